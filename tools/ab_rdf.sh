for i in 1 2; do
for so in "$@"; do
  MOLAR_HIP_PLUGIN=$so python bench.py --workload rdf --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_frame']; print('$so', round(d['value'],1), 'grid %.3f hist %.3f' % (k['grid_build'], k['pair_fill']), 'pairs/frame %.1f' % d['config']['pairs_per_frame'])"
done
done
