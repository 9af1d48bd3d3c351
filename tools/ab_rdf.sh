# A/B of library variants on the C4 bench (frames form), alternating on ONE box: tools/ab_rdf.sh lib1.so lib2.so ...  ("regular" = the regular build)
for i in 1 2 3; do
for so in "$@"; do
  if [ "$so" = regular ]; then unset MOLAR_HIP_PLUGIN; else export MOLAR_HIP_PLUGIN=$so; fi
  python bench.py --workload rdf --steps 1024 --warmup 64 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_frame']; print('$so', round(d['value'],1), 'hist kernel per frame %.4f ms' % k['pair_fill'], 'pairs/frame %.1f' % d['config']['pairs_per_frame'])"
done
done
