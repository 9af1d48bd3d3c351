# as dbg_skip.sh, for the fused histogram workload
for s in ${SKIPS:-0 1 2 4 3 5 6}; do
MOLAR_HIP_DEBUG_SKIP=$s python bench.py --workload rdf --steps 40 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_frame']; print('skip=$s', 'hist %.3f' % k['pair_fill'], d['config']['pairs_per_frame'])"
done
