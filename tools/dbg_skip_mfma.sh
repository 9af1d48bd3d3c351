# per-class count times with the matrix-core count pass on and off (MOLAR_HIP_DEBUG_KNOBS build selected by MOLAR_HIP_PLUGIN)
for s in 0 6 3 5 7; do
for m in mfma valu; do
if [ $m = valu ]; then export MOLAR_HIP_NO_MFMA_COUNT=1; else unset MOLAR_HIP_NO_MFMA_COUNT; fi
MOLAR_HIP_DEBUG_SKIP=$s python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_frame']; print('skip=$s $m', 'count %.3f fill %.3f' % (k['pair_count'], k['pair_fill']), d['config']['pairs_per_frame'])"
done; done
