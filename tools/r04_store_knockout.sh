export MOLAR_HIP_PLUGIN=molar_amd/_ab/libmolar_hip_dbg.so
for s in 0 16 32 0 16 32; do
MOLAR_HIP_DEBUG_SKIP=$s python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pipeline --serial-measure --preheat 0.5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_frame']; print('skip=$s', 'count %.3f fill %.3f' % (k['pair_count'], k['pair_fill']))"
done
