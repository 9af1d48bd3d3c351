# A/B two or more builds with one frame at a time and the fit on the same stream (the kernels' stand-alone times):
#   bash tools/ab_alone.sh molar_amd/_ab/libmolar_hip_A.so molar_amd/libmolar_hip.so     (REPS rounds, default 3)
for i in $(seq 1 ${REPS:-3}); do for so in "$@"; do
MOLAR_HIP_PLUGIN=$so python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-pipeline --serial-measure --preheat 0.5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_frame']; print('$so', 'count %.4f fill %.4f grid %.4f' % (k['pair_count'], k['pair_fill'], k['grid_build']))"
done; done
