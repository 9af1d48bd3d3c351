for i in 1 2; do
for so in molar_amd/libmolar_hip.so NOTASK; do
  if [ $so = NOTASK ]; then export MOLAR_HIP_NO_COUNT_TASK=1; lib=molar_amd/libmolar_hip.so; else unset MOLAR_HIP_NO_COUNT_TASK; lib=$so; fi
  MOLAR_HIP_PLUGIN=$lib python bench.py --steps 60 --warmup 5 --streams 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_frame']; print('$so', round(d['value'],1), 'grid %.3f count %.3f fill %.3f' % (k['grid_build'], k['pair_count'], k['pair_fill']), 'pairs/frame %.1f' % d['config']['pairs_per_frame'])"
done
done
