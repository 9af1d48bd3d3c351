# A/B of library variants on the headline bench line (no secondary legs, no CPU baseline), alternating on ONE box:
# tools/ab_headline.sh ROUNDS regular molar_amd/_ab/libmolar_hip_X.so ...  -> frames/s, ms per step, count / fill event times per variant and round
R=$1; shift
cd /root/repo
for i in $(seq $R); do
for so in "$@"; do
  if [ "$so" = regular ]; then unset MOLAR_HIP_PLUGIN; else export MOLAR_HIP_PLUGIN=$so; fi
  timeout 300 python bench.py --no-secondary --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=l['kernel_ms_per_frame']
print('$so'.split('_')[-1][:12].ljust(12), 'frames/s %.1f  ms_per_step %.4f  count %.3f fill %.3f  critical %.3f' % (l['value'], l['ms_per_step'], k['pair_count'], k['pair_fill'], l['critical_path_ms_per_frame']))"
done
done
