# A/B of library variants on the cutoff sweep (ordered pair list, two frames in flight), alternating on ONE box:
# tools/ab_sweep.sh "0.6 0.8 1.0 1.2" regular molar_amd/_ab/libmolar_hip_X.so ...   -> per variant and cutoff: pipelined ms, count / fill event times
rcs=$1; shift
for i in 1 2; do
for so in "$@"; do
  if [ "$so" = regular ]; then unset MOLAR_HIP_PLUGIN; else export MOLAR_HIP_PLUGIN=$so; fi
  python tools/bench_cutoff_sweep.py $rcs 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['kernel_ms_per_frame']; print('$so'.split('_')[-1][:12].ljust(12), d['cutoff_nm'], 'pipelined %.3f ms' % d['ms_resident_pipelined'], 'count %.3f fill %.3f' % (k['pair_count'], k['pair_fill']), '%.1f Mpairs/ms' % d['mpairs_per_ms_pipelined'])"
done
done
