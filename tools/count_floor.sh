# What the count pass of the ordered list costs BEFORE a slot touches its atoms: the pipelined sweep at 0.6 / 0.8 / 1.0 / 1.2 nm with every
# entry class switched off right behind the slot record (MOLAR_HIP_DEBUG_SKIP=15, a -DMOLAR_HIP_DEBUG_KNOBS build as
# molar_amd/_ab/libmolar_hip_dbg.so) against the same build with nothing skipped.  Event times of the count class per frame.
cd /root/repo
export MOLAR_HIP_PLUGIN=/root/repo/molar_amd/_ab/libmolar_hip_dbg.so
for s in 0 15 0 15; do
MOLAR_HIP_DEBUG_SKIP=$s timeout 300 python tools/bench_cutoff_sweep.py 0.6 0.8 1.0 1.2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d['kernel_ms_per_frame']
        print('skip=$s rc', d['cutoff_nm'], 'count %.3f fill %.3f pipelined %.3f' % (k['pair_count'], k['pair_fill'], d['ms_resident_pipelined']))
"
done
