# Per-class kernel times: MOLAR_HIP_DEBUG_SKIP switches entry classes off (bit0 plain, bit1 wrapped, bit2 same-cell, bit3
# triclinic corner entries).  The knob exists only in builds with -DMOLAR_HIP_DEBUG_KNOBS (release kernels carry no debug
# branch): build such a library first and select it with MOLAR_HIP_PLUGIN, e.g.
#   MOLAR_HIP_EXTRA_FLAGS=-DMOLAR_HIP_DEBUG_KNOBS python -m molar_amd.build && cp molar_amd/libmolar_hip.so molar_amd/_ab/libmolar_hip_dbg.so
#   MOLAR_HIP_PLUGIN=molar_amd/_ab/libmolar_hip_dbg.so sh tools/dbg_skip.sh
# One frame at a time, fit behind the search on the same stream: the event times are those of the kernels alone.
for s in 0 15 14 13 11 7; do
MOLAR_HIP_DEBUG_SKIP=$s python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pipeline --serial-measure --preheat 0.5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_frame']; print('skip=$s', 'count %.3f fill %.3f' % (k['pair_count'], k['pair_fill']), d['config']['pairs_per_frame'])"
done
