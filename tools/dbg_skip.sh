# Per-class kernel times: MOLAR_HIP_DEBUG_SKIP switches entry classes off (bit0 plain, bit1 wrapped, bit2 same-cell).
# The knob exists only in builds with -DMOLAR_HIP_DEBUG_KNOBS (release kernels carry no debug branch): build such a
# library first and select it with MOLAR_HIP_PLUGIN, e.g.
#   MOLAR_HIP_EXTRA_FLAGS=-DMOLAR_HIP_DEBUG_KNOBS python -m molar_amd.build && cp molar_amd/libmolar_hip.so /tmp/dbg.so
#   MOLAR_HIP_PLUGIN=/tmp/dbg.so sh tools/dbg_skip.sh
for s in 0 1 2 4 3 5 6; do
MOLAR_HIP_DEBUG_SKIP=$s python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_frame']; print('skip=$s', 'count %.3f fill %.3f' % (k['pair_count'], k['pair_fill']), d['config']['pairs_per_frame'])"
done
