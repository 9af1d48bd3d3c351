# A/B two or more builds of the library on the same box (the pair count is printed too: a variant that is faster because it
# drops work shows up there): MOLAR_HIP_PLUGIN selects the .so (see molar_amd/_lib.py).  REPS (default 3) rounds.
for i in $(seq 1 ${REPS:-3}); do
  for so in "$@"; do
    MOLAR_HIP_PLUGIN=$so python bench.py --steps ${STEPS:-60} --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_frame']; print('$so', round(d['value'],1), 'grid %.3f count %.3f fill %.3f measure %.3f' % (k['grid_build'], k['pair_count'], k['pair_fill'], k['measure']), 'pairs/frame %.1f' % d['config']['pairs_per_frame'])"
  done
done
