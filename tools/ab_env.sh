# A/B of environment knobs on one library, same box: tools/ab_env.sh "VAR=1" ["VAR2=1" ...]; "" = no knob.  REPS rounds (default 3).
for i in $(seq 1 ${REPS:-3}); do
  for kv in "" "$@"; do
    env $kv python bench.py --steps ${STEPS:-60} --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_frame']; print('[$kv]', round(d['value'],1), 'grid %.3f count %.3f scan %.3f fill %.3f measure %.3f' % (k['grid_build'], k['pair_count'], k['offset_scan'], k['pair_fill'], k['measure']))"
  done
done
