# same library with and without one environment flag: bash tools/ab_flag.sh FLAG   (alternating runs on one box)
FLAG=$1
for i in 1 2 3; do
  for v in off on; do
    if [ $v = on ]; then export $FLAG=1; else unset $FLAG; fi
    python bench.py --steps 60 --warmup 5 --streams 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_frame']; print('$FLAG=$v', round(d['value'],1), 'grid %.3f count %.3f fill %.3f' % (k['grid_build'], k['pair_count'], k['pair_fill']), 'pairs/frame %.1f' % d['config']['pairs_per_frame'])"
  done
done
unset $FLAG
