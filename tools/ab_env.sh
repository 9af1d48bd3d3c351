# same library, different values of one environment variable: bash tools/ab_env.sh VAR v1 v2 ...
VAR=$1; shift
for i in 1 2; do
  for v in "$@"; do
    env $VAR=$v python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_frame']; print('$VAR=$v', round(d['value'],1), 'grid %.3f count %.3f fill %.3f' % (k['grid_build'], k['pair_count'], k['pair_fill']))"
  done
done
