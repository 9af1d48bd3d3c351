for i in 1 2 3; do
  for v in early late; do
    if [ $v = early ]; then export MOLAR_HIP_GRID_EARLY=1; else unset MOLAR_HIP_GRID_EARLY; fi
    for s in 1 2; do
      python bench.py --steps 100 --warmup 5 --streams $s --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_frame']; print('grid $v streams $s', round(d['value'],1), 'grid %.3f count %.3f fill %.3f' % (k['grid_build'], k['pair_count'], k['pair_fill']))"
    done
  done
done
