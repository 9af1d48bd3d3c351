# count pass on the matrix cores against the vector-ALU count pass, same box, alternating (MOLAR_HIP_NO_MFMA_COUNT is read at create)
for i in 1 2 3; do
  python bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_frame']; print('mfma', round(d['value'],1), 'grid %.3f count %.3f fill %.3f' % (k['grid_build'], k['pair_count'], k['pair_fill']), d['config']['pairs_per_frame'])"
  MOLAR_HIP_NO_MFMA_COUNT=1 python bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_frame']; print('valu', round(d['value'],1), 'grid %.3f count %.3f fill %.3f' % (k['grid_build'], k['pair_count'], k['pair_fill']), d['config']['pairs_per_frame'])"
done
