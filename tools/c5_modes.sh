cd /root/repo
for i in 1 2 3 4; do for M in thread defer; do for S in 1 4; do
BENCH_C5_SUMS=$M timeout 300 python bench.py --workload membrane --streams $S --steps 512 --warmup 32 2>/dev/null | python -c "import json,sys;l=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$M','streams',$S,round(l['value'],1),l.get('verify'))"
done; done; done
