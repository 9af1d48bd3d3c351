# the C5 leg over the number of engine contexts, alternating (one box)
cd /root/repo
for i in 1 2; do for S in 2 4 6 8; do
timeout 300 python bench.py --workload membrane --streams $S --steps 768 --warmup 32 2>/dev/null | python -c "import json,sys;l=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('streams',$S,round(l['value'],1))"
done; done
