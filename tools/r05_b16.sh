O=gpurun_out/r05/b16
mkdir -p $O
for S in 1 2 3 4; do
python bench.py --workload membrane --steps 200 --warmup 10 --streams $S 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('streams $S', round(d['value'],1), d['ms_per_step'])" >> $O/membrane.txt
done
python bench.py --workload membrane --steps 100 --warmup 10 --verify 2>/dev/null | tail -1 > $O/membrane_verify.json
cat $O/membrane.txt; python -c "
import json; d=json.loads(open('$O/membrane_verify.json').read()); print(d['value'], d['sums_equal_stage_by_stage_single_rank'], d['config']['engine_contexts_per_gpu'])"
