O=gpurun_out/r05/b18
mkdir -p $O
for i in 1 2; do python bench.py --workload rdf --steps 200 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['ms_per_step'], d['kernel_ms_per_frame'])" >> $O/rdf.txt; done
python bench.py --workload rdf --steps 200 --warmup 10 --verify 2>/dev/null | tail -1 > $O/rdf_verify.json
cat $O/rdf.txt; python -c "
import json; d=json.loads(open('$O/rdf_verify.json').read()); print(d['value'], d['reduced_bins_equal_single_rank'], d['roofline']['frac'])"
