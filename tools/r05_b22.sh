O=gpurun_out/r05/b22
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_search.py tests/test_gpu_within_set.py -x -q -k "pairs_plane or within" 2>&1 | tail -4 > $O/tests.txt
python bench.py --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench.json
python tools/bench_within.py --no-cpu 2>/dev/null > $O/within.jsonl
python tools/bench_vdw.py 2>/dev/null > $O/vdw.jsonl
cat $O/tests.txt; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05/b22/bench.json').read()); print(d['value'], d['kernel_ms_per_frame'], d['pairs_only'], d['verified_against_single_context'])
for l in open('gpurun_out/r05/b22/within.jsonl'):
    d=json.loads(l); print(d['workload'][:60].ljust(60), 'set %.3f hold %.3f stream %.3f' % (d['ms_set'], d['ms_set_grid_held'], d['ms_stream_plus_unique']))
for l in open('gpurun_out/r05/b22/vdw.jsonl'):
    d=json.loads(l); print(d['workload'][:70], 'gpu %.3f resident-sel %.3f cpu %.2f' % (d['ms_gpu_count_fill_to_host'], d['ms_gpu_selections_resident'], d['ms_cpu_restatement']))
PY
