O=gpurun_out/r05/b21
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_search.py -x -q -k "vdw or sparse or large_cells or giant" 2>&1 | tail -4 > $O/tests.txt
python tools/bench_vdw.py 2>/dev/null > $O/vdw.jsonl
cat $O/tests.txt; python - <<'PY'
import json
for l in open('gpurun_out/r05/b21/vdw.jsonl'):
    d=json.loads(l); print(d['workload'][:70], 'gpu %.3f cpu %.2f (t=%s) same %s' % (d['ms_gpu_count_fill_to_host'], d['ms_cpu_restatement'], d.get('cpu_threads_best'), d['identical_to_cpu']))
PY
