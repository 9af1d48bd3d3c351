O=gpurun_out/r05/b14
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_within_set.py -x -q 2>&1 | tail -8 > $O/tests.txt
timeout 900 python -m pytest tests/test_gpu_full_size.py -x -q -k within 2>&1 | tail -3 >> $O/tests.txt
python tools/bench_within.py > $O/within.jsonl 2> $O/within.err
python tools/bench_vdw.py > $O/vdw.jsonl 2> $O/vdw.err
cat $O/tests.txt; python - <<'PY'
import json
for l in open('gpurun_out/r05/b14/within.jsonl'):
    d=json.loads(l); print(d['workload'][:60].ljust(60), 'set %.3f hold %.3f stream %.3f cpu %.2f (t=%s)' % (d['ms_set'], d['ms_set_grid_held'], d['ms_stream_plus_unique'], d.get('ms_cpu_restatement',0), d.get('cpu_threads_best')))
for l in open('gpurun_out/r05/b14/vdw.jsonl'):
    d=json.loads(l); print(d['workload'][:70], 'gpu %.3f cpu %.2f (t=%s) same %s' % (d['ms_gpu_count_fill_to_host'], d['ms_cpu_restatement'], d.get('cpu_threads_best'), d['identical_to_cpu']))
PY
tail -n 3 $O/within.err $O/vdw.err
