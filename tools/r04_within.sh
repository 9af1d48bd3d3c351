# `within` on the reference's benchmark shapes -> gpurun_out/r04/within.jsonl (profiles/r04_within.jsonl)
R=/root/repo; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
python tools/bench_within.py > $O/within.jsonl 2> $O/within.err; tail -3 $O/within.err
python - <<PY
import json
for l in open("$O/within.jsonl"):
    d=json.loads(l); print(d["workload"][:62].ljust(62), "found %7d set %8.3f ms stream %9.3f ms cpu %9.2f ms  x%.1f / x%.1f" % (d["found"], d["ms_set"], d["ms_stream_plus_unique"], d.get("ms_cpu_restatement",0), d["speedup_set_over_stream"], d.get("speedup_set_over_cpu",0)))
PY
