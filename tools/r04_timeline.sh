# kernel trace of the pipelined headline bench -> tools/timeline.py summary (gpurun_out/r04/<tag>_timeline.txt)
cd /tmp && export TMPDIR=/tmp
R=/root/repo; T=${1:-tl}; O=$R/gpurun_out/r04; mkdir -p $O; rm -rf $O/trace_$T
rocprofv3 --kernel-trace --output-format csv -d $O/trace_$T -- python $R/bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-secondary > /dev/null 2>&1
F=$(find $O/trace_$T -name "*kernel_trace.csv" | head -1)
python $R/tools/timeline.py $F > $O/${T}_timeline.txt 2>&1
cat $O/${T}_timeline.txt
python - "$F" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
d=collections.defaultdict(list)
for r in rows: d[r['Kernel_Name'].split('(')[0][-60:]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1]))[:24]: print(k.ljust(62), len(v), '%9.1f us avg' % (sum(v)/len(v)))
PY
rm -rf $O/trace_$T
