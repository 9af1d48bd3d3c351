# kernel trace of the C4 bench in its frames form (bench.py --workload rdf: molar_hip_search_histogram_frames, groups of 8 frames per
# launch): every kernel of two steady-state groups with start / end relative to the first hist_kernel's start, and its queue - what
# overlaps what; then the groups' period per frame.   usage: tools/r06_rdf_trace.sh TAG [extra bench flags]
cd /tmp && export TMPDIR=/tmp
R=/root/repo; T=${1:-tr}; shift; O=$R/gpurun_out/r06; mkdir -p $O; rm -rf $O/trace_$T
rocprofv3 --kernel-trace --output-format csv -d $O/trace_$T -- python $R/bench.py --workload rdf --steps 128 --warmup 16 --no-cpu-baseline "$@" > /dev/null 2>&1
F=$(find $O/trace_$T -name "*kernel_trace.csv" | head -1)
python - "$F" > $O/${T}_rdf_trace.txt <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
def short(n):
    n=n.replace("mh::pairk::","").replace("mh::","").replace("(anonymous namespace)::","")
    return n.split("(")[0][-40:]
ev=sorted((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),short(r["Kernel_Name"]),r.get("Stream_Id",r.get("Queue_Id","?"))) for r in rows)
h=[i for i,e in enumerate(ev) if "hist_kernel" in e[2]]
dur=sorted((ev[i][1]-ev[i][0])/1e3 for i in h)
big=[i for i in h if (ev[i][1]-ev[i][0])/1e3 > 0.5*dur[-1]]      # the full groups of the timed region
a=big[len(big)//2]; b=big[len(big)//2+2] if len(big)//2+2 < len(big) else big[-1]
t0=ev[a][0]
lo=ev[big[len(big)//2-1]][0]
for s,e,n,q in ev:
    if s<lo or s>ev[b][1]: continue
    print(f"{(s-t0)/1e3:9.1f} {(e-t0)/1e3:9.1f} {(e-s)/1e3:8.1f} us  q{q:>3}  {n}")
d=[(ev[j][0]-ev[i][0])/1e3 for i,j in zip(big[2:-1],big[3:])]
d.sort()
if d: print("median group period us", d[len(d)//2], " hist_kernel median us", dur[len(dur)//2], " launches", len(h))
PY
cat $O/${T}_rdf_trace.txt
rm -rf $O/trace_$T
