# kernel trace of the C4 bench (bench.py --workload rdf) for one library: every kernel of two steady-state frames with start / end
# relative to the first hist_kernel's start, and its queue: what overlaps what.   usage: tools/r05_rdf_trace.sh TAG [lib.so]
cd /tmp && export TMPDIR=/tmp
R=/root/repo; T=${1:-tr}; O=$R/gpurun_out/r05; mkdir -p $O; rm -rf $O/trace_$T
[ -n "$2" ] && export MOLAR_HIP_PLUGIN=$R/$2
rocprofv3 --kernel-trace --output-format csv -d $O/trace_$T -- python $R/bench.py --workload rdf --steps 40 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
F=$(find $O/trace_$T -name "*kernel_trace.csv" | head -1)
python - "$F" > $O/${T}_rdf_trace.txt <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
def short(n):
    n=n.replace("mh::pairk::","").replace("mh::","").replace("(anonymous namespace)::","")
    return n.split("(")[0][-40:]
ev=sorted((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),short(r["Kernel_Name"]),r.get("Stream_Id",r.get("Queue_Id","?"))) for r in rows)
h=[i for i,e in enumerate(ev) if "hist_kernel" in e[2]]
a=h[len(h)//2]; b=h[len(h)//2+2]
t0=ev[a][0]
# include side-stream kernels that started up to one frame before
lo=ev[h[len(h)//2-1]][0]
for s,e,n,q in ev:
    if s<lo or s>ev[b][1]: continue
    print(f"{(s-t0)/1e3:9.1f} {(e-t0)/1e3:9.1f} {(e-s)/1e3:8.1f} us  q{q:>3}  {n}")
d=[(ev[j][0]-ev[i][0])/1e3 for i,j in zip(h[5:-1],h[6:])]
d.sort(); print("median frame period us", d[len(d)//2])
PY
cat $O/${T}_rdf_trace.txt
rm -rf $O/trace_$T
