# Timeline of the C5 chain on ONE context: kernel + copy trace of 96 frames, then per-frame busy time and the idle gaps between consecutive
# GPU operations grouped by (operation before, operation after).  usage: tools/r06_c5_timeline.sh TAG -> gpurun_out/r06/TAG_c5_timeline.txt
R=/root/repo; T=${1:-t}; O=$R/gpurun_out/r06; mkdir -p $O; D=/tmp/c5tl_$T; rm -rf $D
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $D -- python $R/bench.py --workload membrane --streams 1 --steps 96 --warmup 16 > $O/${T}_c5_tl.json 2> $O/${T}_c5_tl.err </dev/null
K=$(find $D -name "*kernel_trace.csv" | head -1); C=$(find $D -name "*memory_copy_trace.csv" | head -1)
python - "$K" "$C" <<'PY' > $O/${T}_c5_timeline.txt 2>&1
import csv, sys, re, collections
ops = []
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r'\(anonymous namespace\)::|mh::pairk::|mh::|void ', '', r['Kernel_Name']).split('(')[0][:44]
    ops.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), n, r.get('Queue_Id', r.get('Stream_Id', '0'))))
if len(sys.argv) > 2 and sys.argv[2]:
    for r in csv.DictReader(open(sys.argv[2])):
        ops.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY_' + r['Direction'][-14:], 'copy'))
ops.sort()
# steady state: frames are delimited by k_split_markers (first kernel of B)
marks = [i for i, o in enumerate(ops) if o[2].startswith('k_split_markers')]
lo, hi = marks[len(marks) // 3], marks[-4]
nfr = sum(1 for m in marks if lo <= m < hi)
seg = ops[lo:hi]
span = seg[-1][0] - seg[0][0]
busy = 0; end = seg[0][0]; gaps = collections.defaultdict(lambda: [0, 0])
per = collections.defaultdict(lambda: [0, 0])
prev = None
for s, e, n, q in seg:
    per[n][0] += 1; per[n][1] += e - s
    if s > end:
        if prev: g = gaps[(prev, n)]; g[0] += 1; g[1] += s - end
        busy += e - s
    else:
        busy += max(0, e - max(s, end))
    if e > end: end = e; prev = n
print(f"frames {nfr}  period {span / nfr / 1e3:.1f} us  busy {busy / nfr / 1e3:.1f} us  idle {(span - busy) / nfr / 1e3:.1f} us")
print(f"operations per frame {sum(c for c, t in per.values()) / nfr:.1f} (kernels, fills and copies on all queues), their time {sum(t for c, t in per.values()) / nfr / 1e3:.1f} us")
print("-- operations per frame: count x average us = us per frame")
for n, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{n:46s} {c / nfr:6.2f} x {t / c / 1e3:7.2f} = {t / nfr / 1e3:7.2f}")
print("-- idle gaps per frame (us) by (before -> after)")
for (a, b), (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f"{a:40s} -> {b:40s} {c / nfr:6.2f} x {t / c / 1e3:7.2f} = {t / nfr / 1e3:7.2f}")
PY
rm -rf $D
cat $O/${T}_c5_timeline.txt </dev/null
