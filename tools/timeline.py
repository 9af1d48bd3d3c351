"""Per-frame timeline of the headline bench from a rocprofv3 --kernel-trace CSV: which kernels sit on the critical
path (the stream that runs the fill kernel), how long each takes and how large the gaps between them are.
usage: python tools/timeline.py <kernel_trace.csv> [nframes_to_skip]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 5
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "").replace("mh::pairk::", "").replace("mh::", "")
    for a, b in (("pair_kernel<0, 1, 0>", "FILL"), ("pair_kernel<0, 0, 0>", "COUNT"), ("count_task_kernel<0>", "COUNT_T")):
        if a in n: return b
    return n.split("(")[0].split("<")[0][-28:]
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows))
fills = [i for i, e in enumerate(ev) if e[2] == "FILL"]
if len(fills) < skip + 3: raise SystemExit("too few frames")
q = ev[fills[skip]][3]
main = [e for e in ev if e[3] == q]
fi = [i for i, e in enumerate(main) if e[2] == "FILL"]
busy = collections.defaultdict(float); gaps = collections.defaultdict(float); n = 0
# frames whose period is far above the median are pauses of the host (phases of the benchmark, its self-check), not the pipeline
periods = sorted(main[b][0] - main[a][0] for a, b in zip(fi[skip:-1], fi[skip + 1:]))
med = periods[len(periods) // 2]
kept_ns = 0
for a, b in zip(fi[skip:-1], fi[skip + 1:-0 or None]):
    if main[b][0] - main[a][0] > 2 * med: continue
    kept_ns += main[b][0] - main[a][0]
    seg = main[a:b + 1]
    for x, y in zip(seg[:-1], seg[1:]):
        busy[y[2]] += (y[1] - y[0]) / 1e3
        gaps[x[2] + " -> " + y[2]] += max(0, y[0] - x[1]) / 1e3
    n += 1
print(f"{n} frames on queue {q}: per-frame busy us / gap us")
tb = tg = 0
for k, v in busy.items(): print(f"  busy {k:30s} {v / n:8.1f}"); tb += v / n
for k, v in gaps.items(): print(f"  gap  {k:50s} {v / n:8.1f}"); tg += v / n
print(f"  total busy {tb:.1f} us, gaps {tg:.1f} us, period {kept_ns / 1e3 / max(n, 1):.1f} us (frames with more than twice the median period left out)")
