#!/usr/bin/env python
"""Turns the raw outputs of tools/profile_bench.sh (gpurun_out/$ROUND) into the committed files of profiles/."""
import collections, csv, glob, json, os, shutil
TAG = os.environ.get("ROUND", "r02")
R = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", TAG)
P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
def newest(pat):
    return max(glob.glob(pat, recursive=True), key=os.path.getmtime)
rows, vals = [], {}
for name in ("fetch", "write"):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(newest(f"{R}/{name}/**/*counter_collection.csv"))):
        short = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mh::pairk::", "").replace("(anonymous namespace)::", "")
        acc[(short, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        rows.append((k, c, sum(v) / len(v), len(v))); vals[(k, c)] = sum(v) / len(v)
with open(f"{P}/{TAG}_pmc_hbm_bytes.csv", "w") as f:
    w = csv.writer(f); w.writerow(["k", "Counter_Name", "mean", "count"]); w.writerows(rows)
fk = "pair_kernel<0, 1, 0>"
fetch, write = vals[(fk, "FETCH_SIZE")], vals[(fk, "WRITE_SIZE")]
traffic = (2 * fetch + write) * 1024
json.dump({"pair_fill_hbm_bytes_per_launch": traffic,
           "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (bench.py --steps 3), KB per launch averaged; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B for wide coalesced reads); WRITE_SIZE taken as reported",
           "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write, "round": TAG}, open(f"{P}/traffic.json", "w"), indent=1)
shutil.copy(newest(f"{R}/stats/**/*kernel_stats.csv"), f"{P}/{TAG}_bench_kernel_stats.csv")
d = json.loads(open(f"{R}/bench.json").read().strip().splitlines()[-1]); d["roofline"]["traffic"] = traffic
open(f"{P}/{TAG}_bench.json", "w").write(json.dumps(d) + "\n")
print(round(d["value"], 1), round(d["ms_per_step"], 3), d["kernel_ms_per_frame"], round(d["roofline"]["frac"], 3), d["cpu_baseline"]["value"])
for r in csv.DictReader(open(f"{P}/{TAG}_bench_kernel_stats.csv")):
    if "pair_kernel" in r["Name"]:
        print(r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e6, "ms")
try:
    shutil.copy(newest(f"{R}/rdf_stats/**/*kernel_stats.csv"), f"{P}/{TAG}_rdf_kernel_stats.csv")
    line = [l for l in open(f"{R}/rdf_bench.json").read().splitlines() if l.startswith("{")][-1]
    open(f"{P}/{TAG}_rdf_bench.json", "w").write(line + "\n")
    dr = json.loads(line)
    print("rdf", round(dr["value"], 1), "frames/s", dr["kernel_ms_per_frame"])
except Exception as exc:
    print("no rdf profile:", exc)
