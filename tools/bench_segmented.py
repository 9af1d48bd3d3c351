"""Dense against segmented layout of the resident single search on the headline frame (1M atoms, triclinic box A, rc 1.2 nm),
pipelined begin / end on one context, no fit beside it:

    python tools/bench_segmented.py [--frames 60] [--natoms 1000000] [--pairs-only]

Prints ms per frame of both layouts and the number of kernel launches of the segmented one that were repeats."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from molar_amd import api, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=60)
ap.add_argument("--natoms", type=int, default=1_000_000)
ap.add_argument("--cutoff", type=float, default=1.2)
ap.add_argument("--pairs-only", action="store_true")
ap.add_argument("--ortho", action="store_true", help="orthorhombic box of the same volume (no triclinic corner entries)")
ap.add_argument("--only", choices=["dense", "segmented"], default=None)
args = ap.parse_args()
n, rc = args.natoms, args.cutoff
box = synth.box_a(n)
if args.ortho:
    box = np.diag(np.diag(box)).astype(np.float32)
dev = torch.device("cuda:0")
frames = [torch.from_numpy(synth.frame(n, box, f)).to(dev) for f in range(4)]
eng = api.Engine(0)
if args.pairs_only:
    eng.search_resident_planes(False)
descs = [eng.make_search_desc(api.SEARCH_SINGLE, rc, frames[0], box=box, pbc=7) for _ in range(2)]


def run(k):
    counts = []
    prev = None
    for f in range(k):
        d = descs[f & 1][0]
        d.xyz1 = frames[f % 4].data_ptr()
        t = eng.search_resident_begin(d)
        if prev is not None:
            counts.append(eng.search_resident_end(prev)[0])
        prev = t
    counts.append(eng.search_resident_end(prev)[0])
    return counts


out = {}
for name, seg in (("dense", False), ("segmented", True)):
    if args.only and args.only != name:
        continue
    eng.search_resident_layout(seg)
    run(8)
    eng.synchronize()
    t0 = time.perf_counter()
    c = run(args.frames)
    eng.synchronize()
    dt = time.perf_counter() - t0
    out[name] = {"ms_per_frame": dt / args.frames * 1e3, "frames_per_s": args.frames / dt, "pairs_frame0": c[0]}
    if seg:
        b, nn, nseg, span = eng.search_segments(0)
        out[name]["span_over_results"] = span / max(c[-2 if args.frames > 1 else -1], 1)
# how much does an entry's count move between consecutive frames?  (what the capacities' margin has to cover)
eng.search_resident_layout(True)
cs = []
for f in range(3):
    d = descs[0][0]
    d.xyz1 = frames[f].data_ptr()
    cnt, p, dd = eng.search_resident_desc(d)
    b, nn, nseg, span = eng.search_segments(0)
    cs.append(api.device_view(nn, (nseg,), torch.int32).to(torch.int64).cpu().numpy().copy())
growth = []
for a, b in zip(cs[:-1], cs[1:]):
    m = a > 0
    growth.append((b[m] - a[m]) / (a[m] + 64.0))
z = np.concatenate([(b[a > 0] - a[a > 0]) / np.sqrt(a[a > 0].astype(np.float64)) for a, b in zip(cs[:-1], cs[1:])])
out["count_growth_in_sqrt_units"] = {"p50_abs": float(np.percentile(np.abs(z), 50)), "p99": float(np.percentile(z, 99)), "p99.99": float(np.percentile(z, 99.99)), "max": float(z.max())}
for lo, hi in ((1, 100), (100, 1000), (1000, 4000), (4000, 10**9)):
    sel = np.concatenate([((a >= lo) & (a < hi)) [a > 0] for a in cs[:-1]])
    if sel.any():
        out["count_growth_in_sqrt_units"][f"max_for_counts_{lo}_{hi}"] = float(z[sel].max())
g = np.concatenate(growth)
out["count_growth_between_frames"] = {"entries": int(len(g)), "p50": float(np.percentile(np.abs(g), 50)), "p99": float(np.percentile(g, 99)),
                                      "p99.99": float(np.percentile(g, 99.99)), "max": float(g.max()),
                                      "share_above_1/8": float((g > 0.125).mean()), "share_above_1/4": float((g > 0.25).mean()),
                                      "new_entries_max": int(max((b[a == 0]).max() if (a == 0).any() else 0 for a, b in zip(cs[:-1], cs[1:])))}
print(json.dumps(out))
