#!/usr/bin/env python
"""The fused histogram between TWO selections (distance_search_double_pbc into Histogram1D: a radial distribution between two
species) on the C4 frame - 250k atoms, triclinic box A, rc 1.2 nm, 1200 bins: the two halves of the frame against each other, and
a 5 % selection against the rest - through molar_hip_search_histogram_frames (groups of 16 frames share their launches) and through
one queued call per frame; the bins of the two forms must be equal.  One JSON line per case.
Usage: python tools/bench_hist_two_sets.py [frames]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from molar_amd import api, build, synth
    build.build_library()
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    n, nbins, rc = 250_000, 1200, 1.2
    box = synth.box_a(n)
    eng = api.Engine(0)
    res = 32
    frames = torch.from_numpy(np.stack([synth.frame(n, box, f) for f in range(res)])).cuda()
    rng = np.random.default_rng(3)
    perm = rng.permutation(n)
    for name, k in (("halves", n // 2), ("5 % against the rest", n // 20)):
        i1 = torch.from_numpy(np.sort(perm[:k]).astype(np.int64)).cuda()
        i2 = torch.from_numpy(np.sort(perm[k:]).astype(np.int64)).cuda()
        out = {}
        for form in ("frames", "single calls"):
            bins = torch.zeros(nbins, dtype=torch.int64, device="cuda")
            for timed in (False, True):
                bins.zero_()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                done = 0
                while done < (F if timed else 64):
                    if form == "frames":
                        eng.search_histogram_frames(api.SEARCH_DOUBLE, rc, 0.0, rc, nbins, frames, idx1=i1, box=box, pbc=7, bins=bins, idx2=i2)
                    else:
                        for f in range(res):
                            eng.search_histogram(api.SEARCH_DOUBLE, rc, 0.0, rc, nbins, frames[f], i1, frames[f], i2, box=box, pbc=7, bins=bins, want_count=False)
                    done += res
                eng.synchronize()
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            out[form] = (done / dt, bins.cpu().numpy() // (done // res))
        same = bool(np.array_equal(out["frames"][1], out["single calls"][1]))
        print(json.dumps({"case": name, "n1": int(k), "n2": int(n - k), "pairs_per_frame": int(out["frames"][1].sum()) // res,
                          "frames_per_s_frames_form": round(out["frames"][0], 1), "frames_per_s_single_calls": round(out["single calls"][0], 1),
                          "bins_equal": same}), flush=True)


if __name__ == "__main__":
    main()
