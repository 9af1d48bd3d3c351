#!/usr/bin/env python
"""Randomised test of the fused histogram over blocks of frames (molar_hip_search_histogram_frames): blocks of 1 ... 40 frames of
random size, box kind (one box for the block, a box per frame with NPT-like jitter, now and then a frame whose box gives another
grid), cutoff, periodicity, index, bin count and range go through the frames form on one context - with queued single-frame
calls and host-bins calls in between - and must leave exactly the bins that one waited molar_hip_search_histogram call per frame
leaves on a second context (the form tools/fuzz_search.py checks against the oracle); every 8th block is checked against the
oracle directly.  Exercises the groups of <= 16 frames, both generations of the group buffers, the four pairs of list counters
and the fall-back to single calls.  A third of the blocks are of kind DOUBLE (two selections of the same frames, overlapping
now and then): both sets' grids per frame.
Usage: python tools/fuzz_hist_frames.py [nblocks] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from molar_amd import api, build
    from oracle.oracle import Oracle
    from tools.fuzz_search import random_box
    nblocks = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    build.build_library()
    eng, ref_eng = api.Engine(0), api.Engine(0)
    o = Oracle("f32")
    rng = np.random.default_rng(seed)
    fails = batched = 0
    for blk in range(nblocks):
        box0 = random_box(rng)
        vol = abs(np.linalg.det(box0.astype(np.float64)))
        n = int(min(max(vol * rng.choice([30.0, 100.0, 100.0, 400.0]), 60), rng.choice([2500, 9000, 30000])))
        F = int(rng.choice([1, 2, 3, 7, 16, 17, 33, 40]))
        rc = float(np.float32(rng.uniform(0.3, 1.1)))
        pbc = int(rng.choice([7, 7, 7, 3, 5, 6, 1]))
        nbins = int(rng.choice([1, 17, 300, 1200, 4000]))
        hmin, hmax = (0.0, rc) if rng.random() < 0.6 else (float(rng.uniform(0, 0.3 * rc)), float(rng.uniform(0.5 * rc, 1.3 * rc)))
        mode = rng.choice(["one", "npt", "jump"])
        boxes = np.repeat(box0[None], F, axis=0).astype(np.float32)
        if mode != "one":
            for f in range(F):
                boxes[f] = (box0 * np.float32(1.0 + 0.003 * rng.normal())).astype(np.float32)
            if mode == "jump" and F > 2:
                boxes[rng.integers(0, F)] *= np.float32(rng.choice([0.8, 1.25]))
        frames = np.empty((F, n, 3), np.float32)
        for f in range(F):
            frames[f] = (rng.random((n, 3)) @ boxes[f].astype(np.float64).T + rng.normal(0, rng.choice([0.0, 0.05, 0.4]), (n, 3))).astype(np.float32)
        idx = None if rng.random() < 0.6 else np.sort(rng.choice(n, max(n // 2, 2), replace=False)).astype(np.uint64)
        two = rng.random() < 0.34
        kind = api.SEARCH_DOUBLE if two else api.SEARCH_SINGLE
        idx2 = None
        if two:       # the second selection: a few atoms up to most of them; None = all atoms (every atom of set 1 is in it too)
            idx2 = None if rng.random() < 0.2 else np.sort(rng.choice(n, max(int(n * rng.choice([0.02, 0.3, 0.8])), 1), replace=False)).astype(np.uint64)
        dframes = torch.from_numpy(frames).cuda()
        didx = None if idx is None else torch.from_numpy(idx.astype(np.int64)).cuda()
        didx2 = None if idx2 is None else torch.from_numpy(idx2.astype(np.int64)).cuda()
        second = dict(frames2=None, idx2=didx2) if two else {}
        got = torch.zeros(nbins, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        box_arg = box0 if mode == "one" else boxes
        tag = f"block {blk}: n {n} F {F} rc {rc:.3f} pbc {pbc} nbins {nbins} boxes {mode} idx {idx is not None} two {two} idx2 {idx2 is not None}"

        def one_call(e, xyz, i1, i2, bx, **kw):
            return e.search_histogram(kind, rc, hmin, hmax, nbins, xyz, i1, xyz if two else None, i2 if two else None, box=bx, pbc=pbc, **kw)
        try:
            reps = int(rng.choice([1, 1, 2]))
            extra = np.zeros(nbins, np.int64)
            for r in range(reps):
                eng.search_histogram_frames(kind, rc, hmin, hmax, nbins, dframes, idx1=didx, box=box_arg, pbc=pbc, bins=got, **second)
                if rng.random() < 0.4:        # a queued single-frame call on the same bins in between
                    k = int(rng.integers(0, F))
                    one_call(eng, dframes[k], didx, didx2, boxes[k], bins=got, want_count=False)
                    hb, _ = one_call(ref_eng, frames[k], idx, idx2, boxes[k])
                    extra += hb.astype(np.int64)
            eng.synchronize()
            want = np.zeros(nbins, np.int64)
            for f in range(F):
                hb, _ = one_call(ref_eng, frames[f], idx, idx2, boxes[f])
                want += hb.astype(np.int64)
            if blk % 8 == 0:
                wo = np.zeros(nbins, np.int64)
                for f in range(F):
                    p = frames[f] if idx is None else frames[f][idx.astype(int)]
                    if two:
                        ref = o.search_double_pbc(rc, p, frames[f] if idx2 is None else frames[f][idx2.astype(int)], o.box_from_matrix(boxes[f]), pbc, nthreads=8)
                    else:
                        ref = o.search_single_pbc(rc, p, o.box_from_matrix(boxes[f]), pbc, nthreads=8)
                    wo += o.histogram_add(hmin, hmax, nbins, ref["d"]).astype(np.int64)
                if not np.array_equal(wo, want):
                    fails += 1
                    print("MISMATCH single calls vs oracle", tag)
            if not np.array_equal(got.cpu().numpy(), reps * want + extra):
                fails += 1
                print("MISMATCH", tag, int(got.sum()), int((reps * want + extra).sum()))
            if F >= 2 and mode != "jump":
                batched += 1
        except Exception as exc:
            fails += 1
            print("ERROR", tag, repr(exc))
    print(f"fuzz_hist_frames: {nblocks} blocks, {batched} of them through the batched form, {fails} failures (seed {seed})")
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
