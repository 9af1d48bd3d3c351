#!/usr/bin/env python
"""Randomised test of the kernel choice by occupied cells (molar_hip_search_cell_kernels): inhomogeneous frames - slabs, blobs, a few
dense clusters - in periodic boxes of random shape, a handful of frames of the SAME shape per case through one context (synchronous
resident searches, then pipelined ones), so that the first frames run on the average's choice and later ones on the occupied cells'.
Every frame's ordered list must equal the oracle's bit for bit whichever kernels ran; the case also reports whether a switch happened.
Usage: python tools/fuzz_slab.py [ncases] [seed]"""
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
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    build.build_library()
    o = Oracle("f32")
    rng = np.random.default_rng(seed)
    eng = api.Engine(0)
    fails = switched = 0

    class Dev:
        def __init__(self, ptr, n, typestr):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}
    for case in range(ncases):
        box = random_box(rng) * np.float32(rng.choice([1.0, 2.0, 3.0]))
        vol = abs(np.linalg.det(box.astype(np.float64)))
        shape = rng.choice(["slab", "blob", "clusters"])
        frac = float(rng.choice([0.03, 0.08, 0.2]))                # share of the box that holds the atoms
        n = int(min(max(vol * frac * rng.choice([60.0, 100.0]), 200), 60000))
        rc = float(np.float32(rng.uniform(0.4, 1.3)))
        pbc = int(rng.choice([7, 7, 3, 5]))
        two = rng.random() < 0.3
        nfr = int(rng.integers(3, 6))

        def frame():
            u = rng.random((n, 3))
            if shape == "slab":
                u[:, 2] = 0.4 + frac * u[:, 2]
            elif shape == "blob":
                u = 0.5 + (frac ** (1 / 3)) * (u - 0.5)
            else:
                c = rng.random((4, 3))
                u = c[rng.integers(0, 4, n)] + (frac / 4) ** (1 / 3) * (u - 0.5)
            return (u @ box.astype(np.float64).T).astype(np.float32)
        frames = [frame() for _ in range(nfr)]
        i1 = np.sort(rng.choice(n, max(n // 2, 2), replace=False)).astype(np.uint64) if two else None
        i2 = np.sort(rng.choice(n, max(n // 3, 2), replace=False)).astype(np.uint64) if two else None
        ob = o.box_from_matrix(box)
        tag = f"case {case}: {shape} n {n} rc {rc:.3f} pbc {pbc} two {two} frames {nfr}"
        try:
            refs = []
            for p in frames:
                refs.append(o.search_double_pbc(rc, p[i1.astype(int)], p[i2.astype(int)], ob, pbc, ids1=i1, ids2=i2, nthreads=8) if two
                            else o.search_single_pbc(rc, p, ob, pbc, nthreads=8))
            lanes = []
            for p, ref in zip(frames, refs):
                if two:
                    cnt, _, _ = eng.search_resident(api.SEARCH_DOUBLE, rc, p, i1, p, i2, box=box, pbc=pbc)
                else:
                    cnt, _, _ = eng.search_resident(api.SEARCH_SINGLE, rc, p, box=box, pbc=pbc)
                pr, d = eng.search_fill(cnt)
                lanes.append(eng.search_cell_kernels()[0])
                if not (cnt == len(ref["i"]) and np.array_equal(pr[:, 0], ref["i"]) and np.array_equal(pr[:, 1], ref["j"]) and np.array_equal(d, ref["d"])):
                    fails += 1
                    print("MISMATCH (synchronous)", tag, lanes)
                    break
            if len(set(lanes)) > 1:
                switched += 1
            if not two:            # the same frames pipelined, two in flight, on a context that has not seen the shape
                e2 = api.Engine(0)
                dev = [torch.from_numpy(f).cuda() for f in frames]
                descs = [e2.make_search_desc(api.SEARCH_SINGLE, rc, f, box=box, pbc=pbc) for f in dev]
                prev, k_prev = None, -1
                for k in range(nfr + 1):
                    t = e2.search_resident_begin(descs[k][0]) if k < nfr else None
                    if prev is not None:
                        cnt, pp, dp = e2.search_resident_end(prev)
                        ref = refs[k_prev]
                        ok = cnt == len(ref["i"])
                        if ok and cnt:
                            gp = torch.as_tensor(Dev(pp, cnt * 2, "<i4"), device="cuda").cpu().numpy().view(np.uint32).reshape(-1, 2)
                            gd = torch.as_tensor(Dev(dp, cnt, "<f4"), device="cuda").cpu().numpy()
                            ok = np.array_equal(gp[:, 0], ref["i"]) and np.array_equal(gp[:, 1], ref["j"]) and np.array_equal(gd, ref["d"])
                        if not ok:
                            fails += 1
                            print("MISMATCH (pipelined)", tag, "frame", k_prev)
                            break
                    prev, k_prev = t, k
                del e2
        except Exception as exc:
            fails += 1
            print("ERROR", tag, repr(exc))
    print(f"fuzz_slab: {ncases} cases, {switched} of them changed kernels between frames, {fails} failures (seed {seed})")
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
