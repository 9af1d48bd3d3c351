#!/usr/bin/env python
"""Randomised differential test of the f64 drivers (search_f64.hip: grid by stable radix sort, plan, bounding-box row pruning,
adjacent-image classification of entries across the boundary, LDS output queue) against the f64 build of the CPU oracle: random
boxes (orthorhombic, sheared, strongly triclinic, flat, tiny, large with >= 4 cells per dimension), cutoffs, densities,
periodicity masks, selections, all four kinds, coordinates from host arrays or resident in HBM, atoms outside the cell, pairs
planted at the cutoff edge across the periodic boundary.  Every case must be bit-identical (ids, order, distances).
Usage: python tools/fuzz_search_f64.py [ncases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def random_box(rng):
    kind = rng.integers(0, 7)
    L = rng.uniform(1.5, 6.0, 3)
    if kind == 0:
        m = np.diag(L)
    elif kind == 1:
        m = np.diag(L); m[0, 2] = -rng.uniform(0, 0.3) * L[0]; m[1, 2] = -rng.uniform(0, 0.3) * L[1]
    elif kind == 2:
        m = np.diag(L); m[0, 1] = rng.uniform(-0.5, 0.5) * L[0]; m[0, 2] = rng.uniform(-0.5, 0.5) * L[0]; m[1, 2] = rng.uniform(-0.5, 0.5) * L[1]
    elif kind == 3:
        m = np.diag(L) + rng.uniform(-0.3, 0.3, (3, 3)) * L.min()
    elif kind == 4:
        m = np.diag([L[0] * 2, L[1] * 2, rng.uniform(0.6, 1.2)])
    elif kind == 5:
        m = np.diag(rng.uniform(0.7, 1.6, 3))
    else:                                             # roomy: several cells per dimension even at large cutoffs
        m = np.diag(rng.uniform(5.0, 9.0, 3)); m[0, 1] = rng.uniform(-0.3, 0.3) * m[0, 0]; m[1, 2] = rng.uniform(-0.3, 0.3) * m[1, 1]
    return m.astype(np.float64)


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    import torch
    from molar_amd import api, build
    from oracle.oracle import Oracle
    build.build_library()
    eng = api.Engine(0)
    o = Oracle("f64")
    rng = np.random.default_rng(seed)
    fails = 0
    for case in range(ncases):
        box = random_box(rng)
        vol = abs(np.linalg.det(box))
        dens = rng.choice([20.0, 60.0, 100.0, 300.0])
        n = int(min(max(vol * dens, 30), 8000))
        pos = rng.random((n, 3)) @ box.T + rng.normal(0, rng.choice([0.0, 0.05, 0.5]), (n, 3))
        rc = float(rng.uniform(0.25, 1.3))
        if rng.random() < 0.4 and n >= 200:
            # pairs planted at rc * (1 +- 1e-16 .. 1e-8) around atoms next to the faces of the cell
            k = n // 5
            frac = rng.random((k, 3))
            frac[np.arange(k), rng.integers(0, 3, k)] = rng.choice([0.0, 1.0], k) + rng.normal(0, 0.01, k)
            pa = frac @ box.T
            u = rng.normal(size=(k, 3)); u /= np.linalg.norm(u, axis=1)[:, None]
            e = 10.0 ** rng.uniform(-16.0, -8.0, k) * rng.choice([-1.0, 1.0], k)
            pos[:k] = pa
            pos[k:2 * k] = pa + rc * (1.0 + e)[:, None] * u
        if rng.random() < 0.1:
            pos[rng.integers(0, n)] = np.nan                      # an atom that pairs with nothing
        pbc = int(rng.choice([7, 7, 7, 0, 1, 2, 3, 4, 5, 6]))
        kind = int(rng.choice([0, 0, 0, 1, 1, 2, 3]))
        resident = rng.random() < 0.5
        try:
            ob = o.box_from_matrix(box)
        except Exception:
            continue
        tag = f"case {case}: kind {kind} n {n} rc {rc!r} pbc {pbc} resident {resident} box {box.tolist()}"
        xyz = torch.from_numpy(pos).cuda() if resident else pos

        def dev(a):
            if not resident or a is None:
                return a
            return torch.from_numpy(a.astype(np.int64) if a.dtype == np.uint64 else a).cuda()
        try:
            kw = dict(box=box, pbc=pbc) if pbc else {}
            if kind == 0:
                idx = None if rng.random() < 0.5 else np.sort(rng.choice(n, max(n // 2, 2), replace=False)).astype(np.uint64)
                p = pos if idx is None else pos[idx.astype(int)]
                ref = o.search_single_pbc(rc, p, ob, pbc, ids=idx, nthreads=4) if pbc else o.search_single(rc, p, ids=idx, nthreads=4)
                got = eng.search_f64(api.SEARCH_SINGLE, rc, xyz, dev(idx), **kw)
            elif kind in (1, 3):
                perm = rng.permutation(n)
                i1 = np.sort(perm[: n // 3]).astype(np.uint64); i2 = np.sort(perm[n // 3:]).astype(np.uint64)
                p1, p2 = pos[i1.astype(int)], pos[i2.astype(int)]
                if kind == 1:
                    ref = o.search_double_pbc(rc, p1, p2, ob, pbc, ids1=i1, ids2=i2, nthreads=4) if pbc else o.search_double(rc, p1, p2, ids1=i1, ids2=i2, nthreads=4)
                    got = eng.search_f64(api.SEARCH_DOUBLE, rc, xyz, dev(i1), xyz, dev(i2), **kw)
                else:
                    v1 = rng.uniform(0.1, 0.25, len(i1)); v2 = rng.uniform(0.1, 0.25, len(i2))
                    ref = o.search_double_vdw_pbc(p1, p2, v1, v2, ob, pbc, nthreads=4) if pbc else o.search_double_vdw(p1, p2, v1, v2, nthreads=4)
                    got = eng.search_f64(api.SEARCH_DOUBLE_VDW, None, xyz, dev(i1), xyz, dev(i2), vdw1=dev(v1), vdw2=dev(v2), **kw)
            else:
                i1 = np.arange(n, dtype=np.uint64); i2 = np.sort(rng.choice(n, max(n // 20, 1), replace=False)).astype(np.uint64)
                p2 = pos[i2.astype(int)]
                if pbc:
                    ref = o.search_within_pbc(rc, pos, p2, ob, pbc, i1, i2, nthreads=4)
                    ids = eng.search_f64(api.SEARCH_WITHIN, rc, xyz, dev(i1), xyz, dev(i2), box=box, pbc=pbc)
                else:
                    fin = pos[np.isfinite(pos).all(1)]
                    lo = np.minimum(fin.min(0), 0.0) - (rc + 2.220446049250313e-16); up = np.maximum(fin.max(0), 0.0) + (rc + 2.220446049250313e-16)
                    ref = o.search_within(rc, pos, p2, lo, up, i1, i2, nthreads=4)
                    ids = eng.search_f64(api.SEARCH_WITHIN, rc, xyz, dev(i1), xyz, dev(i2), lower=lo, upper=up)
                if not np.array_equal(ids, ref["i"]):
                    fails += 1; print("MISMATCH", tag, len(ids), len(ref["i"]))
                continue
            i, j, d = got
            ok = len(i) == len(ref["i"]) and np.array_equal(i, ref["i"]) and np.array_equal(j, ref["j"]) and np.array_equal(d, ref["d"])
            if not ok:
                fails += 1; print("MISMATCH", tag, len(i), len(ref["i"]))
        except Exception as exc:      # an engine error on a case the oracle accepts is a failure too
            fails += 1; print("ERROR", tag, repr(exc))
    print(f"{ncases} cases, {fails} failures")
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
