#!/usr/bin/env python
"""Randomised differential test of the HIP search against the CPU oracle: random boxes (orthorhombic, sheared,
strongly triclinic, flat, tiny), cutoffs, densities, periodicity masks, selections and search kinds, through both the
count/fill and the resident entry.  Every case must be bit-identical (ids, order, distances).
Usage: python tools/fuzz_search.py [ncases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def random_box(rng):
    kind = rng.integers(0, 6)
    L = rng.uniform(1.5, 6.0, 3)
    if kind == 0:
        m = np.diag(L)
    elif kind == 1:                                   # benign shear (reference grid complete)
        m = np.diag(L); m[0, 2] = -rng.uniform(0, 0.3) * L[0]; m[1, 2] = -rng.uniform(0, 0.3) * L[1]
    elif kind == 2:                                   # GROMACS-style lower-triangular rows = upper-triangular columns
        m = np.diag(L); m[0, 1] = rng.uniform(-0.5, 0.5) * L[0]; m[0, 2] = rng.uniform(-0.5, 0.5) * L[0]; m[1, 2] = rng.uniform(-0.5, 0.5) * L[1]
    elif kind == 3:                                   # general matrix
        m = np.diag(L) + rng.uniform(-0.3, 0.3, (3, 3)) * L.min()
    elif kind == 4:                                   # flat slab
        m = np.diag([L[0] * 2, L[1] * 2, rng.uniform(0.6, 1.2)])
    else:                                             # tiny box: 1-2 cells per dimension
        m = np.diag(rng.uniform(0.7, 1.6, 3))
    return m.astype(np.float32)


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    from molar_amd import api, build
    from oracle.oracle import Oracle
    build.build_library()
    eng = api.Engine(0)
    o = Oracle("f32")
    rng = np.random.default_rng(seed)
    fails = 0
    small = 0        # fixed-cutoff cases the small-cell kernels took (pair_small.hip: at most 19 atoms per cell on average)
    for case in range(ncases):
        box = random_box(rng)
        vol = abs(np.linalg.det(box.astype(np.float64)))
        dens = rng.choice([20.0, 60.0, 100.0, 300.0])
        n = int(min(max(vol * dens, 30), 6000))
        frac = rng.random((n, 3))
        pos = (frac @ box.astype(np.float64).T + rng.normal(0, rng.choice([0.0, 0.05, 0.5]), (n, 3))).astype(np.float32)
        rc = float(np.float32(rng.uniform(0.25, 1.3)))
        # a fifth of the cases at other length scales: the matrix-core count pass splits coordinates into f16 parts, whose
        # absolute limits (subnormal lo parts below 2^-14, overflow of cutoff^2 above 65504) only show far from MD units
        scale = float(rng.choice([1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.04, 0.01, 0.004, 300.0]))
        if scale != 1.0:
            box = (box.astype(np.float64) * scale).astype(np.float32)
            pos = (pos.astype(np.float64) * scale).astype(np.float32)
            rc = float(np.float32(rc * scale))
        pbc = int(rng.choice([7, 7, 7, 0, 1, 2, 3, 4, 5, 6]))
        kind = int(rng.choice([0, 0, 0, 1, 1, 2, 3]))
        try:
            ob = o.box_from_matrix(box)
        except Exception:
            continue
        tag = f"case {case}: kind {kind} n {n} rc {rc:.3f} pbc {pbc} box {box.tolist()}"
        try:
            if kind == 0:
                idx = None if rng.random() < 0.5 else np.sort(rng.choice(n, max(n // 2, 2), replace=False)).astype(np.uint64)
                p = pos if idx is None else pos[idx.astype(int)]
                ref = o.search_single_pbc(rc, p, ob, pbc, ids=idx, nthreads=4) if pbc else o.search_single(rc, p, ids=idx, nthreads=4)
                kw = dict(box=box, pbc=pbc) if pbc else {}
                cnt = eng.search_count(api.SEARCH_SINGLE, rc, pos, idx, **kw)
                gd = eng.grid_dims(); small += len(p) <= 19 * gd[0] * gd[1] * gd[2]
                pr, d = eng.search_fill(cnt)
                cnt2, _, _ = eng.search_resident(api.SEARCH_SINGLE, rc, pos, idx, **kw)
                pr2, d2 = eng.search_fill(cnt2)
                if case % 3 == 0:
                    # the consumer-fused mode: Histogram1D::add_one (stats.rs:29-35) over the same distance stream,
                    # with a range that also produces out-of-range and negative bins
                    nb = int(rng.integers(1, 900)); hmin = float(np.float32(rng.uniform(-0.2, 0.4) * scale)); hmax = float(np.float32(rng.uniform(0.5, 1.6) * scale))
                    want = o.histogram_add(hmin, hmax, nb, ref["d"]).astype(np.uint64)
                    bins, hc = eng.search_histogram(api.SEARCH_SINGLE, rc, hmin, hmax, nb, pos, idx, **kw)
                    if hc != len(ref["i"]) or not np.array_equal(bins, want):
                        fails += 1; print("MISMATCH histogram", tag, nb, hmin, hmax)
            elif kind in (1, 3):
                perm = rng.permutation(n)
                i1 = np.sort(perm[: n // 3]).astype(np.uint64); i2 = np.sort(perm[n // 3:]).astype(np.uint64)
                p1, p2 = pos[i1.astype(int)], pos[i2.astype(int)]
                kw = dict(box=box, pbc=pbc) if pbc else {}
                if kind == 1:
                    ref = o.search_double_pbc(rc, p1, p2, ob, pbc, ids1=i1, ids2=i2, nthreads=4) if pbc else o.search_double(rc, p1, p2, ids1=i1, ids2=i2, nthreads=4)
                    cnt = eng.search_count(api.SEARCH_DOUBLE, rc, pos, i1, pos, i2, **kw)
                    gd = eng.grid_dims(); small += max(len(i1), len(i2)) <= 19 * gd[0] * gd[1] * gd[2]
                    pr, d = eng.search_fill(cnt)
                    cnt2, _, _ = eng.search_resident(api.SEARCH_DOUBLE, rc, pos, i1, pos, i2, **kw)
                else:
                    v1 = (rng.uniform(0.1, 0.25, len(i1)) * scale).astype(np.float32); v2 = (rng.uniform(0.1, 0.25, len(i2)) * scale).astype(np.float32)
                    ref = o.search_double_vdw_pbc(p1, p2, v1, v2, ob, pbc, nthreads=4) if pbc else o.search_double_vdw(p1, p2, v1, v2, nthreads=4)
                    cnt = eng.search_count(api.SEARCH_DOUBLE_VDW, None, pos, i1, pos, i2, vdw1=v1, vdw2=v2, **kw)
                    pr, d = eng.search_fill(cnt)
                    cnt2, _, _ = eng.search_resident(api.SEARCH_DOUBLE_VDW, None, pos, i1, pos, i2, vdw1=v1, vdw2=v2, **kw)
                pr2, d2 = eng.search_fill(cnt2)
                if kind == 1 and case % 2 == 0:       # fused histogram of the two-set stream (same-cell duplicates included)
                    nb = int(rng.integers(1, 900)); hmin = float(np.float32(rng.uniform(-0.2, 0.4) * scale)); hmax = float(np.float32(rng.uniform(0.5, 1.6) * scale))
                    want = o.histogram_add(hmin, hmax, nb, ref["d"]).astype(np.uint64)
                    bins, hc = eng.search_histogram(api.SEARCH_DOUBLE, rc, hmin, hmax, nb, pos, i1, pos, i2, **kw)
                    if hc != len(ref["i"]) or not np.array_equal(bins, want):
                        fails += 1; print("MISMATCH histogram", tag, nb, hmin, hmax)
                        if os.environ.get("FUZZ_DEBUG"):
                            print(" count", hc, len(ref["i"]), "bins sum", int(bins.sum()), int(want.sum()))
                            bad = np.flatnonzero(bins != want)
                            print(" bad bins", bad[:20], bins[bad[:20]], want[bad[:20]])
                            np.savez(os.path.join(ROOT, "gpurun_out", "fuzz_case.npz"), pos=pos, box=box, i1=i1, i2=i2, rc=rc, pbc=pbc, nb=nb, hmin=hmin, hmax=hmax)
            else:
                i1 = np.arange(n, dtype=np.uint64); i2 = np.sort(rng.choice(n, max(n // 20, 1), replace=False)).astype(np.uint64)
                p2 = pos[i2.astype(int)]
                if pbc:
                    ref = o.search_within_pbc(rc, pos, p2, ob, pbc, i1, i2, nthreads=4)
                    cnt = eng.search_count(api.SEARCH_WITHIN, rc, pos, i1, pos, i2, box=box, pbc=pbc)
                else:
                    lo, up = o.min_max(pos)
                    lo = lo + (np.float32(-rc) - np.float32(1.1920929e-07)); up = up + (np.float32(rc) + np.float32(1.1920929e-07))
                    ref = o.search_within(rc, pos, p2, lo, up, i1, i2, nthreads=4)
                    cnt = eng.search_count(api.SEARCH_WITHIN, rc, pos, i1, pos, i2, lower=lo, upper=up)
                ids = eng.search_fill_ids(cnt)
                ok = cnt == len(ref["i"]) and np.array_equal(ids, ref["i"])
                if not ok:
                    fails += 1; print("MISMATCH", tag, cnt, len(ref["i"]))
                continue
            ok = (cnt == cnt2 == len(ref["i"]) and np.array_equal(pr[:, 0], ref["i"]) and np.array_equal(pr[:, 1], ref["j"])
                  and np.array_equal(d, ref["d"]) and np.array_equal(pr, pr2) and np.array_equal(d, d2))
            if not ok:
                fails += 1; print("MISMATCH", tag, cnt, cnt2, len(ref["i"]))
        except Exception as exc:      # an engine error on a case the oracle accepts is a failure too
            fails += 1; print("ERROR", tag, repr(exc))
    print(f"{ncases} cases ({small} through the small-cell kernels), {fails} failures")
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
