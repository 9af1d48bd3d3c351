#!/usr/bin/env python
"""Randomised differential test of the Measure / Modify entries against the f64 oracle: centres (plain and periodic, every
PbcDims mask), gyration and inertia (plain and periodic), rmsd / rmsd_mw, min_max, unwrap_simple, the CSR-batched
gyration / rmsd - on selections of 1..20000 atoms, compact blobs and box-filling clouds, orthorhombic and triclinic boxes,
clouds up to 400 nm from the origin.  Tolerances: 1e-5 relative, plus the quantisation of f32 results (half an ulp of
the coordinate magnitude) where a result is a coordinate.  Usage: python tools/fuzz_measure.py [ncases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def random_box(rng, L):
    m = np.diag(L * rng.uniform(0.8, 1.3, 3))
    k = rng.integers(0, 3)
    if k == 1:
        m[0, 2] = -rng.uniform(0, 0.3) * m[0, 0]; m[1, 2] = -rng.uniform(0, 0.3) * m[1, 1]
    elif k == 2:
        m[0, 1] = rng.uniform(-0.5, 0.5) * m[0, 0]; m[0, 2] = rng.uniform(-0.5, 0.5) * m[0, 0]; m[1, 2] = rng.uniform(-0.5, 0.5) * m[1, 1]
    return m.astype(np.float32)


def run(ncases=300, seed=1, eng=None):
    from molar_amd import api, build
    from oracle.oracle import Oracle
    build.build_library()
    eng = eng or api.Engine(0)
    o = Oracle("f64")
    o32 = Oracle("f32")      # periodic quantities hinge on per-atom image DECISIONS (round(f) at |f| = 0.5), which only
                             # the f32 restatement makes exactly like the engine; its serial f32 sums cost ~1e-4 relative
    rng = np.random.default_rng(seed)
    fails = 0

    def bad(what, case, got, want):
        nonlocal fails
        fails += 1
        print("MISMATCH", what, case, got, want)

    for case in range(ncases):
        natoms = int(rng.integers(1, 20000))
        m = int(rng.integers(1, natoms + 1))
        L = float(rng.uniform(3.0, 25.0))
        box = random_box(rng, L)
        far = case % 5 == 1
        origin = rng.uniform(-400, 400, 3) if far else np.zeros(3)
        if case % 2:        # compact blob (a molecule), possibly split over the periodic boundary
            xyz = rng.uniform(0, 1, 3) @ box.astype(np.float64).T + rng.normal(0, rng.uniform(0.1, 0.08 * L), (natoms, 3))
        else:               # box-filling cloud
            xyz = rng.uniform(-0.2, 1.2, (natoms, 3)) @ box.astype(np.float64).T
        xyz = (xyz + origin).astype(np.float32)
        mass = rng.uniform(1, 40, natoms).astype(np.float32)
        idx = np.sort(rng.choice(natoms, m, replace=False)).astype(np.uint64)
        ii = idx.astype(np.int64)
        ob = o.box_from_matrix(box)
        scale = max(float(np.abs(xyz[ii]).max()), 1.0)
        q = 6e-8 * scale                              # half an ulp of an f32 coordinate of that magnitude
        rel = lambda a, b, extra=0.0: abs(a - b) <= 1e-5 * abs(b) + extra

        w = o.center_of_mass(xyz, mass, idx); g = eng.center_of_mass(xyz, mass, idx)
        if not np.allclose(g, w, rtol=1e-5, atol=2 * q): bad("com", case, g, w)
        w = o.center_of_geometry(xyz, idx); g = eng.center_of_geometry(xyz, idx)
        if not np.allclose(g, w, rtol=1e-5, atol=2 * q): bad("cog", case, g, w)
        lo, hi = eng.min_max(xyz, idx); wl, wh = o.min_max(xyz, idx)
        if not (np.array_equal(lo, wl.astype(np.float32)) and np.array_equal(hi, wh.astype(np.float32))): bad("min_max", case, (lo, hi), (wl, wh))
        w = o.gyration(xyz, mass, idx); g = eng.gyration(xyz, mass, idx)
        if not rel(g, w, 4 * q): bad("gyration", case, g, w)
        mom, axes, tens = eng.inertia(xyz, mass, idx)
        wm, wa = o.inertia(xyz, mass, idx)
        if not np.allclose(mom, wm, rtol=2e-5, atol=2e-5 * max(abs(wm).max(), 1e-6) + 8 * q * float(mass[ii].sum()) * np.sqrt(max(wm.max(), 0) / max(mass[ii].sum(), 1e-9) + 1e-12)):
            bad("inertia", case, mom, wm)
        if m >= 3 and not np.allclose(axes.T @ axes, np.eye(3), atol=1e-4): bad("axes", case, axes, None)
        if not far:         # periodic variants are defined for atoms near the cell
            ob32 = o32.box_from_matrix(box)
            ptol = dict(rtol=3e-4, atol=3e-4 * max(L, 1.0) * 0.1 + 2 * q)
            for dims in (7, 3, 5, 1):
                w = o32.center_of_mass_pbc_dims(xyz, mass, ob32, dims, idx); g = eng.center_of_mass_pbc(xyz, mass, box, dims, idx)
                if not np.allclose(g, w, **ptol): bad(f"com_pbc{dims}", case, g, w)
            w = o32.center_of_geometry_pbc_dims(xyz, ob32, 7, idx); g = eng.center_of_geometry_pbc(xyz, box, 7, idx)
            if not np.allclose(g, w, **ptol): bad("cog_pbc", case, g, w)
            w = o32.gyration_pbc(xyz, mass, ob32, idx); g = eng.gyration(xyz, mass, idx, box=box)
            if not abs(g - w) <= 3e-4 * max(w, 1e-2): bad("gyration_pbc", case, g, w)
            u = xyz.copy(); eng.unwrap_simple(u, box, 7, idx)
            wu = o32.unwrap_simple_dim(xyz, ob32, 7, idx)
            if not np.array_equal(u, wu): bad("unwrap", case, np.abs(u - wu).max(), 0)      # bit-identical coordinates
        # a second frame for the two-frame measures
        y = (xyz.astype(np.float64) + rng.normal(0, rng.uniform(1e-3, 0.5), xyz.shape)).astype(np.float32)
        w = o.rmsd(xyz, y, idx, idx); g = eng.rmsd(xyz, y, idx, idx)
        if not rel(g, w): bad("rmsd", case, g, w)
        w = o.rmsd_mw(xyz, mass, y, idx, idx); g = eng.rmsd_mw(xyz, mass, y, idx, idx)
        if not rel(g, w): bad("rmsd_mw", case, g, w)
        # CSR batches: split the selection into random runs
        if m >= 4:
            cuts = np.unique(np.concatenate([[0, m], rng.choice(np.arange(1, m), min(m - 1, int(rng.integers(1, 40))), replace=False)]))
            off = cuts.astype(np.uint64)
            gb = eng.gyration_batch(xyz, idx, off, mass)
            rb = eng.rmsd_batch(xyz, y, idx, off)
            for k in range(len(off) - 1):
                sub = idx[int(off[k]):int(off[k + 1])]
                if not rel(gb[k], o.gyration(xyz, mass, sub), 4 * q): bad("gyration_batch", case, gb[k], o.gyration(xyz, mass, sub))
                if not rel(rb[k], o.rmsd(xyz, y, sub, sub)): bad("rmsd_batch", case, rb[k], o.rmsd(xyz, y, sub, sub))
    print(f"{ncases} cases, {fails} failures")
    return fails


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 300, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
