#!/usr/bin/env python
"""Randomised differential test of the f64 Measure / Modify entries (molar_amd/csrc/measure_f64.hip) against the oracle's
f64 build: selection sizes 1..30000, clouds near and far from the origin, random / contiguous / whole selections,
orthorhombic and triclinic boxes, every periodicity mask, batches of 1..6 frames.
Usage: python tools/fuzz_measure_f64.py [ncases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(ncases=300, seed=1):
    from molar_amd import api, build
    from oracle.oracle import Oracle
    build.build_library()
    m64 = api.MeasureF64(api.Engine(0))
    o = Oracle("f64")
    rng = np.random.default_rng(seed)
    fails = 0

    def close(a, b, atol):
        return np.allclose(a, b, rtol=0, atol=atol)

    for case in range(ncases):
        natoms = int(rng.integers(4, 30000))
        kind = case % 4
        if kind == 0:
            idx = None
        elif kind == 1:
            m = int(rng.integers(3, natoms + 1))
            idx = np.sort(rng.choice(natoms, m, replace=False)).astype(np.uint64)
        elif kind == 2:
            m = int(rng.integers(3, natoms + 1)); a = int(rng.integers(0, natoms - m + 1))
            idx = np.arange(a, a + m, dtype=np.uint64)
        else:
            idx = rng.permutation(natoms)[: int(rng.integers(3, natoms + 1))].astype(np.uint64)      # unsorted
        centre = rng.uniform(-500, 500, 3) if case % 5 == 0 else rng.uniform(0, 10, 3)
        sig = float(rng.uniform(0.2, 4.0))
        ref = centre + rng.normal(0, sig, (natoms, 3))
        R0 = api.rotation_from_axis_angle(rng.normal(size=3), float(rng.uniform(-3, 3))).astype(np.float64)
        noise = 10.0 ** rng.uniform(-9, -1)
        cur = np.ascontiguousarray((ref - centre) @ R0.T + centre + rng.uniform(-5, 5, 3) + rng.normal(0, noise, (natoms, 3)))
        mass = rng.uniform(0.5, 40, natoms)
        scale = float(max(np.abs(cur).max(), np.abs(ref).max()))
        ok = True
        ok &= close(m64.center_of_geometry(cur, idx), o.center_of_geometry(cur, idx), 1e-12 * scale)
        ok &= close(m64.center_of_mass(cur, mass, idx), o.center_of_mass(cur, mass, idx), 1e-12 * scale)
        ok &= abs(m64.gyration(cur, mass, idx) - o.gyration(cur, mass, idx)) <= 1e-11 * sig + 1e-13 * scale
        ok &= abs(m64.rmsd(cur, ref, idx, idx) - o.rmsd(cur, ref, idx, idx)) <= 1e-12 * scale
        ok &= abs(m64.rmsd_mw(cur, mass, ref, idx, idx) - o.rmsd_mw(cur, mass, ref, idx, idx)) <= 1e-12 * scale
        lo, hi = m64.min_max(cur, idx); rlo, rhi = o.min_max(cur, idx)
        ok &= np.array_equal(lo, rlo) and np.array_equal(hi, rhi)
        mom, axes, tens = m64.inertia(cur, mass, idx)
        rt = o.inertia_tensor(cur, mass, idx)
        ok &= close(tens, rt, 1e-10 * np.abs(rt).max()) and close(axes @ np.diag(mom) @ axes.T, rt, 1e-9 * np.abs(rt).max())
        R, t = m64.fit_transform(cur, mass, ref, mass, idx, idx)
        Ro, to = o.fit_transform(cur, mass, ref, mass, idx, idx)
        ok &= close(R, Ro, 1e-9) and close(t, to, 1e-8 * scale)
        mv = cur.copy(); m64.apply_transform(mv, R, t, idx)
        want = o.apply_transform(cur, R, t, idx)
        ok &= np.array_equal(mv, want)
        if case % 3 == 0:            # the batched loop
            nf = int(rng.integers(1, 7))
            fr = np.ascontiguousarray(np.stack([cur] + [cur + rng.normal(0, noise, cur.shape) for _ in range(nf - 1)]))
            ob = m64.fit_rmsd_batch(fr.copy(), mass, ref, idx=idx, apply=False)
            for f in range(nf):
                Rf, tf = o.fit_transform(fr[f], mass, ref, mass, idx, idx)
                mvf = o.apply_transform(fr[f], ob["R"][f], ob["t"][f], idx)
                ok &= close(ob["R"][f], Rf, 1e-9) and close(ob["t"][f], tf, 1e-8 * scale)
                ok &= abs(ob["rmsd"][f] - o.rmsd(mvf, ref, idx, idx)) <= 1e-9 * max(o.rmsd(mvf, ref, idx, idx), 1e-300) + 1e-14 * scale
                ok &= close(ob["com"][f], o.center_of_mass(mvf, mass, idx), 1e-11 * scale)
                ok &= abs(ob["gyration"][f] - o.gyration(mvf, mass, idx)) <= 1e-10 * sig + 1e-12 * scale
        if case % 2 == 0:            # periodic entries
            if case % 4 == 0:
                box = np.diag(rng.uniform(3, 9, 3))
            else:
                box = np.diag(rng.uniform(3, 9, 3)); box[0, 1] = rng.uniform(-2, 2); box[0, 2] = rng.uniform(-2, 2); box[1, 2] = rng.uniform(-2, 2)
            blob = rng.normal(0, 0.5, (natoms, 3)) + rng.uniform(0, 5, 3)
            wrapped = np.ascontiguousarray(((blob @ np.linalg.inv(box).T) % 1.0) @ box.T)
            bo = o.box_from_matrix(box)
            dims = int(rng.choice([7, 3, 5, 6, 1, 2, 4]))
            ok &= close(m64.center_of_mass_pbc(wrapped, mass, box, dims, idx), o.center_of_mass_pbc_dims(wrapped, mass, bo, dims, idx), 1e-11)
            ok &= close(m64.center_of_geometry_pbc(wrapped, box, dims, idx), o.center_of_geometry_pbc_dims(wrapped, bo, dims, idx), 1e-11)
            ok &= abs(m64.gyration_pbc(wrapped, mass, box, idx) - o.gyration_pbc(wrapped, mass, bo, idx)) <= 1e-11
            pm, pa, pt = m64.inertia(wrapped, mass, idx, box)
            prt = o.inertia_tensor(wrapped, mass, idx, bo)
            ok &= close(pt, prt, 1e-10 * np.abs(prt).max()) and close(pa @ np.diag(pm) @ pa.T, prt, 1e-9 * np.abs(prt).max())
            un = wrapped.copy(); m64.unwrap_simple(un, box, dims, idx)
            ok &= np.array_equal(un, o.unwrap_simple_dim(wrapped, bo, dims, idx))
        if not ok:
            fails += 1
            print("MISMATCH", case, kind, natoms, None if idx is None else len(idx), noise)
    print(f"{ncases} cases, {fails} failures")
    return fails


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 300, int(sys.argv[2]) if len(sys.argv) > 2 else 1) else 0)
