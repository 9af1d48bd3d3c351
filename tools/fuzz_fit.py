#!/usr/bin/env python
"""Randomised differential test of the fused fit path (molar_hip_fit_rmsd_batch, molar_hip_fit_batch, fit_transform, molar_hip_fit_stream_*)
against the f64 oracle: selection sizes 3..20000, far-from-origin clouds, near-identical frames (RMSD ~ 1e-4), planar and
nearly collinear selections (for the collinear ones the rotation is not unique: only the RMSD after the fit, the
centre and the gyration radius are compared), large rigid motions.  Usage: python tools/fuzz_fit.py [ncases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(ncases=300, seed=1, eng=None):
    from molar_amd import api, build
    from oracle.oracle import Oracle
    build.build_library()
    eng = eng or api.Engine(0)
    o = Oracle("f64")
    rng = np.random.default_rng(seed)
    fails = 0
    for case in range(ncases):
        natoms = int(rng.integers(50, 30000))
        m = int(rng.integers(3, min(natoms, 20000)))
        kind = case % 6
        centre = rng.uniform(-30, 30, 3) if kind != 1 else rng.uniform(-400, 400, 3)
        ref = (centre + rng.normal(0, rng.uniform(0.3, 5.0), (natoms, 3))).astype(np.float32)
        if kind == 2:      # planar cloud
            ref[:, 2] = np.float32(centre[2])
        if kind == 3:      # nearly collinear
            ref = (centre + np.outer(rng.normal(0, 3, natoms), [1, 0.5, -0.2]) + rng.normal(0, 1e-3, (natoms, 3))).astype(np.float32)
        noise = 1e-4 if kind == 4 else rng.uniform(0.01, 0.3)
        Rz = api.rotation_from_axis_angle(rng.normal(size=3), float(rng.uniform(-3, 3))).astype(np.float64)
        cur = ((ref.astype(np.float64) - centre) @ Rz.T + centre + rng.uniform(-5, 5, 3) + rng.normal(0, noise, (natoms, 3))).astype(np.float32)
        mass = rng.uniform(1, 40, natoms).astype(np.float32)
        idx = np.sort(rng.choice(natoms, m, replace=False)).astype(np.uint64)
        out = eng.fit_rmsd_batch(cur[None].copy(), mass, ref, idx=idx, apply=False)
        R, t = o.fit_transform(cur, mass, ref, mass, idx, idx)
        mv = o.apply_transform(cur, R, t, idx)
        w_rmsd, w_com, w_gyr = o.rmsd(mv, ref, idx, idx), o.center_of_mass(mv, mass, idx), o.gyration(mv, mass, idx)
        scale = max(float(np.abs(ref[idx.astype(np.int64)]).max()), float(np.abs(cur[idx.astype(np.int64)]).max()), 1.0)
        # The engine (like the reference) applies an f32 translation: |dt| <= scale * 6e-8 * sqrt(3) is a common shift of
        # the whole selection that the f64 oracle does not have.  It moves the RMSD by up to |dt| * |mean residual| / RMSD
        # <= |dt|: an absolute floor of ~1e-7 * scale on what ANY f32 fit can reproduce (2 % of an RMSD of 1e-4 at |x| ~ 30).
        quant = 1.2e-7 * scale
        ok = abs(out["rmsd"][0] - w_rmsd) <= 2e-5 * w_rmsd + quant \
            and np.allclose(out["com"][0], w_com, atol=2e-5 * scale) and abs(out["gyration"][0] - w_gyr) <= 2e-5 * max(w_gyr, 1e-3)
        if kind != 3 and m > 3:        # planar selections still have a unique proper rotation; collinear ones do not
            ok = ok and np.allclose(out["R"][0], R, atol=3e-5) and np.allclose(out["t"][0], t, atol=3e-5 * scale * 10)
        # the single-call entry agrees with the batched one bit for bit (same kernels)
        R1, t1 = eng.fit_transform(cur, mass, ref, mass, idx, idx)
        ok = ok and np.array_equal(R1, out["R"][0]) and np.array_equal(t1, out["t"][0])
        if case % 5 == 0:              # batches of >= 4 frames take the packed gather: records equal to the per-frame ones
            nf = int(rng.integers(4, 8))   # (4..7 frames: same workgroup count as one frame, so the same partial sums; from 8 on
                                           # blocks_for() gives a thread more atoms and the f64 partials group differently)
            fr = np.stack([cur] + [(cur + rng.normal(0, 0.02, cur.shape)).astype(np.float32) for _ in range(nf - 1)])
            ob = eng.fit_rmsd_batch(fr.copy(), mass, ref, idx=idx, apply=False)
            for f in range(nf):
                o1 = eng.fit_rmsd_batch(fr[f:f + 1].copy(), mass, ref, idx=idx, apply=False)
                ok = ok and all(np.array_equal(ob[k][f], o1[k][0]) for k in ("rmsd", "R", "t", "com", "gyration"))
        if case % 3 == 0:              # the streamed form (host frames, selection packed by host threads): the batch entry's record and frame
            nf = int(rng.integers(1, 6))
            fr = np.stack([cur] + [(cur + rng.normal(0, 0.02, cur.shape)).astype(np.float32) for _ in range(nf - 1)])
            apply = bool(rng.integers(0, 2))
            wb = fr.copy()
            per = [eng.fit_rmsd_batch(wb[f:f + 1], mass, ref, idx=idx, apply=apply) for f in range(nf)]
            ws = fr.copy()
            fs = api.FitStream(eng, natoms, mass, ref, idx=idx, host_threads=int(rng.integers(0, 4)))
            got, pend = [], []
            for f in range(nf):
                pend.append(fs.begin(ws[f], apply=apply))
                if len(pend) == 3:
                    got.append(fs.end(pend.pop(0)))
            while pend:
                got.append(fs.end(pend.pop(0)))
            fs.close()
            for f in range(nf):
                ok = ok and all(np.array_equal(got[f][k], per[f][k][0]) for k in ("rmsd", "R", "t", "com", "gyration"))
            ok = ok and np.array_equal(ws, wb)
        if not ok:
            fails += 1
            print("MISMATCH", case, kind, natoms, m, out["rmsd"][0], w_rmsd, out["gyration"][0], w_gyr, np.abs(out["R"][0] - R).max())
    print(f"{ncases} cases, {fails} failures")
    return fails


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 300, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
