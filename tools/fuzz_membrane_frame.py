#!/usr/bin/env python
"""Randomised differential test of the chained bilayer frame (molar_hip_membrane_frame_*) against the stage-by-stage calls:
random bilayer size, cutoff, order type, iterations, switched-off lipids, sheared boxes, defects that cost lipids mid-way,
host or resident coordinates, one or two frames in flight.  Every array has to agree bit for bit.
Usage: python tools/fuzz_membrane_frame.py [CASES] [SEED]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ARRAYS = ("head", "mid", "tail", "patch_off", "patch_ids", "initial_normals", "valid", "smoothed_head", "normals", "quad_coefs",
          "mean_curv", "gauss_curv", "princ_curvs", "princ_dirs", "area", "nvert", "neib_ids", "voro_vertexes", "fitted_patch_points")


def differ(got, want):
    E = len(want["patch_ids"])
    for k in ARRAYS:
        a, b = np.ascontiguousarray(got[k]), np.ascontiguousarray(want[k])
        if k == "fitted_patch_points":          # (the stage-by-stage state keeps one padding row when there is no patch entry at all)
            a, b = a[:E], b[:E]
        if a.dtype != b.dtype or a.shape != b.shape or a.tobytes() != b.tobytes():
            return f"{k} {a.shape} {b.shape} {a.dtype} {b.dtype}"
    for t, (a, b) in enumerate(zip(got["order"], want["order"])):
        if np.ascontiguousarray(a).tobytes() != np.ascontiguousarray(b).tobytes():
            return f"order[{t}]"
    return None


def main():
    import torch
    from molar_amd import api, build
    from molar_amd import membrane as mb
    build.build_library()
    eng = api.Engine(0)
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    lipids = frames_done = 0
    for case in range(cases):
        rng = np.random.default_rng(seed0 * 100003 + case)
        per = int(rng.integers(12, 400))
        natoms = 2 * per * 52 + int(rng.integers(0, 30000))
        xyz, box, first, tpl, masses = mb.build_bilayer(per, natoms, seed=int(rng.integers(1 << 30)))
        shear = np.eye(3)
        if rng.random() < 0.5:
            shear[0, 1], shear[0, 2], shear[1, 2] = rng.uniform(-0.45, 0.45, 3)
        xyz = (xyz.astype(np.float64) @ shear.T).astype(np.float32)
        box = (shear @ box.astype(np.float64)).astype(np.float32)
        opts = dict(cutoff=float(rng.uniform(0.9, 3.0)), order_type=int(rng.integers(0, 3)), max_smooth_iter=int(rng.integers(1, 4)),
                    unwrap=bool(rng.random() < 0.85))
        if rng.random() < 0.25:
            g = rng.normal(size=3)
            opts["global_normal"] = tuple(float(v) for v in g / np.linalg.norm(g))
        fused = mb.Membrane(eng, len(xyz), first, tpl, masses, mb.MembraneOptions(**opts))
        staged = mb.Membrane(eng, len(xyz), first, tpl, masses, mb.MembraneOptions(fused=False, **opts))
        off = rng.choice(2 * per, size=int(rng.integers(0, max(1, per // 8))), replace=False)
        for m in (fused, staged):
            m.valid[off] = 0
        nfr = int(rng.integers(1, 6))
        fr = []
        for f in range(nfr):
            x = (xyz + rng.normal(0, 0.02, xyz.shape)).astype(np.float32)
            if rng.random() < 0.3:          # a defect: one head group pushed out of its leaflet
                k = int(rng.integers(0, 2 * per))
                x[k * 52: k * 52 + 12] += (shear @ np.array([0, 0, rng.uniform(0.8, 2.0)])).astype(np.float32)
            fr.append(x)
        want = [staged.compute(f.copy(), box) for f in fr]
        resident = rng.random() < 0.5
        bufs = [torch.from_numpy(f.copy()).cuda() if resident else f.copy() for f in fr]
        got = []
        if rng.random() < 0.3:
            got = [fused.compute(b, box) for b in bufs]
        else:
            prev = fused.compute_begin(bufs[0], box)
            for k in range(1, nfr):
                t = fused.compute_begin(bufs[k], box)
                got.append(fused.compute_end(prev))
                prev = t
            got.append(fused.compute_end(prev))
        for k, (g, w) in enumerate(zip(got, want)):
            bad = differ(g, w)
            if bad:
                print(f"MISMATCH seed {seed0} case {case} frame {k}: {bad}  (per {per}, {opts}, resident {resident})")
                sys.exit(1)
        if not np.array_equal(fused.valid, staged.valid):
            print(f"MISMATCH seed {seed0} case {case}: valid flags after the trajectory")
            sys.exit(1)
        lipids += 2 * per * nfr
        frames_done += nfr
    print(f"fuzz_membrane_frame seed {seed0}: {cases} cases, {frames_done} frames, {lipids} lipid-frames bit-identical to the stages")


if __name__ == "__main__":
    main()
