#!/usr/bin/env python
"""Randomised differential test of molar_hip_membrane_smooth (Membrane::smooth, molar_membrane/src/lib.rs:661-812) against
the oracle's restatement: undulating, noisy, tilted sheets of 100..4000 markers in orthorhombic and sheared boxes, with
holes, patch cutoffs from sparse to crowded and some pre-invalidated lipids.  Validity, Voronoi neighbour ids and vertex
counts must be exact, floats within 2e-5 (5e-5 for the curvatures of crowded patches).  Usage: python tools/fuzz_membrane.py [ncases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(ncases=60, seed=1, eng=None):
    from molar_amd import api, build
    from oracle.oracle import Oracle
    build.build_library()
    eng = eng or api.Engine(0)
    o = Oracle("f32")
    rng = np.random.default_rng(seed)
    fails = 0
    tot_valid = tot_invalid = 0
    for case in range(ncases):
        side = int(rng.integers(10, 64))
        spacing = float(rng.uniform(0.6, 1.0))
        L = side * spacing
        g = (np.stack(np.meshgrid(np.arange(side), np.arange(side), indexing="ij"), -1).reshape(-1, 2) + 0.5
             + rng.uniform(0.05, 0.3) * rng.normal(size=(side * side, 2))) * spacing
        amp, noise = rng.uniform(0, 0.6), rng.uniform(0, 0.05)
        z = 5.0 + amp * np.sin(2 * np.pi * g[:, 0] / L) * np.cos(2 * np.pi * g[:, 1] / L) + noise * rng.normal(size=len(g))
        head = np.concatenate([g, z[:, None]], 1)
        if case % 3 == 1:                                   # a hole: lipids next to it get wall vertices / invalid cells
            c = rng.uniform(0.3, 0.7, 2) * L
            head = head[np.linalg.norm(head[:, :2] - c, axis=1) > rng.uniform(1.0, 2.5)]
        box = np.diag([L, L, 12.0]).astype(np.float32)
        if case % 4 == 2:
            box[0, 1] = np.float32(0.3 * L)                  # sheared in the membrane plane
        head = head.astype(np.float32)
        K = len(head)
        rc = float(rng.uniform(1.3, 2.6))
        ob = o.box_from_matrix(box)
        r = o.search_single_pbc(rc, head, ob, 7)
        i = r["i"].astype(np.int64); j = r["j"].astype(np.int64)
        src = np.stack([i, j], 1).reshape(-1); dst = np.stack([j, i], 1).reshape(-1)
        order = np.argsort(src, kind="stable")
        poff = np.concatenate([[0], np.cumsum(np.bincount(src, minlength=K))]).astype(np.uint64)
        pids = dst[order].astype(np.uint64)
        tilt = rng.normal(0, 0.15, (K, 3)); tilt[:, 2] = 1.0
        nrm = (tilt / np.linalg.norm(tilt, axis=1)[:, None]).astype(np.float32)
        valid = (rng.random(K) > 0.03).astype(np.uint8)
        st = api.new_membrane_state(head, nrm, valid, len(pids))
        eng.membrane_smooth(box, st, poff, pids)
        w = o.membrane_smooth(ob, head, nrm, valid, poff, pids)
        ok = np.array_equal(st["valid"], w["valid"])
        good = w["valid"].astype(bool)
        tot_valid += int(good.sum()); tot_invalid += int((~good).sum())
        ok = ok and np.array_equal(st["nvert"][good], w["nvert"][good])
        worst = 0.0
        if ok:
            for k in np.flatnonzero(good):
                s0 = int(poff[k]) + 4 * k; nv = int(w["nvert"][k])
                if not np.array_equal(st["neib_ids"][s0:s0 + nv], w["neib_ids"][s0:s0 + nv]):
                    ok = False; break
                worst = max(worst, float(np.abs(st["voro_vertexes"][s0:s0 + nv] - w["voro"][s0:s0 + nv]).max(initial=0.0)))
            for mine, theirs in (("quad_coefs", "coefs"), ("mean_curv", "mean_curv"), ("gauss_curv", "gauss_curv"), ("area", "area"),
                                 ("princ_curvs", "princ_curvs"), ("normals", "normals"), ("head_markers", "head")):
                a, b = st[mine][good], w[theirs][good]
                d = np.abs(a - b) / (1.0 + np.abs(b))
                worst = max(worst, float(d.max(initial=0.0)))
        if not ok or worst > 5e-5:
            fails += 1
            print("MISMATCH", case, K, round(rc, 2), "valid", int(good.sum()), "worst", worst, "structure ok" if ok else "STRUCTURE DIFFERS")
    print(f"{ncases} cases ({tot_valid} valid and {tot_invalid} invalidated lipids compared), {fails} failures")
    return fails


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
