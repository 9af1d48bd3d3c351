#!/usr/bin/env python
"""Randomised differential test of molar_hip_lipid_tail_order (Measure::lipid_tail_order, measure.rs:270-422) against the
f32 and f64 oracles: random-walk tails of 3..30 carbons, one normal per tail or one per bond, 0..3 double bonds at legal
positions, all three order types, degenerate geometry (straight segments: zero cross products).
Usage: python tools/fuzz_lipid_order.py [ncases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(ncases=100, seed=1, eng=None):
    from molar_amd import api, build
    from oracle.oracle import Oracle
    build.build_library()
    eng = eng or api.Engine(0)
    o32, o64 = Oracle("f32"), Oracle("f64")
    rng = np.random.default_rng(seed)
    fails = 0
    for case in range(ncases):
        ntails = int(rng.integers(1, 60))
        order_type = int(rng.integers(0, 3))
        per_bond = bool(rng.integers(0, 2))
        tails, bonds, normals, pts = [], [], [], []
        used = 0
        for t in range(ntails):
            n = int(rng.integers(3, 31))
            p = np.zeros((n, 3)); p[0] = rng.uniform(-20, 20, 3)
            d = rng.normal(size=3); d /= np.linalg.norm(d)
            wobble = 0.0 if (case % 7 == 3 and t % 3 == 0) else 0.9          # some perfectly straight tails
            for k in range(1, n):
                d = d + wobble * rng.normal(size=3); d /= np.linalg.norm(d)
                p[k] = p[k - 1] + 0.153 * d
            bo = np.ones(n - 1, np.uint8)
            if order_type != 0:
                for _ in range(int(rng.integers(0, 4))):
                    lo, hi = 1, (n - 3 if per_bond else n - 2)               # a double bond needs C(i-1) and, per bond, normal i+1
                    if hi > lo:
                        b = int(rng.integers(lo, hi))
                        if bo[b - 1] == 1 and (b + 1 >= n - 1 or bo[b + 1] == 1):
                            bo[b] = 2
            nn = rng.normal(size=(n - 2 if per_bond else 1, 3)); nn /= np.linalg.norm(nn, axis=1)[:, None]
            tails.append(np.arange(used, used + n, dtype=np.uint64)); used += n
            bonds.append(bo); normals.append(nn.astype(np.float32)); pts.append(p)
        xyz = np.concatenate(pts).astype(np.float32)
        got = eng.lipid_tail_order(xyz, tails, order_type, normals, bonds)
        for t in range(ntails):
            w32 = o32.lipid_tail_order(xyz, order_type, normals[t], bonds[t], idx=tails[t])
            w64 = o64.lipid_tail_order(xyz, order_type, normals[t], bonds[t], idx=tails[t])
            # The engine evaluates the reference's f32 expressions: it must agree with the f32 oracle everywhere, NaNs
            # included (a straight segment has a zero cross product and normalises to NaN in the reference too).  The f64
            # oracle is only a sanity check where the f32 formula is well conditioned (|value - f64| small in the f32 oracle).
            fin = np.isfinite(w32)
            ok = got[t].shape == w32.shape and np.array_equal(np.isfinite(got[t]), fin)
            ok = ok and np.allclose(got[t][fin], w32[fin], atol=5e-5)
            well = fin & np.isfinite(w64) & (np.abs(w32 - w64) < 1e-3)
            ok = ok and np.allclose(got[t][well], w64[well], atol=1.1e-3)
            if not ok:
                fails += 1
                print("MISMATCH", case, t, order_type, per_bond, got[t], w32)
                break
    print(f"{ncases} cases, {fails} failures")
    return fails


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
