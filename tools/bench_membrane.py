#!/usr/bin/env python
"""BASELINE.json configs[4] stress test on one GPU: per-lipid order parameters + neighbour analysis on a
500k-atom synthetic bilayer (4000 lipids).  Frames shard over ranks exactly like bench.py; this tool
reports the single-GPU rate (lipids*frames/s and frames/s; no roofline claim — irregular small work), the
time of the smoothing pass alone and the oracle's serial time for the same pass."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from molar_amd import api, build
    from molar_amd import membrane as mb
    build.build_library()
    eng = api.Engine(0)
    xyz, box, first, tpl, masses = mb.build_bilayer(2000, 500_000)
    m = mb.Membrane(eng, len(xyz), first, tpl, masses, mb.MembraneOptions(cutoff=2.5, order_type=1))
    rng = np.random.default_rng(0)
    frames = [(xyz + rng.normal(0, 0.02, xyz.shape).astype(np.float32)) for _ in range(4)]
    m.compute(frames[0].copy(), box)
    K = 20
    t0 = time.perf_counter()
    acc = None
    for s in range(K):
        res = m.compute(frames[s % 4].copy(), box)
        o = np.concatenate([x.mean(0) for x in res["order"]])
        acc = o if acc is None else acc + o
    dt = (time.perf_counter() - t0) / K
    # same with the frames resident on the GPU (torch tensors are used in place: no re-staging per stage)
    import torch
    dframes = [torch.from_numpy(f).cuda() for f in frames]
    m.compute(dframes[0].clone(), box)
    t1 = time.perf_counter()
    for s in range(K):
        m.compute(dframes[s % 4].clone(), box)
    torch.cuda.synchronize()
    dt_dev = (time.perf_counter() - t1) / K
    print(json.dumps({"workload": "C5 same, frames resident in HBM", "frames_per_s": 1.0 / dt_dev, "ms_per_frame": dt_dev * 1e3,
                      "lipid_frames_per_s": len(first) / dt_dev}))
    print(json.dumps({"workload": "C5 500k-atom bilayer, 4000 lipids: unwrap, markers, patches (rc 2.5 nm), normals, one smoothing pass, Scd order "
                                  "of 8000 tails; host frames (12 MB H2D + D2H of the unwrapped frame per call)",
                      "frames_per_s": 1.0 / dt, "ms_per_frame": dt * 1e3, "lipid_frames_per_s": len(first) / dt,
                      "mean_abs_scd": float(np.abs(acc / K).mean())}))


def smooth_only():
    """Membrane::smooth alone (lib.rs:661-812) on the 4000-lipid bilayer and on a 100k-marker sheet."""
    from molar_amd import api
    from oracle.oracle import Oracle
    eng = api.Engine(0)
    o = Oracle("f32")
    rng = np.random.default_rng(2)
    for side in (64, 316):
        L = side * 0.8
        g = (np.stack(np.meshgrid(np.arange(side), np.arange(side), indexing="ij"), -1).reshape(-1, 2) + 0.5
             + 0.2 * rng.normal(size=(side * side, 2))) * L / side
        z = 5.0 + 0.3 * np.sin(2 * np.pi * g[:, 0] / L) + 0.02 * rng.normal(size=len(g))
        head = np.concatenate([g, z[:, None]], 1).astype(np.float32)
        box = np.diag([L, L, 12.0]).astype(np.float32)
        K = len(head)
        n = eng.search_count(api.SEARCH_SINGLE, 2.5, head, box=box, pbc=7, ids_local=True)
        pairs, _ = eng.search_fill(n)
        i = pairs[:, 0].astype(np.int64); j = pairs[:, 1].astype(np.int64)
        src = np.stack([i, j], 1).reshape(-1); dst = np.stack([j, i], 1).reshape(-1)
        order = np.argsort(src, kind="stable")
        poff = np.concatenate([[0], np.cumsum(np.bincount(src, minlength=K))]).astype(np.uint64)
        pids = dst[order].astype(np.uint64)
        nrm = np.tile(np.array([0, 0, 1], np.float32), (K, 1))
        st = api.new_membrane_state(head, nrm, None, len(pids))
        eng.membrane_smooth(box, st, poff, pids)
        eng.profile_enable(True); eng.profile_read()
        reps = 10
        t0 = time.perf_counter()
        for _ in range(reps):
            st = api.new_membrane_state(head, nrm, None, len(pids))
            eng.membrane_smooth(box, st, poff, pids)
        dt = (time.perf_counter() - t0) / reps
        prof = eng.profile_read(); eng.profile_enable(False)
        ob = o.box_from_matrix(box)
        t1 = time.perf_counter()
        o.membrane_smooth(ob, head, nrm, np.ones(K, np.uint8), poff, pids)
        dc = time.perf_counter() - t1
        print(json.dumps({"workload": f"Membrane::smooth, {K} lipids, {len(pids) / K:.1f} patch markers each (rc 2.5 nm), host arrays",
                          "call_ms": dt * 1e3, "device_ms": prof["measure"][0] / reps, "lipids_per_s": K / dt,
                          "oracle_1core_ms": dc * 1e3, "valid": int(st["valid"].sum())}))


if __name__ == "__main__":
    smooth_only()
    main()
