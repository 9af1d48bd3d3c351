#!/usr/bin/env python
"""BASELINE.json configs[4] through the chained frame call (molar_hip_membrane_frame_*): 500k atoms, 4000 lipids, frames
resident in HBM.  Prints one JSON line per mode: one frame at a time (begin + end), two frames in flight, and with the
per-lipid results fetched to the host every frame."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from molar_amd import api, build
    from molar_amd import membrane as mb
    build.build_library()
    eng = api.Engine(0)
    xyz, box, first, tpl, masses = mb.build_bilayer(2000, 500_000)
    m = mb.Membrane(eng, len(xyz), first, tpl, masses, mb.MembraneOptions(cutoff=2.5, order_type=1))
    plan = m._plan()
    rng = np.random.default_rng(0)
    frames = [torch.from_numpy((xyz + rng.normal(0, 0.02, xyz.shape)).astype(np.float32)).cuda() for _ in range(4)]
    K = int(os.environ.get("FRAMES", "200"))
    small = ["valid", "normals", "mean_curv", "gauss_curv", "area", "nvert", "order"]
    pbox = api.PeriodicBox.from_matrix(box)
    for _ in range(3):
        plan.end(plan.begin(frames[0].clone(), pbox))
    work = [f.clone() for f in frames for _ in range((K + 3) // 4)]

    def run(mode):
        bufs = [w.cpu().numpy().copy() for w in work[:K]] if mode == "host" else [w.clone() for w in work[:K]]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if mode == "one":
            for k in range(K):
                plan.end(plan.begin(bufs[k], pbox))
        else:
            prev = plan.begin(bufs[0], pbox)
            for k in range(1, K):
                t = plan.begin(bufs[k], pbox)
                plan.end(prev)
                if mode in ("fetch", "host"):
                    plan.fetch(prev, small)
                prev = t
            plan.end(prev)
            if mode in ("fetch", "host"):
                plan.fetch(prev, small)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / K

    for mode, what in (("one", "one frame at a time"), ("two", "two frames in flight, results stay in HBM"),
                       ("fetch", "two frames in flight, per-lipid results (valid, normals, curvatures, area, nvert, order) fetched"),
                       ("host", "frames in pageable host memory (6 MB up, the unwrapped frame 6 MB back), two in flight, per-lipid results fetched")):
        dt = min(run(mode) for _ in range(3))
        print(json.dumps({"workload": "C5 500k-atom bilayer, 4000 lipids, chained frame call" + ("; " if mode == "host" else ", frames resident; ") + what,
                          "frames_per_s": round(1.0 / dt, 1), "ms_per_frame": round(dt * 1e3, 4), "lipid_frames_per_s": round(4000 / dt)}))


if __name__ == "__main__":
    main()
