#!/usr/bin/env python
"""Where the host's time goes in the C5 loop on one context (bench.py --workload membrane --streams 1): the same loop with a clock
around each call - begin, end, fetch, the host sums.  Usage: python tools/c5_host_times.py [frames]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(n=400):
    import torch
    from molar_amd import api, build
    from molar_amd import membrane as mb
    build.build_library()
    xyz, box, first, tpl, masses = mb.build_bilayer(2000, 500_000)
    nl = len(first)
    pbox = api.PeriodicBox.from_matrix(box)
    eng = api.Engine(0)
    rng = np.random.default_rng(3)
    src = [torch.from_numpy((xyz + rng.normal(0, 0.02, xyz.shape)).astype(np.float32)).cuda() for _ in range(8)]
    m = mb.Membrane(eng, len(xyz), first, tpl, masses, mb.MembraneOptions(cutoff=2.5, order_type=1))
    plan = m._plan()
    plan.set_valid(None)
    small = ["valid", "normals", "mean_curv", "gauss_curv", "area", "nvert", "order"]
    norder = plan.norder // nl
    T = dict(begin=0.0, end=0.0, fetch=0.0, sums=0.0)
    acc = np.zeros(4 + norder)

    def sums(r):
        ok = r["valid"].astype(bool)
        acc[0] += int(ok.sum()); acc[1] += int(r["nvert"][ok].sum())
        acc[2] += float(r["area"][ok].sum(dtype=np.float64)); acc[3] += float(np.abs(r["mean_curv"][ok]).sum(dtype=np.float64))
        acc[4:] += r["order"].reshape(nl, norder)[ok].sum(axis=0, dtype=np.float64)

    for mode in ("end, fetch, sums", "end with arrays; sums behind the next begin"):
        for k in T: T[k] = 0.0
        bufs = [src[k % 8].clone() for k in range(n)]
        torch.cuda.synchronize()
        t00 = time.perf_counter()
        prev = got = None
        for k in range(n):
            t0 = time.perf_counter(); t = plan.begin(bufs[k], pbox); t1 = time.perf_counter(); T["begin"] += t1 - t0
            if mode.startswith("end,"):
                if prev is not None:
                    plan.end(prev); t2 = time.perf_counter(); T["end"] += t2 - t1
                    r = plan.fetch(prev, small); t3 = time.perf_counter(); T["fetch"] += t3 - t2
                    sums(r); T["sums"] += time.perf_counter() - t3
            else:
                if got is not None:
                    sums(got); got = None
                t2 = time.perf_counter(); T["sums"] += t2 - t1
                if prev is not None:
                    got = plan.end(prev, small)[1]; T["end"] += time.perf_counter() - t2
            prev = t
        plan.end(prev)
        eng.synchronize()
        tot = time.perf_counter() - t00
        print(f"[{mode}] {n} frames, {tot / n * 1e6:.0f} us per frame:", {k: round(v / n * 1e6, 1) for k, v in T.items()})


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 400)
