#!/usr/bin/env python
"""An inhomogeneous frame - a slab of liquid density in a mostly empty periodic box - through the ordered pair list: the small-cell kernels are
chosen from the MEAN cell population; here the occupied cells hold many times the mean.  Prints the pipelined time per frame for the
library MOLAR_HIP_PLUGIN selects (tools/build_variant.sh nosmall "-DMH_NO_SMALL_CELLS" search.hip = the regular kernels for every frame).
usage: python tools/bench_slab.py [rc ...]"""
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
    build.build_library()
    eng = api.Engine(0)
    rng = np.random.default_rng(1)
    L, H, thick = 20.0, 60.0, 5.0
    n = int(L * L * thick * 100)
    box = np.diag([L, L, H]).astype(np.float32)
    frames = []
    for s in range(3):
        p = rng.random((n, 3)) * np.array([L, L, thick]) + np.array([0, 0, 0.5 * (H - thick)])
        frames.append(torch.from_numpy(p.astype(np.float32)).cuda())
    for rc in [float(a) for a in sys.argv[1:]] or [0.6, 0.8, 1.0, 1.2]:
        descs = [eng.make_search_desc(api.SEARCH_SINGLE, rc, f, box=box, pbc=7) for f in frames]
        total = None
        for rep in range(2):
            K = 12 if rep else 4
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            prev = None
            for k in range(K):
                t = eng.search_resident_begin(descs[k % 3][0])
                if prev is not None:
                    total = eng.search_resident_end(prev)[0]
                prev = t
            total = eng.search_resident_end(prev)[0]
            eng.synchronize()
            dt = (time.perf_counter() - t0) / K
        dims = [int(np.floor(x / rc)) for x in (L, L, H)]
        print(json.dumps({"rc": rc, "natoms": n, "cells": dims, "mean_atoms_per_cell": round(n / np.prod(dims), 1),
                          "atoms_per_occupied_cell": round(100 * rc ** 3, 1), "pairs": int(total), "ms_per_frame": round(dt * 1e3, 3)}), flush=True)


if __name__ == "__main__":
    main()
