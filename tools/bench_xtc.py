#!/usr/bin/env python
"""Frame supply rate: XTC decode of 1M-atom frames on host threads into HBM (SURVEY.md §8f rank 3, §8e scaling
risk).  Streams are written by the oracle's encoder (water-like triplets -> small-delta runs with the pair swap)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    from molar_amd import api, build
    from molar_amd.xtc import XtcReader
    from oracle.oracle import Oracle
    from test_xtc_cpu import synthetic_frames
    build.build_library()
    eng = api.Engine(0)
    o = Oracle("f32")
    n, nf = 1_000_000, 16
    frames, box9 = synthetic_frames(n, 2)
    blob = b"".join(o.xtc_encode(frames[k % 2] * 3.5, box9 * 3.5, step=k, time=float(k)) for k in range(nf))
    r = XtcReader(blob, engine=eng)
    dev = torch.empty((nf, n, 3), dtype=torch.float32, device="cuda")
    host = np.empty((nf, n, 3), np.float32)
    t0 = time.perf_counter(); o.xtc_decode(blob, 0); t_or = time.perf_counter() - t0
    for T in (1, 8, 16, 32, 64):
        r.read_frames(0, nf, out=dev, nthreads=T)
        t0 = time.perf_counter()
        r.read_frames(0, nf, out=dev, nthreads=T)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        t0 = time.perf_counter()
        r.read_frames(0, nf, out=host, nthreads=T)
        dh = time.perf_counter() - t0
        print(json.dumps({"workload": f"XTC decode, {nf} frames x {n} atoms, {len(blob) / nf / 1e6:.2f} MB/frame compressed",
                          "threads": T, "to_device_frames_per_s": nf / dt, "to_device_matoms_per_s": nf * n / dt / 1e6,
                          "to_host_frames_per_s": nf / dh, "oracle_1thread_frames_per_s": 1.0 / t_or}))


if __name__ == "__main__":
    main()
