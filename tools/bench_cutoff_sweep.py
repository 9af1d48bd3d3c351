#!/usr/bin/env python
"""The ordered pair list in HBM over the cutoff (profiles/r05_cutoff_sweep.txt): the headline's 1M-atom frame (triclinic box A,
distance_search_single_pbc, distance_search.rs:892-954) at rc = 0.5 ... 2.0 nm through
  count      molar_hip_search_count alone (host-synchronous)
  serial     molar_hip_search_resident, one call at a time
  pipelined  molar_hip_search_resident_begin / _end with two frames in flight (the bench's loop without the fit)
One JSON object per cutoff.  usage: python tools/bench_cutoff_sweep.py [rc ...]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from molar_amd import api, build, synth
    build.build_library()
    eng = api.Engine(0)
    n = int(os.environ.get("SWEEP_NATOMS", "1000000"))
    box = synth.box_a(n)
    frames = [torch.from_numpy(synth.frame(n, box, s)).cuda() for s in (1, 2, 3)]
    torch.cuda.synchronize()
    rcs = [float(a) for a in sys.argv[1:]] or [0.5, 0.6, 0.8, 1.0, 1.2, 1.4, 1.6, 1.8, 2.0]
    for rc in rcs:
        descs = [eng.make_search_desc(api.SEARCH_SINGLE, rc, f, box=box, pbc=7) for f in frames]
        cnt, _, _ = eng.search_resident_desc(descs[0][0])          # sizes the buffers
        reps = max(3, min(40, int(2.0e9 / max(cnt, 1))))

        def timed(fn, k):
            fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(k):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / k
        t_count = timed(lambda: eng.search_count(api.SEARCH_SINGLE, rc, frames[0], box=box, pbc=7), reps)
        dims = eng.grid_dims()
        t_serial = timed(lambda: eng.search_resident_desc(descs[0][0]), reps)
        # two in flight: begin(k + 1) before end(k)
        state = {"k": 0, "t": None, "pairs": 0}

        def step():
            k = state["k"]
            t_new = eng.search_resident_begin(descs[(k + 1) % 3][0])
            if state["t"] is not None:
                c, _, _ = eng.search_resident_end(state["t"])
                state["pairs"] += c
            state["t"] = t_new
            state["k"] = k + 1
        for _ in range(4):
            step()
        state["pairs"] = 0
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3 * reps):
            step()
        c, _, _ = eng.search_resident_end(state["t"]); state["t"] = None
        torch.cuda.synchronize()
        t_pipe = (time.perf_counter() - t0) / (3 * reps)
        # per-class event times of the same pipelined loop in a separate pass (every class carries its own event pair: 5-10 us each)
        eng.profile_enable(True); eng.profile_read()
        for _ in range(reps):
            step()
        c, _, _ = eng.search_resident_end(state["t"]); state["t"] = None
        pr = eng.profile_read(); eng.profile_enable(False)
        kern = {k: round(v[0] / max(reps, 1), 4) for k, v in pr.items() if k in ("grid_build", "pair_count", "offset_scan", "pair_fill")}
        alg = 12.0 * n + 12.0 * cnt              # coordinates read once, every pair written once as (u32, u32, f32)
        print(json.dumps({"natoms": n, "cutoff_nm": rc, "grid_dims": dims, "atoms_per_cell": round(n / (dims[0] * dims[1] * dims[2]), 1),
                          "pairs": cnt, "ms_count": round(t_count * 1e3, 3), "ms_resident_serial": round(t_serial * 1e3, 3),
                          "ms_resident_pipelined": round(t_pipe * 1e3, 3), "mpairs_per_ms_serial": round(cnt / t_serial / 1e9, 1),
                          "mpairs_per_ms_pipelined": round(cnt / t_pipe / 1e9, 1), "frames_per_s_pipelined": round(1.0 / t_pipe, 1),
                          "kernel_ms_per_frame": kern, "algorithmic_bytes": alg,
                          "hbm_roofline_frac_pipelined": round(alg / t_pipe / 8.0e12, 3)}),
              flush=True)


if __name__ == "__main__":
    main()
