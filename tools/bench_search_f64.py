#!/usr/bin/env python
"""The f64 drivers (MolAR built with its `f64` feature) on synthetic boxes: distance_search_single_pbc in double, count + fill,
from host arrays as an unmodified caller passes them and from coordinates resident in HBM, beside the f64 build of the CPU
restatement (best of 1 / 8 / all host threads).  One JSON object per size; lists compared element-wise (ids, order, distances).

    python tools/bench_search_f64.py [--sizes 100000,1000000] [--cutoff 1.0]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timeit(fn, min_reps, min_s=0.3):
    fn()
    t0 = time.perf_counter()
    k = 0
    while k < min_reps or time.perf_counter() - t0 < min_s:
        out = fn()
        k += 1
    return (time.perf_counter() - t0) / k, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="100000,1000000")
    ap.add_argument("--cutoff", type=float, default=1.0)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()
    import torch
    from molar_amd import api, build, synth
    build.build_library()
    eng = api.Engine(0)
    for n in [int(x) for x in args.sizes.split(",")]:
        box = synth.box_a(n).astype(np.float64)
        pos = synth.frame(n, synth.box_a(n), 2).astype(np.float64)
        pos += np.random.default_rng(1).normal(0, 1e-9, pos.shape)          # digits below f32 resolution: a genuine f64 frame
        rc = args.cutoff
        t_host, (i, j, d) = timeit(lambda: eng.search_f64(api.SEARCH_SINGLE, rc, pos, box=box, pbc=7), 2)
        line = {"workload": f"f64 distance_search_single_pbc, {n} atoms, triclinic box A, rc {rc} nm, (usize, usize, f64) columns to the host",
                "natoms": n, "pairs": int(len(i)), "grid_dims": list(eng.grid_dims_f64()), "ms_gpu_host_arrays": t_host * 1e3}
        try:
            dpos = torch.from_numpy(pos).cuda()
            torch.cuda.synchronize()
            t_dev, (i2, j2, d2) = timeit(lambda: eng.search_f64(api.SEARCH_SINGLE, rc, dpos, box=box, pbc=7), 2)
            line["ms_gpu_frame_resident"] = t_dev * 1e3
            line["resident_equals_host_path"] = bool(np.array_equal(i, i2) and np.array_equal(j, j2) and np.array_equal(d, d2))
            outs = (torch.empty(len(i), dtype=torch.int64, device="cuda"), torch.empty(len(i), dtype=torch.int64, device="cuda"),
                    torch.empty(len(i), dtype=torch.float64, device="cuda"))      # the caller's result columns, allocated once
            t_res, (i3, j3, d3) = timeit(lambda: eng.search_f64(api.SEARCH_SINGLE, rc, dpos, box=box, pbc=7, device_out=True, out=outs), 5)
            line["ms_gpu_frame_and_result_resident"] = t_res * 1e3
            line["resident_result_equals_host_path"] = bool(np.array_equal(i, i3.cpu().numpy().view(np.uint64)) and np.array_equal(d, d3.cpu().numpy()))
            del i3, j3, d3, outs
        except (TypeError, AttributeError, ValueError):
            pass          # (api.search_f64 of earlier revisions took host arrays only)
        if not args.no_cpu:
            from oracle.oracle import Oracle
            from tools.cpu_columns import cpu_best
            orc = Oracle("f64")
            ob = orc.box_from_matrix(box)
            t_cpu, ref, cols = cpu_best(lambda nt: orc.search_single_pbc(rc, pos, ob, 7, nthreads=nt), reps=1)
            line.update({"ms_cpu_restatement": t_cpu * 1e3, **cols,
                         "identical_to_cpu": bool(np.array_equal(i, ref["i"]) and np.array_equal(j, ref["j"]) and np.array_equal(d, ref["d"]))})
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
