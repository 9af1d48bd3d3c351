#!/usr/bin/env python
"""The vdW overlap search on the shape of its one call site in the reference (molar_bin/src/command_solvate.rs:94-101:
solvent atoms inside the box against the solute, radii from Atom::vdw(), full PBC): a solvent box of N atoms around a compact
solute, element-like radii 0.12 ... 0.21 nm (cutoff = max r1 + max r2 + eps ~ 0.42 nm, cells of a few atoms), one GPU:
  gpu   molar_hip_search_count(DOUBLE_VDW) + _fill (pairs and distances to the host, as the command uses them)
  cpu   the C restatement of distance_search_double_vdw_pbc, best of 1 / 8 / all host threads (tools/cpu_columns.py)
Prints one JSON object per case; coordinates resident in HBM."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timeit(fn, reps):
    fn(); fn()
    t0 = time.perf_counter()
    k = 0
    while k < reps or time.perf_counter() - t0 < 0.15:
        out = fn()
        k += 1
    return (time.perf_counter() - t0) / k, out


def main():
    import torch
    from molar_amd import api, build, synth
    from oracle.oracle import Oracle
    from tools.cpu_columns import cpu_best
    build.build_library()
    eng = api.Engine(0)
    orc = Oracle("f32")
    ncores = os.cpu_count() or 1
    rng = np.random.default_rng(5)
    only = int(os.environ.get("BENCH_VDW_ONLY", "0"))          # one size only (kernel traces)
    for n, nsolute in ((100_000, 5_000), (1_000_000, 50_000)):
        if only and n != only:
            continue
        box = synth.box_a(n)
        pos = synth.frame(n, box, 1)
        centre = (box @ np.array([0.5, 0.5, 0.5], np.float32)).astype(np.float32)
        order = np.argsort(((pos - centre) ** 2).sum(1))
        solute = np.sort(order[:nsolute]).astype(np.uint64)
        mask = np.ones(n, bool)
        mask[solute.astype(np.int64)] = False
        solvent = np.nonzero(mask)[0].astype(np.uint64)
        vdw = rng.choice(np.array([0.12, 0.152, 0.155, 0.17, 0.18, 0.21], np.float32), n)
        v1, v2 = vdw[solvent.astype(np.int64)], vdw[solute.astype(np.int64)]
        dpos = torch.from_numpy(pos).cuda()
        torch.cuda.synchronize()

        def gpu():
            k = eng.search_count(api.SEARCH_DOUBLE_VDW, None, dpos, solvent, dpos, solute, box=box, pbc=7, vdw1=v1, vdw2=v2)
            pairs, d = eng.search_fill(k)
            return pairs, d
        t_gpu, (pairs, d) = timeit(gpu, 5)
        # the same call with the selections and radii already in HBM (a plugin that keeps a topology's columns resident): what is
        # left when 11 MB of index and radius arrays do not cross the link from pageable memory on every call
        dsolv, dsolu = torch.from_numpy(solvent.astype(np.int64)).cuda(), torch.from_numpy(solute.astype(np.int64)).cuda()
        dv1, dv2 = torch.from_numpy(v1).cuda(), torch.from_numpy(v2).cuda()
        torch.cuda.synchronize()

        def gpu_dev():
            k = eng.search_count(api.SEARCH_DOUBLE_VDW, None, dpos, dsolv, dpos, dsolu, box=box, pbc=7, vdw1=dv1, vdw2=dv2)
            return eng.search_fill(k)
        t_gpu_dev, (pairs2, d2) = timeit(gpu_dev, 5)
        assert np.array_equal(pairs, pairs2) and np.array_equal(d, d2)
        ob = orc.box_from_matrix(box)
        p1, p2 = pos[solvent.astype(np.int64)], pos[solute.astype(np.int64)]
        t_cpu, ref, info = cpu_best(lambda nt: orc.search_double_vdw_pbc(p1, p2, v1, v2, ob, 7, nthreads=nt), 2)
        # ids are positions in the two sets, as in the reference (the command indexes `inside_sel` with them, :104-108)
        same = len(ref["i"]) == len(pairs) and np.array_equal(ref["i"], pairs[:, 0].astype(np.uint64)) and \
            np.array_equal(ref["j"], pairs[:, 1].astype(np.uint64)) and np.array_equal(ref["d"], d)
        print(json.dumps({"workload": f"vdW overlap search, {len(solvent)} solvent atoms against a compact {nsolute}-atom solute, full PBC",
                          "natoms": n, "cutoff_nm": float(v1.max() + v2.max()), "grid_dims": eng.grid_dims(), "overlaps": int(len(pairs)),
                          "ms_gpu_count_fill_to_host": t_gpu * 1e3, "ms_gpu_selections_resident": t_gpu_dev * 1e3, "ms_cpu_restatement": t_cpu * 1e3, **info,
                          "speedup": t_cpu / t_gpu, "identical_to_cpu": bool(same)}), flush=True)


if __name__ == "__main__":
    main()
