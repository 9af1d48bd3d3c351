#!/usr/bin/env python
"""`within` on the reference's own benchmark shapes (molar/benches/comparison_large.rs:10-47: `within 1.0 of protein`;
within_size_bench.rs:13-47: cutoff 0.3 ... 4.2 nm around a group of residues), one GPU:
  set    molar_hip_within_count + _fill   (sorted unique ids, what the selection keeps)
  stream molar_hip_search_count(WITHIN) + _fill_ids + np.unique on the host   (the reference's own two steps)
  cpu    the C restatement of distance_search_within_pbc + np.unique, best of 1 / 8 / all host threads (tools/cpu_columns.py)
Prints one JSON object per case; frames resident in HBM, the id list brought to the host every call (as the selection
language does)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timeit(fn, reps):
    """two warm-up calls (buffers grow on the first), then at least `reps` calls and at least 0.15 s"""
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
    cpu = "--no-cpu" not in sys.argv

    def case(name, n, cutoff, idx1, idx2, box, pos, reps=10, cpu_reps=2):
        dpos = torch.from_numpy(pos).cuda()
        torch.cuda.synchronize()
        ob = orc.box_from_matrix(box)
        t_set, got = timeit(lambda: eng.within_set(cutoff, dpos, idx1, dpos, idx2, box=box, pbc=7), reps)
        # the reference's benchmark asks against ONE frame over and over: with the hold on, the first set's grid is reused
        eng.within_hold(True)
        t_hold, got_h = timeit(lambda: eng.within_set(cutoff, dpos, idx1, dpos, idx2, box=box, pbc=7), reps)
        eng.within_hold(False)
        assert np.array_equal(got, got_h)

        def stream():
            k = eng.search_count(api.SEARCH_WITHIN, cutoff, dpos, idx1, dpos, idx2, box=box, pbc=7)
            return np.unique(eng.search_fill_ids(k)), k
        t_stream, (got2, nstream) = timeit(stream, max(reps // 3, 2))
        assert np.array_equal(got, got2)
        # candidate evaluations of the reference's plan (every first-set atom against every second-set atom of its <= 27 partner cells)
        rec = {"workload": name, "natoms": n, "cutoff_nm": cutoff, "set1": int(len(idx1)), "set2": int(len(idx2)), "found": int(len(got)),
               "stream_len": int(nstream), "ms_set": t_set * 1e3, "ms_set_grid_held": t_hold * 1e3, "ms_stream_plus_unique": t_stream * 1e3,
               "speedup_set_over_stream": t_stream / t_set}
        if cpu:
            p1, p2 = pos[idx1.astype(np.int64)], pos[idx2.astype(np.int64)]

            def cpu_fn(nt):
                r = orc.search_within_pbc(cutoff, p1, p2, ob, 7, idx1, idx2, nthreads=nt)
                return np.unique(r["i"])
            t_cpu, ref, info = cpu_best(cpu_fn, cpu_reps)
            assert np.array_equal(got, ref)
            rec.update({"ms_cpu_restatement": t_cpu * 1e3, **info, "speedup_set_over_cpu": t_cpu / t_set,
                        "speedup_best_gpu_form_over_cpu": t_cpu / min(t_set, t_stream)})
        print(json.dumps(rec), flush=True)

    # ---- comparison_large.rs shape: `within 1.0 of <100k-atom selection>` on the 1M-atom frame
    n = 1_000_000
    box = synth.box_a(n)
    pos = synth.frame(n, box, 1)
    all_idx = np.arange(n, dtype=np.uint64)
    # (a) the selection is a compact solute: the 100k atoms nearest to the box centre
    centre = (box @ np.array([0.5, 0.5, 0.5], np.float32)).astype(np.float32)
    order = np.argsort(((pos - centre) ** 2).sum(1))
    blob = np.sort(order[:100_000]).astype(np.uint64)
    case("within 1.0 of a compact 100k-atom solute, 1M-atom frame", n, 1.0, all_idx, blob, box, pos)
    # (b) the selection is spread over the box: every 10th atom
    case("within 1.0 of every 10th atom (100k), 1M-atom frame", n, 1.0, all_idx, all_idx[::10], box, pos)
    # ---- within_size_bench.rs shape: cutoff sweep around groups of `n_res` residues (here 10-atom residues) in a 100k-atom box
    n = 100_000
    box = synth.box_a(n)
    pos = synth.frame(n, box, 1)
    all_idx = np.arange(n, dtype=np.uint64)
    order = np.argsort(((pos - (box @ np.array([0.5, 0.5, 0.5], np.float32))) ** 2).sum(1))
    for n_res in (1, 20, 60):
        grp = np.sort(order[:10 * (n_res + 1)]).astype(np.uint64)
        for cutoff in (0.3, 0.8, 1.5, 2.5, 4.2):
            case(f"within {cutoff} of {n_res + 1} residues ({len(grp)} atoms), 100k-atom box", n, cutoff, all_idx, grp, box, pos, reps=10, cpu_reps=2)


if __name__ == "__main__":
    main()
