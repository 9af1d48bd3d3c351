#!/usr/bin/env python
"""Secondary workloads of BASELINE.json (configs[2], [3]) on one GPU — informational numbers for
DESIGN.md; the headline metric stays bench.py.  Prints one JSON object per workload."""
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
    dev = torch.device("cuda", 0)

    # ---- C3: 1M-atom frames, 100k-atom selection: Kabsch fit + apply + rmsd + COM + gyration, batched
    n, F = 1_000_000, 64
    box = synth.box_a(n)
    g = torch.Generator(device=dev); g.manual_seed(1)
    base = (torch.rand((n, 3), generator=g, device=dev, dtype=torch.float64) @ torch.from_numpy(box.astype(np.float64)).to(dev).T)
    frames = (base[None] + 0.05 * torch.randn((F, n, 3), generator=g, device=dev, dtype=torch.float32).double()).float().contiguous()
    ref = base.float().contiguous()
    mass = torch.from_numpy(synth.masses(n)).to(dev)
    idx = torch.arange(0, n, 10, device=dev, dtype=torch.int64)
    for batch, apply in ((1, False), (1, True), (64, False), (64, True)):
        work = frames.clone() if apply else frames          # apply moves the selection in place
        eng.fit_rmsd_batch(work[:batch], mass, ref, idx=idx, apply=apply)
        eng.synchronize(); torch.cuda.synchronize()
        reps = 20 if batch == 1 else 5
        # wall time with the engine's event profiling off (two hipEventRecord per call otherwise), kernel time from a
        # second loop with it on
        t0 = time.perf_counter()
        for _ in range(reps):
            for s in range(0, F, batch):
                eng.fit_rmsd_batch(work[s:s + batch], mass, ref, idx=idx, apply=apply)
        eng.synchronize(); torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / (reps * F)
        eng.profile_enable(True); eng.profile_read()
        for s in range(0, F, batch):
            eng.fit_rmsd_batch(work[s:s + batch], mass, ref, idx=idx, apply=apply)
        eng.synchronize(); torch.cuda.synchronize()
        ev_ms, ev_n = eng.profile_read()["measure"]
        eng.profile_enable(False)
        alg = (44.0 if apply else 32.0) * 1e5              # 32*M gather pass (+12*M written by apply_transform)
        print(json.dumps({"workload": "C3 fit+rmsd+COM+gyration, M=1e5 of N=1e6, frames resident", "frames_per_call": batch,
                          "apply_transform": apply, "launches_per_call": (3 if apply else 2) + (1 if batch >= 4 else 0),
                          "frames_per_s": 1.0 / dt, "us_per_frame": dt * 1e6, "kernel_us_per_call": ev_ms * 1e3 / max(ev_n, 1),
                          "algorithmic_GBps": alg / dt / 1e9,
                          "algorithmic_GBps_kernels_only": alg * batch / (ev_ms * 1e-3 / max(ev_n, 1)) / 1e9}))
        del work
    # the same loop for an f64 MolAR (molar_hip_fit_rmsd_batch_f64: three gather passes per frame, not the fused f32 pass)
    m64 = api.MeasureF64(eng)
    F64 = 16
    fr64 = frames[:F64].double().contiguous(); ref64 = ref.double().contiguous(); mass64 = mass.double().contiguous()
    m64.fit_rmsd_batch(fr64, mass64, ref64, idx=idx, apply=False)
    eng.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        m64.fit_rmsd_batch(fr64, mass64, ref64, idx=idx, apply=False)
    eng.synchronize(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (5 * F64)
    print(json.dumps({"workload": "C3 in f64 (fit+rmsd+COM+gyration, M=1e5 of N=1e6, frames resident, 16 frames per call, no apply)",
                      "frames_per_s": 1.0 / dt, "us_per_frame": dt * 1e6,
                      "algorithmic_GBps": (24 + 8 + 8 + 2 * (24 + 24 + 8 + 16)) * 1e5 / dt / 1e9}))
    del fr64, ref64, mass64
    # streamed from host (PCIe): one 12 MB frame per call
    hframe = frames[0].cpu().numpy(); href = ref.cpu().numpy(); hmass = mass.cpu().numpy(); hidx = idx.cpu().numpy().astype(np.uint64)
    eng.fit_rmsd_batch(hframe[None].copy(), hmass, href, idx=hidx, apply=False)
    t0 = time.perf_counter()
    for _ in range(10):
        eng.fit_rmsd_batch(hframe[None], hmass, href, idx=hidx, apply=False)
    dt = (time.perf_counter() - t0) / 10
    print(json.dumps({"workload": "C3 streamed from pageable host memory (frame + reference + mass + idx per call)",
                      "frames_per_s": 1.0 / dt, "ms_per_frame": dt * 1e3}))
    # the same frames through molar_hip_fit_stream_*: the selection is packed by host threads, 1.2 of the frame's 12 MB cross the
    # link, three frames in flight; records equal the batch entry's
    from molar_amd import api as _api
    hframes = [frames[f].cpu().numpy().copy() for f in range(8)]
    want = eng.fit_rmsd_batch(np.stack(hframes), hmass, href, idx=hidx, apply=False)
    for threads in (1, 4, 8, 16):
        for apply in (False, True):
            fs = _api.FitStream(eng, len(hmass), hmass, href, idx=hidx, host_threads=threads)
            work = [h.copy() for h in hframes]
            nrun = 200
            got, pending = [], []
            t0 = time.perf_counter()
            for k in range(nrun):
                pending.append(fs.begin(work[k % 8], apply=apply and k < 8))      # (a frame is moved once: the later laps time the fit alone)
                if len(pending) == 3:
                    got.append(fs.end(pending.pop(0)))
            while pending:
                got.append(fs.end(pending.pop(0)))
            dt = (time.perf_counter() - t0) / nrun
            same = all(np.array_equal(got[k]["rmsd"], want["rmsd"][k]) and np.array_equal(got[k]["R"], want["R"][k]) for k in range(8))
            fs.close()
            print(json.dumps({"workload": "C3 streamed from host memory through molar_hip_fit_stream (selection packed by host threads into pinned "
                                          "staging, three frames in flight)", "host_threads": threads, "apply_first_lap": apply,
                              "frames_per_s": 1.0 / dt, "ms_per_frame": dt * 1e3, "records_equal_batch_entry": bool(same)}))
    del frames

    # ---- C4: 250k atoms, RDF 0..1.2 nm in 1200 bins, fused histogram (no pair list)
    n = 250_000
    box = synth.box_a(n)
    pos = [torch.from_numpy(synth.frame(n, box, f)).to(dev) for f in range(8)]
    bins = np.zeros(1200, np.uint64)
    eng.search_histogram(api.SEARCH_SINGLE, 1.2, 0.0, 1.2, 1200, pos[0], box=box, pbc=7)
    eng.synchronize()
    t0 = time.perf_counter()
    pairs = 0
    K = 40
    for s in range(K):
        bins, c = eng.search_histogram(api.SEARCH_SINGLE, 1.2, 0.0, 1.2, 1200, pos[s % 8], box=box, pbc=7, bins=bins)
        pairs += c
    eng.synchronize()
    dt = (time.perf_counter() - t0) / K
    print(json.dumps({"workload": "C4 250k-atom frame, radial histogram 1200 bins, fused (single pass, no pairs)",
                      "frames_per_s": 1.0 / dt, "ms_per_frame": dt * 1e3, "matom_pairs_per_s": pairs / K / dt / 1e6}))

    # same with the bins resident on the GPU and no per-frame round trip (frames queue up back to back)
    dbins = torch.zeros(1200, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    eng.search_histogram(api.SEARCH_SINGLE, 1.2, 0.0, 1.2, 1200, pos[0], box=box, pbc=7, bins=dbins, want_count=False)
    eng.synchronize()
    dbins.zero_(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(K):
        eng.search_histogram(api.SEARCH_SINGLE, 1.2, 0.0, 1.2, 1200, pos[s % 8], box=box, pbc=7, bins=dbins, want_count=False)
    eng.synchronize()
    dt2 = (time.perf_counter() - t0) / K
    same = bool(np.array_equal(dbins.cpu().numpy().astype(np.uint64), bins))
    print(json.dumps({"workload": "C4 same, bins resident in HBM, no per-frame round trip", "frames_per_s": 1.0 / dt2,
                      "ms_per_frame": dt2 * 1e3, "matom_pairs_per_s": pairs / K / dt2 / 1e6, "bins_equal_host_path": same}))

    # the same frames with an RDF range of 2.0 nm (800 atoms per cell: the lean kernel's BIG instance, blocks of 512 second-cell atoms)
    dbins2 = torch.zeros(1000, dtype=torch.int64, device=dev)
    eng.search_histogram(api.SEARCH_SINGLE, 2.0, 0.0, 2.0, 1000, pos[0], box=box, pbc=7, bins=dbins2, want_count=False)
    eng.synchronize()
    dbins2.zero_(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(K):
        eng.search_histogram(api.SEARCH_SINGLE, 2.0, 0.0, 2.0, 1000, pos[s % 8], box=box, pbc=7, bins=dbins2, want_count=False)
    eng.synchronize()
    dt3 = (time.perf_counter() - t0) / K
    pairs2 = int(dbins2.sum().item()) / K
    print(json.dumps({"workload": "C4 frames, RDF range 2.0 nm (cells of ~800 atoms), 1000 bins, bins resident in HBM", "frames_per_s": 1.0 / dt3,
                      "ms_per_frame": dt3 * 1e3, "pairs_per_frame": pairs2, "matom_pairs_per_s": pairs2 / dt3 / 1e6}))

    # ---- host round trip of the headline search (PCIe-inclusive, never the bench value)
    n = 1_000_000
    box = synth.box_a(n)
    hpos = synth.frame(n, box, 0)
    cnt = eng.search_count(api.SEARCH_SINGLE, 1.2, hpos, box=box, pbc=7)
    t0 = time.perf_counter()
    cnt = eng.search_count(api.SEARCH_SINGLE, 1.2, hpos, box=box, pbc=7)
    pairs_h, dist_h = eng.search_fill(cnt)
    dt = time.perf_counter() - t0
    print(json.dumps({"workload": "C2 with host buffers: 12 MB frame in, 4.3 GB pair list out to pageable host memory "
                                  "(pinned ring + host threads, csrc/hoststream.hpp)",
                      "s_per_frame": dt, "pairs": cnt, "host_GBps": cnt * 12 / dt / 1e9}))
    del pairs_h, dist_h
    from molar_amd._lib import check
    best = 1e9
    for _ in range(3):      # steady state: the caller's arrays are touched (page faults of fresh numpy arrays are the caller's)
        pairs_h = np.empty((cnt, 2), np.uint32); dist_h = np.empty(cnt, np.float32)
        pairs_h.fill(0); dist_h.fill(0)
        t0 = time.perf_counter()
        cnt = eng.search_count(api.SEARCH_SINGLE, 1.2, hpos, box=box, pbc=7)
        check(eng.lib.molar_hip_search_fill(eng.ctx, pairs_h.ctypes.data, dist_h.ctypes.data))
        best = min(best, time.perf_counter() - t0)
    print(json.dumps({"workload": "C2 with host buffers, caller's arrays already mapped (second and later frames of a loop)",
                      "s_per_frame": best, "pairs": cnt, "host_GBps": cnt * 12 / best / 1e9}))
    del pairs_h, dist_h
    i64 = np.zeros(cnt, np.uint64); j64 = np.zeros(cnt, np.uint64); d32 = np.zeros(cnt, np.float32)
    t0 = time.perf_counter()
    cnt = eng.search_count(api.SEARCH_SINGLE, 1.2, hpos, box=box, pbc=7)
    check(eng.lib.molar_hip_search_fill_usize(eng.ctx, i64.ctypes.data, j64.ctypes.data, d32.ctypes.data))
    dt = time.perf_counter() - t0
    print(json.dumps({"workload": "C2 with host buffers, (usize, usize, f32) columns as the Rust binding fills them: 12 B per "
                                  "pair over the link, widened to 20 B by the host threads",
                      "s_per_frame": dt, "pairs": cnt, "link_GBps": cnt * 12 / dt / 1e9, "host_GBps": cnt * 20 / dt / 1e9}))


if __name__ == "__main__":
    main()
