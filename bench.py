#!/usr/bin/env python
"""bench.py — headline benchmark of the MI355X engine (BASELINE.json configs[1] + [2]).

One step = one trajectory frame of the 1M-atom synthetic triclinic box ("box A"),
already resident in HBM, through the whole hot path:
    PBC neighbour search, cutoff 1.2 nm, pair list (u32,u32,f32) materialised in HBM in the
    reference's order            (distance_search_single_pbc, distance_search.rs:928-954)
  + Kabsch fit of the 100k-atom selection (every 10th atom) onto frame 0, apply_transform, rmsd,
    centre of mass, gyration     (measure.rs:485-570, modify.rs:32-36; comparison_small.rs:14-25)
Frames shard embarrassingly over ranks (one process per GPU, weak scaling: each rank owns K frames);
the only collective is the end-of-run reduction of the pair count / RMSD sum (RCCL all_reduce).

Usage:  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload search_fit|rdf|membrane]
        N>1 from a bare shell: the script re-launches itself as N ranks through torch.distributed.run
        (127.0.0.1 rendezvous); under torch.distributed.run it uses the RANK/LOCAL_RANK/WORLD_SIZE it finds.
        --workload rdf is BASELINE.json configs[3] in the same frame-sharded shape: 250k-atom frames, fused
        1200-bin radial histogram with the bins resident in HBM, ONE all_reduce of 1200 x int64 at the end.
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NATOMS = 1_000_000
CUTOFF = 1.2
SEL_STRIDE = 10
HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def make_frames(nframes, rank, box, device):
    """Synthetic frames on the GPU: uniform fractional coordinates through the box matrix (seed
    20240607, shared by all frames = the 'topology'), plus per-frame Gaussian jitter sigma 0.05 nm."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(20240607)
    frac = torch.rand((NATOMS, 3), generator=g, device=device, dtype=torch.float64)
    M = torch.from_numpy(box.astype(np.float64)).to(device)
    base = frac @ M.T
    frames = torch.empty((nframes, NATOMS, 3), device=device, dtype=torch.float32)
    for f in range(nframes):
        g.manual_seed(20240607 + 1 + rank * 100003 + f)
        jit = torch.randn((NATOMS, 3), generator=g, device=device, dtype=torch.float32) * 0.05
        frames[f] = (base + jit.double()).float()
    ref = base.float().contiguous()
    torch.cuda.synchronize()      # made on torch's stream; the engine reads them on its own
    return frames, ref


def cpu_baseline(frame0, ref, box, mass, idx, warmup=5, reps=10):
    """The oracle (a C restatement of MolAR's algorithm, NOT the Rust binary) in the reference's schedule — serial
    grid + plan, thread pool over plan entries (never fewer than 3 per task, one result arena per thread), ordered
    concatenation, serial Measure passes — on one frame of the same workload.  BASELINE.md §3 protocol, bounded:
    `warmup` untimed + `reps` timed repetitions on all host cores, mean and standard deviation; beside it the same search
    with 8 threads and with 1 thread on smaller boxes of the same density and cutoff (scaled by atom count), so that the
    line says at which thread count this host does best (`cpu_threads_best`).  `value` is the best of the three.
    Only the native calls are inside the clock: the search result stays on the C side and is freed after the clock
    stops (Oracle.time_search_single_pbc); the Measure calls return scalars."""
    from oracle.oracle import Oracle
    from molar_amd import synth
    ncores = os.cpu_count() or 1
    o = Oracle("f32")
    ob = o.box_from_matrix(box)
    mem_gb = 0.0
    try:
        import psutil
        mem_gb = psutil.virtual_memory().available / 2 ** 30
    except Exception:
        pass
    n = NATOMS
    sample = f"one 1M-atom frame of the same workload, {warmup} warm-up + {reps} timed repetitions on {ncores} threads"
    pos = frame0
    if mem_gb and mem_gb < 40:     # 3.6e8 pairs x 20 B x 2 (per-thread arenas + ordered concat)
        n = 250_000
        sample = (f"250k-atom sub-box at the same density and cutoff (host RAM < 40 GB), scaled by atom count, "
                  f"{warmup} warm-up + {reps} timed repetitions on {ncores} threads")
        box = synth.box_a(n)
        ob = o.box_from_matrix(box)
        pos = synth.frame(n, box, 0)
        ref = synth.frame(n, box, 1)
        mass = mass[:n]
        idx = idx[idx < n]
    scale = NATOMS / n
    search_s, fit_s, npairs = [], [], 0
    for r in range(warmup + reps):
        dt, npairs = o.time_search_single_pbc(CUTOFF, pos, ob, 7, nthreads=ncores)
        t1 = time.perf_counter()
        R, t = o.fit_transform(pos, mass, ref, mass, idx, idx)
        moved = o.apply_transform(pos, R, t, idx)
        o.rmsd(moved, ref, idx, idx)
        o.center_of_mass(moved, mass, idx)
        o.gyration(moved, mass, idx)
        t2 = time.perf_counter()
        if r >= warmup:
            search_s.append(dt * scale)
            fit_s.append((t2 - t1) * scale)
    per_frame = np.array(search_s) + np.array(fit_s)
    fps = 1.0 / per_frame
    # the same search at 8 threads and at 1 thread (bounded: smaller boxes, 1 warm-up + best of 2), seconds per 1M-atom frame
    sweep = {str(ncores): float(np.mean(search_s))}
    for nt, nn in ((8, 250_000), (1, 62_500)):
        if nt >= ncores:
            continue
        b2 = synth.box_a(nn)
        p2 = synth.frame(nn, b2, 0)
        ob2 = o.box_from_matrix(b2)
        o.time_search_single_pbc(CUTOFF, p2, ob2, 7, nthreads=nt)
        sweep[str(nt)] = min(o.time_search_single_pbc(CUTOFF, p2, ob2, 7, nthreads=nt)[0] for _ in range(2)) * (NATOMS / nn)
    best_nt = min(sweep, key=sweep.get)
    fit_mean = float(np.mean(fit_s))
    best_frame_s = sweep[best_nt] + fit_mean
    return {
        "value": float(1.0 / best_frame_s), "unit": "frames/s", "cores": int(best_nt), "kind": "port",
        "sample": sample,
        "cpu_threads_best": int(best_nt), "host_cores": ncores,
        "search_seconds_per_frame_by_threads": sweep,
        "schedule": "serial grid + plan; dynamic chunks of 3 plan entries over min(threads, entries/3, evaluations/5e5) threads, "
                    "one result arena per thread, ordered parallel concatenation (oracle/molar_oracle.c run_plan)",
        "all_cores": {"value": float(1.0 / per_frame.mean()), "value_std": float(fps.std(ddof=1)) if len(fps) > 1 else 0.0,
                      "seconds_per_frame_mean": float(per_frame.mean()),
                      "seconds_per_frame_std": float(per_frame.std(ddof=1)) if len(per_frame) > 1 else 0.0, "cores": ncores},
        "search_s": float(sweep[best_nt]), "fit_s": fit_mean,
        "matom_pairs_per_sec": npairs * scale / float(sweep[best_nt]) / 1e6,
        "timed_span": "native oracle calls only (no result copies, no frees)",
    }


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args, argv):
    """`python bench.py --gpus N` from a bare shell (no WORLD_SIZE): become the launcher of N ranks, one per GPU,
    through torch.distributed.run with a 127.0.0.1 rendezvous, and exit with its status.  A node with fewer than N
    GPUs is refused here, with a message, before anything is spawned."""
    import subprocess
    if args.backend == "nccl" and not args.share_gpu:
        import torch
        have = torch.cuda.device_count()
        if have < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: this node exposes {have} GPU(s); one rank per GPU is required "
                             f"(run with --gpus {max(have, 1)} or on a node with {args.gpus} GPUs)")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__), *argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this driver (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def launch_check(args, rank, world):
    """Launch + collective plumbing without the GPU work (CPU test of the self-launch path, gloo): every rank
    contributes rank+1 pairs and a 1200-bin vector through the same reductions the workloads use."""
    from molar_amd.distributed import max_over_ranks, reduce_counts
    if args.workload == "rdf" and args.source == "xtc":
        # the frame supply of `--workload rdf --source xtc` without the GPU: rank 0 writes the trajectory with the library's
        # XTC writer, every rank decodes its own block on host threads (into host memory) and the ranks reduce a checksum
        import torch.distributed as dist
        from molar_amd import build
        build.build_library()
        n = 2000
        path, nframes, base = xtc_trajectory(args, rank, world, n, args.steps)
        if world > 1:
            dist.barrier()
        from molar_amd.distributed import shard_frames
        from molar_amd.xtc import XtcReader
        rd = XtcReader(path, nthreads=args.decode_threads or 2)
        mine = shard_frames(len(rd), rank, world)
        got = rd.read_frames(mine.start, len(mine))
        ok = all(np.array_equal(got[k], base[(mine.start + k) % len(base)]) for k in range(len(mine)))
        tot = reduce_counts([len(mine), int(ok)])
        if world > 1:
            dist.barrier()
        if rank == 0:
            os.remove(path)
            print(json.dumps({"launch_check": True, "source": "xtc", "n_gpus": world, "backend": args.backend, "frames_in_file": nframes,
                              "frames_decoded": int(tot[0]), "ranks_with_exact_frames": int(tot[1]), "frames_per_gpu": args.steps}))
        return
    bins = np.full(1200, rank + 1, np.int64)
    tot = reduce_counts(bins)
    pairs = int(reduce_counts([rank + 1])[0])
    t = max_over_ranks(float(rank + 1))
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "backend": args.backend, "pairs": pairs,
                          "bins_sum": int(tot.sum()), "max_over_ranks": t, "frames_per_gpu": args.steps}))


def xtc_trajectory(args, rank, world, natoms, frames_per_rank):
    """Rank 0 writes the synthetic trajectory of `--source xtc`: world x frames_per_rank frames of `natoms` atoms in box A,
    four distinct frames cycled, compressed by the library's own XTC writer (molar_hip_xtc_encode_frame).  Returns (path,
    number of frames, the four frames as the decoder will return them: on the format's 0.001 nm grid)."""
    from molar_amd import synth
    from molar_amd.xtc import encode_frame
    path = args.xtc_path or os.path.join(os.environ.get("TMPDIR", "/tmp"), f"molar_amd_bench_{os.getppid()}_{world}x{frames_per_rank}x{natoms}.xtc")
    box = synth.box_a(natoms)
    frames = [synth.frame(natoms, box, f) for f in range(4)]
    prec = np.float32(1000.0)
    grid = [(np.where(f * prec >= 0, f * prec + np.float32(0.5), f * prec - np.float32(0.5)).astype(np.int32).astype(np.float32) * (np.float32(1.0) / prec))
            for f in frames]
    total = world * frames_per_rank
    if rank == 0:
        box9 = np.ascontiguousarray(box.T, np.float32).reshape(9)      # the file stores the matrix column by column
        blobs = [encode_frame(f, box9, step=k, time=float(k)) for k, f in enumerate(frames)]
        with open(path + ".tmp", "wb") as fh:
            for k in range(total):
                fh.write(blobs[k % 4])
        os.replace(path + ".tmp", path)
    return path, total, grid


def run_rdf_xtc(args, rank, local_rank, world, device, cdev):
    """BASELINE.json configs[3] as stated: an XTC trajectory of 250k-atom frames sharded over the ranks (contiguous blocks,
    molar_amd.distributed.shard_frames), every rank decoding its block - host threads = cores / ranks through pinned staging,
    or one lane per frame on the GPU - into two windows of frames in HBM on a SECOND engine context, while the fused
    histogram consumes the window decoded before; bins stay on the GPU, ONE all_reduce of 1200 x int64 at the end.  The
    frame supply and the consumer overlap: the slower of the two sets frames/s, and the line says which (`binding_side`).
    Reference path: molar/src/io/xtc_handler.rs:64-112 (read_state), io.rs:198-271 (the state iterator), analysis_task.rs:202-267."""
    import torch
    import torch.distributed as dist
    from concurrent.futures import ThreadPoolExecutor
    from molar_amd import api, build, synth
    from molar_amd.distributed import max_over_ranks, reduce_counts, shard_frames, gather_float64
    from molar_amd.xtc import XtcReader
    build.build_library()
    n, nbins, K, W = 250_000, 1200, args.steps, args.warmup
    box = synth.box_a(n)
    path, nframes, grid = xtc_trajectory(args, rank, world, n, K + W)
    if world > 1:
        dist.barrier()
    eng = api.Engine(local_rank)
    dec = api.Engine(local_rank)                 # the decoder's context: its own stream and pinned staging
    threads = args.decode_threads or max(1, (os.cpu_count() or 8) // world)
    rd = XtcReader(path, engine=dec, nthreads=threads)
    mine = shard_frames(len(rd), rank, world)
    win = args.xtc_window or (1024 if args.decoder == "device" else 16)
    win = max(1, min(win, K))
    bufs = [torch.empty((win, n, 3), dtype=torch.float32, device=device) for _ in range(2)]
    bins = torch.zeros(nbins, dtype=torch.int64, device=device)
    torch.cuda.synchronize()

    def decode(first, count, buf):
        t0 = time.perf_counter()
        if args.decoder == "device":
            rd.read_frames_device(first, count, buf[:count])
        else:
            rd.read_frames(first, count, out=buf[:count])          # returns when the frames are in HBM
        return time.perf_counter() - t0

    def consume(buf, count):
        t0 = time.perf_counter()
        if args.rdf_single_calls:
            for q in range(count):
                eng.search_histogram(api.SEARCH_SINGLE, CUTOFF, 0.0, CUTOFF, nbins, buf[q], box=box, pbc=7, bins=bins, want_count=False)
        else:
            eng.search_histogram_frames(api.SEARCH_SINGLE, CUTOFF, 0.0, CUTOFF, nbins, buf[:count], box=box, pbc=7, bins=bins)
        eng.synchronize()                                           # this window's buffer is free again
        return time.perf_counter() - t0

    def barrier():
        if world > 1:
            dist.barrier()
        eng.synchronize()
        torch.cuda.synchronize()

    # warm-up: the rank's first W frames through the same two steps (buffers grow, the library loads its kernels)
    done = 0
    while done < W:
        k = min(win, W - done)
        decode(mine.start + done, k, bufs[0])
        consume(bufs[0], k)
        done += k
    barrier()
    bins.zero_()
    torch.cuda.synchronize()
    first = mine.start + W
    windows = [(first + f, min(win, K - f)) for f in range(0, K, win)]
    pool = ThreadPoolExecutor(1)
    t_dec = t_con = 0.0
    t0 = time.perf_counter()
    fut = pool.submit(decode, windows[0][0], windows[0][1], bufs[0]) if windows else None
    for w, (f, k) in enumerate(windows):
        t_dec += fut.result()
        if w + 1 < len(windows):
            fut = pool.submit(decode, windows[w + 1][0], windows[w + 1][1], bufs[(w + 1) % 2])      # overlaps with the launches below
        t_con += consume(bufs[w % 2], k)
    barrier()
    elapsed = time.perf_counter() - t0
    total_bins = reduce_counts(bins.cpu().numpy(), device=cdev)       # the only collective
    t = max_over_ranks(elapsed, device=cdev)
    sides = gather_float64(np.array([K / max(t_dec, 1e-9), K / max(t_con, 1e-9), K / elapsed]), device=cdev)
    from molar_amd.distributed import collective_view
    comm = collective_view(local_rank) if world > 1 else None
    if rank == 0:
        check = None
        if args.verify:       # the same frames held resident (decoded to the format's grid), on one fresh context
            e2 = api.Engine(local_rank)
            per = []
            for gfr in grid:
                chk = torch.zeros(nbins, dtype=torch.int64, device=device)
                fr = torch.from_numpy(gfr).to(device)
                torch.cuda.synchronize()
                e2.search_histogram(api.SEARCH_SINGLE, CUTOFF, 0.0, CUTOFF, nbins, fr, box=box, pbc=7, bins=chk, want_count=False)
                e2.synchronize()
                per.append(chk.cpu().numpy())
            want = np.zeros(nbins, np.int64)
            for r in range(world):
                blk = shard_frames(nframes, r, world)
                for fr in range(blk.start + W, blk.start + W + K):
                    want += per[fr % 4]
            check = bool(np.array_equal(want, total_bins))
        pairs = float(total_bins.sum())
        dec_fps, con_fps = [float(s_[0]) for s_ in sides], [float(s_[1]) for s_ in sides]
        line = ({
            "metric": "frames/sec, 250k-atom XTC frames -> decode -> HBM -> fused 1200-bin radial distance histogram, bins reduced over ranks",
            "value": K * world / t, "unit": "frames/s", "pairs_binned_per_sec": pairs / t,
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": t / K * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C4 as stated: synthetic XTC trajectory of 250k-atom frames (box A, written by the library's XTC writer), "
                                   "each rank decodes its contiguous block into double-buffered HBM windows on a second context and feeds the "
                                   "fused Histogram1D binning (rc = 1.2 nm, 1200 bins of 0.001 nm); one all_reduce of 1200 x int64",
                       "natoms": n, "nbins": nbins, "frames_per_gpu": K, "frames_in_file": nframes, "pairs_per_frame": pairs / (K * world),
                       "source": "xtc", "decoder": args.decoder, "decode_threads_per_rank": threads if args.decoder == "host" else None,
                       "window_frames": win, "host_cores": os.cpu_count()},
            # each rank's two sides over the time that side was busy (they overlap: the slower one sets the rank's frames/s)
            "decode_fps": dec_fps, "consumer_fps": con_fps, "per_rank_fps": [float(s_[2]) for s_ in sides],
            "binding_side": ["decode" if d < c_ else "histogram" for d, c_ in zip(dec_fps, con_fps)],
            "collective": comm,
            "roofline": None,
            "reduced_bins_equal_resident_frames": check,
        })
        try:
            os.remove(path)
        except OSError:
            pass
        line["_failed"] = check is False
        return line
    return None


def run_rdf(args, rank, local_rank, world, device, cdev):
    """BASELINE.json configs[3] shape: each rank owns K 250k-atom frames (resident in HBM), every frame goes through
    the fused search + Histogram1D binning (molar_membrane/src/stats.rs:29-35) into int64 bins resident on the GPU
    (no per-frame round trip), and ONE all_reduce of 1200 x int64 (RCCL) combines the ranks at the end."""
    import torch
    import torch.distributed as dist
    from molar_amd import api, build, synth
    from molar_amd.distributed import max_over_ranks, reduce_counts
    build.build_library()
    n, nbins, K, W = 250_000, 1200, args.steps, args.warmup
    box = synth.box_a(n)
    g = torch.Generator(device=device)
    g.manual_seed(20240607)
    base = torch.rand((n, 3), generator=g, device=device, dtype=torch.float64) @ torch.from_numpy(box.astype(np.float64)).to(device).T
    nres = min(K, 64)
    frames = torch.empty((nres, n, 3), device=device, dtype=torch.float32)
    for f in range(nres):
        g.manual_seed(20240607 + 1 + rank * 100003 + f)
        frames[f] = (base + (torch.randn((n, 3), generator=g, device=device, dtype=torch.float32) * 0.05).double()).float()
    eng = api.Engine(local_rank)
    bins = torch.zeros(nbins, dtype=torch.int64, device=device)
    torch.cuda.synchronize()      # frames and bins were made on torch's stream; the engine works on its own

    def run(first, count):
        if args.rdf_single_calls:
            for s in range(count):
                eng.search_histogram(api.SEARCH_SINGLE, CUTOFF, 0.0, CUTOFF, nbins, frames[(first + s) % nres], box=box, pbc=7,
                                     bins=bins, want_count=False)
            return
        # the resident frames as blocks of the trajectory (molar_hip_search_histogram_frames: groups of up to 8 frames share
        # their launches); the ring of nres resident frames is walked in contiguous pieces
        s = 0
        while s < count:
            f0 = (first + s) % nres
            k = min(count - s, nres - f0)
            eng.search_histogram_frames(api.SEARCH_SINGLE, CUTOFF, 0.0, CUTOFF, nbins, frames[f0:f0 + k], box=box, pbc=7, bins=bins)
            s += k

    def barrier():
        if world > 1:
            dist.barrier()
        eng.synchronize()
        torch.cuda.synchronize()

    run(0, W)
    barrier()
    bins.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(W, K)
    barrier()
    elapsed = time.perf_counter() - t0
    timed_bins = bins.cpu().numpy().copy()
    # per-kernel event times from a second, untimed pass over the same frames: an event record is a barrier packet on the stream
    # (~5-10 us each, five per frame), so the timed region carries none
    eng.profile_enable(True)
    eng.profile_read()
    KP = min(K, max(args.profile_steps, 1))
    run(W, KP)
    barrier()
    prof = eng.profile_read()
    eng.profile_enable(False)
    total_bins = reduce_counts(timed_bins, device=cdev)       # the only collective
    t = max_over_ranks(elapsed, device=cdev)
    from molar_amd.distributed import collective_view, gather_float64 as _gf
    comm = collective_view(local_rank) if world > 1 else None
    per_rank = [float(v[0]) for v in _gf([K / elapsed], device=cdev)]
    if rank == 0:
        check = None
        if args.verify:       # rank 0 recomputes every rank's frames alone: the reduced bins must be identical
            e2 = api.Engine(local_rank)
            chk = torch.zeros(nbins, dtype=torch.int64, device=device)
            for r in range(world):
                for s in range(K):
                    f = (W + s) % nres
                    g.manual_seed(20240607 + 1 + r * 100003 + f)
                    fr = (base + (torch.randn((n, 3), generator=g, device=device, dtype=torch.float32) * 0.05).double()).float()
                    torch.cuda.synchronize()       # torch made `fr` (and zeroed `chk`) on ITS stream; the engine reads on its own
                    e2.search_histogram(api.SEARCH_SINGLE, CUTOFF, 0.0, CUTOFF, nbins, fr, box=box, pbc=7, bins=chk, want_count=False)
                    e2.synchronize()
            check = bool(np.array_equal(chk.cpu().numpy(), total_bins))
        pairs = float(total_bins.sum())
        hist_ms, hist_n = prof["pair_fill"]
        hist_launches = hist_n
        hist_n = KP          # per FRAME: a launch of the frames form carries up to 8 frames
        # algorithmic work of one frame: the plan's candidate evaluations, 13.5 * N * mean cell
        # population (13 neighbour cells in full, the own cell as a triangle), 9 flop each; the kernels skip part of
        # them by bounding boxes, the reference evaluates all
        ext = np.asarray(box, np.float32).sum(axis=1)              # get_lab_extents: row sums (periodic_box.rs:369-375)
        gd = np.maximum(np.floor(ext / np.float32(CUTOFF)), 1).astype(np.int64)     # Grid::from_cutoff_and_extents (:103-110)
        ncell = max(int(gd[0]) * int(gd[1]) * int(gd[2]), 1)
        cand = 13.5 * n * (n / ncell)
        valu_peak = 157.3
        # what the kernels really evaluate: the lean kernel skips the (row, 64-atom chunk) steps its bounding boxes rule out.
        # Counted by a debug build in a separate run (tools/hist_wave_times.py -> profiles/hist_steps.json), like roofline.traffic.
        executed = None
        try:
            executed = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "hist_steps.json")))["executed_candidate_evaluations"]
        except Exception:
            pass
        line = ({
            "metric": "frames/sec, 250k-atom frames -> fused 1200-bin radial distance histogram, bins reduced over ranks",
            "value": K * world / t, "unit": "frames/s", "pairs_binned_per_sec": pairs / t,
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": t / K * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C4 shape: 250k-atom synthetic triclinic box A frames resident in HBM, rc=1.2 nm, fused "
                                   "Histogram1D binning (1200 bins of 0.001 nm), frames sharded over ranks, one all_reduce "
                                   "of 1200 x int64", "natoms": n, "nbins": nbins, "frames_per_gpu": K,
                       "call_form": "one call per frame" if args.rdf_single_calls else "molar_hip_search_histogram_frames over blocks of the resident frames (groups of <= 8 frames per launch)",
                       "pairs_per_frame": pairs / (K * world)},
            "per_rank_fps": per_rank, "collective": comm,
            "kernel_ms_per_frame": {k: v[0] / KP for k, v in prof.items()},
            "kernel_ms_source": f"HIP events on the engine's stream in a separate untimed pass over {KP} of the same frames",
            "roofline": {"kernel": "hist_plan_kernel + hist_kernel<SINGLE> + pair_kernel<SINGLE,HIST>", "bound": "valu",
                         "note": "12*N + 8*nbins bytes per frame: not an HBM-bound path (SURVEY.md 8d); priced as the plan's "
                                 "candidate evaluations x 9 flop against the fp32 vector peak",
                         "achieved": cand * 9 / (hist_ms / max(hist_n, 1) * 1e-3) / 1e12, "peak": valu_peak, "unit": "TFLOP/s",
                         "frac": cand * 9 / (hist_ms / max(hist_n, 1) * 1e-3) / 1e12 / valu_peak, "traffic": None,
                         "frac_on_executed_evaluations": (executed * 9 / (hist_ms / max(hist_n, 1) * 1e-3) / 1e12 / valu_peak) if executed else None,
                         "executed_candidate_evals_per_frame": executed,
                         "executed_source": "profiles/hist_steps.json: steps counted by a -DMOLAR_HIP_DEBUG_KNOBS build on this workload's frame (separate run)",
                         "avg_launch_ms": hist_ms / max(hist_n, 1), "avg_launch_ms_is": "histogram kernels' time per FRAME",
                         "launches_in_profile_pass": hist_launches, "frames_in_profile_pass": KP, "grid_dims": [int(x) for x in gd],
                         "candidate_evals_per_frame": cand, "candidate_evals_per_sec": cand * K * world / t},
            "reduced_bins_equal_single_rank": check,
        })
        line["_failed"] = check is False
        return line
    return None


def run_membrane(args, rank, local_rank, world, device, cdev):
    """BASELINE.json configs[4] shape: each rank owns K frames of the 500k-atom bilayer (4000 lipids, resident in HBM).  A
    frame is one chained call (molar_hip_membrane_frame_*: unwrap, markers, patches with rc 2.5 nm, initial normals, one
    smoothing pass, Scd of the 8000 tails), two frames in flight; the per-lipid results (flags, normals, curvatures, areas,
    vertex counts, order parameters) come to the host with the end of every frame (molar_hip_membrane_frame_end_fetch) and are
    accumulated there like LipidGroup::frame_update does, while the GPU runs the next frame; ONE all_reduce of the accumulated sums combines the ranks at the end."""
    import torch
    import torch.distributed as dist
    from molar_amd import api, build
    from molar_amd import membrane as mb
    from molar_amd.distributed import max_over_ranks, reduce_counts
    build.build_library()
    K, W = args.steps, args.warmup
    xyz, box, first, tpl, masses = mb.build_bilayer(2000, 500_000)
    nl = len(first)
    pbox = api.PeriodicBox.from_matrix(box)
    small = ["valid", "normals", "mean_curv", "gauss_curv", "area", "nvert", "order"]

    def frame_of(r, f):
        rng = np.random.default_rng(20240607 + 1 + r * 100003 + f)
        return (xyz + rng.normal(0, 0.02, xyz.shape).astype(np.float32)).astype(np.float32)

    nres = min(K + W, 16)
    src = [torch.from_numpy(frame_of(rank, f)).to(device) for f in range(nres)]

    def trajectory(eng, frames_dev, count, first_frame=0, stride=1, offset=0):
        """frames first_frame + offset, + offset + stride, ... (< first_frame + count) through one plan, two in flight; returns
        (per-frame rows of sums in float64 - integers exactly -, seconds)."""
        m = mb.Membrane(eng, len(xyz), first, tpl, masses, mb.MembraneOptions(cutoff=2.5, order_type=1))
        plan = m._plan()
        plan.set_valid(None)
        norder = plan.norder // nl
        ids = list(range(offset, count, stride))
        rows = np.zeros((len(ids), 4 + norder), np.float64)   # valid lipid-frames, vertices, area, |mean curvature|, order per carbon

        def sums(r, k):
            ok = r["valid"].astype(bool)
            rows[k, 0] = int(ok.sum())
            rows[k, 1] = int(r["nvert"][ok].sum())
            rows[k, 2] = float(r["area"][ok].sum(dtype=np.float64))
            rows[k, 3] = float(np.abs(r["mean_curv"][ok]).sum(dtype=np.float64))
            rows[k, 4:] = r["order"].reshape(nl, norder)[ok].sum(axis=0, dtype=np.float64)

        bufs = [frames_dev[(first_frame + s) % len(frames_dev)].clone() for s in ids]     # unwrapped in place: fresh copies
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        prev = None          # (ticket, row) of the frame in flight
        if os.environ.get("BENCH_C5_SUMS", "thread" if S == 1 else "defer") == "thread":
            # one context: the host sums of a frame run on a thread of their own (numpy releases the interpreter lock inside them),
            # beside the calls that feed the GPU.  Several contexts (a feeding thread each already): the sums wait until the next
            # frame has been begun - more threads only fight for the interpreter (measured: 3.7 k against 5.3 k frames/s on four)
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=1) as acc:
                jobs = []
                for k in range(len(ids)):
                    t = plan.begin(bufs[k], pbox)
                    if prev is not None:
                        jobs.append(acc.submit(sums, plan.end(prev[0], small)[1], prev[1]))   # the arrays come with the end: one wait
                    prev = (t, k)
                if prev is not None:
                    jobs.append(acc.submit(sums, plan.end(prev[0], small)[1], prev[1]))
                for j in jobs:
                    j.result()
        else:
            got = None           # (arrays, row) of the frame that ended last: summed while the GPU runs the next frame
            for k in range(len(ids)):
                t = plan.begin(bufs[k], pbox)
                if got is not None:
                    sums(*got)
                    got = None
                if prev is not None:
                    got = (plan.end(prev[0], small)[1], prev[1])
                prev = (t, k)
            if got is not None:
                sums(*got)
            if prev is not None:
                sums(plan.end(prev[0], small)[1], prev[1])
        eng.synchronize()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        plan.close()
        return rows, ids, dt

    S = max(1, args.streams)
    engines = [api.Engine(local_rank) for _ in range(S)]
    eng = engines[0]

    def run_frames(count, first_frame=0):
        """`count` frames over the S engine contexts (context s takes frames s, s + S, ...: one host thread, one stream and one
        chained plan each - the frames of a trajectory are independent, and a frame is ~35 latency-bound launches that leave
        most of the chip idle); sums accumulated in FRAME order, so the result does not depend on S."""
        if S == 1:
            rows, ids, dt = trajectory(eng, src, count, first_frame)
            parts = [(rows, ids)]
        else:
            from concurrent.futures import ThreadPoolExecutor
            t0 = time.perf_counter()
            with ThreadPoolExecutor(S) as pool:
                res = list(pool.map(lambda s_: trajectory(engines[s_], src, count, first_frame, S, s_), range(S)))
            dt = time.perf_counter() - t0
            parts = [(r[0], r[1]) for r in res]
        per_frame = [None] * count
        for rows, ids in parts:
            for k, f in enumerate(ids):
                per_frame[f] = rows[k]
        acc = np.zeros(per_frame[0].shape if count else 4, np.float64)
        for row in per_frame:
            acc += row
        return acc, dt

    def barrier():
        if world > 1:
            dist.barrier()
        for e in engines:
            e.synchronize()
        torch.cuda.synchronize()

    t_pre = time.perf_counter()
    while args.preheat > 0 and time.perf_counter() - t_pre < args.preheat:
        run_frames(16)
    run_frames(max(W, 1))
    barrier()
    acc, elapsed = run_frames(K, first_frame=W)
    barrier()
    # sums over ranks: the integer entries exactly, the float sums in rank order on rank 0
    from molar_amd.distributed import gather_float64
    parts = gather_float64(acc, device=cdev)
    t = max_over_ranks(elapsed, device=cdev)
    from molar_amd.distributed import collective_view
    comm = collective_view(local_rank) if world > 1 else None
    per_rank = [float(v[0]) for v in gather_float64([K / elapsed], device=cdev)]
    if rank == 0:
        total = np.sum(np.stack(parts), axis=0)
        check = None
        if args.verify:        # rank 0 alone: every rank's frames through a fresh plan, stage by stage instead of chained
            e2 = api.Engine(local_rank)
            chk = []
            for r in range(world):
                fr = [torch.from_numpy(frame_of(r, f)).to(device) for f in range(nres)]
                m2 = mb.Membrane(e2, len(xyz), first, tpl, masses, mb.MembraneOptions(cutoff=2.5, order_type=1, fused=False))
                a2 = np.zeros_like(acc)
                norder = (len(acc) - 4)
                for s in range(K):
                    work = fr[(W + s) % nres].clone()
                    torch.cuda.synchronize()       # the copy runs on torch's stream, the engine on its own
                    res = m2.compute(work, box)
                    ok = res["valid"].astype(bool)
                    a2[0] += int(ok.sum()); a2[1] += int(res["nvert"][ok].sum())
                    a2[2] += float(res["area"][ok].sum(dtype=np.float64)); a2[3] += float(np.abs(res["mean_curv"][ok]).sum(dtype=np.float64))
                    a2[4:] += np.concatenate(res["order"], axis=1)[ok].sum(axis=0, dtype=np.float64)
                chk.append(a2)
            check = bool(np.array_equal(np.sum(np.stack(chk), axis=0), total))
        nvalid = total[0]
        line = ({
            "metric": "frames/sec, 500k-atom bilayer (4000 lipids): per-lipid order parameters + neighbour analysis per frame, sums reduced over ranks",
            "value": K * world / t, "unit": "frames/s", "lipid_frames_per_sec": nl * K * world / t,
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": t / K * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C5 shape: 500k-atom synthetic bilayer frames resident in HBM, 2 x 2000 lipids of 52 atoms: unwrap, "
                                   "head / mid / tail markers, patches (rc 2.5 nm), initial normals, one smoothing pass (quadric fit, "
                                   "Voronoi cell, curvatures, area), Scd of 8000 tails; one chained call per frame, two frames in flight per engine context, "
                                   "per-lipid results fetched and accumulated on the host every frame; frames sharded over ranks, one "
                                   "gather of the accumulated sums", "natoms": len(xyz), "nlipids": nl, "frames_per_gpu": K, "engine_contexts_per_gpu": S},
            "results": {"valid_lipid_frames": int(nvalid), "mean_vertices": total[1] / max(nvalid, 1), "mean_area_nm2": total[2] / max(nvalid, 1),
                        "mean_abs_mean_curvature": total[3] / max(nvalid, 1), "mean_abs_scd": float(np.abs(total[4:] / max(nvalid, 1)).mean())},
            "per_rank_fps": per_rank, "collective": comm,
            "roofline": None,
            "roofline_note": "no roofline claim: ~35 latency-bound launches per frame over 4000 lipids (the largest, the per-lipid fit, runs "
                             "63 waves for 0.11 ms) and a serial host pass of 0.2 ms hidden behind them; profiles/r03_membrane_frame_kernel_stats.csv",
            "sums_equal_stage_by_stage_single_rank": check,
        })
        line["_failed"] = check is False
        return line
    return None


def secondary_legs(args, rank, local_rank, world, device, cdev):
    """BASELINE.json configs[3] and [4] beside the headline, on the same clock: `--workload rdf` (resident 250k-atom frames,
    fused 1200-bin histogram), the same fed from a synthetic XTC file (`--source xtc`: configs[3] as stated) and `--workload
    membrane` (500k-atom bilayer, 4000 lipids) as short legs with --verify on
    (rdf: the reduced bins against every frame recomputed alone; membrane: the accumulated sums against the stage-by-stage
    calls).  Returns {"rdf": line, "membrane": line, "seconds": ...}; an exception inside a leg is reported as {"error": ...}."""
    import copy
    import traceback
    out = {}
    t0 = time.perf_counter()
    for name, fn, over in (("rdf", run_rdf, dict(workload="rdf", steps=args.secondary_rdf_steps, warmup=32, verify=True, profile_steps=64)),
                           # C4 as BASELINE.json states it: XTC file -> decoder threads -> HBM windows -> fused histogram (frames form)
                           ("rdf_xtc", run_rdf_xtc, dict(workload="rdf", source="xtc", steps=args.secondary_xtc_steps, warmup=32, verify=True,
                                                         xtc_window=16, decoder="host", decode_threads=0, xtc_path="")),
                           ("membrane", run_membrane, dict(workload="membrane", steps=args.secondary_membrane_steps, warmup=8, verify=True,
                                                           preheat=min(args.preheat, 1.0), streams=4))):
        a2 = copy.copy(args)
        for k, v in over.items():
            setattr(a2, k, v)
        t1 = time.perf_counter()
        try:
            ln = fn(a2, rank, local_rank, world, device, cdev)
        except Exception as exc:       # the headline line is still printed; the run exits with status 1
            ln = {"error": f"{type(exc).__name__}: {exc}", "traceback": traceback.format_exc()[-800:], "_failed": True}
        if ln is not None:
            ln["leg_seconds"] = time.perf_counter() - t1
            out[name] = ln
    out["seconds"] = time.perf_counter() - t0
    out["note"] = ("secondary legs, never `value`: the C4 and C5 shapes of BASELINE.json run after the headline in the same process, "
                   "fewer steps than `python bench.py --workload rdf|membrane` runs by default")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--preheat", type=float, default=1.5, help="seconds of untimed steps in front of the warm-up (clock ramp of a fresh box)")
    ap.add_argument("--profile-steps", type=int, default=50, help="steps of the separate per-kernel timing pass (at most --steps)")
    ap.add_argument("--serial-measure", action="store_true",
                    help="run the Kabsch fit/RMSD/COM/gyration of a frame after its search on the same stream instead of "
                         "concurrently on a second engine context (HIP stream) of the same GPU")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("MOLAR_BENCH_STREAMS", "0")),
                    help="engine contexts (HIP streams) per GPU working on different frames in the timed region.  1 (default): one "
                         "context, two frames in flight through molar_hip_search_resident_begin / _end, the next frame's grid built on "
                         "the side stream under the fill pass.  2: two contexts on alternate frames, one host thread each (until the "
                         "end of round 3 the default: +3..5 %% then; since the non-temporal result stores and the late grid start the "
                         "single context is as fast at 200 steps and 1.5 %% faster at 20).  The per-kernel event times and the roofline "
                         "come from a separate single-context pass either way.")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="one search at a time (molar_hip_search_resident) instead of the begin/end form that keeps two "
                         "frames queued on the engine's stream (kernels still run one after the other, in order)")
    ap.add_argument("--workload", choices=("search_fit", "rdf", "membrane"), default="search_fit",
                    help="search_fit: the headline (configs[1]+[2]); rdf: configs[3] shape (fused histogram + all_reduce)")
    ap.add_argument("--source", choices=("resident", "xtc"), default="resident",
                    help="rdf workload: where the frames come from.  resident: synthetic frames already in HBM.  xtc: BASELINE.json "
                         "configs[3] as stated - every rank decodes its own contiguous block of a synthetic 250k-atom XTC file into "
                         "double-buffered HBM windows on a second engine context while the fused histogram consumes the window before")
    ap.add_argument("--decoder", choices=("host", "device"), default="host",
                    help="--source xtc: host = decoder threads + pinned staging (molar_hip_xtc_read); device = one lane per frame on the "
                         "GPU (molar_hip_xtc_read_device), for hosts with few cores")
    ap.add_argument("--decode-threads", type=int, default=0, help="--source xtc --decoder host: decoder threads per rank (0 = host cores / ranks)")
    ap.add_argument("--xtc-window", type=int, default=0, help="--source xtc: frames per decode window (0 = 16 for the host decoder, 1024 for the device decoder)")
    ap.add_argument("--xtc-path", default="", help="--source xtc: where rank 0 writes the synthetic trajectory (default: a file under $TMPDIR or /tmp)")
    ap.add_argument("--rdf-single-calls", action="store_true",
                    help="rdf workload: one molar_hip_search_histogram call per frame (the form of rounds 2-5) instead of "
                         "molar_hip_search_histogram_frames over blocks of frames")
    ap.add_argument("--no-secondary", action="store_true", help="search_fit: skip the short C4 (rdf) and C5 (membrane) legs attached to the line as `secondary`")
    ap.add_argument("--secondary", action="store_true", help="search_fit: run the secondary legs at N > 1 as well (default: N = 1 only)")
    ap.add_argument("--secondary-rdf-steps", type=int, default=512)
    ap.add_argument("--secondary-membrane-steps", type=int, default=512)
    ap.add_argument("--secondary-xtc-steps", type=int, default=256, help="frames of the XTC-fed C4 leg (a synthetic 250k-atom trajectory of that many + 32 frames is written to $TMPDIR: ~1 MB per frame)")
    ap.add_argument("--no-pairs-only", action="store_true", help="search_fit: skip the extra leg that times the resident search with the (i, j) plane only")
    ap.add_argument("--verify", action="store_true",
                    help="rank 0 recomputes all ranks' frames alone and compares (rdf: the reduced bins; search_fit: every "
                         "frame's pair count and RMSD); a mismatch exits with status 1")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default=os.environ.get("MOLAR_BENCH_BACKEND", "nccl"),
                    help="torch.distributed backend; gloo only with --launch-check (CPU test of the launch path)")
    ap.add_argument("--launch-check", action="store_true", help="launch the ranks and run the collectives only (no GPU work)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="functional check of the N>1 code path on a box with fewer GPUs: ranks share the visible GPUs "
                         "(rank r uses GPU r mod count) and reduce over gloo; the line it prints is NOT a scaling result")
    args = ap.parse_args()
    if args.streams <= 0:         # default: one context for the search workloads; the membrane workload (a frame = ~35 latency-bound
        args.streams = 4 if args.workload == "membrane" else 1      # launches on 4000 lipids) runs four contexts on alternate frames
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}")

    import torch
    import torch.distributed as dist

    if args.launch_check:
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(args.backend, rank=rank, world_size=world)
        launch_check(args, rank, world)
        if world > 1:
            dist.destroy_process_group()
        return
    if args.share_gpu and torch.cuda.is_available() and torch.cuda.device_count() > 0:
        local_rank = local_rank % torch.cuda.device_count()
    if not torch.cuda.is_available() or torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py: rank {rank} needs GPU {local_rank}, but this process sees {torch.cuda.device_count()} "
                         "GPU(s); there is no CPU path")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    cdev = None if args.share_gpu else device          # where the end-of-run reductions run (gloo: host tensors)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    if args.workload in ("rdf", "membrane"):
        fn = run_membrane if args.workload == "membrane" else (run_rdf_xtc if args.source == "xtc" else run_rdf)
        line = fn(args, rank, local_rank, world, device, cdev)
        failed = bool(line.pop("_failed", False)) if line else False
        if line:
            print(json.dumps(line))
        if world > 1:
            dist.destroy_process_group()
        if failed:
            raise SystemExit(1)
        return

    from molar_amd import api, build, synth
    build.build_library()
    exit_code = 0
    # One context = one HIP stream + its own buffers.  Frames are independent, so S contexts work on
    # different frames at the same time: the tail of one frame's kernels (idle CUs) and its host
    # round trips (result count, fit scalars) are covered by the other frame's kernels.
    S = max(1, args.streams)
    engines = [api.Engine(local_rank) for _ in range(S)]
    eng = engines[0]
    # The two halves of a frame's work are independent: the search runs on the engine's stream, the fit / RMSD /
    # COM / gyration of the selection on a second context (its own HIP stream), fed by a persistent host thread, so
    # its ~0.1 ms of small launches and its host round trip hide behind the search kernels.
    overlap = not args.serial_measure
    m_engines = [api.Engine(local_rank) for _ in range(S)] if overlap else engines
    import queue
    import threading
    jobs = [queue.Queue() for _ in range(S)] if overlap else None
    done = [queue.Queue() for _ in range(S)] if overlap else None

    def measure_worker(k):
        torch.cuda.set_device(local_rank)      # the current device is per thread: without this rank r's worker would sync GPU 0
        while True:
            job = jobs[k].get()
            if job is None:
                return
            try:
                # the fit moves the selected atoms in place (apply_transform) while the search of the same frame
                # is reading it on the other stream: work on a copy (12 MB device-to-device)
                work[k].copy_(job, non_blocking=True)
                torch.cuda.current_stream(device).synchronize()
                done[k].put(m_engines[k].fit_rmsd_batch(work[k].unsqueeze(0), mass, ref, idx=idx, apply=True))
            except Exception as exc:           # surface failures in the main thread
                done[k].put(exc)

    workers = []
    work = None

    box = synth.box_a(NATOMS)
    K, W = args.steps, args.warmup
    nres = min(K, 64)                       # frames resident per rank (cycled if K is larger)
    frames, ref = make_frames(nres, rank, box, device)
    mass = torch.from_numpy(synth.masses(NATOMS)).to(device)
    idx_np = np.arange(0, NATOMS, SEL_STRIDE, dtype=np.int64)
    idx = torch.from_numpy(idx_np).to(device)
    torch.cuda.synchronize()

    if overlap:
        work = [torch.empty_like(frames[0]) for _ in range(S)]
        torch.cuda.synchronize()
        for k in range(S):
            th = threading.Thread(target=measure_worker, args=(k,), daemon=True)
            th.start()
            workers.append(th)

    # one search description per context, re-pointed at each frame (no per-step argument marshalling)
    # (two per context: the begin/end form keeps the description of a frame alive until its result is collected)
    descs = [[e.make_search_desc(api.SEARCH_SINGLE, CUTOFF, frames[0], box=box, pbc=7) for _ in range(2)] for e in engines]
    pipelined = not args.no_pipeline

    def step(e, f):
        """Enqueue the frame's fit on the measure context, run its search; the fit result is collected by
        run_steps (it has the whole search to finish in, so nothing waits on it inside the loop)."""
        k = engines.index(e)
        fr = frames[f % nres]
        if overlap:
            jobs[k].put(fr)
            descs[k][0][0].xyz1 = fr.data_ptr()
            cnt, pp_, dp_ = e.search_resident_desc(descs[k][0][0])       # count + scan + fill, one round trip
            return cnt, None, pp_, dp_
        cnt, _, _ = e.search_resident(api.SEARCH_SINGLE, CUTOFF, fr, box=box, pbc=7)
        out = e.fit_rmsd_batch(fr.unsqueeze(0), mass, ref, idx=idx, apply=True)
        return cnt, float(out["rmsd"][0])

    def collect_fits(k, count):
        """The `count` fit results (RMSD after the fit) of context k, in frame order; all of them are part of the timed work."""
        vals = []
        for _ in range(count):
            out = done[k].get()
            if isinstance(out, Exception):
                raise out
            vals.append(float(out["rmsd"][0]))
        return vals

    def barrier():
        if world > 1:
            dist.barrier()
        for e in set(engines) | set(m_engines):
            e.synchronize()
        torch.cuda.synchronize()

    fits = []
    last_results = []      # (frame number, pair count, pairs device address, distances device address) of the newest results

    def checksum_planes(cnt, pp, dp):
        """Order-sensitive 64-bit checksums of a result's pair plane and distance plane where they lie in HBM: sum of
        value_i * (2 i + 1) in wrapping int64 arithmetic over the (i, j) records as u64 words / the distances' bit patterns."""
        if cnt == 0:
            return 0, 0

        class _Dev:
            def __init__(self, ptr, n, typestr):
                self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}
        w = torch.arange(cnt, device=device, dtype=torch.int64) * 2 + 1
        pv = torch.as_tensor(_Dev(pp, cnt, "<i8"), device=device)
        a = int((pv * w).sum().item())
        b = 0
        if dp:                 # (the pairs-only mode has no distance plane)
            dv = torch.as_tensor(_Dev(dp, cnt, "<i4"), device=device).to(torch.int64)
            b = int((dv * w).sum().item())
            del dv
        del w, pv
        return a, b

    def run_steps(first, count):
        """`count` steps starting at frame `first`, dealt round-robin to the S contexts.  Every context runs its frames
        through the begin / end form (two of ITS frames in flight: the next one is enqueued - grid, plan, count, offsets,
        fill: ~14 launches - before the result of the one in front is waited for); with S > 1 one host thread per context
        (ctypes releases the GIL inside the library).  Every frame is begun AND ended in here."""
        res = [None] * count
        rms = [0.0] * count
        newest = [[] for _ in range(S)]       # per context: (frame number, pair count, device addresses) of its newest results

        def context_loop(k):
            e = engines[k]
            mine = list(range(k, count, S))
            if pipelined:
                prev = None
                for n_, s_ in enumerate(mine):
                    fr = frames[(first + s_) % nres]
                    if overlap:
                        jobs[k].put(fr)
                    d = descs[k][n_ & 1][0]
                    d.xyz1 = fr.data_ptr()
                    t = e.search_resident_begin(d)
                    if prev is not None:
                        r_ = e.search_resident_end(prev[1])
                        res[prev[0]] = (r_[0], None)
                        newest[k].append((first + prev[0],) + tuple(r_))
                    prev = (s_, t)
                    if not overlap:      # the fit of the frame runs behind its search on the same stream
                        out = e.fit_rmsd_batch(fr.unsqueeze(0), mass, ref, idx=idx, apply=True)
                        rms[s_] = float(out["rmsd"][0])
                if prev is not None:
                    r_ = e.search_resident_end(prev[1])
                    res[prev[0]] = (r_[0], None)
                    newest[k].append((first + prev[0],) + tuple(r_))
                del newest[k][:-2]       # a context's two result sets hold its last two frames
            else:
                for s_ in mine:
                    res[s_] = step(e, first + s_)
                    if not overlap:
                        rms[s_] = res[s_][1]
                    elif len(res[s_]) == 4:
                        newest[k] = [(first + s_, res[s_][0], res[s_][2], res[s_][3])]     # the context's one result set
            if overlap:
                for s_, v in zip(mine, collect_fits(k, len(mine))):
                    rms[s_] = v

        if S == 1:
            context_loop(0)
        else:
            errors = []

            def worker(k):
                try:
                    context_loop(k)
                except Exception as exc:       # a failure in a stream thread must end the run, not hang the join
                    errors.append(exc)

            th = [threading.Thread(target=worker, args=(k,)) for k in range(S)]
            for t_ in th:
                t_.start()
            for t_ in th:
                t_.join()
            if errors:
                raise errors[0]
        last_results[:] = sorted(x for k in range(S) for x in newest[k])[-2:]
        return [int(r[0]) for r in res], [float(v) for v in rms]

    # Untimed pre-heat: the same steps for --preheat seconds, so that the W warm-up steps and the K timed steps run at
    # the clocks the chip settles at under this load (a fresh box ramps for the first ~1 s: a 20-step run used to read
    # 7 % below a 200-step run of the same command).  Reported as `preheat_ms`; W and K mean what they always meant.
    t_pre = time.perf_counter()
    pre_steps = 0
    while args.preheat > 0 and time.perf_counter() - t_pre < args.preheat:
        run_steps(pre_steps, 16)
        pre_steps += 16
    barrier()
    preheat_ms = (time.perf_counter() - t_pre) * 1e3
    run_steps(0, W)
    barrier()
    # ---- the timed region: exactly K steps, no per-kernel event recording inside it
    t0 = time.perf_counter()
    counts, rms = run_steps(W, K)
    barrier()
    elapsed = time.perf_counter() - t0
    pairs, rsum = sum(counts), sum(rms)
    # ---- self-check of the timed code path (untimed, every run): the pair lists of the LAST TWO timed frames are still in
    # the context's two result sets.  A fresh context recomputes those frames alone (one search at a time, fit on the same
    # stream): pair counts, order-sensitive 64-bit checksums of both result planes and the RMSD must agree.  --verify
    # additionally recomputes ALL frames' counts and RMSDs.
    t_chk = time.perf_counter()
    self_check = None
    if overlap and K >= 2 * S and len(last_results) == 2 and (pipelined or S >= 2):
        self_check = True
        e3 = api.Engine(local_rank)
        for (fno, cnt, pp, dp) in last_results:
            mine = checksum_planes(cnt, pp, dp)
            fr = frames[fno % nres].clone()
            torch.cuda.synchronize()           # the copy runs on torch's stream, the engine on its own
            c2, p2, d2 = e3.search_resident(api.SEARCH_SINGLE, CUTOFF, fr, box=box, pbc=7)
            theirs = checksum_planes(c2, p2, d2)
            out = e3.fit_rmsd_batch(fr.unsqueeze(0), mass, ref, idx=idx, apply=True)
            r_timed, r_alone = rms[fno - W], float(out["rmsd"][0])
            ok = cnt == c2 and mine == theirs and abs(r_timed - r_alone) <= 1e-6 * abs(r_alone) + 1e-9
            if not ok:
                self_check = False
                print(f"bench.py self-check: rank {rank} frame {fno}: pairs {cnt} vs {c2}, checksums {mine} vs {theirs}, "
                      f"rmsd {r_timed!r} vs {r_alone!r}", file=sys.stderr)
        del e3
    self_check_s = time.perf_counter() - t_chk
    # ---- a second, untimed pass of the same steps with HIP events around every kernel group (per-kernel times and
    # the roofline of the fill kernel).  The events sit on each engine's own stream; with more than one context per GPU
    # the other contexts' kernels overlap the bracketed ones, so this pass runs the contexts one frame at a time.
    prof_engines = list(engines) + ([m for m in m_engines if m not in engines])
    for e in prof_engines:
        e.profile_enable(True)
        e.profile_read()
    KP = min(K, args.profile_steps)
    S_timed, S = S, 1
    run_steps(W, KP)
    barrier()
    prof = None
    for e in prof_engines:
        p1 = e.profile_read()
        e.profile_enable(False)
        prof = p1 if prof is None else {k: (prof[k][0] + p1[k][0], prof[k][1] + p1[k][1]) for k in prof}
    # ---- a third untimed pass: ONE event pair per frame around count + offsets + fill (molar_hip_profile_enable(ctx, 2)).  The
    # pass above brackets every kernel group with its own pair, and an event record is a barrier packet of 5-10 us on the
    # stream: its three entries read longer than the kernels run and used to add up to MORE than ms_per_step.
    for e in engines:
        e.profile_enable(2)
        e.profile_read()
    barrier()
    t_span = time.perf_counter()
    run_steps(W, KP)
    barrier()
    frame_span_pass_ms = (time.perf_counter() - t_span) * 1e3 / max(KP, 1)      # this pass's own period per step (its events included)
    frame_span_ms, frame_span_n = 0.0, 0
    for e in engines:
        p2 = e.profile_read()["search_frame"]
        e.profile_enable(False)
        frame_span_ms += p2[0]
        frame_span_n += p2[1]
    S = S_timed

    # ---- a measured MODE, never `value`: the same steps with the resident searches filling the (i, j) plane only - the output
    # form of the reference's (usize, usize) consumers (distance_search.rs:14-20; SearchConnectivity, the membrane's patches):
    # 8 instead of 12 bytes per result and no square roots.  Checked: every frame's pair count against the timed run's, the
    # pair plane of the last frame by checksum against the full mode's.
    pairs_only = None
    if overlap and pipelined and K >= 2 * S and not args.no_pairs_only:
        for e in engines:
            e.search_resident_planes(False)
        K2 = min(K, 100)
        run_steps(0, min(W, 5))
        barrier()
        t2 = time.perf_counter()
        c2, _ = run_steps(W, K2)
        barrier()
        dt2 = time.perf_counter() - t2
        ok_counts = c2 == counts[:K2]
        ok_plane = None
        if len(last_results) >= 1:
            fno, cnt, pp, dp = last_results[-1]
            mine = checksum_planes(cnt, pp, None)[0]
            e4 = api.Engine(local_rank)
            fr = frames[fno % nres].clone()
            torch.cuda.synchronize()
            c4, p4, d4 = e4.search_resident(api.SEARCH_SINGLE, CUTOFF, fr, box=box, pbc=7)
            ok_plane = bool(cnt == c4 and mine == checksum_planes(c4, p4, None)[0] and dp is None)
            del e4
        for e in engines:
            e.search_resident_planes(True)
        pairs_only = (K2, dt2, bool(ok_counts), ok_plane)

    # end-of-run reductions (RCCL when world > 1): integer pair count, max-over-ranks wall time
    from molar_amd.distributed import max_over_ranks, reduce_counts
    from molar_amd.distributed import gather_float64
    total_pairs = float(reduce_counts([pairs], device=cdev)[0])
    t = max_over_ranks(elapsed, device=cdev)
    per_rank_s = [float(v[0]) for v in gather_float64([elapsed], device=cdev)]        # a straggling rank shows up here
    pairs_only_line = None
    if pairs_only is not None:
        t_po = max_over_ranks(pairs_only[1], device=cdev)
        ok_po = int(reduce_counts([0 if (pairs_only[2] and pairs_only[3] is not False) else 1], device=cdev)[0]) == 0
        pairs_only_line = {"value": pairs_only[0] * world / t_po, "unit": "frames/s", "steps": pairs_only[0], "ms_per_step": t_po / pairs_only[0] * 1e3,
                           "bytes_per_result": 8, "pair_counts_and_pair_plane_equal_full_mode": ok_po,
                           "note": "measured mode, not the headline: molar_hip_search_resident_planes(ctx, 0) - the resident searches fill the (i, j) plane "
                                   "only (the reference's (usize, usize) output form, distance_search.rs:14-20), same frames, same pipeline"}
    from molar_amd.distributed import collective_view
    comm = collective_view(local_rank) if world > 1 else None
    checks = reduce_counts([0 if self_check is None else 1, 1 if self_check is False else 0], device=cdev)
    self_check_all = None if int(checks[0]) == 0 else (int(checks[1]) == 0)
    verified = None
    if args.verify:
        # Every rank's per-frame results travel to rank 0 (one integer all_reduce of a [world, K] table of pair counts, and
        # one of the RMSDs as int64 micro-units), and rank 0 recomputes ALL ranks' frames alone, one context, no
        # pipelining: pair counts must be equal, RMSDs equal to 1e-6 relative.  A mismatch ends the run with rc 1.
        tab = np.zeros((world, K), np.int64); tab[rank] = counts
        rtab = np.zeros((world, K), np.int64); rtab[rank] = np.round(np.asarray(rms, np.float64) * 1e9).astype(np.int64)
        tab = reduce_counts(tab.ravel(), device=cdev).reshape(world, K)
        rtab = reduce_counts(rtab.ravel(), device=cdev).reshape(world, K)
        if rank == 0:
            e2 = api.Engine(local_rank)
            verified = True
            for r in range(world):
                fr_r, _ = (frames, None) if r == rank else make_frames(nres, r, box, device)
                for s_ in range(K):
                    fr = fr_r[(W + s_) % nres].clone()
                    torch.cuda.synchronize()       # the copy runs on torch's stream, the engine on its own
                    cnt, _, _ = e2.search_resident(api.SEARCH_SINGLE, CUTOFF, fr, box=box, pbc=7)
                    out = e2.fit_rmsd_batch(fr.unsqueeze(0), mass, ref, idx=idx, apply=True)
                    ok = int(cnt) == int(tab[r, s_]) and abs(float(out["rmsd"][0]) * 1e9 - float(rtab[r, s_])) <= 1e-6 * abs(float(rtab[r, s_])) + 2.0
                    if not ok:
                        verified = False
                        print(f"bench.py --verify: rank {r} frame {s_}: pairs {int(tab[r, s_])} vs {int(cnt)}, rmsd*1e9 {int(rtab[r, s_])} vs "
                              f"{float(out['rmsd'][0]) * 1e9:.0f}", file=sys.stderr)
                del fr_r

    if rank == 0:
        frames_total = K * world
        fill_ms, fill_n = prof["pair_fill"]
        p_per_frame = total_pairs / frames_total
        alg_bytes = 12.0 * NATOMS + 12.0 * p_per_frame            # 12*N + 12*P per frame
        fill_avg_ms = fill_ms / max(fill_n, 1)
        achieved = alg_bytes / (fill_avg_ms * 1e-3) / 1e9 if fill_avg_ms > 0 else 0.0
        traffic = None
        tp = os.path.join(ROOT, "profiles", "traffic.json")        # HBM bytes per fill launch from rocprofv3 PMC passes
        if os.path.exists(tp):
            try:
                traffic = json.load(open(tp)).get("pair_fill_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        copy_peak = eng.copy_bandwidth(1 << 30, 10)                # best float4 copy kernel, same run
        write_peak = eng.write_bandwidth(1 << 30, 10)              # write-only float4 stream, same run
        line = {
            "metric": "frames/sec + Matom-pairs/sec, 1M-atom PBC neighbor search + RMSD fit",
            "value": frames_total / t,
            "unit": "frames/s",
            "matom_pairs_per_sec": total_pairs / t / 1e6,
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": t / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "1M-atom synthetic triclinic box A, per frame: PBC neighbour search rc=1.2 nm "
                            "(ordered pair list in HBM) + Kabsch RMSD fit/COM/gyration of a 100k-atom selection",
                "natoms": NATOMS, "cutoff_nm": CUTOFF, "pairs_per_frame": p_per_frame,
                "selection_atoms": int(len(idx_np)), "frames_per_gpu": K,
                "parallelism": f"frames sharded over {world} rank(s), no data-path collective"
                               + (" - ranks SHARE the GPUs (functional check, not a scaling result)" if args.share_gpu else ""),
                "streams_per_gpu": S * (2 if overlap else 1),
                "frames_in_flight_per_stream": 2 if pipelined else 1,
                "measure_overlapped_with_search": overlap,
            },
            "preheat_ms": preheat_ms,
            "per_rank_fps": [K / v for v in per_rank_s],
            # N > 1: the collective library's own view of the job (RCCL version, world size, every rank's device)
            "collective": comm,
            "pairs_only": pairs_only_line,
            "verified_against_single_context": self_check_all if verified is None else (verified and self_check_all is not False),
            "verification": ("last two timed frames of every rank recomputed on a fresh single context after the timed region: "
                             "pair counts, order-sensitive 64-bit checksums of the pair and distance planes in HBM, RMSD (1e-6 rel)"
                             + ("; --verify: every frame's pair count and RMSD as well" if verified is not None else "")
                             if self_check_all is not None else
                             ("--verify: every frame's pair count and RMSD" if verified is not None else None)),
            "self_check_s": self_check_s,
            "kernel_ms_per_frame": {k: v[0] / KP for k, v in prof.items() if k != "search_frame"},
            "kernel_ms_note": f"HIP-event times from a separate untimed pass of {KP} of the same steps on one context (events are "
                              "not recorded inside the timed region); grid_build (side stream) and measure (second context) "
                              "OVERLAP the search kernels, so the entries do not add up to ms_per_step",
            # the same numbers sorted by what they mean for the frame period: only the first group adds up (to ms_per_step less
            # ~0.05 ms of plan kernels and launch gaps); the second runs beside it and is stretched by the contention
            "critical_stream_ms_per_frame": {k: v[0] / KP for k, v in prof.items() if k in ("pair_count", "offset_scan", "pair_fill")},
            "overlapped_ms_per_frame": {k: v[0] / KP for k, v in prof.items() if k in ("grid_build", "measure")},
            # count + offsets + fill of a frame between ONE pair of events (third pass): what the critical stream spends per frame
            # besides the plan kernels (~0.02 ms) and the gaps between frames.  The pass is a separate one and carries its own event
            # pair per frame (a few us): compare it with ITS period (critical_path_pass_ms_per_step, <= 1 by construction), not to the
            # last digit with ms_per_step of the timed region
            "critical_path_ms_per_frame": frame_span_ms / max(frame_span_n, 1),
            "critical_path_pass_ms_per_step": frame_span_pass_ms,
            "critical_path_frac_of_its_pass": (frame_span_ms / max(frame_span_n, 1)) / frame_span_pass_ms if frame_span_pass_ms else None,
            "critical_path_source": f"one HIP-event pair per frame around count + offsets + fill, separate untimed pass of {KP} steps "
                                    "(the per-class entries above carry one event pair EACH and read 5-10 us longer per class than the kernels run)",
            "critical_path_sum_of_bracketed_classes_ms": sum(v[0] for k, v in prof.items() if k in ("pair_count", "offset_scan", "pair_fill")) / KP,
            "roofline": {
                "kernel": "pair_kernel<SINGLE,FILL>", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_source": "profiles/traffic.json (rocprofv3 PMC FETCH_SIZE/WRITE_SIZE passes, separate run of this command)",
                "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": fill_avg_ms,
                "end_to_end_frac": alg_bytes * frames_total / t / 1e9 / HBM_PEAK_GBS / world,
                "copy_peak_measured": copy_peak, "frac_of_copy_peak": achieved / copy_peak if copy_peak > 0 else None,
                "write_peak_measured": write_peak, "frac_of_write_peak": achieved / write_peak if write_peak > 0 else None,
            },
        }
        if not args.no_cpu_baseline and world == 1:
            f0 = frames[0].cpu().numpy()        # frame 0 was fitted in place during warm-up; any frame serves
            line["cpu_baseline"] = cpu_baseline(f0, ref.cpu().numpy(), box, mass.cpu().numpy(),
                                                idx_np.astype(np.uint64))
            line["speedup_vs_cpu_baseline"] = line["value"] / line["cpu_baseline"]["value"]
    # ---- the two other BASELINE configs on the same clock (never `value`): short legs of --workload rdf (C4) and --workload
    # membrane (C5) with their own self-checks, attached to the line as `secondary`.  N = 1 only unless --secondary is given
    # (a failure inside a leg's collectives must not cost a multi-GPU run its headline line).
    if (world == 1 and not args.no_secondary) or args.secondary:
        sec = secondary_legs(args, rank, local_rank, world, device, cdev)
        if rank == 0:
            line["secondary"] = sec
            if any(v.get("_failed") for v in sec.values() if isinstance(v, dict)):
                exit_code = 1
            for v in sec.values():
                if isinstance(v, dict):
                    v.pop("_failed", None)
    if rank == 0:
        print(json.dumps(line))
        if verified is False or self_check_all is False:
            exit_code = 1
    if overlap:
        for q in jobs:
            q.put(None)
    if world > 1:
        dist.destroy_process_group()
    if exit_code:
        raise SystemExit(exit_code)


if __name__ == "__main__":
    main()
