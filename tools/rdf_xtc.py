#!/usr/bin/env python
"""BASELINE.json configs[3] end to end: an XTC trajectory of 250k-atom frames sharded over the ranks of one node,
each frame decoded by host threads into HBM and fed to the fused radial-distance histogram (bins resident on the GPU,
no per-frame round trip), bins summed over ranks with ONE integer all_reduce (RCCL) at the end.

    python tools/rdf_xtc.py [--frames F] [--natoms N]                      # one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/rdf_xtc.py --frames 512

The trajectory is synthetic: rank 0 writes it once with the test encoder of oracle/ (this tool is measurement
infrastructure, like bench.py's cpu_baseline leg), every rank maps the same file and reads only its block of frames."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--natoms", type=int, default=250_000)
    ap.add_argument("--window", type=int, default=16, help="frames decoded per call")
    ap.add_argument("--threads", type=int, default=0, help="decoder threads per rank (0 = cores / ranks)")
    ap.add_argument("--path", default="/tmp/molar_amd_rdf.xtc")
    ap.add_argument("--decoder", choices=("host", "device"), default="host",
                    help="host: decoder threads + pinned staging (molar_hip_xtc_read); device: one lane per frame on the GPU "
                         "(molar_hip_xtc_read_device; use --window 1024 or more)")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0")); local_rank = int(os.environ.get("LOCAL_RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from molar_amd import api, build, synth
    from molar_amd.distributed import max_over_ranks, reduce_counts, shard_frames
    from molar_amd.xtc import XtcReader
    build.build_library()
    n, F = args.natoms, args.frames
    box = synth.box_a(n)
    if rank == 0:
        from oracle.oracle import Oracle
        o = Oracle("f32")
        base = [synth.frame(n, box, f) for f in range(4)]            # four distinct frames, cycled
        blobs = [o.xtc_encode(b, np.ascontiguousarray(box.T).reshape(9), step=k, time=float(k)) for k, b in enumerate(base)]
        with open(args.path + ".tmp", "wb") as f:
            for k in range(F):
                f.write(blobs[k % 4])
        os.replace(args.path + ".tmp", args.path)
    if world > 1:
        dist.barrier()
    eng = api.Engine(local_rank)
    threads = args.threads or max(1, (os.cpu_count() or 8) // world)
    rd = XtcReader(args.path, engine=eng, nthreads=threads)
    mine = shard_frames(len(rd), rank, world)
    W = args.window
    # two windows of frames in HBM: a second engine context (own stream + pinned staging) decodes window w+1 on host
    # threads while the GPU histograms window w
    bufs = [torch.empty((W, n, 3), dtype=torch.float32, device=dev) for _ in range(2)]
    dec = api.Engine(local_rank)
    rd_dec = XtcReader(args.path, engine=dec, nthreads=threads)
    bins = torch.zeros(1200, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    # warm-up on the first frame of the block (buffers, library load)
    rd_dec.read_frames(mine.start, 1, out=bufs[0][:1])
    eng.search_histogram(api.SEARCH_SINGLE, 1.2, 0.0, 1.2, 1200, bufs[0][0], box=box, pbc=7, bins=bins, want_count=False)
    eng.synchronize(); bins.zero_(); torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(1)
    windows = [(f, min(W, mine.stop - f)) for f in range(mine.start, mine.stop, W)]

    t_decode = [0.0]

    def decode(w):
        f, k = windows[w]
        td = time.perf_counter()
        if args.decoder == "device":
            rd_dec.read_frames_device(f, k, bufs[w % 2][:k])
        else:
            rd_dec.read_frames(f, k, out=bufs[w % 2][:k])             # returns when the frames are in HBM
        t_decode[0] += time.perf_counter() - td
        return k

    t_consume = 0.0
    t0 = time.perf_counter()
    fut = pool.submit(decode, 0) if windows else None
    for w in range(len(windows)):
        k = fut.result()
        if w + 1 < len(windows):
            fut = pool.submit(decode, w + 1)                            # overlaps with the histogram launches below
        tc = time.perf_counter()
        for q in range(k):
            eng.search_histogram(api.SEARCH_SINGLE, 1.2, 0.0, 1.2, 1200, bufs[w % 2][q], box=box, pbc=7, bins=bins, want_count=False)
        eng.synchronize()                                              # this window's buffer is free again
        t_consume += time.perf_counter() - tc
    total_bins = reduce_counts(bins.cpu().numpy(), device=dev)       # the only collective: 1200 x int64
    torch.cuda.synchronize()
    elapsed = max_over_ranks(time.perf_counter() - t0, device=dev)
    if rank == 0:
        nf = len(rd)
        # every 4th frame is the same: the reduced histogram must be (frames/4) x the histogram of the four base frames
        ok = None
        if nf % 4 == 0:
            chk = np.zeros(1200, np.uint64)
            e2 = api.Engine(local_rank)
            for b in range(4):
                fr = rd.read_frames(b, 1)[0]
                chk, _ = e2.search_histogram(api.SEARCH_SINGLE, 1.2, 0.0, 1.2, 1200, fr, box=box, pbc=7, bins=chk)
            ok = bool(np.array_equal(total_bins.astype(np.uint64), chk * np.uint64(nf // 4)))
        print(json.dumps({"workload": f"C4 end to end: XTC decode ({threads} threads/rank) -> HBM -> fused RDF histogram, {nf} frames x {n} atoms",
                          "n_gpus": world, "frames_per_s": nf / elapsed, "pairs_in_histogram": int(total_bins.sum()),
                          "decoder": args.decoder, "window_frames": W,
                          # rank 0's two sides, each over the time it was busy (they overlap: the slower one sets frames_per_s)
                          "decode_frames_per_s": len(mine) / max(t_decode[0], 1e-9),
                          "decode_frames_per_s_per_thread": (len(mine) / max(t_decode[0], 1e-9) / threads) if args.decoder == "host" else None,
                          "consumer_frames_per_s": len(mine) / max(t_consume, 1e-9),
                          "reduced_bins_exact": ok, "collective": "one all_reduce of 1200 x int64"}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
