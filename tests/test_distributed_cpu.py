"""N>1 path on CPU: frame sharding + end-of-run reductions with the gloo backend, world_size 2.

The per-frame kernels need a GPU, so each rank here produces its frames' results with the CPU
oracle; what is under test is that sharded + reduced results equal the single-process results
bit for bit (integer histogram) / exactly (series order)."""
import os
import socket

import numpy as np
import pytest

from molar_amd import synth
from molar_amd.distributed import shard_frames


def test_shard_frames_partition():
    for n in (0, 1, 7, 8, 1000, 10001):
        for w in (1, 2, 3, 8):
            parts = [shard_frames(n, r, w) for r in range(w)]
            flat = [f for p in parts for f in p]
            assert flat == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _frame_results(f, n=1500, nbins=60, cutoff=0.6):
    from oracle.oracle import Oracle
    o = Oracle("f32")
    box = synth.box_a(n)
    pos = synth.frame(n, box, f)
    ref = synth.frame(n, box, 0)
    r = o.search_single_pbc(cutoff, pos, o.box_from_matrix(box), 7)
    hist = o.histogram_add(0.0, cutoff, nbins, r["d"]).astype(np.int64)
    mass = synth.masses(n)
    R, t = o.fit_transform(pos, mass, ref, mass)
    return hist, len(r["i"]), o.rmsd(o.apply_transform(pos, R, t), ref)


def _worker(rank, world, port, nframes, q):
    import torch.distributed as dist
    from molar_amd.distributed import gather_float64, gather_series, max_over_ranks, reduce_counts
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard_frames(nframes, rank, world)
    hist = np.zeros(60, np.int64)
    pairs = 0
    series = []
    for f in mine:
        h, c, r = _frame_results(f)
        hist += h
        pairs += c
        series.append(r)
    hist = reduce_counts(hist)
    pairs = int(reduce_counts([pairs])[0])
    full = gather_series(series, nframes)
    tmax = max_over_ranks(float(rank + 1))
    parts = gather_float64(np.array([rank + 0.25, 1.0 / (rank + 3)]))       # the membrane workload's accumulators travel like this
    from molar_amd.distributed import collective_view
    view = collective_view(None)          # what bench.py prints into the N > 1 line
    dist.barrier()
    if rank == 0:
        assert view["backend"] == "gloo" and view["world_size"] == world and view["nccl_version"] is None
        assert [r["rank"] for r in view["ranks"]] == list(range(world)) and len({r["pid"] for r in view["ranks"]}) == world
        q.put((hist, pairs, full, tmax, parts))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_matches_single_process():
    import torch.multiprocessing as mp
    nframes, world = 5, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nframes, q)) for r in range(world)]
    for p in procs:
        p.start()
    hist, pairs, series, tmax, parts = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    ref_hist = np.zeros(60, np.int64)
    ref_pairs = 0
    ref_series = []
    for f in range(nframes):
        h, c, r = _frame_results(f)
        ref_hist += h
        ref_pairs += c
        ref_series.append(r)
    assert np.array_equal(hist, ref_hist)          # integer bins: bit-identical to the 1-rank result
    assert pairs == ref_pairs
    assert np.array_equal(series, np.array(ref_series))
    assert tmax == 2.0
    assert len(parts) == world and all(np.array_equal(parts[r], np.array([r + 0.25, 1.0 / (r + 3)])) for r in range(world))
