"""Frame sharding and end-of-run reductions for multi-GPU runs (one process per GPU).

Frames of a trajectory are independent (analysis_task.rs:202-267 keeps no cross-frame state
apart from the task's own accumulators), so ranks own contiguous blocks of frames and never
exchange data on the hot path.  The only collectives run once, at the end: an integer
all_reduce of histogram bins / pair counts and a gather of the per-frame scalar series.
`torch.distributed` is the transport: backend "nccl" is RCCL over xGMI on the GPU box, "gloo"
in the CPU tests.  Payloads are tens of bytes to ~10 KB: latency-bound, bucket sizes irrelevant.
"""
from __future__ import annotations

import numpy as np


def shard_frames(nframes: int, rank: int, world: int) -> range:
    """Contiguous block of frame indices owned by `rank` (blocks differ by at most one frame)."""
    base, rem = divmod(nframes, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def _dist():
    import torch.distributed as dist
    return dist


def reduce_counts(local, device=None):
    """Element-wise SUM of an integer array over ranks (histogram bins, pair counts).  Integer
    arithmetic, so the result is bit-identical to a single-rank run over all frames."""
    import torch
    dist = _dist()
    t = torch.as_tensor(np.asarray(local, dtype=np.int64))
    if device is not None:
        t = t.to(device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def gather_series(local, nframes: int, device=None):
    """Concatenate per-frame float series from all ranks in frame order (ranks own the blocks of
    shard_frames).  Returns the full series on every rank."""
    import torch
    dist = _dist()
    local = np.asarray(local, dtype=np.float64)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    width = int(np.prod(local.shape[1:])) if local.ndim > 1 else 1
    maxlen = (nframes + world - 1) // world
    buf = torch.zeros((maxlen, width), dtype=torch.float64)
    buf[: len(local)] = torch.as_tensor(local.reshape(len(local), width))
    if device is not None:
        buf = buf.to(device)
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    out = []
    for r, p in enumerate(parts):
        k = len(shard_frames(nframes, r, world))
        out.append(p[:k].cpu().numpy())
    full = np.concatenate(out, 0)
    return full.reshape((nframes,) + local.shape[1:]) if local.ndim > 1 else full.reshape(nframes)


def gather_float64(local, device=None):
    """Every rank's float64 vector (same length on all ranks), as a list in rank order, on every rank.  Floating-point
    accumulators are combined by the caller in rank order, so the result does not depend on the reduction tree."""
    import torch
    dist = _dist()
    local = np.ascontiguousarray(local, dtype=np.float64).reshape(-1)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [local]
    t = torch.as_tensor(local)
    if device is not None:
        t = t.to(device)
    parts = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, t)
    return [p.cpu().numpy() for p in parts]


def max_over_ranks(value: float, device=None) -> float:
    import torch
    dist = _dist()
    t = torch.tensor([value], dtype=torch.float64)
    if device is not None:
        t = t.to(device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])


def collective_view(local_device_index=None):
    """What the collective library itself reports about the job, gathered over ranks: backend, its version (RCCL's on the
    GPU box), the world size torch.distributed sees and every rank's device (index, name, PCI bus id).  bench.py prints it
    into the N > 1 line, so that a scaling run proves N ranks on N distinct devices were seen.  Collective: call on every rank."""
    import os
    import torch
    dist = _dist()
    view = {"backend": None, "world_size": 1, "nccl_version": None, "ranks": []}
    mine = {"rank": 0, "pid": os.getpid(), "device_index": local_device_index, "device_name": None, "pci_bus_id": None}
    if local_device_index is not None and torch.cuda.is_available():
        try:
            pr = torch.cuda.get_device_properties(local_device_index)
            mine["device_name"] = pr.name
            mine["pci_bus_id"] = f"{getattr(pr, 'pci_domain_id', 0):04x}:{getattr(pr, 'pci_bus_id', 0):02x}:{getattr(pr, 'pci_device_id', 0):02x}"
        except Exception:
            pass
    if dist.is_available() and dist.is_initialized():
        view["backend"] = dist.get_backend()
        view["world_size"] = dist.get_world_size()
        mine["rank"] = dist.get_rank()
        if view["backend"] == "nccl":
            try:
                view["nccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:
                pass
        parts = [None] * view["world_size"]
        dist.all_gather_object(parts, mine)
        view["ranks"] = parts
    else:
        view["ranks"] = [mine]
    view["distinct_devices"] = len({(r["pci_bus_id"], r["device_index"]) for r in view["ranks"]})
    return view
