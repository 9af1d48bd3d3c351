"""Fused histogram with device-resident bins: 64 frames queued back to back against the same frames with a wait after
each, twelve times - the bins have to be identical (integer atomics into the caller's bins, two grid generations in
flight).  python tools/hist_race_check.py"""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from molar_amd import api, build, synth
build.build_library()
n, nbins = 250_000, 1200
box = synth.box_a(n)
g = torch.Generator(device='cuda'); g.manual_seed(1)
base = torch.rand((n, 3), generator=g, device='cuda', dtype=torch.float64) @ torch.from_numpy(box.astype(np.float64)).cuda().T
frames = [(base + torch.randn((n, 3), generator=g, device='cuda') * 0.05).float().contiguous() for _ in range(16)]
torch.cuda.synchronize()
eng = api.Engine(0)
def run(K, sync_each):
    bins = torch.zeros(nbins, dtype=torch.int64, device='cuda')
    torch.cuda.synchronize()
    for s in range(K):
        eng.search_histogram(api.SEARCH_SINGLE, 1.2, 0.0, 1.2, nbins, frames[s % 16], box=box, pbc=7, bins=bins, want_count=False)
        if sync_each: eng.synchronize()
    eng.synchronize(); torch.cuda.synchronize()
    return bins.cpu().numpy()
ref = run(64, True)
bad_q = bad_s = 0
for rep in range(12):
    a = run(64, False)
    b = run(64, True)
    if not np.array_equal(a, ref): bad_q += 1; print("queued differs: sum", a.sum() - ref.sum(), "nbins differing", (a != ref).sum())
    if not np.array_equal(b, ref): bad_s += 1; print("synced differs: sum", b.sum() - ref.sum(), "nbins differing", (b != ref).sum())
print("queued bad", bad_q, "synced bad", bad_s, "total", int(ref.sum()))
