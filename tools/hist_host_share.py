"""How much of a C4-shaped frame is host time: the enqueue loop alone against the loop plus the wait for the GPU
(49 us of 508 us per frame on the round-3 box: the path is not host-bound).  python tools/hist_host_share.py"""
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
eng = api.Engine(0)
bins = torch.zeros(nbins, dtype=torch.int64, device='cuda')
for s in range(20): eng.search_histogram(api.SEARCH_SINGLE, 1.2, 0.0, 1.2, nbins, frames[s % 16], box=box, pbc=7, bins=bins, want_count=False)
eng.synchronize(); torch.cuda.synchronize()
K = 300
t0 = time.perf_counter()
for s in range(K): eng.search_histogram(api.SEARCH_SINGLE, 1.2, 0.0, 1.2, nbins, frames[s % 16], box=box, pbc=7, bins=bins, want_count=False)
t1 = time.perf_counter()
eng.synchronize(); torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue us/frame", (t1 - t0) / K * 1e6, "total us/frame", (t2 - t0) / K * 1e6)
