#!/usr/bin/env python
"""Large-frame check of the pipelined resident search: 300k-1M-atom frames of different sizes and cutoffs go through
molar_hip_search_resident_begin/_end in random order, two in flight (the grid of one frame is built on the side
stream while the pair kernels of the frame before it run); every result must equal count + fill.
Usage: python tools/big_pipeline_check.py"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from molar_amd import api, synth
ref_eng, eng = api.Engine(0), api.Engine(0)
class Dev:
    def __init__(self, ptr, n, t): self.__cuda_array_interface__ = {"shape": (n,), "typestr": t, "data": (ptr, False), "version": 2}
rng = np.random.default_rng(7)
frames = []
for k in range(12):
    n = int(rng.choice([300000, 600000, 1000000]))
    box = synth.box_a(n)
    pos = torch.from_numpy(synth.frame(n, box, k)).cuda()
    rc = float(rng.choice([0.5, 0.6, 0.7]))
    wn = ref_eng.search_count(api.SEARCH_SINGLE, rc, pos, box=box, pbc=7)
    wp, wd = ref_eng.search_fill(wn)
    frames.append((pos, box, rc, wn, wp, wd))
fails = 0
def check(k, res):
    global fails
    cnt, pp, dp = res
    pos, box, rc, wn, wp, wd = frames[k]
    p = torch.as_tensor(Dev(pp, cnt * 2, "<i4"), device="cuda").cpu().numpy().view(np.uint32).reshape(-1, 2)
    d = torch.as_tensor(Dev(dp, cnt, "<f4"), device="cuda").cpu().numpy()
    ok = cnt == wn and np.array_equal(p, wp) and np.array_equal(d, wd)
    if not ok:
        fails += 1; print("MISMATCH", k, cnt, wn)
descs = [eng.make_search_desc(api.SEARCH_SINGLE, f[2], f[0], box=f[1], pbc=7) for f in frames]
for rounds in range(3):
    prev = None
    order = rng.permutation(len(frames))
    for k in order:
        t = eng.search_resident_begin(descs[k][0])
        if prev is not None:
            check(prev[0], eng.search_resident_end(prev[1]))
        prev = (k, t)
    check(prev[0], eng.search_resident_end(prev[1]))
print("big pipeline:", 3 * len(frames), "frames,", fails, "failures")
