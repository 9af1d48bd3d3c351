#!/usr/bin/env python
"""64-frames-per-call fit (C3 shape, batched) in a loop, for
`rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python tools/prof_fit_batch.py [stride|random]`: what the gather of
M = 1e5 selected atoms out of N = 1e6 really pulls from HBM per frame."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from molar_amd import api, synth
eng = api.Engine(0)
n, F = 1_000_000, 64
box = synth.box_a(n)
ref = torch.from_numpy(synth.frame(n, box, 0)).cuda()
frames = torch.stack([torch.from_numpy(synth.frame(n, box, 1 + (f % 4))).cuda() for f in range(F)]).contiguous()
mass = torch.from_numpy(synth.masses(n)).cuda()
if len(sys.argv) > 1 and sys.argv[1] == "random":
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    idx = torch.randperm(n, generator=g, device="cuda")[: n // 10].sort().values.to(torch.int64)
elif len(sys.argv) > 1 and sys.argv[1] == "block":
    idx = torch.arange(n // 3, n // 3 + n // 10, device="cuda", dtype=torch.int64)      # one contiguous molecule
else:
    idx = torch.arange(0, n, 10, device="cuda", dtype=torch.int64)
for _ in range(6):
    eng.fit_rmsd_batch(frames, mass, ref, idx=idx, apply=False)
eng.synchronize()
