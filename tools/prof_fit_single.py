#!/usr/bin/env python
"""One-frame-per-call fit (C3 shape) in a loop, for `rocprofv3 --kernel-trace --stats -- python tools/prof_fit_single.py`."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from molar_amd import api, synth
eng = api.Engine(0)
n = 1_000_000
box = synth.box_a(n)
ref = torch.from_numpy(synth.frame(n, box, 0)).cuda()
cur = torch.from_numpy(synth.frame(n, box, 1)).cuda().unsqueeze(0).contiguous()
mass = torch.from_numpy(synth.masses(n)).cuda()
idx = torch.arange(0, n, 10, device="cuda", dtype=torch.int64)
for _ in range(50):
    eng.fit_rmsd_batch(cur, mass, ref, idx=idx, apply=True)
eng.synchronize()
