"""Times the resident search of the headline frame (1M atoms, box A, rc 1.2) on one context, per kernel class, for
whatever MOLAR_HIP_* knobs the environment holds (A/B runs of the one-pass kernel: MOLAR_HIP_ONEPASS, MOLAR_HIP_OP_RUN,
MOLAR_HIP_OP_DBG).  usage: python tools/op_time.py [natoms] [frames] [cutoff]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from molar_amd import build, synth
from molar_amd.api import Engine, SEARCH_SINGLE


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    rc = float(sys.argv[3]) if len(sys.argv) > 3 else 1.2
    build.build_library()
    eng = Engine(0)
    box = synth.box_a(n)
    pos = [torch.from_numpy(synth.frame(n, box, frame_no=k)).cuda() for k in range(2)]
    descs = [eng.make_search_desc(SEARCH_SINGLE, rc, p, box=box, pbc=7) for p in pos]
    for k in range(6):
        cnt, _, _ = eng.search_resident_desc(descs[k & 1][0])
    eng.profile_enable(True)
    eng.profile_read()
    t0 = time.perf_counter()
    for k in range(frames):
        cnt, _, _ = eng.search_resident_desc(descs[k & 1][0])
    wall = (time.perf_counter() - t0) / frames * 1e3
    prof = eng.profile_read()
    eng.profile_enable(False)
    out = {"natoms": n, "rc": rc, "pairs": cnt, "wall_ms_per_frame": round(wall, 4),
           "env": {k: v for k, v in os.environ.items() if k.startswith("MOLAR_HIP_")}}
    for name, (ms, launches) in prof.items():
        if launches:
            out[name] = round(ms / frames, 4)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
