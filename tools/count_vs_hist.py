import sys; sys.path.insert(0, '/root/repo')
import numpy as np, torch
from molar_amd import api, build, synth
build.build_library()
eng = api.Engine(0)
for n in (250_000, 1_000_000):
    box = synth.box_a(n)
    pos = torch.from_numpy(synth.frame(n, box, 1)).cuda()
    d, keep = eng.make_search_desc(api.SEARCH_SINGLE, 1.2, pos, box=box, pbc=7)
    for _ in range(3): eng.search_resident_desc(d)
    eng.profile_enable(True); eng.profile_read()
    for _ in range(20): cnt, _, _ = eng.search_resident_desc(d)
    eng.synchronize()
    pr = eng.profile_read(); eng.profile_enable(False)
    print(n, cnt, {k: round(v[0] / max(v[1], 1), 4) for k, v in pr.items()}, eng.grid_dims())
