import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from molar_amd import api, build, synth
build.build_library()
e = api.Engine(0)
n = 2000
box = synth.box_a(n); pos = torch.from_numpy(synth.frame(n, box)).cuda()
d, keep = e.make_search_desc(api.SEARCH_SINGLE, 0.5, pos, box=box, pbc=7)
for _ in range(20): e.search_resident_desc(d)
t0 = time.perf_counter()
for _ in range(500): e.search_resident_desc(d)
t1 = time.perf_counter()
print("tiny resident search: %.1f us per call" % ((t1 - t0) / 500 * 1e6))
e.profile_enable(True); e.profile_read()
for _ in range(100): e.search_resident_desc(d)
p = e.profile_read()
print({k: round(v[0] / 100 * 1e3, 1) for k, v in p.items()}, "us per call (GPU event spans)")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(500): e.search_resident_desc(d)
pr.disable(); pstats.Stats(pr).sort_stats('tottime').print_stats(4)
