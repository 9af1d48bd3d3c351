import sys, time, cProfile, pstats
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from molar_amd import api, build
from molar_amd import membrane as mb
build.build_library()
eng = api.Engine(0)
xyz, box, first, tpl, masses = mb.build_bilayer(2000, 500_000)
m = mb.Membrane(eng, len(xyz), first, tpl, masses, mb.MembraneOptions(cutoff=2.5, order_type=1))
d = torch.from_numpy(xyz).cuda()
m.compute(d.clone(), box)
fr = [d.clone() for _ in range(20)]
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for f in fr: m.compute(f, box)
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
