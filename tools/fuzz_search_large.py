import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tools')
import numpy as np
from molar_amd import api, build, synth
from oracle.oracle import Oracle
build.build_library()
eng = api.Engine(0); o = Oracle("f32")
rng = np.random.default_rng(77); fails = 0
for case in range(40):
    n = int(rng.choice([20000, 50000, 120000]))
    boxfn = [synth.box_a, synth.box_b, synth.box_ortho][case % 3]
    dens = float(rng.choice([100.0, 400.0, 1500.0]))          # up to very crowded cells (> 512 atoms: streaming path)
    box = boxfn(n, density=dens)
    pos = synth.frame(n, box, case, sigma=float(rng.choice([0.02, 0.1])))
    rc = float(np.float32(rng.uniform(0.4, 0.9)))
    pbc = int(rng.choice([7, 7, 3, 5]))
    ob = o.box_from_matrix(box)
    ref = o.search_single_pbc(rc, pos, ob, pbc, nthreads=8)
    if len(ref["i"]) > 4e7: continue
    cnt, _, _ = eng.search_resident(api.SEARCH_SINGLE, rc, pos, box=box, pbc=pbc)
    pr, d = eng.search_fill(cnt)
    ok = cnt == len(ref["i"]) and np.array_equal(pr[:,0], ref["i"]) and np.array_equal(pr[:,1], ref["j"]) and np.array_equal(d, ref["d"])
    print(case, n, dens, round(rc,3), pbc, ref["dims"], cnt, "OK" if ok else "MISMATCH", flush=True)
    fails += not ok
print("failures", fails)
