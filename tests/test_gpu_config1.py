"""BASELINE.json configs[0]: a 25k-atom GRO box, single frame - `within 1.0 nm` search + rmsd fit against frame 0 -
read through the engine's GRO plumbing, computed on the GPU, checked against the oracle fed by ITS OWN reader."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_config1_gro_within_and_fit(tmp_path, orc32, orc64):
    from molar_amd import api, build, gro, synth
    from oracle import gro_oracle as og
    build.build_library()
    eng = api.Engine(0)
    n = 25000
    box = synth.box_a(n)
    names = [["OW", "HW1", "HW2", "CA", "N", "C", "O", "P"][k % 8] for k in range(n)]
    resn = [["SOL", "SOL", "SOL", "ALA", "ALA", "ALA", "ALA", "POPE"][k % 8] for k in range(n)]
    top0 = gro.GroTopology(names, resn, np.arange(n) // 8 + 1)
    paths = []
    for f in range(2):
        st = api.State(synth.frame(n, box, f), api.PeriodicBox.from_matrix(box), 10.0 * f)
        p = tmp_path / f"frame{f}.gro"
        gro.write_gro(p, top0, st)
        paths.append(p)
    (top, ref), (_, cur) = gro.read_gro(paths[0]), gro.read_gro(paths[1])
    o0, o1 = og.read_gro(paths[0]), og.read_gro(paths[1])
    assert np.array_equal(cur.coords, o1["xyz"]) and np.array_equal(top.masses, o1["mass"]) and cur.time == 10.0
    assert np.array_equal(cur.pbox.get_matrix(), o1["box"])
    # `within 1.0 pbc of resid 100..140` evaluated inside all atoms (selection/ast.rs:589-631)
    inner_idx = np.flatnonzero((top.resids >= 100) & (top.resids <= 140))
    allsel = api.Sel(top, cur, None, engine=eng)
    inner = api.Sel(top, cur, inner_idx, engine=eng)
    got = allsel.within(1.0, inner, pbc=[True, True, True])
    ob = orc32.box_from_matrix(o1["box"])
    r = orc32.search_within_pbc(1.0, o1["xyz"], o1["xyz"][inner_idx], ob, 7, np.arange(n), inner_idx, nthreads=4)
    assert np.array_equal(got, np.unique(r["i"])) and len(got) > len(inner_idx)
    got_np = allsel.within(1.0, inner)                                   # non-periodic variant
    lo, up = orc32.min_max(o1["xyz"])
    lo = lo + (np.float32(-1.0) - np.float32(1.1920929e-07)); up = up + (np.float32(1.0) + np.float32(1.1920929e-07))
    r = orc32.search_within(1.0, o1["xyz"], o1["xyz"][inner_idx], lo, up, np.arange(n), inner_idx, nthreads=4)
    assert np.array_equal(got_np, np.unique(r["i"]))
    # rmsd fit of the CA atoms against frame 0 (comparison_small.rs:17-24)
    ca = np.flatnonzero(np.array(top.names) == "CA")
    s_cur, s_ref = api.Sel(top, cur, ca, engine=eng), api.Sel(top, ref, ca, engine=eng)
    before = api.rmsd(s_cur, s_ref)
    R, t = api.fit_transform(s_cur, s_ref)
    s_cur.apply_transform((R, t))
    after = api.rmsd(s_cur, s_ref)
    Ro, to = orc64.fit_transform(o1["xyz"], o1["mass"], o0["xyz"], o0["mass"], ca.astype(np.uint64), ca.astype(np.uint64))
    moved = orc64.apply_transform(o1["xyz"], Ro, to, ca.astype(np.uint64))
    want_after = orc64.rmsd(moved, o0["xyz"], ca.astype(np.uint64), ca.astype(np.uint64))
    want_before = orc64.rmsd(o1["xyz"], o0["xyz"], ca.astype(np.uint64), ca.astype(np.uint64))
    assert np.isclose(before, want_before, rtol=1e-5) and np.isclose(after, want_after, rtol=1e-5) and after <= before
    assert np.allclose(R, Ro, atol=1e-5)
