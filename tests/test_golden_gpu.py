"""The HIP path (through the C ABI) against the committed golden fixtures of tests/golden/ — no oracle
involved at run time.  Pair lists and ids: bit-exact incl. order and distances; in-place coordinate ops:
bit-exact; reductions: 1e-5 relative (BASELINE.json north_star).  Provenance of the fixtures: make_golden.py."""
import hashlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def eng():
    from molar_amd import build
    from molar_amd.api import Engine
    build.build_library()
    return Engine(0)


@pytest.mark.parametrize("name", ["ortho", "tric_a", "hex_b", "rhombic_dodecahedron"])
def test_search_fixtures(eng, name):
    from molar_amd import api as a
    g = np.load(os.path.join(G, f"search_{name}.npz"))
    box, pos, rc, i1, i2, vdw = g["box"], g["pos"], float(g["cutoff"]), g["idx1"], g["idx2"], g["vdw"]

    def pairs(key, kind, cutoff, *args, **kw):
        n = eng.search_count(kind, cutoff, *args, **kw)
        p, d = eng.search_fill(n)
        assert n == len(g[key + "_i"]), key
        assert np.array_equal(p[:, 0], g[key + "_i"]) and np.array_equal(p[:, 1], g[key + "_j"]), key
        assert np.array_equal(d, g[key + "_d"]), key
        assert tuple(g[key + "_dims"]) == eng.grid_dims(), key

    pairs("single_pbc7", a.SEARCH_SINGLE, rc, pos, box=box, pbc=7)
    pairs("single_pbc3", a.SEARCH_SINGLE, rc, pos, box=box, pbc=3)
    pairs("single", a.SEARCH_SINGLE, rc, pos)
    pairs("double_pbc7", a.SEARCH_DOUBLE, rc, pos, i1, pos, i2, box=box, pbc=7)
    pairs("double", a.SEARCH_DOUBLE, rc, pos, i1, pos, i2)
    v1, v2 = vdw[i1.astype(int)], vdw[i2.astype(int)]           # radii per SELECTED atom (molar_hip.h)
    pairs("vdw_pbc7", a.SEARCH_DOUBLE_VDW, None, pos, i1, pos, i2, box=box, pbc=7, vdw1=v1, vdw2=v2)
    pairs("vdw", a.SEARCH_DOUBLE_VDW, None, pos, i1, pos, i2, vdw1=v1, vdw2=v2)

    def hist(key, kind, cutoff, *args, **kw):
        """The fused histogram against Histogram1D::add_one (stats.rs:29-35) applied, in numpy float32, to the fixture's
        distance stream: in-range bins, distances below the range and above it."""
        d = g[key + "_d"]
        nb, hmin, hmax = 97, np.float32(0.03), np.float32(0.47)
        fb = np.floor(np.float32(nb) * (d - hmin) / (hmax - hmin))
        ok = (fb >= 0) & (fb < nb)
        want = np.bincount(fb[ok].astype(np.int64), minlength=nb).astype(np.uint64)
        bins, cnt = eng.search_histogram(kind, cutoff, float(hmin), float(hmax), nb, *args, **kw)
        assert cnt == len(d) and np.array_equal(bins, want), key

    hist("single_pbc7", a.SEARCH_SINGLE, rc, pos, box=box, pbc=7)
    hist("single_pbc3", a.SEARCH_SINGLE, rc, pos, box=box, pbc=3)
    hist("single", a.SEARCH_SINGLE, rc, pos)
    hist("double_pbc7", a.SEARCH_DOUBLE, rc, pos, i1, pos, i2, box=box, pbc=7)
    hist("double", a.SEARCH_DOUBLE, rc, pos, i1, pos, i2)
    hist("vdw_pbc7", a.SEARCH_DOUBLE_VDW, None, pos, i1, pos, i2, box=box, pbc=7, vdw1=v1, vdw2=v2)
    n = eng.search_count(a.SEARCH_WITHIN, rc, pos, i1, pos, i2, box=box, pbc=7)
    assert np.array_equal(eng.search_fill_ids(n), g["within_pbc7_i"])
    n = eng.search_count(a.SEARCH_WITHIN, rc, pos, i1, pos, i2, lower=g["within_lower"], upper=g["within_upper"])
    assert np.array_equal(eng.search_fill_ids(n), g["within_i"])


def test_measure_fixture(eng):
    g = np.load(os.path.join(G, "measure.npz"))
    pos, ref, mass, idx, box = g["pos"], g["ref"], g["mass"], g["idx"], g["box"]
    from molar_amd import api as a
    pb = a.PeriodicBox.from_matrix(box)
    lo, hi = eng.min_max(pos, idx)
    assert np.array_equal(lo, g["min"].astype(np.float32)) and np.array_equal(hi, g["max"].astype(np.float32))
    tol = dict(rtol=1e-5, atol=1e-6)
    assert np.allclose(eng.center_of_geometry(pos, idx), g["cog"], **tol)
    assert np.allclose(eng.center_of_mass(pos, mass, idx), g["com"], **tol)
    assert np.allclose(eng.center_of_geometry_pbc(pos, pb, 7, idx), g["cog_pbc7"], **tol)
    assert np.allclose(eng.center_of_mass_pbc(pos, mass, pb, 7, idx), g["com_pbc7"], **tol)
    assert np.allclose(eng.center_of_mass_pbc(pos, mass, pb, 5, idx), g["com_pbc5"], **tol)
    assert np.isclose(eng.gyration(pos, mass, idx), g["gyration"], rtol=1e-5)
    assert np.isclose(eng.gyration(pos, mass, idx, pb), g["gyration_pbc"], rtol=1e-5)
    mom, axes, _ = eng.inertia(pos, mass, idx)
    assert np.allclose(mom, g["inertia_moments"], rtol=1e-5)
    for k in range(3):      # axes up to sign
        assert min(np.abs(axes[:, k] - g["inertia_axes"][:, k]).max(), np.abs(axes[:, k] + g["inertia_axes"][:, k]).max()) < 1e-4
    assert np.isclose(eng.rmsd(pos, ref, idx, idx), g["rmsd"], rtol=1e-5)
    assert np.isclose(eng.rmsd_mw(pos, mass, ref, idx, idx), g["rmsd_mw"], rtol=1e-5)
    R, t = eng.fit_transform(pos, mass, ref, mass, idx, idx)
    assert np.allclose(R, g["fit_R"], atol=1e-5) and np.allclose(t, g["fit_t"], atol=2e-4)
    moved = pos.copy()
    eng.apply_transform(moved, g["fit_R"].astype(np.float32), g["fit_t"].astype(np.float32), idx)
    assert np.array_equal(moved, g["applied_f32"])
    assert np.isclose(eng.rmsd(moved, ref, idx, idx), g["rmsd_after_fit"], rtol=1e-4)
    unw = pos.copy()
    eng.unwrap_simple(unw, pb, 7, idx)
    assert np.array_equal(unw, g["unwrapped_f32"])
    for ot, nm in ((0, "sz"), (1, "scd"), (2, "scd_corr")):
        got = eng.lipid_tail_order(g["tail"], [np.arange(len(g["tail"]), dtype=np.uint64)], ot, [g["tail_normal"]], [g["tail_bonds"]])[0]
        assert np.allclose(got, g["order_" + nm], rtol=1e-4, atol=2e-5), nm


def test_membrane_fixture(eng):
    from molar_amd import api as a
    g = np.load(os.path.join(G, "membrane.npz"))
    st = a.new_membrane_state(g["head"], g["normals"], g["valid"], len(g["patch_ids"]))
    eng.membrane_smooth(g["box"], st, g["patch_off"], g["patch_ids"])
    assert np.array_equal(st["valid"], g["out_valid"])
    ok = st["valid"].astype(bool)
    assert np.array_equal(st["nvert"][ok], g["out_nvert"][ok])
    K = len(ok)
    for k in np.flatnonzero(ok):
        s0 = int(g["patch_off"][k]) + 4 * k
        nv = int(st["nvert"][k])
        assert np.array_equal(st["neib_ids"][s0:s0 + nv], g["out_neib_ids"][s0:s0 + nv])
        assert np.allclose(st["voro_vertexes"][s0:s0 + nv], g["out_voro"][s0:s0 + nv], rtol=2e-5, atol=2e-5)
    for mine, theirs in (("quad_coefs", "coefs"), ("mean_curv", "mean_curv"), ("gauss_curv", "gauss_curv"), ("area", "area"),
                         ("princ_curvs", "princ_curvs"), ("princ_dirs", "princ_dirs"), ("normals", "normals"), ("head_markers", "head")):
        assert np.allclose(st[mine][ok], g["out_" + theirs][ok], rtol=2e-5, atol=2e-5), mine
    assert np.array_equal(st["head_markers"][~ok], g["head"][~ok]) or K == ok.sum()


def test_ordered_pair_digests(eng):
    from molar_amd import api as a, synth
    dig = json.load(open(os.path.join(G, "ordered_pair_digests.json")))
    for name, d in dig.items():
        if d["natoms"] > 100_000:
            continue                    # the 1M-atom entry is hashed in chunks from HBM in tests/test_gpu_full_size.py
        box = (synth.box_a if d["box"] == "tric_a" else synth.box_b)(d["natoms"])
        pos = synth.frame(d["natoms"], box, 0)
        n = eng.search_count(a.SEARCH_SINGLE, d["cutoff"], pos, box=box, pbc=7)
        p, dist = eng.search_fill(n)
        h = hashlib.sha256()
        h.update(np.ascontiguousarray(p[:, 0]).astype("<u4").tobytes()); h.update(np.ascontiguousarray(p[:, 1]).astype("<u4").tobytes())
        h.update(dist.astype("<f4").tobytes())
        assert n == d["npairs"] and list(eng.grid_dims()) == d["dims"] and h.hexdigest() == d["sha256_i_j_d"], name
