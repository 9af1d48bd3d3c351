"""Committed golden fixtures (tests/golden/, provenance in make_golden.py) replayed on the CPU:
  * the reference's PeriodicBox known answers against BOTH oracle builds and the product's host arithmetic
    (libmolar_hip.so's box functions run on the host; no GPU needed);
  * the oracle-generated vectors against the oracle as built now (regression pin of the checker itself)."""
import hashlib
import json
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = json.load(open(os.path.join(G, "periodic_box_known_answers.json")))["cases"]


class OracleBox:
    def __init__(self, o):
        self.o = o

    def matrix(self, rows):
        return self.o.box_from_matrix(np.array(rows, float))

    def va(self, v, a):
        return self.o.box_from_vectors_angles(*v, *a)

    def shortest(self, b, v, dims):
        return self.o.shortest_vector_dims(b, v, tuple(dims))

    def closest(self, b, p, t, dims):
        return self.o.closest_image_dims(b, p, t, tuple(dims))

    def distance(self, b, p1, p2, dims):
        return self.o.distance(b, p1, p2, tuple(dims))

    def nshift(self, b):
        return b.nshift


class ProductBox:
    """The engine's own PeriodicBox (host code of libmolar_hip.so through molar_amd.api)."""

    def __init__(self):
        from molar_amd import api, build
        build.build_library()
        self.api = api

    def matrix(self, rows):
        return self.api.PeriodicBox.from_matrix(np.array(rows, np.float32))

    def va(self, v, a):
        return self.api.PeriodicBox(v, a)

    def shortest(self, b, v, dims):
        return b.shortest_vector(v, dims)

    def closest(self, b, p, t, dims):
        return b.closest_image(p, t, dims)

    def distance(self, b, p1, p2, dims):
        return b.distance(p1, p2, dims)

    def nshift(self, b):
        return b.n_tric_corrections


@pytest.fixture(params=["oracle_f32", "oracle_f64", "product"])
def impl(request, orc32, orc64):
    return {"oracle_f32": lambda: OracleBox(orc32), "oracle_f64": lambda: OracleBox(orc64), "product": ProductBox}[request.param]()


@pytest.mark.parametrize("case", CASES, ids=[c["ref"] for c in CASES])
def test_reference_known_answer(impl, case):
    op = case["op"]
    if op == "from_vectors_angles_fails":
        with pytest.raises(Exception):
            impl.va(case["vectors"], case["angles"])
        return
    b = impl.va(case["vectors"], case["angles"]) if op.endswith("_va") else impl.matrix(case["matrix"])
    if op in ("shortest_vector", "shortest_vector_va"):
        assert np.linalg.norm(np.asarray(impl.shortest(b, case["v"], case["dims"]), float) - case["expect"]) < case["tol"]
    elif op == "closest_image":
        assert np.linalg.norm(np.asarray(impl.closest(b, case["p"], case["target"], case["dims"]), float) - case["expect"]) < case["tol"]
    elif op == "n_tric_corrections":
        assert impl.nshift(b) == case["expect"]
    elif op == "distance":
        assert abs(float(impl.distance(b, case["p1"], case["p2"], case["dims"])) - case["expect"]) < case["tol"]
    elif op == "distance_below":
        assert float(impl.distance(b, case["p1"], case["p2"], case["dims"])) < case["below"]
    elif op == "shortest_norm_vs_brute_force":
        m = np.array(case["matrix"], float); v = np.array(case["v"], float); r = case["brute_range"]
        best = min(np.linalg.norm(v + i * m[:, 0] + j * m[:, 1] + k * m[:, 2])
                   for i in range(-r, r + 1) for j in range(-r, r + 1) for k in range(-r, r + 1))
        assert abs(np.linalg.norm(np.asarray(impl.shortest(b, v, (True, True, True)), float)) - best) < case["tol"]
    else:
        raise AssertionError(op)


SEARCH_BOXES = ["ortho", "tric_a", "hex_b", "rhombic_dodecahedron"]


def oracle_searches(o, g):
    """All nine golden searches of one fixture, recomputed."""
    box, pos, rc = g["box"], g["pos"], float(g["cutoff"])
    ob = o.box_from_matrix(box)
    i1, i2 = g["idx1"], g["idx2"]
    p1, p2 = pos[i1.astype(int)], pos[i2.astype(int)]
    v1, v2 = g["vdw"][i1.astype(int)], g["vdw"][i2.astype(int)]
    return {
        "single_pbc7": o.search_single_pbc(rc, pos, ob, 7), "single_pbc3": o.search_single_pbc(rc, pos, ob, 3),
        "single": o.search_single(rc, pos),
        "double_pbc7": o.search_double_pbc(rc, p1, p2, ob, 7, ids1=i1, ids2=i2), "double": o.search_double(rc, p1, p2, ids1=i1, ids2=i2),
        "vdw_pbc7": o.search_double_vdw_pbc(p1, p2, v1, v2, ob, 7), "vdw": o.search_double_vdw(p1, p2, v1, v2),
        "within_pbc7": o.search_within_pbc(rc, p1, p2, ob, 7, ids1=i1, ids2=i2),
        "within": o.search_within(rc, p1, p2, g["within_lower"], g["within_upper"], ids1=i1, ids2=i2),
    }


@pytest.mark.parametrize("name", SEARCH_BOXES)
def test_oracle_reproduces_search_fixtures(orc32, name):
    g = np.load(os.path.join(G, f"search_{name}.npz"))
    for key, r in oracle_searches(orc32, g).items():
        assert np.array_equal(r["i"].astype(np.uint32), g[key + "_i"]), key
        assert tuple(g[key + "_dims"]) == r["dims"]
        if "j" in r:
            assert np.array_equal(r["j"].astype(np.uint32), g[key + "_j"]) and np.array_equal(r["d"], g[key + "_d"]), key


@pytest.mark.parametrize("name", SEARCH_BOXES)
def test_search_fixtures_agree_with_brute_force_where_grid_is_complete(orc32, name):
    """Independent check of the fixtures themselves: the non-periodic single search equals an O(N^2) scan as a
    set of pairs (the reference's non-PBC grid is always geometrically complete)."""
    g = np.load(os.path.join(G, f"search_{name}.npz"))
    pos = g["pos"].astype(np.float32)
    d = pos[:, None, :] - pos[None, :, :]
    d2 = ((d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]).astype(np.float32) + d[..., 2] * d[..., 2]).astype(np.float32)
    rc = np.float32(g["cutoff"])
    iu = np.triu_indices(len(pos), 1)
    want = set(zip(*(x[d2[iu] <= rc * rc] for x in iu)))
    got = set((min(a, b), max(a, b)) for a, b in zip(g["single_i"].tolist(), g["single_j"].tolist()))
    assert got == want


def test_oracle_reproduces_measure_fixture(orc32, orc64):
    g = np.load(os.path.join(G, "measure.npz"))
    pos, ref, mass, idx, box = g["pos"], g["ref"], g["mass"], g["idx"], g["box"]
    o = orc64
    ob = o.box_from_matrix(box)
    lo, hi = o.min_max(pos, idx)
    assert np.array_equal(lo, g["min"]) and np.array_equal(hi, g["max"])
    assert np.allclose(o.center_of_mass(pos, mass, idx), g["com"], rtol=1e-12)
    assert np.allclose(o.center_of_mass_pbc_dims(pos, mass, ob, 7, idx), g["com_pbc7"], rtol=1e-12)
    assert np.isclose(o.gyration(pos, mass, idx), g["gyration"], rtol=1e-12)
    R, t = o.fit_transform(pos, mass, ref, mass, idx, idx)
    assert np.allclose(R, g["fit_R"], atol=1e-10) and np.allclose(t, g["fit_t"], atol=1e-9)
    assert np.array_equal(orc32.apply_transform(pos, R.astype(np.float32), t.astype(np.float32), idx), g["applied_f32"])
    assert np.array_equal(orc32.unwrap_simple_dim(pos, orc32.box_from_matrix(box), 7, idx), g["unwrapped_f32"])
    for ot, nm in ((0, "sz"), (1, "scd"), (2, "scd_corr")):
        assert np.allclose(o.lipid_tail_order(g["tail"], ot, g["tail_normal"], g["tail_bonds"]), g["order_" + nm], rtol=1e-10, atol=1e-12)
    # the fixture is self-consistent: fitting brings the selection closer to the reference than it was
    assert g["rmsd_after_fit"] <= g["rmsd"]


def test_oracle_reproduces_membrane_fixture(orc32):
    g = np.load(os.path.join(G, "membrane.npz"))
    r = orc32.membrane_smooth(orc32.box_from_matrix(g["box"]), g["head"], g["normals"], g["valid"], g["patch_off"], g["patch_ids"])
    for k, v in r.items():
        assert np.array_equal(v, g["out_" + k]), k


def test_ordered_pair_digests(orc32):
    from molar_amd import synth
    dig = json.load(open(os.path.join(G, "ordered_pair_digests.json")))
    for name, d in dig.items():
        if d["natoms"] > 100_000:
            continue                    # BASELINE config 2 at full size: checked on the GPU box (tests/test_gpu_full_size.py)
        boxfn = synth.box_a if d["box"] == "tric_a" else synth.box_b
        box = boxfn(d["natoms"])
        pos = synth.frame(d["natoms"], box, 0)
        r = orc32.search_single_pbc(d["cutoff"], pos, orc32.box_from_matrix(box), 7, nthreads=4)
        h = hashlib.sha256()
        h.update(r["i"].astype("<u4").tobytes()); h.update(r["j"].astype("<u4").tobytes()); h.update(r["d"].astype("<f4").tobytes())
        assert len(r["i"]) == d["npairs"] and list(r["dims"]) == d["dims"] and h.hexdigest() == d["sha256_i_j_d"], name
