"""Per-lipid membrane analysis (markers, patches, normals, tail order) on the GPU against the same
pipeline assembled from the CPU oracle's primitives (restating molar_membrane/src/lib.rs:435-558)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from molar_amd import build
    from molar_amd.api import Engine
    build.build_library()
    return Engine(0)


def oracle_normals(o, head, tail, patch):
    """lib.rs:456-505 in f32, sequential second pass."""
    f = np.float32
    K = len(head)
    thv = np.zeros((K, 3), f)
    for i in range(K):
        v = (head[i] - tail[i]).astype(f)
        thv[i] = v / f(np.sqrt(f(f(v[0] * v[0] + v[1] * v[1]) + v[2] * v[2])))

    def ang(a, b):
        n1 = f(np.sqrt(f(f(a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]))); n2 = f(np.sqrt(f(f(b[0] * b[0] + b[1] * b[1]) + b[2] * b[2])))
        if n1 == 0 or n2 == 0:
            return f(0)
        c = f(f(f(a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]) / f(n1 * n2))
        return f(np.arccos(np.clip(c, f(-1), f(1))))
    nv = np.zeros((K, 3), f)
    for p in range(2):
        src = thv if p == 0 else nv
        for i in range(K):
            s = np.zeros(3, f)
            for l in patch[i]:
                if ang(src[l], src[i]) <= f(np.pi / 2):
                    s = (s + src[l]).astype(f)
            s = (s + src[i]).astype(f)
            nv[i] = s / f(np.sqrt(f(f(s[0] * s[0] + s[1] * s[1]) + s[2] * s[2])))
    return nv


def check_smooth(got, want, patch_off, K, tol=2e-5):
    """GPU state vs the oracle's: validity, Voronoi neighbour ids and vertex counts exactly; floats to `tol`
    (both sides evaluate the same f32 expressions; libm/rounding of sqrt and division are IEEE on both)."""
    assert np.array_equal(got["valid"], want["valid"])
    ok = want["valid"].astype(bool)
    assert np.array_equal(got["nvert"][ok], want["nvert"][ok])
    for k in np.flatnonzero(ok):
        s0 = int(patch_off[k]) + 4 * k
        nv = int(want["nvert"][k])
        assert np.array_equal(got["neib_ids"][s0:s0 + nv], want["neib_ids"][s0:s0 + nv])
        assert np.allclose(got["voro_vertexes"][s0:s0 + nv], want["voro"][s0:s0 + nv], rtol=tol, atol=tol)
        a, b = int(patch_off[k]), int(patch_off[k + 1])
        assert np.allclose(got["fitted_patch_points"][a:b], want["fitted"][a:b], rtol=tol, atol=tol)
    for g, w in (("quad_coefs", "coefs"), ("mean_curv", "mean_curv"), ("gauss_curv", "gauss_curv"),
                 ("princ_curvs", "princ_curvs"), ("princ_dirs", "princ_dirs"), ("area", "area"), ("normals", "normals"),
                 ("smoothed_head", "head")):
        assert np.allclose(got[g][ok], want[w][ok], rtol=tol, atol=tol), g


def run_pipeline_check(eng, orc32, per_leaflet, natoms, fused=True):
    """Membrane.compute on a synthetic bilayer against the same pipeline assembled from the oracle's primitives
    (fused: the chained molar_hip_membrane_frame_* call; otherwise one call per stage)."""
    from molar_amd import membrane as mb
    xyz, box, first, tpl, masses = mb.build_bilayer(per_leaflet, natoms)
    K = len(first)
    m = mb.Membrane(eng, len(xyz), first, tpl, masses, mb.MembraneOptions(cutoff=1.5, order_type=1, fused=fused))
    assert m.fusable() == fused
    work = xyz.copy()
    res = m.compute(work, box)
    ob = orc32.box_from_matrix(box)
    # unwrap per lipid (exact f32 arithmetic); a lipid's unwrap touches only its own atoms, so the oracle works on slices
    ref_xyz = xyz.copy()
    na = tpl.natoms
    for k in range(K):
        f0 = int(first[k])
        ref_xyz[f0:f0 + na] = orc32.unwrap_simple_dim(ref_xyz[f0:f0 + na], ob, 7)
    assert np.array_equal(work, ref_xyz)
    # markers: centre of mass per sub-selection
    for name, sub in (("head", tpl.head), ("mid", tpl.mid), ("tail", tpl.tail_end)):
        want = np.array([orc32.center_of_mass(ref_xyz[int(first[k]):int(first[k]) + na], masses[int(first[k]):int(first[k]) + na],
                                              sub.astype(np.uint64)) for k in range(K)])
        assert np.allclose(res[name], want, rtol=2e-6, atol=2e-6)
    # patches from the GPU markers (the search itself is bit-exact given identical input)
    r = orc32.search_single_pbc(1.5, res["head"], ob, 7)
    patch = [[] for _ in range(K)]
    for i, j in zip(r["i"].tolist(), r["j"].tolist()):
        patch[i].append(j); patch[j].append(i)
    for k in range(K):
        got = res["patch_ids"][int(res["patch_off"][k]): int(res["patch_off"][k + 1])].tolist()
        assert got == patch[k]
    assert np.mean([len(p) for p in patch]) > 4
    # normals
    want_n = oracle_normals(orc32, res["head"], res["tail"], patch)
    assert np.allclose(res["initial_normals"], want_n, atol=2e-6)
    assert np.allclose(np.linalg.norm(res["initial_normals"], axis=1), 1.0, atol=1e-5)
    # upper leaflet normals point up, lower down (tails towards the mid-plane)
    assert (res["initial_normals"][:per_leaflet, 2] > 0.8).all() and (res["initial_normals"][per_leaflet:, 2] < -0.8).all()
    # one smoothing pass (lib.rs:661-812) against the oracle's restatement on the same inputs
    so = orc32.membrane_smooth(ob, res["head"], res["initial_normals"], np.ones(K, np.uint8), res["patch_off"], res["patch_ids"])
    check_smooth(res, so, res["patch_off"], K)
    # order parameters per tail with the lipid normal
    for t, carbons in enumerate(tpl.tails):
        for k in range(0, K, 7):
            if not res["valid"][k]:
                continue
            f0 = int(first[k])
            want = orc32.lipid_tail_order(ref_xyz[f0:f0 + na], 1, res["normals"][k][None, :], tpl.bond_orders[t],
                                          idx=carbons.astype(np.uint64))
            assert np.allclose(res["order"][t][k], want, atol=3e-5)
    # roughly ordered chains along the normal: mean |Scd| in a sensible range
    assert 0.05 < np.abs(np.concatenate([o.reshape(-1) for o in res["order"]])).mean() < 0.6
    return K, int(np.count_nonzero(res["valid"]))


@pytest.mark.parametrize("fused", [True, False])
def test_membrane_pipeline_matches_oracle(eng, orc32, fused):
    run_pipeline_check(eng, orc32, 200, 40000, fused)


@pytest.mark.timeout(1500)
def test_membrane_pipeline_at_baseline_size(eng, orc32):
    """BASELINE.json configs[4]: the 500k-atom bilayer (2 x 2000 lipids) through Membrane.compute.  Validity, Voronoi
    neighbour ids and vertex counts exact, floats within 2e-5 of the f32 oracle pipeline."""
    K, nvalid = run_pipeline_check(eng, orc32, 2000, 500_000)
    assert K == 4000 and nvalid > 3900


def _patches_from_oracle(o, ob, head, cutoff):
    r = o.search_single_pbc(cutoff, head, ob, 7)
    K = len(head)
    i = r["i"].astype(np.int64); j = r["j"].astype(np.int64)
    src = np.stack([i, j], 1).reshape(-1); dst = np.stack([j, i], 1).reshape(-1)
    order = np.argsort(src, kind="stable")
    return (np.concatenate([[0], np.cumsum(np.bincount(src, minlength=K))]).astype(np.uint64), dst[order].astype(np.uint64))


@pytest.mark.parametrize("tric", [False, True])
def test_smooth_sheet_parity_and_geometry(eng, orc32, orc64, tric):
    """A jittered undulating sheet crossing the periodic boundaries: the GPU pass against the f32 oracle (ids
    exact, floats 2e-5) and the f64 oracle (conditioning of the 6x6 normal equations: 2e-3), plus what the
    geometry must give: six-ish neighbours, cell areas tiling the box."""
    from molar_amd import api
    rng = np.random.default_rng(5)
    side = 40
    L = side * 0.8
    g = (np.stack(np.meshgrid(np.arange(side), np.arange(side), indexing="ij"), -1).reshape(-1, 2) + 0.5
         + 0.2 * rng.normal(size=(side * side, 2))) * L / side
    z = 5.0 + 0.3 * np.sin(2 * np.pi * g[:, 0] / L) * np.cos(2 * np.pi * g[:, 1] / L) + 0.02 * rng.normal(size=len(g))
    head = np.concatenate([g, z[:, None]], 1).astype(np.float32)
    box = np.diag([L, L, 12.0]).astype(np.float32)
    if tric:
        box[0, 1] = 0.3 * L          # b = (0.3L, L, 0): sheared in-plane periodicity
        head[:, 0] += 0.3 * head[:, 1]
    K = len(head)
    nrm = np.tile(np.array([0, 0, 1], np.float32), (K, 1))
    nrm += 0.05 * rng.normal(size=nrm.shape).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    ob = orc32.box_from_matrix(box)
    poff, pids = _patches_from_oracle(orc32, ob, head, 2.0)
    st = api.new_membrane_state(head, nrm, None, len(pids))
    eng.membrane_smooth(box, st, poff, pids)
    st["smoothed_head"] = st["head_markers"]
    w32 = orc32.membrane_smooth(ob, head, nrm, np.ones(K, np.uint8), poff, pids)
    check_smooth(st, w32, poff, K)
    w64 = orc64.membrane_smooth(orc64.box_from_matrix(box), head, nrm, np.ones(K, np.uint8), poff, pids)
    ok = st["valid"].astype(bool) & w64["valid"].astype(bool)
    assert ok.sum() == K
    for gk, wk in (("mean_curv", "mean_curv"), ("gauss_curv", "gauss_curv"), ("area", "area"), ("normals", "normals"),
                   ("smoothed_head", "head"), ("quad_coefs", "coefs")):
        assert np.allclose(st[gk][ok], w64[wk][ok], rtol=2e-3, atol=2e-3), gk
    assert 5.5 < st["nvert"].mean() < 6.5
    assert abs(st["area"].sum() - L * L) < 0.02 * L * L
    # smoothing pulls the markers towards the underlying surface: jitter in z shrinks
    zs = 5.0 + 0.3 * np.sin(2 * np.pi * g[:, 0] / L) * np.cos(2 * np.pi * g[:, 1] / L)
    assert np.abs(st["head_markers"][:, 2] - zs).mean() < np.abs(head[:, 2] - zs).mean()


def test_smooth_sphere_curvature_and_invalidation(eng, orc32):
    """Markers on a sphere cap (the reference's test_curvature_sphere geometry, lib.rs:1097-1134, with enough
    points for a patch): curvatures ~ 1/R; rim lipids have open Voronoi cells and turn invalid; lipids that were
    invalid on entry are left untouched."""
    from molar_amd import api
    rng = np.random.default_rng(11)
    R, n = 10.0, 800
    th = np.arccos(1 - rng.random(n) * (1 - np.cos(0.6))); ph = rng.random(n) * 2 * np.pi
    c = np.array([25.0, 25.0, 10.0])
    pts = (np.stack([R * np.sin(th) * np.cos(ph), R * np.sin(th) * np.sin(ph), R * np.cos(th)], 1) + c).astype(np.float32)
    box = np.diag([50.0, 50.0, 50.0]).astype(np.float32)
    nrm = ((pts - c) / R).astype(np.float32)
    ob = orc32.box_from_matrix(box)
    poff, pids = _patches_from_oracle(orc32, ob, pts, 2.5)
    valid = np.ones(n, np.uint8); valid[::17] = 0
    st = api.new_membrane_state(pts, nrm, valid, len(pids))
    eng.membrane_smooth(box, st, poff, pids)
    st["smoothed_head"] = st["head_markers"]
    want = orc32.membrane_smooth(ob, pts, nrm, valid, poff, pids)
    check_smooth(st, want, poff, n)
    ok = st["valid"].astype(bool)
    assert 0.5 * n < ok.sum() < n - n // 17          # interior valid, rim invalid
    assert abs(st["mean_curv"][ok].mean() - 1 / R) < 0.015
    assert np.all(st["princ_curvs"][ok, 0] >= st["princ_curvs"][ok, 1])
    assert np.allclose(st["princ_curvs"][ok].sum(1) / 2, st["mean_curv"][ok], atol=1e-3)
    assert np.allclose(st["princ_curvs"][ok].prod(1), st["gauss_curv"][ok], atol=1e-3)
    off = valid == 0
    assert np.array_equal(st["head_markers"][off], pts[off]) and np.all(st["mean_curv"][off] == -100.0)
    # degenerate normal along x: to_lab is singular (normal x X = 0) -> invalid (lib.rs:675-679)
    nrm2 = nrm.copy(); nrm2[5] = [1, 0, 0]
    st2 = api.new_membrane_state(pts, nrm2, None, len(pids))
    eng.membrane_smooth(box, st2, poff, pids)
    assert st2["valid"][5] == 0


def test_membrane_shells_and_iterations(eng, orc32):
    """n-th shell patches + two smoothing iterations + curvature averaging run end to end and keep the bilayer
    geometry: normals stay aligned with the leaflet, areas near the lattice area per lipid."""
    from molar_amd import membrane as mb
    xyz, box, first, tpl, masses = mb.build_bilayer(400, 60000)
    K = len(first)
    m = mb.Membrane(eng, len(xyz), first, tpl, masses,
                    mb.MembraneOptions(cutoff=2.0, max_smooth_iter=2, n_shells_patch=3, n_shells_smoothing=2))
    res = m.compute(xyz.copy(), box)
    ok = res["valid"].astype(bool)
    assert ok.sum() > 0.95 * K
    up = np.arange(K) < 400
    assert (res["normals"][ok & up, 2] > 0.8).all() and (res["normals"][ok & ~up, 2] < -0.8).all()
    assert abs(res["area"][ok].mean() - 0.62) < 0.05
    assert np.abs(res["mean_curv"][ok]).mean() < 0.2
    # patches are now Voronoi shells: symmetric-ish, much smaller than the 2 nm disc
    assert np.diff(res["patch_off"].astype(np.int64))[ok].mean() < 30


def test_membrane_groups_accumulate(eng, tmp_path):
    """Groups + per-frame statistics over two frames (lib.rs:448-451, lipid_group.rs, stats.rs)."""
    from molar_amd import membrane as mb
    xyz, box, first, tpl, masses = mb.build_bilayer(300, 50000)
    K = len(first)
    m = mb.Membrane(eng, len(xyz), first, tpl, masses, mb.MembraneOptions(cutoff=2.0))
    m.add_ids_to_group("upper", np.arange(300))
    m.add_ids_to_group("lower", np.arange(300, 600))
    with pytest.raises(ValueError):
        m.add_ids_to_group("upper", [K])
    rng = np.random.default_rng(3)
    areas = []
    for f in range(2):
        res = m.compute((xyz + rng.normal(0, 0.01, xyz.shape)).astype(np.float32), box)
        ok = res["valid"].astype(bool)
        areas.append(res["area"][:300][ok[:300]])
    g = m.groups["upper"].per_species["LIP"]
    allarea = np.concatenate(areas)
    mean, std = g.area.compute()
    assert np.isclose(mean, allarea.mean(), rtol=1e-4) and np.isclose(std, allarea.std(), rtol=2e-2)
    assert 250 < g.num_lip.compute()[0] <= 300 and 5 < g.num_neib.compute()[0] < 7
    assert g.tilt.compute()[0] < 25.0                      # normals roughly along the tail-head vectors
    assert np.isclose(g.neib_species["LIP"].compute()[0], g.num_neib.compute()[0], rtol=1e-3)
    m.finalize(tmp_path)
    assert sorted(p.name for p in tmp_path.iterdir()) == ["gr_lower_neib_stats.dat", "gr_lower_order_LIP.dat", "gr_lower_stats.dat",
                                                          "gr_upper_neib_stats.dat", "gr_upper_order_LIP.dat", "gr_upper_stats.dat"]


def test_membrane_smooth_randomised_differential(eng):
    """A 60-case slice of tools/fuzz_membrane.py: undulating, noisy, tilted sheets with holes, sheared boxes, sparse to
    crowded patches, pre-invalidated lipids - validity / neighbour ids / vertex counts exact, floats within 5e-5."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_membrane
    assert fuzz_membrane.run(60, 2, eng) == 0
