"""The oracle's restatement of Membrane::smooth (molar_membrane/src/lib.rs:661-812) against independent
implementations: scipy's Voronoi tessellation (qhull) for the cell topology and areas, numpy's least squares for
the quadric fit, analytic curvature of a sphere / a saddle.  The reference holds no asserting test for this path
(lib.rs:1097-1134 only prints), so this is what pins the oracle."""
import numpy as np
import pytest


def patches(o, ob, head, cutoff):
    r = o.search_single_pbc(cutoff, head, ob, 7)
    K = len(head)
    i = r["i"].astype(np.int64); j = r["j"].astype(np.int64)
    src = np.stack([i, j], 1).reshape(-1); dst = np.stack([j, i], 1).reshape(-1)
    order = np.argsort(src, kind="stable")
    return (np.concatenate([[0], np.cumsum(np.bincount(src, minlength=K))]).astype(np.uint64), dst[order].astype(np.uint64))


@pytest.fixture(scope="module")
def sheet():
    rng = np.random.default_rng(3)
    side = 24
    L = side * 0.8
    g = (np.stack(np.meshgrid(np.arange(side), np.arange(side), indexing="ij"), -1).reshape(-1, 2) + 0.5
         + 0.2 * rng.normal(size=(side * side, 2))) * L / side
    return g, L


def test_voronoi_topology_and_area_vs_qhull(orc64, sheet):
    from scipy.spatial import Voronoi
    g, L = sheet
    K = len(g)
    head = np.concatenate([g, np.full((K, 1), 5.0)], 1)
    box = np.diag([L, L, 10.0])
    ob = orc64.box_from_matrix(box)
    poff, pids = patches(orc64, ob, head, 2.4)
    r = orc64.membrane_smooth(ob, head, np.tile([0.0, 0.0, 1.0], (K, 1)), np.ones(K, np.uint8), poff, pids)
    assert r["valid"].all()
    # qhull on the 3x3 periodic replication; the central copy's regions are the periodic cells
    shifts = np.array([[a, b] for a in (-1, 0, 1) for b in (-1, 0, 1)]) * L
    allp = np.concatenate([g + s for s in shifts])
    base = 4 * K                                  # shifts[4] == (0, 0)
    vor = Voronoi(allp)
    neigh = [set() for _ in range(K)]
    for a, b in vor.ridge_points:
        if base <= a < base + K:
            neigh[a - base].add(b % K)
        if base <= b < base + K:
            neigh[b - base].add(a % K)
    for k in range(K):
        s0 = int(poff[k]) + 4 * k
        got = r["neib_ids"][s0:s0 + int(r["nvert"][k])].tolist()
        assert len(got) == len(set(got))
        assert set(got) == neigh[k], k
        reg = vor.regions[vor.point_region[base + k]]
        poly = vor.vertices[reg]
        x, y = poly[:, 0], poly[:, 1]
        want_area = 0.5 * abs(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1)))
        assert abs(r["area"][k] - want_area) < 1e-9
        # vertices: same point set (order/rotation differ)
        vv = r["voro"][s0:s0 + len(got), :2] - r["head"][k, :2]
        pv = poly - g[k]
        d = np.linalg.norm(vv[:, None, :] - pv[None, :, :], axis=2)
        assert d.min(1).max() < 1e-9
    assert abs(r["area"].sum() - L * L) < 1e-8
    # flat sheet: zero curvature, markers do not move, fitted normal = -(-z) = +z
    assert np.abs(r["mean_curv"]).max() < 1e-9 and np.abs(r["head"] - head).max() < 1e-9
    assert np.allclose(r["normals"], [0, 0, 1])


def test_quadric_fit_vs_lstsq_and_curvature(orc64):
    """Markers on z = a x^2 + b y^2 + c xy (+ tilt) around one lipid: the normal-equation/Cholesky solution equals
    numpy's least squares; curvatures equal the closed forms at the origin."""
    rng = np.random.default_rng(8)
    a, b, c, d, e = 0.04, -0.03, 0.02, 0.05, -0.02
    n = 40
    xy = rng.uniform(-2, 2, size=(n, 2))
    z = a * xy[:, 0] ** 2 + b * xy[:, 1] ** 2 + c * xy[:, 0] * xy[:, 1] + d * xy[:, 0] + e * xy[:, 1]
    centre = np.array([20.0, 20.0, 20.0])
    # local frame for normal (0,0,1): local x = lab y, local y = -lab x, local z = -lab z (lipid_molecule.rs:190-196)
    lab = np.stack([-xy[:, 1], xy[:, 0], -z], 1) + centre
    head = np.concatenate([[centre], lab])
    K = len(head)
    poff = np.concatenate([[0], [n], np.full(K - 1, n)]).astype(np.uint64)
    poff = np.zeros(K + 1, np.uint64); poff[1:] = n          # only lipid 0 has a patch
    pids = np.arange(1, K, dtype=np.uint64)
    valid = np.zeros(K, np.uint8); valid[0] = 1
    ob = orc64.box_from_matrix(np.diag([40.0, 40.0, 40.0]))
    r = orc64.membrane_smooth(ob, head, np.tile([0.0, 0.0, 1.0], (K, 1)), valid, poff, pids)
    A = np.stack([xy[:, 0] ** 2, xy[:, 1] ** 2, xy[:, 0] * xy[:, 1], xy[:, 0], xy[:, 1], np.ones(n)], 1)
    want = np.linalg.lstsq(A, z, rcond=None)[0]
    assert np.allclose(want, [a, b, c, d, e, 0], atol=1e-10)
    # the cell may be open (random points, no wall check here) - coefficients are only stored for closed cells
    if r["valid"][0]:
        assert np.allclose(r["coefs"][0], want, atol=1e-9)
        E, F, G = 1 + d * d, d * e, 1 + e * e
        Lq, M, N = 2 * a, c, 2 * b
        Z = E * G - F * F
        assert np.isclose(r["gauss_curv"][0], (Lq * N - M * M) / Z)
        assert np.isclose(r["mean_curv"][0], 0.5 * (E * N - 2 * F * M + G * Lq) / Z)
        W = np.array([[E * Lq - F * M, E * M - F * N], [G * M - F * Lq, G * N - F * M]]) / Z
        # W is NOT symmetric when the surface is tilted (d, e != 0) although lipid_molecule.rs:179 says so;
        # nalgebra's symmetric_eigen reads only the lower triangle, i.e. it diagonalises [[W00, W10], [W10, W11]]
        ev = np.linalg.eigvalsh(W, UPLO="L")[::-1]
        assert np.allclose(r["princ_curvs"][0], ev, atol=1e-9)
        assert not np.isclose(W[0, 1], W[1, 0])
        # directions: lab images of the local eigenvectors, orthonormal, in the tangent plane of the frame
        pd = r["princ_dirs"][0]
        assert np.allclose(pd @ pd.T, np.eye(2), atol=1e-9) and np.allclose(pd[:, 2], 0, atol=1e-12)
    assert r["valid"][0], "test geometry should give a closed cell"


def test_sphere_and_f32_vs_f64(orc32, orc64):
    rng = np.random.default_rng(11)
    R, n = 10.0, 600
    th = np.arccos(1 - rng.random(n) * (1 - np.cos(0.6))); ph = rng.random(n) * 2 * np.pi
    c = np.array([25.0, 25.0, 10.0])
    pts = np.stack([R * np.sin(th) * np.cos(ph), R * np.sin(th) * np.sin(ph), R * np.cos(th)], 1) + c
    pts = pts.astype(np.float32).astype(np.float64)
    nrm = (pts - c) / R
    box = np.diag([50.0, 50.0, 50.0])
    res = {}
    for name, o in (("f32", orc32), ("f64", orc64)):
        ob = o.box_from_matrix(box)
        poff, pids = patches(o, ob, pts, 2.5)
        res[name] = o.membrane_smooth(ob, pts, nrm, np.ones(n, np.uint8), poff, pids)
    r = res["f64"]
    ok = r["valid"].astype(bool)
    assert 0.5 * n < ok.sum() < n                      # rim cells are open -> invalid
    assert abs(r["mean_curv"][ok].mean() - 1 / R) < 0.015
    assert abs(r["gauss_curv"][ok].mean() - 1 / R ** 2) < 0.003
    assert ((r["normals"][ok] * nrm[ok]).sum(1) > 0.999).all()
    assert np.array_equal(res["f32"]["valid"], r["valid"])
    for k in ("mean_curv", "gauss_curv", "area", "normals", "head"):
        assert np.allclose(res["f32"][k][ok], r[k][ok], rtol=2e-3, atol=2e-3), k


def test_invalid_rules(orc64):
    """Singular local frame, non-positive-definite normal equations (fewer than 6 patch points), open cell,
    |f| > 0.5 each invalidate the lipid (lib.rs:675-679, 692-696, 721-726, 774-777)."""
    o = orc64
    ob = o.box_from_matrix(np.diag([40.0, 40.0, 40.0]))
    ring = np.array([[np.cos(t), np.sin(t)] for t in np.linspace(0, 2 * np.pi, 9)[:-1]])
    pts2 = np.concatenate([ring * 1.0, ring * 1.9 @ np.array([[np.cos(0.3), -np.sin(0.3)], [np.sin(0.3), np.cos(0.3)]])])

    def run(normal, dz=0.0, npts=None):
        p = pts2 if npts is None else pts2[:npts]
        head = np.concatenate([[[20.0, 20.0, 20.0]], np.concatenate([p, np.full((len(p), 1), dz)], 1) + 20.0])
        K = len(head)
        poff = np.zeros(K + 1, np.uint64); poff[1:] = K - 1
        valid = np.zeros(K, np.uint8); valid[0] = 1
        return o.membrane_smooth(ob, head, np.tile(np.asarray(normal, float), (K, 1)), valid, poff,
                                 np.arange(1, K, dtype=np.uint64))
    assert run([0, 0, 1])["valid"][0] == 1
    assert run([1, 0, 0])["valid"][0] == 0            # normal x X = 0
    assert run([0, 0, 1], npts=5)["valid"][0] == 0    # rank-deficient fit
    assert run([0, 0, 1], dz=0.6)["valid"][0] == 0    # surface 0.6 nm off the marker
    r = run([0, 0, 1], dz=0.3)
    assert r["valid"][0] == 1 and abs(abs(r["coefs"][0, 5]) - 0.3) < 1e-9
    half = np.concatenate([[[20.0, 20.0, 20.0]], np.concatenate([pts2[pts2[:, 0] > -0.2], np.zeros((int((pts2[:, 0] > -0.2).sum()), 1))], 1) + 20.0])
    K = len(half)
    poff = np.zeros(K + 1, np.uint64); poff[1:] = K - 1
    valid = np.zeros(K, np.uint8); valid[0] = 1
    assert o.membrane_smooth(ob, half, np.tile([0.0, 0.0, 1.0], (K, 1)), valid, poff, np.arange(1, K, dtype=np.uint64))["valid"][0] == 0
