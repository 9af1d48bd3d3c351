"""The f64 drivers (MolAR built with its `f64` feature: Float = f64, molar/src/aliases.rs:10-13) against the oracle's f64
build: ids, ORDER and distances bit for bit, for all eight drivers on the four fixture boxes, with partial periodicity,
selections and the cases where f64 arithmetic decides differently from f32 (atoms on cell faces, pairs at the cutoff)."""
import numpy as np
import pytest

from molar_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from molar_amd import build
    from molar_amd.api import Engine
    build.build_library()
    return Engine(0)


@pytest.fixture(scope="module")
def orc64():
    from oracle.oracle import Oracle
    return Oracle("f64")


def boxes(n):
    rd = np.array([[1.0, 0.0, 0.5], [0.0, 1.0, 0.5], [0.0, 0.0, np.sqrt(0.5)]])
    rd = rd * ((n / 100.0) / abs(np.linalg.det(rd))) ** (1 / 3)
    return {"ortho": synth.box_ortho(n).astype(np.float64), "tric_a": synth.box_a(n).astype(np.float64),
            "hex_b": synth.box_b(n).astype(np.float64), "rhombic_dodecahedron": rd}


def same(got, ref):
    i, j, d = got
    assert len(i) == len(ref["i"]), (len(i), len(ref["i"]))
    assert np.array_equal(i, ref["i"]) and np.array_equal(j, ref["j"])
    assert d.dtype == np.float64 and np.array_equal(d, ref["d"])        # same f64 operation order, correctly rounded sqrt


@pytest.mark.parametrize("name", ["ortho", "tric_a", "hex_b", "rhombic_dodecahedron"])
def test_all_eight_drivers_bit_exact(eng, orc64, name):
    import molar_amd.api as a
    n, rc = 2400, 0.55
    box = boxes(n)[name]
    rng = np.random.default_rng(17)
    pos = (rng.random((n, 3)) @ box.T + rng.normal(0, 0.08, (n, 3)))                # f64 coordinates, some outside the cell
    ob = orc64.box_from_matrix(box)
    i1 = np.arange(0, n, 2, dtype=np.uint64); i2 = np.arange(1, n, 2, dtype=np.uint64)
    p1, p2 = pos[0::2], pos[1::2]
    vdw = 0.12 + 0.1 * rng.random(n)
    for pbc in (7, 3, 5, 1):
        same(eng.search_f64(a.SEARCH_SINGLE, rc, pos, box=box, pbc=pbc), orc64.search_single_pbc(rc, pos, ob, pbc))
    ref = orc64.search_single_pbc(rc, pos, ob, 7)
    eng.search_f64(a.SEARCH_SINGLE, rc, pos, box=box, pbc=7)
    assert eng.grid_dims_f64() == tuple(int(x) for x in ref["dims"])
    same(eng.search_f64(a.SEARCH_SINGLE, rc, pos), orc64.search_single(rc, pos))
    sel = np.sort(rng.choice(n, n // 3, replace=False)).astype(np.uint64)
    same(eng.search_f64(a.SEARCH_SINGLE, rc, pos, sel, box=box, pbc=7), orc64.search_single_pbc(rc, pos[sel.astype(int)], ob, 7, ids=sel))
    same(eng.search_f64(a.SEARCH_DOUBLE, rc, pos, i1, pos, i2, box=box, pbc=7), orc64.search_double_pbc(rc, p1, p2, ob, 7, ids1=i1, ids2=i2))
    same(eng.search_f64(a.SEARCH_DOUBLE, rc, pos, i1, pos, i2, box=box, pbc=6), orc64.search_double_pbc(rc, p1, p2, ob, 6, ids1=i1, ids2=i2))
    same(eng.search_f64(a.SEARCH_DOUBLE, rc, pos, i1, pos, i2), orc64.search_double(rc, p1, p2, ids1=i1, ids2=i2))
    same(eng.search_f64(a.SEARCH_DOUBLE_VDW, None, pos, i1, pos, i2, box=box, pbc=7, vdw1=vdw[0::2], vdw2=vdw[1::2]),
         orc64.search_double_vdw_pbc(p1, p2, vdw[0::2], vdw[1::2], ob, 7))
    same(eng.search_f64(a.SEARCH_DOUBLE_VDW, None, pos, i1, pos, i2, vdw1=vdw[0::2], vdw2=vdw[1::2]),
         orc64.search_double_vdw(p1, p2, vdw[0::2], vdw[1::2]))
    ids = eng.search_f64(a.SEARCH_WITHIN, rc, pos, i1, pos, i2, box=box, pbc=7)
    assert np.array_equal(ids, orc64.search_within_pbc(rc, p1, p2, ob, 7, ids1=i1, ids2=i2)["i"])
    lo = p1.min(0) - (rc + 2.220446049250313e-16); up = p1.max(0) + (rc + 2.220446049250313e-16)
    ids = eng.search_f64(a.SEARCH_WITHIN, rc, pos, i1, pos, i2, lower=lo, upper=up)
    assert np.array_equal(ids, orc64.search_within(rc, p1, p2, lo, up, ids1=i1, ids2=i2)["i"])


def test_f64_decides_where_f32_cannot(eng, orc64):
    """Pairs at rc * (1 +- 1e-12) and atoms 1e-13 from a cell face: an f32 search cannot tell them apart, the f64 drivers
    must agree with the f64 reference on every one of them."""
    import molar_amd.api as a
    rng = np.random.default_rng(3)
    rc, L = 0.8, 8.0
    box = np.diag([L, L, L])
    npairs = 4000
    pa = 0.2 * L + 0.6 * L * rng.random((npairs, 3))
    u = rng.normal(size=(npairs, 3)); u /= np.linalg.norm(u, axis=1)[:, None]
    e = 10.0 ** rng.uniform(-14.0, -9.0, npairs) * rng.choice([-1.0, 1.0], npairs)
    pb = pa + rc * (1.0 + e)[:, None] * u
    cell = L / np.floor(L / rc)
    faces = np.stack([np.round(rng.random(2000) * 9) * cell + rng.choice([-1e-13, 0.0, 1e-13], 2000), L * rng.random(2000), L * rng.random(2000)], 1)
    pos = np.concatenate([pa, pb, faces, L * rng.random((3000, 3))])
    ob = orc64.box_from_matrix(box)
    ref = orc64.search_single_pbc(rc, pos, ob, 7)
    same(eng.search_f64(a.SEARCH_SINGLE, rc, pos, box=box, pbc=7), ref)
    near = np.abs(ref["d"] / rc - 1.0)
    assert (near < 1e-9).sum() > 1000


@pytest.mark.parametrize("name", ["ortho", "tric_a", "hex_b"])
def test_f64_pairs_across_the_periodic_boundary_at_the_cutoff_edge(eng, orc64, name):
    """Entries across the periodic boundary are classified by the distance to the second cell's adjacent image and decided by
    PeriodicBox::distance_squared only inside a narrow band around the cutoff: pairs straddling every face of the cell at
    rc * (1 +- 1e-15 .. 1e-8), in boxes with >= 4 cells per dimension, must come out exactly as the f64 reference has them -
    single and double search, full and partial periodicity, coordinates resident in HBM."""
    import torch
    import molar_amd.api as a
    rng = np.random.default_rng(23)
    n = 30_000
    box = boxes(n)[name]
    rc = 0.7
    ob = orc64.box_from_matrix(box)
    npairs = 6000
    # first atoms close to a face of the unit cell (fractional coordinate ~0 or ~1 in one dimension), partners at the cutoff edge
    frac = rng.random((npairs, 3))
    dim = rng.integers(0, 3, npairs)
    frac[np.arange(npairs), dim] = rng.choice([0.0, 1.0], npairs) + rng.normal(0, 0.01, npairs)
    pa = frac @ box.T
    u = rng.normal(size=(npairs, 3)); u /= np.linalg.norm(u, axis=1)[:, None]
    e = 10.0 ** rng.uniform(-15.5, -8.0, npairs) * rng.choice([-1.0, 1.0], npairs)
    pb = pa + rc * (1.0 + e)[:, None] * u
    pos = np.concatenate([pa, pb, rng.random((n - 2 * npairs, 3)) @ box.T])
    for pbc in (7, 5):
        ref = orc64.search_single_pbc(rc, pos, ob, pbc)
        assert min(ref["dims"]) >= 4
        same(eng.search_f64(a.SEARCH_SINGLE, rc, pos, box=box, pbc=pbc), ref)
        if pbc == 7:
            assert (np.abs(ref["d"] / rc - 1.0) < 1e-9).sum() > 1000
    dpos = torch.from_numpy(pos).cuda()
    got = eng.search_f64(a.SEARCH_SINGLE, rc, dpos, box=box, pbc=7, device_out=True)
    ref = orc64.search_single_pbc(rc, pos, ob, 7)
    same((got[0].cpu().numpy().view(np.uint64), got[1].cpu().numpy().view(np.uint64), got[2].cpu().numpy()), ref)
    i1 = np.arange(0, n, 2, dtype=np.uint64); i2 = np.arange(1, n, 2, dtype=np.uint64)
    same(eng.search_f64(a.SEARCH_DOUBLE, rc, pos, i1, pos, i2, box=box, pbc=7),
         orc64.search_double_pbc(rc, pos[0::2], pos[1::2], ob, 7, ids1=i1, ids2=i2))
    ids = eng.search_f64(a.SEARCH_WITHIN, rc, pos, i1, pos, i2, box=box, pbc=7)
    assert np.array_equal(ids, orc64.search_within_pbc(rc, pos[0::2], pos[1::2], ob, 7, ids1=i1, ids2=i2)["i"])


def test_f64_large_cells_and_device_inputs(eng, orc64):
    """Second cells of more than 256 atoms take the chunk-by-chunk loop; selections and radii resident in HBM; the non-periodic
    bounding box reduced on the device; an out-of-range selection index is an error."""
    import torch
    import molar_amd.api as a
    from molar_amd._lib import MolarHipError
    rng = np.random.default_rng(29)
    n = 20_000
    box = np.diag([6.0, 6.0, 6.0])
    pos = rng.random((n, 3)) * 6.0 + rng.normal(0, 0.05, (n, 3))
    ob = orc64.box_from_matrix(box)
    rc = 1.45                                     # 4 x 4 x 4 cells of ~310 atoms
    ref = orc64.search_single_pbc(rc, pos, ob, 7)
    assert tuple(ref["dims"]) == (4, 4, 4)
    dpos = torch.from_numpy(pos).cuda()
    same(eng.search_f64(a.SEARCH_SINGLE, rc, dpos, box=box, pbc=7), ref)
    same(eng.search_f64(a.SEARCH_SINGLE, rc, dpos), orc64.search_single(rc, pos))
    sel1 = np.sort(rng.choice(n, 8000, replace=False)).astype(np.uint64)
    sel2 = np.sort(rng.choice(n, 5000, replace=False)).astype(np.uint64)
    d1, d2 = torch.from_numpy(sel1.astype(np.int64)).cuda(), torch.from_numpy(sel2.astype(np.int64)).cuda()
    want = orc64.search_double(0.6, pos[sel1.astype(int)], pos[sel2.astype(int)], ids1=sel1, ids2=sel2)
    same(eng.search_f64(a.SEARCH_DOUBLE, 0.6, dpos, d1, dpos, d2), want)
    v1, v2 = 0.1 + 0.1 * rng.random(len(sel1)), 0.1 + 0.1 * rng.random(len(sel2))
    same(eng.search_f64(a.SEARCH_DOUBLE_VDW, None, dpos, d1, dpos, d2, box=box, pbc=7, vdw1=torch.from_numpy(v1).cuda(), vdw2=torch.from_numpy(v2).cuda()),
         orc64.search_double_vdw_pbc(pos[sel1.astype(int)], pos[sel2.astype(int)], v1, v2, ob, 7))
    bad = sel1.copy(); bad[17] = n + 5
    with pytest.raises(MolarHipError):
        eng.search_f64(a.SEARCH_SINGLE, 0.5, pos, bad, box=box, pbc=7)


def test_errors_f64(eng):
    import molar_amd.api as a
    from molar_amd._lib import MolarHipError
    pos = np.random.default_rng(0).random((100, 3))
    with pytest.raises(MolarHipError):
        eng.search_f64(a.SEARCH_SINGLE, -1.0, pos)
    with pytest.raises(MolarHipError):
        eng.search_f64(a.SEARCH_WITHIN, 0.5, pos, None, pos, None)          # needs lower / upper without a box
    i, j, d = eng.search_f64(a.SEARCH_SINGLE, 0.01, pos[:1])                # one atom: nothing to pair
    assert len(i) == 0


@pytest.mark.timeout(900)
def test_f64_half_a_million_atoms_against_the_f64_oracle(eng, orc64):
    """The f64 single-selection search at 500k atoms (triclinic box A, rc 1.0 nm, ~1e8 results): grid by the stable device sort,
    plan and scans on the device, bounding-box row pruning, adjacent-image classification, LDS output queue - ids, order and
    distances bit for bit against the f64 oracle on the box's host cores, frame and result resident in HBM."""
    import os
    import torch
    import molar_amd.api as a
    n, rc = 500_000, 1.0
    box = synth.box_a(n).astype(np.float64)
    pos = synth.frame(n, synth.box_a(n), 3).astype(np.float64)
    pos += np.random.default_rng(2).normal(0, 1e-9, pos.shape)                  # digits an f32 frame does not have
    ob = orc64.box_from_matrix(box)
    ref = orc64.search_single_pbc(rc, pos, ob, 7, nthreads=min(os.cpu_count() or 4, 64))
    dpos = torch.from_numpy(pos).cuda()
    i, j, d = eng.search_f64(a.SEARCH_SINGLE, rc, dpos, box=box, pbc=7, device_out=True)
    assert len(i) == len(ref["i"]) > 5e7 and eng.grid_dims_f64() == tuple(int(x) for x in ref["dims"])
    step = 1 << 24
    for k in range(0, len(i), step):
        assert np.array_equal(i[k:k + step].cpu().numpy().view(np.uint64), ref["i"][k:k + step]), k
        assert np.array_equal(j[k:k + step].cpu().numpy().view(np.uint64), ref["j"][k:k + step]), k
        assert np.array_equal(d[k:k + step].cpu().numpy(), ref["d"][k:k + step]), k
