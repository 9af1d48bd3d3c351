"""GPU parity of the Measure / Modify entry points against the CPU oracle.

Tolerance (north_star): 1e-5 relative against the reference arithmetic.  The reference sums
1e4-1e5 f32 terms serially, which itself carries ~1e-5..1e-4 relative noise, so
floats are compared with the oracle's f64 build (MolAR's own `f64` feature) at 1e-5, and with
the f32-faithful build at a stated looser bound."""
import numpy as np
import pytest

from molar_amd import synth

pytestmark = pytest.mark.gpu

RTOL64 = 1e-5     # vs the f64 oracle
RTOL32 = 3e-4     # vs the serial-f32 oracle (its own summation noise)


@pytest.fixture(scope="module")
def eng():
    from molar_amd import build
    from molar_amd.api import Engine
    build.build_library()
    return Engine(0)


@pytest.fixture(scope="module")
def system():
    n = 50_000
    box = synth.box_a(n)
    xyz = synth.frame(n, box, 0)
    rng = np.random.default_rng(3)
    ang = 0.9
    axis = np.array([0.3, -0.5, 0.8]); axis /= np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    Rtrue = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K
    xyz2 = (xyz.astype(np.float64) @ Rtrue.T + np.array([1.5, -2.0, 0.7]) + rng.normal(0, 0.05, xyz.shape)).astype(np.float32)
    return dict(n=n, box=box, xyz=xyz, xyz2=xyz2, mass=synth.masses(n), idx=np.arange(0, n, 10, dtype=np.uint64))


def test_reductions(eng, orc32, orc64, system):
    s = system
    for idx in (s["idx"], None):
        for o, rtol in ((orc64, RTOL64), (orc32, RTOL32)):
            assert np.allclose(eng.center_of_geometry(s["xyz"], idx), o.center_of_geometry(s["xyz"], idx), rtol=rtol)
            assert np.allclose(eng.center_of_mass(s["xyz"], s["mass"], idx), o.center_of_mass(s["xyz"], s["mass"], idx), rtol=rtol)
            assert eng.gyration(s["xyz"], s["mass"], idx) == pytest.approx(o.gyration(s["xyz"], s["mass"], idx), rel=rtol)
            assert eng.rmsd(s["xyz"], s["xyz2"], idx, idx) == pytest.approx(o.rmsd(s["xyz"], s["xyz2"], idx, idx), rel=rtol)
            assert eng.rmsd_mw(s["xyz"], s["mass"], s["xyz2"], idx, idx) == pytest.approx(
                o.rmsd_mw(s["xyz"], s["mass"], s["xyz2"], idx, idx), rel=rtol)
        lo, up = eng.min_max(s["xyz"], idx)
        rlo, rup = orc32.min_max(s["xyz"], idx)
        assert np.array_equal(lo, rlo) and np.array_equal(up, rup)          # exact


def test_inertia(eng, orc64, system):
    s = system
    mom, axes, tens = eng.inertia(s["xyz"], s["mass"], s["idx"])
    rmom, raxes = orc64.inertia(s["xyz"], s["mass"], s["idx"])
    rt = orc64.inertia_tensor(s["xyz"], s["mass"], s["idx"])
    assert np.allclose(tens, rt, rtol=RTOL64, atol=RTOL64 * np.abs(rt).max())
    assert np.allclose(mom, rmom, rtol=1e-4)
    assert np.allclose(axes.T @ axes, np.eye(3), atol=1e-5) and np.linalg.det(axes) == pytest.approx(1.0, abs=1e-5)
    # near-degenerate moments of a uniform box: compare the invariant A diag(m) A^T instead of axis signs
    assert np.allclose(axes @ np.diag(mom) @ axes.T, rt, rtol=1e-4, atol=1e-4 * np.abs(rt).max())


def test_fit_transform_and_apply(eng, orc32, orc64, system):
    s = system
    idx = s["idx"]
    R, t = eng.fit_transform(s["xyz"], s["mass"], s["xyz2"], s["mass"], idx, idx)
    R64, t64 = orc64.fit_transform(s["xyz"], s["mass"], s["xyz2"], s["mass"], idx, idx)
    assert np.allclose(R, R64, atol=1e-5)
    assert np.allclose(t, t64, rtol=1e-5, atol=2e-4)
    R0, t0 = eng.fit_transform(s["xyz"], s["mass"], s["xyz2"], s["mass"], idx, idx, at_origin=True)
    Ro, to = orc64.fit_transform_at_origin(s["xyz"], s["mass"], s["xyz2"], idx, idx)
    assert np.allclose(R0, Ro, atol=1e-5) and not t0.any()
    # apply_transform: same f32 arithmetic as the reference -> bit-identical coordinates
    moved = s["xyz"].copy()
    eng.apply_transform(moved, R, t, idx)
    ref = orc32.apply_transform(s["xyz"], R, t, idx)
    assert np.array_equal(moved, ref)
    assert eng.rmsd(moved, s["xyz2"], idx, idx) < 0.1


def test_pbc_centres_gyration_unwrap(eng, orc32, orc64):
    box = synth.box_a(20000)
    rng = np.random.default_rng(5)
    blob = rng.normal(0, 0.4, (3000, 3))
    inv = np.linalg.inv(box.astype(np.float64))
    wrapped = (((blob @ inv.T) % 1.0) @ box.astype(np.float64).T).astype(np.float32)
    m = rng.uniform(1, 16, 3000).astype(np.float32)
    b64 = orc64.box_from_matrix(box); b32 = orc32.box_from_matrix(box)
    for dims in (7, 3):
        assert np.allclose(eng.center_of_mass_pbc(wrapped, m, box, dims),
                           orc64.center_of_mass_pbc_dims(wrapped, m, b64, dims), rtol=RTOL64, atol=1e-5)
        assert np.allclose(eng.center_of_geometry_pbc(wrapped, box, dims),
                           orc64.center_of_geometry_pbc_dims(wrapped, b64, dims), rtol=RTOL64, atol=1e-5)
    assert eng.gyration(wrapped, m, None, box) == pytest.approx(orc64.gyration_pbc(wrapped, m, b64), rel=1e-4)
    un = wrapped.copy()
    eng.unwrap_simple(un, box, 7)
    assert np.array_equal(un, orc32.unwrap_simple_dim(wrapped, b32, 7))       # same f32 arithmetic: exact


def test_errors(eng, system):
    from molar_amd._lib import MolarHipError
    s = system
    with pytest.raises(MolarHipError) as e:
        eng.rmsd(s["xyz"], s["xyz2"], s["idx"], s["idx"][:-1])
    assert e.value.code == 1                                               # MeasureError::Sizes
    with pytest.raises(MolarHipError) as e:
        eng.center_of_mass(s["xyz"], np.zeros(s["n"], np.float32), s["idx"])
    assert e.value.code == 2                                               # MeasureError::ZeroMass
    with pytest.raises(MolarHipError) as e:
        eng.center_of_mass_pbc(s["xyz"], s["mass"], None, 7, s["idx"])
    assert e.value.code == 4                                               # PeriodicBoxError::NoPbc


def test_fit_rmsd_batch(eng, orc32, orc64):
    """Per-frame loop of benches/comparison_small.rs:14-25 batched over frames."""
    n, F = 20000, 6
    box = synth.box_a(n)
    ref = synth.frame(n, box, 0)
    frames = np.stack([synth.frame(n, box, f + 1) for f in range(F)])
    # rotate/translate each frame differently so the fit is not the identity
    for f in range(F):
        a = 0.3 * (f + 1)
        Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
        frames[f] = (frames[f].astype(np.float64) @ Rz.T + [0.5 * f, -1.0, 2.0]).astype(np.float32)
    mass = synth.masses(n)
    idx = np.arange(0, n, 10, dtype=np.uint64)
    work = frames.copy()
    out = eng.fit_rmsd_batch(work, mass, ref, idx=idx, apply=True)
    for f in range(F):
        R64, t64 = orc64.fit_transform(frames[f], mass, ref, mass, idx, idx)
        assert np.allclose(out["R"][f], R64, atol=1e-5)
        assert np.allclose(out["t"][f], t64, rtol=1e-5, atol=2e-4)
        moved = orc32.apply_transform(frames[f], out["R"][f], out["t"][f], idx)
        assert np.array_equal(work[f], moved)                              # apply: exact f32 arithmetic
        assert out["rmsd"][f] == pytest.approx(orc64.rmsd(moved, ref, idx, idx), rel=RTOL64)
        assert np.allclose(out["com"][f], orc64.center_of_mass(moved, mass, idx), rtol=RTOL64)
        assert out["gyration"][f] == pytest.approx(orc64.gyration(moved, mass, idx), rel=RTOL64)
        assert out["rmsd"][f] < 0.2


def test_fit_rmsd_batch_packed_equals_per_frame(eng, orc64):
    """Batches of >= 4 frames gather the frame-invariant columns once (k_fit_pack) and read them coalesced; the records
    must be those of the per-frame calls bit for bit, also when the reference selection is a different index set
    (fit_transform takes sel2's own masses for cm2, measure.rs:512) and for a selection that is no multiple of anything."""
    from molar_amd import api
    rng = np.random.default_rng(11)
    n, F, m = 30011, 7, 4099
    ref = rng.uniform(0, 12, (n, 3)).astype(np.float32)
    mass = rng.uniform(1, 40, n).astype(np.float32)
    idx = np.sort(rng.choice(n, m, replace=False)).astype(np.uint64)
    ref_idx = np.sort(rng.choice(n, m, replace=False)).astype(np.uint64)
    frames = np.empty((F, n, 3), np.float32)
    for f in range(F):
        R = api.rotation_from_axis_angle(rng.normal(size=3), float(rng.uniform(-3, 3))).astype(np.float64)
        frames[f] = rng.uniform(0, 12, (n, 3))
        frames[f][idx.astype(np.int64)] = (ref[ref_idx.astype(np.int64)].astype(np.float64) @ R.T + rng.uniform(-3, 3, 3)
                                           + rng.normal(0, 0.05, (m, 3))).astype(np.float32)
    for ri in (idx, ref_idx):
        for apply in (False, True):
            wb = frames.copy()
            batch = eng.fit_rmsd_batch(wb, mass, ref, idx=idx, ref_idx=ri, apply=apply)
            for f in range(F):
                w1 = frames[f:f + 1].copy()
                one = eng.fit_rmsd_batch(w1, mass, ref, idx=idx, ref_idx=ri, apply=apply)
                for k in ("rmsd", "R", "t", "com", "gyration"):
                    assert np.array_equal(batch[k][f], one[k][0]), (k, f)
                assert np.array_equal(wb[f], w1[0])
    # and against the oracle (different index sets)
    out = eng.fit_rmsd_batch(frames.copy(), mass, ref, idx=idx, ref_idx=ref_idx, apply=False)
    for f in range(F):
        R64, t64 = orc64.fit_transform(frames[f], mass, ref, mass, idx, ref_idx)
        assert np.allclose(out["R"][f], R64, atol=1e-5)
        assert np.allclose(out["t"][f], t64, rtol=1e-5, atol=2e-4)
        assert out["rmsd"][f] < 0.2


def _random_tails(rng, ntails, natoms):
    """Random-walk 'lipid tails' of 14-18 carbons with C-C ~0.153 nm and bond angle ~112 deg."""
    xyz = rng.uniform(0, 10, (natoms, 3)).astype(np.float32)
    tails, bonds, normals1, normalsN = [], [], [], []
    used = 0
    for t in range(ntails):
        n = int(rng.integers(14, 19))
        idx = np.arange(used, used + n)
        used += n
        p = np.zeros((n, 3))
        p[0] = rng.uniform(1, 9, 3)
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        for k in range(1, n):
            d = d + 0.9 * rng.normal(size=3); d /= np.linalg.norm(d)
            p[k] = p[k - 1] + 0.153 * d
        xyz[idx] = p.astype(np.float32)
        bo = np.ones(n - 1, np.uint8)
        if n > 8 and t % 2 == 0:
            bo[int(rng.integers(2, n - 4))] = 2          # one double bond, not at the ends
        tails.append(idx.astype(np.uint64)); bonds.append(bo)
        nv = rng.normal(size=3); nv /= np.linalg.norm(nv)
        normals1.append(nv[None, :].astype(np.float32))
        nn = rng.normal(size=(n - 2, 3)); nn /= np.linalg.norm(nn, axis=1)[:, None]
        normalsN.append(nn.astype(np.float32))
    return xyz, tails, bonds, normals1, normalsN


@pytest.mark.parametrize("order_type", [0, 1, 2])
def test_lipid_tail_order_batched(eng, orc32, orc64, order_type):
    rng = np.random.default_rng(11)
    xyz, tails, bonds, n1, nN = _random_tails(rng, 300, 6000)
    for normals in (n1, nN):
        got = eng.lipid_tail_order(xyz, tails, order_type, normals, bonds)
        for t in range(len(tails)):
            want = orc64.lipid_tail_order(xyz, order_type, normals[t], bonds[t], idx=tails[t])
            want32 = orc32.lipid_tail_order(xyz, order_type, normals[t], bonds[t], idx=tails[t])
            assert got[t].shape == want.shape
            # f32 trig (acos/cos) on both sides: agreement with the f32 oracle to a few ulp of O(1) values,
            # with the f64 oracle to f32 roundoff amplified by the acos near |c|=1
            assert np.allclose(got[t], want32, atol=2e-5), (t, got[t], want32)
            assert np.allclose(got[t], want, atol=2e-4)


def test_lipid_tail_order_errors(eng):
    from molar_amd._lib import MolarHipError
    rng = np.random.default_rng(12)
    xyz, tails, bonds, n1, nN = _random_tails(rng, 4, 200)
    with pytest.raises(MolarHipError) as e:
        eng.lipid_tail_order(xyz, [tails[0][:2]], 0, [n1[0]], [bonds[0][:1]])
    assert e.value.code == 7                                       # TailTooShort (measure.rs:281-283)
    with pytest.raises(MolarHipError) as e:
        eng.lipid_tail_order(xyz, [tails[0]], 0, [nN[0][:3]], [bonds[0]])
    assert e.value.code == 8                                       # NormalsCount (:285-287)
    with pytest.raises(MolarHipError) as e:
        eng.lipid_tail_order(xyz, [tails[0]], 1, [n1[0]], [bonds[0][:-1]])
    assert e.value.code == 9                                       # BondOrderCount (:289-291)


def test_lipid_tail_order_misplaced_double_bond(eng):
    """A double bond at bond 0 has no C(i-1), one at the last bond no normal for atom i+1 when normals are per bond:
    the reference indexes out of range there (measure.rs:361-364,385).  The engine refuses the tail and touches
    nothing outside it."""
    from molar_amd._lib import MolarHipError
    rng = np.random.default_rng(13)
    xyz, tails, bonds, n1, nN = _random_tails(rng, 4, 200)
    t, n = tails[0], len(tails[0])
    lead = np.ones(n - 1, np.uint8); lead[0] = 2
    trail = np.ones(n - 1, np.uint8); trail[n - 3] = 2
    for order_type in (1, 2):
        for normals, b in ((n1[0], lead), (nN[0], lead), (nN[0], trail)):
            with pytest.raises(MolarHipError) as e:
                eng.lipid_tail_order(xyz, [t], order_type, [normals], [b])
            assert e.value.code == 50
        # with ONE normal for the whole tail the trailing double bond is inside the reference's ranges
        got = eng.lipid_tail_order(xyz, [t], order_type, [n1[0]], [trail])
        assert np.isfinite(got[0]).all() and got[0].shape == (n - 2,)


def test_empty_selection_is_refused(eng):
    """The reference cannot build an empty selection (sel.rs:13-19); with n == 0 the PBC centre / gyration / inertia /
    unwrap entry points return INVALID_ARGUMENT before any launch instead of reading atom 0."""
    from molar_amd import synth
    from molar_amd._lib import MolarHipError
    box = synth.box_a(1000)
    xyz = synth.frame(1000, box, 0)
    mass = synth.masses(1000)
    none = np.zeros(0, np.uint64)
    for call in (lambda: eng.center_of_mass_pbc(xyz, mass, box, 7, none),
                 lambda: eng.center_of_geometry_pbc(xyz, box, 7, none),
                 lambda: eng.gyration(xyz, mass, none, box=box),
                 lambda: eng.inertia(xyz, mass, none, box=box),
                 lambda: eng.unwrap_simple(xyz.copy(), box, 7, none)):
        with pytest.raises(MolarHipError) as e:
            call()
        assert e.value.code == 50


def test_principal_transform_translate_rotate(eng, orc64):
    """measure.rs:100-109,646-649 and modify.rs:16-30 through the C ABI (molar_hip_principal_transform, _translate,
    _rotate): after principal_transform the inertia
    tensor is diagonal with ascending moments and the centre of mass has not moved; translate/rotate equal the
    float64 formulas."""
    from molar_amd import api, synth
    n = 5000
    box = synth.box_ortho(n)
    xyz = (synth.frame(n, box, 3) * np.array([1.0, 0.6, 0.3], np.float32)).astype(np.float32)
    # skew the cloud so the principal axes are not the lab axes
    Rz = api.rotation_from_axis_angle([0.3, -0.5, 0.8], 0.7)
    xyz = (xyz @ Rz.T).astype(np.float32)
    top = api.Topology(synth.masses(n))
    st = api.State(xyz.copy(), api.PeriodicBox.from_matrix(box))
    sel = api.Sel(top, st, np.arange(0, n, 2), engine=eng)
    cm0 = sel.center_of_mass()
    R, t = sel.principal_transform()
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-5) and np.isclose(np.linalg.det(R), 1.0, atol=1e-5)
    sel.apply_transform((R, t))
    assert np.allclose(sel.center_of_mass(), cm0, atol=2e-4)
    mom, axes = sel.inertia()
    ii = sel.index.astype(np.int64)
    d = st.coords[ii].astype(np.float64) - np.asarray(sel.center_of_mass(), np.float64)
    mm = top.masses[ii].astype(np.float64)
    tens = np.einsum("k,kij->ij", mm, (d * d).sum(1)[:, None, None] * np.eye(3) - d[:, :, None] * d[:, None, :])
    off = tens - np.diag(np.diag(tens))
    assert np.abs(off).max() < 2e-4 * np.abs(np.diag(tens)).max()
    assert np.all(np.diff(np.diag(tens)) >= 0)          # ascending moments along x, y, z
    # translate / rotate
    before = st.coords[sel.index.astype(np.int64)].astype(np.float64)
    untouched = st.coords[1::2].copy()
    sel.translate([0.5, -1.25, 2.0])
    assert np.allclose(st.coords[sel.index.astype(np.int64)], before + [0.5, -1.25, 2.0], atol=1e-5)
    before = st.coords[sel.index.astype(np.int64)].astype(np.float64)
    ax = np.array([1.0, 2.0, -0.5]); ax /= np.linalg.norm(ax)
    sel.rotate(ax, 0.9)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    Rr = np.eye(3) + np.sin(0.9) * K + (1 - np.cos(0.9)) * (K @ K)
    assert np.allclose(st.coords[sel.index.astype(np.int64)], before @ Rr.T, atol=2e-5)
    assert np.array_equal(st.coords[1::2], untouched)


def _random_csr(rng, natoms, nsel, lo=3, hi=200):
    sizes = rng.integers(lo, hi, nsel)
    idx = np.concatenate([np.sort(rng.choice(natoms, int(k), replace=False)) for k in sizes]).astype(np.uint64)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    return idx, off


def test_csr_batched_gyration_rmsd_fit(eng, orc32, orc64):
    """molar_hip_gyration_batch / _rmsd_batch / _fit_batch: one wave per selection, what MolAR runs from rayon over a
    ParSplit (selection/system.rs:193-213).  Every selection against the per-selection oracle calls."""
    from molar_amd import synth
    rng = np.random.default_rng(31)
    n, K = 20000, 300
    box = synth.box_a(n)
    x1 = synth.frame(n, box, 1)
    x2 = synth.frame(n, box, 2)
    mass = synth.masses(n)
    idx, off = _random_csr(rng, n, K)
    sels = [idx[int(off[k]):int(off[k + 1])] for k in range(K)]
    # gyration, plain and periodic
    g = eng.gyration_batch(x1, idx, off, mass)
    gp = eng.gyration_batch(x1, idx, off, mass, box=box)
    ob64 = orc64.box_from_matrix(box)
    for k in range(K):
        w = orc64.gyration(x1, mass, sels[k])
        assert abs(g[k] - w) <= 1e-5 * w, (k, g[k], w)
        wp = orc64.gyration_pbc(x1, mass, ob64, sels[k])
        assert abs(gp[k] - wp) <= 2e-5 * wp, (k, gp[k], wp)
    # rmsd / rmsd_mw
    r = eng.rmsd_batch(x1, x2, idx, off)
    rw = eng.rmsd_batch(x1, x2, idx, off, mass=mass)
    for k in range(K):
        w = orc64.rmsd(x1, x2, sels[k], sels[k]); ww = orc64.rmsd_mw(x1, mass, x2, sels[k], sels[k])
        assert abs(r[k] - w) <= 1e-5 * w and abs(rw[k] - ww) <= 1e-5 * ww
    # fit: every selection of frame 1 onto the same atoms of frame 2, rotated + shifted so the fit has work to do
    Rz = np.asarray(__import__("molar_amd.api", fromlist=["x"]).rotation_from_axis_angle([0.1, 0.7, -0.4], 1.1), np.float64)
    y1 = (x1.astype(np.float64) @ Rz.T + np.array([1.5, -0.5, 0.25])).astype(np.float32)
    out = eng.fit_batch(y1, mass, x2, idx, off, apply=False)
    moved_all = y1.copy()
    for k in range(K):
        R, t = orc64.fit_transform(y1, mass, x2, mass, sels[k], sels[k])
        assert np.allclose(out["R"][k], R, atol=2e-5), k
        assert np.allclose(out["t"][k], t, atol=2e-4), k
        mv = orc64.apply_transform(y1, R, t, sels[k])
        w = orc64.rmsd(mv, x2, sels[k], sels[k])
        assert abs(out["rmsd"][k] - w) <= 2e-5 * max(w, 1e-3), (k, out["rmsd"][k], w)
        assert np.allclose(out["com"][k], orc64.center_of_mass(mv, mass, sels[k]), atol=2e-4)
        wg = orc64.gyration(mv, mass, sels[k])
        assert abs(out["gyration"][k] - wg) <= 2e-5 * wg
    # apply moves exactly the selected atoms; selections overlap, so compare one selection at a time on fresh copies
    for k in (0, 7, K - 1):
        z = y1.copy()
        o1 = eng.fit_batch(z, mass, x2, sels[k], np.array([0, len(sels[k])], np.uint64), apply=True)
        R, t = orc64.fit_transform(y1, mass, x2, mass, sels[k], sels[k])
        want = orc64.apply_transform(y1, R, t, sels[k])
        assert np.abs(z - want).max() < 2e-4
        rest = np.ones(n, bool); rest[sels[k].astype(np.int64)] = False
        assert np.array_equal(z[rest], y1[rest])
    # errors: an empty selection in the batch is refused, zero masses reported
    from molar_amd._lib import MolarHipError
    with pytest.raises(MolarHipError) as e:
        eng.gyration_batch(x1, idx, np.array([0, 5, 5, 9], np.uint64), mass)
    assert e.value.code == 50
    with pytest.raises(MolarHipError) as e:
        eng.gyration_batch(x1, idx, off, np.zeros(n, np.float32))
    assert e.value.code == 2


def test_principal_transform_abi_matches_host_formula(eng, orc64):
    """molar_hip_principal_transform(_pbc) against T(cm) * inverse(axes) * T(-cm) built from the engine's own inertia /
    centre of mass (measure.rs:102-109, 246-257, 646-649)."""
    from molar_amd import synth
    n = 4000
    box = synth.box_a(n)
    xyz = synth.frame(n, box, 2)
    mass = synth.masses(n)
    idx = np.arange(5, n, 3, dtype=np.uint64)
    for bx in (None, box):
        R, t = eng.principal_transform(xyz, mass, idx, box=bx)
        _, axes, _ = eng.inertia(xyz, mass, idx, box=bx)
        cm = eng.center_of_mass(xyz, mass, idx) if bx is None else eng.center_of_mass_pbc(xyz, mass, bx, 7, idx)
        Rw = np.linalg.inv(np.asarray(axes, np.float64))
        assert np.allclose(R, Rw, atol=1e-5)
        assert np.allclose(t, cm + Rw @ (-np.asarray(cm, np.float64)), atol=1e-4)


def test_fit_randomised_differential(eng):
    """A 300-case slice of tools/fuzz_fit.py: selection sizes 3..20000, clouds far from the origin, near-identical
    frames, planar and nearly collinear selections, large rigid motions - batched and single-call fits against the f64
    oracle (RMSD / COM / gyration of the fitted selection always; R and t wherever the rotation is unique)."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_fit
    assert fuzz_fit.run(300, 5, eng) == 0


def test_measure_randomised_differential(eng):
    """A 300-case slice of tools/fuzz_measure.py: centres (plain / periodic, all PbcDims masks), gyration and inertia
    (plain / periodic), rmsd, rmsd_mw, min_max, unwrap_simple and the CSR-batched gyration / rmsd on random selections,
    boxes and blobs, some 400 nm from the origin, against the f64 oracle."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_measure
    assert fuzz_measure.run(300, 4, eng) == 0


def test_lipid_order_randomised_differential(eng):
    """A 150-case slice of tools/fuzz_lipid_order.py: tails of 3..30 carbons, one normal per tail or per bond, up to three
    double bonds, all order types, straight segments (NaN in the reference's formula, reproduced) - against the f32 oracle."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_lipid_order
    assert fuzz_lipid_order.run(150, 2, eng) == 0


def test_csr_batches_with_device_resident_inputs(eng):
    """The CSR-batched entries take device pointers for every array (frames that never leave HBM) and give the same
    numbers as with host arrays; apply moves the device frame in place."""
    import torch
    from molar_amd import synth
    rng = np.random.default_rng(41)
    n, K = 30000, 500
    box = synth.box_a(n)
    x1, x2, mass = synth.frame(n, box, 1), synth.frame(n, box, 2), synth.masses(n)
    idx, off = _random_csr(rng, n, K, 3, 120)
    d = lambda a, t=None: torch.from_numpy(a if t is None else a.astype(t)).cuda()
    dx1, dx2, dm, di, do = d(x1), d(x2), d(mass), d(idx, np.int64), d(off, np.int64)
    assert np.array_equal(eng.gyration_batch(dx1, di, do, dm), eng.gyration_batch(x1, idx, off, mass))
    assert np.array_equal(eng.gyration_batch(dx1, di, do, dm, box=box), eng.gyration_batch(x1, idx, off, mass, box=box))
    assert np.array_equal(eng.rmsd_batch(dx1, dx2, di, do, mass=dm), eng.rmsd_batch(x1, x2, idx, off, mass=mass))
    a = eng.fit_batch(dx1, dm, dx2, di, do, apply=False)
    b = eng.fit_batch(x1, mass, x2, idx, off, apply=False)
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    # disjoint selections so that apply is well defined for the whole batch
    perm = rng.permutation(n)[: 6000].astype(np.uint64)
    idx2 = np.sort(perm.reshape(100, 60), axis=1).reshape(-1); off2 = (np.arange(101) * 60).astype(np.uint64)
    h = x1.copy()
    eng.fit_batch(h, mass, x2, idx2, off2, apply=True)
    dv = dx1.clone()
    eng.fit_batch(dv, dm, dx2, d(idx2, np.int64), d(off2, np.int64), apply=True)
    eng.synchronize()
    assert np.array_equal(dv.cpu().numpy(), h)
    rest = np.ones(n, bool); rest[idx2.astype(np.int64)] = False
    assert np.array_equal(h[rest], x1[rest]) and not np.array_equal(h[~rest], x1[~rest])


def test_fit_stream_equals_the_batch_entry(eng):
    """molar_hip_fit_stream_*: frames in host memory, the selection packed by host threads, three frames in flight.  Every
    record - and, with apply, every moved frame - must equal molar_hip_fit_rmsd_batch's on the same frame bit for bit (same
    kernels on the packed selection, same terms in the same order); also with a reference selection of its own index set,
    with the identity selection, and with a selection too small for the thread pool."""
    from molar_amd import api
    rng = np.random.default_rng(23)
    n, F = 120011, 9
    ref = rng.uniform(0, 12, (n, 3)).astype(np.float32)
    mass = rng.uniform(1, 40, n).astype(np.float32)
    for m, same_ref, threads in ((12007, True, 0), (12007, False, 3), (n, True, 4), (900, True, 0)):
        idx = None if m == n else np.sort(rng.choice(n, m, replace=False)).astype(np.uint64)
        ref_idx = idx if same_ref else np.sort(rng.choice(n, m, replace=False)).astype(np.uint64)
        sel = np.arange(n) if idx is None else idx.astype(np.int64)
        rsel = np.arange(n) if ref_idx is None else ref_idx.astype(np.int64)
        frames = np.empty((F, n, 3), np.float32)
        for f in range(F):
            R = api.rotation_from_axis_angle(rng.normal(size=3), float(rng.uniform(-3, 3))).astype(np.float64)
            frames[f] = rng.uniform(0, 12, (n, 3))
            frames[f][sel] = (ref[rsel].astype(np.float64) @ R.T + rng.uniform(-3, 3, 3) + rng.normal(0, 0.05, (len(sel), 3))).astype(np.float32)
        for apply in (False, True):
            want_frames = frames.copy()
            want = eng.fit_rmsd_batch(want_frames, mass, ref, idx=idx, ref_idx=ref_idx, apply=apply)
            got_frames = frames.copy()
            fs = api.FitStream(eng, n, mass, ref, idx=idx, ref_idx=ref_idx, host_threads=threads)
            got, pending = [None] * F, []
            for f in range(F):
                pending.append((f, fs.begin(got_frames[f], apply=apply)))
                if len(pending) == 3:
                    g, t = pending.pop(0)
                    got[g] = fs.end(t)
            for g, t in pending:
                got[g] = fs.end(t)
            fs.close()
            for f in range(F):
                for k in ("rmsd", "R", "t", "com", "gyration"):
                    assert np.array_equal(got[f][k], want[k][f]), (m, same_ref, apply, k, f)
            assert np.array_equal(got_frames, want_frames)
            assert max(float(g["rmsd"]) for g in got) < 0.2
    # a fourth begin without an end is refused
    fs = api.FitStream(eng, n, mass, ref, idx=None)
    work = frames[:4].copy()
    ts = [fs.begin(work[f]) for f in range(3)]
    from molar_amd._lib import MolarHipError
    with pytest.raises(MolarHipError):
        fs.begin(work[3])
    for t in ts:
        fs.end(t)
    fs.close()
