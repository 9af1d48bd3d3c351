"""GPU parity of the f64 Measure / Modify entries (MolAR's `f64` feature, molar/src/aliases.rs:10-13) against the
oracle's f64 build.  Both sides do every per-atom term in f64; the engine's sums are grouped per thread / wave /
workgroup, the oracle's are serial: agreement is at the level of f64 summation noise, stated below."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL = 1e-12          # reductions of <= 1e6 f64 terms
RTOL_ROT = 1e-10      # Horn's quaternion (engine) against the oracle's SVD-based Kabsch


@pytest.fixture(scope="module")
def m64():
    from molar_amd import build
    from molar_amd.api import Engine, MeasureF64
    build.build_library()
    return MeasureF64(Engine(0))


def _system(n, seed, far=False):
    from molar_amd import api
    rng = np.random.default_rng(seed)
    centre = rng.uniform(-400, 400, 3) if far else rng.uniform(0, 20, 3)
    ref = centre + rng.normal(0, 3.0, (n, 3))
    R = api.rotation_from_axis_angle(rng.normal(size=3), float(rng.uniform(-3, 3))).astype(np.float64)
    cur = (ref - centre) @ R.T + centre + rng.uniform(-5, 5, 3) + rng.normal(0, 0.05, (n, 3))
    mass = rng.uniform(1, 40, n)
    return np.ascontiguousarray(cur), np.ascontiguousarray(ref), mass, rng


@pytest.mark.parametrize("n,m,far", [(50_000, 5_000, False), (1_000_000, 100_000, False), (20_000, 20_000, True), (7, 3, False)])
def test_f64_measure_matches_f64_oracle(m64, orc64, n, m, far):
    cur, ref, mass, rng = _system(n, 100 + n % 97, far)
    idx = None if m == n else np.sort(rng.choice(n, m, replace=False)).astype(np.uint64)
    scale = max(np.abs(cur).max(), np.abs(ref).max())
    assert np.allclose(m64.center_of_geometry(cur, idx), orc64.center_of_geometry(cur, idx), rtol=0, atol=RTOL * scale)
    assert np.allclose(m64.center_of_mass(cur, mass, idx), orc64.center_of_mass(cur, mass, idx), rtol=0, atol=RTOL * scale)
    assert m64.gyration(cur, mass, idx) == pytest.approx(orc64.gyration(cur, mass, idx), rel=RTOL)
    assert m64.rmsd(cur, ref, idx, idx) == pytest.approx(orc64.rmsd(cur, ref, idx, idx), rel=RTOL)
    assert m64.rmsd_mw(cur, mass, ref, idx, idx) == pytest.approx(orc64.rmsd_mw(cur, mass, ref, idx, idx), rel=RTOL)
    # (at_origin on a cloud 400 nm from the origin is a rank-one covariance plus noise: neither side's rotation is
    # determined to better than ~1e-7 there, so that combination is left out)
    lo, hi = m64.min_max(cur, idx)
    rlo, rhi = orc64.min_max(cur, idx)
    assert np.array_equal(lo, rlo) and np.array_equal(hi, rhi)
    mom, axes, tens = m64.inertia(cur, mass, idx)
    rmom, raxes = orc64.inertia(cur, mass, idx)
    rt = orc64.inertia_tensor(cur, mass, idx)
    assert np.allclose(tens, rt, rtol=0, atol=1e-11 * np.abs(rt).max())
    assert np.allclose(mom, rmom, rtol=1e-10)
    assert np.allclose(axes.T @ axes, np.eye(3), atol=1e-13) and np.linalg.det(axes) == pytest.approx(1.0, abs=1e-13)
    assert np.allclose(axes @ np.diag(mom) @ axes.T, rt, rtol=0, atol=1e-10 * np.abs(rt).max())   # axis signs are open
    tr = cur.copy()
    m64.translate(tr, [0.25, -1.5, 3.0], idx)
    assert np.array_equal(tr, orc64.translate(cur, [0.25, -1.5, 3.0], idx))
    for at_origin in ((False,) if far else (False, True)):
        R, t = m64.fit_transform(cur, mass, ref, mass, idx, idx, at_origin=at_origin)
        Ro, to = (orc64.fit_transform_at_origin(cur, mass, ref, idx, idx) if at_origin
                  else orc64.fit_transform(cur, mass, ref, mass, idx, idx))
        assert np.allclose(R, Ro, rtol=0, atol=RTOL_ROT)
        assert np.allclose(t, to, rtol=0, atol=RTOL_ROT * scale * 10)
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-13) and np.linalg.det(R) == pytest.approx(1.0, abs=1e-13)
    R, t = m64.fit_transform(cur, mass, ref, mass, idx, idx)
    moved = cur.copy()
    m64.apply_transform(moved, R, t, idx)
    assert np.array_equal(moved, orc64.apply_transform(cur, R, t, idx))          # same f64 operations, same bits
    # the fit brings the selection onto the reference: RMSD ~ the noise that was added, far below the f32 floor
    assert m64.rmsd(moved, ref, idx, idx) == pytest.approx(orc64.rmsd(orc64.apply_transform(cur, R, t, idx), ref, idx, idx), rel=1e-9)
    assert m64.rmsd(moved, ref, idx, idx) < 0.2


def test_f64_resolves_what_f32_cannot(m64, orc64):
    """A selection 400 nm from the origin moved by 1e-9 nm: the f64 entries see it (RMSD to 1e-3 relative), f32 cannot
    represent the displacement at all (ulp(400) = 3e-5)."""
    rng = np.random.default_rng(5)
    n = 4096
    ref = np.array([400.0, -350.0, 380.0]) + rng.normal(0, 1.0, (n, 3))
    cur = ref + rng.normal(0, 1e-9, (n, 3))
    got, want = m64.rmsd(cur, ref), orc64.rmsd(cur, ref)
    assert got == pytest.approx(want, rel=1e-12) and 1e-9 < got < 3e-9
    assert (np.float32(cur) != np.float32(ref)).mean() < 1e-3        # only values that sat on a rounding boundary differ


def test_f64_device_resident_and_errors(m64, orc64):
    import torch
    from molar_amd.api import MolarHipError
    cur, ref, mass, rng = _system(30_000, 9)
    idx = np.sort(rng.choice(30_000, 3_000, replace=False)).astype(np.uint64)
    d_cur, d_ref, d_mass = (torch.from_numpy(a).cuda() for a in (cur, ref, mass))
    d_idx = torch.from_numpy(idx.astype(np.int64)).cuda()
    assert m64.gyration(d_cur, d_mass, d_idx) == m64.gyration(cur, mass, idx)              # same kernels, same grouping
    R, t = m64.fit_transform(d_cur, d_mass, d_ref, d_mass, d_idx, d_idx)
    R2, t2 = m64.fit_transform(cur, mass, ref, mass, idx, idx)
    assert np.array_equal(R, R2) and np.array_equal(t, t2)
    m64.apply_transform(d_cur, R, t, d_idx)
    assert np.array_equal(d_cur.cpu().numpy(), orc64.apply_transform(cur, R, t, idx))
    with pytest.raises(MolarHipError) as e:
        m64.center_of_mass(cur, np.zeros_like(mass), idx)
    assert e.value.code == 2                                                                 # MeasureError::ZeroMass
    with pytest.raises(MolarHipError) as e:
        m64.rmsd(cur, ref, idx, idx[:-1])
    assert e.value.code == 1                                                                 # MeasureError::Sizes


@pytest.mark.parametrize("n,m,F,resident", [(20_000, 2_000, 5, False), (1_000_000, 100_000, 3, True)])
def test_f64_fit_rmsd_batch(m64, orc64, n, m, F, resident):
    """The per-frame loop (benches/comparison_small.rs:14-25) in f64, also at the C3 size, against the f64 oracle run
    frame by frame: fit, apply, then rmsd / centre of mass / gyration of the moved selection."""
    from molar_amd import api
    rng = np.random.default_rng(21)
    ref = rng.uniform(0, 20, (n, 3))
    mass = rng.uniform(1, 40, n)
    idx = np.sort(rng.choice(n, m, replace=False)).astype(np.uint64)
    ref_idx = idx if resident else np.sort(rng.choice(n, m, replace=False)).astype(np.uint64)
    frames = np.empty((F, n, 3))
    for f in range(F):
        R = api.rotation_from_axis_angle(rng.normal(size=3), float(rng.uniform(-3, 3))).astype(np.float64)
        frames[f] = rng.uniform(0, 20, (n, 3))
        frames[f][idx.astype(np.int64)] = ref[ref_idx.astype(np.int64)] @ R.T + rng.uniform(-3, 3, 3) + rng.normal(0, 10.0 ** -(2 + 3 * f), (m, 3))
    work = frames.copy()
    if resident:
        import torch
        d = torch.from_numpy(work).cuda()
        out = m64.fit_rmsd_batch(d, torch.from_numpy(mass).cuda(), torch.from_numpy(ref).cuda(),
                                 idx=torch.from_numpy(idx.astype(np.int64)).cuda(), apply=True)
        work = d.cpu().numpy()
    else:
        out = m64.fit_rmsd_batch(work, mass, ref, idx=idx, ref_idx=ref_idx, apply=True)
    for f in range(F):
        Ro, to = orc64.fit_transform(frames[f], mass, ref, mass, idx, ref_idx)
        assert np.allclose(out["R"][f], Ro, rtol=0, atol=RTOL_ROT)
        assert np.allclose(out["t"][f], to, rtol=0, atol=RTOL_ROT * 400)
        moved = orc64.apply_transform(frames[f], out["R"][f], out["t"][f], idx)
        assert np.array_equal(work[f], moved)                                  # same f64 operations, same bits
        # residuals down to 1e-11 nm (the last frames) are resolved: the RMSD comes from the fitted positions themselves
        assert out["rmsd"][f] == pytest.approx(orc64.rmsd(moved, ref, idx, ref_idx), rel=1e-9)
        assert np.allclose(out["com"][f], orc64.center_of_mass(moved, mass, idx), rtol=0, atol=1e-11 * 20)
        assert out["gyration"][f] == pytest.approx(orc64.gyration(moved, mass, idx), rel=1e-11)


@pytest.mark.parametrize("boxkind", ["triclinic", "orthorhombic"])
def test_f64_periodic_centres_gyration_unwrap(m64, orc64, boxkind):
    """center_of_*_pbc_dims, gyration_pbc and unwrap_simple_dim with an f64 PeriodicBox, against the oracle's f64 build."""
    from molar_amd import synth
    box = synth.box_a(20000).astype(np.float64) if boxkind == "triclinic" else np.diag([6.0, 7.5, 5.25])
    rng = np.random.default_rng(8)
    blob = rng.normal(0, 0.4, (3000, 3)) + rng.uniform(0, 5, 3)
    inv = np.linalg.inv(box)
    wrapped = np.ascontiguousarray(((blob @ inv.T) % 1.0) @ box.T)          # split over the periodic images
    m = rng.uniform(1, 16, 3000)
    idx = np.sort(rng.choice(3000, 1200, replace=False)).astype(np.uint64)
    b64 = orc64.box_from_matrix(box)
    for sel in (None, idx):
        for dims in (7, 3, 5):
            assert np.allclose(m64.center_of_mass_pbc(wrapped, m, box, dims, sel),
                               orc64.center_of_mass_pbc_dims(wrapped, m, b64, dims, sel), rtol=0, atol=1e-12)
            assert np.allclose(m64.center_of_geometry_pbc(wrapped, box, dims, sel),
                               orc64.center_of_geometry_pbc_dims(wrapped, b64, dims, sel), rtol=0, atol=1e-12)
        assert m64.gyration_pbc(wrapped, m, box, sel) == pytest.approx(orc64.gyration_pbc(wrapped, m, b64, sel), rel=1e-12)
        un = wrapped.copy()
        m64.unwrap_simple(un, box, 7, sel)
        assert np.array_equal(un, orc64.unwrap_simple_dim(wrapped, b64, 7, sel))      # same f64 operations, same bits
    # the unwrapped blob is whole again: its plain gyration radius is the periodic one, up to the reference's centre
    # quirk (center_of_mass_pbc adds the first position unweighted, measure.rs:197-220: the centre is off by ~1e-3 nm)
    assert m64.gyration(un, m, idx) == pytest.approx(m64.gyration_pbc(wrapped, m, box, idx), rel=1e-3)
    from molar_amd.api import MolarHipError
    with pytest.raises(MolarHipError) as e:
        m64.lib.molar_hip_gyration_pbc_f64    # bound
        from molar_amd._lib import check
        import ctypes as C
        out = C.c_double(0)
        check(m64.lib.molar_hip_gyration_pbc_f64(m64.ctx, wrapped.ctypes.data, 3000, None, 0, m.ctypes.data, None, C.byref(out)))
    assert e.value.code == 4                                                           # PeriodicBoxError::NoPbc


def test_f64_randomised_differential():
    """A 200-case slice of tools/fuzz_measure_f64.py: every f64 entry on random selections (whole, sorted, contiguous,
    unsorted), clouds up to 500 nm from the origin, residuals 1e-9..1e-1 nm, batches of 1..6 frames, orthorhombic and
    triclinic boxes with every periodicity mask - against the oracle's f64 build."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_measure_f64
    assert fuzz_measure_f64.run(200, 3) == 0


def test_f64_principal_transform_and_rotate(m64):
    """measure.rs:100-109,646-649 and modify.rs:25-30 in f64: after principal_transform the inertia tensor is diagonal
    with ascending moments (to f64 working precision) and the centre of mass has not moved; rotate is Rodrigues' formula."""
    from molar_amd import api
    rng = np.random.default_rng(12)
    n = 6000
    xyz = rng.normal(0, 1.0, (n, 3)) * [3.0, 1.5, 0.6] + [12.0, -7.0, 30.0]
    xyz = np.ascontiguousarray(xyz @ api.rotation_from_axis_angle([0.3, -0.5, 0.8], 0.7).astype(np.float64).T)
    mass = rng.uniform(1, 30, n)
    idx = np.arange(0, n, 2, dtype=np.uint64)
    cm0 = m64.center_of_mass(xyz, mass, idx)
    R, t = m64.principal_transform(xyz, mass, idx)
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-13) and np.isclose(np.linalg.det(R), 1.0, atol=1e-13)
    moved = xyz.copy()
    m64.apply_transform(moved, R, t, idx)
    assert np.allclose(m64.center_of_mass(moved, mass, idx), cm0, rtol=0, atol=1e-12)
    mom, axes, tens = m64.inertia(moved, mass, idx)
    off = tens - np.diag(np.diag(tens))
    assert np.abs(off).max() < 1e-11 * np.abs(np.diag(tens)).max()
    assert np.all(np.diff(np.diag(tens)) >= 0)                      # ascending moments along x, y, z
    assert np.array_equal(moved[1::2], xyz[1::2])
    ax = np.array([1.0, 2.0, -0.5]); ax /= np.linalg.norm(ax)
    rot = xyz.copy()
    m64.rotate(rot, ax, 0.9, idx)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    Rr = np.eye(3) + np.sin(0.9) * K + (1 - np.cos(0.9)) * (K @ K)
    assert np.allclose(rot[::2], xyz[::2] @ Rr.T, rtol=0, atol=1e-13 * 40)
    assert np.array_equal(rot[1::2], xyz[1::2])


@pytest.mark.parametrize("order_type", [0, 1, 2])
def test_f64_lipid_tail_order(m64, orc64, order_type):
    """Measure::lipid_tail_order (measure.rs:270-422) in f64 over 300 random tails (14-18 carbons, a double bond in every
    third one, one normal per tail or per bond) against the oracle's f64 build, tail by tail."""
    rng = np.random.default_rng(40 + order_type)
    natoms, used = 8000, 0
    xyz = rng.uniform(0, 10, (natoms, 3))
    tails, bonds, normals = [], [], []
    for t in range(300):
        n = int(rng.integers(14, 19))
        idx = np.arange(used, used + n); used += n
        p = np.zeros((n, 3)); p[0] = rng.uniform(1, 9, 3)
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        for k in range(1, n):
            d = d + 0.9 * rng.normal(size=3); d /= np.linalg.norm(d)
            p[k] = p[k - 1] + 0.153 * d
        xyz[idx] = p
        bo = np.ones(n - 1, np.uint8)
        if t % 3 == 0:
            bo[int(rng.integers(1, n - 3))] = 2
        nn = 1 if t % 2 == 0 else n - 2
        nv = rng.normal(size=(nn, 3)); nv /= np.linalg.norm(nv, axis=1)[:, None]
        tails.append(idx.astype(np.uint64)); bonds.append(bo); normals.append(nv)
    got = m64.lipid_tail_order(xyz, tails, order_type, normals, bonds)
    for t in range(300):
        want = orc64.lipid_tail_order(xyz, order_type, normals[t], bonds[t], tails[t])
        assert np.allclose(got[t], want, rtol=0, atol=1e-12, equal_nan=True), t
