"""The chained frame call (molar_hip_membrane_frame_begin / _end / _fetch: unwrap -> markers -> marker search -> patches
-> initial normals -> smoothing -> order without a host round trip) against the stage-by-stage calls it replaces: the
same kernels and the same arithmetic, so every array has to agree bit for bit.  The stage-by-stage path is compared with
the CPU checker in tests/test_gpu_membrane.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ARRAYS = ("head", "mid", "tail", "patch_off", "patch_ids", "initial_normals", "valid", "smoothed_head", "normals", "quad_coefs",
          "mean_curv", "gauss_curv", "princ_curvs", "princ_dirs", "area", "nvert", "neib_ids", "voro_vertexes", "fitted_patch_points")


@pytest.fixture(scope="module")
def eng():
    from molar_amd import build
    from molar_amd.api import Engine
    build.build_library()
    return Engine(0)


def same_bits(a, b, what):
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
    assert a.dtype == b.dtype and a.shape == b.shape, (what, a.dtype, b.dtype, a.shape, b.shape)
    if a.tobytes() != b.tobytes():
        bad = np.flatnonzero(a.reshape(-1).view(np.uint8 if a.dtype.itemsize == 1 else f"u{a.dtype.itemsize}")
                             != b.reshape(-1).view(np.uint8 if b.dtype.itemsize == 1 else f"u{b.dtype.itemsize}"))
        raise AssertionError(f"{what}: {len(bad)} of {a.size} elements differ, first at {bad[:5]}: {a.reshape(-1)[bad[:5]]} vs {b.reshape(-1)[bad[:5]]}")


def same_result(got, want, what=""):
    for k in ARRAYS:
        same_bits(got[k], want[k], f"{what}{k}")
    assert len(got["order"]) == len(want["order"])
    for t, (a, b) in enumerate(zip(got["order"], want["order"])):
        same_bits(a, b, f"{what}order[{t}]")


def pair(eng, per_leaflet, natoms, **opts):
    from molar_amd import membrane as mb
    xyz, box, first, tpl, masses = mb.build_bilayer(per_leaflet, natoms)
    mk = lambda fused: mb.Membrane(eng, len(xyz), first, tpl, masses, mb.MembraneOptions(fused=fused, **opts))
    return xyz, box, mk(True), mk(False)


def frames_of(xyz, n, seed=5, sigma=0.02):
    rng = np.random.default_rng(seed)
    return [(xyz + rng.normal(0, sigma, xyz.shape)).astype(np.float32) for _ in range(n)]


@pytest.mark.parametrize("opts", [dict(cutoff=1.5, order_type=1), dict(cutoff=2.5, order_type=2), dict(cutoff=1.2, order_type=0, max_smooth_iter=3),
                                  dict(cutoff=1.5, order_type=1, unwrap=False), dict(cutoff=1.5, order_type=2, global_normal=(0.0, 0.0, 1.0))])
def test_chained_frame_equals_the_stages(eng, opts):
    xyz, box, fused, staged = pair(eng, 150, 30000, **opts)
    assert fused.fusable() and not staged.fusable()
    for f, frame in enumerate(frames_of(xyz, 3)):
        a, b = frame.copy(), frame.copy()
        got, want = fused.compute(a, box), staged.compute(b, box)
        same_bits(a, b, "unwrapped frame")
        same_result(got, want, f"frame {f}: ")
        assert np.array_equal(fused.valid, staged.valid)
    assert np.count_nonzero(fused.valid) > (250 if opts.get("unwrap", True) and opts.get("max_smooth_iter", 1) == 1 else 10)


def test_chained_frame_on_resident_coordinates(eng):
    import torch
    xyz, box, fused, staged = pair(eng, 120, 20000, cutoff=1.5, order_type=1)
    for frame in frames_of(xyz, 2):
        d = torch.from_numpy(frame).cuda()
        b = frame.copy()
        got, want = fused.compute(d, box), staged.compute(b, box)
        same_bits(d.cpu().numpy(), b, "unwrapped frame")
        same_result(got, want)


def test_invalid_lipids_stay_out_and_flags_carry_over(eng):
    """LipidMolecule::valid is sticky (lib.rs:269-273): lipids switched off by the caller never enter a patch, lipids the
    smoothing drops in one frame are out in the next; reset_valid_lipids brings all of them back."""
    xyz, box, fused, staged = pair(eng, 150, 30000, cutoff=1.5, order_type=1)
    off = np.array([0, 7, 8, 149, 150, 222, 299])
    for m in (fused, staged):
        m.valid[off] = 0
    fr = frames_of(xyz, 3)
    # frame 1 carries a defect: one lipid's head far from the leaflet, so that its neighbours' fits move > 0.5 nm or fail
    bad = fr[1].copy()
    bad[40 * 52: 40 * 52 + 12, 2] += 1.5
    fr[1] = bad
    for f, frame in enumerate(fr):
        got, want = fused.compute(frame.copy(), box), staged.compute(frame.copy(), box)
        same_result(got, want, f"frame {f}: ")
        assert not got["valid"][off].any()
        po = got["patch_off"]
        assert all(po[k + 1] == po[k] for k in off)
        assert not np.isin(got["patch_ids"], off).any()
    assert np.array_equal(fused.valid, staged.valid)
    dropped = np.flatnonzero(fused.valid == 0)
    assert len(dropped) > len(off), "the defect was meant to cost at least one more lipid"
    for m in (fused, staged):
        m.reset_valid_lipids()
    got, want = fused.compute(fr[0].copy(), box), staged.compute(fr[0].copy(), box)
    same_result(got, want, "after reset: ")
    po = got["patch_off"]
    assert all(po[k + 1] > po[k] for k in off) and np.isin(off, got["patch_ids"]).all()     # back in the patches


def test_two_frames_in_flight(eng):
    """begin(k+1) before end(k): same results as one frame at a time, flags chained on the device."""
    xyz, box, fused, staged = pair(eng, 150, 30000, cutoff=1.5, order_type=1)
    fr = frames_of(xyz, 6)
    bad = fr[2].copy()
    bad[77 * 52: 77 * 52 + 12, 2] += 1.5          # costs lipids in frame 2: frames 3.. must see them gone
    fr[2] = bad
    want = [staged.compute(f.copy(), box) for f in fr]
    bufs = [f.copy() for f in fr]
    got = []
    t_prev = fused.compute_begin(bufs[0], box)
    for k in range(1, len(fr)):
        t = fused.compute_begin(bufs[k], box)
        got.append(fused.compute_end(t_prev))
        t_prev = t
    got.append(fused.compute_end(t_prev))
    for k, (g, w) in enumerate(zip(got, want)):
        same_result(g, w, f"frame {k}: ")
    assert np.array_equal(fused.valid, staged.valid) and np.count_nonzero(fused.valid == 0) > 0


def test_end_with_its_arrays_equals_end_then_fetch(eng):
    """molar_hip_membrane_frame_end_fetch: the per-lipid arrays that come with the end of a frame (stored behind its last
    kernel, one wait) are the arrays a fetch brings afterwards, with two frames in flight and flags that change on the way;
    arrays sized by the patch entries are refused there."""
    from molar_amd import api, membrane as mb
    xyz, box, first, tpl, masses = mb.build_bilayer(150, 30000)
    fr = frames_of(xyz, 5, seed=9)
    fr[1][31 * 52: 31 * 52 + 12, 2] += 1.5
    names = ["head", "mid", "tail", "patch_offsets", "initial_normals", "valid", "smoothed_head", "normals", "quad_coefs", "mean_curv",
             "gauss_curv", "princ_curvs", "princ_dirs", "area", "nvert", "order"]
    res = []
    for with_end in (False, True):
        m = mb.Membrane(eng, len(xyz), first, tpl, masses, mb.MembraneOptions(cutoff=1.5, order_type=2))
        plan = m._plan()
        plan.set_valid(None)
        bufs = [f.copy() for f in fr]
        out, prev = [], None
        for k in range(len(fr) + 1):
            t = plan.begin(bufs[k], box) if k < len(fr) else None
            if prev is not None:
                if with_end:
                    v, r = plan.end(prev, names)
                    assert v.nlipids == m.K
                    r["neib_ids"] = plan.fetch(prev, ["neib_ids"])["neib_ids"]       # a later fetch of the same frame still works
                else:
                    plan.end(prev)
                    r = plan.fetch(prev, names + ["neib_ids"])
                out.append(r)
            prev = t
        if with_end:
            t = plan.begin(bufs[0].copy(), box)
            with pytest.raises(ValueError):
                plan.end(t, ["valid", "patch_ids"])
            o = api.MembraneOut()
            ids = np.zeros(8, np.uint64)
            o.patch_ids = ids.ctypes.data
            assert plan.lib.molar_hip_membrane_frame_end_fetch(plan.handle, t, None, __import__("ctypes").byref(o)) == 50        # MOLAR_HIP_ERR_INVALID_ARGUMENT
            plan.end(t)
        plan.close()
        res.append(out)
    assert len(res[0]) == len(fr)
    for k, (a, b) in enumerate(zip(*res)):
        assert a["valid"].sum() > 0
        for n in names + ["neib_ids"]:
            same_bits(b[n], a[n], f"frame {k} {n}")
    assert any((r["valid"] == 0).any() for r in res[1])


def test_a_frame_that_outgrows_its_buffers_is_repeated(eng):
    """Pair and patch capacities come from earlier frames.  A frame with several times as many neighbours (the box and the
    bilayer squeezed laterally) overflows them with a younger frame already enqueued behind it: both are repeated
    inside end() and come out right."""
    from molar_amd import membrane as mb
    xyz, box, first, tpl, masses = mb.build_bilayer(150, 30000)
    fused = mb.Membrane(eng, len(xyz), first, tpl, masses, mb.MembraneOptions(cutoff=1.5, order_type=1))
    staged = mb.Membrane(eng, len(xyz), first, tpl, masses, mb.MembraneOptions(cutoff=1.5, order_type=1, fused=False))
    squeeze = np.array([0.5, 0.5, 1.0], np.float32)
    fr = frames_of(xyz, 4)
    boxes = [box, box, (box * squeeze[:, None]).astype(np.float32), box]
    fr[2] = (fr[2] * squeeze).astype(np.float32)
    want = [staged.compute(f.copy(), b) for f, b in zip(fr, boxes)]
    n_pairs = [len(w["patch_ids"]) for w in want]
    assert n_pairs[2] > 2.5 * n_pairs[1]
    bufs = [f.copy() for f in fr]
    got = []
    t_prev = fused.compute_begin(bufs[0], boxes[0])
    for k in range(1, 4):
        t = fused.compute_begin(bufs[k], boxes[k])
        got.append(fused.compute_end(t_prev))
        t_prev = t
    got.append(fused.compute_end(t_prev))
    for k, (g, w) in enumerate(zip(got, want)):
        same_result(g, w, f"frame {k}: ")


@pytest.mark.timeout(900)
def test_chained_frame_with_ten_thousand_lipids(eng):
    xyz, box, fused, staged = pair(eng, 5300, 560_000, cutoff=1.5, order_type=1)
    a, b = xyz.copy(), xyz.copy()
    got, want = fused.compute(a, box), staged.compute(b, box)
    same_bits(a, b, "unwrapped frame")
    same_result(got, want)
    assert np.count_nonzero(got["valid"]) > 10000


@pytest.mark.parametrize("seed", range(8))
def test_randomised_bilayers_sheared_boxes_pipelined(eng, seed):
    """Random size, cutoff, order type, iterations, switched-off lipids and a sheared (triclinic) box; three frames with two
    in flight against the stage-by-stage path."""
    from molar_amd import membrane as mb
    rng = np.random.default_rng(1000 + seed)
    per = int(rng.integers(30, 320))
    natoms = 2 * per * 52 + int(rng.integers(0, 20000))
    xyz, box, first, tpl, masses = mb.build_bilayer(per, natoms, seed=int(rng.integers(1 << 30)))
    shear = np.eye(3, dtype=np.float64)
    if seed % 2:
        shear[0, 1], shear[0, 2], shear[1, 2] = rng.uniform(-0.4, 0.4, 3)
    xyz = (xyz.astype(np.float64) @ shear.T).astype(np.float32)
    box = (shear @ box.astype(np.float64)).astype(np.float32)
    opts = dict(cutoff=float(rng.uniform(1.0, 2.8)), order_type=int(rng.integers(0, 3)), max_smooth_iter=int(rng.integers(1, 3)),
                unwrap=True)
    if seed % 3 == 0:
        opts["global_normal"] = (0.0, 0.0, 1.0)
    fused = mb.Membrane(eng, len(xyz), first, tpl, masses, mb.MembraneOptions(**opts))
    staged = mb.Membrane(eng, len(xyz), first, tpl, masses, mb.MembraneOptions(fused=False, **opts))
    off = rng.choice(2 * per, size=int(rng.integers(0, 6)), replace=False)
    for m in (fused, staged):
        m.valid[off] = 0
    fr = frames_of(xyz, 3, seed=seed)
    want = [staged.compute(f.copy(), box) for f in fr]
    bufs = [f.copy() for f in fr]
    got = []
    t_prev = fused.compute_begin(bufs[0], box)
    for k in range(1, 3):
        t = fused.compute_begin(bufs[k], box)
        got.append(fused.compute_end(t_prev))
        t_prev = t
    got.append(fused.compute_end(t_prev))
    for k, (g, w) in enumerate(zip(got, want)):
        same_result(g, w, f"seed {seed} frame {k}: ")
    assert np.array_equal(fused.valid, staged.valid)


@pytest.mark.parametrize("per", [1, 2, 5])
def test_degenerate_bilayers(eng, per):
    """One, two, five lipids per leaflet (no patch at all, or patches too short for the quadric fit): every lipid drops out,
    the same way on both paths; and a bilayer whose lipids are all switched off from the start."""
    from molar_amd import membrane as mb
    xyz, box, first, tpl, masses = mb.build_bilayer(per, 2 * per * 52 + 500)
    big = (box * np.float32(3.0)).astype(np.float32)            # room for the cutoff: the search needs a box of >= 2 cutoffs
    fused = mb.Membrane(eng, len(xyz), first, tpl, masses, mb.MembraneOptions(cutoff=1.5, order_type=1))
    staged = mb.Membrane(eng, len(xyz), first, tpl, masses, mb.MembraneOptions(cutoff=1.5, order_type=1, fused=False))
    for f in frames_of(xyz, 2):
        got, want = fused.compute(f.copy(), big), staged.compute(f.copy(), big)
        for k in ARRAYS:
            a, b = np.ascontiguousarray(got[k]), np.ascontiguousarray(want[k])
            if k == "fitted_patch_points":
                a, b = a[: len(want["patch_ids"])], b[: len(want["patch_ids"])]
            assert a.tobytes() == b.tobytes(), k
    assert np.array_equal(fused.valid, staged.valid)
    for m in (fused, staged):
        m.valid[:] = 0
    got, want = fused.compute(xyz.copy(), big), staged.compute(xyz.copy(), big)
    assert len(got["patch_ids"]) == 0 and not got["valid"].any() and np.array_equal(got["normals"], want["normals"])


def test_plan_argument_errors(eng):
    from molar_amd import api, membrane as mb
    xyz, box, first, tpl, masses = mb.build_bilayer(20, 3000)
    m = mb.Membrane(eng, len(xyz), first, tpl, masses, mb.MembraneOptions(cutoff=1.5, order_type=1))
    plan = m._plan()
    with pytest.raises(api.MolarHipError):
        plan.end(0)                                    # nothing in flight
    t0 = plan.begin(xyz.copy(), box)
    t1 = plan.begin(xyz.copy(), box)
    with pytest.raises(api.MolarHipError):
        plan.begin(xyz.copy(), box)                    # both tickets taken
    with pytest.raises(api.MolarHipError):
        plan.end(t1)                                   # the older frame ends first
    with pytest.raises(api.MolarHipError):
        plan.set_valid(None)                           # frames in flight
    plan.end(t0); plan.end(t1)
    with pytest.raises(api.MolarHipError):
        plan.begin(xyz.copy(), np.zeros((3, 3), np.float32))     # PeriodicBox::from_matrix refuses it
    bad_first = first.copy()
    with pytest.raises(api.MolarHipError):
        api.MembranePlan(eng, 10, m.lipid_idx, m.lipid_off, m.marker_idx, m.marker_off, masses[:10], m.tail_idx, m.tail_off,
                         np.repeat(np.arange(m.K, dtype=np.uint32), m.ntails), m.tail_bonds, 1.5, 1)     # indices beyond natoms


def test_regular_marker_search_after_a_small_cell_one_on_the_same_context():
    """A bilayer whose marker search runs the small-cell kernels (no hit history: none is allocated) followed, on the SAME engine
    context, by a smaller bilayer with a larger cutoff whose marker search runs the regular kernels: the first attempt of that
    search finds no room for its hit history, its fill pass leaves the wrapped entries' results unwritten - stale pairs of the
    larger system - and the patch kernels chained behind it must not touch them (k_patch_begin: the list counts as not there;
    round 6 met a memory fault here).  The frame is repeated with the history grown and must equal a fresh context's."""
    from molar_amd import api, build
    from molar_amd import membrane as mb
    build.build_library()
    e_shared, e_fresh = api.Engine(0), api.Engine(0)
    big = mb.build_bilayer(800, 120_000, seed=101)
    small = mb.build_bilayer(450, 60_000, seed=103)
    m_big = mb.Membrane(e_shared, len(big[0]), big[2], big[3], big[4], mb.MembraneOptions(cutoff=2.5, order_type=1))
    r0 = m_big.compute(big[0].copy(), big[1])
    assert int(r0["valid"].sum()) > 1000
    opt = mb.MembraneOptions(cutoff=3.5, order_type=1, max_smooth_iter=2)
    got = mb.Membrane(e_shared, len(small[0]), small[2], small[3], small[4], opt).compute(small[0].copy(), small[1])
    want = mb.Membrane(e_fresh, len(small[0]), small[2], small[3], small[4], opt).compute(small[0].copy(), small[1])
    assert int(want["valid"].sum()) > 500
    for k, v in want.items():
        if isinstance(v, np.ndarray):
            assert np.array_equal(got[k], v), k
