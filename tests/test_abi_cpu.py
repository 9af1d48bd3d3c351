"""CPU-only checks of the C-ABI library: it loads, exports every symbol include/molar_hip.h
declares, its host-side PeriodicBox arithmetic equals the oracle bit for bit, and the product
path fails loudly when no GPU is present (no compute calls here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from molar_amd import _lib, build
    build.build_library()
    return _lib.load()


def header_functions():
    text = open(os.path.join(ROOT, "include", "molar_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(molar_hip_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from molar_amd import _lib
    names = header_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/molar_hip.h but not exported"
    assert set(names) == set(_lib.SYMBOLS), set(names) ^ set(_lib.SYMBOLS)


def test_version_and_device_count(lib):
    assert b"gfx950" in lib.molar_hip_version()
    assert lib.molar_hip_device_count() >= 0


def test_box_matches_oracle_bitwise(lib, orc32):
    from molar_amd.api import PeriodicBox
    rng = np.random.default_rng(0)
    mats = [
        np.diag([10.0, 20.0, 30.0]),
        np.array([[10.0, 4.0, -4.0], [0, 10.0, 0], [0, 0, 10.0]]),
        np.array([[21.544, 0, -3.0], [0, 21.544, -3.0], [0, 0, 21.544]]),
        np.array([[6.0, 0, 3.0], [0, 6.0, 3.0], [0, 0, 6.0]]),
        np.array([[8.0, 4.0, 0], [0, 6.9282, 0], [0, 0, 9.0]]),
    ]
    for m in mats:
        pb = PeriodicBox.from_matrix(m)
        ob = orc32.box_from_matrix(m)
        assert np.array_equal(np.array(pb._b.m), np.array(ob.m))
        assert np.array_equal(np.array(pb._b.inv), np.array(ob.inv))
        assert pb._b.nshift == ob.nshift
        assert np.array_equal(np.array(pb._b.shifts)[: 3 * ob.nshift], np.array(ob.shifts)[: 3 * ob.nshift])
        assert np.array_equal(pb.get_lab_extents(), orc32.lab_extents(ob))
        assert np.array_equal(pb.get_box_extents(), orc32.box_extents(ob))
        for _ in range(50):
            q = rng.uniform(-30, 30, 3).astype(np.float32)
            assert np.array_equal(pb.wrap_point(q), orc32.wrap_point(ob, q))
            assert pb.is_inside(q) == orc32.is_inside(ob, q)
        for _ in range(200):
            v = rng.uniform(-60, 60, 3).astype(np.float32)
            for dims in (7, 0, 1, 3, 5):
                assert np.array_equal(pb.shortest_vector(v, dims), orc32.shortest_vector_dims(ob, v, dims))


def test_box_known_answers(lib):
    """periodic_box.rs:559-575 and test_2.py:233-245 through the product's host arithmetic."""
    from molar_amd.api import PeriodicBox
    pb = PeriodicBox.from_matrix([[10.0, 4.0, -4.0], [0, 10.0, 0], [0, 0, 10.0]])
    assert abs(pb.distance([38.9214, 40.0078, -34.0795], [-26.6187, 40.8926, 30.9709]) - 5.353627) < 1e-3
    b = PeriodicBox([1, 2, 3], [90, 90, 90])
    v = b.shortest_vector([0.9, 0.5, 0.6])
    assert np.allclose(v, [-0.1, 0.5, 0.6], atol=1e-6)
    assert PeriodicBox.from_matrix(np.diag([10.0, 20.0, 30.0])).n_tric_corrections == 0


def test_box_errors(lib):
    from molar_amd.api import PeriodicBox
    from molar_amd._lib import MolarHipError
    with pytest.raises(MolarHipError) as e:
        PeriodicBox.from_matrix(np.zeros((3, 3)))
    assert e.value.code == 5
    with pytest.raises(MolarHipError) as e:
        PeriodicBox.from_matrix([[1, 1, 0], [1, 1, 0], [0, 0, 1]])
    assert e.value.code == 6
    with pytest.raises(MolarHipError) as e:
        PeriodicBox([10.0, 0.2, 15.0], [90.0, 9.0, 90.0])       # periodic_box.rs:448-454
    assert e.value.code == 10


@pytest.mark.parametrize("hmin,hmax,nbins", [(0.0, 1.2, 1200), (0.35, 0.9, 450), (-0.2, 0.7, 350), (0.0, 0.6, 1), (0.1, 2.5, 8192)])
def test_histogram_edges_are_the_formula(lib, orc32, hmin, hmax, nbins):
    """The table the fused histogram bins with (molar_hip_histogram_edges) against Histogram1D::add_one
    (stats.rs:29-35): at every edge the distance d = sqrt(edge) falls into a bin >= b, at the float just below it into a
    bin < b; random squared distances bin identically through the table and through the oracle."""
    from molar_amd import api
    e = api.histogram_edges(hmin, hmax, nbins)
    assert e.shape == (nbins + 1,) and np.all(np.diff(e) >= 0) and e[0] >= 0

    def formula(d2):         # numpy float32 arithmetic is IEEE like the oracle's C (cross-checked against it below)
        d = np.sqrt(np.asarray(d2, np.float32)).astype(np.float32)
        return np.floor(np.float32(nbins) * (d - np.float32(hmin)) / (np.float32(hmax) - np.float32(hmin)))
    rng = np.random.default_rng(3)
    x = (rng.uniform(0, 1.1 * max(hmax, 0.1), 20000).astype(np.float32)) ** 2
    want = orc32.histogram_add(hmin, hmax, nbins, np.sqrt(x)).astype(np.int64)
    fb = formula(x)
    ok = (fb >= 0) & (fb < nbins)
    assert np.array_equal(np.bincount(fb[ok].astype(np.int64), minlength=nbins), want)
    tb = np.searchsorted(e, x, side="right") - 1          # largest b with e[b] <= x
    assert np.array_equal(tb[ok], fb[ok].astype(np.int64))
    assert np.all((tb[~ok] < 0) | (tb[~ok] >= nbins))
    # every edge is tight
    assert np.all(formula(e) >= np.arange(nbins + 1))
    pos = e > 0
    below = np.nextafter(e[pos], np.float32(-1.0), dtype=np.float32)
    assert np.all(formula(below) < np.arange(nbins + 1)[pos])


def test_histogram_edges_rejects_degenerate_ranges(lib):
    from molar_amd import api
    from molar_amd._lib import MolarHipError
    for lo, hi in ((1.0, 1.0), (2.0, 1.0), (0.0, float("inf")), (float("nan"), 1.0)):
        with pytest.raises(MolarHipError):
            api.histogram_edges(lo, hi, 10)


def test_isa_audit_flags_a_copy_ahead_of_an_exec_restore(tmp_path):
    """molar_amd.build.exec_restore_hazards on the pattern ROCm 7.2's compiler produced in hist_kernel (a live-range
    split copy of a VGPR in a join block before `s_or_b64 exec, exec`), and on the harmless form."""
    from molar_amd.build import exec_restore_hazards
    bad = tmp_path / "bad.s"
    bad.write_text("_Z4kernv:\n\tv_add_f32_e32 v59, -1.0, v1\n.LBB2_1115:\n\ts_mov_b64 s[10:11], 0\n\tv_writelane_b32 v63, s10, 11\n"
                   "\tv_mov_b32_e32 v62, v59\n\ts_nop 0\n\ts_or_b64 exec, exec, s[12:13]\n.LBB2_1261:\n\tv_mov_b32_e32 v59, v62\n\ts_endpgm\n")
    good = tmp_path / "good.s"
    good.write_text("_Z4kernv:\n.LBB2_1:\n\ts_or_b64 exec, exec, s[12:13]\n\tv_mov_b32_e32 v62, v59\n.LBB2_2:\n\tv_mov_b64_e32 v[0:1], v[50:51]\n"
                    "\ts_or_b64 exec, exec, s[2:3]\n\ts_endpgm\n")      # a conditional assignment at the end of a then-block
    hits = exec_restore_hazards(str(bad))
    assert len(hits) == 1 and "v_mov_b32_e32 v62, v59" in hits[0] and "_Z4kernv" in hits[0]
    assert exec_restore_hazards(str(good)) == []
    # a copy at the top of a JOIN block (the label the skip branch s_cbranch_execz jumps to) is flagged even when it is
    # never undone; the same copy in a then-body entered by s_cbranch_execnz is not
    join = tmp_path / "join.s"
    join.write_text("_Z5kern2v:\n\ts_and_saveexec_b64 s[2:3], vcc\n\ts_cbranch_execz .LBB0_2\n\tv_add_f32_e32 v1, v1, v1\n.LBB0_2:\n"
                    "\tv_mov_b32_e32 v7, v9\n\ts_or_b64 exec, exec, s[2:3]\n\ts_endpgm\n")
    then = tmp_path / "then.s"
    then.write_text("_Z5kern3v:\n\ts_and_saveexec_b64 s[2:3], vcc\n\ts_cbranch_execnz .LBB0_2\n\ts_branch .LBB0_3\n.LBB0_2:\n"
                    "\tv_mov_b32_e32 v7, v9\n\ts_or_b64 exec, exec, s[2:3]\n.LBB0_3:\n\ts_endpgm\n")
    assert len(exec_restore_hazards(str(join))) == 1 and exec_restore_hazards(str(then)) == []


def test_fails_loudly_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from molar_amd.api import Engine
    from molar_amd._lib import MolarHipError
    with pytest.raises(MolarHipError):
        Engine(0)
    assert "HIP device" in lib.molar_hip_last_error().decode() or "no HIP" in lib.molar_hip_last_error().decode()


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under molar_amd/ may reference it."""
    pkg = os.path.join(ROOT, "molar_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("no CPU fallback", ""), f"{f} mentions the oracle"
