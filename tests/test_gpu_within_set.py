"""GPU parity of `within` as a set (molar_hip_within_count / _fill) against the oracle.

The reference's callers keep SortedSet::from_unsorted(stream) (selection/ast.rs:589-631, selection_expr.rs:112) of what
distance_search_within(_pbc) emits (distance_search.rs:271-322,519-598).  The set form must equal np.unique of the
oracle's stream - including on sheared boxes, where the reference's half-shell plan does not see every neighbour and
the set is NOT the brute-force set - and np.unique of the engine's own stream form.  Ids: bit-exact."""
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


def stream_unique(eng, a, cutoff, pos, idx1, pos2, idx2, **kw):
    n = eng.search_count(a.SEARCH_WITHIN, cutoff, pos, idx1, pos2, idx2, **kw)
    return np.unique(eng.search_fill_ids(n))


@pytest.mark.parametrize("boxfn,n,cutoff,pbc", [
    (synth.box_ortho, 6000, 0.45, 7),
    (synth.box_a, 8000, 0.6, 7),
    (synth.box_a, 30000, 0.9, 7),
    (synth.box_b, 6000, 0.5, 7),          # sheared: the reference's plan is incomplete; parity with it, not with brute force
    (synth.box_ortho, 4000, 0.5, 3),      # z non-periodic: drop rule
    (synth.box_ortho, 4000, 0.5, 5),
    (synth.box_a, 4000, 0.5, 1),
    (synth.box_a, 5000, 2.5, 7),          # few, crowded cells: several row blocks per cell, partners repeat
])
def test_within_set_pbc(eng, orc32, boxfn, n, cutoff, pbc):
    import molar_amd.api as a
    box = boxfn(n)
    pos = synth.frame(n, box)
    rng = np.random.default_rng(n + pbc)
    ob = orc32.box_from_matrix(box)
    for idx1, idx2 in [
        (np.arange(n, dtype=np.uint64), np.arange(100, 400, dtype=np.uint64)),                      # a blob of consecutive atoms
        (np.arange(0, n, 3, dtype=np.uint64), np.sort(rng.choice(n, n // 10, replace=False)).astype(np.uint64)),   # scattered
        (np.arange(n // 2, n, dtype=np.uint64), np.arange(0, n // 2, 7, dtype=np.uint64)),
    ]:
        ref = orc32.search_within_pbc(cutoff, pos[idx1.astype(int)], pos[idx2.astype(int)], ob, pbc, idx1, idx2, nthreads=4)
        want = np.unique(ref["i"])
        got = eng.within_set(cutoff, pos, idx1, pos, idx2, box=box, pbc=pbc)
        assert got.dtype == np.uint64 and np.array_equal(got, want), (len(got), len(want))
        assert np.array_equal(got, stream_unique(eng, a, cutoff, pos, idx1, pos, idx2, box=box, pbc=pbc))
        # local ids
        got_l = eng.within_set(cutoff, pos, idx1, pos, idx2, box=box, pbc=pbc, ids_local=True)
        assert np.array_equal(idx1[got_l.astype(int)], want)


def test_within_set_nonperiodic(eng, orc32):
    import molar_amd.api as a
    n = 9000
    box = synth.box_a(n)
    pos = synth.frame(n, box)
    idx1 = np.arange(n, dtype=np.uint64)
    idx2 = np.arange(2000, 2300, dtype=np.uint64)
    lo, up = orc32.min_max(pos)                                    # selection/ast.rs:597-602
    lo = lo + (np.float32(-0.6) - np.float32(1.1920929e-07))
    up = up + (np.float32(0.6) + np.float32(1.1920929e-07))
    ref = orc32.search_within(0.6, pos, pos[2000:2300], lo, up, idx1, idx2, nthreads=4)
    got = eng.within_set(0.6, pos, idx1, pos, idx2, lower=lo, upper=up)
    assert np.array_equal(got, np.unique(ref["i"]))
    bf = orc32.brute_double(0.6, pos, pos[2000:2300], None, ids1=idx1, ids2=idx2)
    assert np.array_equal(got, np.unique(bf["i"]))


def test_within_set_edge_cases(eng, orc32):
    n = 3000
    box = synth.box_ortho(n)
    pos = synth.frame(n, box)
    idx1 = np.arange(n, dtype=np.uint64)
    # (nearly) nobody in range: a cutoff far below the mean spacing
    got = eng.within_set(0.03, pos, idx1[:1500], pos, idx1[1500:], box=box, pbc=7)
    ob = orc32.box_from_matrix(box)
    ref = orc32.search_within_pbc(0.03, pos[:1500], pos[1500:], ob, 7, idx1[:1500], idx1[1500:])
    assert np.array_equal(got, np.unique(ref["i"]))
    # a selection "within" itself: every atom is within 0 of itself
    got = eng.within_set(0.3, pos, idx1, pos, idx1, box=box, pbc=7)
    assert np.array_equal(got, idx1)
    # one inner atom (WithinPoint's shape, ast.rs:633-665)
    got = eng.within_set(0.8, pos, idx1, pos, idx1[7:8], box=box, pbc=7)
    ref = orc32.search_within_pbc(0.8, pos, pos[7:8], ob, 7, idx1, idx1[7:8])
    assert np.array_equal(got, np.unique(ref["i"])) and len(got) > 1
    # wrong kind is refused
    import molar_amd.api as a
    from molar_amd.api import MolarHipError
    d, keep = eng._search_desc(a.SEARCH_DOUBLE, 0.5, pos, idx1, pos, idx1, box=box, pbc=7)
    import ctypes as C
    cnt = C.c_uint64(0)
    assert eng.lib.molar_hip_within_count(eng.ctx, C.byref(d), C.byref(cnt)) != 0


def test_within_set_device_output_and_selection(eng, orc32):
    import torch
    import molar_amd.api as a
    n = 20000
    box = synth.box_a(n)
    pos = synth.frame(n, box)
    ob = orc32.box_from_matrix(box)
    idx1 = np.arange(n, dtype=np.uint64)
    idx2 = np.arange(0, n, 10, dtype=np.uint64)
    ref = orc32.search_within_pbc(0.5, pos, pos[::10], ob, 7, idx1, idx2, nthreads=4)
    dpos = torch.from_numpy(pos).cuda()
    torch.cuda.synchronize()
    out = eng.within_set(0.5, dpos, idx1, dpos, idx2, box=box, pbc=7,
                         device_out=lambda k: torch.empty(k, dtype=torch.int64, device="cuda"))
    eng.synchronize()
    assert np.array_equal(out.cpu().numpy().astype(np.uint64), np.unique(ref["i"]))


@pytest.mark.parametrize("boxfn,n,cutoff,pbc,n2", [
    (synth.box_a, 40000, 0.7, 7, 6000),        # second sets above the small-set path's limit: the partner-list path
    (synth.box_b, 30000, 0.6, 7, 9000),
    (synth.box_ortho, 30000, 0.8, 5, 5000),
])
def test_within_set_large_second_sets(eng, orc32, boxfn, n, cutoff, pbc, n2):
    import molar_amd.api as a
    box = boxfn(n)
    pos = synth.frame(n, box)
    rng = np.random.default_rng(n2)
    ob = orc32.box_from_matrix(box)
    idx1 = np.arange(n, dtype=np.uint64)
    idx2 = np.sort(rng.choice(n, n2, replace=False)).astype(np.uint64)
    want = np.unique(orc32.search_within_pbc(cutoff, pos, pos[idx2.astype(int)], ob, pbc, idx1, idx2, nthreads=8)["i"])
    got = eng.within_set(cutoff, pos, idx1, pos, idx2, box=box, pbc=pbc)
    assert np.array_equal(got, want)
    # a small second set right after (flags left by the partner-list path are cleared), then the large one again
    small = idx2[:40]
    want_s = np.unique(orc32.search_within_pbc(cutoff, pos, pos[small.astype(int)], ob, pbc, idx1, small, nthreads=4)["i"])
    assert np.array_equal(eng.within_set(cutoff, pos, idx1, pos, small, box=box, pbc=pbc), want_s)
    assert np.array_equal(eng.within_set(cutoff, pos, idx1, pos, idx2, box=box, pbc=pbc), want)


def test_within_hold_reuses_the_first_sets_grid(eng, orc32):
    """molar_hip_within_hold: consecutive requests against one frame (within_size_bench.rs:13-47: groups of residues, a cutoff
    sweep) - every answer equals the oracle's whether the grid was rebuilt or reused; a new cutoff (another grid), another
    first set, a device tensor and a plain search in between all end the reuse without a wrong answer; after the hold is
    switched off, changed coordinates are seen."""
    import torch
    import molar_amd.api as a
    n = 50000
    box = synth.box_a(n)
    pos = synth.frame(n, box, 3)
    ob = orc32.box_from_matrix(box)
    idx1 = np.arange(n, dtype=np.uint64)
    order = np.argsort(((pos - (box @ np.array([0.5, 0.5, 0.5], np.float32))) ** 2).sum(1))

    def want(cutoff, p, i1, i2):
        return np.unique(orc32.search_within_pbc(cutoff, p[i1.astype(int)], p[i2.astype(int)], ob, 7, i1, i2, nthreads=4)["i"])

    e2 = a.Engine(0)
    e2.within_hold(True)
    dpos = torch.from_numpy(pos).cuda()
    torch.cuda.synchronize()
    for src in (pos, dpos):
        for cutoff in (0.4, 0.4, 0.9, 0.4):
            for nres in (1, 20, 1, 60):
                grp = np.sort(order[:10 * (nres + 1)]).astype(np.uint64)
                assert np.array_equal(e2.within_set(cutoff, src, idx1, src, grp, box=box, pbc=7), want(cutoff, pos, idx1, grp))
        # another first set, then a plain search on the same context, then the first request again
        half = idx1[::2].copy()
        grp = np.sort(order[:50]).astype(np.uint64)
        assert np.array_equal(e2.within_set(0.4, src, half, src, grp, box=box, pbc=7), want(0.4, pos, half, grp))
        cnt = e2.search_count(a.SEARCH_SINGLE, 0.3, src, idx1[:5000], box=box, pbc=7)
        assert cnt == len(orc32.search_single_pbc(0.3, pos[:5000], ob, 7, nthreads=4)["i"])
        assert np.array_equal(e2.within_set(0.4, src, idx1, src, grp, box=box, pbc=7), want(0.4, pos, idx1, grp))
    # hold off: the same array with other coordinates is staged again
    e2.within_hold(False)
    pos2 = pos.copy()
    assert np.array_equal(e2.within_set(0.4, pos2, idx1, pos2, grp, box=box, pbc=7), want(0.4, pos2, idx1, grp))
    pos2[:] = synth.frame(n, box, 4)
    assert np.array_equal(e2.within_set(0.4, pos2, idx1, pos2, grp, box=box, pbc=7), want(0.4, pos2, idx1, grp))


def test_within_hold_sees_a_host_frame_updated_in_place(eng, orc32):
    """With the hold left on across a trajectory loop, a host array that is overwritten with the next frame (same address,
    same sizes) must not be answered from the staged copy of the frame before: the hold keeps a fingerprint of a host-memory
    first set and stages it again when it differs."""
    import molar_amd.api as a
    n = 30000
    box = synth.box_a(n)
    ob = orc32.box_from_matrix(box)
    idx1 = np.arange(n, dtype=np.uint64)
    grp = np.arange(200, 260, dtype=np.uint64)
    e2 = a.Engine(0)
    e2.within_hold(True)
    buf = np.empty((n, 3), np.float32)
    for f in (3, 4, 4, 5):
        buf[:] = synth.frame(n, box, f)
        want = np.unique(orc32.search_within_pbc(0.5, buf, buf[grp.astype(int)], ob, 7, idx1, grp, nthreads=4)["i"])
        assert np.array_equal(e2.within_set(0.5, buf, idx1, buf, grp, box=box, pbc=7), want)
