"""The segmented layout of the resident searches (molar_hip_search_resident_layout): one segment of the result planes per
element of the reference's search plan (distance_search.rs:103-269), capacities from the search before, no count pass.
The segments read one after the other must be the dense list - the reference's output (distance_search.rs:432-517,
324-373), which the dense layout of the same context is checked against the oracle for in test_gpu_search.py - bit for
bit, distances included: on the first search of a plan (kernel run twice), on later ones (capacities from the frame before),
when the planes are too short, and when an entry outgrows its segment (the frame is repeated with its own counts)."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

from molar_amd import synth

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture()
def eng():
    from molar_amd import build
    from molar_amd.api import Engine
    build.build_library()
    return Engine(0)


def _dense(eng, desc):
    """(pairs (n, 2) uint32, dist (n,) float32) of the dense layout, on the host"""
    import torch
    from molar_amd import api as a
    eng.search_resident_layout(False)
    cnt, p, d = eng.search_resident_desc(desc)
    if cnt == 0:
        return np.zeros((0, 2), np.uint32), np.zeros(0, np.float32)
    pairs = a.device_view(p, (cnt, 2), torch.int32).cpu().numpy().view(np.uint32).copy()
    dist = a.device_view(d, (cnt,), torch.float32).cpu().numpy().copy()
    return pairs, dist


def _from_segments(eng, result_set, cnt, paddr, daddr):
    """the segments read one after the other, gathered on the host from the raw planes"""
    import torch
    from molar_amd import api as a
    b, n, nseg, span = eng.search_segments(result_set)
    if nseg == 0 or cnt == 0:
        return np.zeros((0, 2), np.uint32), np.zeros(0, np.float32), (b, n, nseg, span)
    base = a.device_view(b, (nseg + 1,), torch.int64).cpu().numpy()
    num = a.device_view(n, (nseg,), torch.int32).cpu().numpy().view(np.uint32).astype(np.int64)
    assert int(num.sum()) == cnt
    assert int(base[nseg]) == span
    assert np.all(base[:-1] % 64 == 0), "segment starts are 256-byte aligned"
    assert np.all(base[1:] - base[:-1] >= num), "every segment holds its results"
    planes_p = a.device_view(paddr, (span, 2), torch.int32).cpu().numpy().view(np.uint32)
    idx = np.concatenate([np.arange(s, s + k) for s, k in zip(base[:-1], num) if k]) if cnt else np.zeros(0, np.int64)
    pairs = planes_p[idx]
    dist = None
    if daddr:
        dist = a.device_view(daddr, (span,), torch.float32).cpu().numpy()[idx]
    return pairs, dist, (b, n, nseg, span)


def _compact(eng, result_set, cnt, with_dist=True):
    import torch
    p = torch.empty((max(cnt, 1), 2), dtype=torch.int32, device="cuda")
    d = torch.empty(max(cnt, 1), dtype=torch.float32, device="cuda") if with_dist else None
    eng.search_segments_compact(result_set, p.data_ptr(), d.data_ptr() if with_dist else None)
    return p[:cnt].cpu().numpy().view(np.uint32), (d[:cnt].cpu().numpy() if with_dist else None)


def _box(kind, n):
    if kind == "tric":
        return synth.box_a(n)
    e = (n / 100.0) ** (1.0 / 3.0)
    return np.diag([e, 1.1 * e, 0.9 * e]).astype(np.float32)


@pytest.mark.parametrize("shape,pbc", [("tric", 7), ("ortho", 7), ("tric", 3), ("ortho", 0), ("tric", 5)])
def test_single_segments_are_the_dense_list(eng, shape, pbc):
    import torch
    from molar_amd import api as a
    n, rc = 30_000, 0.9
    box = _box(shape, n)
    pos = torch.from_numpy(synth.frame(n, box, 0)).cuda()
    desc, keep = eng.make_search_desc(a.SEARCH_SINGLE, rc, pos, box=box, pbc=pbc)
    want_p, want_d = _dense(eng, desc)
    assert len(want_p) > 1000
    eng.search_resident_layout(True)
    for rep in range(3):          # first search of the plan (counts, then exact capacities), then capacities from the one before
        cnt, p, d = eng.search_resident_desc(desc)
        assert cnt == len(want_p)
        got_p, got_d, (_, _, nseg, span) = _from_segments(eng, 0, cnt, p, d)
        assert np.array_equal(got_p, want_p) and np.array_equal(got_d, want_d), rep
        cp, cd = _compact(eng, 0, cnt)
        assert np.array_equal(cp, want_p) and np.array_equal(cd, want_d), rep
    assert nseg == 14 * int(np.prod(_grid_dims(eng, desc)))
    eng.search_resident_layout(False)


def _grid_dims(eng, desc):
    cnt = C.c_uint64(0)
    from molar_amd._lib import check
    check(eng.lib.molar_hip_search_count(eng.ctx, C.byref(desc), C.byref(cnt)))
    return eng.grid_dims()


@pytest.mark.parametrize("pbc", [7, 0, 6])
def test_double_segments_are_the_dense_list(eng, pbc):
    import torch
    from molar_amd import api as a
    n = 24_000
    box = synth.box_a(n)
    xyz = torch.from_numpy(synth.frame(n, box, 1)).cuda()
    rng = np.random.default_rng(5)
    idx1 = np.sort(rng.choice(n, 9000, replace=False)).astype(np.uint64)
    idx2 = np.sort(rng.choice(n, 7000, replace=False)).astype(np.uint64)      # overlaps idx1: same-cell duplicates included
    desc, keep = eng.make_search_desc(a.SEARCH_DOUBLE, 1.0, xyz, idx1=idx1, xyz2=xyz, idx2=idx2, box=box, pbc=pbc)
    want_p, want_d = _dense(eng, desc)
    assert len(want_p) > 1000
    eng.search_resident_layout(True)
    for rep in range(2):
        cnt, p, d = eng.search_resident_desc(desc)
        assert cnt == len(want_p)
        got_p, got_d, _ = _from_segments(eng, 0, cnt, p, d)
        assert np.array_equal(got_p, want_p) and np.array_equal(got_d, want_d), rep
    eng.search_resident_layout(False)


def test_pipelined_frames_capacities_from_the_frame_before(eng):
    """begin / end with two searches in flight over a jittered trajectory: every frame's segments are that frame's dense list;
    the first frames grow the planes of both result sets."""
    import torch
    from molar_amd import api as a
    n, rc, nframes = 40_000, 1.0, 7
    box = synth.box_a(n)
    frames = [torch.from_numpy(synth.frame(n, box, f)).cuda() for f in range(nframes)]
    descs = [eng.make_search_desc(a.SEARCH_SINGLE, rc, f, box=box, pbc=7) for f in frames]
    want = [_dense(eng, d[0]) for d in descs]
    eng.search_resident_layout(True)
    tickets = [eng.search_resident_begin(descs[0][0])]
    for f in range(nframes):
        if f + 1 < nframes:
            tickets.append(eng.search_resident_begin(descs[f + 1][0]))
        cnt, p, d = eng.search_resident_end(tickets[f])
        assert cnt == len(want[f][0]), f
        got_p, got_d, _ = _from_segments(eng, tickets[f], cnt, p, d)
        assert np.array_equal(got_p, want[f][0]) and np.array_equal(got_d, want[f][1]), f
    eng.search_resident_layout(False)


def test_an_entry_that_outgrows_its_segment_repeats_the_frame(eng):
    """Same box, same grid, but the second frame has a third of its atoms squeezed into one corner: cell pairs there hold many
    times the results their capacities (from the uniform frame) allow.  The frame is repeated with its own counts; the result
    is the dense list."""
    import torch
    from molar_amd import api as a
    n, rc = 30_000, 0.8
    box = np.diag([7.0, 7.0, 7.0]).astype(np.float32)
    rng = np.random.default_rng(11)
    uniform = (rng.random((n, 3)) * 7.0).astype(np.float32)
    squeezed = uniform.copy()
    squeezed[: n // 3] = (rng.random((n // 3, 3)) * 1.6).astype(np.float32)
    fa, fb = torch.from_numpy(uniform).cuda(), torch.from_numpy(squeezed).cuda()
    da, ka = eng.make_search_desc(a.SEARCH_SINGLE, rc, fa, box=box, pbc=7)
    db, kb = eng.make_search_desc(a.SEARCH_SINGLE, rc, fb, box=box, pbc=7)
    want_a, want_b = _dense(eng, da), _dense(eng, db)
    assert len(want_b[0]) > 3 * len(want_a[0])
    eng.search_resident_layout(True)
    for desc, want in ((da, want_a), (db, want_b), (da, want_a), (db, want_b)):
        cnt, p, d = eng.search_resident_desc(desc)
        assert cnt == len(want[0])
        got_p, got_d, _ = _from_segments(eng, 0, cnt, p, d)
        assert np.array_equal(got_p, want[0]) and np.array_equal(got_d, want[1])
    # and pipelined: the overflowing frame is repeated inside _end while a younger search is in flight
    t0 = eng.search_resident_begin(da)
    t1 = eng.search_resident_begin(db)
    for t, want in ((t0, want_a), (t1, want_b)):
        cnt, p, d = eng.search_resident_end(t)
        got_p, got_d, _ = _from_segments(eng, t, cnt, p, d)
        assert np.array_equal(got_p, want[0]) and np.array_equal(got_d, want[1])
    eng.search_resident_layout(False)


def test_pairs_plane_only_and_a_changed_plan(eng):
    """(i, j) plane only (molar_hip_search_resident_planes(0)) in the segmented layout, and a cutoff change between searches
    (another grid: the capacities of the old plan are not used)."""
    import torch
    from molar_amd import api as a
    n = 20_000
    box = synth.box_a(n)
    pos = torch.from_numpy(synth.frame(n, box, 2)).cuda()
    eng.search_resident_layout(True)
    eng.search_resident_planes(False)
    for rc in (0.7, 1.1, 0.7):
        desc, keep = eng.make_search_desc(a.SEARCH_SINGLE, rc, pos, box=box, pbc=7)
        eng.search_resident_layout(False)
        eng.search_resident_planes(True)
        want_p, want_d = _dense(eng, desc)
        eng.search_resident_layout(True)
        eng.search_resident_planes(False)
        cnt, p, d = eng.search_resident_desc(desc)
        assert cnt == len(want_p) and d is None
        got_p, got_d, _ = _from_segments(eng, 0, cnt, p, d)
        assert got_d is None and np.array_equal(got_p, want_p)
        cp, _ = _compact(eng, 0, cnt, with_dist=False)
        assert np.array_equal(cp, want_p)
    eng.search_resident_planes(True)
    eng.search_resident_layout(False)


def test_empty_result_and_errors(eng):
    import torch
    from molar_amd import api as a
    from molar_amd._lib import MolarHipError
    box = np.diag([20.0, 20.0, 20.0]).astype(np.float32)
    pos = torch.from_numpy((np.arange(30, dtype=np.float32).reshape(10, 3) * 1.9) % 19.0).cuda().contiguous()
    with pytest.raises(MolarHipError):
        eng.search_segments(0)                      # dense layout: no segments
    eng.search_resident_layout(True)
    desc, keep = eng.make_search_desc(a.SEARCH_SINGLE, 0.05, pos, box=box, pbc=7)
    cnt, p, d = eng.search_resident_desc(desc)
    assert cnt == 0
    b, n, nseg, span = eng.search_segments(0)
    cp, cd = _compact(eng, 0, 0)
    assert len(cp) == 0
    vd = np.full(10, 0.1, np.float32)
    with pytest.raises(MolarHipError):
        eng.search_resident(a.SEARCH_DOUBLE_VDW, 0.0, pos, xyz2=pos, box=box, pbc=7, vdw1=vd, vdw2=vd)
    t = eng.search_resident_begin(desc)
    with pytest.raises(MolarHipError):
        eng.search_resident_layout(False)           # a pipelined search is in flight
    eng.search_resident_end(t)
    eng.search_resident_layout(False)



@pytest.mark.timeout(900)
def test_c2_one_million_atoms_segments_have_the_committed_digest(eng):
    """BASELINE config 2 in the segmented layout: the segments of the 1M-atom frame, compacted on the device, carry the SHA-256
    committed in tests/golden/ordered_pair_digests.json (the oracle's list) - on the first search and on the one after it."""
    import torch
    from molar_amd import api as a
    n, rc = 1_000_000, 1.2
    box = synth.box_a(n)
    want = json.load(open(os.path.join(G, "ordered_pair_digests.json")))["tric_a_1000000_rc1.2"]
    dpos = torch.from_numpy(synth.frame(n, box, 0)).cuda()
    desc, keep = eng.make_search_desc(a.SEARCH_SINGLE, rc, dpos, box=box, pbc=7)
    eng.search_resident_layout(True)
    for rep in range(2):
        cnt, paddr, daddr = eng.search_resident_desc(desc)
        assert cnt == want["npairs"]
        b, nn, nseg, span = eng.search_segments(0)
        assert nseg == 15 * 15 * 17 * 14 and span < cnt * 1.25
        pairs = torch.empty((cnt, 2), dtype=torch.int32, device="cuda")
        dist = torch.empty(cnt, dtype=torch.float32, device="cuda")
        eng.search_segments_compact(0, pairs.data_ptr(), dist.data_ptr())
        step = 1 << 24
        h = hashlib.sha256()
        for col in (0, 1):
            for k in range(0, cnt, step):
                h.update(pairs[k:k + step, col].contiguous().cpu().numpy().astype("<u4").tobytes())
        for k in range(0, cnt, step):
            h.update(dist[k:k + step].cpu().numpy().astype("<f4").tobytes())
        assert h.hexdigest() == want["sha256_i_j_d"], rep
        del pairs, dist
    eng.search_resident_layout(False)
