"""Parity at the sizes BASELINE.json quotes, element by element against the oracle run on the GPU box's host cores.

  C2  1M-atom triclinic box A, rc = 1.2 nm: the 3.6e8 ordered (i, j, d) triples of the HIP search against (a) the
      oracle's list, compared chunk by chunk, and (b) the SHA-256 committed in tests/golden/ordered_pair_digests.json
      (generated in the build container by tests/golden/make_golden.py c2 - an independent run of the oracle).
  C3  fit + RMSD + COM + gyration of the 100k-atom selection of a 1M-atom frame: single-call and batched paths against
      the f64 oracle at 1e-5 (measure.rs:485-570,613-643).
  C5  Membrane.compute on the 500k-atom bilayer (4000 lipids): validity, neighbour ids and vertex counts exact, floats
      within 2e-5 of the f32 oracle pipeline.
"""
import hashlib
import json
import os

import numpy as np
import pytest

from molar_amd import synth

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NCPU = os.cpu_count() or 8


@pytest.fixture(scope="module")
def eng():
    from molar_amd import build
    from molar_amd.api import Engine
    build.build_library()
    return Engine(0)


@pytest.mark.timeout(900)
def test_c2_host_buffers_streamed_result_has_the_committed_digest(eng):
    """What an unmodified MolAR caller sees (distance_search.rs:928-954 returns host Vecs): 12 MB of coordinates in from
    ordinary memory, the 4.3 GB ordered pair list out into ordinary memory.  Large results travel through the context's
    ring of pinned chunks and host threads (csrc/hoststream.hpp) instead of the runtime's pageable copy; both the
    (u32, u32, f32) entry and the (usize, usize, f32) entry - widened on the host threads - must carry exactly the
    committed list."""
    import time
    from molar_amd import api as a
    n, rc = 1_000_000, 1.2
    box = synth.box_a(n)
    pos = synth.frame(n, box, 0)
    want = json.load(open(os.path.join(G, "ordered_pair_digests.json")))["tric_a_1000000_rc1.2"]

    def digest(i, j, d):
        h = hashlib.sha256()
        step = 1 << 24
        for col in (i, j):
            for k in range(0, len(col), step):
                h.update(np.ascontiguousarray(col[k:k + step]).astype("<u4").tobytes())
        for k in range(0, len(d), step):
            h.update(d[k:k + step].astype("<f4").tobytes())
        return h.hexdigest()

    cnt = eng.search_count(a.SEARCH_SINGLE, rc, pos, box=box, pbc=7)
    assert cnt == want["npairs"]
    t0 = time.perf_counter()
    pairs, dist = eng.search_fill(cnt)
    dt = time.perf_counter() - t0
    assert digest(pairs[:, 0], pairs[:, 1], dist) == want["sha256_i_j_d"]
    del pairs
    i, j, d = eng.search_fill_usize(cnt)
    assert i.dtype == np.uint64 and digest(i, j, d) == want["sha256_i_j_d"]
    print(f"fill into pageable host memory: {dt:.3f} s, {cnt * 12 / dt / 1e9:.1f} GB/s")


@pytest.mark.timeout(1500)
def test_c2_one_million_atoms_element_wise_and_digest(eng, orc32):
    import torch
    from molar_amd import api as a
    n, rc = 1_000_000, 1.2
    box = synth.box_a(n)
    pos = synth.frame(n, box, 0)
    dpos = torch.from_numpy(pos).cuda()
    cnt, paddr, daddr = eng.search_resident(a.SEARCH_SINGLE, rc, dpos, box=box, pbc=7)
    assert eng.grid_dims() == (15, 15, 17)
    # the engine's result buffers as tensors (no copy); hashed and compared in chunks to bound host memory
    pairs = a.device_view(paddr, (cnt, 2), torch.int32)
    dist = a.device_view(daddr, (cnt,), torch.float32)
    want = json.load(open(os.path.join(G, "ordered_pair_digests.json")))["tric_a_1000000_rc1.2"]
    assert cnt == want["npairs"]
    step = 1 << 24
    h = hashlib.sha256()
    for col in (0, 1):
        for k in range(0, cnt, step):
            h.update(pairs[k:k + step, col].contiguous().cpu().numpy().astype("<u4").tobytes())
    for k in range(0, cnt, step):
        h.update(dist[k:k + step].cpu().numpy().astype("<f4").tobytes())
    assert h.hexdigest() == want["sha256_i_j_d"]
    # and against the oracle run right here on the box's host cores (the call bench.py's cpu_baseline leg times)
    ref = orc32.search_single_pbc(rc, pos, orc32.box_from_matrix(box), 7, nthreads=NCPU)
    assert len(ref["i"]) == cnt and tuple(ref["dims"]) == (15, 15, 17)
    for k in range(0, cnt, step):
        blk = pairs[k:k + step].cpu().numpy().view(np.uint32)
        assert np.array_equal(blk[:, 0], ref["i"][k:k + step].astype(np.uint32)), k
        assert np.array_equal(blk[:, 1], ref["j"][k:k + step].astype(np.uint32)), k
        assert np.array_equal(dist[k:k + step].cpu().numpy(), ref["d"][k:k + step]), k


@pytest.mark.timeout(900)
def test_c3_fit_rmsd_com_gyration_at_baseline_size(eng, orc64):
    """M = 1e5 of N = 1e6 (every 10th atom), masses cycled, reference = frame 0: the single-call entry points and the
    batched pipeline against the f64 oracle.  1e-5 relative for RMSD / gyration / COM, 1e-5 absolute for R."""
    import torch
    n = 1_000_000
    box = synth.box_a(n)
    ref = synth.frame(n, box, 0)
    mass = synth.masses(n)
    idx = np.arange(0, n, 10, dtype=np.uint64)
    frames = np.stack([synth.frame(n, box, f) for f in (1, 2, 3)])
    # rotate + shift frame 2 so the fit has something to undo
    from molar_amd import api as a
    Rz = a.rotation_from_axis_angle([0.2, 0.5, -0.8], 0.6).astype(np.float64)
    frames[1] = (frames[1].astype(np.float64) @ Rz.T + np.array([0.7, -1.1, 0.4])).astype(np.float32)
    want = []
    for f in range(3):
        cur = frames[f]
        R, t = orc64.fit_transform(cur, mass, ref, mass, idx, idx)
        moved = orc64.apply_transform(cur, R, t, idx)
        want.append(dict(R=R, t=t, rmsd=orc64.rmsd(moved, ref, idx, idx), com=orc64.center_of_mass(moved, mass, idx),
                         gyr=orc64.gyration(moved, mass, idx), moved=moved))
    # batched path, frames resident
    d_frames = torch.from_numpy(frames).cuda()
    out = eng.fit_rmsd_batch(d_frames, torch.from_numpy(mass).cuda(), torch.from_numpy(ref).cuda(),
                             idx=torch.from_numpy(idx.astype(np.int64)).cuda(), apply=True)
    got_moved = d_frames.cpu().numpy()
    for f in range(3):
        w = want[f]
        assert abs(out["rmsd"][f] - w["rmsd"]) <= 1e-5 * w["rmsd"], (f, out["rmsd"][f], w["rmsd"])
        assert abs(out["gyration"][f] - w["gyr"]) <= 1e-5 * w["gyr"]
        assert np.allclose(out["com"][f], w["com"], rtol=1e-5, atol=1e-5)
        assert np.allclose(out["R"][f], w["R"], atol=1e-5)
        # t = c2 - R c1 cancels two centres of ~10 nm: its error scales with THEIR magnitude, not with |t|
        assert np.allclose(out["t"][f], w["t"], rtol=1e-5, atol=1e-5 * max(np.abs(w["com"]).max(), np.abs(w["t"]).max()))
        sel = idx.astype(np.int64)
        assert np.abs(got_moved[f][sel] - w["moved"][sel]).max() < 1e-4          # f32 coordinates of ~20 nm: 4 ulp
        rest = np.ones(n, bool); rest[sel] = False
        assert np.array_equal(got_moved[f][rest], frames[f][rest])              # atoms outside the selection untouched
    # single-call path on host arrays
    for f in (0, 1):
        cur = frames[f].copy()
        R, t = eng.fit_transform(cur, mass, ref, mass, idx, idx)
        assert np.allclose(R, want[f]["R"], atol=1e-5)
        assert np.allclose(t, want[f]["t"], rtol=1e-5, atol=1e-5 * max(np.abs(want[f]["com"]).max(), np.abs(want[f]["t"]).max()))
        eng.apply_transform(cur, R, t, idx)
        r = eng.rmsd(cur, ref, idx, idx)
        assert abs(r - want[f]["rmsd"]) <= 1e-5 * want[f]["rmsd"]
        g = eng.gyration(cur, mass, idx)
        assert abs(g - want[f]["gyr"]) <= 1e-5 * want[f]["gyr"]
        assert np.allclose(eng.center_of_mass(cur, mass, idx), want[f]["com"], rtol=1e-5, atol=1e-5)
        rm = eng.rmsd_mw(cur, mass, ref, idx, idx)
        assert abs(rm - orc64.rmsd_mw(cur, mass, ref, idx, idx)) <= 1e-5 * rm


@pytest.mark.timeout(900)
def test_within_set_at_1m_atoms_against_the_oracle(eng, orc32):
    """`within 1.0 of <100k-atom selection>` on the 1M-atom frame (molar/benches/comparison_large.rs:29-40): the set form
    (molar_hip_within_count / _fill) against np.unique of the oracle's distance_search_within_pbc stream, for a compact
    solute and for a selection spread over the whole box; the engine's own stream form must give the same set."""
    from molar_amd import api as a
    n = 1_000_000
    box = synth.box_a(n)
    pos = synth.frame(n, box, 0)
    ob = orc32.box_from_matrix(box)
    all_idx = np.arange(n, dtype=np.uint64)
    centre = (box @ np.array([0.5, 0.5, 0.5], np.float32)).astype(np.float32)
    blob = np.sort(np.argsort(((pos - centre) ** 2).sum(1))[:100_000]).astype(np.uint64)
    for idx2 in (blob, all_idx[::10]):
        ref = orc32.search_within_pbc(1.0, pos, pos[idx2.astype(np.int64)], ob, 7, all_idx, idx2, nthreads=NCPU)
        want = np.unique(ref["i"])
        got = eng.within_set(1.0, pos, all_idx, pos, idx2, box=box, pbc=7)
        assert np.array_equal(got, want), (len(got), len(want))
        k = eng.search_count(a.SEARCH_WITHIN, 1.0, pos, all_idx, pos, idx2, box=box, pbc=7)
        assert k == len(ref["i"])
        assert np.array_equal(np.unique(eng.search_fill_ids(k)), want)
