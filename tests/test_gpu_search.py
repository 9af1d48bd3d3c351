"""GPU parity: the HIP search (through the C ABI) against the CPU oracle on identical inputs.

Bar (BASELINE.json north_star): neighbour indices and counts BIT-EXACT, in the reference's
output order (plan order, then i-major / j-minor, distance_search.rs:949-953).  Distances use
the same f32 operation order and a correctly rounded sqrt, so they are compared for exact
equality too (tolerance allowed by north_star: 1e-5 relative).
"""
import os

import numpy as np
import pytest

from molar_amd import synth

pytestmark = pytest.mark.gpu

PBC_FULL = 7


@pytest.fixture(scope="module")
def eng():
    from molar_amd import build
    from molar_amd.api import Engine
    build.build_library()
    return Engine(0)


def api():
    import molar_amd.api as a
    return a


def assert_same_pairs(got_i, got_j, got_d, ref, exact_d=True):
    assert len(got_i) == len(ref["i"]), (len(got_i), len(ref["i"]))
    assert np.array_equal(got_i.astype(np.uint64), ref["i"])
    assert np.array_equal(got_j.astype(np.uint64), ref["j"])
    if exact_d:
        assert np.array_equal(got_d, ref["d"])
    else:
        assert np.allclose(got_d, ref["d"], rtol=1e-5, atol=0)


def run_single(eng, cutoff, pos, box=None, pbc=0, idx=None, ids_local=False):
    a = api()
    n = eng.search_count(a.SEARCH_SINGLE, cutoff, pos, idx, box=box, pbc=pbc, ids_local=ids_local)
    pairs, d = eng.search_fill(n)
    return pairs[:, 0], pairs[:, 1], d, n


@pytest.mark.parametrize("boxfn,n,cutoff,pbc", [
    (synth.box_ortho, 4000, 0.45, 7),
    (synth.box_a, 4000, 0.5, 7),
    (synth.box_a, 20000, 0.8, 7),
    (synth.box_b, 6000, 0.5, 7),          # reference grid incomplete here: parity with the reference, not with brute force
    (synth.box_ortho, 3000, 0.5, 3),      # z non-periodic: drop rule + clamped cells
    (synth.box_ortho, 3000, 0.5, 5),
    (synth.box_a, 3000, 0.5, 1),          # partial pbc on a triclinic box: no triclinic correction (:304)
])
def test_single_pbc_bit_exact(eng, orc32, boxfn, n, cutoff, pbc):
    box = boxfn(n)
    pos = synth.frame(n, box, sigma=0.08)
    ob = orc32.box_from_matrix(box)
    ref = orc32.search_single_pbc(cutoff, pos, ob, pbc, nthreads=4)
    gi, gj, gd, cnt = run_single(eng, cutoff, pos, box, pbc)
    assert cnt == len(ref["i"]) > 0
    assert eng.grid_dims() == ref["dims"]
    assert_same_pairs(gi, gj, gd, ref)


def test_single_pbc_selection_global_and_local_ids(eng, orc32):
    n = 6000
    box = synth.box_a(n)
    pos = synth.frame(n, box)
    idx = np.arange(1, n, 3, dtype=np.uint64)
    ob = orc32.box_from_matrix(box)
    ref = orc32.search_single_pbc(0.7, pos[idx.astype(int)], ob, 7, ids=idx)
    gi, gj, gd, _ = run_single(eng, 0.7, pos, box, 7, idx=idx)
    assert_same_pairs(gi, gj, gd, ref)
    ref_l = orc32.search_single_pbc(0.7, pos[idx.astype(int)], ob, 7)
    gi, gj, gd, _ = run_single(eng, 0.7, pos, box, 7, idx=idx, ids_local=True)      # modify.rs:78 style
    assert_same_pairs(gi, gj, gd, ref_l)


def test_usize_fill_matches_u32_fill(eng):
    a = api()
    n = 3000
    box = synth.box_ortho(n)
    pos = synth.frame(n, box)
    cnt = eng.search_count(a.SEARCH_SINGLE, 0.5, pos, box=box, pbc=7)
    pairs, d = eng.search_fill(cnt)
    i, j, d2 = eng.search_fill_usize(cnt)
    assert i.dtype == np.uint64 and np.array_equal(i, pairs[:, 0]) and np.array_equal(j, pairs[:, 1])
    assert np.array_equal(d, d2)


def test_large_cells_streaming_path(eng, orc32):
    """> 512 atoms per cell exercises the path that re-reads the second cell from memory."""
    n = 6000
    box = np.diag([3.3, 3.3, 3.3]).astype(np.float32)       # dims 3x3x3 at cutoff 1.0 -> ~220/cell ... use 2 sets
    pos = synth.frame(n, box)
    ob = orc32.box_from_matrix(box)
    ref = orc32.search_single_pbc(1.05, pos, ob, 7, nthreads=4)
    assert ref["dims"] == (3, 3, 3)
    gi, gj, gd, _ = run_single(eng, 1.05, pos, box, 7)
    assert_same_pairs(gi, gj, gd, ref)
    n = 5000
    box = np.diag([2.4, 2.4, 7.5]).astype(np.float32)       # 2x2x6 cells (duplicates quirk) with ~208/cell
    pos = synth.frame(n, box)
    ob = orc32.box_from_matrix(box)
    ref = orc32.search_single_pbc(1.2, pos, ob, 7, nthreads=4)
    gi, gj, gd, _ = run_single(eng, 1.2, pos, box, 7)
    assert_same_pairs(gi, gj, gd, ref)
    # one cell holding everything (dims 1,1,1) with 1500 atoms: streaming + triangle
    n = 1500
    box = np.diag([1.9, 1.9, 1.9]).astype(np.float32)
    pos = synth.frame(n, box)
    ob = orc32.box_from_matrix(box)
    ref = orc32.search_single_pbc(1.0, pos, ob, 7, nthreads=4)
    assert ref["dims"] == (1, 1, 1)
    gi, gj, gd, _ = run_single(eng, 1.0, pos, box, 7)
    assert_same_pairs(gi, gj, gd, ref)


@pytest.mark.parametrize("case", ["single_tric", "single_pbc_xy", "single_no_box", "double_tric"])
def test_cells_of_513_to_1024_atoms_stay_in_registers(eng, orc32, case):
    """Frames whose cells hold more than 448 atoms on average run the count / fill instances with 128 registers per lane, which keep
    up to 16 chunks (1024 atoms) of the second cell resident: plain, same-cell and band-classified wrapped entries (evaluated in
    both passes, no hit history), resident and count + fill entries, against the oracle - ids, order, distances."""
    a = api()
    n, rc = 120_000, 1.6
    box = synth.box_a(n)                                    # 5 x 5 x 6 cells of ~800 atoms
    pos = synth.frame(n, box, 4)
    ob = orc32.box_from_matrix(box)
    if case.startswith("single"):
        pbc = {"single_tric": 7, "single_pbc_xy": 3, "single_no_box": 0}[case]
        if pbc:
            ref = orc32.search_single_pbc(rc, pos, ob, pbc, nthreads=16)
            kw = dict(box=box, pbc=pbc)
            if pbc == 7:
                assert tuple(ref["dims"]) == (5, 5, 6)
        else:
            ref = orc32.search_single(rc, pos, nthreads=16)
            kw = {}
        cnt = eng.search_count(a.SEARCH_SINGLE, rc, pos, **kw)
        pr, d = eng.search_fill(cnt)
        cnt2, _, _ = eng.search_resident(a.SEARCH_SINGLE, rc, pos, **kw)
        pr2, d2 = eng.search_fill(cnt2)
    else:
        rng = np.random.default_rng(9)
        i1 = np.sort(rng.choice(n, n // 2, replace=False)).astype(np.uint64)
        i2 = np.sort(rng.choice(n, (3 * n) // 4, replace=False)).astype(np.uint64)        # overlaps i1: same-cell duplicates
        ref = orc32.search_double_pbc(rc, pos[i1.astype(int)], pos[i2.astype(int)], ob, 7, ids1=i1, ids2=i2, nthreads=16)
        cnt = eng.search_count(a.SEARCH_DOUBLE, rc, pos, i1, pos, i2, box=box, pbc=7)
        pr, d = eng.search_fill(cnt)
        cnt2, _, _ = eng.search_resident(a.SEARCH_DOUBLE, rc, pos, i1, pos, i2, box=box, pbc=7)
        pr2, d2 = eng.search_fill(cnt2)
    assert cnt == cnt2 == len(ref["i"]) > 1e7
    assert np.array_equal(pr[:, 0], ref["i"]) and np.array_equal(pr[:, 1], ref["j"]) and np.array_equal(d, ref["d"])
    assert np.array_equal(pr, pr2) and np.array_equal(d, d2)


@pytest.mark.parametrize("case", ["single_tric_4x4x5", "single_tric_3x3x4", "single_pbc_xy", "single_no_box", "double_tric", "single_above_2048"])
def test_cells_of_1025_to_2048_atoms_stay_in_registers(eng, orc32, case):
    """Frames whose cells hold more than 1000 atoms on average (rc >= 1.9 nm at water density; the reference's sweep goes to 4.2 nm,
    benches/within_size_bench.rs:13-47) run the count / fill instances with 168 registers per lane: up to 32 chunks (2048 atoms) of the
    second cell resident, chunk count rounded up to 16 / 20 / 24 / 28 / 32.  Plain, same-cell and wrapped entries (band-classified
    with >= 4 cells per periodic dimension, exact below), both entry forms, against the oracle - ids, order, distances; and a frame
    of cells ABOVE 2048 atoms, which the same instances still stream (distance_search.rs:432-517)."""
    a = api()
    n, rc = {"single_tric_4x4x5": (120_000, 2.0), "single_tric_3x3x4": (60_000, 2.0), "single_pbc_xy": (60_000, 2.0), "single_no_box": (60_000, 2.0),
             "double_tric": (100_000, 2.0), "single_above_2048": (40_000, 2.4)}[case]
    box = synth.box_a(n)
    pos = synth.frame(n, box, 5)
    ob = orc32.box_from_matrix(box)
    if case.startswith("single"):
        pbc = {"single_pbc_xy": 3, "single_no_box": 0}.get(case, 7)
        if pbc:
            ref = orc32.search_single_pbc(rc, pos, ob, pbc, nthreads=16)
            kw = dict(box=box, pbc=pbc)
        else:
            ref = orc32.search_single(rc, pos, nthreads=16)
            kw = {}
        cnt = eng.search_count(a.SEARCH_SINGLE, rc, pos, **kw)
        dims = eng.grid_dims()
        pr, d = eng.search_fill(cnt)
        cnt2, _, _ = eng.search_resident(a.SEARCH_SINGLE, rc, pos, **kw)
        pr2, d2 = eng.search_fill(cnt2)
        per_cell = n / (dims[0] * dims[1] * dims[2])
    else:
        rng = np.random.default_rng(9)
        i1 = np.sort(rng.choice(n, (2 * n) // 3, replace=False)).astype(np.uint64)
        i2 = np.sort(rng.choice(n, (5 * n) // 6, replace=False)).astype(np.uint64)        # overlaps i1: same-cell duplicates
        ref = orc32.search_double_pbc(rc, pos[i1.astype(int)], pos[i2.astype(int)], ob, 7, ids1=i1, ids2=i2, nthreads=16)
        cnt = eng.search_count(a.SEARCH_DOUBLE, rc, pos, i1, pos, i2, box=box, pbc=7)
        dims = eng.grid_dims()
        pr, d = eng.search_fill(cnt)
        cnt2, _, _ = eng.search_resident(a.SEARCH_DOUBLE, rc, pos, i1, pos, i2, box=box, pbc=7)
        pr2, d2 = eng.search_fill(cnt2)
        per_cell = len(i2) / (dims[0] * dims[1] * dims[2])
    if case == "single_tric_4x4x5":
        assert tuple(dims) == (4, 4, 5)
    if case != "single_no_box":          # (without a box the grid comes from the bounding box padded by the cutoff: other cells)
        assert per_cell > (2048 if case == "single_above_2048" else 1000)
        if case != "single_above_2048":
            assert per_cell < 2048
    assert cnt == cnt2 == len(ref["i"]) > 3e7
    assert np.array_equal(pr[:, 0], ref["i"]) and np.array_equal(pr[:, 1], ref["j"]) and np.array_equal(d, ref["d"])
    assert np.array_equal(pr, pr2) and np.array_equal(d, d2)


@pytest.mark.parametrize("rc", [0.35, 0.47])
@pytest.mark.parametrize("case", ["single_tric", "single_pbc_xy", "single_no_box", "double_tric"])
def test_small_cells_several_slots_per_wave(eng, orc32, case, rc):
    """Frames of cells of a few atoms (up to 13 per cell on average: 16 lanes per slot, up to 19: 32) run the kernels of
    pair_small.hip - 2 or 4 slots per wave, every distance by the exact formula in both passes.  Plain, same-cell, wrapped and
    triclinic corner entries, crowded cells inside the sparse grid (second cells longer than a slot's lanes, first cells of more
    than 64 rows: several slots per entry), resident and count + fill entries, against the oracle - ids, order, distances
    (distance_search.rs:324-373,432-517)."""
    a = api()
    n = 60_000
    box = synth.box_a(n)
    pos = synth.frame(n, box, 6)
    rng = np.random.default_rng(21)
    for size in (40, 100, 150):                              # crowded cells
        centre = box @ rng.uniform(0.2, 0.8, 3).astype(np.float32)
        at = rng.choice(n, size, replace=False)
        pos[at] = (centre + rng.normal(0, 0.05, (size, 3))).astype(np.float32)
    ob = orc32.box_from_matrix(box)
    if case.startswith("single"):
        pbc = {"single_tric": 7, "single_pbc_xy": 3, "single_no_box": 0}[case]
        if pbc:
            ref = orc32.search_single_pbc(rc, pos, ob, pbc, nthreads=8)
            kw = dict(box=box, pbc=pbc)
        else:
            ref = orc32.search_single(rc, pos, nthreads=8)
            kw = {}
        cnt = eng.search_count(a.SEARCH_SINGLE, rc, pos, **kw)
        dims = eng.grid_dims()
        pr, d = eng.search_fill(cnt)
        cnt2, _, _ = eng.search_resident(a.SEARCH_SINGLE, rc, pos, **kw)
        pr2, d2 = eng.search_fill(cnt2)
        n_max = n
    else:
        i1 = np.sort(rng.choice(n, n // 2, replace=False)).astype(np.uint64)
        i2 = np.sort(rng.choice(n, (3 * n) // 4, replace=False)).astype(np.uint64)        # overlaps i1: same-cell duplicates
        ref = orc32.search_double_pbc(rc, pos[i1.astype(int)], pos[i2.astype(int)], ob, 7, ids1=i1, ids2=i2, nthreads=8)
        cnt = eng.search_count(a.SEARCH_DOUBLE, rc, pos, i1, pos, i2, box=box, pbc=7)
        dims = eng.grid_dims()
        pr, d = eng.search_fill(cnt)
        cnt2, _, _ = eng.search_resident(a.SEARCH_DOUBLE, rc, pos, i1, pos, i2, box=box, pbc=7)
        pr2, d2 = eng.search_fill(cnt2)
        n_max = len(i2)
    assert n_max <= 19 * dims[0] * dims[1] * dims[2]        # the frame is one of those the small-cell kernels take
    assert cnt == cnt2 == len(ref["i"]) > 1e5
    assert np.array_equal(pr[:, 0], ref["i"]) and np.array_equal(pr[:, 1], ref["j"]) and np.array_equal(d, ref["d"])
    assert np.array_equal(pr, pr2) and np.array_equal(d, d2)


@pytest.mark.parametrize("n,rc", [(500_000, 0.55), (200_000, 0.3)])
def test_plans_of_more_than_2_18_entries(eng, orc32, n, rc):
    """Plans of 2^18 ... 2^22 entries (a large frame at a contact cutoff) take the plan in three launches: tile totals, totals of
    groups of 128 tiles, offsets and slot records (plan_groups_kernel).  500k atoms at 0.55 nm: 2.9e5 entries through the regular
    count / fill kernels; 200k atoms at 0.3 nm: 6.5e5 entries through the small-cell kernels.  Resident and count + fill entries
    against the oracle - ids, order, distances (search_plan, distance_search.rs:217-269)."""
    a = api()
    box = synth.box_a(n)
    pos = synth.frame(n, box, 8)
    ob = orc32.box_from_matrix(box)
    ref = orc32.search_single_pbc(rc, pos, ob, 7, nthreads=16)
    cnt = eng.search_count(a.SEARCH_SINGLE, rc, pos, box=box, pbc=7)
    dims = eng.grid_dims()
    assert (1 << 18) < 14 * dims[0] * dims[1] * dims[2] < (1 << 22)
    pr, d = eng.search_fill(cnt)
    cnt2, _, _ = eng.search_resident(a.SEARCH_SINGLE, rc, pos, box=box, pbc=7)
    pr2, d2 = eng.search_fill(cnt2)
    assert cnt == cnt2 == len(ref["i"]) > 1e6
    assert np.array_equal(pr[:, 0], ref["i"]) and np.array_equal(pr[:, 1], ref["j"]) and np.array_equal(d, ref["d"])
    assert np.array_equal(pr, pr2) and np.array_equal(d, d2)


def test_tiny_inputs(eng, orc32):
    box = np.diag([5.0, 5.0, 5.0]).astype(np.float32)
    ob = orc32.box_from_matrix(box)
    for pos in ([[1.0, 1.0, 1.0]], [[1.0, 1.0, 1.0], [1.2, 1.0, 1.0]], [[0.1, 0.1, 0.1], [4.9, 4.9, 4.9]],
                [[1.0, 1.0, 1.0], [3.0, 3.0, 3.0]]):
        pos = np.array(pos, np.float32)
        ref = orc32.search_single_pbc(0.5, pos, ob, 7)
        gi, gj, gd, cnt = run_single(eng, 0.5, pos, box, 7)
        assert cnt == len(ref["i"])
        assert_same_pairs(gi, gj, gd, ref)
        ref = orc32.search_single(0.5, pos)
        gi, gj, gd, cnt = run_single(eng, 0.5, pos)
        assert_same_pairs(gi, gj, gd, ref)


def test_single_nonpbc_bit_exact(eng, orc32):
    n = 5000
    box = synth.box_ortho(n)
    pos = synth.frame(n, box) - 1.5
    ref = orc32.search_single(0.5, pos, nthreads=4)
    gi, gj, gd, _ = run_single(eng, 0.5, pos)
    assert eng.grid_dims() == ref["dims"]
    assert_same_pairs(gi, gj, gd, ref)
    # all-positive coordinates: the zero-seeded bounding box (distance_search.rs:602-616) still contains the origin
    pos2 = synth.frame(n, box) + 5.0
    ref = orc32.search_single(0.5, pos2, nthreads=4)
    gi, gj, gd, _ = run_single(eng, 0.5, pos2)
    assert eng.grid_dims() == ref["dims"]
    assert_same_pairs(gi, gj, gd, ref)


@pytest.mark.parametrize("pbc", [7, 0, 6])
def test_double_bit_exact_with_duplicates(eng, orc32, pbc):
    a = api()
    n = 6000
    box = synth.box_a(n)
    pos = synth.frame(n, box, sigma=0.08)
    idx1 = np.arange(0, n, 2, dtype=np.uint64)
    idx2 = np.arange(1, n, 2, dtype=np.uint64)
    ob = orc32.box_from_matrix(box)
    p1, p2 = pos[idx1.astype(int)], pos[idx2.astype(int)]
    if pbc:
        ref = orc32.search_double_pbc(0.6, p1, p2, ob, pbc, idx1, idx2, nthreads=4)
        cnt = eng.search_count(a.SEARCH_DOUBLE, 0.6, pos, idx1, pos, idx2, box=box, pbc=pbc)
    else:
        ref = orc32.search_double(0.6, p1, p2, idx1, idx2, nthreads=4)
        cnt = eng.search_count(a.SEARCH_DOUBLE, 0.6, pos, idx1, pos, idx2)
    pairs, d = eng.search_fill(cnt)
    assert_same_pairs(pairs[:, 0], pairs[:, 1], d, ref)
    # the reference's same-cell double emission (distance_search.rs:741-749) is reproduced
    _, c = np.unique(pairs, axis=0, return_counts=True)
    assert c.max() == 2


@pytest.mark.parametrize("kind", ["double", "vdw"])
def test_sparse_plan_small_set_in_a_large_one(eng, orc32, kind):
    """A compact solute against its solvent (command_solvate.rs:94-101, `within`-like selections): most plan entries have
    an empty cell on one side.  The host-synchronous search reads the plan's real slot count back and launches one workgroup
    per slot that exists (size_plan); the result has to be what the full launch gave."""
    a = api()
    n = 120000
    box = synth.box_a(n)
    pos = synth.frame(n, box, sigma=0.08)
    centre = (box @ np.array([0.5, 0.5, 0.5], np.float32)).astype(np.float32)
    order = np.argsort(((pos - centre) ** 2).sum(1))
    solute = np.sort(order[:3000]).astype(np.uint64)
    mask = np.ones(n, bool)
    mask[solute.astype(int)] = False
    solvent = np.nonzero(mask)[0].astype(np.uint64)
    ob = orc32.box_from_matrix(box)
    p1, p2 = pos[solvent.astype(int)], pos[solute.astype(int)]
    if kind == "double":
        ref = orc32.search_double_pbc(0.35, p1, p2, ob, 7, solvent, solute, nthreads=4)
        cnt = eng.search_count(a.SEARCH_DOUBLE, 0.35, pos, solvent, pos, solute, box=box, pbc=7)
    else:
        rng = np.random.default_rng(3)
        vdw = rng.choice(np.array([0.12, 0.152, 0.17, 0.21], np.float32), n)
        v1, v2 = vdw[solvent.astype(int)], vdw[solute.astype(int)]
        ref = orc32.search_double_vdw_pbc(p1, p2, v1, v2, ob, 7, nthreads=4)        # ids local to the two sets, as in the reference
        cnt = eng.search_count(a.SEARCH_DOUBLE_VDW, None, pos, solvent, pos, solute, box=box, pbc=7, vdw1=v1, vdw2=v2)
    assert cnt == len(ref["i"]) > 0
    assert int(np.prod(eng.grid_dims())) * 28 > 65536      # a plan large enough for the slot count to be read back (size_plan)
    pairs, d = eng.search_fill(cnt)
    assert_same_pairs(pairs[:, 0], pairs[:, 1], d, ref)


@pytest.mark.parametrize("pbc", [7, 0])
def test_within_bit_exact(eng, orc32, pbc):
    a = api()
    n = 8000
    box = synth.box_a(n)
    pos = synth.frame(n, box)
    idx1 = np.arange(0, n, dtype=np.uint64)
    idx2 = np.arange(100, 400, dtype=np.uint64)
    ob = orc32.box_from_matrix(box)
    p2 = pos[idx2.astype(int)]
    if pbc:
        ref = orc32.search_within_pbc(0.6, pos, p2, ob, pbc, idx1, idx2, nthreads=4)
        cnt = eng.search_count(a.SEARCH_WITHIN, 0.6, pos, idx1, pos, idx2, box=box, pbc=pbc)
    else:
        lo, up = orc32.min_max(pos)                                    # selection/ast.rs:597-602
        lo = lo + (np.float32(-0.6) - np.float32(1.1920929e-07))
        up = up + (np.float32(0.6) + np.float32(1.1920929e-07))
        ref = orc32.search_within(0.6, pos, p2, lo, up, idx1, idx2, nthreads=4)
        cnt = eng.search_count(a.SEARCH_WITHIN, 0.6, pos, idx1, pos, idx2, lower=lo, upper=up)
    ids = eng.search_fill_ids(cnt)
    assert cnt == len(ref["i"]) > 0
    assert np.array_equal(ids, ref["i"])


@pytest.mark.parametrize("pbc", [7, 0])
def test_vdw_bit_exact(eng, orc32, pbc):
    a = api()
    n = 6000
    box = synth.box_ortho(n, density=60.0)
    pos = synth.frame(n, box)
    idx1 = np.arange(0, n // 2, dtype=np.uint64)
    idx2 = np.arange(n // 2, n, dtype=np.uint64)
    rng = np.random.default_rng(2)
    v1 = rng.uniform(0.1, 0.2, len(idx1)).astype(np.float32)
    v2 = rng.uniform(0.1, 0.2, len(idx2)).astype(np.float32)
    ob = orc32.box_from_matrix(box)
    p1, p2 = pos[idx1.astype(int)], pos[idx2.astype(int)]
    if pbc:
        ref = orc32.search_double_vdw_pbc(p1, p2, v1, v2, ob, pbc, nthreads=4)
        cnt = eng.search_count(a.SEARCH_DOUBLE_VDW, None, pos, idx1, pos, idx2, box=box, pbc=pbc, vdw1=v1, vdw2=v2)
    else:
        ref = orc32.search_double_vdw(p1, p2, v1, v2, nthreads=4)
        cnt = eng.search_count(a.SEARCH_DOUBLE_VDW, None, pos, idx1, pos, idx2, vdw1=v1, vdw2=v2)
    pairs, d = eng.search_fill(cnt)
    assert cnt > 0
    assert_same_pairs(pairs[:, 0], pairs[:, 1], d, ref)     # LOCAL ids (:791-792)


def test_vdw_nan_radii_are_ignored_by_the_cutoff_maximum(eng, orc32):
    """vdw.iter().cloned().reduce(Float::max) ignores NaN (distance_search.rs:781-783). Every radius a wave's lane 0 reads
    is NaN here (every 64th entry): the maximum has to come from the other lanes, and a pair with a NaN radius never hits."""
    a = api()
    n = 6000
    box = synth.box_ortho(n, density=60.0)
    pos = synth.frame(n, box)
    idx1 = np.arange(0, n // 2, dtype=np.uint64)
    idx2 = np.arange(n // 2, n, dtype=np.uint64)
    rng = np.random.default_rng(5)
    v1 = rng.uniform(0.1, 0.2, len(idx1)).astype(np.float32)
    v2 = rng.uniform(0.1, 0.2, len(idx2)).astype(np.float32)
    v1[::64] = np.nan
    v2[::64] = np.nan
    ob = orc32.box_from_matrix(box)
    p1, p2 = pos[idx1.astype(int)], pos[idx2.astype(int)]
    ref = orc32.search_double_vdw_pbc(p1, p2, v1, v2, ob, 7, nthreads=4)
    cnt = eng.search_count(a.SEARCH_DOUBLE_VDW, None, pos, idx1, pos, idx2, box=box, pbc=7, vdw1=v1, vdw2=v2)
    pairs, d = eng.search_fill(cnt)
    assert cnt > 0
    assert_same_pairs(pairs[:, 0], pairs[:, 1], d, ref)


@pytest.mark.parametrize("pbc", [0, 7])
def test_vdw_sparse_grid_with_crowded_cells(eng, orc32, pbc):
    """Grids of many cells of a few atoms are placed by 16 lanes per cell (place_small_kernel); a few crowded cells in such a grid
    (blobs of 20-150 atoms inside one cell) take the same path in several trips and keep the push order
    (distance_search.rs:180,203-209)."""
    a = api()
    n = 9000
    box = synth.box_ortho(n, density=25.0)
    pos = synth.frame(n, box)
    rng = np.random.default_rng(11)
    L = np.diag(box)
    for blob, size in enumerate((20, 33, 64, 150)):
        centre = rng.uniform(0.2, 0.8, 3) * L
        at = rng.choice(n, size, replace=False)
        pos[at] = (centre + rng.normal(0, 0.03, (size, 3))).astype(np.float32)
    idx1 = np.sort(rng.choice(n, 6000, replace=False)).astype(np.uint64)
    idx2 = np.setdiff1d(np.arange(n, dtype=np.uint64), idx1)
    v1 = rng.uniform(0.1, 0.2, len(idx1)).astype(np.float32)
    v2 = rng.uniform(0.1, 0.2, len(idx2)).astype(np.float32)
    ob = orc32.box_from_matrix(box)
    p1, p2 = pos[idx1.astype(int)], pos[idx2.astype(int)]
    if pbc:
        ref = orc32.search_double_vdw_pbc(p1, p2, v1, v2, ob, pbc, nthreads=4)
        cnt = eng.search_count(a.SEARCH_DOUBLE_VDW, None, pos, idx1, pos, idx2, box=box, pbc=pbc, vdw1=v1, vdw2=v2)
    else:
        ref = orc32.search_double_vdw(p1, p2, v1, v2, nthreads=4)
        cnt = eng.search_count(a.SEARCH_DOUBLE_VDW, None, pos, idx1, pos, idx2, vdw1=v1, vdw2=v2)
    dims = eng.grid_dims()
    assert len(idx1) < 16 * dims[0] * dims[1] * dims[2]          # the sparse-grid placement is the one that ran
    pairs, d = eng.search_fill(cnt)
    assert cnt > 500
    assert_same_pairs(pairs[:, 0], pairs[:, 1], d, ref)


def test_pymolar_style_dispatch(eng, orc32):
    """distance_search(cutoff, sel1, sel2=None, dims=None) dispatch table (molar_python/src/lib.rs:271-362)."""
    a = api()
    n = 4000
    box = synth.box_a(n)
    pos = synth.frame(n, box)
    pb = a.PeriodicBox.from_matrix(box)
    top = a.Topology(synth.masses(n), vdw=np.full(n, 0.15, np.float32))
    st = a.State(pos, pb)
    s1 = a.Sel(top, st, np.arange(0, n, 2), engine=eng)
    s2 = a.Sel(top, st, np.arange(1, n, 2), engine=eng)
    ob = orc32.box_from_matrix(box)
    pairs, d = a.distance_search(0.5, s1, dims=[True, True, True])
    ref = orc32.search_single_pbc(0.5, pos[s1.index.astype(int)], ob, 7, ids=s1.index)
    assert pairs.shape == (len(ref["i"]), 2) and pairs.dtype == np.uint64
    assert np.array_equal(pairs[:, 0], ref["i"]) and np.array_equal(pairs[:, 1], ref["j"]) and np.array_equal(d, ref["d"])
    pairs, d = a.distance_search(0.5, s1)
    ref = orc32.search_single(0.5, pos[s1.index.astype(int)], ids=s1.index)
    assert np.array_equal(pairs[:, 0], ref["i"]) and np.array_equal(d, ref["d"])
    pairs, d = a.distance_search(0.5, s1, s2, dims=[True, True, True])
    ref = orc32.search_double_pbc(0.5, pos[s1.index.astype(int)], pos[s2.index.astype(int)], ob, 7, s1.index, s2.index)
    assert np.array_equal(pairs[:, 0], ref["i"]) and np.array_equal(pairs[:, 1], ref["j"])
    pairs, d = a.distance_search("vdw", s1, s2)
    ref = orc32.search_double_vdw(pos[s1.index.astype(int)], pos[s2.index.astype(int)],
                                  top.vdw[s1.index.astype(int)], top.vdw[s2.index.astype(int)])
    assert np.array_equal(pairs[:, 0], s1.index[ref["i"].astype(int)])       # local -> global (:348-354)
    assert np.array_equal(pairs[:, 1], s2.index[ref["j"].astype(int)])
    with pytest.raises(NotImplementedError):
        a.distance_search("vdw", s1)
    with pytest.raises(TypeError):
        a.distance_search("foo", s1, s2)
    assert len(d) == len(pairs) and (d >= 0).all()        # test_2.py:248-257


def test_medium_box_against_oracle(eng, orc32):
    """250k atoms, the config-4 frame size, cutoff 1.2 nm: ~9e7 pairs compared element by element."""
    n = 250_000
    box = synth.box_a(n)
    pos = synth.frame(n, box)
    ob = orc32.box_from_matrix(box)
    ref = orc32.search_single_pbc(1.2, pos, ob, 7, nthreads=8)
    gi, gj, gd, cnt = run_single(eng, 1.2, pos, box, 7)
    assert cnt == len(ref["i"])
    assert_same_pairs(gi, gj, gd, ref)


def test_full_size_properties(eng):
    """Config 2 (1M atoms, box A, rc = 1.2 nm): size-independent properties of the result."""
    import torch
    a = api()
    n = 1_000_000
    box = synth.box_a(n)
    pos = synth.frame(n, box)
    dpos = torch.from_numpy(pos).cuda()
    cnt = eng.search_count(a.SEARCH_SINGLE, 1.2, dpos, box=box, pbc=7)
    assert eng.grid_dims() == (15, 15, 17)
    assert 3.3e8 < cnt < 3.9e8                               # ~724 neighbours/atom at rho = 100 nm^-3
    pairs = torch.empty((cnt, 2), dtype=torch.int32, device="cuda")
    dist = torch.empty(cnt, dtype=torch.float32, device="cuda")
    eng.search_fill_into(pairs, dist)
    eng.synchronize()
    assert cnt == eng.search_count(a.SEARCH_SINGLE, 1.2, dpos, box=box, pbc=7)     # idempotent
    i = pairs[:, 0].long(); j = pairs[:, 1].long()
    assert int((i == j).sum()) == 0
    assert float(dist.max()) <= float(np.float32(1.2)) and float(dist.min()) >= 0.0      # sqrt(d2 <= rc*rc) in f32
    # no duplicate pairs: the (min,max) keys are unique
    key = torch.minimum(i, j) * n + torch.maximum(i, j)
    assert int(torch.unique(key).numel()) == cnt
    del key
    # every reported distance is the minimum-image distance of its pair (float64 recomputation)
    sel = torch.randint(0, cnt, (2_000_000,), device="cuda")
    M = torch.from_numpy(box.astype(np.float64)).cuda()
    v = (dpos[j[sel]] - dpos[i[sel]]).double()
    f = v @ torch.linalg.inv(M).T
    f = f - torch.round(f)
    best = None
    for sx in (-1, 0, 1):
        for sy in (-1, 0, 1):
            for sz in (-1, 0, 1):
                w = (f + torch.tensor([sx, sy, sz], device="cuda", dtype=torch.float64)) @ M.T
                dd = w.norm(dim=1)
                best = dd if best is None else torch.minimum(best, dd)
    assert float((best - dist[sel].double()).abs().max()) < 5e-5
    # degree sum: each atom's neighbour count is consistent with the density
    deg = torch.bincount(i, minlength=n) + torch.bincount(j, minlength=n)
    assert abs(float(deg.float().mean()) - 2.0 * cnt / n) < 1e-3
    assert 600 < float(deg.float().mean()) < 800


@pytest.mark.parametrize("n,cutoff,nbins", [(20000, 0.8, 400), (250_000, 1.2, 1200)])
def test_fused_histogram_bit_identical(eng, orc32, n, cutoff, nbins):
    """Config 4's consumer: radial distance histogram without materialising pairs.  Integer bins must
    equal Histogram1D::add_one (stats.rs:29-35) applied to the reference's distance stream."""
    a = api()
    box = synth.box_a(n)
    pos = synth.frame(n, box, 3)
    ob = orc32.box_from_matrix(box)
    ref = orc32.search_single_pbc(cutoff, pos, ob, 7, nthreads=8)
    want = orc32.histogram_add(0.0, cutoff, nbins, ref["d"]).astype(np.uint64)
    bins, cnt = eng.search_histogram(a.SEARCH_SINGLE, cutoff, 0.0, cutoff, nbins, pos, box=box, pbc=7)
    assert cnt == len(ref["i"])
    assert np.array_equal(bins, want)
    # accumulation over frames: a second call adds into the same bins
    bins2, cnt2 = eng.search_histogram(a.SEARCH_SINGLE, cutoff, 0.0, cutoff, nbins, pos, box=box, pbc=7, bins=bins)
    assert np.array_equal(bins2, 2 * want) and cnt2 == cnt
    # out-of-range distances are dropped (b >= n): histogram over half the range
    half, _ = eng.search_histogram(a.SEARCH_SINGLE, cutoff, 0.0, cutoff / 2, nbins, pos, box=box, pbc=7)
    want_half = orc32.histogram_add(0.0, cutoff / 2, nbins, ref["d"]).astype(np.uint64)
    assert np.array_equal(half, want_half)


@pytest.mark.parametrize("case", ["rhombic", "ortho", "big_cells", "big_cells_band", "big_cells_two_sets", "big_cells_no_box", "pbc_xy", "no_box", "few_cells", "hmin", "vdw", "two_sets_rhombic"])
def test_fused_histogram_equals_the_distance_stream(eng, orc32, case):
    """Every class of plan entry of the fused histogram (plain / same-cell / band-classified wrapped entries in the lean
    kernel; triclinic corner entries, cells > 512 atoms, boxes without the band classification, vdW radii in the generic
    one) against Histogram1D::add_one (stats.rs:29-35) applied to the distance stream of the pair search, which the
    tests above pin to the oracle."""
    a = api()
    rng = np.random.default_rng(11)
    kind, n, cutoff, nbins, hmin, pbc, box, kw = a.SEARCH_SINGLE, 30000, 0.9, 450, 0.0, 7, None, {}
    if case == "rhombic":
        box = synth.box_b(n)
    elif case == "ortho":
        box = synth.box_ortho(n)
    elif case == "big_cells":
        cutoff, nbins = 2.1, 700            # ~900 atoms per cell, 3 cells per dimension: plain entries in blocks of 512 second-cell atoms, wrapped ones exact
        box = synth.box_a(n)
    elif case == "big_cells_band":
        n, cutoff, nbins = 100_000, 1.9, 950     # ~800 atoms per cell, 5 cells per dimension: band-classified wrapped entries in blocks too
        box = synth.box_a(n)
    elif case == "big_cells_two_sets":
        kind, n, cutoff, nbins = a.SEARCH_DOUBLE, 100_000, 2.0, 800
        box = synth.box_a(n)
    elif case == "big_cells_no_box":
        cutoff, nbins, pbc = 2.1, 700, 0
        box = synth.box_a(n)
    elif case == "pbc_xy":
        box, pbc = synth.box_a(n), 3
    elif case == "no_box":
        box, pbc = synth.box_a(n), 0
    elif case == "few_cells":
        n, cutoff = 4000, 1.2               # 2-3 cells per dimension: wrapped entries evaluated exactly
        box = synth.box_b(n)
    elif case == "hmin":
        box, hmin = synth.box_a(n), 0.35
    elif case == "vdw":
        kind, box = a.SEARCH_DOUBLE_VDW, synth.box_a(n)
    elif case == "two_sets_rhombic":
        kind, box = a.SEARCH_DOUBLE, synth.box_b(n)
    pos = synth.frame(n, box, 5)
    hmax = cutoff
    if kind == a.SEARCH_SINGLE:
        args = dict(xyz1=pos)
    else:
        idx1 = np.sort(rng.choice(n, n // 2, replace=False)).astype(np.uint64)
        idx2 = np.setdiff1d(np.arange(n, dtype=np.uint64), idx1)
        args = dict(xyz1=pos, idx1=idx1, xyz2=pos, idx2=idx2)
        if kind == a.SEARCH_DOUBLE_VDW:
            cutoff, hmax, nbins = None, 0.5, 250
            kw = dict(vdw1=rng.uniform(0.1, 0.22, len(idx1)).astype(np.float32),
                      vdw2=rng.uniform(0.1, 0.22, len(idx2)).astype(np.float32))
    use_box = box if case not in ("no_box", "big_cells_no_box") else None
    cnt = eng.search_count(kind, cutoff, box=use_box, pbc=pbc, **args, **kw)
    _, dist = eng.search_fill(cnt)
    assert cnt > 1000
    want = orc32.histogram_add(hmin, hmax, nbins, dist).astype(np.uint64)
    bins, c2 = eng.search_histogram(kind, cutoff, hmin, hmax, nbins, box=use_box, pbc=pbc, **args, **kw)
    assert c2 == cnt
    assert np.array_equal(bins, want)


@pytest.mark.parametrize("nbins", [1, 2, 4096, 8191, 8192])
def test_fused_histogram_extreme_bin_counts(eng, orc32, nbins):
    """1 bin, and the largest table the histogram kernel keeps in LDS (8192 counters + 8193 bin edges next to the queues)."""
    a = api()
    n = 20000
    box = synth.box_a(n)
    pos = synth.frame(n, box, 2)
    ref = orc32.search_single_pbc(0.8, pos, orc32.box_from_matrix(box), 7, nthreads=8)
    want = orc32.histogram_add(0.0, 0.8, nbins, ref["d"]).astype(np.uint64)
    bins, cnt = eng.search_histogram(a.SEARCH_SINGLE, 0.8, 0.0, 0.8, nbins, pos, box=box, pbc=7)
    assert cnt == len(ref["i"]) and np.array_equal(bins, want)
    with pytest.raises(Exception):
        eng.search_histogram(a.SEARCH_SINGLE, 0.8, 0.0, 0.8, 8193, pos, box=box, pbc=7)


@pytest.mark.parametrize("hmin,hmax,nbins", [(0.749, 0.75, 8192), (0.7499, 0.75, 8192), (0.59999, 0.6, 4096), (0.0, 1.0e-3, 8192)])
def test_fused_histogram_bins_narrower_than_an_ulp(eng, orc32, hmin, hmax, nbins):
    """Bins narrower than the spacing of the floats they cover: Histogram1D::add_one's formula then skips bins, neighbouring
    entries of the kernel's exact edge table coincide and its first guess of the bin (v_sqrt_f32, one multiply) can be several
    bins off - hist_add walks the monotone edge table until the squared distance lies inside the bin (advisor finding, round 2)."""
    a = api()
    n = 20000
    box = synth.box_a(n)
    pos = synth.frame(n, box, 4)
    ref = orc32.search_single_pbc(0.8, pos, orc32.box_from_matrix(box), 7, nthreads=8)
    want = orc32.histogram_add(hmin, hmax, nbins, ref["d"]).astype(np.uint64)
    bins, cnt = eng.search_histogram(a.SEARCH_SINGLE, 0.8, hmin, hmax, nbins, pos, box=box, pbc=7)
    assert cnt == len(ref["i"]) and np.array_equal(bins, want)
    if hmin > 0.0:
        assert want.sum() > 50           # the window is not empty


def test_fused_histogram_regression_case_of_the_fuzzer(eng, orc32):
    """tools/fuzz_search.py, seed 123, case 500: a two-set search in a triclinic box periodic in y only, grid (1, 4, 2),
    ~400 atoms of the second set per cell (7 chunks: the register-hungriest instance of the histogram kernel's wrapped
    path).  ROCm 7.2's compiler had placed a VGPR copy of a wave-uniform value ahead of an EXEC restore there, and 1.6 %
    of the hits landed in bin 1 (molar_amd/build.py now audits the ISA for the pattern; the kernel keeps such values in
    SGPRs).  Fixture: tests/golden/hist_regression_case.npz (positions, box, the two index sets, parameters)."""
    a = api()
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hist_regression_case.npz"))
    pos, box, i1, i2 = g["pos"], g["box"], g["i1"], g["i2"]
    rc, pbc, nb, hmin, hmax = float(g["rc"]), int(g["pbc"]), int(g["nb"]), float(g["hmin"]), float(g["hmax"])
    ob = orc32.box_from_matrix(box)
    for s1, s2 in ((i1, i2), (i2, i2), (i2, i1)):
        ref = orc32.search_double_pbc(rc, pos[s1.astype(int)], pos[s2.astype(int)], ob, pbc, ids1=s1, ids2=s2, nthreads=4)
        for lo, hi, n in ((hmin, hmax, nb), (0.0, hmax, nb), (0.0, rc, 300)):
            want = orc32.histogram_add(lo, hi, n, ref["d"]).astype(np.uint64)
            bins, cnt = eng.search_histogram(a.SEARCH_DOUBLE, rc, lo, hi, n, pos, s1, pos, s2, box=box, pbc=pbc)
            assert cnt == len(ref["i"]) and np.array_equal(bins, want), (len(s1), len(s2), lo, hi, n)


def test_fused_histogram_two_sets(eng, orc32):
    a = api()
    n = 12000
    box = synth.box_a(n)
    pos = synth.frame(n, box, 1)
    idx1 = np.arange(0, n, 2, dtype=np.uint64); idx2 = np.arange(1, n, 2, dtype=np.uint64)
    ob = orc32.box_from_matrix(box)
    ref = orc32.search_double_pbc(0.7, pos[idx1.astype(int)], pos[idx2.astype(int)], ob, 7, idx1, idx2)
    want = orc32.histogram_add(0.0, 0.7, 350, ref["d"]).astype(np.uint64)
    bins, cnt = eng.search_histogram(a.SEARCH_DOUBLE, 0.7, 0.0, 0.7, 350, pos, idx1, pos, idx2, box=box, pbc=7)
    assert cnt == len(ref["i"]) and np.array_equal(bins, want)     # includes the reference's same-cell duplicates


def test_within_selection_end_to_end(eng, orc32):
    """`within 0.6 [pbc] [self] of <inner>` (selection/ast.rs:589-631) = search stream + sort/dedup."""
    a = api()
    n = 9000
    box = synth.box_a(n)
    pos = synth.frame(n, box)
    top = a.Topology(synth.masses(n))
    st = a.State(pos, a.PeriodicBox.from_matrix(box))
    outer = a.Sel(top, st, np.arange(0, n, 1), engine=eng)
    inner = a.Sel(top, st, np.arange(2000, 2300), engine=eng)
    ob = orc32.box_from_matrix(box)
    # periodic
    got = outer.within(0.6, inner, pbc=[True, True, True])
    ref = orc32.search_within_pbc(0.6, pos, pos[2000:2300], ob, 7, np.arange(n), np.arange(2000, 2300))
    assert np.array_equal(got, np.unique(ref["i"]))
    # brute-force meaning of the selection: min-image distance to any inner atom <= cutoff
    bf = orc32.brute_double(0.6, pos, pos[2000:2300], ob, 7, np.arange(n), np.arange(2000, 2300))
    assert np.array_equal(got, np.unique(bf["i"]))
    # non-periodic + `self`
    got = outer.within(0.6, inner, include_inner=True)
    bf = orc32.brute_double(0.6, pos, pos[2000:2300], None, ids1=np.arange(n), ids2=np.arange(2000, 2300))
    assert np.array_equal(got, np.unique(np.concatenate([bf["i"], np.arange(2000, 2300)])))


def test_unwrap_connectivity(eng, orc32):
    """modify.rs:72-131: chains broken across the box are made whole again."""
    a = api()
    rng = np.random.default_rng(4)
    box = np.diag([4.0, 4.0, 4.0]).astype(np.float32)
    chains = []
    for c in range(12):                                    # 12 random-walk chains of 40 beads, bond 0.15 nm
        p = np.zeros((40, 3)); p[0] = rng.uniform(0, 4, 3)
        for k in range(1, 40):
            d = rng.normal(size=3); d /= np.linalg.norm(d)
            p[k] = p[k - 1] + 0.15 * d
        chains.append(p)
    whole = np.concatenate(chains)
    wrapped = (whole % 4.0).astype(np.float32)
    top = a.Topology(np.ones(len(wrapped), np.float32))
    st = a.State(wrapped.copy(), a.PeriodicBox.from_matrix(box))
    sel = a.Sel(top, st, engine=eng)
    groups = sel.unwrap_connectivity(0.2)
    un = st.coords
    # every bond is restored to its length, whatever image the chain ended up in
    for c in range(12):
        seg = un[40 * c: 40 * (c + 1)].astype(np.float64)
        assert np.allclose(np.linalg.norm(np.diff(seg, axis=0), axis=1), 0.15, atol=1e-4)
    assert sum(len(g) for g in groups) <= len(wrapped) and len(groups) >= 1
    # same walk with the oracle's primitives gives the same coordinates
    ob = orc32.box_from_matrix(box)
    r = orc32.search_single_pbc(0.2, wrapped, ob, 7)
    conn = [[] for _ in range(len(wrapped))]
    for i, j in zip(r["i"].tolist(), r["j"].tolist()):
        conn[i].append(j); conn[j].append(i)
    ref = wrapped.copy(); used = np.zeros(len(ref), bool); todo = [0]; used[0] = True
    while True:
        while todo:
            c = todo.pop(); p0 = ref[c].copy()
            for ind in conn[c]:
                if not used[ind]:
                    ref[ind] = orc32.closest_image_dims(ob, ref[ind], p0, 7); todo.append(ind); used[ind] = True
        rest = np.nonzero(~used)[0]
        if not len(rest):
            break
        todo.append(int(rest[0])); used[int(rest[0])] = True
    assert np.array_equal(un, ref)


@pytest.mark.parametrize("boxkind", ["ortho", "tric"])
def test_wrapped_pairs_at_the_cutoff_edge(eng, orc32, boxkind):
    """Wrapped cell pairs are classified with a cheap distance and decided exactly only inside a narrow band
    around cutoff^2 (pair_kernels.hpp, run_fast).  Stress exactly that band: thousands of pairs straddling the
    periodic boundary whose distance is cutoff*(1 +- 3e-4), where one ulp decides membership."""
    rng = np.random.default_rng(21)
    L = 9.0
    box = np.diag([L, L, L]).astype(np.float32)
    if boxkind == "tric":
        box[0, 2] = -1.5; box[1, 2] = -1.5; box[0, 1] = 0.7
    rc = 1.0
    npairs = 6000
    M = box.astype(np.float64)
    pts = []
    for k in range(npairs):
        # a near the far face of a random periodic dim, b = a + u*d mapped back into the box
        fa = rng.random(3)
        dim = int(rng.integers(0, 3))
        fa[dim] = 1.0 - 0.04 * rng.random()
        a = M @ fa
        u = rng.normal(size=3); u /= np.linalg.norm(u)
        d = rc * (1.0 + rng.uniform(-3e-4, 3e-4))
        b = a + d * u
        fb = np.linalg.solve(M, b)
        if not ((fb < 0) | (fb >= 1)).any():
            continue                                   # keep only pairs that really cross the boundary
        fb = fb % 1.0
        pts.append(a); pts.append(M @ fb)
    pos = np.array(pts, np.float32)
    # background atoms so that cells are populated like a real system
    pos = np.concatenate([pos, synth.frame(4000, box, 5)])
    ob = orc32.box_from_matrix(box)
    ref = orc32.search_single_pbc(rc, pos, ob, 7, nthreads=4)
    assert min(ref["dims"]) >= 4                        # the approximate classification is active
    gi, gj, gd, cnt = run_single(eng, rc, pos, box, 7)
    assert cnt == len(ref["i"])
    assert_same_pairs(gi, gj, gd, ref)
    # the band really was exercised: many reference distances lie within 3e-4 of the cutoff
    assert (np.abs(ref["d"] / rc - 1.0) < 3e-4).sum() > 1000
    # and the fused histogram agrees too (flush-time recomputation of wrapped d2)
    a = api()
    want = orc32.histogram_add(0.0, rc, 500, ref["d"]).astype(np.uint64)
    bins, c2 = eng.search_histogram(a.SEARCH_SINGLE, rc, 0.0, rc, 500, pos, box=box, pbc=7)
    assert c2 == cnt and np.array_equal(bins, want)


def _boundary_pairs(box, rc, npairs, seed, emin=-6.0, emax=-4.0):
    """Pairs that cross a periodic face with distance rc*(1 +- 10^e), e uniform in [emin, emax]."""
    rng = np.random.default_rng(seed)
    M = box.astype(np.float64)
    L = float(np.abs(M).sum(1).max())
    pts = []
    while len(pts) < 2 * npairs:
        fa = rng.random(3)
        dim = int(rng.integers(0, 3))
        fa[dim] = 1.0 - (0.5 * rc / L) * rng.random()
        a = M @ fa
        u = rng.normal(size=3); u /= np.linalg.norm(u)
        e = 10.0 ** rng.uniform(emin, emax) * rng.choice([-1.0, 1.0])
        b = a + rc * (1.0 + e) * u
        fb = np.linalg.solve(M, b)
        if not ((fb < 0) | (fb >= 1)).any():
            continue
        pts.append(a); pts.append(M @ (fb % 1.0))
    return np.array(pts, np.float32)


@pytest.mark.parametrize("case", ["ortho_30nm", "tric_12nm", "no_box_far_from_origin", "crowded_cells", "nan_atom"])
def test_matrix_core_count_band_adversarial(eng, orc32, case):
    """The count pass of plain and same-cell entries decides d2 <= cutoff^2 from a matrix-core evaluation of f16 hi/lo
    splits and re-decides exactly inside an error band (pair_kernels.hpp, run_count_mfma).  Stress that band: tens of
    thousands of INTERIOR pairs (no periodic image involved) at rc * (1 +- 1e-8 .. 1e-4), where one ulp of the f32
    formula decides membership; plus the guards - coordinates far from the origin without a box, cells beyond the
    register-resident size, a NaN coordinate.  A single wrong count shifts every later slot's output offset, so the lists
    must be bit-identical to the oracle's."""
    a = api()
    rng = np.random.default_rng(5)
    rc, use_box, pbc, shift = 1.0, True, 7, 0.0
    if case == "ortho_30nm":
        box = np.diag([30.0, 30.0, 30.0]).astype(np.float32)
    elif case == "tric_12nm":
        box = np.array([[12.0, 0.0, -2.0], [0.0, 12.0, -2.0], [0.0, 0.0, 12.0]], np.float32)
    elif case == "no_box_far_from_origin":
        box = np.diag([14.0, 14.0, 14.0]).astype(np.float32); use_box, pbc, shift = False, 0, 30.0     # the error bound works on coordinates relative to the cell
    elif case == "crowded_cells":
        box = np.diag([8.0, 8.0, 8.0]).astype(np.float32)
    else:
        box = np.diag([10.0, 10.0, 10.0]).astype(np.float32)
    M = box.astype(np.float64)
    npairs = 20000
    fa = 0.15 + 0.7 * rng.random((npairs, 3))                     # interior: no pair reaches a periodic face
    pa = fa @ M.T
    u = rng.normal(size=(npairs, 3)); u /= np.linalg.norm(u, axis=1)[:, None]
    e = 10.0 ** rng.uniform(-8.0, -4.0, npairs) * rng.choice([-1.0, 1.0], npairs)
    pb = pa + rc * (1.0 + e)[:, None] * u
    pos = np.concatenate([pa, pb]).astype(np.float32)
    nbg = 60000 if case == "crowded_cells" else 8000              # crowded: ~120 atoms/nm^3 -> cells beyond 320 atoms
    pos = np.concatenate([pos, (rng.random((nbg, 3)) @ M.T).astype(np.float32)]) + np.float32(shift)
    if case == "nan_atom":
        pos[12345] = np.nan
    ob = orc32.box_from_matrix(box)
    if use_box:
        ref = orc32.search_single_pbc(rc, pos, ob, pbc, nthreads=8)
        cnt = eng.search_count(a.SEARCH_SINGLE, rc, pos, box=box, pbc=pbc)
    else:
        ref = orc32.search_single(rc, pos, nthreads=8)
        cnt = eng.search_count(a.SEARCH_SINGLE, rc, pos)
    pr, d = eng.search_fill(cnt)
    near = np.abs(ref["d"].astype(np.float64) / rc - 1.0)
    assert (near < 1e-4).sum() > 8000 and (near < 1e-6).sum() > 1500       # the half of the constructed pairs inside the cutoff
    assert cnt == len(ref["i"])
    assert np.array_equal(pr[:, 0], ref["i"]) and np.array_equal(pr[:, 1], ref["j"]) and np.array_equal(d, ref["d"])


@pytest.mark.parametrize("scale", [0.01, 0.02, 0.04, 300.0])
def test_matrix_core_count_band_at_other_length_scales(eng, orc32, scale, monkeypatch):
    """The f16 operands of the matrix-core count pass have ABSOLUTE limits the relative error terms do not see: a lo part
    below 2^-14 is a subnormal f16 with a fixed quantum of 2^-24 (cutoffs of a few hundredths of a unit: the error no longer
    shrinks with the coordinates), and |a|^2 - cutoff^2 overflows f16 once cutoff^2 exceeds 65504 (pair_kernels.hpp,
    mfma_error_bound / mfma_bound_usable).  The band stress of the test above, scaled: tens of thousands of interior pairs at
    rc * (1 +- 1e-8 .. 1e-4); at scale 300 (cutoff^2 = 9e4) a blob whose cells are tight enough for the path to be tried.
    A wrong count shifts every later slot's output, so lists must equal the oracle's bit for bit - with the matrix-core
    count and, in a build with the A/B knobs, with the vector count (MOLAR_HIP_NO_MFMA_COUNT) alike."""
    a = api()
    rng = np.random.default_rng(11)
    rc = np.float32(1.0 * scale)
    if scale < 1.0:
        box = (np.diag([12.0, 12.0, 12.0]) * scale).astype(np.float32)
        M = box.astype(np.float64)
        npairs = 20000
        pa = (0.15 + 0.7 * rng.random((npairs, 3))) @ M.T
        u = rng.normal(size=(npairs, 3)); u /= np.linalg.norm(u, axis=1)[:, None]
        e = 10.0 ** rng.uniform(-8.0, -4.0, npairs) * rng.choice([-1.0, 1.0], npairs)
        pb = pa + float(rc) * (1.0 + e)[:, None] * u
        pos = np.concatenate([pa, pb, rng.random((8000, 3)) @ M.T]).astype(np.float32)
        ob = orc32.box_from_matrix(box)
        ref = orc32.search_single_pbc(float(rc), pos, ob, 7, nthreads=8)
        kw = dict(box=box, pbc=7)
    else:
        # no box: one blob of radius ~20 units, cutoff 300: a single cell (of <= 320 atoms per block column: 300 atoms)
        pos = (rng.normal(size=(300, 3)) * 8.0).astype(np.float32)
        ref = orc32.search_single(float(rc), pos, nthreads=4)
        kw = {}
    # the release library has one code path; a build with -DMOLAR_HIP_AB_KNOBS (tools/build_variant.sh) also runs the vector count
    from molar_amd import _lib
    lib_path = os.environ.get("MOLAR_HIP_PLUGIN") or _lib.DEFAULT_LIB
    has_knobs = b"MOLAR_HIP_NO_MFMA_COUNT" in open(lib_path, "rb").read()
    for env in ((None, "1") if has_knobs else (None,)):
        if env:
            monkeypatch.setenv("MOLAR_HIP_NO_MFMA_COUNT", env)
        e2 = a.Engine(0)          # the knob is read when a context is created
        cnt = e2.search_count(a.SEARCH_SINGLE, float(rc), pos, **kw)
        pr, d = e2.search_fill(cnt)
        assert cnt == len(ref["i"]), (scale, env)
        assert np.array_equal(pr[:, 0], ref["i"]) and np.array_equal(pr[:, 1], ref["j"]) and np.array_equal(d, ref["d"])
    if scale < 1.0:
        near = np.abs(ref["d"].astype(np.float64) / float(rc) - 1.0)
        assert (near < 1e-4).sum() > 8000


@pytest.mark.timeout(900)
@pytest.mark.parametrize("boxkind,L", [("dodecahedron", 60.0), ("sheared", 40.0), ("sheared_huge", 160.0)])
def test_wrapped_band_adversarial_large_sheared_boxes(eng, orc32, boxkind, L):
    """The band / pruning margins of the wrapped fast path scale with ulp(L) and with the conditioning of the box
    (search.hip make_params).  Adversarial cases at large L/rc: a 60-degree rhombic dodecahedron of 60 nm and a strongly
    sheared 40 nm box at rc = 0.3 with >= 1e4 boundary-crossing pairs at rc*(1 +- 1e-6..1e-4), where the f32 evaluation
    of inv*v / M*f carries errors comparable to the distance from the cutoff; and a 160 nm sheared box at rc = 1.0 (the
    conditioning term dominates the margin).  Lists must be bit-identical to the oracle's."""
    rc = 0.3 if L < 100 else 1.0
    if boxkind == "dodecahedron":
        box = np.array([[L, 0, L / 2], [0, L, L / 2], [0, 0, L * np.sqrt(2) / 2]], np.float32)
    else:
        box = np.array([[L, 0.45 * L, 0.9 * L], [0, L, 0.8 * L], [0, 0, L]], np.float32)
    pos = _boundary_pairs(box, rc, 12000, seed=int(L))
    ob = orc32.box_from_matrix(box)
    ref = orc32.search_single_pbc(rc, pos, ob, 7, nthreads=8)
    assert min(ref["dims"]) >= 4
    near = np.abs(ref["d"].astype(np.float64) / rc - 1.0)
    # (the reference's half-shell grid is incomplete in positively sheared boxes, so not every constructed pair is found)
    assert len(ref["i"]) > 5000 and (near < 1e-4).sum() > 1000 and (near < 1e-5).sum() > 100
    gi, gj, gd, cnt = run_single(eng, rc, pos, box, 7)
    assert cnt == len(ref["i"])
    assert_same_pairs(gi, gj, gd, ref)
    a = api()
    c2, pa, da = eng.search_resident(a.SEARCH_SINGLE, rc, pos, box=box, pbc=7)
    assert c2 == cnt


def test_non_finite_coordinates_and_degenerate_inputs(eng, orc32):
    """NaN / inf coordinates never compare as hits in the reference (and land in cell 0 through the
    saturating `as usize` cast); the engine must agree and must not fault.  Also: coincident atoms (d = 0)."""
    a = api()
    n = 3000
    box = synth.box_a(n)
    pos = synth.frame(n, box)
    pos[10] = [np.nan, 1.0, 1.0]
    pos[11] = [np.inf, 1.0, 1.0]
    pos[12] = [1.0, -np.inf, np.nan]
    pos[500] = pos[499]                      # coincident pair: d2 == 0, sqrt(0) == 0 exactly
    ob = orc32.box_from_matrix(box)
    ref = orc32.search_single_pbc(0.5, pos, ob, 7)
    gi, gj, gd, cnt = run_single(eng, 0.5, pos, box, 7)
    assert cnt == len(ref["i"])
    assert_same_pairs(gi, gj, gd, ref)
    assert not np.isin([10, 11, 12], np.concatenate([gi, gj])).any()
    hit = (gi == 499) & (gj == 500)
    assert hit.sum() == 1 and gd[hit][0] == 0.0
    # non-periodic driver: NaN only (an infinite coordinate makes the reference's zero-seeded bounding box,
    # hence its grid, infinite - it cannot run there either)
    pos2 = pos.copy()
    pos2[11] = [2.0, 1.0, 1.0]
    pos2[12] = [1.0, 2.0, np.nan]
    ref = orc32.search_single(0.5, pos2)
    gi, gj, gd, cnt = run_single(eng, 0.5, pos2)
    assert_same_pairs(gi, gj, gd, ref)
    # an empty second set
    cnt = eng.search_count(a.SEARCH_DOUBLE, 0.5, pos, np.arange(100, dtype=np.uint64), pos, np.zeros(0, np.uint64), box=box, pbc=7)
    assert cnt == 0


def test_slab_with_a_large_cutoff_moves_to_the_large_cell_instances(orc32):
    """The same for large cells: 160 atoms per cell on average (regular kernels), ~900 per occupied cell - from the second frame on the
    instances that keep up to 1024 atoms of the second cell in registers; the lists stay the oracle's."""
    a = api()
    from molar_amd.api import Engine
    e = Engine(0)
    rng = np.random.default_rng(22)
    L, H, thick, rc = 8.0, 30.0, 3.0, 2.2
    n = int(L * L * thick * 100)
    box = np.diag([L, L, H]).astype(np.float32)
    ob = orc32.box_from_matrix(box)
    kinds = []
    for f in range(3):
        pos = (rng.random((n, 3)) * np.array([L, L, thick]) + np.array([0, 0, 0.5 * (H - thick)])).astype(np.float32)
        ref = orc32.search_single_pbc(rc, pos, ob, 7, nthreads=8)
        cnt, _, _ = e.search_resident(a.SEARCH_SINGLE, rc, pos, box=box, pbc=7)
        pr, d = e.search_fill(cnt)
        assert cnt == len(ref["i"]) > 1e6
        assert np.array_equal(pr[:, 0], ref["i"]) and np.array_equal(pr[:, 1], ref["j"]) and np.array_equal(d, ref["d"]), f
        kinds.append(e.search_cell_kernels()[0])
    assert kinds[0] == 0 and kinds[-1] in (-1, -2), kinds


@pytest.mark.parametrize("kind", ["single", "double"])
def test_slab_in_a_mostly_empty_box_switches_kernels_between_frames(orc32, kind):
    """A slab of liquid density in a periodic box that is mostly empty: few atoms per cell on average, many per OCCUPIED cell.  The
    first searches of a context judge by the average (small-cell kernels); the grid build counts the occupied cells, the host
    learns the count with the result sizes, and later searches of the same shape go to the regular kernels (hit history and all).
    Every frame, before and after the switch, synchronous and pipelined, is the oracle's ordered list bit for bit."""
    a = api()
    from molar_amd.api import Engine
    e = Engine(0)
    rng = np.random.default_rng(21)
    L, H, thick, rc = 8.0, 30.0, 3.0, 0.9
    n = int(L * L * thick * 100)
    box = np.diag([L, L, H]).astype(np.float32)
    ob = orc32.box_from_matrix(box)
    i1 = np.sort(rng.choice(n, n // 2, replace=False)).astype(np.uint64)
    i2 = np.sort(rng.choice(n, n // 2, replace=False)).astype(np.uint64)
    frames = [(rng.random((n, 3)) * np.array([L, L, thick]) + np.array([0, 0, 0.5 * (H - thick)])).astype(np.float32) for _ in range(6)]

    def want(pos):
        if kind == "single":
            return orc32.search_single_pbc(rc, pos, ob, 7, nthreads=8)
        return orc32.search_double_pbc(rc, pos[i1.astype(int)], pos[i2.astype(int)], ob, 7, ids1=i1, ids2=i2, nthreads=8)

    def run(pos):
        if kind == "single":
            cnt, _, _ = e.search_resident(a.SEARCH_SINGLE, rc, pos, box=box, pbc=7)
        else:
            cnt, _, _ = e.search_resident(a.SEARCH_DOUBLE, rc, pos, i1, pos, i2, box=box, pbc=7)
        return e.search_fill(cnt)
    lanes = []
    for f in range(4):
        ref = want(frames[f])
        pr, d = run(frames[f])
        assert len(ref["i"]) > 1e5
        assert np.array_equal(pr[:, 0], ref["i"]) and np.array_equal(pr[:, 1], ref["j"]) and np.array_equal(d, ref["d"]), f
        lanes.append(e.search_cell_kernels()[0])
    assert lanes[0] in (16, 32) and lanes[-1] == 0, lanes          # the average said small cells; the occupied cells said otherwise
    # the stream form (count, then fill) learns the same way
    e3 = Engine(0)
    lanes3 = []
    for f in range(3):
        ref = want(frames[f])
        if kind == "single":
            cnt = e3.search_count(a.SEARCH_SINGLE, rc, frames[f], box=box, pbc=7)
        else:
            cnt = e3.search_count(a.SEARCH_DOUBLE, rc, frames[f], i1, frames[f], i2, box=box, pbc=7)
        pr, d = e3.search_fill(cnt)
        assert np.array_equal(pr[:, 0], ref["i"]) and np.array_equal(pr[:, 1], ref["j"]) and np.array_equal(d, ref["d"]), f
        lanes3.append(e3.search_cell_kernels()[0])
    assert lanes3[0] in (16, 32) and lanes3[1] == 0 and lanes3[2] == 0, lanes3
    if kind == "single":
        import torch
        e2 = Engine(0)
        dev = [torch.from_numpy(f).cuda() for f in frames]
        descs = [e2.make_search_desc(a.SEARCH_SINGLE, rc, f, box=box, pbc=7) for f in dev]
        counts, prev = [], None
        for k in range(len(dev)):
            t = e2.search_resident_begin(descs[k][0])
            if prev is not None:
                counts.append(e2.search_resident_end(prev)[0])
            prev = t
        cnt, pp, dp = e2.search_resident_end(prev)
        counts.append(cnt)
        assert counts == [len(want(f)["i"]) for f in frames]
        class Dev:
            def __init__(self, ptr, n, typestr):
                self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}
        ref = want(frames[-1])
        got_pairs = torch.as_tensor(Dev(pp, cnt * 2, "<i4"), device="cuda").cpu().numpy().view(np.uint32).reshape(-1, 2)
        got_dist = torch.as_tensor(Dev(dp, cnt, "<f4"), device="cuda").cpu().numpy()
        assert np.array_equal(got_pairs[:, 0], ref["i"]) and np.array_equal(got_pairs[:, 1], ref["j"]) and np.array_equal(got_dist, ref["d"])
        assert e2.search_cell_kernels()[0] == 0 and e2.search_cell_kernels()[1][0] > 0


def test_resident_single_round_trip_matches_count_then_fill():
    """molar_hip_search_resident: same ordered result as count + fill, through the grow-and-repeat logic (fresh
    context: every buffer starts empty; then larger and smaller systems on the same context)."""
    import torch
    a = api()
    from molar_amd.api import Engine
    e1, e2 = Engine(0), Engine(0)
    for n, boxfn, rc in ((3000, synth.box_a, 0.5), (20000, synth.box_a, 0.7), (5000, synth.box_b, 0.6), (20000, synth.box_a, 0.7),
                         (4000, synth.box_ortho, 0.45)):
        box = boxfn(n)
        pos = synth.frame(n, box, sigma=0.08)
        want_n = e1.search_count(a.SEARCH_SINGLE, rc, pos, box=box, pbc=7)
        want_p, want_d = e1.search_fill(want_n)
        cnt, pp, dp = e2.search_resident(a.SEARCH_SINGLE, rc, pos, box=box, pbc=7)
        assert cnt == want_n > 0
        # the engine-owned device buffers hold the ordered result
        class Dev:
            def __init__(self, ptr, n, typestr):
                self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}
        e2.synchronize()
        got_pairs = torch.as_tensor(Dev(pp, cnt * 2, "<i4"), device="cuda").cpu().numpy().view(np.uint32).reshape(-1, 2)
        got_dist = torch.as_tensor(Dev(dp, cnt, "<f4"), device="cuda").cpu().numpy()
        assert np.array_equal(got_pairs, want_p) and np.array_equal(got_dist, want_d)
        # the cached search of the resident call serves the host fill too
        got_p, got_d = e2.search_fill(cnt)
        assert np.array_equal(got_p, want_p) and np.array_equal(got_d, want_d)
    # two-set search and vdw through the same entry
    n = 6000
    box = synth.box_ortho(n, density=60.0)
    pos = synth.frame(n, box)
    i1 = np.arange(0, n // 2, dtype=np.uint64); i2 = np.arange(n // 2, n, dtype=np.uint64)
    want_n = e1.search_count(a.SEARCH_DOUBLE, 0.6, pos, i1, pos, i2, box=box, pbc=7)
    want = e1.search_fill(want_n)
    cnt, _, _ = e2.search_resident(a.SEARCH_DOUBLE, 0.6, pos, i1, pos, i2, box=box, pbc=7)
    got = e2.search_fill(cnt)
    assert cnt == want_n and np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    v = np.random.default_rng(2).uniform(0.1, 0.2, n // 2).astype(np.float32)
    want_n = e1.search_count(a.SEARCH_DOUBLE_VDW, None, pos, i1, pos, i2, vdw1=v, vdw2=v)
    want = e1.search_fill(want_n)
    cnt, _, _ = e2.search_resident(a.SEARCH_DOUBLE_VDW, None, pos, i1, pos, i2, vdw1=v, vdw2=v)
    got = e2.search_fill(cnt)
    assert cnt == want_n and np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    from molar_amd._lib import MolarHipError
    with pytest.raises(MolarHipError):
        e2.search_resident(a.SEARCH_WITHIN, 0.5, pos, i1, pos, i2, box=box, pbc=7)


def test_resident_pipelined_begin_end_matches_the_plain_call():
    """molar_hip_search_resident_begin/_end: two frames in flight on one stream, results per ticket equal to
    count + fill - through the repeat logic (fresh context: the first frames outgrow every buffer while a
    younger search is already queued), with frames that grow and shrink, and the misuse errors."""
    import torch
    a = api()
    from molar_amd.api import Engine
    from molar_amd._lib import MolarHipError
    e1, e2 = Engine(0), Engine(0)

    class Dev:
        def __init__(self, ptr, n, typestr):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}

    cases = [(3000, synth.box_a, 0.5), (3000, synth.box_a, 0.5), (20000, synth.box_a, 0.7), (5000, synth.box_b, 0.6),
             (20000, synth.box_a, 0.7), (21000, synth.box_a, 0.72), (4000, synth.box_ortho, 0.45), (20000, synth.box_a, 0.7)]
    frames, want = [], []
    for k, (n, boxfn, rc) in enumerate(cases):
        box = boxfn(n)
        pos = torch.from_numpy(synth.frame(n, box, k, sigma=0.08)).cuda()
        frames.append((pos, box, rc))
        wn = e1.search_count(a.SEARCH_SINGLE, rc, pos, box=box, pbc=7)
        want.append((wn,) + e1.search_fill(wn))
    descs = [e2.make_search_desc(a.SEARCH_SINGLE, rc, pos, box=box, pbc=7) for pos, box, rc in frames]

    def check(k, res):
        cnt, pp, dp = res
        wn, wp, wd = want[k]
        assert cnt == wn > 0
        got_pairs = torch.as_tensor(Dev(pp, cnt * 2, "<i4"), device="cuda").cpu().numpy().view(np.uint32).reshape(-1, 2)
        got_dist = torch.as_tensor(Dev(dp, cnt, "<f4"), device="cuda").cpu().numpy()
        assert np.array_equal(got_pairs, wp) and np.array_equal(got_dist, wd), f"frame {k}"

    for rounds in range(2):          # second round: every buffer already large enough (no repeats)
        prev = None
        for k in range(len(frames)):
            t = e2.search_resident_begin(descs[k][0])
            if prev is not None:
                check(prev[0], e2.search_resident_end(prev[1]))
            prev = (k, t)
        check(prev[0], e2.search_resident_end(prev[1]))
    # both tickets out: a third begin is refused, so is ending a ticket twice; the plain call still works afterwards
    t0 = e2.search_resident_begin(descs[0][0])
    t1 = e2.search_resident_begin(descs[2][0])
    assert {t0, t1} == {0, 1}
    with pytest.raises(MolarHipError):
        e2.search_resident_begin(descs[3][0])
    check(2, e2.search_resident_end(t1))          # out of order is fine
    check(0, e2.search_resident_end(t0))
    with pytest.raises(MolarHipError):
        e2.search_resident_end(t0)
    check(3, e2.search_resident_desc(descs[3][0]))


def test_pipelined_large_frames_privatised_binning():
    """Frames of >= 2^19 atoms whose grid is built on the side stream are binned with counters privatised in LDS
    (bin_tile_kernel): three 600k-atom frames through begin / end, two in flight, against count + fill on another context
    (which bins with one atomic per atom on its main stream); order-sensitive comparison on the device."""
    import torch
    a = api()
    from molar_amd.api import Engine
    e1, e2 = Engine(0), Engine(0)
    n = 600_000
    box = synth.box_a(n)
    frames = [torch.from_numpy(synth.frame(n, box, k)).cuda() for k in range(3)]
    descs = [e2.make_search_desc(a.SEARCH_SINGLE, 1.0, f, box=box, pbc=7) for f in frames]

    def view(pp, dp, cnt):
        return a.device_view(pp, (cnt, 2), torch.int32), a.device_view(dp, (cnt,), torch.float32)

    def check(k, res):
        cnt, pp, dp = res
        wn = e1.search_count(a.SEARCH_SINGLE, 1.0, frames[k], box=box, pbc=7)
        assert cnt == wn > 0
        wp, wd = e1.search_fill_device()
        e1.synchronize()              # the fill is only enqueued on e1's stream
        gp, gd = view(pp, dp, cnt)
        assert torch.equal(gp, a.device_view(wp, (cnt, 2), torch.int32)) and torch.equal(gd, a.device_view(wd, (cnt,), torch.float32)), f"frame {k}"

    prev = None
    for k in range(3):
        t = e2.search_resident_begin(descs[k][0])
        if prev is not None:
            check(prev[0], e2.search_resident_end(prev[1]))
        prev = (k, t)
    check(prev[0], e2.search_resident_end(prev[1]))


def test_resident_launch_follows_the_previous_plan_and_repeats_when_short():
    """Resident searches launch their passes over the slots the plan of the search before came to (+ 3 % + 512) instead of the
    host's bound; a frame that needs more slots than that - same box, same grid, twice the atoms - is repeated over the bound
    where its sizes are read (molar_hip_search_resident, and _end with a younger search in flight).  Every result against
    count + fill on another context, order-sensitive."""
    import torch
    a = api()
    from molar_amd.api import Engine
    e1, e2 = Engine(0), Engine(0)
    nA, nB = 120_000, 260_000
    box = synth.box_a(nB)
    rng = np.random.default_rng(41)
    fa = torch.from_numpy((rng.random((nA, 3)) @ box.astype(np.float64).T).astype(np.float32)).cuda()
    fb = torch.from_numpy((rng.random((nB, 3)) @ box.astype(np.float64).T).astype(np.float32)).cuda()
    frames = {"A": fa, "B": fb}
    descs = {k: e2.make_search_desc(a.SEARCH_SINGLE, 1.0, f, box=box, pbc=7) for k, f in frames.items()}

    def check(k, res):
        cnt, pp, dp = res
        wn = e1.search_count(a.SEARCH_SINGLE, 1.0, frames[k], box=box, pbc=7)
        assert cnt == wn > 0, k
        wp, wd = e1.search_fill_device()
        e1.synchronize()
        assert torch.equal(a.device_view(pp, (cnt, 2), torch.int32), a.device_view(wp, (cnt, 2), torch.int32)), k
        assert torch.equal(a.device_view(dp, (cnt,), torch.float32), a.device_view(wd, (cnt,), torch.float32)), k

    for k in "ABABBA":                       # B after A needs 2.2 x the slots A's plan came to
        check(k, e2.search_resident_desc(descs[k][0]))
    order = "AABBABAAB"
    prev = None
    for k in order:
        t = e2.search_resident_begin(descs[k][0])
        if prev is not None:
            check(prev[0], e2.search_resident_end(prev[1]))
        prev = (k, t)
    check(prev[0], e2.search_resident_end(prev[1]))


def test_randomised_differential(monkeypatch):
    """tools/fuzz_search.py: random boxes / cutoffs / densities / periodicity masks / selections / kinds, count+fill and
    resident entries, against the oracle - every case bit-identical (9000 cases were run this way in round 1)."""
    import importlib.util
    import os
    import sys
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_search.py")
    spec = importlib.util.spec_from_file_location("fuzz_search", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    monkeypatch.setattr(sys, "argv", ["fuzz_search.py", "250", "11"])
    assert mod.main() == 0


def test_randomised_pipeline():
    """tools/fuzz_pipeline.py: random frames through molar_hip_search_resident_begin/_end with two searches in flight
    (two grid generations, grid build on the side stream, grow-and-repeat), each result equal to count + fill."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_pipeline.py")
    spec = importlib.util.spec_from_file_location("fuzz_pipeline", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(150, 5, verbose=False) == 0


def test_histogram_device_bins_async(eng):
    """Device-resident bins, no per-frame round trip: several frames queued back to back give the same integer bins
    as the host-bins path."""
    import torch
    a = api()
    n = 20000
    box = synth.box_a(n)
    frames = [synth.frame(n, box, f) for f in range(4)]
    hb = np.zeros(300, np.uint64)
    tot = 0
    for f in frames:
        hb, c = eng.search_histogram(a.SEARCH_SINGLE, 0.9, 0.0, 0.9, 300, f, box=box, pbc=7, bins=hb)
        tot += c
    db = torch.zeros(300, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    dev_frames = [torch.from_numpy(f).cuda() for f in frames]
    torch.cuda.synchronize()
    for f in dev_frames[:-1]:
        _, c = eng.search_histogram(a.SEARCH_SINGLE, 0.9, 0.0, 0.9, 300, f, box=box, pbc=7, bins=db, want_count=False)
        assert c is None
    _, c_last = eng.search_histogram(a.SEARCH_SINGLE, 0.9, 0.0, 0.9, 300, dev_frames[-1], box=box, pbc=7, bins=db)
    eng.synchronize()
    assert np.array_equal(db.cpu().numpy().astype(np.uint64), hb)
    assert 0 < int(hb.sum()) <= tot and c_last > 0          # a distance equal to the upper edge falls outside the last bin


def test_unwrap_connectivity_100k_atoms_against_the_oracle_entry(eng, orc32):
    """molar_hip_unwrap_connectivity (GPU search + host walk in the library) against the oracle's
    orc_unwrap_connectivity_dim (modify.rs:72-131): 2000 molecules of 60 atoms, wrapped into a triclinic box so that
    hundreds of them are split over a boundary; coordinates bit-identical, group lists equal; also through a selection
    and with the frame resident on the device."""
    import torch
    a = api()
    rng = np.random.default_rng(11)
    # helices of 60 atoms (radius 0.3, 27 degrees and 0.05 nm rise per atom: neighbours 0.15 nm apart, everything else > 0.17)
    # on a 13 x 13 x 12 lattice, shifted as a whole so that the box faces cut through hundreds of them
    nx, ny, nz, length = 13, 13, 12, 60
    nmol = nx * ny * nz - 28
    box = np.diag([nx * 2.4, ny * 2.4, nz * 3.6]).astype(np.float32)
    box[0, 2] = -2.0; box[1, 2] = -1.5
    k = np.arange(length)
    th = np.deg2rad(27.0) * k
    helix = np.stack([0.3 * np.cos(th), 0.3 * np.sin(th), 0.05 * k], 1)
    parts = []
    for m in range(nmol):
        cx, cy, cz = m % nx, (m // nx) % ny, m // (nx * ny)
        rot = rng.uniform(0, 2 * np.pi)
        R = np.array([[np.cos(rot), -np.sin(rot), 0], [np.sin(rot), np.cos(rot), 0], [0, 0, 1.0]])
        parts.append(helix @ R.T + np.array([cx * 2.4 + 1.2, cy * 2.4 + 1.2, cz * 3.6 + 0.3]) + rng.normal(0, 0.005, (length, 3)))
    whole = np.concatenate(parts) + np.array([0.9, 1.1, 1.7])
    fr = whole @ np.linalg.inv(box.astype(np.float64)).T
    wrapped = ((fr - np.floor(fr)) @ box.astype(np.float64).T).astype(np.float32)
    n = len(wrapped)
    assert n >= 100_000
    ob = orc32.box_from_matrix(box)
    import os
    ref, rgroups = orc32.unwrap_connectivity(wrapped, ob, 0.17, 7, nthreads=os.cpu_count() or 4)
    assert len(rgroups) >= 1000
    split = sum(1 for m in range(nmol) if np.abs(np.diff(wrapped[m * length:(m + 1) * length], axis=0)).max() > 1.0)
    assert split > 200
    got = wrapped.copy()
    groups = eng.unwrap_connectivity(got, box, 0.17, 7)
    assert np.array_equal(got, ref)
    assert len(groups) == len(rgroups) and all(np.array_equal(x, y) for x, y in zip(groups, rgroups))
    # device-resident frame
    dgot = torch.from_numpy(wrapped.copy()).cuda()
    torch.cuda.synchronize()
    g2 = eng.unwrap_connectivity(dgot, box, 0.17, 7)
    assert np.array_equal(dgot.cpu().numpy(), ref) and len(g2) == len(rgroups)
    # through a selection (every other molecule), partial dims
    idx = np.concatenate([np.arange(m * length, (m + 1) * length) for m in range(0, nmol, 2)]).astype(np.uint64)
    ref3, rg3 = orc32.unwrap_connectivity(wrapped, ob, 0.17, 3, idx=idx, nthreads=os.cpu_count() or 4)
    got3 = wrapped.copy()
    g3 = eng.unwrap_connectivity(got3, box, 0.17, 3, idx=idx)
    assert np.array_equal(got3, ref3) and len(g3) == len(rg3) and all(np.array_equal(x, y) for x, y in zip(g3, rg3))
    # no box: the reference's require_box error
    with pytest.raises(a.MolarHipError):
        eng.unwrap_connectivity(wrapped.copy(), None, 0.17, 7)


@pytest.mark.parametrize("boxfn,n,cutoff,local", [(synth.box_a, 30000, 0.25, True), (synth.box_b, 12000, 0.3, True),
                                                 (synth.box_ortho, 20000, 0.2, False), (synth.box_a, 3000, 1.0, True)])
def test_search_connectivity_csr_in_push_order(eng, orc32, boxfn, n, cutoff, local):
    """SearchConnectivity::from_iter (connectivity.rs:19-35) on the device: conn[i].push(j); conn[j].push(i) in pair order.
    The CSR must hold exactly those lists, element by element - including the dense case (cutoff 1.0: lists of hundreds)."""
    box = boxfn(n)
    pos = synth.frame(n, box)
    idx = np.sort(np.random.default_rng(n).choice(n, n * 3 // 4, replace=False)).astype(np.uint64)
    ob = orc32.box_from_matrix(box)
    ref = orc32.search_single_pbc(cutoff, pos[idx.astype(int)], ob, 7, ids=None if local else idx, nthreads=8)
    rows = len(idx) if local else n
    lists = [[] for _ in range(rows)]
    for i, j in zip(ref["i"].tolist(), ref["j"].tolist()):
        lists[i].append(j)
        lists[j].append(i)
    off, nb = eng.search_connectivity(cutoff, pos, idx, box=box, pbc=7, ids_local=local)
    assert len(off) == rows + 1 and off[0] == 0 and off[-1] == 2 * len(ref["i"]) == len(nb)
    want_off = np.concatenate([[0], np.cumsum([len(l) for l in lists])]).astype(np.uint64)
    assert np.array_equal(off, want_off)
    assert np.array_equal(nb, np.fromiter((v for l in lists for v in l), np.uint64, count=len(nb)))


def test_resident_search_pairs_plane_only(eng, orc32):
    """molar_hip_search_resident_planes(ctx, 0): the DistanceSearchOutput of (usize, usize) (distance_search.rs:14-20) - the
    resident searches (plain and begin / end) fill the (i, j) plane only; it must be the full mode's pair plane bit for bit,
    the distance address comes back NULL, and switching back restores the distances."""
    import torch
    a = api()
    n = 40000
    box = synth.box_a(n)
    pos = torch.from_numpy(synth.frame(n, box)).cuda()
    pos2 = torch.from_numpy(synth.frame(n, box, 1)).cuda()
    torch.cuda.synchronize()
    e2 = a.Engine(0)
    ref = orc32.search_single_pbc(0.9, pos.cpu().numpy(), orc32.box_from_matrix(box), 7, nthreads=8)

    def planes(cnt, pp, dp):
        class _Dev:
            def __init__(self, ptr, m, t):
                self.__cuda_array_interface__ = {"shape": (m,), "typestr": t, "data": (ptr, False), "version": 2}
        pr = torch.as_tensor(_Dev(pp, 2 * cnt, "<u4"), device="cuda").cpu().numpy().reshape(-1, 2)
        d = torch.as_tensor(_Dev(dp, cnt, "<f4"), device="cuda").cpu().numpy() if dp else None
        return pr, d

    e2.search_resident_planes(False)
    cnt, pp, dp = e2.search_resident(a.SEARCH_SINGLE, 0.9, pos, box=box, pbc=7)
    assert cnt == len(ref["i"]) and dp is None
    pr, _ = planes(cnt, pp, None)
    assert np.array_equal(pr[:, 0], ref["i"]) and np.array_equal(pr[:, 1], ref["j"])
    # two frames in flight
    d1 = e2.make_search_desc(a.SEARCH_SINGLE, 0.9, pos, box=box, pbc=7)
    d2 = e2.make_search_desc(a.SEARCH_SINGLE, 0.9, pos2, box=box, pbc=7)
    t1 = e2.search_resident_begin(d1[0])
    t2 = e2.search_resident_begin(d2[0])
    with pytest.raises(Exception):
        e2.search_resident_planes(True)            # not while searches are in flight
    c1, p1, q1 = e2.search_resident_end(t1)
    c2, p2, q2 = e2.search_resident_end(t2)
    assert q1 is None and q2 is None and c1 == cnt
    assert np.array_equal(planes(c1, p1, None)[0], pr)
    ref2 = orc32.search_single_pbc(0.9, pos2.cpu().numpy(), orc32.box_from_matrix(box), 7, nthreads=8)
    pr2, _ = planes(c2, p2, None)
    assert c2 == len(ref2["i"]) and np.array_equal(pr2[:, 0], ref2["i"]) and np.array_equal(pr2[:, 1], ref2["j"])
    e2.search_resident_planes(True)
    cnt3, pp3, dp3 = e2.search_resident(a.SEARCH_SINGLE, 0.9, pos, box=box, pbc=7)
    pr3, d3 = planes(cnt3, pp3, dp3)
    assert np.array_equal(pr3, pr) and np.array_equal(d3, ref["d"])


def test_fill_into_misaligned_device_views(eng):
    """The fill pass writes two results per lane and store instruction (dwordx4 / dwordx2 relative to the output bases), so it
    wants 16-byte (pairs) / 8-byte (distances) alignment.  A caller's DEVICE view that is not aligned like that - an offset
    slice of a larger tensor - is filled through the context's own buffers and copied device to device: the same result
    (it used to be refused with MOLAR_HIP_ERR_INVALID_ARGUMENT)."""
    import ctypes as C
    import torch
    a = api()
    n = 20000
    box = synth.box_a(n)
    pos = synth.frame(n, box)
    cnt = eng.search_count(a.SEARCH_SINGLE, 0.6, pos, box=box, pbc=7)
    assert cnt > 0
    pr, d = eng.search_fill(cnt)
    for poff, doff in ((8, 0), (0, 4), (8, 4), (0, 0)):      # bytes past an aligned base
        pbuf = torch.full((2 * cnt + 8,), -1, dtype=torch.int32, device="cuda")
        dbuf = torch.full((cnt + 8,), -1.0, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        a.check(eng.lib.molar_hip_search_fill(eng.ctx, C.c_void_p(pbuf.data_ptr() + poff), C.c_void_p(dbuf.data_ptr() + doff)))
        eng.synchronize()
        got_p = pbuf[poff // 4: poff // 4 + 2 * cnt].cpu().numpy().view(np.uint32).reshape(-1, 2)
        got_d = dbuf[doff // 4: doff // 4 + cnt].cpu().numpy()
        assert np.array_equal(got_p, pr) and np.array_equal(got_d, d), (poff, doff)
        # nothing outside the view was touched
        assert int(pbuf[poff // 4 + 2 * cnt:].max()) == -1 and float(dbuf[doff // 4 + cnt:].max()) == -1.0
        if poff:
            assert int(pbuf[:poff // 4].max()) == -1
        if doff:
            assert float(dbuf[:doff // 4].max()) == -1.0


def test_fused_histogram_call_forms_interleaved(eng, orc32):
    """The one-kernel plan of the fused histogram alternates between two pairs of list counters from frame to frame (the
    kernel of frame k zeroes the pair frame k + 1 will use).  Queued calls with device bins, waited calls with host bins, a
    vdW histogram (regular plan, no lists) and systems of different size in between must all leave the counters right:
    every call's bins equal the oracle's."""
    import torch
    a = api()
    e2 = a.Engine(0)
    rng = np.random.default_rng(77)
    systems = []
    for n, cutoff in ((3000, 0.7), (40000, 0.9), (200, 0.5), (12000, 1.1)):
        box = synth.box_a(n)
        pos = synth.frame(n, box, n % 5)
        ref = orc32.search_single_pbc(cutoff, pos, orc32.box_from_matrix(box), 7, nthreads=8)
        systems.append((n, cutoff, box, pos, torch.from_numpy(pos).cuda(), orc32.histogram_add(0.0, cutoff, 300, ref["d"]).astype(np.int64)))
    torch.cuda.synchronize()
    dev_bins = [torch.zeros(300, dtype=torch.int64, device="cuda") for _ in systems]
    times = [0] * len(systems)
    torch.cuda.synchronize()
    order = [0, 1, 1, 2, 0, 3, 3, 3, 1, 2, 2, 0, 1, 3, 0]
    for step, k in enumerate(order):
        n, cutoff, box, pos, dpos, want = systems[k]
        if step % 4 == 3:          # a waited call with host bins in between the queued ones
            bins, cnt = e2.search_histogram(a.SEARCH_SINGLE, cutoff, 0.0, cutoff, 300, pos, box=box, pbc=7)
            assert np.array_equal(bins.astype(np.int64), want) and cnt >= want.sum()
        else:
            e2.search_histogram(a.SEARCH_SINGLE, cutoff, 0.0, cutoff, 300, dpos, box=box, pbc=7, bins=dev_bins[k], want_count=False)
            times[k] += 1
        if step == 6:              # a histogram of another kind: the regular plan, no slot lists
            m = 2000
            bx = synth.box_ortho(m)
            p = synth.frame(m, bx)
            i1, i2 = np.arange(0, m // 2, dtype=np.uint64), np.arange(m // 2, m, dtype=np.uint64)
            vd = rng.uniform(0.1, 0.2, m).astype(np.float32)
            r = orc32.search_double_vdw_pbc(p[i1.astype(int)], p[i2.astype(int)], vd[i1.astype(int)], vd[i2.astype(int)], orc32.box_from_matrix(bx), 7, nthreads=4)
            hb, hc = e2.search_histogram(a.SEARCH_DOUBLE_VDW, None, 0.0, 0.5, 64, p, i1, p, i2, box=bx, pbc=7, vdw1=vd[i1.astype(int)], vdw2=vd[i2.astype(int)])
            assert hc == len(r["i"]) and np.array_equal(hb, orc32.histogram_add(0.0, 0.5, 64, r["d"]).astype(np.uint64))
    e2.synchronize()
    torch.cuda.synchronize()
    for k, (n, cutoff, box, pos, dpos, want) in enumerate(systems):
        assert np.array_equal(dev_bins[k].cpu().numpy(), want * times[k]), (k, times[k])


@pytest.mark.parametrize("boxkind,n,cutoff,nframes,pbc,strided", [
    ("a", 30000, 0.9, 11, 7, False),        # triclinic box A: corner entries through the generic kernel's joint list; 8 + 3 frames
    ("ortho", 20000, 0.8, 8, 7, False),
    ("a", 140000, 1.0, 5, 7, False),        # >= 2^17 atoms: the tiled binning kernel
    ("b", 9000, 0.7, 17, 7, True),          # sheared box, an index, frames with a gap between them; 8 + 8 + 1 (a group of one: single call)
    ("ortho", 12000, 0.9, 6, 5, False),     # y not periodic: drop rule, fewer wrapped entries
    ("npt", 25000, 0.85, 9, 7, False),      # a box per frame (same grid), one frame whose box gives ANOTHER grid: that group goes frame by frame
])
def test_fused_histogram_frames_equals_single_calls(eng, orc32, boxkind, n, cutoff, nframes, pbc, strided):
    """molar_hip_search_histogram_frames: a block of a trajectory through the fused histogram in groups of up to eight frames
    that share their launches (grids of a group built by frame-indexed kernels, one plan launch, one persistent kernel over
    the joint slot list with the frame's number in every record).  The bins are integers: they must equal those of one
    molar_hip_search_histogram call per frame exactly, and the oracle's distance stream binned by Histogram1D::add_one
    (molar_membrane/src/stats.rs:29-35).  Mixed with single-frame calls on the same context (the list counters rotate)."""
    import torch
    a = api()
    e2 = a.Engine(0)
    nbins = 400
    base_box = {"a": synth.box_a, "b": synth.box_b, "ortho": synth.box_ortho, "npt": synth.box_a}[boxkind](n)
    boxes = np.repeat(base_box[None], nframes, axis=0).astype(np.float32)
    if boxkind == "npt":
        for f in range(nframes):
            boxes[f] *= np.float32(1.0 + 0.002 * np.sin(f))
        boxes[4] *= np.float32(1.15)        # another grid in the middle of the first group
    frames_np = np.stack([synth.frame(n, boxes[f], f) for f in range(nframes)])
    idx = np.sort(np.random.default_rng(5).choice(n, n - n // 7, replace=False)).astype(np.uint64) if strided else None
    if strided:
        store = torch.zeros((nframes, n + 37, 3), dtype=torch.float32, device="cuda")
        store[:, :n] = torch.from_numpy(frames_np).cuda()
        dframes = store[:, :n]              # stride (n + 37) * 3 floats between frames
    else:
        dframes = torch.from_numpy(frames_np).cuda()
    didx = None if idx is None else torch.from_numpy(idx.astype(np.int64)).cuda()
    bins_f = torch.zeros(nbins, dtype=torch.int64, device="cuda")
    bins_s = torch.zeros(nbins, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    want = np.zeros(nbins, np.int64)
    for f in range(nframes):
        p = frames_np[f] if idx is None else frames_np[f][idx.astype(int)]
        ref = orc32.search_single_pbc(cutoff, p, orc32.box_from_matrix(boxes[f]), pbc, nthreads=8)
        want += orc32.histogram_add(0.0, cutoff, nbins, ref["d"]).astype(np.int64)
    box_arg = boxes if boxkind == "npt" else base_box
    for rep in range(3):                    # three blocks back to back: both generations of the group buffers, all four list slots
        e2.search_histogram_frames(a.SEARCH_SINGLE, cutoff, 0.0, cutoff, nbins, dframes, idx1=didx, box=box_arg, pbc=pbc, bins=bins_f)
        if rep == 1:                        # a queued single-frame call in between
            e2.search_histogram(a.SEARCH_SINGLE, cutoff, 0.0, cutoff, nbins, dframes[0], idx1=didx, box=boxes[0], pbc=pbc, bins=bins_s, want_count=False)
    e2.synchronize()
    got = bins_f.cpu().numpy()
    assert want.sum() > 0
    assert np.array_equal(got, 3 * want), (int(got.sum()), int(3 * want.sum()))
    one = bins_s.cpu().numpy().copy()
    for f in range(1, nframes):
        e2.search_histogram(a.SEARCH_SINGLE, cutoff, 0.0, cutoff, nbins, dframes[f], idx1=didx, box=boxes[f], pbc=pbc, bins=bins_s, want_count=False)
    e2.synchronize()
    assert np.array_equal(bins_s.cpu().numpy(), want)
    assert one.sum() > 0


@pytest.mark.parametrize("boxkind,n,cutoff,nframes,pbc,own_frames", [
    ("a", 24000, 0.9, 9, 7, False),         # two selections of one trajectory (ions / water): same frames, two indices
    ("ortho", 16000, 0.8, 5, 7, True),      # the second set has frames of its own (another stride), no index on it
    ("b", 150000, 0.9, 4, 7, False),        # a first set of >= 2^17 atoms (tiled binning) against a small second one
    ("ortho", 10000, 0.9, 6, 3, False),     # z not periodic
])
def test_fused_histogram_frames_two_sets(eng, orc32, boxkind, n, cutoff, nframes, pbc, own_frames):
    """The frames form for distance_search_double_pbc (a radial distribution between two selections): groups of frames share
    their launches, both sets' grids built by the frame-indexed kernels; integer bins equal to one call per frame and to the
    oracle's distance stream, same-cell duplicates included (overlapping selections)."""
    import torch
    a = api()
    e2 = a.Engine(0)
    nbins = 300
    box = {"a": synth.box_a, "b": synth.box_b, "ortho": synth.box_ortho}[boxkind](n)
    frames_np = np.stack([synth.frame(n, box, 40 + f) for f in range(nframes)])
    rng = np.random.default_rng(11)
    n1 = n - n // 5 if boxkind == "b" else n // 3
    idx1 = np.sort(rng.choice(n, n1, replace=False)).astype(np.uint64)
    dframes = torch.from_numpy(frames_np).cuda()
    if own_frames:
        m = n // 2
        f2_np = np.stack([synth.frame(m, box, 90 + f) for f in range(nframes)])
        store = torch.zeros((nframes, m + 11, 3), dtype=torch.float32, device="cuda")
        store[:, :m] = torch.from_numpy(f2_np).cuda()
        d2, idx2 = store[:, :m], None
    else:
        f2_np = frames_np
        n2 = n // 40 if boxkind == "b" else n // 4
        idx2 = np.sort(rng.choice(n, n2, replace=False)).astype(np.uint64)      # overlaps idx1: pairs of an atom with itself at d = 0
        d2 = None
    didx1 = torch.from_numpy(idx1.astype(np.int64)).cuda()
    didx2 = None if idx2 is None else torch.from_numpy(idx2.astype(np.int64)).cuda()
    want = np.zeros(nbins, np.int64)
    ob = orc32.box_from_matrix(box)
    for f in range(nframes):
        p1 = frames_np[f][idx1.astype(int)]
        p2 = f2_np[f] if idx2 is None else f2_np[f][idx2.astype(int)]
        ref = orc32.search_double_pbc(cutoff, p1, p2, ob, pbc, nthreads=8)
        want += orc32.histogram_add(0.0, cutoff, nbins, ref["d"]).astype(np.int64)
    bins_f = torch.zeros(nbins, dtype=torch.int64, device="cuda")
    bins_s = torch.zeros(nbins, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    for rep in range(2):
        e2.search_histogram_frames(a.SEARCH_DOUBLE, cutoff, 0.0, cutoff, nbins, dframes, idx1=didx1, box=box, pbc=pbc, bins=bins_f, frames2=d2, idx2=didx2)
    e2.search_histogram_frames(a.SEARCH_SINGLE, cutoff, 0.0, cutoff, nbins, dframes[:3], idx1=didx1, box=box, pbc=pbc, bins=bins_s)   # the other kind in between
    e2.search_histogram_frames(a.SEARCH_DOUBLE, cutoff, 0.0, cutoff, nbins, dframes, idx1=didx1, box=box, pbc=pbc, bins=bins_f, frames2=d2, idx2=didx2)
    e2.synchronize()
    assert want.sum() > 0
    got = bins_f.cpu().numpy()
    assert np.array_equal(got, 3 * want), (int(got.sum()), int(3 * want.sum()))
    bins_1 = torch.zeros(nbins, dtype=torch.int64, device="cuda")
    for f in range(nframes):
        e2.search_histogram(a.SEARCH_DOUBLE, cutoff, 0.0, cutoff, nbins, dframes[f], didx1, dframes[f] if d2 is None else d2[f], didx2, box=box, pbc=pbc,
                            bins=bins_1, want_count=False)
    e2.synchronize()
    assert np.array_equal(bins_1.cpu().numpy(), want)


def test_fused_histogram_frames_host_inputs(eng, orc32):
    """Frames in host memory (or host bins) are walked frame by frame: the same sums."""
    a = api()
    n, cutoff, nbins, nframes = 6000, 0.8, 200, 3
    box = synth.box_a(n)
    frames_np = np.stack([synth.frame(n, box, f) for f in range(nframes)])
    want = np.zeros(nbins, np.int64)
    for f in range(nframes):
        ref = orc32.search_single_pbc(cutoff, frames_np[f], orc32.box_from_matrix(box), 7, nthreads=4)
        want += orc32.histogram_add(0.0, cutoff, nbins, ref["d"]).astype(np.int64)
    bins = eng.search_histogram_frames(a.SEARCH_SINGLE, cutoff, 0.0, cutoff, nbins, frames_np, box=box, pbc=7)
    assert np.array_equal(bins.astype(np.int64), want)
    # two sets from host memory: the second set a trajectory of its own
    m = 2500
    f2 = np.stack([synth.frame(m, box, 70 + f) for f in range(nframes)])
    idx1 = np.arange(0, n, 3, dtype=np.uint64)
    want2 = np.zeros(nbins, np.int64)
    for f in range(nframes):
        ref = orc32.search_double_pbc(cutoff, frames_np[f][idx1.astype(int)], f2[f], orc32.box_from_matrix(box), 7, nthreads=4)
        want2 += orc32.histogram_add(0.0, cutoff, nbins, ref["d"]).astype(np.int64)
    bins2 = eng.search_histogram_frames(a.SEARCH_DOUBLE, cutoff, 0.0, cutoff, nbins, frames_np, idx1=idx1, box=box, pbc=7, frames2=f2)
    assert want2.sum() > 0 and np.array_equal(bins2.astype(np.int64), want2)
