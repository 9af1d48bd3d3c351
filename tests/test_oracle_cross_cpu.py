"""Two independent restatements of the reference's distance search must agree: oracle/molar_oracle.c (the checker of the
GPU path) against oracle/ref_search.py (a literal Python re-reading of distance_search.rs / periodic_box.rs).  Brute
force can only witness boxes where the reference's half-shell grid is geometrically complete; GROMACS-style and strongly
sheared boxes, one- and two-cell grids, partial periodicity and atoms outside the cell are exactly where it cannot, and
where this test compares the two restatements element by element: ids, order and distances."""
import numpy as np
import pytest

from oracle import ref_search as R


def boxes(rng):
    L = rng.uniform(1.2, 4.0, 3)
    kinds = []
    kinds.append(np.diag(L))
    m = np.diag(L); m[0, 2] = -rng.uniform(0, 0.3) * L[0]; m[1, 2] = -rng.uniform(0, 0.3) * L[1]; kinds.append(m)           # benign shear
    m = np.diag(L); m[0, 1] = rng.uniform(-0.5, 0.5) * L[0]; m[0, 2] = rng.uniform(-0.5, 0.5) * L[0]; m[1, 2] = rng.uniform(-0.5, 0.5) * L[1]; kinds.append(m)   # GROMACS-style
    kinds.append(np.diag(L) + rng.uniform(-0.3, 0.3, (3, 3)) * L.min())                                                     # general
    d = L[0]; kinds.append(np.array([[d, 0, d / 2], [0, d, d / 2], [0, 0, d * np.sqrt(2) / 2]]))                             # rhombic dodecahedron
    kinds.append(np.diag(rng.uniform(0.6, 1.3, 3)))                                                                          # 1-2 cells per dimension
    return [k.astype(np.float32) for k in kinds]


def same(ref, got, within=False):
    if within:
        return [int(x) for x in ref["i"]] == list(got)
    if len(ref["i"]) != len(got):
        return False
    gi = np.array([g[0] for g in got], np.uint64); gj = np.array([g[1] for g in got], np.uint64)
    gd = np.array([g[2] for g in got], np.float32)
    return np.array_equal(ref["i"], gi) and np.array_equal(ref["j"], gj) and np.array_equal(ref["d"], gd)


@pytest.mark.parametrize("seed", range(3))
def test_c_oracle_equals_python_restatement(orc32, seed):
    rng = np.random.default_rng(100 + seed)
    checked = 0
    for box in boxes(rng):
        n = int(rng.integers(40, 160))
        pos = (rng.random((n, 3)) @ box.astype(np.float64).T + rng.normal(0, rng.choice([0.0, 0.05, 0.4]), (n, 3))).astype(np.float32)
        rc = float(np.float32(rng.uniform(0.3, 0.9)))
        ob = orc32.box_from_matrix(box)
        rb = R.Box(box)
        assert np.array_equal(np.array(rb.inv, np.float32), orc32.box_inv(ob)) and len(rb.shifts) == len(orc32.box_shifts(ob))
        perm = rng.permutation(n)
        i1 = np.sort(perm[: n // 3]).astype(np.uint64); i2 = np.sort(perm[n // 3:]).astype(np.uint64)
        p1, p2 = pos[i1.astype(int)], pos[i2.astype(int)]
        v1 = rng.uniform(0.1, 0.3, len(i1)).astype(np.float32); v2 = rng.uniform(0.1, 0.3, len(i2)).astype(np.float32)
        for dims in (7, 3, 5, 6, 1):
            ref = orc32.search_single_pbc(rc, pos, ob, dims)
            got, gd = R.single(rc, pos, None, rb, dims)
            assert tuple(ref["dims"]) == tuple(gd) and same(ref, got), (seed, box.tolist(), dims)
            ref = orc32.search_double_pbc(rc, p1, p2, ob, dims, ids1=i1, ids2=i2)
            got, _ = R.double(rc, p1, p2, i1, i2, rb, dims)
            assert same(ref, got), ("double", seed, box.tolist(), dims)
            ref = orc32.search_within_pbc(rc, p1, p2, ob, dims, i1, i2)
            got, _ = R.double(rc, p1, p2, i1, i2, rb, dims, within=True)
            assert same(ref, got, within=True), ("within", seed, box.tolist(), dims)
            ref = orc32.search_double_vdw_pbc(p1, p2, v1, v2, ob, dims)
            got, _ = R.double(None, p1, p2, None, None, rb, dims, vdw=(v1, v2))
            assert same(ref, got), ("vdw", seed, box.tolist(), dims)
            checked += 4
        # the non-periodic drivers (zero-seeded bounding box, drop rule)
        shifted = (pos + np.float32(rng.uniform(-3, 3))).astype(np.float32)
        ref = orc32.search_single(rc, shifted)
        got, gd = R.single(rc, shifted)
        assert tuple(ref["dims"]) == tuple(gd) and same(ref, got)
        s1, s2 = shifted[i1.astype(int)], shifted[i2.astype(int)]
        assert same(orc32.search_double(rc, s1, s2, ids1=i1, ids2=i2), R.double(rc, s1, s2, i1, i2)[0])
        assert same(orc32.search_double_vdw(s1, s2, v1, v2), R.double(None, s1, s2, vdw=(v1, v2))[0])
        lo, up = orc32.min_max(s1)
        lo = lo + (np.float32(-rc) - np.float32(1.1920929e-07)); up = up + (np.float32(rc) + np.float32(1.1920929e-07))
        assert same(orc32.search_within(rc, s1, s2, lo, up, i1, i2), R.double(rc, s1, s2, i1, i2, within=True, lower=lo, upper=up)[0], within=True)
        checked += 4
    assert checked == 6 * 24


def test_lipid_tail_order_second_restatement(orc64):
    """Measure::lipid_tail_order (measure.rs:270-422): the C oracle (f64 build) against a separate numpy re-reading, on
    random tails with one normal or one per bond and double bonds at every legal position."""
    from oracle import ref_measure as M
    rng = np.random.default_rng(17)
    for case in range(300):
        n = int(rng.integers(3, 24))
        p = np.cumsum(rng.normal(0, 0.1, (n, 3)), axis=0) + rng.uniform(-5, 5, 3)
        order_type = case % 3
        per_bond = bool(rng.integers(0, 2))
        bo = np.ones(n - 1, np.uint8)
        if order_type:
            for _ in range(int(rng.integers(0, 3))):
                hi = n - 3 if per_bond else n - 2
                if hi > 1:
                    b = int(rng.integers(1, hi))
                    if bo[b - 1] == 1 and (b + 1 >= n - 1 or bo[b + 1] == 1):
                        bo[b] = 2
        nn = rng.normal(size=(n - 2 if per_bond else 1, 3)); nn /= np.linalg.norm(nn, axis=1)[:, None]
        want = M.lipid_tail_order(p, order_type, nn, bo)
        got = orc64.lipid_tail_order(p, order_type, nn, bo)
        assert np.allclose(got, want, atol=1e-9, equal_nan=True), (case, order_type, bo)
