"""The two host-arithmetic entry points of the membrane path (no GPU needed): patch lists from the marker pairs in push
order (molar_membrane/src/lib.rs:548-557) and compute_initial_normals (:456-505, second pass in place in lipid order)
against plain restatements of the reference loops.  Also what tools/asan_host.sh runs under ASan + UBSan."""
import numpy as np
import pytest

from molar_amd import api
from molar_amd._lib import MolarHipError


def patches_loop(pairs, K):
    lists = [[] for _ in range(K)]
    for i, j in pairs:
        lists[i].append(j); lists[j].append(i)
    off = np.concatenate([[0], np.cumsum([len(l) for l in lists])]).astype(np.uint64)
    return off, np.array([x for l in lists for x in l], np.uint64)


def normals_loop(head, tail, lists, valid):
    """lib.rs:456-505 with every operation rounded to f32; pass 2 reads the normals pass 2 has already written"""
    f = np.float32
    K = len(head)

    def norm(v):
        return f(np.sqrt(f(f(v[0] * v[0] + v[1] * v[1]) + v[2] * v[2])))

    def within(a, b):
        n1, n2 = norm(a), norm(b)
        if n1 == 0 or n2 == 0:
            return True
        c = f(f(f(a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]) / f(n1 * n2))
        return bool(f(np.arccos(np.clip(c, f(-1), f(1)))) <= f(np.pi / 2))
    thv = np.zeros((K, 3), f)
    for i in range(K):
        if valid[i]:
            v = (head[i] - tail[i]).astype(f)
            thv[i] = v / norm(v)
    nv = np.zeros((K, 3), f)
    for p in range(2):
        src = thv if p == 0 else nv
        for i in range(K):
            if not valid[i]:
                continue
            s = np.zeros(3, f)
            for l in lists[i]:
                if within(src[l], src[i]):
                    s = (s + src[l]).astype(f)
            s = (s + src[i]).astype(f)
            nv[i] = s / norm(s)
    return nv


@pytest.mark.parametrize("seed", range(4))
def test_patch_lists_in_push_order(seed):
    rng = np.random.default_rng(seed)
    K = int(rng.integers(1, 200))
    n = int(rng.integers(0, 2000))
    i = rng.integers(0, K, n); j = rng.integers(0, K, n)
    pairs = np.stack([i, j], 1).astype(np.uint32)
    off, ids = api.membrane_patches_from_pairs(pairs, K)
    woff, wids = patches_loop(pairs.tolist(), K)
    assert np.array_equal(off, woff) and np.array_equal(ids, wids)


def test_patch_lists_refuse_ids_out_of_range():
    with pytest.raises(MolarHipError):
        api.membrane_patches_from_pairs(np.array([[0, 7]], np.uint32), 7)


@pytest.mark.parametrize("seed", range(6))
def test_initial_normals_follow_the_reference_loops(seed):
    rng = np.random.default_rng(100 + seed)
    side = int(rng.integers(4, 12))
    K = side * side
    g = (np.stack(np.meshgrid(np.arange(side), np.arange(side), indexing="ij"), -1).reshape(-1, 2) + rng.normal(0, 0.2, (K, 2))) * 0.8
    order = rng.permutation(K) if seed % 2 else np.arange(K)            # lattice order: the longest chains through pass 2
    g = g[order]
    z = 0.4 * np.sin(g[:, 0]) + rng.normal(0, 0.05, K)
    head = np.concatenate([g, z[:, None] + 2.0], 1).astype(np.float32)
    tilt = rng.normal(0, 0.5, (K, 3)); tilt[:, 2] = -1.5
    if seed == 3:
        tilt[::5] *= -1.0                                               # some lipids upside down: the 90-degree filter decides
    tail = (head + tilt).astype(np.float32)
    d = np.linalg.norm(g[:, None, :] - g[None, :, :], axis=2)
    ii, jj = np.nonzero(np.triu(d < 1.7, 1))
    perm = rng.permutation(len(ii))
    pairs = np.stack([ii[perm], jj[perm]], 1).astype(np.uint32)
    valid = np.ones(K, np.uint8)
    valid[rng.choice(K, K // 10, replace=False)] = 0
    pairs = pairs[valid[pairs[:, 0]].astype(bool) & valid[pairs[:, 1]].astype(bool)]      # patches hold valid lipids only (lib.rs:540-546)
    off, ids = api.membrane_patches_from_pairs(pairs, K)
    got = api.membrane_initial_normals(head, tail, off, ids, valid=valid)
    lists = [ids[int(off[k]): int(off[k + 1])].astype(int).tolist() for k in range(K)]
    want = normals_loop(head, tail, lists, valid)
    assert np.allclose(got, want, atol=2e-6)
    ok = valid.astype(bool)
    assert np.allclose(np.linalg.norm(got[ok], axis=1), 1.0, atol=1e-5) and not got[~ok].any()


def test_initial_normals_at_exactly_ninety_degrees():
    """angle <= FRAC_PI_2 includes the right angle itself (acos(0) rounds to the f32 pi/2)."""
    head = np.array([[0, 0, 1], [1, 0, 0]], np.float32)
    tail = np.zeros((2, 3), np.float32)
    off = np.array([0, 1, 2], np.uint64); ids = np.array([1, 0], np.uint64)
    got = api.membrane_initial_normals(head, tail, off, ids)
    s = np.float32(1) / np.sqrt(np.float32(2))
    assert np.allclose(got[0], [s, 0, s], atol=1e-6)                    # pass 1: both unit vectors summed, then pass 2 of equal normals
    assert np.allclose(got[1], [s, 0, s], atol=1e-6)


def random_neighbour_state(rng, K):
    """A Voronoi-like neighbour graph in the slotted layout of molar_hip_membrane_state: lipid i owns slots
    [patch_off[i] + 4 i, ...) of neib_ids and fills the first nvert[i]."""
    plen = rng.integers(0, 9, K)
    patch_off = np.concatenate([[0], np.cumsum(plen)]).astype(np.uint64)
    patch_ids = rng.integers(0, K, int(patch_off[-1])).astype(np.uint64)
    valid = (rng.random(K) < 0.85).astype(np.uint8)
    nvert = np.zeros(K, np.uint32)
    neib = np.zeros(int(patch_off[-1]) + 4 * K, np.uint64)
    lists = []
    for i in range(K):
        room = int(plen[i]) + 4
        n = int(rng.integers(0, room + 1)) if (valid[i] or rng.random() < 0.5) else 0        # lipids dropped late keep their neighbours
        ids = rng.choice(K, size=min(n, K), replace=False)
        nvert[i] = len(ids)
        s0 = int(patch_off[i]) + 4 * i
        neib[s0:s0 + len(ids)] = ids
        lists.append([int(x) for x in ids])
    return valid, patch_off, patch_ids, nvert, neib, lists


def shell_of(i, lists, n):
    """lib.rs:572-578: a set of the direct neighbours, extended (n - 2) times by the neighbours of its members"""
    s = set(lists[i])
    for _ in range(2, n):
        for m in list(s):
            s.update(lists[m])
    return sorted(s)


@pytest.mark.parametrize("seed", range(5))
@pytest.mark.parametrize("n_shells", [1, 2, 3, 4])
def test_nth_shell_patches(seed, n_shells):
    rng = np.random.default_rng(300 + seed)
    K = int(rng.integers(1, 120))
    valid, patch_off, patch_ids, nvert, neib, lists = random_neighbour_state(rng, K)
    off, ids = api.membrane_nth_shell_patches(valid, patch_off, patch_ids, nvert, neib, n_shells)
    for i in range(K):
        got = ids[int(off[i]): int(off[i + 1])].astype(int).tolist()
        want = shell_of(i, lists, n_shells) if valid[i] else patch_ids[int(patch_off[i]): int(patch_off[i + 1])].astype(int).tolist()
        assert got == want, (i, n_shells)
    if n_shells >= 3:       # a lipid with a neighbour that points back is a member of its own shell, as in the reference
        back = [i for i in range(K) if valid[i] and any(i in lists[m] for m in lists[i])]
        assert all(i in ids[int(off[i]): int(off[i + 1])] for i in back)


@pytest.mark.parametrize("seed", range(5))
def test_smooth_curvature(seed):
    rng = np.random.default_rng(400 + seed)
    K = int(rng.integers(2, 150))
    valid, patch_off, patch_ids, nvert, neib, lists = random_neighbour_state(rng, K)
    mean = rng.normal(0, 0.3, K).astype(np.float32); gauss = rng.normal(0, 0.1, K).astype(np.float32)
    for n in (1, 2, 3):
        gm, gg = api.membrane_smooth_curvature(valid, patch_off, nvert, neib, n, mean, gauss)
        f = np.float32
        for i in range(K):
            if not valid[i]:
                assert gm[i] == mean[i] and gg[i] == gauss[i]
                continue
            m = g = f(0)
            nv = 0
            for l in shell_of(i, lists, n):
                if valid[l]:
                    m = f(m + mean[l]); g = f(g + gauss[l]); nv += 1
            assert gm[i] == f(f(mean[i] + m) / f(nv + 1)) and gg[i] == f(f(gauss[i] + g) / f(nv + 1))
    gm, gg = api.membrane_smooth_curvature(valid, patch_off, nvert, neib, 0, mean, gauss)
    assert np.array_equal(gm, mean) and np.array_equal(gg, gauss)


def test_shell_argument_errors():
    v = np.ones(2, np.uint8); po = np.array([0, 1, 2], np.uint64); pi = np.array([1, 0], np.uint64)
    nv = np.array([9, 1], np.uint32); nb = np.zeros(10, np.uint64)
    with pytest.raises(MolarHipError):
        api.membrane_nth_shell_patches(v, po, pi, nv, nb, 2)              # more vertices than the lipid's slots hold
    with pytest.raises(MolarHipError):
        api.membrane_nth_shell_patches(v, po, pi, np.array([1, 1], np.uint32), nb, 0)       # n_shells < 1


def test_invalid_lipid_with_oversized_vertex_count_is_not_read_past_its_slots():
    """A lipid the smoothing pass dropped keeps whatever vertex count it had; as a member of a valid lipid's shell its
    neighbour list is walked, and a stale / garbage count must not send the walk past the lipid's slots (the ABI has no
    array lengths: round-3 advisor finding, a segfault with nvert = 4e8)."""
    valid = np.array([1, 0], np.uint8)
    po = np.array([0, 1, 2], np.uint64)
    pi = np.array([1, 0], np.uint64)
    nb = np.full(10, 2**40, np.uint64)            # ids >= K are skipped
    nb[0] = 1                                     # lipid 0 (slots 0..4): neighbour 1
    nb[5] = 0                                     # lipid 1 (slots 5..9): neighbour 0
    for bad in (6, 400_000_000, 0xFFFFFFFF):
        nv = np.array([1, bad], np.uint32)
        off, ids = api.membrane_nth_shell_patches(valid, po, pi, nv, nb, 3)
        assert ids[int(off[0]): int(off[1])].tolist() == [0, 1]
        assert ids[int(off[1]): int(off[2])].tolist() == [0]          # not valid: keeps its patch
        m, g = api.membrane_smooth_curvature(valid, po, nv, nb, 3, np.array([1.0, 5.0], np.float32), np.array([2.0, 7.0], np.float32))
        assert m[0] == np.float32(1.0) and g[0] == np.float32(2.0) and m[1] == np.float32(5.0)   # own value + itself via the way back... 
