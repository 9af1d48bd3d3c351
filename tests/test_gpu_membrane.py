"""Per-lipid membrane analysis (markers, patches, normals, tail order) on the GPU against the same
pipeline assembled from the CPU oracle's primitives (restating molar_membrane/src/lib.rs:435-558)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from molar_amd import build
    from molar_amd.api import Engine
    build.build_library()
    return Engine(0)


def oracle_normals(o, head, tail, patch):
    """lib.rs:456-505 in f32, sequential second pass."""
    f = np.float32
    K = len(head)
    thv = np.zeros((K, 3), f)
    for i in range(K):
        v = (head[i] - tail[i]).astype(f)
        thv[i] = v / f(np.sqrt(f(f(v[0] * v[0] + v[1] * v[1]) + v[2] * v[2])))

    def ang(a, b):
        n1 = f(np.sqrt(f(f(a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]))); n2 = f(np.sqrt(f(f(b[0] * b[0] + b[1] * b[1]) + b[2] * b[2])))
        if n1 == 0 or n2 == 0:
            return f(0)
        c = f(f(f(a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]) / f(n1 * n2))
        return f(np.arccos(np.clip(c, f(-1), f(1))))
    nv = np.zeros((K, 3), f)
    for p in range(2):
        src = thv if p == 0 else nv
        for i in range(K):
            s = np.zeros(3, f)
            for l in patch[i]:
                if ang(src[l], src[i]) <= f(np.pi / 2):
                    s = (s + src[l]).astype(f)
            s = (s + src[i]).astype(f)
            nv[i] = s / f(np.sqrt(f(f(s[0] * s[0] + s[1] * s[1]) + s[2] * s[2])))
    return nv


def test_membrane_pipeline_matches_oracle(eng, orc32):
    from molar_amd import membrane as mb
    xyz, box, first, tpl, masses = mb.build_bilayer(200, 40000)
    K = len(first)
    m = mb.Membrane(eng, len(xyz), first, tpl, masses, mb.MembraneOptions(cutoff=1.5, order_type=1))
    work = xyz.copy()
    res = m.compute(work, box)
    ob = orc32.box_from_matrix(box)
    # unwrap per lipid (exact f32 arithmetic)
    ref_xyz = xyz.copy()
    for k in range(K):
        idx = np.arange(first[k], first[k] + tpl.natoms, dtype=np.uint64)
        ref_xyz = orc32.unwrap_simple_dim(ref_xyz, ob, 7, idx)
    assert np.array_equal(work, ref_xyz)
    # markers: centre of mass per sub-selection
    for name, sub in (("head", tpl.head), ("mid", tpl.mid), ("tail", tpl.tail_end)):
        want = np.array([orc32.center_of_mass(ref_xyz, masses, first[k] + sub.astype(np.uint64)) for k in range(K)])
        assert np.allclose(res[name], want, rtol=2e-6, atol=2e-6)
    # patches from the GPU markers (the search itself is bit-exact given identical input)
    r = orc32.search_single_pbc(1.5, res["head"], ob, 7)
    patch = [[] for _ in range(K)]
    for i, j in zip(r["i"].tolist(), r["j"].tolist()):
        patch[i].append(j); patch[j].append(i)
    for k in range(K):
        got = res["patch_ids"][int(res["patch_off"][k]): int(res["patch_off"][k + 1])].tolist()
        assert got == patch[k]
    assert np.mean([len(p) for p in patch]) > 4
    # normals
    want_n = oracle_normals(orc32, res["head"], res["tail"], patch)
    assert np.allclose(res["normals"], want_n, atol=2e-6)
    assert np.allclose(np.linalg.norm(res["normals"], axis=1), 1.0, atol=1e-5)
    # upper leaflet normals point up, lower down (tails towards the mid-plane)
    assert (res["normals"][:200, 2] > 0.8).all() and (res["normals"][200:, 2] < -0.8).all()
    # order parameters per tail with the lipid normal
    for t, carbons in enumerate(tpl.tails):
        for k in range(0, K, 7):
            want = orc32.lipid_tail_order(ref_xyz, 1, res["normals"][k][None, :], tpl.bond_orders[t],
                                          idx=first[k] + carbons.astype(np.uint64))
            assert np.allclose(res["order"][t][k], want, atol=3e-5)
    # roughly ordered chains along the normal: mean |Scd| in a sensible range
    assert 0.05 < np.abs(np.concatenate([o.reshape(-1) for o in res["order"]])).mean() < 0.6
