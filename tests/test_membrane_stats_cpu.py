"""molar_membrane's accumulators (stats.rs, lipid_group.rs) on the host mirror: binning rule against the oracle's
restatement, MeanStd/MeanStdVec against numpy, a LipidGroup frame update on a hand-made frame, file formats."""
import os

import numpy as np
import pytest

from molar_amd.membrane_stats import Histogram1D, LipidGroup, MeanStd, MeanStdVec


def test_histogram_matches_oracle_binning(orc32):
    rng = np.random.default_rng(0)
    v = np.concatenate([rng.uniform(-0.2, 0.2, 5000), [-0.15, 0.15, np.nan, np.inf, -np.inf, 0.1499999]]).astype(np.float32)
    h = Histogram1D(-0.15, 0.15, 100)          # main.rs:39
    h.add_many(v)
    assert np.array_equal(h.bins, orc32.histogram_add(-0.15, 0.15, 100, v))
    total = h.bins.sum()
    h.normalize_density()
    assert np.isclose(h.bins.sum() * (0.3 / 100), 1.0, rtol=1e-5) and total > 0


def test_meanstd():
    rng = np.random.default_rng(1)
    x = rng.normal(3.0, 0.5, 1000).astype(np.float32)
    a, b = MeanStd(), MeanStd()
    for v in x:
        a.add(v)
    b.add_many(x)
    for acc in (a, b):
        m, s = acc.compute()
        assert np.isclose(m, x.mean(), rtol=1e-4) and np.isclose(s, x.std(), rtol=2e-3)
    c = MeanStd()
    for _ in range(5):
        c.add(2.0)
    assert c.compute() == (np.float32(2.0), np.float32(0.0))        # x2/n == mean^2 -> 0, not NaN
    with pytest.raises(ValueError):
        MeanStd().compute()
    mv = MeanStdVec(3)
    rows = rng.normal(size=(50, 3)).astype(np.float32)
    mv.add_many(rows[:25])
    for r in rows[25:]:
        mv.add(r)
    m, s = mv.compute()
    assert np.allclose(m, rows.mean(0), atol=1e-5) and np.allclose(s, rows.std(0), atol=1e-4)
    with pytest.raises(ValueError):
        mv.add(np.zeros(4))


def test_group_frame_update_and_files(tmp_path):
    K = 6
    names = ["POPE", "POPG"]
    species = np.array([0, 0, 1, 0, 1, 0])
    patch_off = np.array([0, 2, 4, 6, 8, 10, 12], np.uint64)
    nvert = np.array([2, 2, 1, 2, 2, 2], np.uint32)
    neib = np.zeros(12 + 4 * K, np.uint64)
    nb = {0: [1, 2], 1: [0, 3], 2: [4], 3: [1, 5], 4: [2, 5], 5: [3, 4]}
    for i in range(K):
        s0 = int(patch_off[i]) + 4 * i
        neib[s0:s0 + len(nb[i])] = nb[i]
    normals = np.tile(np.array([0, 0, 1], np.float32), (K, 1))
    thv = normals.copy(); thv[1] = [0, np.sin(np.radians(30)), np.cos(np.radians(30))]
    res = dict(valid=np.array([1, 1, 1, 0, 1, 1], np.uint8), patch_off=patch_off, nvert=nvert, neib_ids=neib,
               area=np.array([0.6, 0.7, 0.65, 9.9, 0.62, 0.58], np.float32), normals=normals,
               mean_curv=np.linspace(-0.1, 0.1, K).astype(np.float32), gauss_curv=np.zeros(K, np.float32),
               order=[np.arange(K * 3, dtype=np.float32).reshape(K, 3) / 10, np.ones((K, 2), np.float32)])
    g = LipidGroup(names, {n: [5, 4] for n in names})
    g.lipid_ids = np.array([0, 1, 2, 3, 5])            # lipid 4 not in the group, lipid 3 invalid
    g.frame_update(res, species, thv)
    g.frame_update(res, species, thv)
    pe, pg = g.per_species["POPE"], g.per_species["POPG"]
    assert pe.num_lip.compute()[0] == 3 and pg.num_lip.compute()[0] == 1           # valid group members per species
    assert np.isclose(pe.area.compute()[0], np.mean([0.6, 0.7, 0.58]), rtol=1e-6)
    assert np.isclose(pe.tilt.compute()[0], 10.0, atol=1e-3)                       # (0 + 30 + 0) / 3 degrees
    assert np.isclose(pe.num_neib.compute()[0], 2.0) and np.isclose(pg.num_neib.compute()[0], 1.0)
    # neighbours of POPE lipids 0,1,5: {1,2},{0,3},{3,4} -> POPE 4 (1,0,3,3), POPG 2 (2,4); per lipid: 4/3 and 2/3
    assert np.isclose(pe.neib_species["POPE"].compute()[0], 4 / 3, rtol=1e-6)
    assert np.isclose(pe.neib_species["POPG"].compute()[0], 2 / 3, rtol=1e-6)
    assert np.allclose(pe.order[0].compute()[0], res["order"][0][[0, 1, 5]].mean(0), atol=1e-6)
    g.save(tmp_path, "upper")
    txt = open(os.path.join(tmp_path, "gr_upper_stats.dat")).read().splitlines()
    assert txt[0].startswith("#species\tnum\tnum_std\tarea") and txt[1].startswith("POPE\t   3.000\t   0.000\t   0.627")
    order = open(os.path.join(tmp_path, "gr_upper_order_POPE.dat")).read().splitlines()
    assert order[0] == "# time\taver\ttail1\ttail2" and order[3].endswith("\t--") and order[1].startswith("1.000\t")
    assert "POPG" in open(os.path.join(tmp_path, "gr_upper_neib_stats.dat")).read()
