"""GRO plumbing of BASELINE config 1 on the host mirror (molar_amd/gro.py) against the oracle's restatement and
against hand-checked rules of the reference (gro_handler.rs:55-288, atom.rs:238-291)."""
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def gro():
    from molar_amd import build
    build.build_library()
    from molar_amd import gro as g
    return g


def test_element_and_mass_guessing(gro):
    from oracle import gro_oracle as og
    cases = [("CA", "CA", 20), ("CA", "ALA", 6), ("CL", "CL", 17), ("CLA", "CLA", 17), ("NA", "NA", 11), ("SOD", "SOD", 11),
             ("POT", "POT", 19), ("1HB2", "ALA", 1), ("OW", "SOL", 8), ("HW1", "SOL", 1), ("FE", "HEM", 26), ("ZN", "ZN", 30),
             ("N", "ALA", 7), ("P", "POPE", 15), ("C21", "POPE", 6), ("MG", "MG", 12), ("123", "XXX", 0), ("OH2", "TIP3", 8)]
    for name, res, z in cases:
        assert gro.guess_element(name, res) == z, (name, res)
        assert og.element_of(name, res) == gro.ELEMENT_NAME[z], (name, res)
    assert np.isclose(gro.ELEMENT_MASS[6], 12.0107) and np.isclose(gro.ELEMENT_VDW[1] * 0.1, 0.12)


def test_read_write_roundtrip_and_box_order(gro, tmp_path):
    from molar_amd import api
    from oracle import gro_oracle as og
    rng = np.random.default_rng(0)
    n = 500
    names = [["OW", "HW1", "HW2"][k % 3] for k in range(n)]
    resn = ["SOL"] * n
    resid = np.arange(n) // 3 + 1
    box = np.array([[5.0, 1.25, -0.5], [0.0, 4.5, 0.75], [0.0, 0.0, 6.0]], np.float32)      # columns a, b, c
    xyz = (rng.random((n, 3)) * 5).astype(np.float32)
    top = gro.GroTopology(names, resn, resid)
    st = api.State(xyz, api.PeriodicBox.from_matrix(box), 12.5)
    p = tmp_path / "w.gro"
    gro.write_gro(p, top, st)
    text = open(p).read().splitlines()
    assert text[0] == "Created by Molar, t= 12.500" and text[1] == "500"
    assert text[2] == "    1SOL     OW    1%8.3f%8.3f%8.3f" % tuple(xyz[0])
    # box line: xx yy zz  a_y a_z  b_x b_z  c_x c_y
    assert [float(x) for x in text[-1].split()] == [5.0, 4.5, 6.0, 0.0, 0.0, 1.25, 0.0, -0.5, 0.75]
    top2, st2 = gro.read_gro(p)
    o = og.read_gro(p)
    assert np.allclose(st2.coords, xyz, atol=5.1e-4) and np.array_equal(st2.coords, o["xyz"])
    assert np.array_equal(st2.pbox.get_matrix(), box) and np.array_equal(o["box"], box)
    assert st2.time == 12.5 and top2.names == names and np.array_equal(top2.resids, resid)
    assert np.array_equal(top2.masses, o["mass"])
    assert np.allclose(top2.masses[:3], [15.9994, 1.00794, 1.00794])
    # orthorhombic boxes are written with three numbers and read back with zero off-diagonals
    st3 = api.State(xyz, api.PeriodicBox.from_matrix(np.diag([3.0, 4.0, 5.0])), 0.0)
    gro.write_gro(p, top, st3)
    assert len(open(p).read().splitlines()[-1].split()) == 3
    assert np.array_equal(gro.read_gro(p)[1].pbox.get_matrix(), np.diag([3.0, 4.0, 5.0]).astype(np.float32))
