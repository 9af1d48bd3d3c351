"""Pins the oracle's PeriodicBox arithmetic to the reference's own known answers.

Each test restates the INPUTS and EXPECTED VALUES of one reference test
(molar/src/periodic_box.rs:456-620 and molar_python/tests/test_2.py:233-245) — data only.
"""
import numpy as np
import pytest

from oracle.oracle import PBC_FULL, PBC_NONE

EPS = 1e-6


def diag_box(o, x, y, z):
    return o.box_from_matrix(np.diag([x, y, z]))


@pytest.fixture(params=["f32", "f64"])
def o(request, orc32, orc64):
    return orc32 if request.param == "f32" else orc64


# periodic_box.rs:456-496
@pytest.mark.parametrize("dims,expect", [
    (PBC_NONE, (8.0, 8.0, 8.0)),
    (PBC_FULL, (-2.0, -2.0, -2.0)),
    ((True, False, False), (-2.0, 8.0, 8.0)),
    ((True, True, False), (-2.0, -2.0, 8.0)),
])
def test_shortest_vector_dims(o, dims, expect):
    b = diag_box(o, 10, 10, 10)
    r = o.shortest_vector_dims(b, [8.0, 8.0, 8.0], dims)
    assert np.linalg.norm(r - np.array(expect)) < EPS


# periodic_box.rs:498-542
@pytest.mark.parametrize("dims,expect", [
    (PBC_NONE, (8.0, 8.0, 8.0)),
    (PBC_FULL, (-2.0, -2.0, -2.0)),
    ((True, False, False), (-2.0, 8.0, 8.0)),
    ((True, True, False), (-2.0, -2.0, 8.0)),
])
def test_closest_image(o, dims, expect):
    b = diag_box(o, 10, 10, 10)
    r = o.closest_image_dims(b, [8.0, 8.0, 8.0], [0.0, 0.0, 0.0], dims)
    assert np.linalg.norm(r - np.array(expect)) < EPS


# periodic_box.rs:546-551
def test_orthogonal_has_no_tric_corrections(o):
    b = diag_box(o, 10, 20, 30)
    assert b.nshift == 0


# periodic_box.rs:559-575 (issue #6 regression; mdtraj/MDAnalysis/brute force agree on 5.353627)
def test_triclinic_mdtraj_box_matches_brute_force(o):
    m = [[10.0, 4.0, -4.0], [0.0, 10.0, 0.0], [0.0, 0.0, 10.0]]   # columns are box vectors
    b = o.box_from_matrix(m)
    d = o.distance(b, [38.9214, 40.0078, -34.0795], [-26.6187, 40.8926, 30.9709], PBC_FULL)
    assert abs(d - 5.353627) < 1e-3


# periodic_box.rs:580-603
def test_triclinic_corner_matches_brute_force(o):
    m = np.array([[6.0, 0.0, 3.0], [0.0, 6.0, 3.0], [0.0, 0.0, 6.0]])
    b = o.box_from_matrix(m)
    dx = np.array([2.9, 2.9, 2.9])
    a, bb, c = m[:, 0], m[:, 1], m[:, 2]
    best = min(np.linalg.norm(dx + i * a + j * bb + k * c)
               for i in range(-2, 3) for j in range(-2, 3) for k in range(-2, 3))
    got = np.linalg.norm(o.shortest_vector_dims(b, dx, PBC_FULL))
    assert abs(got - best) < 1e-5


# periodic_box.rs:607-619
def test_triclinic_far_apart_reduction(o):
    m = [[10.0, 4.0, -4.0], [0.0, 10.0, 0.0], [0.0, 0.0, 10.0]]
    b = o.box_from_matrix(m)
    d = o.distance(b, [0.1, 0.2, 0.3], [60.1, 0.2, 0.3], PBC_FULL)
    assert d < 1e-4


# periodic_box.rs:448-454
def test_invalid_from_vec_ang(o):
    with pytest.raises(ValueError):
        o.box_from_vectors_angles(10.0, 0.2, 15.0, 90.0, 9.0, 90.0)


# molar_python/tests/test_2.py:233-245
def test_python_known_answer(o):
    b = o.box_from_vectors_angles(1, 2, 3, 90, 90, 90)
    v = o.shortest_vector_dims(b, [0.9, 0.5, 0.6], PBC_FULL)
    assert v[0] == pytest.approx(-0.1, abs=1e-6)
    assert v[1] == pytest.approx(0.5, abs=1e-6)
    assert v[2] == pytest.approx(0.6, abs=1e-6)


def test_lab_extents_are_row_sums(o):
    # periodic_box.rs:369-375
    m = np.array([[21.544, 0.0, -3.0], [0.0, 21.544, -3.0], [0.0, 0.0, 21.544]])
    b = o.box_from_matrix(m)
    assert np.allclose(o.lab_extents(b), m.sum(axis=1), rtol=1e-6)
    assert np.allclose(o.box_extents(b), np.linalg.norm(m, axis=0), rtol=1e-6)


def test_inverse_matches_numpy(o):
    m = np.array([[10.0, 4.0, -4.0], [0.0, 10.0, 0.0], [0.0, 0.0, 10.0]])
    b = o.box_from_matrix(m)
    assert np.allclose(o.box_inv(b), np.linalg.inv(m), rtol=1e-5, atol=1e-7)
