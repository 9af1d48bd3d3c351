"""Second restatement of Measure::lipid_tail_order (molar/src/measure.rs:270-422) in float64 numpy, from a separate reading
of the Rust source - test infrastructure: tests/test_oracle_cross_cpu.py compares it with oracle/molar_oracle.c (f64 build)."""
from __future__ import annotations

import numpy as np


def _angle(a, b):
    """nalgebra Vector::angle: acos(clamp(a.b / (|a||b|), -1, 1)); 0 if either vector has zero norm."""
    na, nb = np.linalg.norm(a), np.linalg.norm(b)
    if na == 0.0 or nb == 0.0:
        return 0.0
    return float(np.arccos(np.clip(np.dot(a, b) / (na * nb), -1.0, 1.0)))


def _unit(v):
    with np.errstate(all="ignore"):
        return v / np.linalg.norm(v)          # nalgebra normalize(): no zero check (NaN for a zero vector)


def lipid_tail_order(pos, order_type, normals, bond_orders):
    """pos: [n,3] tail carbons in chain order; order_type 0 Sz, 1 Scd, 2 ScdCorr; normals: [1,3] or [n-2,3]; bond_orders: n-1."""
    p = np.asarray(pos, np.float64)
    nrm = np.asarray(normals, np.float64).reshape(-1, 3)
    bo = list(bond_orders)
    n = len(p)
    if n < 3:
        raise ValueError("TailTooShort")
    if len(nrm) != 1 and len(nrm) != n - 2:
        raise ValueError("NormalsCount")
    if len(bo) != n - 1:
        raise ValueError("BondOrderCount")
    N = (lambda k: nrm[0]) if len(nrm) == 1 else (lambda k: nrm[k])
    order = np.zeros(n - 2)
    if order_type == 0:
        for at in range(1, n - 1):
            ang = _angle(p[at + 1] - p[at - 1], N(at - 1))
            order[at - 1] = 1.5 * np.cos(ang) ** 2 - 0.5
        return order
    for i in range(n - 2):
        if bo[i] == 1:
            if bo[i + 1] == 1:
                p1, p2, p3 = p[i], p[i + 1], p[i + 2]
                lz = _unit(p3 - p1)
                lx = _unit(np.cross(p1 - p2, p3 - p2))
                ly = np.cross(lx, lz)
                nv = N(i)
                sxx = 0.5 * (3.0 * np.cos(_angle(lx, nv)) ** 2 - 1.0)
                syy = 0.5 * (3.0 * np.cos(_angle(ly, nv)) ** 2 - 1.0)
                order[i] = -(2.0 * sxx + syy) / 3.0
        else:
            p1, p2, p3, p4 = p[i - 1], p[i], p[i + 1], p[i + 2]
            a1 = 0.5 * (np.pi - _angle(p1 - p2, p3 - p2))
            a2 = 0.5 * (np.pi - _angle(p2 - p3, p4 - p3))
            lz = _unit(p3 - p2)
            for first in (True, False):
                lx = _unit(np.cross(p1 - p2 if first else p3 - p4, lz))
                ly = np.cross(lx, lz)
                nv = N(i if first else i + 1)
                cy, cz = np.cos(_angle(ly, nv)), np.cos(_angle(lz, nv))
                szz = 0.5 * (3.0 * cz ** 2 - 1.0)
                syy = 0.5 * (3.0 * cy ** 2 - 1.0)
                syz = 1.5 * cy * cz
                a, sg = (a1, -1.0) if first else (a2, 1.0)
                if order_type == 2:
                    v = -(np.cos(a) ** 2 * syy + np.sin(a) ** 2 * szz + sg * 2.0 * np.cos(a) * np.sin(a) * syz)
                else:
                    v = -(szz / 4.0 + 3.0 * syy / 4.0 + sg * np.sqrt(3.0) * syz / 2.0)
                order[i - 1 if first else i] = v
    return order
