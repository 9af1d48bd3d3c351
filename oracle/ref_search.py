"""A SECOND, independent restatement of the reference's distance search - test infrastructure, like everything in oracle/.

oracle/molar_oracle.c was written from a first reading of molar/src/distance_search.rs and periodic_box.rs and is
validated against brute force only where the reference's half-shell grid is geometrically complete.  On GROMACS-style
and strongly sheared boxes the grid is NOT complete, brute force is no witness, and the C restatement stood alone.  This
module restates the same functions again, from a separate reading of the Rust source, in the most literal form Python
allows (lists of lists for the cells, the plan as a list of tuples, one loop per Rust loop, np.float32 scalars with the
operation order written out), so that tests/test_oracle_cross_cpu.py can demand that both restatements produce identical
ordered results on exactly those boxes.  Pure-Python loops: a few hundred atoms at most.

Line numbers refer to /root/reference/molar/src/."""
from __future__ import annotations

import math

import numpy as np

f32 = np.float32
EPS = f32(1.1920929e-07)          # Float::EPSILON

# distance_search.rs:39-60
MASK = [((0, 0, 0), (0, 0, 0)),
        ((0, 0, 0), (1, 0, 0)), ((0, 0, 0), (0, 1, 0)), ((0, 0, 0), (0, 0, 1)),
        ((0, 0, 0), (1, 1, 0)), ((0, 0, 0), (1, 0, 1)), ((0, 0, 0), (0, 1, 1)),
        ((0, 0, 0), (1, 1, 1)),
        ((1, 0, 0), (0, 1, 0)), ((1, 0, 0), (0, 0, 1)), ((0, 1, 0), (0, 0, 1)),
        ((1, 1, 0), (0, 0, 1)), ((1, 0, 1), (0, 1, 0)), ((0, 1, 1), (1, 0, 0))]


def _matvec(m, v):
    """nalgebra Matrix3 * Vector3 (column-axpy gemv): y_i = (m_i0 v0 + m_i1 v1) + m_i2 v2.  m[r][c], f32."""
    return [f32(f32(f32(m[r][0] * v[0]) + f32(m[r][1] * v[1])) + f32(m[r][2] * v[2])) for r in range(3)]


def _norm2(v):
    return f32(f32(f32(v[0] * v[0]) + f32(v[1] * v[1])) + f32(v[2] * v[2]))


def _round_away(x):
    """f32::round: half away from zero."""
    x = float(x)
    return f32(math.floor(abs(x) + 0.5) * (1.0 if x >= 0 else -1.0)) if math.isfinite(x) else f32(x)


def _as_usize(x):
    """Rust `as usize` on a float: saturating, NaN -> 0."""
    x = float(x)
    if x != x or x <= 0:
        return 0
    return int(min(x, 1.8e19))


class Box:
    """PeriodicBox (periodic_box.rs:15-23): matrix with COLUMNS a, b, c; inverse; triclinic correction shifts."""

    def __init__(self, matrix):
        m = [[f32(matrix[r][c]) for c in range(3)] for r in range(3)]
        self.m = m
        # try_inverse (:167-169), nalgebra's closed form for 3x3
        a, d, g = m[0][0], m[1][0], m[2][0]
        b, e, h = m[0][1], m[1][1], m[2][1]
        c, f, i = m[0][2], m[1][2], m[2][2]
        minor_bf = f32(f32(e * i) - f32(h * f))
        minor_af = f32(f32(d * i) - f32(g * f))
        minor_ae = f32(f32(d * h) - f32(g * e))
        det = f32(f32(f32(a * minor_bf) - f32(b * minor_af)) + f32(c * minor_ae))
        if det == 0:
            raise ValueError("InverseFailed")
        self.inv = [[f32(minor_bf / det), f32(f32(f32(c * h) - f32(i * b)) / det), f32(f32(f32(b * f) - f32(e * c)) / det)],
                    [f32(-minor_af / det), f32(f32(f32(a * i) - f32(g * c)) / det), f32(f32(f32(c * d) - f32(f * a)) / det)],
                    [f32(minor_ae / det), f32(f32(f32(b * g) - f32(h * a)) / det), f32(f32(f32(a * e) - f32(d * b)) / det)]]
        self.shifts = self._tric_corrections()

    def _tric_corrections(self):          # periodic_box.rs:25-66
        m = self.m
        if all(m[r][c] == 0 for r in range(3) for c in range(3) if r != c):
            return []
        col = lambda k: [m[0][k], m[1][k], m[2][k]]
        a, b, c = col(0), col(1), col(2)
        add = lambda u, v: [f32(u[k] + v[k]) for k in range(3)]
        sub = lambda u, v: [f32(u[k] - v[k]) for k in range(3)]
        neg = lambda u: [f32(-u[k]) for k in range(3)]
        norm = lambda u: f32(np.sqrt(_norm2(u)))
        n1 = norm(add(add(a, b), c)); n2 = norm(sub(add(a, b), c)); n3 = norm(add(sub(a, b), c)); n4 = norm(add(add(neg(a), b), c))
        half = f32(f32(0.5) * max(max(max(n1, n2), n3), n4))
        t = f32(f32(2.0) * half)
        bound2 = f32(t * t)
        out = []
        for i in (-1, 0, 1):
            for j in (-1, 0, 1):
                for k in (-1, 0, 1):
                    if i == 0 and j == 0 and k == 0:
                        continue
                    s = [f32(f32(f32(f32(i) * a[q]) + f32(f32(j) * b[q])) + f32(f32(k) * c[q])) for q in range(3)]
                    if _norm2(s) < bound2:
                        out.append(s)
        return out

    def lab_extents(self):                # :369-375 (row sums)
        m = self.m
        return [f32(f32(m[r][0] + m[r][1]) + m[r][2]) for r in range(3)]

    def shortest_vector_dims(self, v, dims):      # :286-318
        bv = _matvec(self.inv, v)
        for k in range(3):
            if dims >> k & 1:
                bv[k] = f32(bv[k] - _round_away(bv[k]))
        start = _matvec(self.m, bv)
        if not self.shifts or dims != 7:
            return start
        best, best2 = start, _norm2(start)
        for s in self.shifts:
            cand = [f32(start[k] + s[k]) for k in range(3)]
            n2 = _norm2(cand)
            if n2 < best2:
                best2, best = n2, cand
        return best

    def distance_squared(self, p1, p2, dims):     # :379-381
        return _norm2(self.shortest_vector_dims([f32(p2[k] - p1[k]) for k in range(3)], dims))


class Grid:
    """distance_search.rs:33-214.  cells[c] = list of (id, position)."""

    def __init__(self, dims):
        self.dims = list(dims)
        self.cells = [[] for _ in range(dims[0] * dims[1] * dims[2])]

    @staticmethod
    def from_cutoff_and_extents(cutoff, ext):     # :103-110
        return Grid([max(_as_usize(np.floor(f32(ext[d] / cutoff))), 1) for d in range(3)])

    def loc_to_ind(self, loc):                    # :85-87
        return loc[0] + loc[1] * self.dims[0] + loc[2] * self.dims[0] * self.dims[1]

    def populate(self, pos, ids, lower, upper):   # :120-142
        sz = [f32(upper[d] - lower[d]) for d in range(3)]
        for id_, p in zip(ids, pos):
            loc, keep = [0, 0, 0], True
            for d in range(3):
                with np.errstate(all="ignore"):
                    n = np.floor(f32(f32(f32(self.dims[d]) * f32(p[d] - lower[d])) / sz[d]))
                n = float(n)
                n = 0 if n != n else int(max(min(n, 9.2e18), -9.2e18))        # `as isize`
                if n < 0 or n >= self.dims[d]:
                    keep = False
                    break
                loc[d] = n
            if keep:
                self.cells[self.loc_to_ind(loc)].append((int(id_), [f32(p[0]), f32(p[1]), f32(p[2])]))

    def populate_pbc(self, pos, ids, box, dims):  # :144-210
        wrapped = []
        for id_, p in zip(ids, pos):
            p = [f32(p[0]), f32(p[1]), f32(p[2])]
            rel = _matvec(box.inv, p)
            correct, dropped = True, False
            for d in range(3):
                if rel[d] < 0 or rel[d] >= 1:
                    if not (dims >> d & 1):
                        dropped = True
                    else:
                        correct = False
                    break
            if dropped:
                continue
            loc = [0, 0, 0]
            if correct:
                for d in range(3):
                    loc[d] = min(max(_as_usize(np.floor(f32(rel[d] * f32(self.dims[d])))), 0), self.dims[d] - 1)
                self.cells[self.loc_to_ind(loc)].append((int(id_), p))
            else:
                for d in range(3):
                    if dims >> d & 1:
                        fr = f32(np.trunc(rel[d]))
                        rel[d] = f32(rel[d] - fr)             # f32::fract
                        if rel[d] < 0:
                            rel[d] = f32(f32(1.0) + rel[d])
                    loc[d] = min(max(_as_usize(np.floor(f32(rel[d] * f32(self.dims[d])))), 0), self.dims[d] - 1)
                wrapped.append((self.loc_to_ind(loc), int(id_), _matvec(box.m, rel)))
        for c, id_, wp in wrapped:                # :203-209: wrapped atoms go in AFTER all in-box atoms
            self.cells[c].append((id_, wp))


def search_plan(g1, g2, dims):                    # :217-269
    plan = []
    for x in range(g1.dims[0]):
        for y in range(g1.dims[1]):
            for z in range(g1.dims[2]):
                for v1, v2 in MASK:
                    c = [[x + v1[0], y + v1[1], z + v1[2]], [x + v2[0], y + v2[1], z + v2[2]]]
                    wrapped, skip = 0, False
                    for i in (0, 1):
                        for d in range(3):
                            if c[i][d] == g1.dims[d]:
                                if dims >> d & 1:
                                    c[i][d] = 0
                                    wrapped |= 1 << d
                                else:
                                    skip = True
                                    break
                        if skip:
                            break
                    if skip:
                        continue
                    i1, i2 = g1.loc_to_ind(c[0]), g1.loc_to_ind(c[1])
                    if g2 is not None:
                        if (g1.cells[i1] and g2.cells[i2]) or (g2.cells[i1] and g1.cells[i2]):
                            plan.append((i1, i2, wrapped))
                    elif g1.cells[i1] and g1.cells[i2]:
                        plan.append((i1, i2, wrapped))
    return plan


def _d2(p1, p2, wrap, box):
    if box is not None and wrap:                  # pair.2.any()
        return box.distance_squared(p1, p2, wrap)
    return _norm2([f32(p2[k] - p1[k]) for k in range(3)])


def _pair_single(cut2, g, pair, box, out):        # :432-517
    c1, c2, wrap = pair
    if c1 == c2:
        cell = g.cells[c1]
        for i in range(len(cell) - 1):
            for j in range(i + 1, len(cell)):
                d2 = _d2(cell[i][1], cell[j][1], wrap, box)
                if d2 <= cut2:
                    out.append((cell[i][0], cell[j][0], f32(np.sqrt(d2))))
    else:
        for a in g.cells[c1]:
            for b in g.cells[c2]:
                d2 = _d2(a[1], b[1], wrap, box)
                if d2 <= cut2:
                    out.append((a[0], b[0], f32(np.sqrt(d2))))


def _pair_double(cut2, g1, g2, pair, box, out, vdw=None, within=False):     # :271-430
    c1, c2, wrap = pair
    for a in g1.cells[c1]:
        for b in g2.cells[c2]:
            d2 = _d2(a[1], b[1], wrap, box)
            if vdw is not None:
                cut = f32(f32(vdw[0][a[0]] + vdw[1][b[0]]) + EPS)
                hit = d2 <= f32(cut * cut)
            else:
                hit = d2 <= cut2
            if hit:
                if within:
                    out.append(a[0])
                    break
                out.append((a[0], b[0], f32(np.sqrt(d2))))


def _min_max(pos):                                # :602-616, seeded with zeros
    lo, hi = [f32(0)] * 3, [f32(0)] * 3
    for p in pos:
        for d in range(3):
            if f32(p[d]) < lo[d]:
                lo[d] = f32(p[d])
            if f32(p[d]) > hi[d]:
                hi[d] = f32(p[d])
    return lo, hi


def _bbox(cutoff, *sets):                         # :618-646
    mm = [_min_max(s) for s in sets]
    lo = [min(m[0][d] for m in mm) for d in range(3)]
    hi = [max(m[1][d] for m in mm) for d in range(3)]
    return [f32(lo[d] + f32(f32(-cutoff) - EPS)) for d in range(3)], [f32(hi[d] + f32(cutoff + EPS)) for d in range(3)]


def single(cutoff, pos, ids=None, box=None, dims=0):            # :892-954
    cutoff = f32(cutoff)
    ids = range(len(pos)) if ids is None else ids
    if box is not None:
        g = Grid.from_cutoff_and_extents(cutoff, box.lab_extents())
        g.populate_pbc(pos, ids, box, dims)
        plan = search_plan(g, None, dims)
    else:
        lo, hi = _bbox(cutoff, pos)
        g = Grid.from_cutoff_and_extents(cutoff, [f32(hi[d] - lo[d]) for d in range(3)])
        g.populate(pos, ids, lo, hi)
        plan = search_plan(g, None, 0)
    out = []
    for pair in plan:
        _pair_single(f32(cutoff * cutoff), g, pair, box, out)
    return out, g.dims


def double(cutoff, pos1, pos2, ids1=None, ids2=None, box=None, dims=0, vdw=None, within=False, lower=None, upper=None):
    """distance_search_double(_pbc) :659-754, _double_vdw(_pbc) :767-879 (ids local, cutoff from the radii),
    _within(_pbc) :519-598 (lower/upper given by the caller for the non-periodic form)."""
    ids1 = range(len(pos1)) if ids1 is None else ids1
    ids2 = range(len(pos2)) if ids2 is None else ids2
    if vdw is not None:
        m1 = vdw[0][0]
        for v in vdw[0][1:]:
            m1 = max(m1, v)
        m2 = vdw[1][0]
        for v in vdw[1][1:]:
            m2 = max(m2, v)
        cutoff = f32(f32(f32(m1) + f32(m2)) + EPS)
        ids1, ids2 = range(len(pos1)), range(len(pos2))
    cutoff = f32(cutoff)
    if box is not None:
        g1 = Grid.from_cutoff_and_extents(cutoff, box.lab_extents())
        g2 = Grid(g1.dims)
        g1.populate_pbc(pos1, ids1, box, dims)
        g2.populate_pbc(pos2, ids2, box, dims)
        plan = search_plan(g1, g2, dims)
    else:
        if within:
            lo, hi = [f32(x) for x in lower], [f32(x) for x in upper]
        else:
            lo, hi = _bbox(cutoff, pos1, pos2)
        g1 = Grid.from_cutoff_and_extents(cutoff, [f32(hi[d] - lo[d]) for d in range(3)])
        g2 = Grid(g1.dims)
        g1.populate(pos1, ids1, lo, hi)
        g2.populate(pos2, ids2, lo, hi)
        plan = search_plan(g1, g2, 0)
    out = []
    cut2 = f32(cutoff * cutoff)
    for c1, c2, wrap in plan:
        _pair_double(cut2, g1, g2, (c1, c2, wrap), box, out, vdw, within)
        _pair_double(cut2, g1, g2, (c2, c1, wrap), box, out, vdw, within)       # swapped call, unguarded (:686-693)
    return out, g1.dims
