"""ctypes front-end for the CPU oracle (oracle/molar_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package (molar_amd/) never imports this.
Parity pinning statement: see oracle/molar_oracle.h.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

PBC_FULL = 7
PBC_NONE = 0


def build(force: bool = False) -> None:
    """Compile liboracle_f32.so / liboracle_f64.so with gcc (oracle/Makefile)."""
    targets = [os.path.join(_HERE, f"liboracle_{p}.so") for p in ("f32", "f64")]
    srcs = [os.path.join(_HERE, f) for f in ("molar_oracle.c", "molar_oracle.h")]
    newest = max(os.path.getmtime(s) for s in srcs)
    if force or any(not os.path.exists(t) or os.path.getmtime(t) < newest for t in targets):
        subprocess.run(["make", "-C", _HERE, "-B"], check=True, capture_output=True)


class _Pairs(C.Structure):
    _fields_ = [
        ("n", C.c_size_t),
        ("i", C.POINTER(C.c_uint64)),
        ("j", C.POINTER(C.c_uint64)),
        ("d", C.c_void_p),
        ("dims", C.c_uint64 * 3),
        ("plan_len", C.c_size_t),
        ("threads_used", C.c_int),
    ]


def pbc_mask(dims) -> int:
    """PbcDims::new(x,y,z) (periodic_box.rs:101-107): bit d <=> dimension d."""
    if isinstance(dims, (int, np.integer)):
        return int(dims)
    return (1 if dims[0] else 0) | (2 if dims[1] else 0) | (4 if dims[2] else 0)


class Oracle:
    """One precision of the oracle: 'f32' (MolAR default) or 'f64' (MolAR `f64` feature)."""

    def __init__(self, precision: str = "f32"):
        build()
        self.precision = precision
        self.real = np.float32 if precision == "f32" else np.float64
        self.creal = C.c_float if precision == "f32" else C.c_double
        self.lib = C.CDLL(os.path.join(_HERE, f"liboracle_{precision}.so"))
        assert self.lib.orc_sizeof_real() == np.dtype(self.real).itemsize

        class Box(C.Structure):
            _fields_ = [
                ("m", self.creal * 9),
                ("inv", self.creal * 9),
                ("nshift", C.c_int32),
                ("shifts", self.creal * 78),
            ]

        self.Box = Box
        L = self.lib
        R = self.creal
        P = C.c_void_p
        L.orc_distance_squared.restype = R
        L.orc_distance.restype = R
        for name in (
            "orc_search_single", "orc_search_single_pbc", "orc_search_double", "orc_search_double_pbc",
            "orc_search_double_vdw", "orc_search_double_vdw_pbc", "orc_search_within", "orc_search_within_pbc",
            "orc_brute_single", "orc_brute_double",
        ):
            getattr(L, name).restype = C.POINTER(_Pairs)
        L.orc_search_single.argtypes = [R, P, P, C.c_size_t, C.c_int]
        L.orc_search_single_pbc.argtypes = [R, P, P, C.c_size_t, P, C.c_uint8, C.c_int]
        L.orc_search_double.argtypes = [R, P, P, C.c_size_t, P, P, C.c_size_t, C.c_int]
        L.orc_search_double_pbc.argtypes = [R, P, P, C.c_size_t, P, P, C.c_size_t, P, C.c_uint8, C.c_int]
        L.orc_search_double_vdw.argtypes = [P, C.c_size_t, P, C.c_size_t, P, P, C.c_int]
        L.orc_search_double_vdw_pbc.argtypes = [P, C.c_size_t, P, C.c_size_t, P, P, P, C.c_uint8, C.c_int]
        L.orc_search_within.argtypes = [R, P, P, C.c_size_t, P, P, C.c_size_t, P, P, C.c_int]
        L.orc_search_within_pbc.argtypes = [R, P, P, C.c_size_t, P, P, C.c_size_t, P, C.c_uint8, C.c_int]
        L.orc_brute_single.argtypes = [R, P, P, C.c_size_t, P, C.c_uint8]
        L.orc_brute_double.argtypes = [R, P, P, C.c_size_t, P, P, C.c_size_t, P, C.c_uint8]
        L.orc_pairs_free.argtypes = [C.POINTER(_Pairs)]
        L.orc_box_from_vectors_angles.argtypes = [R, R, R, R, R, R, P]
        L.orc_histogram_add.argtypes = [R, R, C.c_size_t, P, C.c_size_t, P]
        L.orc_bounding_box_single.argtypes = [R, P, C.c_size_t, P, P]
        L.orc_bounding_box_double.argtypes = [R, P, C.c_size_t, P, C.c_size_t, P, P]

    # ------------------------------------------------------------------ helpers
    def arr(self, a, shape=None):
        a = np.ascontiguousarray(a, dtype=self.real)
        if shape is not None:
            a = a.reshape(shape)
        return a

    @staticmethod
    def _p(a):
        return None if a is None else a.ctypes.data_as(C.c_void_p)

    @staticmethod
    def _ids(ids):
        return None if ids is None else np.ascontiguousarray(ids, dtype=np.uint64)

    def _take(self, ptr, within=False):
        if not ptr:
            raise RuntimeError("oracle search refused the grid (too many cells)")
        pr = ptr.contents
        n = pr.n
        i = np.ctypeslib.as_array(pr.i, shape=(max(n, 1),))[:n].copy()
        res = {"i": i, "dims": tuple(int(x) for x in pr.dims), "plan_len": int(pr.plan_len), "threads_used": int(pr.threads_used)}
        if not within:
            j = np.ctypeslib.as_array(pr.j, shape=(max(n, 1),))[:n].copy()
            dptr = C.cast(pr.d, C.POINTER(self.creal))
            d = np.ctypeslib.as_array(dptr, shape=(max(n, 1),))[:n].copy()
            res["j"] = j
            res["d"] = d
        self.lib.orc_pairs_free(ptr)
        return res

    # ------------------------------------------------------------------ box
    def box_from_matrix(self, m):
        """m: 3x3 array-like with COLUMNS = box vectors (PeriodicBox::from_matrix)."""
        m = np.asarray(m, dtype=self.real).reshape(3, 3)
        flat = np.ascontiguousarray(m.T).reshape(9)  # column-major
        b = self.Box()
        rc = self.lib.orc_box_from_matrix(self._p(flat), C.byref(b))
        if rc:
            raise ValueError(f"box_from_matrix failed: status {rc}")
        return b

    def box_from_vectors_angles(self, a, b, c, alpha, beta, gamma):
        bx = self.Box()
        rc = self.lib.orc_box_from_vectors_angles(a, b, c, alpha, beta, gamma, C.byref(bx))
        if rc:
            raise ValueError(f"box_from_vectors_angles failed: status {rc}")
        return bx

    def box_matrix(self, b):
        return np.array(b.m, dtype=self.real).reshape(3, 3).T.copy()

    def box_inv(self, b):
        return np.array(b.inv, dtype=self.real).reshape(3, 3).T.copy()

    def box_shifts(self, b):
        return np.array(b.shifts, dtype=self.real).reshape(26, 3)[: b.nshift].copy()

    def shortest_vector_dims(self, b, v, dims=PBC_FULL):
        v = self.arr(v)
        out = np.zeros(3, self.real)
        self.lib.orc_shortest_vector_dims(C.byref(b), self._p(v), C.c_uint8(pbc_mask(dims)), self._p(out))
        return out

    def distance(self, b, p1, p2, dims=PBC_FULL):
        p1, p2 = self.arr(p1), self.arr(p2)
        return float(self.lib.orc_distance(C.byref(b), self._p(p1), self._p(p2), C.c_uint8(pbc_mask(dims))))

    def distance_squared(self, b, p1, p2, dims=PBC_FULL):
        p1, p2 = self.arr(p1), self.arr(p2)
        return float(self.lib.orc_distance_squared(C.byref(b), self._p(p1), self._p(p2), C.c_uint8(pbc_mask(dims))))

    def closest_image_dims(self, b, p, target, dims=PBC_FULL):
        p, target = self.arr(p), self.arr(target)
        out = np.zeros(3, self.real)
        self.lib.orc_closest_image_dims(C.byref(b), self._p(p), self._p(target), C.c_uint8(pbc_mask(dims)), self._p(out))
        return out

    def lab_extents(self, b):
        out = np.zeros(3, self.real)
        self.lib.orc_lab_extents(C.byref(b), self._p(out))
        return out

    def box_extents(self, b):
        out = np.zeros(3, self.real)
        self.lib.orc_box_extents(C.byref(b), self._p(out))
        return out

    def wrap_point(self, b, p):
        p = self.arr(p)
        out = np.zeros(3, self.real)
        self.lib.orc_wrap_point(C.byref(b), self._p(p), self._p(out))
        return out

    def is_inside(self, b, p):
        p = self.arr(p)
        return bool(self.lib.orc_is_inside(C.byref(b), self._p(p)))

    # ------------------------------------------------------------------ searches
    def search_single(self, cutoff, pos, ids=None, nthreads=1):
        pos = self.arr(pos, (-1, 3)); ids = self._ids(ids)
        return self._take(self.lib.orc_search_single(cutoff, self._p(pos), self._p(ids), len(pos), nthreads))

    def search_single_pbc(self, cutoff, pos, box, dims=PBC_FULL, ids=None, nthreads=1):
        pos = self.arr(pos, (-1, 3)); ids = self._ids(ids)
        return self._take(self.lib.orc_search_single_pbc(
            cutoff, self._p(pos), self._p(ids), len(pos), C.cast(C.byref(box), C.c_void_p), pbc_mask(dims), nthreads))

    def time_search_single_pbc(self, cutoff, pos, box, dims=PBC_FULL, nthreads=1):
        """Benchmark helper: runs orc_search_single_pbc and returns (seconds spent inside the C call, number of pairs).
        The result stays on the C side and is freed after the clock has stopped: no numpy copies in the timed span."""
        import time
        pos = self.arr(pos, (-1, 3))
        pp, bp, m = self._p(pos), C.cast(C.byref(box), C.c_void_p), pbc_mask(dims)
        t0 = time.perf_counter()
        ptr = self.lib.orc_search_single_pbc(cutoff, pp, None, len(pos), bp, m, nthreads)
        dt = time.perf_counter() - t0
        if not ptr:
            raise RuntimeError("oracle search refused the grid (too many cells)")
        n = int(ptr.contents.n)
        self.lib.orc_pairs_free(ptr)
        return dt, n

    def search_double(self, cutoff, pos1, pos2, ids1=None, ids2=None, nthreads=1):
        pos1 = self.arr(pos1, (-1, 3)); pos2 = self.arr(pos2, (-1, 3)); ids1 = self._ids(ids1); ids2 = self._ids(ids2)
        return self._take(self.lib.orc_search_double(
            cutoff, self._p(pos1), self._p(ids1), len(pos1), self._p(pos2), self._p(ids2), len(pos2), nthreads))

    def search_double_pbc(self, cutoff, pos1, pos2, box, dims=PBC_FULL, ids1=None, ids2=None, nthreads=1):
        pos1 = self.arr(pos1, (-1, 3)); pos2 = self.arr(pos2, (-1, 3)); ids1 = self._ids(ids1); ids2 = self._ids(ids2)
        return self._take(self.lib.orc_search_double_pbc(
            cutoff, self._p(pos1), self._p(ids1), len(pos1), self._p(pos2), self._p(ids2), len(pos2),
            C.cast(C.byref(box), C.c_void_p), pbc_mask(dims), nthreads))

    def search_double_vdw(self, pos1, pos2, vdw1, vdw2, nthreads=1):
        pos1 = self.arr(pos1, (-1, 3)); pos2 = self.arr(pos2, (-1, 3)); vdw1 = self.arr(vdw1); vdw2 = self.arr(vdw2)
        return self._take(self.lib.orc_search_double_vdw(
            self._p(pos1), len(pos1), self._p(pos2), len(pos2), self._p(vdw1), self._p(vdw2), nthreads))

    def search_double_vdw_pbc(self, pos1, pos2, vdw1, vdw2, box, dims=PBC_FULL, nthreads=1):
        pos1 = self.arr(pos1, (-1, 3)); pos2 = self.arr(pos2, (-1, 3)); vdw1 = self.arr(vdw1); vdw2 = self.arr(vdw2)
        return self._take(self.lib.orc_search_double_vdw_pbc(
            self._p(pos1), len(pos1), self._p(pos2), len(pos2), self._p(vdw1), self._p(vdw2),
            C.cast(C.byref(box), C.c_void_p), pbc_mask(dims), nthreads))

    def search_within(self, cutoff, pos1, pos2, lower, upper, ids1=None, ids2=None, nthreads=1):
        pos1 = self.arr(pos1, (-1, 3)); pos2 = self.arr(pos2, (-1, 3)); ids1 = self._ids(ids1); ids2 = self._ids(ids2)
        lower = self.arr(lower); upper = self.arr(upper)
        return self._take(self.lib.orc_search_within(
            cutoff, self._p(pos1), self._p(ids1), len(pos1), self._p(pos2), self._p(ids2), len(pos2),
            self._p(lower), self._p(upper), nthreads), within=True)

    def search_within_pbc(self, cutoff, pos1, pos2, box, dims=PBC_FULL, ids1=None, ids2=None, nthreads=1):
        pos1 = self.arr(pos1, (-1, 3)); pos2 = self.arr(pos2, (-1, 3)); ids1 = self._ids(ids1); ids2 = self._ids(ids2)
        return self._take(self.lib.orc_search_within_pbc(
            cutoff, self._p(pos1), self._p(ids1), len(pos1), self._p(pos2), self._p(ids2), len(pos2),
            C.cast(C.byref(box), C.c_void_p), pbc_mask(dims), nthreads), within=True)

    def brute_single(self, cutoff, pos, box=None, dims=PBC_FULL, ids=None):
        pos = self.arr(pos, (-1, 3)); ids = self._ids(ids)
        bp = C.cast(C.byref(box), C.c_void_p) if box is not None else None
        return self._take(self.lib.orc_brute_single(cutoff, self._p(pos), self._p(ids), len(pos), bp, pbc_mask(dims)))

    def brute_double(self, cutoff, pos1, pos2, box=None, dims=PBC_FULL, ids1=None, ids2=None):
        pos1 = self.arr(pos1, (-1, 3)); pos2 = self.arr(pos2, (-1, 3)); ids1 = self._ids(ids1); ids2 = self._ids(ids2)
        bp = C.cast(C.byref(box), C.c_void_p) if box is not None else None
        return self._take(self.lib.orc_brute_double(
            cutoff, self._p(pos1), self._p(ids1), len(pos1), self._p(pos2), self._p(ids2), len(pos2), bp, pbc_mask(dims)))

    def bounding_box_single(self, cutoff, pos):
        pos = self.arr(pos, (-1, 3))
        lo = np.zeros(3, self.real); up = np.zeros(3, self.real)
        self.lib.orc_bounding_box_single(cutoff, self._p(pos), len(pos), self._p(lo), self._p(up))
        return lo, up

    # ------------------------------------------------------------------ measure / modify
    def _sel(self, xyz, idx):
        xyz = self.arr(xyz, (-1, 3))
        idx = self._ids(idx)
        n = len(xyz) if idx is None else len(idx)
        return xyz, idx, n

    def _bp(self, box):
        return C.cast(C.byref(box), C.c_void_p) if box is not None else None

    def min_max(self, xyz, idx=None):
        xyz, idx, n = self._sel(xyz, idx)
        lo = np.zeros(3, self.real); up = np.zeros(3, self.real)
        self.lib.orc_min_max(self._p(xyz), self._p(idx), C.c_size_t(n), self._p(lo), self._p(up))
        return lo, up

    def center_of_geometry(self, xyz, idx=None):
        xyz, idx, n = self._sel(xyz, idx)
        out = np.zeros(3, self.real)
        self.lib.orc_center_of_geometry(self._p(xyz), self._p(idx), C.c_size_t(n), self._p(out))
        return out

    def _chk(self, rc):
        if rc:
            raise MeasureError(rc)

    def center_of_mass(self, xyz, mass, idx=None):
        xyz, idx, n = self._sel(xyz, idx); mass = self.arr(mass)
        out = np.zeros(3, self.real)
        self._chk(self.lib.orc_center_of_mass(self._p(xyz), self._p(idx), C.c_size_t(n), self._p(mass), self._p(out)))
        return out

    def center_of_geometry_pbc_dims(self, xyz, box, dims=PBC_FULL, idx=None):
        xyz, idx, n = self._sel(xyz, idx)
        out = np.zeros(3, self.real)
        self._chk(self.lib.orc_center_of_geometry_pbc_dims(
            self._p(xyz), self._p(idx), C.c_size_t(n), self._bp(box), C.c_uint8(pbc_mask(dims)), self._p(out)))
        return out

    def center_of_mass_pbc_dims(self, xyz, mass, box, dims=PBC_FULL, idx=None):
        xyz, idx, n = self._sel(xyz, idx); mass = self.arr(mass)
        out = np.zeros(3, self.real)
        self._chk(self.lib.orc_center_of_mass_pbc_dims(
            self._p(xyz), self._p(idx), C.c_size_t(n), self._p(mass), self._bp(box), C.c_uint8(pbc_mask(dims)),
            self._p(out)))
        return out

    def gyration(self, xyz, mass, idx=None):
        xyz, idx, n = self._sel(xyz, idx); mass = self.arr(mass)
        out = np.zeros(1, self.real)
        self._chk(self.lib.orc_gyration(self._p(xyz), self._p(idx), C.c_size_t(n), self._p(mass), self._p(out)))
        return float(out[0])

    def gyration_pbc(self, xyz, mass, box, idx=None):
        xyz, idx, n = self._sel(xyz, idx); mass = self.arr(mass)
        out = np.zeros(1, self.real)
        self._chk(self.lib.orc_gyration_pbc(
            self._p(xyz), self._p(idx), C.c_size_t(n), self._p(mass), self._bp(box), self._p(out)))
        return float(out[0])

    def inertia(self, xyz, mass, idx=None, box=None):
        xyz, idx, n = self._sel(xyz, idx); mass = self.arr(mass)
        mom = np.zeros(3, self.real); axes = np.zeros(9, self.real)
        if box is None:
            self._chk(self.lib.orc_inertia(self._p(xyz), self._p(idx), C.c_size_t(n), self._p(mass), self._p(mom),
                                           self._p(axes)))
        else:
            self._chk(self.lib.orc_inertia_pbc(self._p(xyz), self._p(idx), C.c_size_t(n), self._p(mass),
                                               self._bp(box), self._p(mom), self._p(axes)))
        return mom, axes.reshape(3, 3).T.copy()

    def inertia_tensor(self, xyz, mass, idx=None, box=None):
        xyz, idx, n = self._sel(xyz, idx); mass = self.arr(mass)
        t = np.zeros(9, self.real)
        self._chk(self.lib.orc_inertia_tensor(self._p(xyz), self._p(idx), C.c_size_t(n), self._p(mass),
                                              self._bp(box), self._p(t)))
        return t.reshape(3, 3).T.copy()

    def rmsd(self, xyz1, xyz2, idx1=None, idx2=None):
        xyz1, idx1, n1 = self._sel(xyz1, idx1); xyz2, idx2, n2 = self._sel(xyz2, idx2)
        out = np.zeros(1, self.real)
        self._chk(self.lib.orc_rmsd(self._p(xyz1), self._p(idx1), C.c_size_t(n1), self._p(xyz2), self._p(idx2),
                                    C.c_size_t(n2), self._p(out)))
        return float(out[0])

    def rmsd_mw(self, xyz1, mass1, xyz2, idx1=None, idx2=None):
        xyz1, idx1, n1 = self._sel(xyz1, idx1); xyz2, idx2, n2 = self._sel(xyz2, idx2); mass1 = self.arr(mass1)
        out = np.zeros(1, self.real)
        self._chk(self.lib.orc_rmsd_mw(self._p(xyz1), self._p(idx1), C.c_size_t(n1), self._p(mass1), self._p(xyz2),
                                       self._p(idx2), C.c_size_t(n2), self._p(out)))
        return float(out[0])

    def fit_transform(self, xyz1, mass1, xyz2, mass2, idx1=None, idx2=None):
        """Returns (R 3x3 row/col natural numpy matrix, t) with p -> R @ p + t."""
        xyz1, idx1, n1 = self._sel(xyz1, idx1); xyz2, idx2, n2 = self._sel(xyz2, idx2)
        mass1 = self.arr(mass1); mass2 = self.arr(mass2)
        R = np.zeros(9, self.real); t = np.zeros(3, self.real)
        self._chk(self.lib.orc_fit_transform(
            self._p(xyz1), self._p(idx1), C.c_size_t(n1), self._p(mass1), self._p(xyz2), self._p(idx2),
            C.c_size_t(n2), self._p(mass2), self._p(R), self._p(t)))
        return R.reshape(3, 3).T.copy(), t

    def fit_transform_at_origin(self, xyz1, mass1, xyz2, idx1=None, idx2=None):
        xyz1, idx1, n1 = self._sel(xyz1, idx1); xyz2, idx2, n2 = self._sel(xyz2, idx2); mass1 = self.arr(mass1)
        R = np.zeros(9, self.real); t = np.zeros(3, self.real)
        self._chk(self.lib.orc_fit_transform_at_origin(
            self._p(xyz1), self._p(idx1), C.c_size_t(n1), self._p(mass1), self._p(xyz2), self._p(idx2),
            C.c_size_t(n2), self._p(R), self._p(t)))
        return R.reshape(3, 3).T.copy(), t

    def apply_transform(self, xyz, R, t, idx=None):
        """Returns a transformed COPY of xyz (selected atoms moved)."""
        xyz, idx, n = self._sel(xyz, idx)
        xyz = xyz.copy()
        Rf = np.ascontiguousarray(np.asarray(R, dtype=self.real).T).reshape(9); t = self.arr(t)
        self.lib.orc_apply_transform(self._p(xyz), self._p(idx), C.c_size_t(n), self._p(Rf), self._p(t))
        return xyz

    def translate(self, xyz, shift, idx=None):
        """Returns a translated COPY of xyz (modify.rs:16-23)."""
        xyz, idx, n = self._sel(xyz, idx)
        xyz = xyz.copy()
        sh = self.arr(shift)
        self.lib.orc_translate(self._p(xyz), self._p(idx), C.c_size_t(n), self._p(sh))
        return xyz

    def unwrap_simple_dim(self, xyz, box, dims=PBC_FULL, idx=None):
        xyz, idx, n = self._sel(xyz, idx)
        xyz = xyz.copy()
        self._chk(self.lib.orc_unwrap_simple_dim(self._p(xyz), self._p(idx), C.c_size_t(n), self._bp(box),
                                                 C.c_uint8(pbc_mask(dims))))
        return xyz

    def unwrap_connectivity(self, xyz, box, cutoff, dims=PBC_FULL, idx=None, nthreads=1):
        """modify.rs:72-131.  Returns (unwrapped copy of xyz, list of groups of LOCAL indices)."""
        xyz, idx, n = self._sel(xyz, idx)
        xyz = xyz.copy()
        goff = np.zeros(n + 1, np.uint64); gids = np.zeros(max(n, 1), np.uint64)
        ng = C.c_size_t(0)
        fn = self.lib.orc_unwrap_connectivity_dim
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, self.creal, C.c_uint8, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        self._chk(fn(self._p(xyz), self._p(idx), n, self._bp(box), cutoff, pbc_mask(dims), self._p(goff), self._p(gids), C.byref(ng), nthreads))
        g = int(ng.value)
        return xyz, [gids[int(goff[k]):int(goff[k + 1])].copy() for k in range(g)]

    def lipid_tail_order(self, xyz, order_type, normals, bond_orders, idx=None):
        xyz, idx, n = self._sel(xyz, idx)
        normals = self.arr(normals, (-1, 3))
        bo = np.ascontiguousarray(bond_orders, dtype=np.uint8)
        out = np.zeros(max(n - 2, 1), self.real)
        self._chk(self.lib.orc_lipid_tail_order(
            self._p(xyz), self._p(idx), C.c_size_t(n), C.c_int(order_type), self._p(normals),
            C.c_size_t(len(normals)), self._p(bo), C.c_size_t(len(bo)), self._p(out)))
        return out[: n - 2]

    def membrane_smooth(self, box, head, normals, valid, patch_off, patch_ids):
        """One Membrane::smooth iteration (molar_membrane/src/lib.rs:661-812).  Returns a dict; inputs are
        copied, not modified."""
        head = self.arr(head, (-1, 3)).copy(); normals = self.arr(normals, (-1, 3)).copy()
        K = len(head)
        valid = np.ascontiguousarray(valid, dtype=np.uint8).copy()
        poff = np.ascontiguousarray(patch_off, dtype=np.uint64); pids = np.ascontiguousarray(patch_ids, dtype=np.uint64)
        E = int(poff[-1]); S = E + 4 * K
        r = self.real
        out = dict(coefs=np.zeros((K, 6), r), mean_curv=np.full(K, -100.0, r), gauss_curv=np.full(K, -100.0, r),
                   princ_curvs=np.zeros((K, 2), r), princ_dirs=np.zeros((K, 2, 3), r), area=np.zeros(K, r),
                   nvert=np.zeros(K, np.uint32), neib_ids=np.zeros(max(S, 1), np.uint64),
                   voro=np.zeros((max(S, 1), 3), r), fitted=np.zeros((max(E, 1), 3), r))
        self._chk(self.lib.orc_membrane_smooth(
            C.byref(box), C.c_size_t(K), self._p(head), self._p(normals), self._p(valid), self._p(poff), self._p(pids),
            self._p(out["coefs"]), self._p(out["mean_curv"]), self._p(out["gauss_curv"]), self._p(out["princ_curvs"]),
            self._p(out["princ_dirs"]), self._p(out["area"]), self._p(out["nvert"]), self._p(out["neib_ids"]),
            self._p(out["voro"]), self._p(out["fitted"])))
        out.update(head=head, normals=normals, valid=valid)
        return out

    # ------------------------------------------------------------------ XTC codec (xtc_oracle.c)
    def xtc_index(self, data: bytes):
        buf = np.frombuffer(data, np.uint8)
        self.lib.orc_xtc_index.restype = C.c_size_t
        n = self.lib.orc_xtc_index(self._p(buf), C.c_size_t(len(buf)), None, C.c_size_t(0))
        off = np.zeros(max(n, 1), np.uint64)
        self.lib.orc_xtc_index(self._p(buf), C.c_size_t(len(buf)), self._p(off), C.c_size_t(n))
        return off[:n]

    def xtc_header(self, data: bytes, offset=0):
        buf = np.frombuffer(data, np.uint8)[int(offset):]
        self.lib.orc_xtc_frame_header.restype = C.c_size_t
        nat, step, t, prec = C.c_int32(0), C.c_int32(0), C.c_float(0), C.c_float(0)
        box = np.zeros(9, np.float32)
        ln = self.lib.orc_xtc_frame_header(self._p(buf), C.c_size_t(len(buf)), C.byref(nat), C.byref(step), C.byref(t),
                                           self._p(box), C.byref(prec))
        if not ln:
            raise ValueError("not an XTC frame header")
        return dict(natoms=nat.value, step=step.value, time=t.value, box9=box, precision=prec.value, length=int(ln))

    def xtc_decode(self, data: bytes, offset=0):
        h = self.xtc_header(data, offset)
        buf = np.frombuffer(data, np.uint8)[int(offset):]
        xyz = np.zeros((h["natoms"], 3), np.float32)
        rc = self.lib.orc_xtc_decode_frame(self._p(buf), C.c_size_t(len(buf)), self._p(xyz))
        if rc:
            raise ValueError(f"xtc decode failed: {rc}")
        return xyz, h

    def xtc_encode(self, xyz, box9, step=0, time=0.0, precision=1000.0, magic=1995) -> bytes:
        xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        box9 = np.ascontiguousarray(box9, np.float32).reshape(9)
        cap = 128 + 16 * len(xyz)
        out = np.zeros(cap, np.uint8)
        self.lib.orc_xtc_encode_frame.restype = C.c_size_t
        n = self.lib.orc_xtc_encode_frame(self._p(xyz), C.c_int32(len(xyz)), self._p(box9), C.c_int32(step), C.c_float(time),
                                          C.c_float(precision), C.c_uint32(magic), self._p(out), C.c_size_t(cap))
        if not n:
            raise ValueError("xtc encode failed")
        return out[:n].tobytes()

    def histogram_add(self, minv, maxv, nbins, vals):
        vals = self.arr(vals)
        bins = np.zeros(nbins, self.real)
        self.lib.orc_histogram_add(minv, maxv, nbins, self._p(vals), len(vals), self._p(bins))
        return bins


class MeasureError(RuntimeError):
    """Mirrors MeasureError / PeriodicBoxError / LipidOrderError status codes (molar_oracle.h)."""

    def __init__(self, code):
        super().__init__(f"measure error status {code}")
        self.code = code
