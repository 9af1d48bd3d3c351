"""Host-side mirror of the pymolar surface for the accelerated path, over the C ABI.

Mirrors (same names, argument meaning, error behaviour) the Python front-end of the reference
for this path — molar_python/python/pymolar/molar.pyi:130-213 and molar_python/src/lib.rs:
    distance_search(cutoff | "vdw", sel1, sel2=None, dims=None) -> (pairs[N,2], dist[N])
    fit_transform(sel1, sel2), fit_transform_at_origin, rmsd(sel1, sel2) [rmsd_py], rmsd_mw
    Sel.com(dims) / cog(dims) / gyration() / inertia() / min_max() / apply_transform / unwrap_simple
    PeriodicBox(vectors, angles) / PeriodicBox.from_matrix, shortest_vector, ...
Only the numeric path is here: no selection language, no file IO (out of scope).
Arrays may be numpy (host) or torch CUDA tensors (device-resident, used in place).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import (MEMBRANE_ARRAYS, Box, MembraneDesc, MembraneOut, MembraneView, MolarHipError, SearchDesc, SearchDescF64,
                   check)

PBC_FULL = 7
PBC_NONE = 0

SEARCH_SINGLE, SEARCH_DOUBLE, SEARCH_WITHIN, SEARCH_DOUBLE_VDW = 0, 1, 2, 3


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


def _addr(x):
    """(address, keepalive) of a numpy array or a torch tensor; None -> (None, None)."""
    if x is None:
        return None, None
    if _is_torch(x):
        assert x.is_contiguous()
        return x.data_ptr(), x
    return x.ctypes.data, x


class _DevicePointer:
    """__cuda_array_interface__ holder for memory the engine owns (result buffers of the resident search)."""

    def __init__(self, addr, count, typestr):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": typestr, "data": (int(addr), False), "version": 2}


def device_view(addr, shape, dtype):
    """A torch CUDA tensor over `addr` (no copy).  The memory stays owned by the engine: the view is valid until the
    next search on the same context (or result set, for the begin/end form) reuses the buffer."""
    import torch
    count = int(np.prod(shape))
    if count == 0:
        return torch.empty(shape, dtype=dtype, device="cuda")
    typestr = {torch.int32: "<i4", torch.float32: "<f4", torch.int64: "<i8", torch.uint8: "|u1"}[dtype]
    return torch.as_tensor(_DevicePointer(addr, count, typestr), device="cuda").view(*shape)


def _f32(x, shape=None):
    if x is None:
        return None
    if _is_torch(x):
        import torch
        assert x.dtype == torch.float32
        return x.contiguous()
    a = np.ascontiguousarray(x, dtype=np.float32)
    return a.reshape(shape) if shape is not None else a


def _u64(x):
    if x is None:
        return None
    if _is_torch(x):
        import torch
        if x.dtype == torch.int64:
            return x.contiguous()     # non-negative int64 has the same bits as uint64
        raise TypeError("index tensors must be int64")
    return np.ascontiguousarray(x, dtype=np.uint64)


def pbc_mask(dims) -> int:
    """PbcDims::new (periodic_box.rs:101-107); None -> all False like the Python front-end."""
    if dims is None:
        return 0
    if isinstance(dims, (int, np.integer)):
        return int(dims) & 7
    return (1 if dims[0] else 0) | (2 if dims[1] else 0) | (4 if dims[2] else 0)


class PeriodicBox:
    """periodic_box.rs:146-435 (host arithmetic of the engine, identical to its device code)."""

    def __init__(self, vectors=None, angles=None, _box=None):
        lib = _lib.load()
        self._b = Box()
        if _box is not None:
            self._b = _box
        else:
            v, a = vectors, angles
            check(lib.molar_hip_box_from_vectors_angles(v[0], v[1], v[2], a[0], a[1], a[2], C.byref(self._b)))

    @classmethod
    def from_matrix(cls, m):
        """m: 3x3 with COLUMNS = box vectors a,b,c (periodic_box.rs:7-13)."""
        lib = _lib.load()
        m = np.asarray(m, dtype=np.float32).reshape(3, 3)
        flat = np.ascontiguousarray(m.T).reshape(9)
        b = Box()
        check(lib.molar_hip_box_from_matrix(flat.ctypes.data, C.byref(b)))
        return cls(_box=b)

    def get_matrix(self):
        return np.array(self._b.m, dtype=np.float32).reshape(3, 3).T.copy()

    def colmajor9(self):
        return np.array(self._b.m, dtype=np.float32)

    def shortest_vector(self, v, dims=PBC_FULL):
        v = np.ascontiguousarray(v, dtype=np.float32)
        out = np.zeros(3, np.float32)
        _lib.load().molar_hip_box_shortest_vector(C.byref(self._b), v.ctypes.data, pbc_mask(dims), out.ctypes.data)
        return out

    def closest_image(self, point, target, dims=PBC_FULL):
        point = np.asarray(point, np.float32); target = np.asarray(target, np.float32)
        return target + self.shortest_vector(point - target, dims)

    def distance(self, p1, p2, dims=PBC_FULL):
        p1 = np.asarray(p1, np.float32); p2 = np.asarray(p2, np.float32)
        s = self.shortest_vector(p2 - p1, dims)
        return float(np.sqrt(np.float32(s[0] * s[0] + s[1] * s[1]) + np.float32(s[2] * s[2])))

    def get_lab_extents(self):
        out = np.zeros(3, np.float32)
        _lib.load().molar_hip_box_lab_extents(C.byref(self._b), out.ctypes.data)
        return out

    def _v3(self, fn, v):
        v = np.ascontiguousarray(v, dtype=np.float32)
        out = np.zeros(3, np.float32)
        getattr(_lib.load(), fn)(C.byref(self._b), v.ctypes.data, out.ctypes.data)
        return out

    def to_box_coords(self, v):
        return self._v3("molar_hip_box_to_box_coords", v)

    def to_lab_coords(self, v):
        return self._v3("molar_hip_box_to_lab_coords", v)

    def wrap_point(self, p):
        return self._v3("molar_hip_box_wrap_point", p)

    def is_inside(self, p):
        p = np.ascontiguousarray(p, dtype=np.float32)
        return bool(_lib.load().molar_hip_box_is_inside(C.byref(self._b), p.ctypes.data))

    def get_box_extents(self):
        out = np.zeros(3, np.float32)
        _lib.load().molar_hip_box_extents(C.byref(self._b), out.ctypes.data)
        return out

    def is_triclinic(self):
        m = self._b.m
        return any(m[k] != 0.0 for k in (1, 2, 3, 5, 6, 7))

    @property
    def n_tric_corrections(self):
        return int(self._b.nshift)


class Engine:
    """One molar_hip_ctx: a GPU, a stream and reusable device buffers.

    The context runs on a stream of its own unless `stream` is given (e.g. `torch.cuda.current_stream().cuda_stream`).
    Device tensors handed to it are read where they are, on THAT stream: a tensor another stream is still producing (a
    `.clone()`, a `torch.randn(...)` enqueued a moment ago) has to be complete first - `torch.cuda.synchronize()`, or one
    shared stream."""

    def __init__(self, device: int = 0, stream=None):
        self.lib = _lib.load()
        self.ctx = self.lib.molar_hip_create(device)
        if not self.ctx:
            raise MolarHipError(100, _lib.last_error())
        self.device = device
        if stream is not None:
            check(self.lib.molar_hip_set_stream(self.ctx, C.c_void_p(stream)))

    def close(self):
        if getattr(self, "ctx", None):
            for ref in list(getattr(self, "_plans", ())):        # plans hold the context: they go first
                plan = ref()
                if plan is not None:
                    plan.close()
            self.lib.molar_hip_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _adopt(self, plan):
        import weakref
        if not hasattr(self, "_plans"):
            self._plans = []
        self._plans = [r for r in self._plans if r() is not None] + [weakref.ref(plan)]

    def synchronize(self):
        check(self.lib.molar_hip_synchronize(self.ctx))

    PROFILE_CLASSES = ("grid_build", "pair_count", "offset_scan", "pair_fill", "measure", "search_frame")

    def profile_enable(self, on=True):
        """True / 1: a span per kernel class; 2: one span per resident search (class search_frame); False: off."""
        check(self.lib.molar_hip_profile_enable(self.ctx, int(on)))

    def profile_read(self):
        """{class: (milliseconds, launches)} accumulated since the last read (HIP events)."""
        ms = (C.c_float * 8)(); ln = (C.c_uint64 * 8)()
        check(self.lib.molar_hip_profile_read(self.ctx, ms, ln))
        return {name: (float(ms[k]), int(ln[k])) for k, name in enumerate(self.PROFILE_CLASSES)}

    # ------------------------------------------------------------ search
    def _search_desc(self, kind, cutoff, xyz1, idx1=None, xyz2=None, idx2=None, box=None, pbc=0, vdw1=None,
                     vdw2=None, ids_local=False, lower=None, upper=None):
        xyz1 = _f32(xyz1); xyz2 = _f32(xyz2); idx1 = _u64(idx1); idx2 = _u64(idx2)
        vdw1 = _f32(vdw1); vdw2 = _f32(vdw2)
        d = SearchDesc()
        d.kind = kind
        d.cutoff = float(cutoff) if cutoff is not None else 0.0
        keep = []
        for name, arr in (("xyz1", xyz1), ("idx1", idx1), ("xyz2", xyz2), ("idx2", idx2), ("vdw1", vdw1),
                          ("vdw2", vdw2)):
            a, k = _addr(arr)
            setattr(d, name, a)
            keep.append(k)
        d.natoms1 = 0 if xyz1 is None else xyz1.shape[0] if xyz1.ndim == 2 else xyz1.shape[0] // 3
        d.natoms2 = 0 if xyz2 is None else xyz2.shape[0] if xyz2.ndim == 2 else xyz2.shape[0] // 3
        d.n1 = 0 if idx1 is None else idx1.shape[0]
        d.n2 = 0 if idx2 is None else idx2.shape[0]
        d.ids_local = 1 if ids_local else 0
        if box is not None:
            b9 = box.colmajor9() if isinstance(box, PeriodicBox) else np.ascontiguousarray(
                np.asarray(box, np.float32).reshape(3, 3).T).reshape(9)
            keep.append(b9)
            d.box9 = b9.ctypes.data
        d.pbc = pbc_mask(pbc)
        if lower is not None:
            lo = np.ascontiguousarray(lower, np.float32); up = np.ascontiguousarray(upper, np.float32)
            keep += [lo, up]
            d.lower3 = lo.ctypes.data
            d.upper3 = up.ctypes.data
        return d, keep

    def search_count(self, kind, cutoff, xyz1, idx1=None, xyz2=None, idx2=None, box=None, pbc=0, vdw1=None,
                     vdw2=None, ids_local=False, lower=None, upper=None) -> int:
        d, keep = self._search_desc(kind, cutoff, xyz1, idx1, xyz2, idx2, box, pbc, vdw1, vdw2, ids_local, lower, upper)
        cnt = C.c_uint64(0)
        check(self.lib.molar_hip_search_count(self.ctx, C.byref(d), C.byref(cnt)))
        self._keep = keep
        return int(cnt.value)

    def make_search_desc(self, kind, cutoff, xyz1, **kw):
        """A reusable search description for per-frame loops: build once, point `desc.xyz1` (and xyz2) at each new
        frame (`desc.xyz1 = frame.data_ptr()`), run with search_resident_desc / search_count_desc.  Returns
        (desc, keepalive)."""
        return self._search_desc(kind, cutoff, xyz1, **kw)

    def search_resident_desc(self, desc):
        cnt = C.c_uint64(0); p = C.c_void_p(); dd = C.c_void_p()
        check(self.lib.molar_hip_search_resident(self.ctx, C.byref(desc), C.byref(cnt), C.byref(p), C.byref(dd)))
        return int(cnt.value), p.value, dd.value

    def search_cell_kernels(self):
        """molar_hip_search_cell_kernels: (lanes, occupied cells of the two grids) of the context's last fixed-cutoff search -
        lanes 16 / 32 = small-cell kernels, 0 = regular ones."""
        lanes = C.c_int32(-1)
        occ = (C.c_uint64 * 2)()
        check(self.lib.molar_hip_search_cell_kernels(self.ctx, C.byref(lanes), occ))
        return int(lanes.value), (int(occ[0]), int(occ[1]))

    def search_resident_planes(self, want_dist=True):
        """molar_hip_search_resident_planes: want_dist=False makes the resident searches fill the (i, j) plane only
        (DistanceSearchOutput of (usize, usize)); the distance addresses they return are then None."""
        check(self.lib.molar_hip_search_resident_planes(self.ctx, 1 if want_dist else 0))

    def search_resident_begin(self, desc):
        """Enqueue a whole resident search and return its ticket without waiting (molar_hip_search_resident_begin).
        `desc` (and what it points to) must stay alive and unchanged until search_resident_end(ticket)."""
        t = C.c_int32(-1)
        check(self.lib.molar_hip_search_resident_begin(self.ctx, C.byref(desc), C.byref(t)))
        return int(t.value)

    def search_resident_end(self, ticket):
        """Wait for the search behind `ticket`; returns (count, pairs_device_address, dist_device_address)."""
        cnt = C.c_uint64(0); p = C.c_void_p(); dd = C.c_void_p()
        check(self.lib.molar_hip_search_resident_end(self.ctx, C.c_int32(ticket), C.byref(cnt), C.byref(p), C.byref(dd)))
        return int(cnt.value), p.value, dd.value

    def search_resident(self, kind, cutoff, xyz1, idx1=None, xyz2=None, idx2=None, box=None, pbc=0, vdw1=None,
                        vdw2=None, ids_local=False, lower=None, upper=None):
        """Count + fill into engine-owned device buffers with a single host round trip
        (molar_hip_search_resident).  Returns (count, pairs_device_address, dist_device_address)."""
        d, keep = self._search_desc(kind, cutoff, xyz1, idx1, xyz2, idx2, box, pbc, vdw1, vdw2, ids_local, lower, upper)
        cnt = C.c_uint64(0); p = C.c_void_p(); dd = C.c_void_p()
        check(self.lib.molar_hip_search_resident(self.ctx, C.byref(d), C.byref(cnt), C.byref(p), C.byref(dd)))
        self._keep = keep
        return int(cnt.value), p.value, dd.value

    def search_fill(self, count):
        pairs = np.empty((count, 2), np.uint32)
        dist = np.empty(count, np.float32)
        check(self.lib.molar_hip_search_fill(self.ctx, pairs.ctypes.data, dist.ctypes.data))
        return pairs, dist

    def search_fill_usize(self, count):
        i = np.empty(count, np.uint64); j = np.empty(count, np.uint64); d = np.empty(count, np.float32)
        check(self.lib.molar_hip_search_fill_usize(self.ctx, i.ctypes.data, j.ctypes.data, d.ctypes.data))
        return i, j, d

    def within_set(self, cutoff, xyz1, idx1=None, xyz2=None, idx2=None, box=None, pbc=0, ids_local=False, lower=None,
                   upper=None, device_out=None):
        """`within` as the set its callers keep (molar_hip_within_count + _fill): sorted, de-duplicated ids of the atoms of
        set 1 with an atom of set 2 within the cutoff - np.unique of the stream search_count(SEARCH_WITHIN) +
        search_fill_ids returns, without the stream.  Returns a uint64 array (or, with `device_out` = a callable that
        allocates a device int64 tensor of a given length, that tensor)."""
        d, keep = self._search_desc(SEARCH_WITHIN, cutoff, xyz1, idx1, xyz2, idx2, box, pbc, None, None, ids_local, lower, upper)
        cnt = C.c_uint64(0)
        check(self.lib.molar_hip_within_count(self.ctx, C.byref(d), C.byref(cnt)))
        self._keep = keep
        n = int(cnt.value)
        if device_out is not None:
            out = device_out(n)
            if n:
                check(self.lib.molar_hip_within_fill(self.ctx, out.data_ptr()))
            return out
        ids = np.empty(n, np.uint64)
        if n:
            check(self.lib.molar_hip_within_fill(self.ctx, ids.ctypes.data))
        return ids

    def search_connectivity(self, cutoff, xyz, idx=None, box=None, pbc=0, ids_local=True):
        """SearchConnectivity::from_iter over the single-selection search (connectivity.rs:19-35), built on the device from
        the resident pair list (molar_hip_search_connectivity / _fill): CSR (offsets uint64[rows + 1], neigh uint64[2 * pairs])
        with every list in the reference's push order; rows = len(selection) for local ids, natoms for global ones."""
        d, keep = self._search_desc(SEARCH_SINGLE, cutoff, xyz, idx, None, None, box, pbc, None, None, ids_local, None, None)
        rows, ent = C.c_uint64(0), C.c_uint64(0)
        check(self.lib.molar_hip_search_connectivity(self.ctx, C.byref(d), C.byref(rows), C.byref(ent)))
        self._keep = keep
        off = np.empty(rows.value + 1, np.uint64)
        nb = np.empty(max(ent.value, 1), np.uint64)
        check(self.lib.molar_hip_search_connectivity_fill(self.ctx, off.ctypes.data, nb.ctypes.data))
        return off, nb[:ent.value]

    def within_hold(self, on=True):
        """molar_hip_within_hold: while on, within_set calls that name the same first set (same array / tensor, same index,
        same box) and come to the same grid reuse its staged coordinates and its grid.  The caller promises not to change
        those coordinates meanwhile."""
        check(self.lib.molar_hip_within_hold(self.ctx, 1 if on else 0))

    def search_fill_ids(self, count):
        ids = np.empty(count, np.uint64)
        check(self.lib.molar_hip_search_fill_ids(self.ctx, ids.ctypes.data))
        return ids

    def search_fill_device(self):
        """Fill ctx-owned device buffers; returns (pairs_ptr, dist_ptr) device addresses."""
        p = C.c_void_p(); d = C.c_void_p()
        check(self.lib.molar_hip_search_fill_device(self.ctx, C.byref(p), C.byref(d)))
        return p.value, d.value

    def search_fill_into(self, pairs_t, dist_t):
        """Fill caller-owned torch CUDA tensors (uint32/int32 [N,2], float32 [N]) in place."""
        pa, _ = _addr(pairs_t); da, _ = _addr(dist_t)
        check(self.lib.molar_hip_search_fill(self.ctx, pa, da))

    # ------------------------------------------------------------ search, f64 build of MolAR (Float = f64)
    def search_f64(self, kind, cutoff, xyz1, idx1=None, xyz2=None, idx2=None, box=None, pbc=0, vdw1=None, vdw2=None,
                   ids_local=False, lower=None, upper=None, device_out=False, out=None):
        """The distance_search drivers with every operation in double (molar_hip_search_count_f64 + fill): returns
        (i, j, d) as uint64 / uint64 / float64 arrays in the reference's order, or the uint64 ids for SEARCH_WITHIN.
        Coordinates, indices and radii may be torch tensors in HBM (used in place); device_out=True leaves the result
        there too (int64 / int64 / float64 tensors); `out` = (i, j, d) device tensors of at least the result's length to fill
        instead of fresh ones (views of the result's length are returned)."""
        def f64(a):
            if a is None:
                return None
            if hasattr(a, "data_ptr"):          # a torch tensor in HBM is used in place (float64, contiguous)
                if str(a.dtype) != "torch.float64" or not a.is_contiguous():
                    raise TypeError("search_f64: device tensors must be contiguous float64")
                return a
            return np.ascontiguousarray(a, np.float64)

        def addr(a):
            return None if a is None else (a.data_ptr() if hasattr(a, "data_ptr") else a.ctypes.data)
        xyz1, xyz2, vdw1, vdw2 = f64(xyz1), f64(xyz2), f64(vdw1), f64(vdw2)
        idx1 = idx1 if hasattr(idx1, "data_ptr") else _u64(idx1)       # (device index tensors: int64)
        idx2 = idx2 if hasattr(idx2, "data_ptr") else _u64(idx2)
        d = SearchDescF64()
        d.kind = kind
        d.cutoff = float(cutoff) if cutoff is not None else 0.0
        keep = [xyz1, xyz2, vdw1, vdw2, idx1, idx2]
        for name, arr in (("xyz1", xyz1), ("idx1", idx1), ("xyz2", xyz2), ("idx2", idx2), ("vdw1", vdw1), ("vdw2", vdw2)):
            setattr(d, name, addr(arr))
        d.natoms1 = 0 if xyz1 is None else xyz1.reshape(-1, 3).shape[0]
        d.natoms2 = 0 if xyz2 is None else xyz2.reshape(-1, 3).shape[0]
        d.n1 = 0 if idx1 is None else idx1.shape[0]
        d.n2 = 0 if idx2 is None else idx2.shape[0]
        d.ids_local = 1 if ids_local else 0
        if box is not None:
            b9 = np.ascontiguousarray(np.asarray(box, np.float64).T).reshape(9)      # column-major: columns a, b, c
            keep.append(b9)
            d.box9 = b9.ctypes.data
            d.pbc = int(pbc)
        if lower is not None:
            lo, up = f64(lower), f64(upper)
            keep += [lo, up]
            d.lower3, d.upper3 = lo.ctypes.data, up.ctypes.data
        n = C.c_uint64()
        check(self.lib.molar_hip_search_count_f64(self.ctx, C.byref(d), C.byref(n)))
        n = int(n.value)
        if device_out:
            import torch
            dev = torch.device("cuda", self.device)
            if kind == SEARCH_WITHIN:
                ids = torch.empty(n, dtype=torch.int64, device=dev)
                if n:
                    check(self.lib.molar_hip_search_fill_ids_f64(self.ctx, ids.data_ptr()))
                return ids
            if out is not None:
                if min(len(out[0]), len(out[1]), len(out[2])) < n:
                    raise ValueError("search_f64: `out` tensors are shorter than the result")
                i, j, dist = out[0][:n], out[1][:n], out[2][:n]
            else:
                i = torch.empty(n, dtype=torch.int64, device=dev); j = torch.empty(n, dtype=torch.int64, device=dev)
                dist = torch.empty(n, dtype=torch.float64, device=dev)
            if n:
                check(self.lib.molar_hip_search_fill_f64(self.ctx, i.data_ptr(), j.data_ptr(), dist.data_ptr()))
            return i, j, dist
        if kind == SEARCH_WITHIN:
            ids = np.empty(n, np.uint64)
            check(self.lib.molar_hip_search_fill_ids_f64(self.ctx, ids.ctypes.data))
            return ids
        i = np.empty(n, np.uint64); j = np.empty(n, np.uint64); dist = np.empty(n, np.float64)
        check(self.lib.molar_hip_search_fill_f64(self.ctx, i.ctypes.data, j.ctypes.data, dist.ctypes.data))
        return i, j, dist

    def grid_dims_f64(self):
        dims = (C.c_uint64 * 3)()
        check(self.lib.molar_hip_search_grid_dims_f64(self.ctx, dims))
        return tuple(int(x) for x in dims)

    def grid_dims(self):
        dims = (C.c_uint64 * 3)()
        check(self.lib.molar_hip_search_grid_dims(self.ctx, dims))
        return tuple(int(x) for x in dims)

    def search_histogram(self, kind, cutoff, hmin, hmax, nbins, xyz1, idx1=None, xyz2=None, idx2=None, box=None,
                         pbc=0, vdw1=None, vdw2=None, bins=None, want_count=True):
        """Consumer-fused search: every emitted distance goes through Histogram1D::add_one
        (molar_membrane/src/stats.rs:29-35); pairs are never materialised.  `bins` (uint64[nbins], numpy or a torch
        int64 CUDA tensor) is accumulated into, so frames can be summed.  Returns (bins, number_of_pairs).  With
        device-resident bins and want_count=False the call does not wait for the GPU (count is returned as None):
        the frames of a trajectory queue up back to back; synchronize() before reading the bins."""
        xyz1 = _f32(xyz1); xyz2 = _f32(xyz2); idx1 = _u64(idx1); idx2 = _u64(idx2)
        vdw1 = _f32(vdw1); vdw2 = _f32(vdw2)
        d = SearchDesc()
        d.kind = kind
        d.cutoff = float(cutoff) if cutoff is not None else 0.0
        keep = []
        for name, arr in (("xyz1", xyz1), ("idx1", idx1), ("xyz2", xyz2), ("idx2", idx2), ("vdw1", vdw1),
                          ("vdw2", vdw2)):
            a, k = _addr(arr)
            setattr(d, name, a)
            keep.append(k)
        d.natoms1 = 0 if xyz1 is None else xyz1.shape[0]
        d.natoms2 = 0 if xyz2 is None else xyz2.shape[0]
        d.n1 = 0 if idx1 is None else idx1.shape[0]
        d.n2 = 0 if idx2 is None else idx2.shape[0]
        if box is not None:
            ba, kb = self._box9(box)
            keep.append(kb)
            d.box9 = ba
        d.pbc = pbc_mask(pbc)
        if bins is None:
            bins = np.zeros(nbins, np.uint64)
        cnt = C.c_uint64(0)
        ba_, kb_ = _addr(bins)
        check(self.lib.molar_hip_search_histogram(self.ctx, C.byref(d), float(hmin), float(hmax), nbins,
                                                  ba_, C.byref(cnt) if want_count else None))
        self._keep = keep
        return bins, (int(cnt.value) if want_count else None)

    def search_histogram_frames(self, kind, cutoff, hmin, hmax, nbins, frames, idx1=None, box=None, pbc=0, bins=None, frames2=None, idx2=None):
        """molar_hip_search_histogram_frames: the frames of a trajectory block, `frames` = [nframes, natoms, 3] float32 (one
        contiguous array or CUDA tensor), through the fused histogram; the same sums as nframes calls of search_histogram.
        `box`: one 3x3 matrix for all frames or [nframes, 3, 3].  SEARCH_DOUBLE: the second set is `idx2` of `frames2` (None:
        of the same frames - two selections of one trajectory).  With frames, indices and bins in device memory the frames go
        through the GPU in groups that share their launches, and the call does not wait (synchronize() before reading)."""
        def block(fr):
            stride = None
            if _is_torch(fr) and fr.ndim == 3 and fr.stride(2) == 1 and fr.stride(1) == 3:
                import torch
                assert fr.dtype == torch.float32
                stride = int(fr.stride(0))          # frames with a gap between them (a window of a larger buffer): no copy
                fa, k = fr.data_ptr(), fr
            else:
                fr = _f32(fr)
                fa, k = _addr(fr)
            assert fr.ndim == 3 and fr.shape[2] == 3
            return fr, fa, k, (int(fr.shape[1]) * 3 if stride is None else stride)
        frames, fa, k1, stride = block(frames)
        idx1 = _u64(idx1)
        nframes, natoms = int(frames.shape[0]), int(frames.shape[1])
        d = SearchDesc()
        d.kind = kind
        d.cutoff = float(cutoff)
        ia, k2 = _addr(idx1)
        d.xyz1 = fa
        d.idx1 = ia
        d.natoms1 = natoms
        d.n1 = 0 if idx1 is None else idx1.shape[0]
        keep = [k1, k2]
        stride2 = 0
        if kind in (SEARCH_DOUBLE, SEARCH_DOUBLE_VDW):
            f2, fa2, k3, stride2 = block(frames if frames2 is None else frames2)
            assert int(f2.shape[0]) == nframes
            idx2 = _u64(idx2)
            ia2, k4 = _addr(idx2)
            d.xyz2 = fa2
            d.idx2 = ia2
            d.natoms2 = int(f2.shape[1])
            d.n2 = 0 if idx2 is None else idx2.shape[0]
            keep += [k3, k4]
        boxes_ptr = None
        if box is not None:
            b = np.asarray(box.get_matrix() if isinstance(box, PeriodicBox) else box, np.float32)
            if b.ndim == 3:
                b9 = np.ascontiguousarray(np.transpose(b, (0, 2, 1))).reshape(nframes, 9)      # column-major per frame
                boxes_ptr = b9.ctypes.data
                d.box9 = boxes_ptr
            else:
                b9 = np.ascontiguousarray(b.reshape(3, 3).T).reshape(9)
                d.box9 = b9.ctypes.data
            keep.append(b9)
        d.pbc = pbc_mask(pbc)
        if bins is None:
            bins = np.zeros(nbins, np.uint64)
        ba_, kb_ = _addr(bins)
        check(self.lib.molar_hip_search_histogram_frames(self.ctx, C.byref(d), nframes, stride, stride2, boxes_ptr, float(hmin), float(hmax),
                                                         nbins, ba_))
        self._keep = keep
        return bins

    # ------------------------------------------------------------ measure
    def _sel_args(self, xyz, idx):
        xyz = _f32(xyz); idx = _u64(idx)
        xa, k1 = _addr(xyz); ia, k2 = _addr(idx)
        natoms = xyz.shape[0] if xyz.ndim == 2 else xyz.shape[0] // 3
        n = 0 if idx is None else idx.shape[0]
        return xa, natoms, ia, n, (k1, k2)

    @staticmethod
    def _box9(box):
        if box is None:
            return None, None
        b9 = box.colmajor9() if isinstance(box, PeriodicBox) else np.ascontiguousarray(
            np.asarray(box, np.float32).reshape(3, 3).T).reshape(9)
        return b9.ctypes.data, b9

    def min_max(self, xyz, idx=None):
        xa, na, ia, n, k = self._sel_args(xyz, idx)
        lo = np.zeros(3, np.float32); up = np.zeros(3, np.float32)
        check(self.lib.molar_hip_min_max(self.ctx, xa, na, ia, n, lo.ctypes.data, up.ctypes.data))
        return lo, up

    def center_of_geometry(self, xyz, idx=None):
        xa, na, ia, n, k = self._sel_args(xyz, idx)
        out = np.zeros(3, np.float32)
        check(self.lib.molar_hip_center_of_geometry(self.ctx, xa, na, ia, n, out.ctypes.data))
        return out

    def center_of_mass(self, xyz, mass, idx=None):
        xa, na, ia, n, k = self._sel_args(xyz, idx)
        mass = _f32(mass); ma, km = _addr(mass)
        out = np.zeros(3, np.float32)
        check(self.lib.molar_hip_center_of_mass(self.ctx, xa, na, ia, n, ma, out.ctypes.data))
        return out

    def center_of_geometry_pbc(self, xyz, box, dims=PBC_FULL, idx=None):
        xa, na, ia, n, k = self._sel_args(xyz, idx)
        ba, kb = self._box9(box)
        out = np.zeros(3, np.float32)
        check(self.lib.molar_hip_center_of_geometry_pbc(self.ctx, xa, na, ia, n, ba, pbc_mask(dims), out.ctypes.data))
        return out

    def center_of_mass_pbc(self, xyz, mass, box, dims=PBC_FULL, idx=None):
        xa, na, ia, n, k = self._sel_args(xyz, idx)
        mass = _f32(mass); ma, km = _addr(mass)
        ba, kb = self._box9(box)
        out = np.zeros(3, np.float32)
        check(self.lib.molar_hip_center_of_mass_pbc(self.ctx, xa, na, ia, n, ma, ba, pbc_mask(dims), out.ctypes.data))
        return out

    def gyration(self, xyz, mass, idx=None, box=None):
        xa, na, ia, n, k = self._sel_args(xyz, idx)
        mass = _f32(mass); ma, km = _addr(mass)
        ba, kb = self._box9(box)
        out = C.c_float(0)
        check(self.lib.molar_hip_gyration(self.ctx, xa, na, ia, n, ma, ba, C.byref(out)))
        return float(out.value)

    def inertia(self, xyz, mass, idx=None, box=None):
        """(moments[3] ascending, axes 3x3 with axes as columns, raw tensor 3x3)."""
        xa, na, ia, n, k = self._sel_args(xyz, idx)
        mass = _f32(mass); ma, km = _addr(mass)
        ba, kb = self._box9(box)
        mom = np.zeros(3, np.float32); axes = np.zeros(9, np.float32); tens = np.zeros(9, np.float32)
        check(self.lib.molar_hip_inertia(self.ctx, xa, na, ia, n, ma, ba, mom.ctypes.data, axes.ctypes.data,
                                         tens.ctypes.data))
        return mom, axes.reshape(3, 3).T.copy(), tens.reshape(3, 3).T.copy()

    def rmsd(self, xyz1, xyz2, idx1=None, idx2=None):
        a1 = self._sel_args(xyz1, idx1); a2 = self._sel_args(xyz2, idx2)
        out = C.c_float(0)
        check(self.lib.molar_hip_rmsd(self.ctx, *a1[:4], *a2[:4], C.byref(out)))
        return float(out.value)

    def rmsd_mw(self, xyz1, mass1, xyz2, idx1=None, idx2=None):
        a1 = self._sel_args(xyz1, idx1); a2 = self._sel_args(xyz2, idx2)
        mass1 = _f32(mass1); ma, km = _addr(mass1)
        out = C.c_float(0)
        check(self.lib.molar_hip_rmsd_mw(self.ctx, *a1[:4], ma, *a2[:4], C.byref(out)))
        return float(out.value)

    def fit_transform(self, xyz1, mass1, xyz2, mass2, idx1=None, idx2=None, at_origin=False):
        """(R, t) with p -> R @ p + t  (IsometryMatrix3, measure.rs:507-535)."""
        a1 = self._sel_args(xyz1, idx1); a2 = self._sel_args(xyz2, idx2)
        mass1 = _f32(mass1); m1, k1 = _addr(mass1)
        mass2 = _f32(mass2); m2, k2 = _addr(mass2)
        R = np.zeros(9, np.float32); t = np.zeros(3, np.float32)
        check(self.lib.molar_hip_fit_transform(self.ctx, *a1[:4], m1, *a2[:4], m2, 1 if at_origin else 0,
                                               R.ctypes.data, t.ctypes.data))
        return R.reshape(3, 3).T.copy(), t

    def apply_transform(self, xyz, R, t, idx=None):
        """In place on xyz (numpy float32 C-contiguous array or torch CUDA tensor)."""
        if not _is_torch(xyz):
            assert xyz.dtype == np.float32 and xyz.flags.c_contiguous, "apply_transform works in place"
        xa, na, ia, n, k = self._sel_args(xyz, idx)
        Rf = np.ascontiguousarray(np.asarray(R, np.float32).T).reshape(9)
        tf = np.ascontiguousarray(t, np.float32)
        check(self.lib.molar_hip_apply_transform(self.ctx, xa, na, ia, n, Rf.ctypes.data, tf.ctypes.data))
        return xyz

    def unwrap_simple(self, xyz, box, dims=PBC_FULL, idx=None):
        if not _is_torch(xyz):
            assert xyz.dtype == np.float32 and xyz.flags.c_contiguous, "unwrap_simple works in place"
        xa, na, ia, n, k = self._sel_args(xyz, idx)
        ba, kb = self._box9(box)
        check(self.lib.molar_hip_unwrap_simple(self.ctx, xa, na, ia, n, ba, pbc_mask(dims)))
        return xyz

    def unwrap_connectivity(self, xyz, box, cutoff, dims=PBC_FULL, idx=None):
        """Modify::unwrap_connectivity_dim (modify.rs:72-131) in place: GPU neighbour search with local ids, adjacency in
        pair order and the reference's stack walk inside the library (molar_hip_unwrap_connectivity).  Returns the list of
        groups of LOCAL indices the reference returns as selections."""
        if not _is_torch(xyz):
            assert xyz.dtype == np.float32 and xyz.flags.c_contiguous, "unwrap_connectivity works in place"
        if idx is not None and len(idx) == 0:      # (a NULL index means "all atoms" to the C ABI: never pass an empty one as NULL)
            raise ValueError("unwrap_connectivity: empty selection")
        xa, na, ia, n, k = self._sel_args(xyz, idx)
        nsel = n if idx is not None else na
        ba, kb = self._box9(box)
        goff = np.zeros(nsel + 1, np.uint64); gids = np.zeros(max(nsel, 1), np.uint64)
        ng = C.c_size_t(0)
        check(self.lib.molar_hip_unwrap_connectivity(self.ctx, xa, na, ia, n, ba, float(cutoff), pbc_mask(dims), goff.ctypes.data,
                                                     gids.ctypes.data, C.byref(ng)))
        return [gids[int(goff[g]):int(goff[g + 1])].copy() for g in range(int(ng.value))]

    def center_batch(self, xyz, idx, offsets, mass=None):
        """Centres of K selections given as CSR (idx, offsets[K+1]): center_of_mass if `mass` is given,
        else center_of_geometry.  One wave per selection; returns float32 [K,3]."""
        xyz = _f32(xyz); idx = _u64(idx); offsets = _u64(offsets); mass = _f32(mass)
        xa, k1 = _addr(xyz); ia, k2 = _addr(idx); oa, k3 = _addr(offsets); ma, k4 = _addr(mass)
        K = offsets.shape[0] - 1
        out = np.zeros((K, 3), np.float32)
        check(self.lib.molar_hip_center_batch(self.ctx, xa, xyz.shape[0], ia, oa, K, ma, out.ctypes.data))
        return out

    def unwrap_simple_batch(self, xyz, idx, offsets, box, dims=PBC_FULL):
        """unwrap_simple on each of K CSR selections, in place."""
        if not _is_torch(xyz):
            assert xyz.dtype == np.float32 and xyz.flags.c_contiguous
        idx = _u64(idx); offsets = _u64(offsets)
        xa, k1 = _addr(xyz); ia, k2 = _addr(idx); oa, k3 = _addr(offsets)
        ba, kb = self._box9(box)
        check(self.lib.molar_hip_unwrap_simple_batch(self.ctx, xa, xyz.shape[0], ia, oa, offsets.shape[0] - 1, ba,
                                                     pbc_mask(dims)))
        return xyz

    def lipid_tail_order_csr(self, xyz, idx, tail_offsets, order_type, normals, normal_offsets, bond_orders):
        """Same as lipid_tail_order with the CSR arrays prepared by the caller (no per-call Python loops)."""
        xyz = _f32(xyz)
        xa, kx = _addr(xyz)
        idx = _u64(idx); tail_offsets = np.ascontiguousarray(tail_offsets, np.uint64)
        normal_offsets = np.ascontiguousarray(normal_offsets, np.uint64)
        normals = np.ascontiguousarray(normals, np.float32)
        if not _is_torch(bond_orders):                      # idx and bond_orders may live on the GPU (torch int64 / uint8)
            bond_orders = np.ascontiguousarray(bond_orders, np.uint8)
        ia, ki = _addr(idx); ba, kb = _addr(bond_orders)
        K = len(tail_offsets) - 1
        nout = int(tail_offsets[-1]) - 2 * K
        out = np.zeros(max(nout, 1), np.float32)
        check(self.lib.molar_hip_lipid_tail_order(self.ctx, xa, xyz.shape[0], ia, tail_offsets.ctypes.data, K,
                                                  int(order_type), normals.ctypes.data, normal_offsets.ctypes.data,
                                                  ba, out.ctypes.data))
        return out[:nout]

    def copy_bandwidth(self, nbytes=1 << 30, reps=10) -> float:
        """Measured device-to-device copy rate in GB/s (read + write), best of the library's float4 copy kernels."""
        out = C.c_float(0)
        check(self.lib.molar_hip_copy_bandwidth(self.ctx, int(nbytes), int(reps), C.byref(out)))
        return float(out.value)

    def write_bandwidth(self, nbytes=1 << 30, reps=10) -> float:
        """Measured write-only float4 stream rate in GB/s."""
        out = C.c_float(0)
        check(self.lib.molar_hip_write_bandwidth(self.ctx, int(nbytes), int(reps), C.byref(out)))
        return float(out.value)

    def membrane_smooth(self, box, state, patch_offsets, patch_ids):
        """One iteration of Membrane::smooth (molar_membrane/src/lib.rs:661-812) on the GPU.  `state` is a dict of
        per-lipid arrays updated IN PLACE (see new_membrane_state); a lipid that turns invalid keeps its old
        values, like the fields of the reference's LipidMolecule."""
        pb = box if isinstance(box, PeriodicBox) else PeriodicBox.from_matrix(box)
        po = _u64(patch_offsets); pi = _u64(patch_ids)
        K = len(po) - 1
        E = int(po[-1]); slots = E + 4 * K
        st = state
        if st["neib_ids"].shape[0] != max(slots, 1):                       # patch structure changed: re-slot
            st["neib_ids"] = np.zeros(max(slots, 1), np.uint64)
            st["voro_vertexes"] = np.zeros((max(slots, 1), 3), np.float32)
            st["fitted_patch_points"] = np.zeros((max(E, 1), 3), np.float32)
            st["nvert"][:] = 0
        P = _MembranePatches(K, po.ctypes.data, pi.ctypes.data if E else None)
        S = _MembraneState(*[st[k].ctypes.data for k in _MEMBRANE_FIELDS])
        m9 = pb.colmajor9()
        check(self.lib.molar_hip_membrane_smooth(self.ctx, C.byref(P), m9.ctypes.data, C.byref(S)))
        return st

    def lipid_tail_order(self, xyz, tails, order_type, normals, bond_orders):
        """Batched Measure::lipid_tail_order (measure.rs:270-422).  tails: list of index arrays (the
        tail carbons, in chain order); normals: list of [1,3] or [n-2,3] arrays; bond_orders: list of
        n-1 arrays of 1/2.  order_type: 0 Sz, 1 Scd, 2 ScdCorr.  Returns a list of n-2 arrays."""
        xyz = _f32(xyz)
        xa, kx = _addr(xyz)
        natoms = xyz.shape[0]
        lens = np.array([len(t) for t in tails], dtype=np.uint64)
        toff = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        idx = np.ascontiguousarray(np.concatenate([np.asarray(t, np.uint64) for t in tails]))
        nl = [np.asarray(n, np.float32).reshape(-1, 3) for n in normals]
        noff = np.concatenate([[0], np.cumsum([len(n) for n in nl])]).astype(np.uint64)
        nrm = np.ascontiguousarray(np.concatenate(nl))
        bo = np.ascontiguousarray(np.concatenate([np.asarray(b, np.uint8) for b in bond_orders])) if bond_orders is not None else None
        if bo is not None and len(bo) != int(toff[-1]) - len(tails):
            raise MolarHipError(9, "for N tail carbons # of bond orders should be N-1")       # LipidOrderError::BondOrderCount
        nout = max(int(toff[-1]) - 2 * len(tails), 0)
        out = np.zeros(max(nout, 1), np.float32)
        check(self.lib.molar_hip_lipid_tail_order(self.ctx, xa, natoms, idx.ctypes.data, toff.ctypes.data, len(tails),
                                                  int(order_type), nrm.ctypes.data, noff.ctypes.data,
                                                  None if bo is None else bo.ctypes.data, out.ctypes.data))
        res, pos = [], 0
        for n in lens:
            res.append(out[pos:pos + int(n) - 2].copy())
            pos += int(n) - 2
        return res

    # ------------------------------------------------------------ CSR-batched Measure (ParSplit-style loops)
    def gyration_batch(self, xyz, idx, offsets, mass, box=None):
        """Measure::gyration (measure.rs:78-87; gyration_pbc :222-232 with a box) of each of the K selections
        idx[offsets[k]:offsets[k+1]] - what MolAR runs from rayon over a ParSplit (system.rs:193-213)."""
        xyz = _f32(xyz); idx = _u64(idx); offsets = _u64(offsets); mass = _f32(mass)
        xa, k1 = _addr(xyz); ia, k2 = _addr(idx); oa, k3 = _addr(offsets); ma, k4 = _addr(mass)
        ba, kb = self._box9(box)
        K = len(offsets) - 1
        out = np.zeros(max(K, 1), np.float32)
        check(self.lib.molar_hip_gyration_batch(self.ctx, xa, xyz.shape[0], ia, oa, K, ma, ba, out.ctypes.data))
        return out[:K]

    def rmsd_batch(self, xyz1, xyz2, idx, offsets, mass=None, idx2=None):
        """rmsd (measure.rs:485-504; rmsd_mw :538-558 with `mass`) of selection k in frame 1 against selection k in frame 2."""
        xyz1 = _f32(xyz1); xyz2 = _f32(xyz2); idx = _u64(idx); idx2 = _u64(idx2); offsets = _u64(offsets); mass = _f32(mass)
        a1, k1 = _addr(xyz1); a2, k2 = _addr(xyz2); ia, k3 = _addr(idx); ja, k4 = _addr(idx2); oa, k5 = _addr(offsets)
        ma, k6 = _addr(mass)
        K = len(offsets) - 1
        out = np.zeros(max(K, 1), np.float32)
        check(self.lib.molar_hip_rmsd_batch(self.ctx, a1, xyz1.shape[0], ia, a2, xyz2.shape[0], ja, oa, K, ma, out.ctypes.data))
        return out[:K]

    def fit_batch(self, xyz1, mass1, xyz2, idx, offsets, idx2=None, mass2=None, apply=False):
        """fit_transform (measure.rs:507-522) of each selection of frame 1 onto its counterpart in frame 2; `apply`
        moves the selections of xyz1 in place.  Returns dict(R[K,3,3], t[K,3], rmsd[K], com[K,3], gyration[K]); the last
        three describe the fitted selections."""
        if apply and not _is_torch(xyz1):
            assert xyz1.dtype == np.float32 and xyz1.flags.c_contiguous, "apply works in place"
        xyz1 = _f32(xyz1); xyz2 = _f32(xyz2); idx = _u64(idx); idx2 = _u64(idx2); offsets = _u64(offsets)
        mass1 = _f32(mass1); mass2 = _f32(mass2)
        a1, k1 = _addr(xyz1); a2, k2 = _addr(xyz2); ia, k3 = _addr(idx); ja, k4 = _addr(idx2); oa, k5 = _addr(offsets)
        m1, k6 = _addr(mass1); m2, k7 = _addr(mass2)
        K = len(offsets) - 1
        R = np.zeros((max(K, 1), 9), np.float32); t = np.zeros((max(K, 1), 3), np.float32)
        rm = np.zeros(max(K, 1), np.float32); com = np.zeros((max(K, 1), 3), np.float32); gy = np.zeros(max(K, 1), np.float32)
        check(self.lib.molar_hip_fit_batch(self.ctx, a1, xyz1.shape[0], ia, m1, a2, xyz2.shape[0], ja, m2, oa, K,
                                           1 if apply else 0, R.ctypes.data, t.ctypes.data, rm.ctypes.data,
                                           com.ctypes.data, gy.ctypes.data))
        return dict(R=R[:K].reshape(K, 3, 3).transpose(0, 2, 1).copy(), t=t[:K], rmsd=rm[:K], com=com[:K], gyration=gy[:K])

    def translate(self, xyz, shift, idx=None):
        """Modify::translate (modify.rs:16-23), in place."""
        if not _is_torch(xyz):
            assert xyz.dtype == np.float32 and xyz.flags.c_contiguous, "translate works in place"
        xa, na, ia, n, k = self._sel_args(xyz, idx)
        sh = np.ascontiguousarray(shift, np.float32).reshape(3)
        check(self.lib.molar_hip_translate(self.ctx, xa, na, ia, n, sh.ctypes.data))
        return xyz

    def rotate(self, xyz, unit_axis, angle, idx=None):
        """Modify::rotate (modify.rs:25-30): Rotation3::from_axis_angle about the origin, in place."""
        if not _is_torch(xyz):
            assert xyz.dtype == np.float32 and xyz.flags.c_contiguous, "rotate works in place"
        xa, na, ia, n, k = self._sel_args(xyz, idx)
        ax = np.ascontiguousarray(unit_axis, np.float32).reshape(3)
        check(self.lib.molar_hip_rotate(self.ctx, xa, na, ia, n, ax.ctypes.data, float(angle)))
        return xyz

    def principal_transform(self, xyz, mass, idx=None, box=None):
        """Measure::principal_transform (measure.rs:102-109; _pbc :246-257 with a box) as (R[3,3], t[3]), p -> R p + t."""
        xa, na, ia, n, k = self._sel_args(xyz, idx)
        mass = _f32(mass); ma, km = _addr(mass)
        ba, kb = self._box9(box)
        R = np.zeros(9, np.float32); t = np.zeros(3, np.float32)
        check(self.lib.molar_hip_principal_transform(self.ctx, xa, na, ia, n, ma, ba, R.ctypes.data, t.ctypes.data))
        return R.reshape(3, 3).T.copy(), t

    def fit_rmsd_batch(self, frames, mass, ref_xyz, idx=None, ref_idx=None, apply=True):
        """frames: [F, natoms, 3] (numpy, modified in place if apply; or torch CUDA tensor).
        Returns dict(rmsd[F], R[F,3,3], t[F,3], com[F,3], gyration[F])."""
        if _is_torch(frames):
            F, natoms = frames.shape[0], frames.shape[1]
            fa, kf = _addr(frames.contiguous())
        else:
            assert frames.dtype == np.float32 and frames.flags.c_contiguous
            F, natoms = frames.shape[0], frames.shape[1]
            fa, kf = _addr(frames)
        idx = _u64(idx); ref_idx = _u64(ref_idx) if ref_idx is not None else idx
        ia, ki = _addr(idx); ra, kr = _addr(ref_idx)
        n = 0 if idx is None else idx.shape[0]
        mass = _f32(mass); ma, km = _addr(mass)
        ref_xyz = _f32(ref_xyz); xa, kx = _addr(ref_xyz)
        ref_natoms = ref_xyz.shape[0] if ref_xyz.ndim == 2 else ref_xyz.shape[0] // 3
        rm = np.zeros(F, np.float32); R = np.zeros((F, 9), np.float32); t = np.zeros((F, 3), np.float32)
        com = np.zeros((F, 3), np.float32); gy = np.zeros(F, np.float32)
        check(self.lib.molar_hip_fit_rmsd_batch(self.ctx, fa, F, natoms, ia, n, ma, xa, ref_natoms, ra,
                                                1 if apply else 0, rm.ctypes.data, R.ctypes.data, t.ctypes.data,
                                                com.ctypes.data, gy.ctypes.data))
        return dict(rmsd=rm, R=R.reshape(F, 3, 3).transpose(0, 2, 1).copy(), t=t, com=com, gyration=gy)


class FitStream:
    """molar_hip_fit_stream_*: the per-frame fit loop of benches/comparison_small.rs:14-25 for frames in HOST memory (numpy),
    three in flight - `t = fs.begin(frame_k1); out = fs.end(t_k)`.  The selected atoms are packed by host threads and only
    they cross the link; every record equals fit_rmsd_batch's for the same frame.  apply=True moves the selection inside the
    numpy frame handed to begin() (keep it alive and untouched until end())."""

    def __init__(self, engine: "Engine", natoms, mass, ref_xyz, idx=None, ref_idx=None, host_threads=0):
        self.eng, self.lib = engine, engine.lib
        idx = None if idx is None else np.ascontiguousarray(idx, np.uint64)
        ref_idx = idx if ref_idx is None else np.ascontiguousarray(ref_idx, np.uint64)
        mass = np.ascontiguousarray(mass, np.float32)
        ref_xyz = np.ascontiguousarray(ref_xyz, np.float32)
        self.natoms = int(natoms)
        h = C.c_void_p()
        check(self.lib.molar_hip_fit_stream_create(engine.ctx, self.natoms, None if idx is None else idx.ctypes.data,
                                                   0 if idx is None else len(idx), mass.ctypes.data, ref_xyz.ctypes.data,
                                                   ref_xyz.shape[0] if ref_xyz.ndim == 2 else ref_xyz.shape[0] // 3,
                                                   None if ref_idx is None else ref_idx.ctypes.data, int(host_threads), C.byref(h)))
        self.h = h
        self._frames = {}

    def begin(self, frame, apply=False) -> int:
        assert isinstance(frame, np.ndarray) and frame.dtype == np.float32 and frame.flags.c_contiguous and frame.size == self.natoms * 3
        t = C.c_int32(-1)
        check(self.lib.molar_hip_fit_stream_begin(self.h, frame.ctypes.data, 1 if apply else 0, C.byref(t)))
        self._frames[t.value] = frame
        return t.value

    def end(self, ticket):
        rm = C.c_float(); gy = C.c_float()
        R = np.zeros(9, np.float32); t = np.zeros(3, np.float32); com = np.zeros(3, np.float32)
        check(self.lib.molar_hip_fit_stream_end(self.h, ticket, C.byref(rm), R.ctypes.data, t.ctypes.data, com.ctypes.data, C.byref(gy)))
        self._frames.pop(ticket, None)
        return dict(rmsd=np.float32(rm.value), R=R.reshape(3, 3).T.copy(), t=t, com=com, gyration=np.float32(gy.value))

    def close(self):
        if self.h:
            self.lib.molar_hip_fit_stream_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _f64(x):
    if x is None:
        return None
    if _is_torch(x):
        import torch
        assert x.dtype == torch.float64
        return x.contiguous()
    return np.ascontiguousarray(x, dtype=np.float64)


class MeasureF64:
    """The Measure / Modify methods for MolAR's `f64` feature (Float = f64, molar/src/aliases.rs:10-13) on
    an Engine's context: float64 coordinates and masses (numpy or torch CUDA), float64 results.  Same argument meaning
    as the Engine methods of the same name; the search exists in f32 only."""

    def __init__(self, engine: "Engine"):
        self.eng, self.lib, self.ctx = engine, engine.lib, engine.ctx

    @staticmethod
    def _sel(xyz, idx):
        xyz = _f64(xyz); idx = _u64(idx)
        xa, k1 = _addr(xyz); ia, k2 = _addr(idx)
        natoms = xyz.shape[0] if xyz.ndim == 2 else xyz.shape[0] // 3
        return xa, natoms, ia, (0 if idx is None else idx.shape[0]), (k1, k2)

    def center_of_geometry(self, xyz, idx=None):
        a = self._sel(xyz, idx)
        out = np.zeros(3, np.float64)
        check(self.lib.molar_hip_center_of_geometry_f64(self.ctx, *a[:4], out.ctypes.data))
        return out

    def center_of_mass(self, xyz, mass, idx=None):
        a = self._sel(xyz, idx)
        mass = _f64(mass); ma, km = _addr(mass)
        out = np.zeros(3, np.float64)
        check(self.lib.molar_hip_center_of_mass_f64(self.ctx, *a[:4], ma, out.ctypes.data))
        return out

    def gyration(self, xyz, mass, idx=None):
        a = self._sel(xyz, idx)
        mass = _f64(mass); ma, km = _addr(mass)
        out = C.c_double(0)
        check(self.lib.molar_hip_gyration_f64(self.ctx, *a[:4], ma, C.byref(out)))
        return float(out.value)

    def rmsd(self, xyz1, xyz2, idx1=None, idx2=None):
        a1 = self._sel(xyz1, idx1); a2 = self._sel(xyz2, idx2)
        out = C.c_double(0)
        check(self.lib.molar_hip_rmsd_f64(self.ctx, *a1[:4], *a2[:4], C.byref(out)))
        return float(out.value)

    def rmsd_mw(self, xyz1, mass1, xyz2, idx1=None, idx2=None):
        a1 = self._sel(xyz1, idx1); a2 = self._sel(xyz2, idx2)
        mass1 = _f64(mass1); ma, km = _addr(mass1)
        out = C.c_double(0)
        check(self.lib.molar_hip_rmsd_mw_f64(self.ctx, *a1[:4], ma, *a2[:4], C.byref(out)))
        return float(out.value)

    def fit_transform(self, xyz1, mass1, xyz2, mass2, idx1=None, idx2=None, at_origin=False):
        """(R, t) with p -> R @ p + t."""
        a1 = self._sel(xyz1, idx1); a2 = self._sel(xyz2, idx2)
        mass1 = _f64(mass1); m1, k1 = _addr(mass1)
        mass2 = _f64(mass2); m2, k2 = _addr(mass2)
        R = np.zeros(9, np.float64); t = np.zeros(3, np.float64)
        check(self.lib.molar_hip_fit_transform_f64(self.ctx, *a1[:4], m1, *a2[:4], m2, 1 if at_origin else 0,
                                                   R.ctypes.data, t.ctypes.data))
        return R.reshape(3, 3).T.copy(), t

    @staticmethod
    def _box9(box):
        """box: 3x3 with COLUMNS = box vectors (PeriodicBox::from_matrix), float64."""
        b9 = np.ascontiguousarray(np.asarray(box, np.float64).reshape(3, 3).T).reshape(9)
        return b9.ctypes.data, b9

    def center_of_geometry_pbc(self, xyz, box, dims=PBC_FULL, idx=None):
        a = self._sel(xyz, idx); ba, kb = self._box9(box)
        out = np.zeros(3, np.float64)
        check(self.lib.molar_hip_center_of_geometry_pbc_f64(self.ctx, *a[:4], ba, pbc_mask(dims), out.ctypes.data))
        return out

    def center_of_mass_pbc(self, xyz, mass, box, dims=PBC_FULL, idx=None):
        a = self._sel(xyz, idx); ba, kb = self._box9(box)
        mass = _f64(mass); ma, km = _addr(mass)
        out = np.zeros(3, np.float64)
        check(self.lib.molar_hip_center_of_mass_pbc_f64(self.ctx, *a[:4], ma, ba, pbc_mask(dims), out.ctypes.data))
        return out

    def gyration_pbc(self, xyz, mass, box, idx=None):
        a = self._sel(xyz, idx); ba, kb = self._box9(box)
        mass = _f64(mass); ma, km = _addr(mass)
        out = C.c_double(0)
        check(self.lib.molar_hip_gyration_pbc_f64(self.ctx, *a[:4], ma, ba, C.byref(out)))
        return float(out.value)

    def unwrap_simple(self, xyz, box, dims=PBC_FULL, idx=None):
        if not _is_torch(xyz):
            assert xyz.dtype == np.float64 and xyz.flags.c_contiguous, "unwrap_simple works in place"
        a = self._sel(xyz, idx); ba, kb = self._box9(box)
        check(self.lib.molar_hip_unwrap_simple_f64(self.ctx, *a[:4], ba, pbc_mask(dims)))

    def fit_rmsd_batch(self, frames, mass, ref_xyz, idx=None, ref_idx=None, apply=True):
        """frames: float64 [F, natoms, 3] (numpy, modified in place if apply; or torch CUDA tensor).
        Returns dict(rmsd[F], R[F,3,3], t[F,3], com[F,3], gyration[F]) in float64."""
        if not _is_torch(frames):
            assert frames.dtype == np.float64 and frames.flags.c_contiguous
        else:
            import torch
            assert frames.dtype == torch.float64 and frames.is_contiguous()
        F, natoms = frames.shape[0], frames.shape[1]
        fa, kf = _addr(frames)
        idx = _u64(idx); ref_idx = _u64(ref_idx) if ref_idx is not None else idx
        ia, ki = _addr(idx); ra, kr = _addr(ref_idx)
        n = 0 if idx is None else idx.shape[0]
        mass = _f64(mass); ma, km = _addr(mass)
        ref_xyz = _f64(ref_xyz); xa, kx = _addr(ref_xyz)
        ref_natoms = ref_xyz.shape[0] if ref_xyz.ndim == 2 else ref_xyz.shape[0] // 3
        rm = np.zeros(F, np.float64); R = np.zeros((F, 9), np.float64); t = np.zeros((F, 3), np.float64)
        com = np.zeros((F, 3), np.float64); gy = np.zeros(F, np.float64)
        check(self.lib.molar_hip_fit_rmsd_batch_f64(self.ctx, fa, F, natoms, ia, n, ma, xa, ref_natoms, ra,
                                                    1 if apply else 0, rm.ctypes.data, R.ctypes.data, t.ctypes.data,
                                                    com.ctypes.data, gy.ctypes.data))
        return dict(rmsd=rm, R=R.reshape(F, 3, 3).transpose(0, 2, 1).copy(), t=t, com=com, gyration=gy)

    def min_max(self, xyz, idx=None):
        a = self._sel(xyz, idx)
        lo = np.zeros(3, np.float64); hi = np.zeros(3, np.float64)
        check(self.lib.molar_hip_min_max_f64(self.ctx, *a[:4], lo.ctypes.data, hi.ctypes.data))
        return lo, hi

    def inertia(self, xyz, mass, idx=None, box=None):
        """(moments[3] ascending, axes 3x3 with axes as columns, raw tensor 3x3); inertia_pbc with a box."""
        a = self._sel(xyz, idx)
        mass = _f64(mass); ma, km = _addr(mass)
        mom = np.zeros(3, np.float64); axes = np.zeros(9, np.float64); tens = np.zeros(9, np.float64)
        if box is None:
            check(self.lib.molar_hip_inertia_f64(self.ctx, *a[:4], ma, mom.ctypes.data, axes.ctypes.data, tens.ctypes.data))
        else:
            ba, kb = self._box9(box)
            check(self.lib.molar_hip_inertia_pbc_f64(self.ctx, *a[:4], ma, ba, mom.ctypes.data, axes.ctypes.data,
                                                     tens.ctypes.data))
        return mom, axes.reshape(3, 3).T.copy(), tens.reshape(3, 3).T.copy()

    def translate(self, xyz, shift, idx=None):
        if not _is_torch(xyz):
            assert xyz.dtype == np.float64 and xyz.flags.c_contiguous, "translate works in place"
        a = self._sel(xyz, idx)
        sh = np.ascontiguousarray(shift, np.float64)
        check(self.lib.molar_hip_translate_f64(self.ctx, *a[:4], sh.ctypes.data))

    def lipid_tail_order(self, xyz, tails, order_type, normals, bond_orders):
        """Batched Measure::lipid_tail_order (measure.rs:270-422) in f64; arguments as Engine.lipid_tail_order."""
        xyz = _f64(xyz)
        xa, kx = _addr(xyz)
        lens = np.array([len(t) for t in tails], dtype=np.uint64)
        toff = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        idx = np.ascontiguousarray(np.concatenate([np.asarray(t, np.uint64) for t in tails]))
        nl = [np.asarray(n, np.float64).reshape(-1, 3) for n in normals]
        noff = np.concatenate([[0], np.cumsum([len(n) for n in nl])]).astype(np.uint64)
        nrm = np.ascontiguousarray(np.concatenate(nl))
        bo = np.ascontiguousarray(np.concatenate([np.asarray(b, np.uint8) for b in bond_orders])) if bond_orders is not None else None
        if bo is not None and len(bo) != int(toff[-1]) - len(tails):
            raise MolarHipError(9, "for N tail carbons # of bond orders should be N-1")       # LipidOrderError::BondOrderCount
        nout = max(int(toff[-1]) - 2 * len(tails), 0)
        out = np.zeros(max(nout, 1), np.float64)
        check(self.lib.molar_hip_lipid_tail_order_f64(self.ctx, xa, xyz.shape[0], idx.ctypes.data, toff.ctypes.data, len(tails),
                                                      int(order_type), nrm.ctypes.data, noff.ctypes.data,
                                                      None if bo is None else bo.ctypes.data, out.ctypes.data))
        res, pos = [], 0
        for n in lens:
            res.append(out[pos:pos + int(n) - 2].copy()); pos += int(n) - 2
        return res

    def rotate(self, xyz, unit_axis, angle, idx=None):
        if not _is_torch(xyz):
            assert xyz.dtype == np.float64 and xyz.flags.c_contiguous, "rotate works in place"
        a = self._sel(xyz, idx)
        ax = np.ascontiguousarray(unit_axis, np.float64)
        check(self.lib.molar_hip_rotate_f64(self.ctx, *a[:4], ax.ctypes.data, float(angle)))

    def principal_transform(self, xyz, mass, idx=None, box=None):
        """(R, t) of Translation(cm) * Rotation(axes^-1) * Translation(-cm); principal_transform_pbc with a box."""
        a = self._sel(xyz, idx)
        mass = _f64(mass); ma, km = _addr(mass)
        ba, kb = self._box9(box) if box is not None else (None, None)
        R = np.zeros(9, np.float64); t = np.zeros(3, np.float64)
        check(self.lib.molar_hip_principal_transform_f64(self.ctx, *a[:4], ma, ba, R.ctypes.data, t.ctypes.data))
        return R.reshape(3, 3).T.copy(), t

    def apply_transform(self, xyz, R, t, idx=None):
        """In place on xyz (numpy float64 C-contiguous array or torch CUDA tensor)."""
        if not _is_torch(xyz):
            assert xyz.dtype == np.float64 and xyz.flags.c_contiguous, "apply_transform works in place"
        a = self._sel(xyz, idx)
        R9 = np.ascontiguousarray(np.asarray(R, np.float64).reshape(3, 3).T).reshape(9)
        t3 = np.ascontiguousarray(t, np.float64)
        check(self.lib.molar_hip_apply_transform_f64(self.ctx, *a[:4], R9.ctypes.data, t3.ctypes.data))


_MEMBRANE_FIELDS = ("head_markers", "normals", "valid", "quad_coefs", "mean_curv", "gauss_curv", "princ_curvs",
                    "princ_dirs", "area", "nvert", "neib_ids", "voro_vertexes", "fitted_patch_points")


class _MembranePatches(C.Structure):     # molar_hip_membrane_patches
    _fields_ = [("nlipids", C.c_size_t), ("patch_offsets", C.c_void_p), ("patch_ids", C.c_void_p)]


class _MembraneState(C.Structure):       # molar_hip_membrane_state
    _fields_ = [(k, C.c_void_p) for k in _MEMBRANE_FIELDS]


def new_membrane_state(head_markers, normals, valid=None, npatch_entries=0):
    """Per-lipid state with the defaults of Membrane::new (molar_membrane/src/lib.rs:152-177)."""
    head = np.array(head_markers, np.float32, order="C").reshape(-1, 3)
    K = len(head)
    slots = max(npatch_entries + 4 * K, 1)
    return dict(
        head_markers=head, normals=np.array(normals, np.float32, order="C").reshape(K, 3),
        valid=np.ones(K, np.uint8) if valid is None else np.array(valid, np.uint8, order="C"),
        quad_coefs=np.zeros((K, 6), np.float32), mean_curv=np.full(K, -100.0, np.float32),
        gauss_curv=np.full(K, -100.0, np.float32), princ_curvs=np.zeros((K, 2), np.float32),
        princ_dirs=np.zeros((K, 2, 3), np.float32), area=np.zeros(K, np.float32), nvert=np.zeros(K, np.uint32),
        neib_ids=np.zeros(slots, np.uint64), voro_vertexes=np.zeros((slots, 3), np.float32),
        fitted_patch_points=np.zeros((max(npatch_entries, 1), 3), np.float32))


class MembranePlan:
    """molar_hip_membrane_plan: one frame of Membrane::compute (molar_membrane/src/lib.rs:410-454) per begin/end pair,
    chained on the engine's stream without a host round trip inside; two frames may be in flight.  `valid` is carried
    from frame to frame on the device like LipidMolecule::valid."""

    # (dtype, shape as a function of K, E, slots, norder)
    _SHAPES = {
        "head": (np.float32, lambda K, E, S, N: (K, 3)), "mid": (np.float32, lambda K, E, S, N: (K, 3)),
        "tail": (np.float32, lambda K, E, S, N: (K, 3)), "patch_offsets": (np.uint64, lambda K, E, S, N: (K + 1,)),
        "patch_ids": (np.uint64, lambda K, E, S, N: (E,)), "initial_normals": (np.float32, lambda K, E, S, N: (K, 3)),
        "valid": (np.uint8, lambda K, E, S, N: (K,)), "smoothed_head": (np.float32, lambda K, E, S, N: (K, 3)),
        "normals": (np.float32, lambda K, E, S, N: (K, 3)), "quad_coefs": (np.float32, lambda K, E, S, N: (K, 6)),
        "mean_curv": (np.float32, lambda K, E, S, N: (K,)), "gauss_curv": (np.float32, lambda K, E, S, N: (K,)),
        "princ_curvs": (np.float32, lambda K, E, S, N: (K, 2)), "princ_dirs": (np.float32, lambda K, E, S, N: (K, 2, 3)),
        "area": (np.float32, lambda K, E, S, N: (K,)), "nvert": (np.uint32, lambda K, E, S, N: (K,)),
        "neib_ids": (np.uint64, lambda K, E, S, N: (S,)), "voro_vertexes": (np.float32, lambda K, E, S, N: (S, 3)),
        "fitted_patch_points": (np.float32, lambda K, E, S, N: (E, 3)), "order": (np.float32, lambda K, E, S, N: (N,)),
    }

    def __init__(self, engine, natoms, lipid_idx, lipid_off, marker_idx, marker_off, masses, tail_idx, tail_off, tail_lipid,
                 tail_bonds, cutoff, order_type, max_smooth_iter=1, unwrap=True, global_normal=None):
        self.eng = engine
        self.lib = engine.lib
        keep = [np.ascontiguousarray(lipid_idx, np.uint64), np.ascontiguousarray(lipid_off, np.uint64),
                np.ascontiguousarray(marker_idx, np.uint64), np.ascontiguousarray(marker_off, np.uint64),
                np.ascontiguousarray(masses, np.float32), np.ascontiguousarray(tail_idx, np.uint64),
                np.ascontiguousarray(tail_off, np.uint64), np.ascontiguousarray(tail_lipid, np.uint32),
                None if tail_bonds is None else np.ascontiguousarray(tail_bonds, np.uint8)]
        d = MembraneDesc()
        d.natoms = int(natoms); d.nlipids = len(keep[1]) - 1
        d.lipid_idx, d.lipid_offsets, d.marker_idx, d.marker_offsets, d.masses = (a.ctypes.data for a in keep[:5])
        d.ntails = len(keep[6]) - 1
        d.tail_idx, d.tail_offsets, d.tail_lipid = keep[5].ctypes.data, keep[6].ctypes.data, keep[7].ctypes.data
        d.tail_bonds = None if keep[8] is None else keep[8].ctypes.data
        d.cutoff = float(cutoff); d.order_type = int(order_type); d.max_smooth_iter = int(max_smooth_iter); d.unwrap = int(bool(unwrap))
        d.use_global_normal = int(global_normal is not None)
        g = np.zeros(3, np.float32) if global_normal is None else np.asarray(global_normal, np.float32).reshape(3)
        d.global_normal[:] = [float(v) for v in g]
        self.K = int(d.nlipids)
        self.norder = int(keep[6][-1]) - 2 * int(d.ntails)
        self.handle = C.c_void_p()
        check(self.lib.molar_hip_membrane_plan_create(engine.ctx, C.byref(d), C.byref(self.handle)))
        engine._adopt(self)
        self._views = {}
        self._keep = {}

    def close(self):
        if getattr(self, "handle", None) and getattr(self.eng, "ctx", None):
            self.lib.molar_hip_membrane_plan_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_valid(self, valid=None):
        """valid flags for the frames to come (None: reset_valid_lipids, lib.rs:269-273)."""
        v = None if valid is None else np.ascontiguousarray(valid, np.uint8)
        check(self.lib.molar_hip_membrane_plan_set_valid(self.handle, None if v is None else v.ctypes.data))

    def begin(self, xyz, box):
        """Enqueue one frame; xyz: float32 [N,3] torch CUDA tensor (unwrapped in place) or numpy array (unwrapped in place
        as well, through a copy).  Returns the ticket."""
        pb = box if isinstance(box, PeriodicBox) else PeriodicBox.from_matrix(box)
        m9 = pb.colmajor9()
        xa, kx = _addr(xyz)
        t = C.c_int32(-1)
        check(self.lib.molar_hip_membrane_frame_begin(self.handle, xa, m9.ctypes.data, C.byref(t)))
        self._keep[t.value] = kx
        return t.value

    _PATCH_SIZED = ("patch_ids", "neib_ids", "voro_vertexes", "fitted_patch_points")

    def end(self, ticket, names=None):
        """Wait for the frame; returns its MembraneView (device addresses, sizes) - or, with `names` (per-lipid arrays and
        "order"), the pair (view, dict of numpy arrays): the arrays leave behind the frame's last kernel, one wait for both
        (molar_hip_membrane_frame_end_fetch)."""
        v = MembraneView()
        out = None
        if names is None:
            rc = self.lib.molar_hip_membrane_frame_end(self.handle, int(ticket), C.byref(v))
        else:
            if any(k in self._PATCH_SIZED for k in names):
                raise ValueError("end(names=...): per-lipid arrays and order only; fetch() brings the patch-sized arrays")
            out, o = self._arrays(names, self.K, 0)
            rc = self.lib.molar_hip_membrane_frame_end_fetch(self.handle, int(ticket), C.byref(v), C.byref(o))
        if rc == 0 or int(ticket) not in (0, 1) or v.nlipids:      # the frame is over (even if one of its stages failed)
            self._keep.pop(int(ticket), None)
        check(rc)
        self._views[int(ticket)] = v
        return v if names is None else (v, out)

    def _arrays(self, names, K, E):
        out, o = {}, MembraneOut()
        for k in names:
            dt, shp = self._SHAPES[k]
            a = np.empty(shp(K, E, E + 4 * K, self.norder), dt)         # (every element is written by the fetch)
            out[k] = a
            setattr(o, k, a.ctypes.data if a.size else None)
        return out, o

    def fetch(self, ticket, names=MEMBRANE_ARRAYS):
        """The named arrays of an ended frame as numpy arrays."""
        v = self._views[int(ticket)]
        out, o = self._arrays(names, self.K, int(v.patch_entries))
        check(self.lib.molar_hip_membrane_frame_fetch(self.handle, int(ticket), C.byref(o)))
        return out


def histogram_edges(hmin, hmax, nbins):
    """Exact bin edges of Histogram1D::add_one (molar_membrane/src/stats.rs:29-35) over the SQUARED distance: float32
    [nbins + 1], edges[b] = the smallest d2 >= 0 whose bin is >= b (host arithmetic of the engine, no GPU)."""
    e = np.zeros(nbins + 1, np.float32)
    check(_lib.load().molar_hip_histogram_edges(float(hmin), float(hmax), int(nbins), e.ctypes.data))
    return e


def membrane_patches_from_pairs(pairs, nlipids):
    """compute_patches' list building (molar_membrane/src/lib.rs:548-557): CSR (offsets, ids) in push order."""
    pairs = np.ascontiguousarray(pairs, np.uint32).reshape(-1, 2)
    off = np.zeros(nlipids + 1, np.uint64)
    ids = np.zeros(max(2 * len(pairs), 1), np.uint64)
    check(_lib.load().molar_hip_membrane_patches_from_pairs(pairs.ctypes.data, len(pairs), nlipids, off.ctypes.data, ids.ctypes.data))
    return off, ids[: 2 * len(pairs)]


def membrane_nth_shell_patches(valid, patch_offsets, patch_ids, nvert, neib_ids, n_shells):
    """patches_from_nth_shell (molar_membrane/src/lib.rs:562-583): new patch CSR (offsets, ids) - the n-th Voronoi
    neighbour shell for valid lipids (ids ascending), the old patch for the others.  Host arithmetic of the engine."""
    lib = _lib.load()
    v = np.ascontiguousarray(valid, np.uint8); po = np.ascontiguousarray(patch_offsets, np.uint64)
    pi = np.ascontiguousarray(patch_ids, np.uint64); nv = np.ascontiguousarray(nvert, np.uint32)
    nb = np.ascontiguousarray(neib_ids, np.uint64)
    K = len(v)
    off = np.zeros(K + 1, np.uint64)
    need = C.c_size_t(0)
    args = (K, v.ctypes.data, po.ctypes.data, pi.ctypes.data if len(pi) else None, nv.ctypes.data, nb.ctypes.data, int(n_shells))
    check(lib.molar_hip_membrane_nth_shell_patches(*args, off.ctypes.data, None, 0, C.byref(need)))
    ids = np.zeros(max(need.value, 1), np.uint64)
    check(lib.molar_hip_membrane_nth_shell_patches(*args, off.ctypes.data, ids.ctypes.data, need.value, C.byref(need)))
    return off, ids[: need.value]


def membrane_smooth_curvature(valid, patch_offsets, nvert, neib_ids, n_shells, mean_curv, gauss_curv):
    """smooth_curvature (lib.rs:584-621): returns the smoothed (mean, gaussian) curvature arrays."""
    lib = _lib.load()
    v = np.ascontiguousarray(valid, np.uint8); po = np.ascontiguousarray(patch_offsets, np.uint64)
    nv = np.ascontiguousarray(nvert, np.uint32); nb = np.ascontiguousarray(neib_ids, np.uint64)
    m = np.array(mean_curv, np.float32, order="C"); g = np.array(gauss_curv, np.float32, order="C")
    check(lib.molar_hip_membrane_smooth_curvature(len(v), v.ctypes.data, po.ctypes.data, nv.ctypes.data, nb.ctypes.data, int(n_shells),
                                                  m.ctypes.data, g.ctypes.data))
    return m, g


def membrane_initial_normals(head_markers, tail_markers, patch_offsets, patch_ids, valid=None, normals=None):
    """Membrane::compute_initial_normals (molar_membrane/src/lib.rs:456-505); host arithmetic of the engine."""
    lib = _lib.load()
    head = np.ascontiguousarray(head_markers, np.float32); tail = np.ascontiguousarray(tail_markers, np.float32)
    po = np.ascontiguousarray(patch_offsets, np.uint64); pi = np.ascontiguousarray(patch_ids, np.uint64)
    K = len(head)
    out = np.zeros((K, 3), np.float32) if normals is None else np.ascontiguousarray(normals, np.float32)
    v = None if valid is None else np.ascontiguousarray(valid, np.uint8)
    check(lib.molar_hip_membrane_initial_normals(K, head.ctypes.data, tail.ctypes.data, po.ctypes.data,
                                                 pi.ctypes.data if len(pi) else None, None if v is None else v.ctypes.data,
                                                 out.ctypes.data))
    return out


# ---------------------------------------------------------------- pymolar-style front-end

_default_engine = None


def default_engine() -> Engine:
    global _default_engine
    if _default_engine is None:
        _default_engine = Engine(0)
    return _default_engine


class State:
    """Coordinates + box of one frame (state.rs:22-28)."""

    def __init__(self, coords, box: PeriodicBox | None = None, time: float = 0.0):
        self.coords = _f32(coords)
        self.pbox = box
        self.time = time

    def __len__(self):
        return self.coords.shape[0]


class Topology:
    """Only the per-atom columns the path reads: masses and vdW radii (atom_storage.rs:272)."""

    def __init__(self, masses, vdw=None):
        self.masses = _f32(masses)
        self.vdw = _f32(vdw)


def rotation_from_axis_angle(axis, angle):
    """nalgebra Rotation3::from_axis_angle (Rodrigues), f32; `axis` is normalised like Unit::new_normalize."""
    f = np.float32
    u = np.asarray(axis, f)
    u = (u / f(np.sqrt(f(f(u[0] * u[0] + u[1] * u[1]) + u[2] * u[2])))).astype(f)
    ux, uy, uz = u
    s, c = f(np.sin(f(angle))), f(np.cos(f(angle)))
    k = f(1) - c
    sqx, sqy, sqz = ux * ux, uy * uy, uz * uz
    return np.array([[sqx + (f(1) - sqx) * c, ux * uy * k - uz * s, ux * uz * k + uy * s],
                     [ux * uy * k + uz * s, sqy + (f(1) - sqy) * c, uy * uz * k - ux * s],
                     [ux * uz * k - uy * s, uy * uz * k + ux * s, sqz + (f(1) - sqz) * c]], dtype=f)


class Sel:
    """A bound selection: sorted, non-empty index set over (Topology, State) (sel.rs:10-31)."""

    def __init__(self, top: Topology, state: State, index=None, engine: Engine | None = None):
        n = len(state)
        if index is None:
            index = np.arange(n, dtype=np.uint64)
        index = np.unique(np.asarray(index, dtype=np.uint64))     # SVec: sorted + dedup
        if len(index) == 0:
            raise ValueError("selection is empty")                 # sel.rs:13-19
        self.top, self.state, self.index = top, state, index
        self._engine = engine

    @property
    def engine(self):
        """The GPU context is created on first use, so selections can be built without a device."""
        if self._engine is None:
            self._engine = default_engine()
        return self._engine

    def __len__(self):
        return len(self.index)

    def require_box(self) -> PeriodicBox:
        if self.state.pbox is None:
            raise MolarHipError(4, "pbc operation without periodic box")
        return self.state.pbox

    # molar_python/src/selection.rs:816-829 — com(dims) always goes through the pbc variant
    def com(self, dims=None):
        if dims is None or pbc_mask(dims) == 0:
            return self.engine.center_of_mass(self.state.coords, self.top.masses, self.index)
        return self.engine.center_of_mass_pbc(self.state.coords, self.top.masses, self.require_box(), dims, self.index)

    def cog(self, dims=None):
        if dims is None or pbc_mask(dims) == 0:
            return self.engine.center_of_geometry(self.state.coords, self.index)
        return self.engine.center_of_geometry_pbc(self.state.coords, self.require_box(), dims, self.index)

    def center_of_mass(self):
        return self.engine.center_of_mass(self.state.coords, self.top.masses, self.index)

    def center_of_geometry(self):
        return self.engine.center_of_geometry(self.state.coords, self.index)

    def gyration(self):
        return self.engine.gyration(self.state.coords, self.top.masses, self.index)

    def gyration_pbc(self):
        return self.engine.gyration(self.state.coords, self.top.masses, self.index, self.require_box())

    def inertia(self):
        m, a, _ = self.engine.inertia(self.state.coords, self.top.masses, self.index)
        return m, a

    def inertia_pbc(self):
        m, a, _ = self.engine.inertia(self.state.coords, self.top.masses, self.index, self.require_box())
        return m, a

    def min_max(self):
        return self.engine.min_max(self.state.coords, self.index)

    # measure.rs:100-109, 246-257, 646-649: T(cm) * inverse(axes) * T(-cm), returned as (R, t) of p -> R p + t
    def principal_transform(self):
        return self.engine.principal_transform(self.state.coords, self.top.masses, self.index)

    def principal_transform_pbc(self):
        return self.engine.principal_transform(self.state.coords, self.top.masses, self.index, self.require_box())

    # modify.rs:16-30
    def translate(self, shift):
        self.engine.translate(self.state.coords, shift, self.index)

    def rotate(self, axis, angle):
        """Rotation3::from_axis_angle about a unit axis through the origin (modify.rs:25-30); `axis` is normalised
        like Unit::new_normalize."""
        f = np.float32
        u = np.asarray(axis, f)
        u = (u / f(np.sqrt(f(f(u[0] * u[0] + u[1] * u[1]) + u[2] * u[2])))).astype(f)
        self.engine.rotate(self.state.coords, u, angle, self.index)

    def apply_transform(self, tr):
        R, t = tr
        self.engine.apply_transform(self.state.coords, R, t, self.index)

    def unwrap_simple(self, dims=PBC_FULL):
        self.engine.unwrap_simple(self.state.coords, self.require_box(), dims, self.index)

    def within(self, cutoff, inner: "Sel", pbc=None, include_inner=False):
        """`within <cutoff> [pbc xyz] [self] of <inner>` evaluated inside this selection
        (LogicalNode::Within, selection/ast.rs:589-631): atoms of `self` closer than cutoff to any atom
        of `inner`; the raw search stream is sorted + de-duplicated like SVec::from_unsorted
        (selection_expr.rs:112); `include_inner` is the `self` keyword (:627-629)."""
        mask = pbc_mask(pbc)
        eng = self.engine
        if mask == 0:
            lo, up = self.min_max()                                   # ast.rs:600-602
            lo = lo + (np.float32(-cutoff) - np.float32(1.1920929e-07))
            up = up + (np.float32(cutoff) + np.float32(1.1920929e-07))
            ids = eng.within_set(cutoff, self.state.coords, self.index, inner.state.coords, inner.index, lower=lo, upper=up)
        else:
            ids = eng.within_set(cutoff, self.state.coords, self.index, inner.state.coords, inner.index,
                                 box=self.require_box(), pbc=mask)
        if include_inner:
            ids = np.union1d(ids, inner.index)
        return ids.astype(np.uint64)

    def unwrap_connectivity(self, cutoff, dims=PBC_FULL):
        """Modify::unwrap_connectivity_dim (modify.rs:72-131): neighbour search with LOCAL ids under full
        PBC (:77-78), adjacency in pair order (SearchConnectivity, connectivity.rs:19-35), then the
        reference's stack walk that pulls every connected atom to the closest image of the atom it was
        reached from - all inside the library (molar_hip_unwrap_connectivity).  Coordinates are modified in place;
        returns the list of local-index groups the reference returns as selections."""
        return self.engine.unwrap_connectivity(self.state.coords, self.require_box(), cutoff, dims, self.index)


def distance_search(cutoff, data1: Sel, data2: Sel | None = None, dims=None):
    """molar_python/src/lib.rs:259-376 — same dispatch table:
    float & data2 & any dim -> double_pbc; float & data2 -> double; float & any dim -> single_pbc;
    float -> single; "vdw" needs data2 (local ids converted to global, :348-354);
    "vdw" with one selection -> NotImplementedError (:355-358)."""
    eng = data1.engine
    pbc = pbc_mask(dims)
    if isinstance(cutoff, str):
        if cutoff != "vdw":
            raise TypeError(f"Unknown cutoff type {cutoff}")
        if data2 is None:
            raise NotImplementedError("VdW distance search is not yet supported for single selection")
        v1 = data1.top.vdw[data1.index.astype(np.int64)]
        v2 = data2.top.vdw[data2.index.astype(np.int64)]
        box = data1.require_box() if pbc else None
        n = eng.search_count(SEARCH_DOUBLE_VDW, None, data1.state.coords, data1.index, data2.state.coords,
                             data2.index, box=box, pbc=pbc, vdw1=v1, vdw2=v2, ids_local=True)
        i, j, d = eng.search_fill_usize(n)
        i = data1.index[i.astype(np.int64)]
        j = data2.index[j.astype(np.int64)]
        return np.stack([i, j], 1), d
    if not isinstance(cutoff, (float, int, np.floating, np.integer)):
        raise TypeError("cutoff must be a float or 'vdw'")
    if data2 is not None:
        box = data1.require_box() if pbc else None
        n = eng.search_count(SEARCH_DOUBLE, cutoff, data1.state.coords, data1.index, data2.state.coords, data2.index,
                             box=box, pbc=pbc)
    else:
        box = data1.require_box() if pbc else None
        n = eng.search_count(SEARCH_SINGLE, cutoff, data1.state.coords, data1.index, box=box, pbc=pbc)
    i, j, d = eng.search_fill_usize(n)
    return np.stack([i, j], 1), d


def fit_transform(sel1: Sel, sel2: Sel):
    return sel1.engine.fit_transform(sel1.state.coords, sel1.top.masses, sel2.state.coords, sel2.top.masses,
                                     sel1.index, sel2.index)


def fit_transform_at_origin(sel1: Sel, sel2: Sel):
    return sel1.engine.fit_transform(sel1.state.coords, sel1.top.masses, sel2.state.coords, sel2.top.masses,
                                     sel1.index, sel2.index, at_origin=True)


def rmsd(sel1: Sel, sel2: Sel) -> float:
    return sel1.engine.rmsd(sel1.state.coords, sel2.state.coords, sel1.index, sel2.index)


rmsd_py = rmsd


def rmsd_mw(sel1: Sel, sel2: Sel) -> float:
    return sel1.engine.rmsd_mw(sel1.state.coords, sel1.top.masses, sel2.state.coords, sel1.index, sel2.index)
