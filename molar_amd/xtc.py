"""XTC trajectory reader over the engine's C ABI — the host-side mirror of MolAR's XtcFileHandler
(molar/src/io/xtc_handler.rs) and of the state iterator that feeds AnalysisTask (io.rs:198-271):

    read_state()          next frame as a State (coords nm, time, PeriodicBox)      xtc_handler.rs:64-112
    seek_frame(fr)        position on frame fr                                      :200-218
    seek_time(t)          position on the first frame with time >= t                :220-229, 282-297
    iteration             frames until Eof                                          io.rs:198-271

`read_frames(first, count, out=...)` is the batched entry the GPU path uses: frames are decoded in parallel on
host threads straight into a torch/HIP device buffer (or a numpy array).

    encode_frame / XtcWriter   write_state (xtc_handler.rs:117-168): GROMACS' compressed frames, written by the library
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .api import Engine, PeriodicBox, State, check, _addr


def encode_frame(xyz, box9, step=0, time=0.0, precision=1000.0) -> bytes:
    """One XTC frame (molar_hip_xtc_encode_frame): xyz float32 [natoms, 3] in nm, box9 = the 9 floats the file stores."""
    lib = _lib.load()
    xyz = np.ascontiguousarray(xyz, np.float32)
    box9 = np.ascontiguousarray(box9, np.float32).reshape(9)
    n = xyz.shape[0] if xyz.ndim == 2 else xyz.size // 3
    out = np.empty(96 + 16 * n, np.uint8)
    ln = C.c_size_t(0)
    check(lib.molar_hip_xtc_encode_frame(xyz.ctypes.data, n, box9.ctypes.data, int(step), float(time), float(precision),
                                         out.ctypes.data, out.nbytes, C.byref(ln)))
    return out[:ln.value].tobytes()


class XtcWriter:
    """FileFormatHandler::create + write_state for XTC (xtc_handler.rs:54-62, 117-168): frames appended to one file."""

    def __init__(self, path, precision=1000.0):
        self.f = open(path, "wb")
        self.precision = precision
        self.nframes = 0

    def write_state(self, state: State, step=None):
        # the file's 9 floats fill the matrix column by column (read_state below): the column-major form
        box9 = np.zeros(9, np.float32) if state.pbox is None else np.asarray(state.pbox.colmajor9(), np.float32).reshape(9)
        self.write(state.coords, box9, self.nframes if step is None else step, state.time)

    def write(self, xyz, box9, step, time):
        self.f.write(encode_frame(xyz, box9, step, time, self.precision))
        self.nframes += 1

    def close(self):
        if self.f:
            self.f.close()
            self.f = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class XtcReader:
    def __init__(self, source, engine: Engine | None = None, nthreads: int = 0):
        self.lib = _lib.load()
        self.engine = engine
        self.nthreads = nthreads
        self._keep = None
        if isinstance(source, (bytes, bytearray, memoryview, np.ndarray)):
            buf = np.frombuffer(source, np.uint8) if not isinstance(source, np.ndarray) else np.ascontiguousarray(source, np.uint8)
            self._keep = buf
            self.h = self.lib.molar_hip_xtc_open_memory(buf.ctypes.data, len(buf))
        else:
            self.h = self.lib.molar_hip_xtc_open(str(source).encode())
        if not self.h:
            raise _lib.MolarHipError(53, self.lib.molar_hip_last_error().decode())
        self.cur_fr = 0

    def close(self):
        if getattr(self, "h", None):
            self.lib.molar_hip_xtc_close(self.h)
            self.h = None

    __del__ = close

    def __len__(self):
        return int(self.lib.molar_hip_xtc_nframes(self.h))

    @property
    def natoms(self):
        return int(self.lib.molar_hip_xtc_natoms(self.h))

    def frame_info(self, fr):
        nat, step, t, prec = C.c_int32(0), C.c_int32(0), C.c_float(0), C.c_float(0)
        box = np.zeros(9, np.float32)
        check(self.lib.molar_hip_xtc_frame_info(self.h, fr, C.byref(nat), C.byref(step), C.byref(t), box.ctypes.data, C.byref(prec)))
        return dict(natoms=nat.value, step=step.value, time=t.value, precision=prec.value, box9=box)

    def read_frames(self, first, count, out=None, nthreads=None):
        """Decode frames [first, first+count) into out[count, natoms, 3] (numpy array, torch CUDA tensor or None)."""
        nat = self.frame_info(first)["natoms"] if count else 0
        if out is None:
            out = np.empty((count, nat, 3), np.float32)
        addr, keep = _addr(out)
        ctx = self.engine.ctx if self.engine is not None else None
        check(self.lib.molar_hip_xtc_read(ctx, self.h, first, count, addr, self.nthreads if nthreads is None else nthreads))
        return out

    def read_frames_device(self, first, count, out):
        """Decode frames [first, first+count) ON THE GPU (one lane per frame, molar_hip_xtc_read_device) into the torch CUDA
        tensor out[count, natoms, 3]: the compressed bytes cross the link, not the coordinates.  For windows of hundreds to
        thousands of frames; bit-identical to read_frames."""
        if self.engine is None:
            raise ValueError("read_frames_device needs an engine")
        # the C entry point takes an address and no capacity: everything it relies on is checked here
        import torch
        need = count * (self.frame_info(first)["natoms"] if count else 0) * 3
        if not (isinstance(out, torch.Tensor) and out.is_cuda and out.dtype == torch.float32 and out.is_contiguous()):
            raise ValueError("read_frames_device: `out` must be a contiguous float32 CUDA tensor")
        if out.device.index != self.engine.device:
            raise ValueError(f"read_frames_device: `out` lives on cuda:{out.device.index}, the engine on cuda:{self.engine.device}")
        if out.numel() < need:
            raise ValueError(f"read_frames_device: `out` holds {out.numel()} floats, {count} frames need {need}")
        addr, keep = _addr(out)
        check(self.lib.molar_hip_xtc_read_device(self.engine.ctx, self.h, first, count, addr))
        return out

    def histogram(self, first, count, cutoff, hmin, hmax, nbins, idx=None, pbc=7, bins=None, nthreads=None, idx2=None, two_sets=False):
        """molar_hip_xtc_histogram: the distances of distance_search_single_pbc of frames [first, first + count) - selection idx
        (None = all atoms) of every frame against itself, every frame's own box - through Histogram1D::add_one, decode and the
        fused histogram overlapped inside the call.  With `idx2` (or two_sets=True: idx against all atoms) the distances are those
        of distance_search_double_pbc between the two selections (molar_hip_xtc_histogram_double).  Adds into `bins` (numpy
        uint64[nbins], made if None) and returns it."""
        if self.engine is None:
            raise ValueError("histogram needs an engine")
        from .api import pbc_mask
        if bins is None:
            bins = np.zeros(nbins, np.uint64)
        assert bins.dtype == np.uint64 and bins.flags.c_contiguous and len(bins) == nbins
        idx = None if idx is None else np.ascontiguousarray(idx, np.uint64)
        idx2 = None if idx2 is None else np.ascontiguousarray(idx2, np.uint64)
        nt = self.nthreads if nthreads is None else nthreads
        a1, n1 = (None, 0) if idx is None else (idx.ctypes.data, len(idx))
        if idx2 is not None or two_sets:
            a2, n2 = (None, 0) if idx2 is None else (idx2.ctypes.data, len(idx2))
            check(self.lib.molar_hip_xtc_histogram_double(self.engine.ctx, self.h, first, count, a1, n1, a2, n2, float(cutoff), pbc_mask(pbc),
                                                          float(hmin), float(hmax), nbins, bins.ctypes.data, nt))
        else:
            check(self.lib.molar_hip_xtc_histogram(self.engine.ctx, self.h, first, count, a1, n1, float(cutoff), pbc_mask(pbc), float(hmin),
                                                   float(hmax), nbins, bins.ctypes.data, nt))
        return bins

    # ---- FileFormatHandler mirror
    def seek_frame(self, fr):
        if fr > len(self):
            raise _lib.MolarHipError(53, f"seek to frame {fr} failed")
        self.cur_fr = fr

    def seek_time(self, t):
        fr = C.c_size_t(0)
        check(self.lib.molar_hip_xtc_seek_time(self.h, C.c_float(t), C.byref(fr)))
        self.cur_fr = fr.value

    def read_state(self) -> State:
        if self.cur_fr >= len(self):
            raise EOFError("end of trajectory")                     # FileFormatError::Eof
        info = self.frame_info(self.cur_fr)
        xyz = self.read_frames(self.cur_fr, 1, nthreads=1)[0]
        self.cur_fr += 1
        box = PeriodicBox.from_matrix(info["box9"].reshape(3, 3).T)    # 9 floats fill the matrix column by column
        return State(xyz, box, info["time"])

    def __iter__(self):
        while self.cur_fr < len(self):
            yield self.read_state()
