"""XTC trajectory reader over the engine's C ABI — the host-side mirror of MolAR's XtcFileHandler
(molar/src/io/xtc_handler.rs) and of the state iterator that feeds AnalysisTask (io.rs:198-271):

    read_state()          next frame as a State (coords nm, time, PeriodicBox)      xtc_handler.rs:64-112
    seek_frame(fr)        position on frame fr                                      :200-218
    seek_time(t)          position on the first frame with time >= t                :220-229, 282-297
    iteration             frames until Eof                                          io.rs:198-271

`read_frames(first, count, out=...)` is the batched entry the GPU path uses: frames are decoded in parallel on
host threads straight into a torch/HIP device buffer (or a numpy array).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .api import Engine, PeriodicBox, State, check, _addr


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
        addr, keep = _addr(out)
        check(self.lib.molar_hip_xtc_read_device(self.engine.ctx, self.h, first, count, addr))
        return out

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
