"""Trajectory-analysis plugin surface for Python callers of the engine.

Written from the behavioural contract of MolAR's task driver (molar/src/analysis_task.rs), not from any Python source:

  command line   -f/--files STRUCTURE TRAJ [TRAJ ...]   --log N   -b/--begin BOUND   -e/--end BOUND   --skip N
                 --use_struct_file   (analysis_task.rs:12-38) and --add-time (time offset across files)
  BOUND          "" = open; "42" or "42fr" = absolute frame index; "5ps" / "1.5ns" / "2us" = time, converted to ps;
                 anything else is an error (process_suffix, :82-110; its five test groups :329-366 are replayed in
                 tests/test_analysis_task_py_cpu.py)
  stream         all trajectory files form ONE stream: frame indices, the begin/end window and the --skip cadence run
                 across file boundaries (:181-184); with a single trajectory `begin` is reached by seeking (:187-197),
                 with several it is filtered frame by frame against the global position (:203-214)
  end            exclusive for a frame bound (global index >= end stops everything), inclusive for a time bound
                 (time > end stops) (:218-222)
  skip           phase counted from the begin frame, which is therefore always processed (:228-234)
  hooks          the task object is initialised on the first frame that passes the window (pre_process), every passing
                 frame goes to process_frame, post_process runs once at the end; a run that passed no frame at all is an
                 error (:236-277)

The hook names (`register_args`, `pre_process`, `process_frame`, `post_process`) and the fields a task sees (`args`,
`top`, `state`, `src`, `consumed_frames`, `trj_ind`) are those of the reference's Python front-end; `consumed_frames`
already counts the frame being processed when `process_frame` runs.

The pieces are separate on purpose: `parse_bound` (text -> Bound), `FrameWindow` (the begin/end/skip decisions, no IO),
`frame_stream` (files -> admitted frames) and `AnalysisTask` (hooks + CLI).  File formats are outside the engine: `.xtc`
goes through the engine's reader, anything else through the overridable `open_trajectory` / `read_topology`."""
from __future__ import annotations

import argparse
import enum
import logging
from dataclasses import dataclass
from typing import Callable, Iterable, Iterator, Optional, Sequence, Tuple

import numpy as np

from . import api

log = logging.getLogger("molar_amd.analysis")


class AnalysisError(Exception):
    """Base of the driver's errors (AnalysisError, analysis_task.rs:41-76)."""


class InvalidSuffix(AnalysisError):
    def __init__(self, text):
        super().__init__(f"invalid time suffix in {text!r}: 'fr', 'ps', 'ns', 'us' allowed")


class NoTrajectory(AnalysisError):
    def __init__(self):
        super().__init__("at least one trajectory required if 'use_struct_file' is not set")


class NoFramesConsumed(AnalysisError):
    def __init__(self):
        super().__init__("no frames consumed")


# ---------------------------------------------------------------------------------------------------------------------
# bounds and the window

_TIME_UNITS_PS = (("ps", 1.0), ("ns", 1.0e3), ("us", 1.0e6))


@dataclass(frozen=True)
class Bound:
    """One side of the processing window: an absolute frame index, a time in ps, or neither."""
    frame: Optional[int] = None
    time: Optional[float] = None

    @property
    def open(self) -> bool:
        return self.frame is None and self.time is None


def _unsigned(text: str) -> int:
    # Rust's usize::from_str: optional '+', decimal digits only
    body = text[1:] if text[:1] == "+" else text
    if not body.isascii() or not body.isdigit():
        raise ValueError(f"invalid frame number {text!r}")
    return int(body)


def parse_bound(text: str) -> Bound:
    """`-b` / `-e` value -> Bound (process_suffix, analysis_task.rs:82-110)."""
    text = text.strip()
    if not text:
        return Bound()
    try:
        return Bound(frame=_unsigned(text))
    except ValueError:
        pass
    if text.endswith("fr"):
        return Bound(frame=_unsigned(text[:-2].strip()))
    for unit, scale in _TIME_UNITS_PS:
        if text.endswith(unit):
            return Bound(time=float(np.float32(float(text[:-len(unit)].strip())) * np.float32(scale)))
    raise InvalidSuffix(text)


class Verdict(enum.Enum):
    BEFORE = 0      # in front of the window: not counted for the cadence
    TAKE = 1        # goes to the task
    PASS = 2        # inside the window but dropped by --skip
    STOP = 3        # window exhausted: nothing later can be admitted


class FrameWindow:
    """Decides, frame by frame, what happens to the frames of the concatenated stream.  Holds the two counters the
    decisions need: the absolute position in the stream and the cadence phase since `begin`."""

    def __init__(self, begin: Bound = Bound(), end: Bound = Bound(), skip: int = 1):
        if skip < 1:
            raise ValueError("--skip must be >= 1")
        self.begin, self.end, self.skip = begin, end, skip
        self.position = 0       # absolute index of the next frame to be judged
        self.phase = 0

    def jump(self, position: int) -> None:
        """The reader was moved to `position` by random access."""
        self.position = position

    def judge(self, time: float, filter_begin: bool) -> Verdict:
        if filter_begin:
            if self.begin.frame is not None:
                early = self.position < self.begin.frame
            elif self.begin.time is not None:
                early = time < self.begin.time
            else:
                early = False
            if early:
                self.position += 1
                return Verdict.BEFORE
        if (self.end.frame is not None and self.position >= self.end.frame) or \
           (self.end.time is not None and time > self.end.time):
            return Verdict.STOP
        taken = self.phase % self.skip == 0
        self.phase += 1
        self.position += 1
        return Verdict.TAKE if taken else Verdict.PASS


def frame_stream(trajectories: Sequence, window: FrameWindow, opener: Callable, add_time: bool = False
                 ) -> Iterator[Tuple[int, int, "api.State"]]:
    """Yields (trajectory index, absolute frame index, state) for every frame the window admits."""
    seekable = len(trajectories) == 1
    offset = 0.0
    for which, path in enumerate(trajectories):
        log.info("Processing trajectory '%s'...", path)
        reader = opener(path)
        if seekable:
            if window.begin.frame is not None:
                if window.begin.frame > 0:
                    reader.skip_to_frame(window.begin.frame)
                    window.jump(window.begin.frame)
            elif window.begin.time is not None:
                reader.skip_to_time(window.begin.time)
        last_time = None
        for state in reader:
            last_time = state.time
            shown = state.time + offset
            verdict = window.judge(shown, filter_begin=not seekable)
            if verdict is Verdict.STOP:
                return
            if verdict is Verdict.TAKE:
                state.time = shown
                yield which, window.position - 1, state
        log.info("Finished with '%s'.", path)
        if add_time and last_time is not None:
            offset += last_time


def format_time(t_ps: float) -> str:
    """get_log_time, analysis_task.rs:316-324."""
    for limit, div, unit in ((1.0e3, 1.0, "ps"), (1.0e6, 1.0e3, "ns")):
        if t_ps < limit:
            return f"{t_ps / div} {unit}"
    return f"{t_ps / 1.0e6} us"


# ---------------------------------------------------------------------------------------------------------------------
# what a task works on

class System:
    """Topology + current State (selection/system.rs); `sel(index)` binds a selection to the current state."""

    def __init__(self, top: api.Topology, state: api.State, engine: api.Engine | None = None):
        if len(top.masses) != len(state):
            raise ValueError("topology and state sizes differ")
        self.top, self.state, self.engine = top, state, engine

    def replace_state_deep(self, state: api.State):
        if len(state) != len(self.state):
            raise ValueError("states of different sizes")           # system.rs:230-236
        self.state = state

    set_state = replace_state_deep

    def sel(self, index=None) -> api.Sel:
        return api.Sel(self.top, self.state, index, engine=self.engine)


class _XtcSource:
    """The three things the stream needs from a trajectory file, over the engine's XTC reader."""

    def __init__(self, path, engine=None):
        from .xtc import XtcReader
        self.reader = XtcReader(path, engine=engine)

    def skip_to_frame(self, index: int) -> None:
        self.reader.seek_frame(index)

    def skip_to_time(self, t_ps: float) -> None:
        self.reader.seek_time(float(t_ps))

    def __iter__(self):
        return iter(self.reader)


class AnalysisTask:
    """Subclass, implement the hooks, construct: `MyTask()` parses sys.argv (or the `argv` list), streams the frames and
    calls `pre_process` (first admitted frame), `process_frame` (every admitted frame) and `post_process` (once)."""

    engine: api.Engine | None = None

    def __init__(self, argv: Optional[Iterable[str]] = None):
        logging.basicConfig(format="[%(levelname)s] (%(name)s) %(message)s", level=logging.INFO)
        log.info('Executing task "%s"...', type(self).__name__)
        self.args = self._parse(argv)
        self.top = self.state = self.src = None
        self.consumed_frames = 0
        self.trj_ind = 0
        self._run()

    # ---- command line
    def _parse(self, argv):
        cli = argparse.ArgumentParser(prog="analysis")
        cli.add_argument("-f", "--files", nargs="+", required=True, metavar="FILE",
                         help="structure file followed by the trajectories")
        cli.add_argument("--log", type=int, default=100, metavar="N", help="report every N-th processed frame")
        cli.add_argument("-b", "--begin", default="0", help="first frame (N, Nfr) or time (Nps, Nns, Nus)")
        cli.add_argument("-e", "--end", default="", help="end frame (exclusive) or time (inclusive)")
        cli.add_argument("--skip", type=int, default=1, metavar="N", help="process every N-th frame from begin")
        cli.add_argument("--use_struct_file", action="store_true", help="the structure file's coordinates are frame 0")
        cli.add_argument("--add-time", dest="add_time", action="store_true",
                         help="continue the clock of each trajectory from the end of the previous one")
        self.register_args(cli)
        args = cli.parse_args(None if argv is None else list(argv))
        if args.skip < 1:
            cli.error("--skip must be >= 1")
        if args.log < 1:
            cli.error("--log must be >= 1")
        return args

    # ---- the run
    def _accept(self, state, position):
        if self.consumed_frames % self.args.log == 0:
            log.info("At frame %d, time %s", position, format_time(state.time))
        self.state = state
        first = self.src is None
        if first:
            if self.top is None:
                self.top = self.read_topology(self.args.files[0], state)
            self.src = System(self.top, state, self.engine)
        else:
            self.src.replace_state_deep(state)
        self.consumed_frames += 1
        if first:
            self.pre_process()
        self.process_frame()

    def _run(self):
        a = self.args
        trajectories = a.files[1:]
        if not a.use_struct_file and not trajectories:
            raise NoTrajectory()
        window = FrameWindow(parse_bound(a.begin), parse_bound(a.end), a.skip)
        if a.use_struct_file:                                           # analysis_task.rs:170-181
            log.info("Using structure file for task initialization")
            self.top, state = self.read_structure(a.files[0])
            self._accept(state, 0)
        for which, position, state in frame_stream(trajectories, window, self.open_trajectory, a.add_time):
            self.trj_ind = which
            self._accept(state, position)
        if self.src is None:
            raise NoFramesConsumed()
        log.info("Post-processing...")
        self.post_process()

    # ---- format hooks (file formats other than XTC are outside the engine)
    def open_trajectory(self, path):
        if str(path).endswith(".xtc"):
            return _XtcSource(path, self.engine)
        raise AnalysisError(f"unsupported trajectory format: {path} (override open_trajectory)")

    def read_topology(self, path, first_state) -> api.Topology:
        """Default: an .npz with `masses` (and optionally `vdw`); otherwise unit masses for every atom of the first frame."""
        if str(path).endswith(".npz"):
            z = np.load(path)
            return api.Topology(z["masses"], z["vdw"] if "vdw" in z else None)
        return api.Topology(np.ones(len(first_state), np.float32))

    def read_structure(self, path):
        """Topology AND coordinates of the structure file (--use_struct_file).  Default: MolAR-style .gro."""
        if str(path).endswith(".gro"):
            from . import gro
            return gro.read_gro(path)
        raise AnalysisError(f"cannot read coordinates from {path} (override read_structure)")

    # ---- user hooks
    def register_args(self, parser: argparse.ArgumentParser) -> None:
        """Add task-specific options to `parser`."""

    def pre_process(self) -> None:
        """Called once, on the first admitted frame, before its process_frame."""

    def process_frame(self) -> None:
        """Called for every admitted frame; `self.state` / `self.src` hold it."""

    def post_process(self) -> None:
        """Called once after the last frame."""
