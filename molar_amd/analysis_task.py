"""Python mirror of pymolar's AnalysisTask (molar_python/python/pymolar/__init__.py:9-146): the plugin surface users
write trajectory analyses against.  Same CLI (-f/--files, --log, -b/--begin, -e/--end, --skip, --add-time), same hook
names (register_args, pre_process, process_frame, post_process), same fields (args, top, state, src,
consumed_frames, trj_ind) and the same frame-window rules, including their quirks:

  * `-b`/`-e` take a frame number or a time with a ps/ns/us suffix (_process_suffix, :9-23; a one-character value is
    not numeric by the `s[-2:]` test only if it is not a digit string - kept as is);
  * the end frame is compared with the number of frames CONSUMED so far, the end time with `time + added_time`;
  * `--skip` counts valid frames from the first one of the run (not per file);
  * `--add-time` adds the last time of each finished trajectory to the following ones.

File formats are out of the engine's scope: trajectories with the extension .xtc go through the engine's XTC reader
(molar_amd.xtc.XtcReader); anything else through `open_trajectory`, and the topology through `read_topology`, both
overridable.  `src` is a small System(top, state) holder with `replace_state_deep` and selection construction."""
from __future__ import annotations

import argparse
import logging

import numpy as np

from . import api


def _process_suffix(s):
    if s == '':
        return (None, None)
    fr = None
    t = None
    if s[-2:].isnumeric():
        fr = int(s)                 # no suffix: a frame number
    elif s[-2:] == 'ps':
        t = int(s[:-2])
    elif s[-2:] == 'ns':
        t = int(s[:-2]) * 1000
    elif s[-2:] == 'us':
        t = int(s[:-2]) * 1000_000
    return (fr, t)


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

    def sel(self, index=None) -> api.Sel:
        return api.Sel(self.top, self.state, index, engine=self.engine)


class _XtcHandler:
    """FileHandler facade over XtcReader: skip_to_frame / skip_to_time / iteration (io.rs:198-271, 691-760)."""

    def __init__(self, path, engine=None):
        from .xtc import XtcReader
        self.r = XtcReader(path, engine=engine)

    def skip_to_frame(self, fr):
        self.r.seek_frame(fr)

    def skip_to_time(self, t):
        self.r.seek_time(float(t))

    def __iter__(self):
        return iter(self.r)


class AnalysisTask:
    """Subclass and implement `register_args`, `pre_process`, `process_frame`, `post_process`; constructing the object
    parses the command line (or `argv`), streams the frames and calls the hooks in processing order."""

    engine: api.Engine | None = None

    def __init__(self, argv=None):
        logging.basicConfig(format='[%(levelname)s] (%(name)s) %(message)s')
        logging.getLogger().setLevel(logging.INFO)
        logging.info(f'Executing task "{type(self).__name__}"...')

        parser = argparse.ArgumentParser('molar_amd trajectory processor')
        parser.add_argument('-f', '--files', nargs='+')
        parser.add_argument('--log', default=100, type=int)
        parser.add_argument('-b', '--begin', default='')
        parser.add_argument('-e', '--end', default='')
        parser.add_argument('--skip', default=1, type=int)
        parser.add_argument('--add-time', action="store_true")
        self.register_args(parser)
        self.args = parser.parse_args(argv)

        if not self.args.files or len(self.args.files) < 2:
            raise Exception('At least one trajectory file is required')

        self.top = None
        bfr, bt = _process_suffix(self.args.begin)
        efr, et = _process_suffix(self.args.end)

        self.consumed_frames = 0
        valid_frames = 0
        added_time = 0.0
        self.state = None

        for trj_ind, trj_file in enumerate(self.args.files[1:]):
            logging.info(f'Processing trajectory "{trj_file}"...')
            self.trj_ind = trj_ind
            trj_handler = self.open_trajectory(trj_file)
            if bfr:
                trj_handler.skip_to_frame(bfr)
            elif bt:
                trj_handler.skip_to_time(bt)
            for st in trj_handler:
                if efr and self.consumed_frames >= efr:
                    break
                if et and st.time + added_time > et:
                    break
                valid_frames += 1
                if (valid_frames - 1) % self.args.skip > 0:
                    continue
                st.time += added_time
                self.state = st
                if self.consumed_frames == 0:
                    self.top = self.read_topology(self.args.files[0], st)
                    self.src = System(self.top, self.state, self.engine)
                    self.pre_process()
                else:
                    self.src.replace_state_deep(self.state)
                if self.consumed_frames % self.args.log == 0:
                    self.__log_time()
                self.consumed_frames += 1
                self.process_frame()
            if self.args.add_time and self.state is not None:
                added_time += self.state.time

        self.post_process()

    def __log_time(self):
        if self.state.time < 1000.0:
            t = f"{self.state.time} ps"
        elif self.state.time < 1000_000.0:
            t = f"{self.state.time / 1000.0} ns"
        else:
            t = f"{self.state.time / 1000_000.0} us"
        logging.info(f'At frame {self.consumed_frames}, time {t}')

    # ---- format hooks (file formats other than XTC are outside the engine)
    def open_trajectory(self, path):
        if str(path).endswith('.xtc'):
            return _XtcHandler(path, self.engine)
        raise Exception(f'unsupported trajectory format: {path} (override open_trajectory)')

    def read_topology(self, path, first_state) -> api.Topology:
        """Default: an .npz with `masses` (and optionally `vdw`); otherwise unit masses for every atom of the first frame."""
        if str(path).endswith('.npz'):
            z = np.load(path)
            return api.Topology(z['masses'], z['vdw'] if 'vdw' in z else None)
        return api.Topology(np.ones(len(first_state), np.float32))

    # ---- user hooks
    def register_args(self, parser):
        pass

    def pre_process(self):
        pass

    def process_frame(self):
        pass

    def post_process(self):
        pass
