"""Python task driver (molar_amd/analysis_task.py) against the behaviour of MolAR's driver (molar/src/analysis_task.rs):
the five `process_suffix` test groups of :329-366 value for value, the window logic (:181-277) on synthetic streams with
no IO, and the whole driver on the reference's benzene.xtc (5 frames, times 4032..4040 ps, 12 atoms)."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
XTC = os.path.join(G, "benzene.xtc")


@pytest.fixture(scope="module")
def at():
    from molar_amd import build
    build.build_library()
    from molar_amd import analysis_task
    return analysis_task


@pytest.fixture(scope="module")
def task_cls(at):
    class Times(at.AnalysisTask):
        def register_args(self, parser):
            parser.add_argument('--tag', default='x')

        def pre_process(self):
            self.log = [("pre", self.state.time, len(self.src.state))]

        def process_frame(self):
            self.log.append(("frame", self.state.time, self.consumed_frames, self.trj_ind))

        def post_process(self):
            self.log.append(("post", self.consumed_frames))
    return Times


def times(t):
    return [e[1] for e in t.log if e[0] == "frame"]


# ---- analysis_task.rs:329-366, one test per group
def test_suffix_empty_is_no_limit(at):
    assert at.parse_bound("") == at.Bound(None, None) and at.parse_bound("   ").open


def test_suffix_bare_number_is_frame(at):
    for text, fr in (("0", 0), ("5", 5), ("42", 42), ("100", 100)):
        assert at.parse_bound(text) == at.Bound(fr, None)


def test_suffix_explicit_frame(at):
    assert at.parse_bound("5fr") == at.Bound(5, None) and at.parse_bound("100fr") == at.Bound(100, None)


def test_suffix_time_units_convert_to_ps(at):
    assert at.parse_bound("5ps") == at.Bound(None, 5.0)
    assert at.parse_bound("2ns") == at.Bound(None, 2000.0)
    assert at.parse_bound("1us") == at.Bound(None, 1_000_000.0)
    assert at.parse_bound("1.5ns") == at.Bound(None, 1500.0)


def test_suffix_invalid(at):
    with pytest.raises(at.InvalidSuffix):
        at.parse_bound("5km")
    with pytest.raises(ValueError):          # a unit with no number is a parse error
        at.parse_bound("fr")
    with pytest.raises(at.InvalidSuffix):
        at.parse_bound("-3")                 # not a usize, no unit
    with pytest.raises(ValueError):
        at.parse_bound("-3fr")               # usize


# ---- the window on synthetic streams (no files): the reference's loop, analysis_task.rs:181-277
class FakeState:
    def __init__(self, t):
        self.time = float(t)


class FakeTraj:
    def __init__(self, t):
        self.t, self.at = list(t), 0
        self.seeks = []

    def skip_to_frame(self, fr):
        self.seeks.append(("fr", fr)); self.at = fr

    def skip_to_time(self, t):
        self.seeks.append(("t", t))
        self.at = next((k for k, x in enumerate(self.t) if x >= t), len(self.t))

    def __iter__(self):
        while self.at < len(self.t):
            self.at += 1
            yield FakeState(self.t[self.at - 1])


def run_window(at, files, begin="0", end="", skip=1, add_time=False):
    w = at.FrameWindow(at.parse_bound(begin), at.parse_bound(end), skip)
    out = list(at.frame_stream(list(files), w, lambda name: files[name], add_time))
    return [(which, pos, st.time) for which, pos, st in out]


def test_window_single_file_seeks(at):
    f = {"a": FakeTraj(range(0, 100, 10))}
    assert [p for _, p, _ in run_window(at, f, "3", "6")] == [3, 4, 5] and f["a"].seeks == [("fr", 3)]
    f = {"a": FakeTraj(range(0, 100, 10))}
    assert [p for _, p, _ in run_window(at, f, "0", "2")] == [0, 1] and f["a"].seeks == []      # -b 0: no seek (:190)
    f = {"a": FakeTraj(range(0, 100, 10))}
    assert [t for _, _, t in run_window(at, f, "25ps", "60ps")] == [30, 40, 50, 60] and f["a"].seeks == [("t", 25.0)]


def test_window_skip_phase_counts_from_begin(at):
    f = {"a": FakeTraj(range(10))}
    assert [p for _, p, _ in run_window(at, f, "3", "", skip=3)] == [3, 6, 9]       # begin frame always processed (:228-234)
    f = {"a": FakeTraj(range(10)), "b": FakeTraj(range(10, 20))}
    got = run_window(at, f, "7", "16", skip=4)                                      # cadence and bounds across files
    assert [(w, p) for w, p, _ in got] == [(0, 7), (1, 11), (1, 15)]
    assert f["a"].seeks == [] and f["b"].seeks == []                                # several files: serial filtering (:203-214)


def test_window_end_is_absolute_and_stops_all_files(at):
    f = {"a": FakeTraj(range(5)), "b": FakeTraj(range(5, 10)), "c": FakeTraj(range(10, 15))}
    got = run_window(at, f, "", "7")
    assert [p for _, p, _ in got] == list(range(7)) and f["c"].at == 0               # break 'files (:221)
    f = {"a": FakeTraj(range(5)), "b": FakeTraj(range(5, 10))}
    assert [t for _, _, t in run_window(at, f, "3ps", "6ps")] == [3, 4, 5, 6]        # time end is inclusive (:219)


def test_window_add_time(at):
    f = {"a": FakeTraj([0, 10, 20]), "b": FakeTraj([0, 10, 20])}
    assert [t for _, _, t in run_window(at, f, add_time=True)] == [0, 10, 20, 20, 30, 40]
    f = {"a": FakeTraj([0, 10, 20]), "b": FakeTraj([0, 10, 20])}
    assert [t for _, _, t in run_window(at, f, end="30ps", add_time=True)] == [0, 10, 20, 20, 30]


def test_log_time_units(at):
    assert at.format_time(999.0) == "999.0 ps" and at.format_time(1500.0) == "1.5 ns" and at.format_time(2.0e6) == "2.0 us"


# ---- the whole driver on the reference's trajectory
def test_hooks_fields_and_custom_args(task_cls):
    t = task_cls(['-f', 'top.none', XTC, '--tag', 'y'])
    assert times(t) == [4032.0, 4034.0, 4036.0, 4038.0, 4040.0] and t.args.tag == 'y'
    assert t.log[0] == ("pre", 4032.0, 12) and t.log[-1] == ("post", 5)
    assert [e[2] for e in t.log if e[0] == "frame"] == [1, 2, 3, 4, 5]


def test_windows_on_xtc(task_cls):
    assert times(task_cls(['-f', 'top.none', XTC, '-b', '2'])) == [4036.0, 4038.0, 4040.0]
    assert times(task_cls(['-f', 'top.none', XTC, '-b', '2fr'])) == [4036.0, 4038.0, 4040.0]
    assert times(task_cls(['-f', 'top.none', XTC, '-b', '0', '-e', '2'])) == [4032.0, 4034.0]
    assert times(task_cls(['-f', 'top.none', XTC, '-b', '1', '-e', '3'])) == [4034.0, 4036.0]      # end is absolute
    assert times(task_cls(['-f', 'top.none', XTC, '-b', '4035ps', '-e', '4038ps'])) == [4036.0, 4038.0]
    assert times(task_cls(['-f', 'top.none', XTC, '-b', '4.035ns', '-e', '4.038ns'])) == [4036.0, 4038.0]
    assert times(task_cls(['-f', 'top.none', XTC, '--skip', '2'])) == [4032.0, 4036.0, 4040.0]
    assert times(task_cls(['-f', 'top.none', XTC, '-b', '1', '--skip', '3'])) == [4034.0, 4040.0]


def test_errors(at, task_cls):
    with pytest.raises(at.NoTrajectory):
        task_cls(['-f', XTC])
    with pytest.raises(at.NoFramesConsumed):                        # analysis_task.rs:275-277
        task_cls(['-f', 'top.none', XTC, '-b', '5'])
    with pytest.raises(at.NoFramesConsumed):
        task_cls(['-f', 'top.none', XTC, '-e', '0'])
    with pytest.raises(at.InvalidSuffix):
        task_cls(['-f', 'top.none', XTC, '-b', '5km'])
    with pytest.raises(SystemExit):                                 # clap range(1..) on --skip
        task_cls(['-f', 'top.none', XTC, '--skip', '0'])


def test_two_files_and_add_time(task_cls):
    t = task_cls(['-f', 'top.none', XTC, XTC])
    assert len(times(t)) == 10 and [e[3] for e in t.log if e[0] == "frame"] == [0] * 5 + [1] * 5
    t = task_cls(['-f', 'top.none', XTC, XTC, '--add-time'])
    assert times(t)[5:] == [4032.0 + 4040.0, 4034.0 + 4040.0, 4036.0 + 4040.0, 4038.0 + 4040.0, 4040.0 + 4040.0]
    assert len(times(task_cls(['-f', 'top.none', XTC, XTC, '-e', '7']))) == 7
    assert times(task_cls(['-f', 'top.none', XTC, XTC, '--skip', '2'])) == [4032.0, 4036.0, 4040.0, 4034.0, 4038.0]
    assert times(task_cls(['-f', 'top.none', XTC, XTC, '-b', '6'])) == [4034.0, 4036.0, 4038.0, 4040.0]   # global begin
    assert times(task_cls(['-f', 'top.none', XTC, XTC, '-b', '3', '-e', '8', '--skip', '2'])) == [4038.0, 4032.0, 4036.0]


def test_topology_from_npz(task_cls, tmp_path):
    p = tmp_path / "top.npz"
    np.savez(p, masses=np.full(12, 12.011, np.float32))
    t = task_cls(['-f', str(p), XTC, '-e', '1'])
    assert np.allclose(t.top.masses, 12.011) and len(t.src.sel()) == 12


def test_use_struct_file(at, task_cls, tmp_path):
    from molar_amd import gro, api
    r = next(iter(at._XtcSource(XTC)))
    top = gro.GroTopology(["C"] * 12, ["BNZ"] * 12, np.ones(12))
    p = tmp_path / "s.gro"
    gro.write_gro(p, top, api.State(r.coords, r.pbox, 7.0))
    t = task_cls(['-f', str(p), '--use_struct_file'])                # no trajectory needed (:139-141)
    assert times(t) == [7.0] and t.log[-1] == ("post", 1)
    t = task_cls(['-f', str(p), XTC, '--use_struct_file', '-e', '2'])
    assert times(t) == [7.0, 4032.0, 4034.0] and t.log[0][0] == "pre"
