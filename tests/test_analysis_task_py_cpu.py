"""Python AnalysisTask mirror (molar_python/python/pymolar/__init__.py:9-146) on the reference's benzene.xtc (5 frames,
times 4032..4040 ps): suffix parsing, begin/end/skip windows, add-time over two files, hook order."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
XTC = os.path.join(G, "benzene.xtc")


@pytest.fixture(scope="module")
def task_cls():
    from molar_amd import build
    build.build_library()
    from molar_amd.analysis_task import AnalysisTask

    class Times(AnalysisTask):
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


def test_process_suffix():
    from molar_amd.analysis_task import _process_suffix as ps
    assert ps('') == (None, None) and ps('12') == (12, None) and ps('100') == (100, None)
    assert ps('10ps') == (None, 10) and ps('3ns') == (None, 3000) and ps('2us') == (None, 2000000)
    assert ps('5') == (5, None)                      # '5'[-2:] == '5' is numeric


def test_windows_and_hooks(task_cls):
    t = task_cls(['-f', 'top.none', XTC, '--tag', 'y'])
    assert times(t) == [4032.0, 4034.0, 4036.0, 4038.0, 4040.0] and t.args.tag == 'y'
    assert t.log[0] == ("pre", 4032.0, 12) and t.log[-1] == ("post", 5)
    assert [e[2] for e in t.log if e[0] == "frame"] == [1, 2, 3, 4, 5]      # consumed_frames is incremented before the hook
    assert times(task_cls(['-f', 'top.none', XTC, '-b', '2'])) == [4036.0, 4038.0, 4040.0]
    assert times(task_cls(['-f', 'top.none', XTC, '-e', '2'])) == [4032.0, 4034.0]         # consumed >= 2 stops
    assert times(task_cls(['-f', 'top.none', XTC, '-b', '4035ps', '-e', '4038ps'])) == [4036.0, 4038.0]
    assert times(task_cls(['-f', 'top.none', XTC, '--skip', '2'])) == [4032.0, 4036.0, 4040.0]
    assert times(task_cls(['-f', 'top.none', XTC, '-b', '1', '--skip', '3'])) == [4034.0, 4040.0]   # phase counted from begin
    with pytest.raises(Exception):
        task_cls(['-f', XTC])


def test_two_files_and_add_time(task_cls):
    t = task_cls(['-f', 'top.none', XTC, XTC])
    assert len(times(t)) == 10 and [e[3] for e in t.log if e[0] == "frame"] == [0] * 5 + [1] * 5
    t = task_cls(['-f', 'top.none', XTC, XTC, '--add-time'])
    assert times(t)[5:] == [4032.0 + 4040.0, 4034.0 + 4040.0, 4036.0 + 4040.0, 4038.0 + 4040.0, 4040.0 + 4040.0]
    # the end frame counts consumed frames across files; skip phase runs across files too
    assert len(times(task_cls(['-f', 'top.none', XTC, XTC, '-e', '7']))) == 7
    assert times(task_cls(['-f', 'top.none', XTC, XTC, '--skip', '2'])) == [4032.0, 4036.0, 4040.0, 4034.0, 4038.0]


def test_topology_from_npz(task_cls, tmp_path):
    p = tmp_path / "top.npz"
    np.savez(p, masses=np.full(12, 12.011, np.float32))
    t = task_cls(['-f', str(p), XTC, '-e', '1'])
    assert np.allclose(t.top.masses, 12.011) and len(t.src.sel()) == 12
