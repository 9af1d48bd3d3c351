"""XTC frames decoded by host threads straight into device memory feed the neighbour search: same bits as the
host-side decode, and the pair list of a decoded frame equals the one of the same coordinates handed in directly."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def eng():
    from molar_amd import build
    from molar_amd.api import Engine
    build.build_library()
    return Engine(0)


def test_decode_to_device_and_search(eng, orc32):
    import torch
    from molar_amd import api
    from molar_amd.xtc import XtcReader
    from test_xtc_cpu import synthetic_frames
    n = 30000
    frames, box9 = synthetic_frames(n, 6)
    blob = b"".join(orc32.xtc_encode(f, box9, step=k, time=float(k)) for k, f in enumerate(frames))
    r = XtcReader(blob, engine=eng, nthreads=4)
    host = r.read_frames(0, 6)
    dev = torch.empty((6, n, 3), dtype=torch.float32, device="cuda")
    r.read_frames(0, 6, out=dev)
    assert np.array_equal(dev.cpu().numpy(), host)
    for k, o in enumerate(orc32.xtc_index(blob)):
        assert np.array_equal(host[k], orc32.xtc_decode(blob, o)[0])
    # a frame taken from the device buffer goes through the search exactly like host coordinates
    box = box9.reshape(3, 3).T                       # columns a, b, c
    ref = orc32.search_single_pbc(0.6, host[2], orc32.box_from_matrix(box), 7, nthreads=4)
    cnt = eng.search_count(api.SEARCH_SINGLE, 0.6, dev[2], box=box, pbc=7)
    pairs, d = eng.search_fill(cnt)
    assert cnt == len(ref["i"]) > 0
    assert np.array_equal(pairs[:, 0], ref["i"]) and np.array_equal(pairs[:, 1], ref["j"]) and np.array_equal(d, ref["d"])


def test_reference_benzene_states_on_device(eng):
    import torch
    from molar_amd.xtc import XtcReader
    r = XtcReader(os.path.join(G, "benzene.xtc"), engine=eng)
    dev = torch.zeros((5, 12, 3), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()             # the engine copies on ITS stream: torch's fill kernel must be done first
    r.read_frames(0, 5, out=dev)
    host = r.read_frames(0, 5)
    assert np.array_equal(dev.cpu().numpy(), host)
    st = list(r)
    assert len(st) == 5 and np.array_equal(st[4].coords, host[4])
    # centre of geometry of a decoded frame through the engine
    assert np.allclose(eng.center_of_geometry(dev[0]), host[0].astype(np.float64).mean(0), atol=1e-5)
