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


@pytest.mark.timeout(900)
def test_c4_shape_xtc_window_to_fused_histogram(eng, orc32):
    """BASELINE config 4 at its frame size: a window of 250k-atom XTC frames (box A, 100 atoms/nm^3) is decoded by the
    engine's host threads straight into HBM and every frame goes through the fused search + Histogram1D binning
    (1200 bins of 0.001 nm, bins resident on the GPU, no per-frame round trip).  The summed integer bins must equal
    Histogram1D::add_one (stats.rs:29-35) applied to the oracle's distance stream of the oracle-decoded frames."""
    import torch
    from molar_amd import api, synth
    from molar_amd.xtc import XtcReader
    n, nframes, rc, nbins = 250_000, 3, 1.2, 1200
    box = synth.box_a(n)                                      # columns a, b, c
    box9 = np.ascontiguousarray(box.T).reshape(9)             # the file stores rows a, b, c
    blob = b"".join(orc32.xtc_encode(synth.frame(n, box, f), box9, step=f, time=float(f)) for f in range(nframes))
    r = XtcReader(blob, engine=eng, nthreads=8)
    assert len(r) == nframes and r.natoms == n
    dev = torch.empty((nframes, n, 3), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    r.read_frames(0, nframes, out=dev)
    bins = torch.zeros(nbins, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    for f in range(nframes):
        eng.search_histogram(api.SEARCH_SINGLE, rc, 0.0, rc, nbins, dev[f], box=box, pbc=7, bins=bins, want_count=False)
    eng.synchronize()
    got = bins.cpu().numpy().astype(np.uint64)
    want = np.zeros(nbins, np.uint64)
    ob = orc32.box_from_matrix(box)
    ncpu = os.cpu_count() or 8
    for f, off in enumerate(orc32.xtc_index(blob)):
        xyz = orc32.xtc_decode(blob, off)[0]
        assert np.array_equal(dev[f].cpu().numpy(), xyz)     # same bits as the oracle's decoder
        ref = orc32.search_single_pbc(rc, xyz, ob, 7, nthreads=ncpu)
        want += orc32.histogram_add(0.0, rc, nbins, ref["d"]).astype(np.uint64)
        del ref
    assert want.sum() > 2.5e8 and np.array_equal(got, want)


@pytest.mark.timeout(900)
def test_c5_shape_xtc_window_to_chained_membrane_frames(eng, orc32):
    """BASELINE.json configs[4] fed the way a trajectory arrives: a window of XTC frames of the 500k-atom bilayer decoded
    by host threads into HBM, every frame through the chained membrane call with two in flight.  The decode is compared
    with the CPU checker's, the analysis with the stage-by-stage path on the same decoded coordinates."""
    import torch
    from molar_amd import membrane as mb
    from molar_amd.xtc import XtcReader
    xyz, box, first, tpl, masses = mb.build_bilayer(2000, 500_000)
    n, nfr = len(xyz), 3
    rng = np.random.default_rng(11)
    frames = [(xyz + rng.normal(0, 0.02, xyz.shape)).astype(np.float32) for _ in range(nfr)]
    box9 = np.ascontiguousarray(box.T.reshape(-1), np.float32)             # column-major, columns a, b, c
    blob = b"".join(orc32.xtc_encode(f, box9, step=k, time=float(k)) for k, f in enumerate(frames))
    r = XtcReader(blob, engine=eng, nthreads=8)
    dev = torch.empty((nfr, n, 3), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    r.read_frames(0, nfr, out=dev)
    host = dev.cpu().numpy()
    offs = orc32.xtc_index(blob)
    assert np.array_equal(host[1], orc32.xtc_decode(blob, offs[1])[0])     # (XTC rounds to 0.001 nm: not the input bits)
    fused = mb.Membrane(eng, n, first, tpl, masses, mb.MembraneOptions(cutoff=2.5, order_type=1))
    staged = mb.Membrane(eng, n, first, tpl, masses, mb.MembraneOptions(cutoff=2.5, order_type=1, fused=False))
    got = []
    prev = fused.compute_begin(dev[0], box)
    for k in range(1, nfr):
        t = fused.compute_begin(dev[k], box)
        got.append(fused.compute_end(prev))
        prev = t
    got.append(fused.compute_end(prev))
    for k in range(nfr):
        want = staged.compute(host[k].copy(), box)
        for key in ("patch_off", "patch_ids", "valid", "normals", "mean_curv", "gauss_curv", "area", "nvert", "neib_ids", "smoothed_head"):
            assert np.ascontiguousarray(got[k][key]).tobytes() == np.ascontiguousarray(want[key]).tobytes(), (k, key)
        for a, b in zip(got[k]["order"], want["order"]):
            assert a.tobytes() == b.tobytes()
        assert np.count_nonzero(got[k]["valid"]) > 3900


@pytest.mark.parametrize("natoms,magic,precision,nframes", [(5, 1995, 1000.0, 3), (9, 1995, 1000.0, 70), (10, 1995, 1000.0, 5), (3000, 1995, 1000.0, 130),
                                                            (3001, 2023, 1000.0, 7), (20000, 1995, 100.0, 64), (1000, 1995, 100000.0, 9),
                                                            (1000, 1995, 3.0e6, 4)])
def test_device_decoder_gives_the_bits_of_the_host_decoder(eng, orc32, natoms, magic, precision, nframes):
    """molar_hip_xtc_read_device (one lane per frame) against molar_hip_xtc_read and the oracle's codec: small frames stored as
    plain floats, the adaptive small-delta index, both magics, the separately coded > 24-bit integers, and (last case)
    triples wider than 64 bits, which the call hands to the host decoder frame by frame."""
    import torch
    from molar_amd.xtc import XtcReader
    from test_xtc_cpu import synthetic_frames
    base, box9 = synthetic_frames(natoms, min(nframes, 6))
    if precision > 1e4:
        base = [f * 300.0 for f in base]
    blob = b"".join(orc32.xtc_encode(base[k % len(base)] + np.float32(0.001 * k), box9, step=k, time=float(k), precision=precision, magic=magic)
                    for k in range(nframes))
    r = XtcReader(blob, engine=eng, nthreads=4)
    assert len(r) == nframes
    host = r.read_frames(0, nframes)
    dev = torch.full((nframes, natoms, 3), float("nan"), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    r.read_frames_device(0, nframes, dev)
    assert np.array_equal(dev.cpu().numpy(), host)
    off = orc32.xtc_index(blob)
    for k in (0, nframes - 1):
        assert np.array_equal(host[k], orc32.xtc_decode(blob, off[k])[0])
    # a window in the middle of the file
    if nframes > 4:
        dev2 = torch.empty((nframes - 3, natoms, 3), dtype=torch.float32, device="cuda")
        r.read_frames_device(2, nframes - 3, dev2)
        assert np.array_equal(dev2.cpu().numpy(), host[2:nframes - 1])


def test_device_decoder_reports_corrupt_streams(eng, orc32):
    import torch
    from molar_amd._lib import MolarHipError
    from molar_amd.xtc import XtcReader
    from test_xtc_cpu import synthetic_frames
    frames, box9 = synthetic_frames(600, 3)
    blob = bytearray(b"".join(orc32.xtc_encode(f, box9, step=k, time=float(k)) for k, f in enumerate(frames)))
    off = orc32.xtc_index(bytes(blob))
    blob[off[1] + 84 + 3] = 200                       # smallidx of frame 1 out of range
    r = XtcReader(bytes(blob), engine=eng)
    dev = torch.empty((3, 600, 3), dtype=torch.float32, device="cuda")
    with pytest.raises(MolarHipError):
        r.read_frames_device(0, 3, dev)
    r.read_frames_device(0, 1, dev[:1])               # the intact frame in front still decodes


def test_xtc_histogram_one_call(eng, orc32):
    """molar_hip_xtc_histogram: BASELINE config 4's whole path as one call for a caller without device memory - frames of an XTC
    block decoded on host threads into 16-frame windows in HBM (helper thread, auxiliary context) while the window before is in
    the frames form of the fused histogram, every frame's own box.  Integer bins: they must equal Histogram1D::add_one
    (stats.rs:29-35) over the oracle's distance stream of the oracle-decoded frames - all atoms and a selection, a block that is
    not a multiple of the window, frames whose boxes differ, a sub-range of the file; repeated calls add up."""
    from molar_amd import synth
    from molar_amd.xtc import XtcReader
    n, nframes, rc, nbins = 30_000, 37, 0.9, 500
    box = synth.box_a(n)
    blobs, boxes = [], []
    for f in range(nframes):
        b = (box * np.float32(1.0 + 0.002 * np.sin(f))).astype(np.float32)
        boxes.append(b)
        blobs.append(orc32.xtc_encode(synth.frame(n, b, f), np.ascontiguousarray(b.T).reshape(9), step=f, time=float(f)))
    blob = b"".join(blobs)
    r = XtcReader(blob, engine=eng, nthreads=4)
    assert len(r) == nframes
    idx = np.sort(np.random.default_rng(3).choice(n, n // 3, replace=False)).astype(np.uint64)
    offs = orc32.xtc_index(blob)

    def want(first, count, sel):
        w = np.zeros(nbins, np.uint64)
        for f in range(first, first + count):
            xyz = orc32.xtc_decode(blob, offs[f])[0]
            p = xyz if sel is None else xyz[sel.astype(int)]
            # the box the FILE holds (f32 through the encoder's header), as the engine reads it from the frame's header
            fb = r.frame_info(f)["box9"].reshape(3, 3).T
            ref = orc32.search_single_pbc(rc, p, orc32.box_from_matrix(fb), 7, nthreads=8)
            w += orc32.histogram_add(0.0, rc, nbins, ref["d"]).astype(np.uint64)
        return w
    w_all = want(0, nframes, None)
    got = r.histogram(0, nframes, rc, 0.0, rc, nbins)
    assert w_all.sum() > 1e8 and np.array_equal(got, w_all)
    r.histogram(0, nframes, rc, 0.0, rc, nbins, bins=got)              # added into
    assert np.array_equal(got, 2 * w_all)
    assert np.array_equal(r.histogram(5, 18, rc, 0.0, rc, nbins, idx=idx), want(5, 18, idx))
    assert np.array_equal(r.histogram(36, 1, rc, 0.0, rc, nbins), want(36, 1, None))
    # between two selections of every frame (molar_hip_xtc_histogram_double): overlapping ones, and one against all atoms
    idx2 = np.sort(np.random.default_rng(4).choice(n, n // 10, replace=False)).astype(np.uint64)

    def want2(first, count, s1, s2):
        w = np.zeros(nbins, np.uint64)
        for f in range(first, first + count):
            xyz = orc32.xtc_decode(blob, offs[f])[0]
            fb = r.frame_info(f)["box9"].reshape(3, 3).T
            ref = orc32.search_double_pbc(rc, xyz[s1.astype(int)], xyz if s2 is None else xyz[s2.astype(int)], orc32.box_from_matrix(fb), 7, nthreads=8)
            w += orc32.histogram_add(0.0, rc, nbins, ref["d"]).astype(np.uint64)
        return w
    assert np.intersect1d(idx, idx2).size > 0
    assert np.array_equal(r.histogram(2, 21, rc, 0.0, rc, nbins, idx=idx, idx2=idx2), want2(2, 21, idx, idx2))
    assert np.array_equal(r.histogram(30, 7, rc, 0.0, rc, nbins, idx=idx2, two_sets=True), want2(30, 7, idx2, None))
    from molar_amd._lib import MolarHipError
    with pytest.raises(MolarHipError):
        r.histogram(30, 10, rc, 0.0, rc, nbins)
