"""XTC frames: (1) the oracle's codec pinned by the reference's own asserting test (tests/test_netcdf.rs:37-80:
benzene.xtc decodes to benzene.nc within 1e-3 nm, times within 0.01) on copies of those two data files;
(2) the product's host-thread decoder (libmolar_hip.so, no GPU needed for host output) BIT-IDENTICAL to the oracle
on the reference's files and on large synthetic streams; (3) the handler semantics (seek_frame, seek_time, Eof)."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REF = "/root/reference"


@pytest.fixture(scope="module")
def reader_cls():
    from molar_amd import build
    build.build_library()
    from molar_amd.xtc import XtcReader
    return XtcReader


def nc_frames():
    from scipy.io import netcdf_file
    nc = netcdf_file(os.path.join(G, "benzene.nc"), "r", mmap=False)
    return nc.variables["coordinates"][:] * 0.1, nc.variables["time"][:].copy()      # Angstrom -> nm


def test_oracle_benzene_matches_netcdf(orc32):
    data = open(os.path.join(G, "benzene.xtc"), "rb").read()
    coords, times = nc_frames()
    off = orc32.xtc_index(data)
    assert len(off) == len(coords) == 5
    for k, o in enumerate(off):
        xyz, h = orc32.xtc_decode(data, o)
        assert xyz.shape == coords[k].shape
        assert abs(h["time"] - times[k]) < 0.01
        assert np.linalg.norm(xyz - coords[k], axis=1).max() < 1e-3


def test_product_benzene_matches_netcdf_and_oracle(reader_cls, orc32):
    path = os.path.join(G, "benzene.xtc")
    data = open(path, "rb").read()
    coords, times = nc_frames()
    for src in (path, data):
        r = reader_cls(src)
        assert len(r) == 5 and r.natoms == 12
        got = r.read_frames(0, 5)
        for k, o in enumerate(orc32.xtc_index(data)):
            want, h = orc32.xtc_decode(data, o)
            assert np.array_equal(got[k], want)
            info = r.frame_info(k)
            assert info["step"] == h["step"] and info["time"] == h["time"] and np.array_equal(info["box9"], h["box9"])
            assert np.linalg.norm(got[k] - coords[k], axis=1).max() < 1e-3 and abs(info["time"] - times[k]) < 0.01
        r.close()


def test_handler_semantics(reader_cls):
    r = reader_cls(os.path.join(G, "benzene.xtc"))
    states = list(r)
    assert len(states) == 5 and [s.time for s in states] == [4032.0, 4034.0, 4036.0, 4038.0, 4040.0]
    with pytest.raises(EOFError):
        r.read_state()
    r.seek_frame(3)
    assert r.read_state().time == 4038.0
    r.seek_time(4035.0)                 # first frame with time >= t (xtc_handler.rs:282-297)
    assert r.read_state().time == 4036.0
    r.seek_time(4032.0)
    assert r.read_state().time == 4032.0
    from molar_amd._lib import MolarHipError
    with pytest.raises(MolarHipError):
        r.seek_time(5000.0)
    assert states[0].pbox is not None and states[0].coords.shape == (12, 3)
    with pytest.raises(MolarHipError):
        reader_cls("/nonexistent/file.xtc")


def synthetic_frames(n, nframes, seed=20240607):
    """Water-like triplets (small intra-molecular deltas exercise the run path) in a triclinic box."""
    rng = np.random.default_rng(seed)
    nmol = n // 3
    box = np.array([[6.0, 0.0, 0.0], [1.0, 6.0, 0.0], [-1.0, 0.5, 6.0]], np.float32)     # rows = a, b, c as in the file
    centres = rng.random((nmol, 3)) @ box
    out = []
    for f in range(nframes):
        c = centres + 0.02 * rng.normal(size=centres.shape)
        mol = c[:, None, :] + 0.08 * rng.normal(size=(nmol, 3, 3))
        xyz = mol.reshape(-1, 3)
        if n % 3:
            xyz = np.concatenate([xyz, rng.random((n % 3, 3)) * 6])
        out.append(xyz.astype(np.float32))
    return out, box.reshape(9)


@pytest.mark.parametrize("natoms,magic,precision", [(5, 1995, 1000.0), (9, 1995, 1000.0), (10, 1995, 1000.0), (3000, 1995, 1000.0),
                                                    (3001, 2023, 1000.0), (20000, 1995, 100.0), (1000, 1995, 100000.0)])
def test_encode_decode_roundtrip_oracle_and_product(reader_cls, orc32, natoms, magic, precision):
    frames, box9 = synthetic_frames(natoms, 3)
    if precision > 1e4:                       # force the >24-bit (separately coded) integer path
        frames = [f * 300.0 for f in frames]
    blob = b"".join(orc32.xtc_encode(f, box9, step=10 * k, time=2.0 * k, precision=precision, magic=magic) for k, f in enumerate(frames))
    off = orc32.xtc_index(blob)
    assert len(off) == 3
    r = reader_cls(blob, nthreads=2)
    assert len(r) == 3 and r.natoms == natoms
    got = r.read_frames(0, 3)
    for k, o in enumerate(off):
        want, h = orc32.xtc_decode(blob, o)
        assert h["step"] == 10 * k and h["time"] == 2.0 * k
        assert np.array_equal(got[k], want)                           # product == oracle, bit for bit
        tol = 0.0 if natoms <= 9 else 0.5 / precision * 1.01 + 1e-6 * np.abs(frames[k]).max()
        assert np.abs(want - frames[k]).max() <= tol                  # lossy by at most half a grid step
    # a truncated last frame is not indexed (Eof semantics)
    assert len(reader_cls(blob[:-8])) == 2
    assert len(orc32.xtc_index(blob[:-8])) == 2


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference data files not mounted")
@pytest.mark.parametrize("rel,nframes,natoms", [("molar/tests/new.xtc", 10, 4295), ("molar_membrane/tests/traj_comp.xtc", 11, 87349)])
def test_reference_trajectories_product_equals_oracle(reader_cls, orc32, rel, nframes, natoms):
    """The GROMACS-written files of the reference's test suite (adaptive small index, all run lengths): every
    block is consumed exactly by both decoders and the outputs are bit-identical."""
    path = os.path.join(REF, rel)
    data = open(path, "rb").read()
    off = orc32.xtc_index(data)
    r = reader_cls(path, nthreads=4)
    assert len(off) == len(r) == nframes and r.natoms == natoms
    got = r.read_frames(0, nframes)
    total = 0
    for k, o in enumerate(off):
        want, h = orc32.xtc_decode(data, o)
        total += h["length"]
        assert np.array_equal(got[k], want)
        M = h["box9"].reshape(3, 3)            # rows a, b, c
        frac = want @ np.linalg.inv(M.astype(np.float64))
        assert frac.min() > -0.6 and frac.max() < 1.6          # molecules whole or wrapped: near the unit cell
    assert total == len(data)
    # consecutive frames of an MD trajectory are close
    d = np.abs(got[1:] - got[:-1])
    assert np.median(d) < 0.3


def test_corrupt_streams_do_not_crash(reader_cls, orc32):
    """Bit flips, truncations and hostile header fields: the decoder reports an error or returns numbers, it never
    reads outside the buffer (the reader bounds every fetch by the block length of the index)."""
    from molar_amd._lib import MolarHipError
    frames, box9 = synthetic_frames(600, 2)
    good = b"".join(orc32.xtc_encode(f, box9, step=k, time=float(k)) for k, f in enumerate(frames))
    rng = np.random.default_rng(7)
    outcomes = {"ok": 0, "error": 0, "short": 0}
    for trial in range(300):
        b = bytearray(good)
        kind = trial % 3
        if kind == 0:                                   # random bit flips in the compressed block
            for _ in range(int(rng.integers(1, 6))):
                pos = int(rng.integers(92, len(b)))
                b[pos] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:                                 # hostile header words (natoms, sizes, smallidx, byte count)
            pos = int(rng.choice([4, 52, 60, 64, 68, 72, 76, 80, 84, 88]))
            b[pos:pos + 4] = rng.integers(0, 256, 4, dtype=np.uint8).tobytes()
        else:                                           # truncation
            b = b[: int(rng.integers(1, len(b)))]
        r = reader_cls(bytes(b), nthreads=1)
        n = len(r)
        if n < 2:
            outcomes["short"] += 1
        try:
            if n:
                r.read_frames(0, n)
            outcomes["ok"] += 1
        except MolarHipError:
            outcomes["error"] += 1
        r.close()
    assert outcomes["error"] > 0 and outcomes["ok"] > 0 and outcomes["short"] > 0
    # a coordinate range of zero size (maxint = minint - 1) would make a radix of the mixed-radix triples zero
    for k in range(3):
        b = bytearray(good)
        lo = int.from_bytes(b[60 + 4 * k: 64 + 4 * k], "big", signed=True)
        b[72 + 4 * k: 76 + 4 * k] = (lo - 1).to_bytes(4, "big", signed=True)
        r = reader_cls(bytes(b), nthreads=1)
        with pytest.raises(MolarHipError):
            r.read_frames(0, 1)
        r.close()
    # extreme ranges: the integer arithmetic of the decoder wraps, it does not overflow (run under UBSan by tools/asan_host.sh)
    b = bytearray(good)
    b[60:64] = (2**31 - 5).to_bytes(4, "big", signed=True)
    b[72:76] = (2**31 - 1).to_bytes(4, "big", signed=True)
    r = reader_cls(bytes(b), nthreads=1)
    try:
        r.read_frames(0, 1)
    except MolarHipError:
        pass
    r.close()


def test_xtc_randomised_differential():
    """A 300-case slice of tools/fuzz_xtc.py: random atom counts (incl. the uncompressed <= 9 atom form), precisions,
    magic numbers, water-like triplets, lattices, chains, kilometre-sized coordinates; product decoder == oracle codec."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_xtc
    assert fuzz_xtc.run(300, 3) == 0
