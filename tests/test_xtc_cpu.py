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


# ---------------------------------------------------------------- writer (molar_hip_xtc_encode_frame, xtc_handler.rs:117-168)

def _water_like(rng, nmol, L):
    """O, H, H per molecule, 0.1 nm bonds at 104.5 degrees in random orientations: the layout the format's runs of small
    deltas are made for (a lower bound on the distance between consecutive atoms is what sets the first delta size)"""
    o = rng.random((nmol, 3)) * L
    a = rng.normal(size=(nmol, 3)); a /= np.linalg.norm(a, axis=1)[:, None]
    b = rng.normal(size=(nmol, 3)); b -= (b * a).sum(1)[:, None] * a; b /= np.linalg.norm(b, axis=1)[:, None]
    th = np.deg2rad(104.5)
    return np.stack([o, o + 0.1 * a, o + 0.1 * (np.cos(th) * a + np.sin(th) * b)], 1).reshape(-1, 3).astype(np.float32)


@pytest.mark.parametrize("case", ["water", "random", "mixed", "ten_atoms", "nine_atoms", "one_atom", "huge_extent", "negative", "clustered"])
def test_writer_round_trips_through_both_decoders(reader_cls, orc32, case):
    """What the library writes, the library's decoder AND the oracle's xdrfile-style decoder read back as the same
    integers: every coordinate equals round(x * precision) / precision computed as the format defines it, frame headers
    survive, and water-like input takes fewer bits than scattered atoms (the small-delta runs are really used)."""
    from molar_amd.xtc import encode_frame
    rng = np.random.default_rng(abs(hash(case)) % 2**31)
    prec = np.float32(1000.0)
    if case == "water":
        xyz = _water_like(rng, 4000, 6.0)
    elif case == "random":
        xyz = (rng.random((12000, 3)) * 6.0).astype(np.float32)
    elif case == "mixed":
        xyz = np.concatenate([_water_like(rng, 1500, 5.0), (rng.random((3001, 3)) * 5.0).astype(np.float32), _water_like(rng, 700, 5.0)])
    elif case == "ten_atoms":
        xyz = (rng.random((10, 3)) * 2.0).astype(np.float32)
    elif case == "nine_atoms":
        xyz = (rng.random((9, 3)) * 2.0).astype(np.float32)
    elif case == "one_atom":
        xyz = np.array([[0.5, -1.25, 3.0]], np.float32)
    elif case == "huge_extent":          # sizes above 2^24 grid units: every coordinate in its own bit field
        xyz = (rng.random((500, 3)) * 4.0e4 - 2.0e4).astype(np.float32)
    elif case == "negative":
        xyz = _water_like(rng, 900, 3.0) - np.float32(7.5)
    else:                                # a tight cluster, then far jumps: the adaptive delta size moves both ways
        xyz = np.concatenate([rng.normal(0, 0.004, (3000, 3)), rng.normal(0, 0.5, (3000, 3)), rng.normal(0, 0.02, (3000, 3))]).astype(np.float32) + np.float32(4.0)
    box9 = np.array([6, 0, 0, 0.5, 6, 0, 0.25, 0.75, 6], np.float32)
    blob = encode_frame(xyz, box9, step=42, time=17.5, precision=float(prec))
    blob2 = encode_frame(xyz[::-1].copy(), box9, step=43, time=18.5, precision=float(prec))
    data = blob + blob2
    off = orc32.xtc_index(data)
    assert list(off) == [0, len(blob)]
    want_int = np.where(xyz * prec >= 0, xyz * prec + np.float32(0.5), xyz * prec - np.float32(0.5)).astype(np.int32)
    want = want_int.astype(np.float32) * (np.float32(1.0) / prec) if len(xyz) > 9 else xyz
    r = reader_cls(data)
    assert len(r) == 2 and r.natoms == len(xyz)
    got = r.read_frames(0, 2, nthreads=2)
    ref0, h0 = orc32.xtc_decode(data, 0)
    ref1, h1 = orc32.xtc_decode(data, int(off[1]))
    assert np.array_equal(got[0], ref0) and np.array_equal(got[1], ref1)
    assert np.array_equal(got[0], want) and np.array_equal(got[1], want[::-1])
    assert h0["step"] == 42 and h0["time"] == 17.5 and h1["step"] == 43 and np.array_equal(h0["box9"], box9)
    info = r.frame_info(1)
    assert info["step"] == 43 and info["time"] == 18.5 and np.array_equal(info["box9"], box9)
    r.close()


def test_writer_uses_runs_for_water(reader_cls):
    from molar_amd.xtc import encode_frame
    rng = np.random.default_rng(5)
    box9 = np.eye(3, dtype=np.float32).reshape(9) * 6
    water = len(encode_frame(_water_like(rng, 5000, 6.0), box9))
    scattered = len(encode_frame((rng.random((15000, 3)) * 6.0).astype(np.float32), box9))
    assert water < 0.8 * scattered, (water, scattered)


def test_writer_rejects_bad_input(reader_cls):
    from molar_amd.xtc import encode_frame
    from molar_amd._lib import MolarHipError
    box9 = np.zeros(9, np.float32)
    with pytest.raises(MolarHipError):
        encode_frame(np.full((20, 3), 1.0e9, np.float32), box9)            # 1e12 grid units
    with pytest.raises(MolarHipError):
        encode_frame(np.zeros((20, 3), np.float32), box9, precision=0.0)
    with pytest.raises(MolarHipError):
        encode_frame(np.full((20, 3), np.nan, np.float32), box9)


def test_xtc_writer_file_round_trip(reader_cls, tmp_path):
    from molar_amd.xtc import XtcWriter
    from molar_amd.api import PeriodicBox, State
    rng = np.random.default_rng(9)
    box = PeriodicBox.from_matrix(np.array([[5, 0.5, 0.25], [0, 5, 0.75], [0, 0, 5]], np.float32))
    frames = [_water_like(rng, 300, 5.0) for _ in range(3)]
    p = tmp_path / "w.xtc"
    with XtcWriter(p) as w:
        for k, f in enumerate(frames):
            w.write_state(State(f, box, time=2.0 * k))
    r = reader_cls(str(p))
    states = list(r)
    assert len(states) == 3 and [s.time for s in states] == [0.0, 2.0, 4.0]
    for s, f in zip(states, frames):
        assert np.abs(s.coords - f).max() <= 0.5001e-3
        assert np.array_equal(s.pbox.get_matrix(), box.get_matrix())
