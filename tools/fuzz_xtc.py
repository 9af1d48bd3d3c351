#!/usr/bin/env python
"""Randomised differential test of the XTC decoder (molar_amd/csrc/xtc.hip, host threads) against the oracle's
xdrfile-style codec: frames of 1..6000 atoms encoded by the oracle with random precisions (10..1e5), both magic numbers,
coordinate ranges from a few nm to kilometres (the >24-bit separately coded path), water-like triplets (runs, the
small-index adaptation and the first-pair swap), lattices (zero deltas) and uniform noise.  Decoded coordinates must be
bit-identical.  Runs on the CPU.  Usage: python tools/fuzz_xtc.py [ncases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(ncases=200, seed=1):
    from molar_amd import build
    from molar_amd.xtc import XtcReader
    from oracle.oracle import Oracle
    build.build_library()
    o = Oracle("f32")
    rng = np.random.default_rng(seed)
    fails = 0
    for case in range(ncases):
        natoms = int(rng.choice([1, 2, 5, 9, 10, 11, int(rng.integers(12, 6000))]))
        kind = case % 5
        span = float(10.0 ** rng.uniform(0, 2.5)) if kind != 4 else float(10.0 ** rng.uniform(4, 6))
        if kind == 0:        # water-like: triplets within 0.1 nm
            c = rng.uniform(0, span, ((natoms + 2) // 3, 3))
            xyz = (np.repeat(c, 3, axis=0)[:natoms] + rng.normal(0, 0.06, (natoms, 3)))
        elif kind == 1:      # lattice with repeated positions
            xyz = np.round(rng.uniform(0, span, (natoms, 3)) * 2) / 2
        elif kind == 2:      # chain: small steps (long runs of small deltas)
            xyz = np.cumsum(rng.normal(0, 0.05, (natoms, 3)), axis=0) + span / 2
        else:                # uniform / huge coordinates
            xyz = rng.uniform(-span, span, (natoms, 3))
        xyz = xyz.astype(np.float32)
        precision = float(rng.choice([10.0, 100.0, 1000.0, 1000.0, 10000.0, 100000.0]))
        if kind == 4:
            precision = float(rng.choice([10.0, 100.0, 1000.0]))
        magic = int(rng.choice([1995, 2023]))
        box9 = np.diag(rng.uniform(1, 50, 3)).astype(np.float32).reshape(9)
        nfr = int(rng.integers(1, 4))
        try:
            blob = b"".join(o.xtc_encode(xyz + np.float32(0.01 * k), box9, step=k, time=float(k), precision=precision, magic=magic) for k in range(nfr))
        except Exception as e:      # the encoder refuses what xdrfile refuses (integer overflow of the scaled coordinates)
            continue
        off = o.xtc_index(blob)
        r = XtcReader(blob, nthreads=int(rng.integers(1, 4)))
        ok = len(r) == len(off) == nfr and r.natoms == natoms
        if ok:
            got = r.read_frames(0, nfr)
            for k, q in enumerate(off):
                want, h = o.xtc_decode(blob, q)
                ok = ok and np.array_equal(got[k], want) and r.frame_info(k)["step"] == h["step"]
        if not ok:
            fails += 1
            print("MISMATCH", case, kind, natoms, precision, magic, span)
    print(f"{ncases} cases, {fails} failures")
    return fails


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
