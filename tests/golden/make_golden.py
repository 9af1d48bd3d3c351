#!/usr/bin/env python
"""Generates the committed golden fixtures of tests/golden/.

PROVENANCE — read before trusting these files.  MolAR is Rust and cannot be built or imported in this
environment (no cargo/rustc, no network), so NONE of these vectors comes from running the reference:
  * periodic_box_known_answers.json  — inputs and expected values TRANSCRIBED (as data) from the reference's own
                                       asserting tests (molar/src/periodic_box.rs:456-620,
                                       molar_python/tests/test_2.py:233-245).  These pin the oracle.
  * search_*.npz, measure.npz, membrane.npz — ORACLE-GENERATED (oracle/molar_oracle.c, f32 build for integer /
                                       ordered results, f64 build for float results).  They pin the oracle and
                                       the HIP path against regressions and against each other; by themselves
                                       they do not prove agreement with MolAR ("parity unpinned").
  * ordered_pair_digests.json        — SHA-256 of the oracle's ordered pair lists on larger seeded frames, including
                                       BASELINE config 2 at full size (1M atoms, rc 1.2: `python make_golden.py c2`).
Run from the repository root:  python tests/golden/make_golden.py          (regenerates the .npz / .json fixtures)
                               python tests/golden/make_golden.py bin      (re-exports them as raw .bin + manifest for
                                                                            rust/molar_hip/tests/parity.rs)
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from molar_amd import synth  # noqa: E402  (input generator only: numpy, no GPU)
from oracle.oracle import Oracle  # noqa: E402


def boxes(n):
    rd = np.array([[1.0, 0.0, 0.5], [0.0, 1.0, 0.5], [0.0, 0.0, np.sqrt(0.5)]])       # GROMACS rhombic dodecahedron (xy-square)
    vol = n / 100.0
    rd = (rd * (vol / abs(np.linalg.det(rd))) ** (1 / 3)).astype(np.float32)
    return {"ortho": synth.box_ortho(n), "tric_a": synth.box_a(n), "hex_b": synth.box_b(n), "rhombic_dodecahedron": rd}


def pairs_dict(prefix, r, out, within=False):
    out[prefix + "_i"] = r["i"].astype(np.uint32)
    if not within:
        out[prefix + "_j"] = r["j"].astype(np.uint32)
        out[prefix + "_d"] = r["d"].astype(np.float32)
    out[prefix + "_dims"] = np.array(r["dims"], np.uint32)


def make_search(o32):
    n, rc = 600, 0.5
    for name, box in boxes(n).items():
        pos = synth.frame(n, box, 0, sigma=0.08)
        ob = o32.box_from_matrix(box)
        i1 = np.arange(0, n, 2, dtype=np.uint64); i2 = np.arange(1, n, 2, dtype=np.uint64)
        p1, p2 = pos[0::2], pos[1::2]
        vdw = (0.12 + 0.1 * np.random.default_rng(7).random(n)).astype(np.float32)
        out = dict(box=box, pos=pos, cutoff=np.float32(rc), idx1=i1, idx2=i2, vdw=vdw)
        pairs_dict("single_pbc7", o32.search_single_pbc(rc, pos, ob, 7), out)
        pairs_dict("single_pbc3", o32.search_single_pbc(rc, pos, ob, 3), out)
        pairs_dict("single", o32.search_single(rc, pos), out)
        pairs_dict("double_pbc7", o32.search_double_pbc(rc, p1, p2, ob, 7, ids1=i1, ids2=i2), out)
        pairs_dict("double", o32.search_double(rc, p1, p2, ids1=i1, ids2=i2), out)
        pairs_dict("vdw_pbc7", o32.search_double_vdw_pbc(p1, p2, vdw[0::2], vdw[1::2], ob, 7), out)
        pairs_dict("vdw", o32.search_double_vdw(p1, p2, vdw[0::2], vdw[1::2]), out)
        pairs_dict("within_pbc7", o32.search_within_pbc(rc, p1, p2, ob, 7, ids1=i1, ids2=i2), out, within=True)
        lo = p1.min(0) - np.float32(rc + 1.1920929e-07); hi = p1.max(0) + np.float32(rc + 1.1920929e-07)
        out["within_lower"] = lo.astype(np.float32); out["within_upper"] = hi.astype(np.float32)
        pairs_dict("within", o32.search_within(rc, p1, p2, out["within_lower"], out["within_upper"], ids1=i1, ids2=i2), out, within=True)
        np.savez_compressed(os.path.join(HERE, f"search_{name}.npz"), **out)
        print(name, {k: len(v) for k, v in out.items() if k.endswith("_i")})


def make_measure(o32, o64):
    n = 900
    box = synth.box_a(n)
    pos = synth.frame(n, box, 0)
    ref = synth.frame(n, box, 1)
    mass = synth.masses(n)
    idx = np.arange(0, n, 3, dtype=np.uint64)
    ob = o64.box_from_matrix(box)
    out = dict(box=box, pos=pos, ref=ref, mass=mass, idx=idx)
    lo, hi = o64.min_max(pos, idx)
    out["min"], out["max"] = lo, hi
    out["cog"] = o64.center_of_geometry(pos, idx)
    out["com"] = o64.center_of_mass(pos, mass, idx)
    out["cog_pbc7"] = o64.center_of_geometry_pbc_dims(pos, ob, 7, idx)
    out["com_pbc7"] = o64.center_of_mass_pbc_dims(pos, mass, ob, 7, idx)
    out["com_pbc5"] = o64.center_of_mass_pbc_dims(pos, mass, ob, 5, idx)
    out["gyration"] = np.float64(o64.gyration(pos, mass, idx))
    out["gyration_pbc"] = np.float64(o64.gyration_pbc(pos, mass, ob, idx))
    mom, axes = o64.inertia(pos, mass, idx)
    out["inertia_moments"], out["inertia_axes"] = mom, axes
    out["rmsd"] = np.float64(o64.rmsd(pos, ref, idx, idx))
    out["rmsd_mw"] = np.float64(o64.rmsd_mw(pos, mass, ref, idx, idx))
    R, t = o64.fit_transform(pos, mass, ref, mass, idx, idx)
    out["fit_R"], out["fit_t"] = R, t
    moved32 = o32.apply_transform(pos, R.astype(np.float32), t.astype(np.float32), idx)
    out["applied_f32"] = moved32                                   # bit-level expectation for apply_transform
    out["rmsd_after_fit"] = np.float64(o64.rmsd(o64.apply_transform(pos, R, t, idx), ref, idx, idx))
    unw = o32.unwrap_simple_dim(pos, o32.box_from_matrix(box), 7, idx)
    out["unwrapped_f32"] = unw
    # a lipid tail: zig-zag chain of 16 carbons, one double bond
    rng = np.random.default_rng(3)
    tail = np.cumsum(np.concatenate([[[5.0, 5.0, 5.0]], np.stack([0.05 * rng.normal(size=15) + 0.04 * (-1) ** np.arange(15),
                                                                     0.05 * rng.normal(size=15), 0.12 + 0.01 * rng.normal(size=15)], 1)]), 0).astype(np.float32)
    bo = np.ones(15, np.uint8); bo[7] = 2
    out["tail"], out["tail_bonds"] = tail, bo
    nrm = np.array([[0.1, -0.05, 1.0]]); nrm = (nrm / np.linalg.norm(nrm)).astype(np.float32)
    out["tail_normal"] = nrm
    for ot, nm in ((0, "sz"), (1, "scd"), (2, "scd_corr")):
        out["order_" + nm] = o64.lipid_tail_order(tail, ot, nrm, bo)
    np.savez_compressed(os.path.join(HERE, "measure.npz"), **out)


def make_membrane(o32):
    rng = np.random.default_rng(5)
    side = 18
    L = side * 0.8
    g = (np.stack(np.meshgrid(np.arange(side), np.arange(side), indexing="ij"), -1).reshape(-1, 2) + 0.5
         + 0.2 * rng.normal(size=(side * side, 2))) * L / side
    z = 5.0 + 0.3 * np.sin(2 * np.pi * g[:, 0] / L) * np.cos(2 * np.pi * g[:, 1] / L) + 0.02 * rng.normal(size=len(g))
    head = np.concatenate([g, z[:, None]], 1).astype(np.float32)
    box = np.diag([L, L, 12.0]).astype(np.float32)
    K = len(head)
    nrm = np.tile(np.array([0, 0, 1], np.float32), (K, 1)) + 0.05 * rng.normal(size=(K, 3)).astype(np.float32)
    nrm = (nrm / np.linalg.norm(nrm, axis=1, keepdims=True)).astype(np.float32)
    ob = o32.box_from_matrix(box)
    r = o32.search_single_pbc(2.0, head, ob, 7)
    i = r["i"].astype(np.int64); j = r["j"].astype(np.int64)
    src = np.stack([i, j], 1).reshape(-1); dst = np.stack([j, i], 1).reshape(-1)
    order = np.argsort(src, kind="stable")
    poff = np.concatenate([[0], np.cumsum(np.bincount(src, minlength=K))]).astype(np.uint64)
    pids = dst[order].astype(np.uint64)
    valid = np.ones(K, np.uint8); valid[::29] = 0
    s = o32.membrane_smooth(ob, head, nrm, valid, poff, pids)
    out = dict(box=box, head=head, normals=nrm, valid=valid, patch_off=poff, patch_ids=pids)
    for k, v in s.items():
        out["out_" + k] = v
    np.savez_compressed(os.path.join(HERE, "membrane.npz"), **out)


def make_digests(o32):
    dig = {}
    for name, boxfn, n, rc in (("tric_a_20000_rc0.8", synth.box_a, 20000, 0.8), ("hex_b_12000_rc0.6", synth.box_b, 12000, 0.6)):
        box = boxfn(n)
        pos = synth.frame(n, box, 0)
        r = o32.search_single_pbc(rc, pos, o32.box_from_matrix(box), 7, nthreads=4)
        h = hashlib.sha256()
        h.update(r["i"].astype("<u4").tobytes()); h.update(r["j"].astype("<u4").tobytes()); h.update(r["d"].astype("<f4").tobytes())
        dig[name] = dict(natoms=n, cutoff=rc, box=name.split("_")[0] + "_" + name.split("_")[1], npairs=int(len(r["i"])),
                         dims=list(r["dims"]), sha256_i_j_d=h.hexdigest())
    json.dump(dig, open(os.path.join(HERE, "ordered_pair_digests.json"), "w"), indent=1)


def make_c2_digest(o32, nthreads=8):
    """BASELINE.json configs[1] at its full size: the 1M-atom box-A frame 0 at rc = 1.2 nm (3.6e8 ordered pairs).  Needs
    ~25 GB of host memory and a few minutes on 8 cores; kept out of the default run (python make_golden.py c2)."""
    n, rc = 1_000_000, 1.2
    box = synth.box_a(n)
    pos = synth.frame(n, box, 0)
    r = o32.search_single_pbc(rc, pos, o32.box_from_matrix(box), 7, nthreads=nthreads)
    h = hashlib.sha256()
    step = 1 << 24
    for key, dt in (("i", "<u4"), ("j", "<u4"), ("d", "<f4")):
        for k in range(0, len(r[key]), step):
            h.update(r[key][k:k + step].astype(dt).tobytes())
    path = os.path.join(HERE, "ordered_pair_digests.json")
    dig = json.load(open(path))
    dig["tric_a_1000000_rc1.2"] = dict(natoms=n, cutoff=rc, box="tric_a", npairs=int(len(r["i"])), dims=list(r["dims"]),
                                       sha256_i_j_d=h.hexdigest())
    json.dump(dig, open(path, "w"), indent=1)


RUST_FIXTURES = os.path.join(ROOT, "rust", "molar_hip", "tests", "fixtures")
RUST_EXPORTS = ("search_ortho", "search_tric_a", "search_hex_b", "search_rhombic_dodecahedron", "measure")
_DT = {"float32": "f32", "float64": "f64", "uint8": "u8", "uint32": "u32", "uint64": "u64", "int64": "i64"}


def export_for_rust():
    """The search and Measure fixtures once more, in a format a Rust test can read without any crate: one raw
    little-endian .bin per array (C order) + manifest.json {fixture: {key: {dtype, shape, file}}}.  The consumer is
    rust/molar_hip/tests/parity.rs, which runs MolAR ITSELF on the inputs and compares with the committed outputs -
    the one-command route from "parity unpinned by the reference" to pinned once a Rust toolchain is at hand.
    tests/test_rust_parity_cpu.py keeps the export equal to the .npz files."""
    manifest = {}
    for name in RUST_EXPORTS:
        z = np.load(os.path.join(HERE, name + ".npz"))
        os.makedirs(os.path.join(RUST_FIXTURES, name), exist_ok=True)
        entry = {}
        for key in z.files:
            a = np.ascontiguousarray(z[key])
            dt = _DT[str(a.dtype)]
            rel = f"{name}/{key}.bin"
            a.astype(a.dtype.newbyteorder("<")).tofile(os.path.join(RUST_FIXTURES, rel))
            entry[key] = {"dtype": dt, "shape": list(a.shape), "file": rel}
        manifest[name] = entry
    json.dump({"format": "raw little-endian arrays, C order", "generator": "tests/golden/make_golden.py bin", "fixtures": manifest},
              open(os.path.join(RUST_FIXTURES, "manifest.json"), "w"), indent=1, sort_keys=True)
    print("exported", sum(len(v) for v in manifest.values()), "arrays to", RUST_FIXTURES)


if __name__ == "__main__":
    if sys.argv[1:] == ["bin"]:
        export_for_rust()
        sys.exit(0)
    o32, o64 = Oracle("f32"), Oracle("f64")
    if sys.argv[1:] == ["c2"]:
        make_c2_digest(o32)
        sys.exit(0)
    make_search(o32)
    make_measure(o32, o64)
    make_membrane(o32)
    make_digests(o32)
    print("fixtures written to", HERE)
