"""The membrane fixture of the Rust-side parity test (rust/molar_hip/tests/membrane.rs, source only): the committed files
equal a fresh computation with the CPU checker, the manifest describes them, and the Rust test reads every array."""
import json
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = os.path.join(ROOT, "rust", "molar_hip", "tests", "fixtures", "membrane_cg")
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


@pytest.fixture(scope="module")
def gen():
    import make_membrane_fixture as g
    return g


def test_committed_structure_and_options_are_the_generators(gen, tmp_path):
    xyz, box, per = gen.bilayer()
    p = tmp_path / "bilayer.gro"
    gen.write_structure(str(p), xyz, box)
    assert open(p).read() == open(os.path.join(FIX, "bilayer.gro")).read()
    assert open(os.path.join(FIX, "options.toml")).read() == gen.TOML
    # the options name exactly the beads of the structure
    names = {ln[10:15].strip() for ln in open(p).read().split("\n")[2:-2]}
    assert names == set(gen.BEADS)
    for t, idx in zip(re.findall(r'"((?:\w+-)+\w+)"', gen.TOML), gen.TAILS):
        assert t.split("-") == [gen.BEADS[i] for i in idx]


def test_committed_arrays_equal_a_fresh_computation(gen):
    man = json.load(open(os.path.join(FIX, "manifest.json")))
    e = gen.expected(os.path.join(FIX, "bilayer.gro"))
    assert man["nlipids"] == e.pop("nlipids") and man["cutoff"] == gen.CUTOFF
    assert set(man["arrays"]) == set(e)
    on_disk = {f[:-4] for f in os.listdir(FIX) if f.endswith(".bin")}
    assert on_disk == set(e)
    for k, a in e.items():
        a = np.ascontiguousarray(a)
        assert man["arrays"][k] == {"dtype": a.dtype.name, "shape": list(a.shape)}
        got = np.fromfile(os.path.join(FIX, k + ".bin"), dtype=a.dtype.newbyteorder("<")).reshape(a.shape)
        assert np.array_equal(got, a), k
    K = man["nlipids"]
    assert e["valid"].sum() > 0.9 * K and abs(e["area"][e["valid"] > 0].mean() - 0.64) < 0.05
    # edge lipids really are split in the file: the reference has to make them whole
    xyz, box, per = gen.bilayer()
    span = np.ptp(xyz.reshape(K, len(gen.BEADS), 3)[:, :, :2], axis=1).max(axis=1)
    assert (span > 0.5 * box[0, 0]).sum() >= 4


def test_rust_test_reads_every_array(gen):
    src = open(os.path.join(ROOT, "rust", "molar_hip", "tests", "membrane.rs")).read()
    man = json.load(open(os.path.join(FIX, "manifest.json")))
    read = set(re.findall(r'(?:f32s|u64s|u32s|raw)\("(\w+)"\)', src))
    assert read == set(man["arrays"]), (sorted(read), sorted(man["arrays"]))
    assert "Membrane::new(&mut sys, &toml)" in src and "memb.compute(&sys)" in src
    cargo = open(os.path.join(ROOT, "rust", "molar_hip", "Cargo.toml")).read()
    assert re.search(r"^molar_membrane\s*=", cargo, flags=re.M)
