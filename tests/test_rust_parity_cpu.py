"""rust/molar_hip/tests/parity.rs runs MolAR itself on fixture inputs and compares with the committed outputs - the route
from "parity unpinned by the reference" to pinned.  No Rust toolchain exists here, so this file checks what can be checked
without one: the raw export equals the .npz fixtures bit for bit, the manifest covers every key, and the Rust test reads
(or explicitly skips) every exported key."""
import json
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = os.path.join(ROOT, "rust", "molar_hip", "tests", "fixtures")
GOLD = os.path.join(ROOT, "tests", "golden")
NP = {"f32": "<f4", "f64": "<f8", "u8": "u1", "u32": "<u4", "u64": "<u8", "i64": "<i8"}


def manifest():
    return json.load(open(os.path.join(FIX, "manifest.json")))["fixtures"]


def test_export_equals_the_npz_fixtures():
    m = manifest()
    assert set(m) == {"search_ortho", "search_tric_a", "search_hex_b", "search_rhombic_dodecahedron", "measure"}
    for name, entry in m.items():
        z = np.load(os.path.join(GOLD, name + ".npz"))
        assert set(entry) == set(z.files), name          # the manifest covers every fixture key
        for key, d in entry.items():
            raw = np.fromfile(os.path.join(FIX, d["file"]), dtype=NP[d["dtype"]])
            want = np.ascontiguousarray(z[key])
            assert list(want.shape) == d["shape"], (name, key)
            assert raw.size == want.size and raw.tobytes() == want.astype(NP[d["dtype"]]).tobytes(), (name, key)


def test_no_stray_files_in_the_export():
    m = manifest()
    listed = {d["file"] for e in m.values() for d in e.values()}
    found = {os.path.relpath(os.path.join(r, f), FIX) for r, _, fs in os.walk(FIX) for f in fs if f.endswith(".bin")}
    found = {f for f in found if not f.startswith("membrane_cg" + os.sep)}      # tests/membrane.rs' fixture has its own manifest and test
    assert listed == found


def test_parity_rs_reads_or_skips_every_key():
    src = open(os.path.join(ROOT, "rust", "molar_hip", "tests", "parity.rs")).read()
    skipped = set(re.findall(r'\("(\w+)",\s*"[^"]*"\)', src.split("pub const SKIPPED")[1].split("];")[0]))
    quoted = set(re.findall(r'"(\w+)"', src))
    m = manifest()
    # list-valued search outputs are read through same_list(.., prefix): prefix_i / prefix_j / prefix_d
    prefixes = set(re.findall(r'same_list\(&got, name, "(\w+)"\)', src))
    for name, entry in m.items():
        for key in entry:
            via_prefix = any(key == p + s for p in prefixes for s in ("_i", "_j", "_d"))
            assert key in quoted or via_prefix or key in skipped, f"{name}/{key} is neither compared nor listed in SKIPPED"
    # every search variant of the fixtures that MolAR exports publicly is compared
    assert prefixes == {"single_pbc7", "single_pbc3", "single", "double_pbc7", "double", "vdw_pbc7", "vdw"}
    for fn in ("distance_search_single_pbc", "distance_search_single", "distance_search_double_pbc", "distance_search_double",
               "distance_search_double_vdw_pbc", "distance_search_double_vdw", "fit_transform", "rmsd", "rmsd_mw", "gyration",
               "gyration_pbc", "inertia", "lipid_tail_order", "unwrap_simple_dim", "apply_transform", "center_of_mass_pbc_dims"):
        assert re.search(r"\b" + fn + r"\b", src), fn
