"""The GPU bilayer analysis on the fixture the Rust-side parity test feeds to MolAR's own molar_membrane
(rust/molar_hip/tests/fixtures/membrane_cg/, rust/molar_hip/tests/membrane.rs): the structure is read from the GRO file the
way MolAR reads it, the lipid description is the one of options.toml, and every number the Rust test compares is compared
here - so the day `cargo test --test membrane` is green, these rows of the GPU path are pinned by the reference."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = os.path.join(ROOT, "rust", "molar_hip", "tests", "fixtures", "membrane_cg")


def ring(ids):
    """Neighbours of a Voronoi cell as a canonical ring: the clipping walk of voronoi_cell.rs leaves two vertices on one
    neighbour's edge wherever a later bisector passes within rounding of an existing vertex, and where it starts the ring
    depends on the last cut - both flip with the last bit of the inputs (the GPU sums the markers in f64, MolAR and the
    checker in f32).  Consecutive repeats are collapsed and the ring is rotated to its smallest id."""
    ids = [int(x) for x in ids]
    out = [x for k, x in enumerate(ids) if x != ids[k - 1]] if len(ids) > 1 else ids
    if not out:
        out = ids[:1]
    k = out.index(min(out))
    return out[k:] + out[:k]


def load(key, man):
    d = man["arrays"][key]
    return np.fromfile(os.path.join(FIX, key + ".bin"), dtype=np.dtype(d["dtype"]).newbyteorder("<")).reshape(d["shape"])


@pytest.mark.parametrize("fused", [True, False])
def test_gpu_membrane_on_the_rust_fixture(fused):
    from molar_amd import build, gro
    from molar_amd import membrane as mb
    from molar_amd.api import Engine
    build.build_library()
    eng = Engine(0)
    man = json.load(open(os.path.join(FIX, "manifest.json")))
    top, st = gro.read_gro(os.path.join(FIX, "bilayer.gro"))
    # the lipid of options.toml: head "name P N", mid "name C1 C2", tails C1A-C2A-C3A-C4A / C1B-C2B-C3B-C4B, tail marker =
    # the last carbon of each tail (lib.rs:127-133)
    names = top.names[:12]
    at = {n: i for i, n in enumerate(names)}
    tails = [np.array([at[n] for n in t.split("-")]) for t in ("C1A-C2A-C3A-C4A", "C1B-C2B-C3B-C4B")]
    tpl = mb.LipidTemplate(12, np.array([at["N"], at["P"]]), np.array([at["C1"], at["C2"]]), np.array([t[-1] for t in tails]), tails,
                           [np.ones(3, np.uint8), np.ones(3, np.uint8)])
    K = man["nlipids"]
    first = np.arange(K) * 12
    m = mb.Membrane(eng, len(st.coords), first, tpl, top.masses, mb.MembraneOptions(cutoff=man["cutoff"], order_type=2, fused=fused))
    xyz = np.ascontiguousarray(st.coords, np.float32).copy()
    box = st.pbox.get_matrix()
    res = m.compute(xyz, box)
    tol = dict(rtol=2e-5, atol=2e-5)
    assert np.allclose(res["head"], load("head_marker_new", man), **tol)
    assert np.allclose(res["mid"], load("mid_marker_new", man), **tol)
    assert np.allclose(res["tail"], load("tail_marker_new", man), **tol)
    assert np.array_equal(res["patch_off"], load("patch_offsets", man)) and np.array_equal(res["patch_ids"], load("patch_ids", man))
    valid = load("valid", man)
    assert np.array_equal(res["valid"], valid)
    ok = valid.astype(bool)
    noff, nids, nvert = load("neib_offsets", man), load("neib_ids", man), load("nvert", man)
    exact = 0
    for k in np.flatnonzero(ok):
        s0 = int(res["patch_off"][k]) + 4 * k
        got, want = res["neib_ids"][s0:s0 + int(res["nvert"][k])], nids[int(noff[k]): int(noff[k + 1])]
        assert ring(got) == ring(want), k
        exact += int(res["nvert"][k] == nvert[k] and np.array_equal(got, want))
    assert exact >= 0.95 * ok.sum(), exact      # nearly every cell comes out vertex for vertex
    for got, want in (("smoothed_head", "head_marker"), ("normals", "normal"), ("mean_curv", "mean_curv"), ("gauss_curv", "gaussian_curv")):
        assert np.allclose(res[got][ok], load(want, man)[ok], **tol), got
    assert np.allclose(res["area"][ok], load("area", man)[ok], rtol=1e-3, atol=0)        # the fan area sees the doubled vertices
    order = load("order", man)
    for t in range(2):
        assert np.allclose(res["order"][t][ok], order[ok, t], rtol=3e-5, atol=3e-5)
