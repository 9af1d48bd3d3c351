#!/usr/bin/env python
"""Fixture for the Rust-side parity test of the bilayer analysis (rust/molar_hip/tests/membrane.rs): a small coarse-grained
bilayer as a GRO file, the options as TOML, and what one Membrane::new + Membrane::compute
(molar_membrane/src/lib.rs:88-200, 410-454) must leave in every LipidMolecule - computed here with the CPU checker's
primitives assembled in the reference's order (the same assembly tests/test_gpu_membrane.py holds the GPU path against).

    python tests/golden/make_membrane_fixture.py            # rewrites rust/molar_hip/tests/fixtures/membrane_cg/

Test infrastructure: uses oracle/, never imported by the product.  tests/test_rust_membrane_fixture_cpu.py regenerates
everything in memory and compares with the committed files."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "rust", "molar_hip", "tests", "fixtures", "membrane_cg")

# bead names MolAR's element guessing gives a mass for (atom.rs:238-291): a MARTINI-like 12-bead lipid with atomistic names
BEADS = ["N", "P", "C1", "C2", "C1A", "C2A", "C3A", "C4A", "C1B", "C2B", "C3B", "C4B"]
HEAD, MID = [0, 1], [2, 3]                     # "name P N", "name C1 C2"
TAILS = [[4, 5, 6, 7], [8, 9, 10, 11]]         # "C1A-C2A-C3A-C4A", "C1B-C2B-C3B-C4B"
BONDS = [[1, 1, 1], [1, 1, 1]]
CUTOFF = 2.5
TOML = '''sel = "resname POPC"
order_type = "ScdCorr"
output_dir = "./membrane_cg_results"
max_smooth_iter = 1
cutoff = 2.5

[lipids.POPC]
whole = "resname POPC"
head = "name P N"
mid = "name C1 C2"
tails = [
    "C1A-C2A-C3A-C4A",
    "C1B-C2B-C3B-C4B",
]
'''


def bilayer(side=10, seed=20240611):
    """2 x side^2 lipids on a jittered square lattice of 0.8 nm, a gentle undulation, wrapped into the box so that the
    lipids at the edges are split over the periodic boundary (Membrane::new makes them whole, lib.rs:116-118).  The box is
    8 nm across: more than three cutoffs, so the search grid has three cells per dimension and the pair list holds no
    duplicates (with two cells the reference emits pairs twice, a patch then holds a neighbour twice, and the second
    insertion of the same point into a Voronoi cell doubles vertices at random)."""
    rng = np.random.default_rng(seed)
    L, Lz = side * 0.8, 9.0
    per = side * side
    xyz = np.zeros((2 * per * len(BEADS), 3))
    for k in range(2 * per):
        leaf, a = divmod(k, per)
        sgn = 1.0 if leaf == 0 else -1.0
        cx = (a % side + 0.5 + 0.15 * rng.normal()) * 0.8
        cy = (a // side + 0.5 + 0.15 * rng.normal()) * 0.8
        z0 = Lz / 2 + 0.25 * np.sin(2 * np.pi * cx / L) * np.cos(2 * np.pi * cy / L)
        p = np.zeros((len(BEADS), 3))
        p[0] = [cx, cy, z0 + sgn * 2.25]; p[1] = [cx + 0.05, cy, z0 + sgn * 2.0]
        p[2] = [cx - 0.15, cy, z0 + sgn * 1.65]; p[3] = [cx + 0.15, cy, z0 + sgn * 1.65]
        for t, (x0, beads) in enumerate(((-0.2, TAILS[0]), (0.2, TAILS[1]))):
            for c, b in enumerate(beads):
                p[b] = [cx + x0 + 0.03 * (c % 2), cy + 0.02 * c, z0 + sgn * (1.3 - 0.38 * c)]
        p += rng.normal(0, 0.03, p.shape)
        xyz[k * len(BEADS):(k + 1) * len(BEADS)] = p
    xyz[:, 0] %= L; xyz[:, 1] %= L
    box = np.diag([L, L, Lz]).astype(np.float32)
    return np.round(xyz, 3).astype(np.float32), box, per


def write_structure(path, xyz, box):
    from molar_amd import api, gro
    n = len(xyz)
    names = [BEADS[i % len(BEADS)] for i in range(n)]
    top = gro.GroTopology(names, ["POPC"] * n, [i // len(BEADS) + 1 for i in range(n)])
    gro.write_gro(path, top, api.State(xyz, api.PeriodicBox.from_matrix(box), 0.0))


def normals_two_passes(head, tail, lists):
    """compute_initial_normals (lib.rs:456-505) in f32; the second pass reads what it has already written"""
    f = np.float32
    K = len(head)

    def norm(v):
        return f(np.sqrt(f(f(v[0] * v[0] + v[1] * v[1]) + v[2] * v[2])))

    def within(a, b):
        n1, n2 = norm(a), norm(b)
        if n1 == 0 or n2 == 0:
            return True
        c = f(f(f(a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]) / f(n1 * n2))
        return bool(f(np.arccos(np.clip(c, f(-1), f(1)))) <= f(np.pi / 2))
    thv = np.zeros((K, 3), f)
    for i in range(K):
        v = (head[i] - tail[i]).astype(f)
        thv[i] = v / norm(v)
    nv = np.zeros((K, 3), f)
    for p in range(2):
        src = thv if p == 0 else nv
        for i in range(K):
            s = np.zeros(3, f)
            for l in lists[i]:
                if within(src[l], src[i]):
                    s = (s + src[l]).astype(f)
            s = (s + src[i]).astype(f)
            nv[i] = s / norm(s)
    return nv


def expected(gro_path):
    """What Membrane::new + one Membrane::compute leave, from the file as MolAR reads it."""
    from oracle import gro_oracle
    from oracle.oracle import Oracle
    o = Oracle("f32")
    g = gro_oracle.read_gro(gro_path)
    xyz = np.ascontiguousarray(g["xyz"], np.float32)
    masses = np.ascontiguousarray(g["mass"], np.float32)
    box = np.ascontiguousarray(g["box"], np.float32)
    ob = o.box_from_matrix(box)
    nb = len(BEADS)
    K = len(xyz) // nb
    # Membrane::new: every lipid made whole (:116-118), then the three markers (:135-137); the tail marker is the centre
    # of mass of the LAST carbon of each tail (:127-133)
    for k in range(K):
        xyz[k * nb:(k + 1) * nb] = o.unwrap_simple_dim(xyz[k * nb:(k + 1) * nb], ob, 7)
    u64 = lambda a: np.asarray(a, np.uint64)
    com = lambda k, sub: o.center_of_mass(xyz[k * nb:(k + 1) * nb], masses[k * nb:(k + 1) * nb], u64(sub))
    head = np.array([com(k, HEAD) for k in range(K)], np.float32)
    mid = np.array([com(k, MID) for k in range(K)], np.float32)
    tail = np.array([com(k, [t[-1] for t in TAILS]) for k in range(K)], np.float32)
    # compute_patches (:539-558)
    r = o.search_single_pbc(CUTOFF, head, ob, 7)
    lists = [[] for _ in range(K)]
    for i, j in zip(r["i"].tolist(), r["j"].tolist()):
        lists[i].append(j); lists[j].append(i)
    patch_off = np.concatenate([[0], np.cumsum([len(l) for l in lists])]).astype(np.uint64)
    patch_ids = np.array([x for l in lists for x in l], np.uint64)
    n0 = normals_two_passes(head, tail, lists)
    s = o.membrane_smooth(ob, head, n0, np.ones(K, np.uint8), patch_off, patch_ids)
    valid = s["valid"].astype(np.uint8)
    neib_off = np.concatenate([[0], np.cumsum(np.where(valid > 0, s["nvert"], 0))]).astype(np.uint64)
    neib = np.concatenate([s["neib_ids"][int(patch_off[k]) + 4 * k: int(patch_off[k]) + 4 * k + int(s["nvert"][k])] if valid[k] else
                           np.zeros(0, np.uint64) for k in range(K)]).astype(np.uint64)
    # compute_order (:435-443): ScdCorr with the lipid's fitted normal, valid lipids only
    order = np.zeros((K, len(TAILS), 2), np.float32)
    for k in range(K):
        if not valid[k]:
            continue
        for t, carbons in enumerate(TAILS):
            order[k, t] = o.lipid_tail_order(xyz[k * nb:(k + 1) * nb], 2, s["normals"][k][None, :], np.asarray(BONDS[t], np.uint8), idx=u64(carbons))
    return dict(nlipids=K, head_marker_new=head, mid_marker_new=mid, tail_marker_new=tail, patch_offsets=patch_off, patch_ids=patch_ids,
                valid=valid, head_marker=s["head"].astype(np.float32), normal=s["normals"].astype(np.float32),
                mean_curv=s["mean_curv"].astype(np.float32), gaussian_curv=s["gauss_curv"].astype(np.float32),
                area=s["area"].astype(np.float32), nvert=s["nvert"].astype(np.uint32), neib_offsets=neib_off, neib_ids=neib, order=order)


def main():
    os.makedirs(OUT, exist_ok=True)
    xyz, box, per = bilayer()
    gro_path = os.path.join(OUT, "bilayer.gro")
    write_structure(gro_path, xyz, box)
    open(os.path.join(OUT, "options.toml"), "w").write(TOML)
    e = expected(gro_path)
    manifest = {"nlipids": int(e.pop("nlipids")), "cutoff": CUTOFF, "structure": "bilayer.gro", "options": "options.toml", "arrays": {}}
    for k, a in e.items():
        a = np.ascontiguousarray(a)
        a.astype(a.dtype.newbyteorder("<")).tofile(os.path.join(OUT, k + ".bin"))
        manifest["arrays"][k] = {"dtype": a.dtype.name, "shape": list(a.shape)}
    json.dump(manifest, open(os.path.join(OUT, "manifest.json"), "w"), indent=1, sort_keys=True)
    nvalid = int(e["valid"].sum())
    print(f"{manifest['nlipids']} lipids, {nvalid} valid, {len(e['patch_ids'])} patch entries, mean area {e['area'][e['valid'] > 0].mean():.3f} nm^2, "
          f"mean |Scd| {np.abs(e['order'][e['valid'] > 0]).mean():.3f}")


if __name__ == "__main__":
    main()
