"""Per-lipid bilayer analysis on the GPU engine — the part of molar_membrane::Membrane::compute
(molar_membrane/src/lib.rs:410-454) that runs per frame on atom coordinates:

    markers      head / mid / tail-end centres of mass per lipid       lipid_molecule.rs:65-99, lib.rs:135-137
    patches      PBC neighbour search among head markers               lib.rs:539-558
    normals      tail->head vectors, two neighbour-averaging passes    lib.rs:456-505
    order        lipid_tail_order per tail with the lipid's normal     lib.rs:435-443, lipid_molecule.rs:48-59

The iterative surface smoothing (quadric fit, Voronoi cells, curvature: lib.rs:661-812) is NOT implemented;
see DESIGN.md.  All per-lipid loops are batched: one launch per stage for all lipids of a frame.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from . import api


@dataclass
class LipidTemplate:
    """Atom offsets inside one lipid (all lipids of a species share them; cf. lipid_species.rs)."""
    natoms: int
    head: np.ndarray            # offsets of the head-marker atoms
    mid: np.ndarray
    tail_end: np.ndarray
    tails: list                 # list of arrays: carbons of each tail, in chain order
    bond_orders: list           # list of uint8 arrays (len n-1) per tail


@dataclass
class MembraneOptions:          # molar_membrane/src/lib.rs:53-85 (subset)
    cutoff: float = 2.5
    order_type: int = 1         # 0 Sz, 1 Scd, 2 ScdCorr
    unwrap: bool = True


def pope_like_template() -> LipidTemplate:
    """A 52-bead POPE-like lipid: 12 head-group atoms, glycerol bridge, an oleoyl-like tail (18 C, one double
    bond) and a palmitoyl-like tail (16 C)."""
    head = np.arange(0, 12)
    mid = np.arange(12, 18)
    t1 = np.arange(18, 36)      # 18 carbons
    t2 = np.arange(36, 52)      # 16 carbons
    bo1 = np.ones(17, np.uint8); bo1[8] = 2
    bo2 = np.ones(15, np.uint8)
    return LipidTemplate(52, head, mid, np.array([t1[-1], t1[-2], t2[-1], t2[-2]]), [t1, t2], [bo1, bo2])


def build_bilayer(nlipids_per_leaflet: int, natoms_total: int, seed: int = 20240607, area_per_lipid: float = 0.62):
    """Synthetic bilayer in an orthorhombic box: two leaflets on a jittered square lattice, tails pointing to
    the mid-plane, remaining atoms as 'water' filling the box.  Returns (xyz float32 [N,3], box 3x3,
    lipid_first_atom int array, template, masses)."""
    rng = np.random.default_rng(seed)
    tpl = pope_like_template()
    side = int(np.ceil(np.sqrt(nlipids_per_leaflet)))
    L = side * np.sqrt(area_per_lipid)
    nlip = 2 * nlipids_per_leaflet
    nwater = natoms_total - nlip * tpl.natoms
    assert nwater >= 0
    Lz = max(8.0, natoms_total / 100.0 / (L * L))      # ~100 atoms/nm^3 overall
    box = np.diag([L, L, Lz]).astype(np.float32)
    xyz = np.zeros((natoms_total, 3), np.float32)
    first = np.arange(nlip) * tpl.natoms
    zmid = Lz / 2
    k = 0
    for leaflet, sgn in ((0, 1.0), (1, -1.0)):
        for a in range(nlipids_per_leaflet):
            gx, gy = a % side, a // side
            cx = (gx + 0.5 + 0.2 * rng.normal()) * L / side
            cy = (gy + 0.5 + 0.2 * rng.normal()) * L / side
            p = np.zeros((tpl.natoms, 3))
            # head group: blob 1.9-2.1 nm from the mid-plane
            p[tpl.head] = [cx, cy, zmid + sgn * 2.0] + 0.15 * rng.normal(size=(len(tpl.head), 3))
            p[tpl.mid] = [cx, cy, zmid + sgn * 1.6] + 0.10 * rng.normal(size=(len(tpl.mid), 3))
            for t, carbons in enumerate(tpl.tails):
                z = zmid + sgn * 1.5
                x, y = cx + (0.25 if t == 0 else -0.25), cy
                for c, off in enumerate(carbons):
                    z -= sgn * 0.09 + 0.01 * rng.normal() * sgn
                    x += 0.04 * rng.normal() + (0.03 if c % 2 else -0.03)
                    y += 0.04 * rng.normal()
                    p[off] = [x, y, z]
            xyz[first[k]: first[k] + tpl.natoms] = p
            k += 1
    if nwater:
        w = rng.random((nwater, 3)) * [L, L, Lz]
        # keep water out of the hydrophobic core
        core = np.abs(w[:, 2] - zmid) < 1.7
        w[core, 2] = (w[core, 2] + Lz / 2) % Lz
        xyz[nlip * tpl.natoms:] = w
    xyz %= np.array([L, L, Lz], np.float32)             # wrap: lipids near the edges get split over PBC
    masses = np.resize(np.array([12.011, 12.011, 15.999, 14.007], np.float32), natoms_total)
    return xyz.astype(np.float32), box, first, tpl, masses


class Membrane:
    def __init__(self, engine: api.Engine, natoms: int, lipid_first_atom, template: LipidTemplate, masses,
                 options: MembraneOptions | None = None):
        self.eng = engine
        self.opt = options or MembraneOptions()
        self.tpl = template
        self.first = np.asarray(lipid_first_atom, np.uint64)
        self.K = len(self.first)
        self.masses = np.ascontiguousarray(masses, np.float32)
        f = self.first[:, None]
        # CSR of the three marker selections, lipid-major: [head_0, mid_0, tail_0, head_1, ...]
        parts, lens = [], []
        for sub in (template.head, template.mid, template.tail_end):
            parts.append(f + np.asarray(sub, np.uint64)[None, :]); lens.append(len(sub))
        self.marker_idx = np.ascontiguousarray(np.concatenate(parts, axis=1).reshape(-1))
        self.marker_off = np.concatenate([[0], np.cumsum(np.tile(lens, self.K))]).astype(np.uint64)
        # whole-lipid CSR (unwrap) and tails CSR (order)
        self.lipid_idx = np.ascontiguousarray((f + np.arange(template.natoms, dtype=np.uint64)[None, :]).reshape(-1))
        self.lipid_off = (np.arange(self.K + 1, dtype=np.uint64) * template.natoms)
        tl = [np.asarray(t, np.uint64) for t in template.tails]
        self.tail_idx = np.ascontiguousarray(np.concatenate([f + t[None, :] for t in tl], axis=1).reshape(-1))
        self.tail_off = np.concatenate([[0], np.cumsum(np.tile([len(t) for t in tl], self.K))]).astype(np.uint64)
        self.tail_bonds = np.ascontiguousarray(np.tile(np.concatenate(template.bond_orders), self.K))
        self.ntails = len(tl)
        self.tail_lens = [len(t) for t in tl]

    def compute(self, xyz, box):
        """One frame.  xyz: float32 [N,3] (numpy; unwrapped in place when options.unwrap).  Returns dict with
        head/mid/tail markers [K,3], patch CSR, normals [K,3], order: list over tails of [K, n_t-2]."""
        e, K = self.eng, self.K
        pb = box if isinstance(box, api.PeriodicBox) else api.PeriodicBox.from_matrix(box)
        if self.opt.unwrap:                                                     # lipid_molecule.rs:75-76
            e.unwrap_simple_batch(xyz, self.lipid_idx, self.lipid_off, pb)
        mk = e.center_batch(xyz, self.marker_idx, self.marker_off, self.masses).reshape(K, 3, 3)
        head, mid, tail = mk[:, 0].copy(), mk[:, 1].copy(), mk[:, 2].copy()
        # compute_patches (lib.rs:539-558): ids are lipid ids -> local ids of the marker array
        n = e.search_count(api.SEARCH_SINGLE, self.opt.cutoff, head, box=pb, pbc=api.PBC_FULL, ids_local=True)
        pairs, _ = e.search_fill(n)
        i = pairs[:, 0].astype(np.int64); j = pairs[:, 1].astype(np.int64)
        # patch_ids[i].push(j); patch_ids[j].push(i) in pair order
        src = np.stack([i, j], 1).reshape(-1); dst = np.stack([j, i], 1).reshape(-1)
        order = np.argsort(src, kind="stable")
        patch_ids = dst[order].astype(np.uint64)
        patch_off = np.concatenate([[0], np.cumsum(np.bincount(src, minlength=K))]).astype(np.uint64)
        normals = api.membrane_initial_normals(head, tail, patch_off, patch_ids)
        # compute_order (lib.rs:435-443): one normal per lipid, shared by its tails
        nrm = np.repeat(normals, self.ntails, axis=0)
        noff = np.arange(K * self.ntails + 1, dtype=np.uint64)
        flat = e.lipid_tail_order_csr(xyz, self.tail_idx, self.tail_off, self.opt.order_type, nrm, noff, self.tail_bonds)
        per_lipid = sum(l - 2 for l in self.tail_lens)
        flat = flat.reshape(K, per_lipid)
        out, pos = [], 0
        for l in self.tail_lens:
            out.append(flat[:, pos:pos + l - 2].copy()); pos += l - 2
        return dict(head=head, mid=mid, tail=tail, patch_off=patch_off, patch_ids=patch_ids, normals=normals, order=out)
