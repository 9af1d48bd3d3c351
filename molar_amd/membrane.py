"""Per-lipid bilayer analysis on the GPU engine — the part of molar_membrane::Membrane::compute
(molar_membrane/src/lib.rs:410-454) that runs per frame on atom coordinates:

    markers      head / mid / tail-end centres of mass per lipid       lipid_molecule.rs:65-99, lib.rs:135-137
    patches      PBC neighbour search among head markers               lib.rs:539-558
    normals      tail->head vectors, two neighbour-averaging passes    lib.rs:456-505
    smoothing    quadric fit, Voronoi cell, curvature, marker update   lib.rs:661-812, voronoi_cell.rs, lipid_molecule.rs:102-196
    n-th shell   patches / curvature averaging over neighbour shells    lib.rs:562-621
    order        lipid_tail_order per tail with the lipid's normal     lib.rs:435-443, lipid_molecule.rs:48-59

All per-lipid loops are batched: one launch per stage for all lipids of a frame.  The reference computes the
markers once in Membrane::new and its per-frame refresh is commented out (lipid_molecule.rs:65-99); this class
refreshes them every frame, which is what the commented code does.  Where the reference iterates a HashSet
(n-th shell patches, curvature averaging: molar_hip_membrane_nth_shell_patches / _smooth_curvature, host arithmetic of the
engine) the order is unspecified there; here it is ascending lipid id.
Group statistics and their text output (stats.rs, lipid_group.rs) are host bookkeeping: membrane_stats.py.
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
    order_type: int = 2         # 0 Sz, 1 Scd, 2 ScdCorr (the reference's default)
    max_smooth_iter: int = 1
    n_shells_patch: int = 0     # >0: re-define patches as the n-th Voronoi neighbour shell after a first pass
    n_shells_smoothing: int = 0 # >0: average curvatures over the n-th neighbour shell
    global_normal: object = None
    unwrap: bool = True
    fused: bool = True          # one chained call per frame (molar_hip_membrane_frame_*) where the options allow it


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
        self.valid = np.ones(self.K, np.uint8)                                  # LipidMolecule::valid, sticky across frames
        self.species_names = ["LIP"]                                            # one template = one species
        self.species_of_lipid = np.zeros(self.K, np.int64)
        self.groups = {}

    # ---- groups (lib.rs:261-345, lipid_group.rs)
    def add_ids_to_group(self, name, ids):
        from .membrane_stats import LipidGroup
        ids = np.asarray(ids, np.int64)
        if len(ids) and (ids.min() < 0 or ids.max() >= self.K):
            raise ValueError(f"lipid id out of bounds 0:{self.K}")                # lib.rs:303-309
        if name not in self.groups:
            self.groups[name] = LipidGroup(self.species_names, {sp: self.tail_lens for sp in self.species_names})
        self.groups[name].lipid_ids = np.concatenate([self.groups[name].lipid_ids, ids])

    def reset_groups(self):                                                     # lib.rs:261-267
        for g in self.groups.values():
            g.lipid_ids = np.zeros(0, np.int64)

    def finalize(self, output_dir="."):                                         # lib.rs:517-537
        for name, g in self.groups.items():
            g.save(output_dir, name)

    def reset_valid_lipids(self):                                               # lib.rs:269-273
        self.valid[:] = 1
        self._valid_dev = None

    @staticmethod
    def _patch_csr(K, i, j):
        """patch_ids[i].push(j); patch_ids[j].push(i) in pair order (lib.rs:553-556)."""
        src = np.stack([i, j], 1).reshape(-1); dst = np.stack([j, i], 1).reshape(-1)
        order = np.argsort(src, kind="stable")
        return (np.concatenate([[0], np.cumsum(np.bincount(src, minlength=K))]).astype(np.uint64),
                dst[order].astype(np.uint64))

    def _constants(self, xyz):
        """The per-trajectory index / mass columns: with frames resident on the GPU they are uploaded once and stay
        there (the C ABI takes device pointers); with host frames they are the numpy arrays."""
        if not api._is_torch(xyz):
            return dict(lipid_idx=self.lipid_idx, marker_idx=self.marker_idx, masses=self.masses, tail_idx=self.tail_idx,
                        tail_bonds=self.tail_bonds)
        if getattr(self, "_dev", None) is None or self._dev["device"] != xyz.device:
            import torch
            up = lambda a: torch.from_numpy(a.view(np.int64) if a.dtype == np.uint64 else a).to(xyz.device)
            self._dev = dict(device=xyz.device, lipid_idx=up(self.lipid_idx), marker_idx=up(self.marker_idx),
                             masses=up(self.masses), tail_idx=up(self.tail_idx), tail_bonds=up(self.tail_bonds))
        return self._dev

    # ---- the chained form: one begin/end pair per frame, two frames in flight
    def fusable(self):
        """The options the chained call covers (everything but the n-th shell variants, lib.rs:562-621)."""
        return self.opt.fused and self.opt.n_shells_patch == 0 and self.opt.n_shells_smoothing == 0

    def _plan(self):
        if getattr(self, "_plan_obj", None) is None:
            tail_lipid = np.repeat(np.arange(self.K, dtype=np.uint32), self.ntails)
            self._plan_obj = api.MembranePlan(self.eng, len(self.masses), self.lipid_idx, self.lipid_off, self.marker_idx,
                                              self.marker_off, self.masses, self.tail_idx, self.tail_off, tail_lipid,
                                              self.tail_bonds, self.opt.cutoff, self.opt.order_type, self.opt.max_smooth_iter,
                                              self.opt.unwrap, self.opt.global_normal)
            self._valid_dev = None              # the flags the plan is known to hold (None: push self.valid first)
            self._inflight = 0
        return self._plan_obj

    def compute_begin(self, xyz, box):
        """Enqueue one frame without waiting for it; returns a ticket for compute_end.  Frames are chained in begin
        order, the valid flags included, so begin(k+1) may come before end(k)."""
        plan = self._plan()
        if self._inflight == 0 and (self._valid_dev is None or not np.array_equal(self._valid_dev, self.valid)):
            plan.set_valid(self.valid)              # first frame, reset_valid_lipids, or flags edited by the caller
        t = plan.begin(xyz, box)
        self._inflight += 1
        return t

    def compute_end(self, ticket, names=None):
        """Wait for a frame begun with compute_begin; returns the same dict as compute (all arrays, or only `names`)."""
        plan = self._plan()
        want = list(api.MEMBRANE_ARRAYS) if names is None else list(dict.fromkeys(list(names) + ["valid"]))
        # the per-lipid arrays come with the frame's end (one wait); the arrays sized by its patch entries need the view first
        late = [k for k in want if k in api.MembranePlan._PATCH_SIZED]
        _, r = plan.end(ticket, [k for k in want if k not in late])
        if late:
            r.update(plan.fetch(ticket, late))
        r = {k: r[k] for k in want}
        self.valid[:] = r["valid"]
        self._inflight -= 1
        self._valid_dev = self.valid.copy() if self._inflight == 0 else None     # a younger frame is ahead of self.valid
        res = dict(r)
        if "order" in r:
            per_lipid = sum(l - 2 for l in self.tail_lens)
            flat = r["order"].reshape(self.K, per_lipid)
            out, pos = [], 0
            for l in self.tail_lens:
                out.append(flat[:, pos:pos + l - 2].copy()); pos += l - 2
            res["order"] = out
        if "patch_offsets" in r:
            res["patch_off"] = res.pop("patch_offsets")
        if self.groups and names is None:
            d = (res["head"] - res["tail"]).astype(np.float32)
            thv = d / np.sqrt((d * d).sum(1, dtype=np.float32))[:, None]
            for g in self.groups.values():
                g.frame_update(res, self.species_of_lipid, thv)
        return res

    def compute(self, xyz, box):
        """One frame (Membrane::compute, lib.rs:410-454).  xyz: float32 [N,3] (numpy; unwrapped in place when
        options.unwrap).  Returns a dict: markers, patch CSR, per-lipid state (valid, normals, curvatures, area,
        Voronoi neighbours/vertices) and order: list over tails of [K, n_t-2]."""
        if self.fusable():
            return self.compute_end(self.compute_begin(xyz, box))
        e, K, opt = self.eng, self.K, self.opt
        pb = box if isinstance(box, api.PeriodicBox) else api.PeriodicBox.from_matrix(box)
        cst = self._constants(xyz)
        if opt.unwrap:                                                          # lipid_molecule.rs:75-76
            e.unwrap_simple_batch(xyz, cst["lipid_idx"], self.lipid_off, pb)
        mk = e.center_batch(xyz, cst["marker_idx"], self.marker_off, cst["masses"]).reshape(K, 3, 3)
        head, mid, tail = mk[:, 0].copy(), mk[:, 1].copy(), mk[:, 2].copy()
        # compute_patches (lib.rs:539-558): search among the valid lipids' head markers, ids = lipid ids
        vidx = np.flatnonzero(self.valid).astype(np.uint64)
        n = e.search_count(api.SEARCH_SINGLE, opt.cutoff, head, idx1=vidx, box=pb, pbc=api.PBC_FULL, ids_local=False)
        pairs, _ = e.search_fill(n)
        patch_off, patch_ids = api.membrane_patches_from_pairs(pairs, K)
        normals = api.membrane_initial_normals(head, tail, patch_off, patch_ids, valid=self.valid)
        st = api.new_membrane_state(head, normals, self.valid, len(patch_ids))
        it = 0
        while True:                                                             # lib.rs:417-432 (at least one pass)
            if opt.n_shells_patch > 0 and it == 0:
                e.membrane_smooth(pb, st, patch_off, patch_ids)
                # patches_from_nth_shell (lib.rs:562-583); the slots of neib_ids follow the patch lists they were made with
                patch_off, patch_ids = api.membrane_nth_shell_patches(st["valid"], patch_off, patch_ids, st["nvert"], st["neib_ids"],
                                                                      opt.n_shells_patch)
            e.membrane_smooth(pb, st, patch_off, patch_ids)
            it += 1
            if it >= opt.max_smooth_iter:
                break
        self.valid[:] = st["valid"]
        # compute_order (lib.rs:435-443): one normal per lipid (or the global one), shared by its tails
        nl = st["normals"] if opt.global_normal is None else np.tile(np.asarray(opt.global_normal, np.float32), (K, 1))
        nrm = np.repeat(nl, self.ntails, axis=0)
        noff = np.arange(K * self.ntails + 1, dtype=np.uint64)
        flat = e.lipid_tail_order_csr(xyz, cst["tail_idx"], self.tail_off, opt.order_type, nrm, noff, cst["tail_bonds"])
        per_lipid = sum(l - 2 for l in self.tail_lens)
        flat = flat.reshape(K, per_lipid)
        out, pos = [], 0
        for l in self.tail_lens:
            out.append(flat[:, pos:pos + l - 2].copy()); pos += l - 2
        if opt.n_shells_smoothing > 0:              # smooth_curvature (lib.rs:584-621)
            st["mean_curv"], st["gauss_curv"] = api.membrane_smooth_curvature(st["valid"], patch_off, st["nvert"], st["neib_ids"],
                                                                              opt.n_shells_smoothing, st["mean_curv"], st["gauss_curv"])
        res = dict(head=head, mid=mid, tail=tail, patch_off=patch_off, patch_ids=patch_ids, normals=st["normals"],
                   initial_normals=normals, order=out, valid=st["valid"].copy(), smoothed_head=st["head_markers"])
        for k in ("quad_coefs", "mean_curv", "gauss_curv", "princ_curvs", "princ_dirs", "area", "nvert", "neib_ids",
                  "voro_vertexes", "fitted_patch_points"):
            res[k] = st[k]
        if self.groups:                                                         # lib.rs:448-451
            d = (head - tail).astype(np.float32)
            thv = d / np.sqrt((d * d).sum(1, dtype=np.float32))[:, None]        # tail_head_vec (lib.rs:459-461)
            for g in self.groups.values():
                g.frame_update(res, self.species_of_lipid, thv)
        return res
