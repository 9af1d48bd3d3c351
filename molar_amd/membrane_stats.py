"""Per-group / per-species accumulators of molar_membrane (host bookkeeping over the per-lipid arrays the GPU
passes produce):

    Histogram1D       stats.rs:13-55     binning rule b = floor(n*(v-min)/(max-min)), in-range only; density norm
    MeanStd           stats.rs:296-324   sum, sum of squares, count -> mean, stddev (0 if x2/n <= mean^2)
    MeanStdVec        stats.rs:346-390   the same per vector component (no clamp: sqrt of a negative is NaN)
    SpeciesStats      stats.rs:121-236   area, tilt (deg), curvatures, order per tail, neighbour counts per species
    LipidGroup        lipid_group.rs:9-37  frame_update over the group's valid lipids
    save_*            stats.rs:63-118, 238-293  the reference's text formats

Arithmetic is float32 like the reference's `Float`; sums over the lipids of a frame are vectorised (the reference adds
them one by one, so the last bits of the f32 accumulators can differ)."""
from __future__ import annotations

import os

import numpy as np

F = np.float32


class Histogram1D:
    def __init__(self, vmin, vmax, n_bins):
        self.min, self.max = F(vmin), F(vmax)
        self.bins = np.zeros(n_bins, F)

    def add_many(self, vals):
        v = np.asarray(vals, F).reshape(-1)
        n = len(self.bins)
        with np.errstate(invalid="ignore", over="ignore"):
            fb = np.floor(F(n) * (v - self.min) / (self.max - self.min))
        fb = np.where(np.isnan(fb), F(0), fb)                       # NaN as isize == 0
        ok = (fb >= 0) & (fb < n)
        np.add.at(self.bins, fb[ok].astype(np.int64), F(1))

    def add_one(self, val):
        self.add_many([val])

    def normalize_density(self):
        d = (self.max - self.min) / F(len(self.bins))
        self.bins = (self.bins / (self.bins.sum(dtype=F) * d)).astype(F)

    def save_to_file(self, fname):
        d = (self.max - self.min) / F(len(self.bins))
        with open(fname, "w") as f:
            for i, val in enumerate(self.bins):
                f.write(f"{float(self.min + F(i) * d + F(0.5) * d)} {float(val)}\n")


class MeanStd:
    def __init__(self):
        self.x = F(0); self.x2 = F(0); self.n = F(0)

    def add(self, val):
        val = F(val)
        self.x = F(self.x + val); self.x2 = F(self.x2 + val * val); self.n = F(self.n + F(1))

    def add_many(self, vals):
        v = np.asarray(vals, F).reshape(-1)
        self.x = F(self.x + v.sum(dtype=F)); self.x2 = F(self.x2 + (v * v).sum(dtype=F)); self.n = F(self.n + F(len(v)))

    def compute(self):
        if self.n == 0:
            raise ValueError("no values accumulated in MeanStd")
        mean = F(self.x / self.n); x2n = F(self.x2 / self.n); m2 = F(mean * mean)
        return mean, (F(np.sqrt(F(x2n - m2))) if x2n > m2 else F(0))


class MeanStdVec:
    def __init__(self, size):
        self.x = np.zeros(size, F); self.x2 = np.zeros(size, F); self.n = F(0)

    def add_many(self, rows):
        r = np.asarray(rows, F).reshape(-1, len(self.x))
        self.x = (self.x + r.sum(0, dtype=F)).astype(F); self.x2 = (self.x2 + (r * r).sum(0, dtype=F)).astype(F)
        self.n = F(self.n + F(len(r)))

    def add(self, val):
        val = np.asarray(val, F)
        if len(val) != len(self.x):
            raise ValueError(f"incompatible vector size in MeanStdVec::add: {len(val)} provided, {len(self.x)} expected")
        self.add_many(val[None, :])

    def compute(self):
        if self.n == 0:
            raise ValueError("no values accumulated in MeanStd")
        mean = (self.x / self.n).astype(F)
        with np.errstate(invalid="ignore"):
            std = np.sqrt((self.x2 / self.n).astype(F) - mean * mean).astype(F)
        return mean, std


class SpeciesStats:
    def __init__(self, tail_lens, all_species):
        self.num_lip = MeanStd(); self.area = MeanStd(); self.tilt = MeanStd(); self.num_neib = MeanStd()
        self.mean_curv = MeanStd(); self.gauss_curv = MeanStd()
        self.order = [MeanStdVec(l - 2) for l in tail_lens]          # bond_orders.len() - 1 = n_carbons - 2
        self.neib_species = {sp: MeanStd() for sp in all_species}


class LipidGroup:
    """lipid_group.rs + GroupProperties: ids of the group, one SpeciesStats per species present in the membrane."""

    def __init__(self, species_names, species_tail_lens):
        self.lipid_ids = np.zeros(0, np.int64)
        self.names = list(species_names)
        self.per_species = {sp: SpeciesStats(species_tail_lens[sp], self.names) for sp in self.names}

    def frame_update(self, res, species_of_lipid, tail_head_vec):
        """res: the dict Membrane.compute returns; species_of_lipid: int array (index into names) per lipid."""
        valid = res["valid"].astype(bool)
        ids = self.lipid_ids[valid[self.lipid_ids]]
        K = len(valid)
        slot0 = res["patch_off"][:-1].astype(np.int64) + 4 * np.arange(K)
        for si, sp in enumerate(self.names):
            st = self.per_species[sp]
            sel = ids[species_of_lipid[ids] == si]
            n_cur = len(sel)
            if n_cur:
                st.area.add_many(res["area"][sel])
                nrm = res["normals"][sel].astype(F); thv = tail_head_vec[sel].astype(F)
                # nalgebra Vector::angle, to_degrees (stats.rs:170-176)
                n1 = np.sqrt((nrm * nrm).sum(1, dtype=F)); n2 = np.sqrt((thv * thv).sum(1, dtype=F))
                with np.errstate(invalid="ignore", divide="ignore"):
                    c = np.clip((nrm * thv).sum(1, dtype=F) / (n1 * n2), F(-1), F(1))
                ang = np.where((n1 == 0) | (n2 == 0), F(0), np.arccos(c)).astype(F)
                st.tilt.add_many(np.degrees(ang).astype(F))
                st.mean_curv.add_many(res["mean_curv"][sel]); st.gauss_curv.add_many(res["gauss_curv"][sel])
                for t, acc in enumerate(st.order):
                    acc.add_many(res["order"][t][sel])
                nv = res["nvert"][sel].astype(np.int64)
                st.num_neib.add_many(nv.astype(F))
                # species of every Voronoi neighbour of the selected lipids
                pos = np.arange(nv.sum()) - np.repeat(np.concatenate([[0], np.cumsum(nv)[:-1]]), nv) + np.repeat(slot0[sel], nv)
                nsp = species_of_lipid[res["neib_ids"][pos].astype(np.int64)]
                counts = np.bincount(nsp, minlength=len(self.names))
            else:
                counts = np.zeros(len(self.names), np.int64)
            st.num_lip.add(n_cur)                                     # finish_frame_update (stats.rs:228-236)
            for k, other in enumerate(self.names):
                with np.errstate(invalid="ignore", divide="ignore"):
                    st.neib_species[other].add(F(counts[k]) / F(n_cur))   # 0/0 = NaN when the species is absent, as there

    def save(self, out_dir, gr_name):
        """gr_<name>_stats.dat, gr_<name>_neib_stats.dat, gr_<name>_order_<species>.dat (stats.rs:63-118, 238-293)."""
        os.makedirs(out_dir, exist_ok=True)
        s = "#species\tnum\tnum_std\tarea\tarea_std\ttilt\ttilt_std\tmean_curv\tmean_curv_std\tgauss_curv\tgauss_curv_std\n"
        for sp, st in self.per_species.items():
            vals = []
            for acc in (st.num_lip, st.area, st.tilt, st.mean_curv, st.gauss_curv):
                vals.extend(acc.compute())
            s += sp + "\t" + "\t".join(f"{float(v):>8.3f}" for v in vals) + "\n"
        open(os.path.join(out_dir, f"gr_{gr_name}_stats.dat"), "w").write(s)
        s = ""
        for sp, st in self.per_species.items():
            m, sd = st.num_neib.compute()
            s += f"{sp}:\t\t{float(m):>8.3f}\t{float(sd):>8.3f}\n"
            for nsp, acc in st.neib_species.items():
                m, sd = acc.compute()
                s += f"\t{nsp}\t{float(m):>8.3f}\t{float(sd):>8.3f}\n"
            s += "\n"
        open(os.path.join(out_dir, f"gr_{gr_name}_neib_stats.dat"), "w").write(s)
        for sp, st in self.per_species.items():
            means = [acc.compute()[0] for acc in st.order]
            max_len = max(len(m) for m in means)
            s = "# time\taver\t" + "\t".join(f"tail{t + 1}" for t in range(len(means))) + "\n"
            for i in range(max_len):
                have = [m[i] for m in means if i < len(m)]
                ave = F(sum(have, F(0))) / F(len(have))
                cols = [f"{float(m[i]):.3f}" if i < len(m) else "--" for m in means]
                s += f"{float(i + 1):.3f}\t{float(ave):.3f}\t" + "\t".join(cols) + "\n"
            open(os.path.join(out_dir, f"gr_{gr_name}_order_{sp}.dat"), "w").write(s)
