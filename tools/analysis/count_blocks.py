"""How many 32x32 blocks does the matrix-core count pass of the headline frame evaluate under different skipping
schemes?  CPU model on the synthetic 1M-atom frame (a sample of first cells, all 13 neighbour entries each).
  S0  two row blocks x all tiles (round 3)
  S1  rows pruned against the second cell's box and compacted (round 4)
  S2  S1 + second cell in Morton order, 32-atom tiles skipped when their box is farther than the cutoff from the FIRST CELL's box
  S3  S1 + tiles skipped against the box of the slot's LIVE rows
  S4  S3 with the live rows sorted by distance to the second cell's box and the test per (row block, tile)
usage: python tools/analysis/count_blocks.py [ncells_sampled]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from molar_amd import synth

N = 1_000_000
rc = 1.2
box = synth.box_a(N)
pos = synth.frame(N, box, 1)
inv = np.linalg.inv(box.astype(np.float64))
rel = pos.astype(np.float64) @ inv.T
rel -= np.floor(rel)
ext = box.sum(axis=1)
dims = np.maximum(np.floor(ext / rc), 1).astype(int)
loc = np.minimum((rel * dims).astype(int), dims - 1)
cell = loc[:, 0] + loc[:, 1] * dims[0] + loc[:, 2] * dims[0] * dims[1]
wp = (rel @ box.astype(np.float64).T)
order = np.argsort(cell, kind="stable")
cs = np.searchsorted(cell[order], np.arange(dims.prod() + 1))
MASK = [(0,0,0,1,0,0),(0,0,0,0,1,0),(0,0,0,0,0,1),(0,0,0,1,1,0),(0,0,0,1,0,1),(0,0,0,0,1,1),(0,0,0,1,1,1),
        (1,0,0,0,1,0),(1,0,0,0,0,1),(0,1,0,0,0,1),(1,1,0,0,0,1),(1,0,1,0,1,0),(0,1,1,1,0,0)]
def morton(p, lo, hi):
    sc = np.where(hi > lo, 8.0 / (hi - lo), 0.0)
    q = np.clip(((p - lo) * sc).astype(int), 0, 7)
    key = np.zeros(len(p), int)
    for b in range(3):
        key |= ((q[:, 0] >> b) & 1) << (3 * b) | ((q[:, 1] >> b) & 1) << (3 * b + 1) | ((q[:, 2] >> b) & 1) << (3 * b + 2)
    return np.argsort(key, kind="stable")
def boxdist2(lo1, hi1, lo2, hi2):
    g = np.maximum(0, np.maximum(lo2 - hi1, lo1 - hi2))
    return (g * g).sum(-1)
rng = np.random.default_rng(0)
nsample = int(sys.argv[1]) if len(sys.argv) > 1 else 60
tot = np.zeros(5)
kinds = {"face": np.zeros(5), "edge": np.zeros(5), "corner": np.zeros(5)}
# interior cells only (no wrap): x,y,z < dims-1
cands = [(x, y, z) for x in range(dims[0] - 1) for y in range(dims[1] - 1) for z in range(dims[2] - 1)]
for ci in rng.choice(len(cands), nsample, replace=False):
    x, y, z = cands[ci]
    for m in MASK:
        c1 = (x + m[0]) + (y + m[1]) * dims[0] + (z + m[2]) * dims[0] * dims[1]
        c2 = (x + m[3]) + (y + m[4]) * dims[0] + (z + m[5]) * dims[0] * dims[1]
        A = wp[order[cs[c1]:cs[c1 + 1]]]; B = wp[order[cs[c2]:cs[c2 + 1]]]
        d = np.abs(np.array(m[:3]) - np.array(m[3:])).sum()
        kind = {1: "face", 2: "edge", 3: "corner"}[d]
        blo, bhi = B.min(0), B.max(0); alo, ahi = A.min(0), A.max(0)
        Bm = B[morton(B, blo, bhi)]
        nt = (len(B) + 31) // 32
        tl = np.array([Bm[t * 32:(t + 1) * 32].min(0) for t in range(nt)]); th = np.array([Bm[t * 32:(t + 1) * 32].max(0) for t in range(nt)])
        r = np.zeros(5)
        for i0 in range(0, len(A), 64):
            rows = A[i0:i0 + 64]
            r[0] += 2 * nt if len(rows) > 32 else nt
            g = np.maximum(0, np.maximum(blo - rows, rows - bhi)); dd = (g * g).sum(1)
            live = rows[dd <= rc * rc]; dl = dd[dd <= rc * rc]
            nl = len(live)
            if nl == 0: continue
            nrb = 2 if nl > 32 else 1
            r[1] += nrb * nt
            r[2] += nrb * (boxdist2(alo, ahi, tl, th) <= rc * rc).sum()
            r[3] += nrb * (boxdist2(live.min(0), live.max(0), tl, th) <= rc * rc).sum()
            ls = live[np.argsort(dl, kind="stable")]
            for b in range(nrb):
                rb = ls[b * 32:(b + 1) * 32]
                r[4] += (boxdist2(rb.min(0), rb.max(0), tl, th) <= rc * rc).sum()
        tot += r; kinds[kind] += r
print("blocks per first cell (13 neighbour entries), schemes S0..S4:", np.round(tot / nsample, 1))
for k, v in kinds.items(): print(f"  {k:7s}", np.round(v / nsample, 1))
print("relative to S0:", np.round(tot / tot[0], 3))
