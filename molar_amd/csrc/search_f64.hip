// search_f64.hip - the eight distance_search drivers for MolAR built with its `f64` feature (Float = f64:
// molar/src/aliases.rs:10-13, molar/Cargo.toml:56-60), correct before fast.
//
// The f32 search (search.hip, pair_kernels.hpp) is the tuned product path; an f64 build of MolAR flips `Float` for the
// whole crate including distance_search, so results there are decided by f64 arithmetic: which cell an atom near a cell
// face lands in, which pairs at the cutoff are hits, and the distances themselves.  This file restates the drivers once
// more with every operation in double:
//   * inputs are used where they lie if they are device memory (staged otherwise); grid dims on the host from the box or
//     from a zero-seeded bounding box reduced on the device (distance_search.rs:103-110, 602-646);
//   * populate / populate_pbc (drop rule, wrap, later-dimension quirk, :120-210) per selected atom on the device; the
//     reference's push order - in-box atoms of a cell before its wrapped atoms, each in selection order - is a STABLE radix
//     sort (rocPRIM) by the key 2 * cell + wrapped; cell starts by binary search; the 14-mask plan, its scan and one record
//     per 64-row slot by small kernels (:217-269).  Three small read-backs (vdW radii maximum / bounding box, number of
//     slots, number of results);
//   * one 64-lane wave per slot evaluates the exact f64 predicate - plain |p2 - p1|^2 or PeriodicBox::distance_squared
//     (periodic_box.rs:286-318) for entries across the periodic boundary - with the second cell's atoms resident in
//     registers (up to 256 of them; larger cells chunk by chunk from memory) and the rows handed round with v_readlane:
//     first to count, then, behind an exclusive scan of the slot counts, to write (i, j, sqrt(d2)) at its place of the
//     reference's output order (plan order, then i-major / j-minor, :949-953);
//   * records are 24-byte positions + 8-byte ids; results are (usize, usize, f64) columns or usize ids (within).
// 1M atoms, rc 1.0 nm, 2.1e8 results, frame and result resident: see tools/bench_search_f64.py, profiles/r05_search_f64.jsonl.
// tests/test_gpu_search_f64.py compares all eight drivers bit for bit with the f64 build of the CPU checker.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "boxmath64.hpp"
#include "common.hpp"
#include "hoststream.hpp"
#include "stages.hpp"

using namespace mh;

struct molar_hip_search64_state {
    int kind = 0;
    uint64_t dims[3] = {1, 1, 1};
    uint64_t total = 0;
    uint32_t nslots = 0;
    bool have = false;
    bool use_box = false;
    uint8_t pbc = 0;
    double cutoff = 0.0;
    double prune_limit2 = 0.0; // (cutoff + margin)^2: rows of wrapped entries farther than this from the second cell's image box are skipped
    DevBuf aabbA, aabbB;       // per cell: lo[3], hi[3] of its items
    bool approx = false;       // wrapped entries classified by the distance to the adjacent image (count and fill alike)
    double band_lo = 1.0, band_hi = 1.0;
    DevBuf posA, idA, vdwA, posB, idB, vdwB, slots, slot_cnt, slot_base, box, out_i, out_j, out_d;
    // device pipeline of the grid and the plan
    DevBuf in_xyz[2], in_idx[2], in_vdw[2];      // staging of host inputs
    DevBuf key_in, key_out, val_in, val_out, pos3, startA, startB, task_ns, task_first, cub_tmp, partial, flags;
};

namespace {

// MASK of distance_search.rs:39-60
__constant__ uint8_t MASKS64[14][6] = {
    {0, 0, 0, 0, 0, 0},
    {0, 0, 0, 1, 0, 0}, {0, 0, 0, 0, 1, 0}, {0, 0, 0, 0, 0, 1},
    {0, 0, 0, 1, 1, 0}, {0, 0, 0, 1, 0, 1}, {0, 0, 0, 0, 1, 1},
    {0, 0, 0, 1, 1, 1},
    {1, 0, 0, 0, 1, 0}, {1, 0, 0, 0, 0, 1}, {0, 1, 0, 0, 0, 1},
    {1, 1, 0, 0, 0, 1}, {1, 0, 1, 0, 1, 0}, {0, 1, 1, 1, 0, 0},
};

// Rust `as usize` / `as isize` on f64: saturating, NaN -> 0
MH64_HD uint64_t as_usize(double x) {
    if (!(x > 0.0)) return 0;
    if (x >= 18446744073709551615.0) return ~0ull;
    return (uint64_t)x;
}
MH64_HD int64_t as_isize(double x) {
    if (x != x) return 0;
    if (x >= 9223372036854775807.0) return INT64_MAX;
    if (x <= -9223372036854775808.0) return INT64_MIN;
    return (int64_t)x;
}

struct Slot64 {                    // what a wave needs for one 64-row slot
    uint32_t a0, rows, i0, b0, n2, flags;     // flags: wrap | tri << 8
    uint32_t cb, pad1;                         // cb: the second cell (its bounding box prunes rows)
};

// one set of the search as the caller gave it (device addresses)
struct SetIn {
    const double *xyz;
    const unsigned long long *idx;     // NULL: all atoms
    const double *vdw;                 // per SELECTED atom, or NULL
    uint64_t natoms;
    uint32_t nsel;
    int ids_local;
};
struct GridP {
    BoxD box;
    int use_box;
    uint32_t pbc;
    uint32_t dims[3];
    uint32_t ncells;
    double lower[3], upper[3];
};

// Grid::populate (:120-142) / populate_pbc (:144-210) for one selected atom: its cell, whether it was wrapped (the reference
// pushes the atoms found inside the box first, the wrapped ones after them, :180, :203-209) and the position stored with it.
// Sort key = 2 * cell + wrapped; dropped atoms get the key 2 * ncells.
__global__ void __launch_bounds__(256) assign64_kernel(SetIn S, GridP G, uint32_t *__restrict__ key, uint32_t *__restrict__ val,
                                                       double *__restrict__ pos3, int *__restrict__ err) {
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= S.nsel) return;
    val[k] = k;
    const uint64_t a = S.idx ? S.idx[k] : (uint64_t)k;
    const uint32_t dropped = 2u * G.ncells;
    if (a >= S.natoms) {
        *err = 1;
        key[k] = dropped;
        return;
    }
    const double p[3] = {S.xyz[3 * a], S.xyz[3 * a + 1], S.xyz[3 * a + 2]};
    double q[3] = {p[0], p[1], p[2]};
    uint64_t loc[3] = {0, 0, 0};
    uint32_t phase = 0;
    if (!G.use_box) {
        bool ok = true;
        for (int d = 0; d < 3; ++d) {
            const double dim_sz = G.upper[d] - G.lower[d];
            const int64_t nn = as_isize(floor((double)G.dims[d] * (p[d] - G.lower[d]) / dim_sz));    // :131
            if (nn < 0 || nn >= (int64_t)G.dims[d]) { ok = false; break; }
            loc[d] = (uint64_t)nn;
        }
        if (!ok) {
            key[k] = dropped;
            return;
        }
    } else {
        const D3 rel = mat_vec(G.box.inv, D3{p[0], p[1], p[2]});                             // :156
        double r[3] = {rel.x, rel.y, rel.z};
        bool correct = true, drop = false;
        for (int d = 0; d < 3; ++d)                                                          // :161-171
            if (r[d] < 0.0 || r[d] >= 1.0) {
                if (!((G.pbc >> d) & 1u)) { drop = true; break; }
                correct = false;
                break;
            }
        if (drop) {
            key[k] = dropped;
            return;
        }
        if (!correct) {                                                                      // :181-199
            for (int d = 0; d < 3; ++d)
                if ((G.pbc >> d) & 1u) {
                    r[d] = r[d] - trunc(r[d]);                                               // fract()
                    if (r[d] < 0.0) r[d] = 1.0 + r[d];
                }
            const D3 w = mat_vec(G.box.m, D3{r[0], r[1], r[2]});                              // :196
            q[0] = w.x; q[1] = w.y; q[2] = w.z;
            phase = 1;
        }
        for (int d = 0; d < 3; ++d) {
            uint64_t l = as_usize(floor(r[d] * (double)G.dims[d]));                          // :175, :191
            if (l > G.dims[d] - 1u) l = G.dims[d] - 1u;
            loc[d] = l;
        }
    }
    const uint32_t cell = (uint32_t)(loc[0] + loc[1] * G.dims[0] + loc[2] * (uint64_t)G.dims[0] * G.dims[1]);
    key[k] = 2u * cell + phase;
    pos3[3 * (size_t)k] = q[0];
    pos3[3 * (size_t)k + 1] = q[1];
    pos3[3 * (size_t)k + 2] = q[2];
}

// the grid's items in the reference's push order (the radix sort is stable: equal keys keep the selection's order)
__global__ void __launch_bounds__(256) gather64_kernel(SetIn S, uint32_t n, uint32_t ncells, const uint32_t *__restrict__ key_sorted,
                                                       const uint32_t *__restrict__ val_sorted, const double *__restrict__ pos3,
                                                       double *__restrict__ pos, unsigned long long *__restrict__ id,
                                                       double *__restrict__ vdw) {
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s >= n || key_sorted[s] >= 2u * ncells) return;
    const uint32_t k = val_sorted[s];
    pos[3 * (size_t)s] = pos3[3 * (size_t)k];
    pos[3 * (size_t)s + 1] = pos3[3 * (size_t)k + 1];
    pos[3 * (size_t)s + 2] = pos3[3 * (size_t)k + 2];
    id[s] = S.ids_local ? (unsigned long long)k : (S.idx ? S.idx[k] : (unsigned long long)k);
    if (vdw) vdw[s] = S.vdw[k];
}

// start[c] = first sorted item with key >= 2 c, c = 0 .. ncells (start[ncells] = items kept)
__global__ void __launch_bounds__(256) cellstart64_kernel(const uint32_t *__restrict__ key_sorted, uint32_t n, uint32_t ncells,
                                                          uint32_t *__restrict__ start) {
    const uint32_t c = blockIdx.x * 256u + threadIdx.x;
    if (c > ncells) return;
    const uint32_t want = 2u * c;
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (key_sorted[mid] < want) lo = mid + 1;
        else hi = mid;
    }
    start[c] = lo;
}

// bounding box of every cell's items (row pruning of the pair kernels): one wave per cell
__global__ void __launch_bounds__(256) aabb64_kernel(const uint32_t *__restrict__ start, uint32_t ncells, const double *__restrict__ pos,
                                                     double *__restrict__ aabb) {
    const uint32_t c = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (c >= ncells) return;
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t k = start[c] + lane; k < start[c + 1]; k += 64u)
        for (int d = 0; d < 3; ++d) {
            const double v = pos[3 * (size_t)k + d];
            lo[d] = fmin(lo[d], v);
            hi[d] = fmax(hi[d], v);
        }
    for (int d = 0; d < 3; ++d)
        for (int off = 32; off > 0; off >>= 1) {
            lo[d] = fmin(lo[d], __shfl_xor(lo[d], off, 64));
            hi[d] = fmax(hi[d], __shfl_xor(hi[d], off, 64));
        }
    if (lane == 0)
        for (int d = 0; d < 3; ++d) { aabb[6 * (size_t)c + d] = lo[d]; aabb[6 * (size_t)c + 3 + d] = hi[d]; }
}

// compute_min_max (:602-616, seeded with zeros) over the selected atoms: per-workgroup partials {lo[3], hi[3]}
__global__ void __launch_bounds__(256) minmax64_kernel(SetIn S, double *__restrict__ partial, int *__restrict__ err) {
    double lo[3] = {0.0, 0.0, 0.0}, hi[3] = {0.0, 0.0, 0.0};
    for (uint32_t k = blockIdx.x * 256u + threadIdx.x; k < S.nsel; k += gridDim.x * 256u) {
        const uint64_t a = S.idx ? S.idx[k] : (uint64_t)k;
        if (a >= S.natoms) { *err = 1; continue; }
        for (int d = 0; d < 3; ++d) {
            const double v = S.xyz[3 * a + d];
            if (v < lo[d]) lo[d] = v;
            if (v > hi[d]) hi[d] = v;
        }
    }
    __shared__ double sh[4][6];
    for (int d = 0; d < 3; ++d)
        for (int off = 32; off > 0; off >>= 1) {
            const double l2 = __shfl_xor(lo[d], off, 64), h2 = __shfl_xor(hi[d], off, 64);
            if (l2 < lo[d]) lo[d] = l2;
            if (h2 > hi[d]) hi[d] = h2;
        }
    if ((threadIdx.x & 63u) == 0u)
        for (int d = 0; d < 3; ++d) { sh[threadIdx.x >> 6][d] = lo[d]; sh[threadIdx.x >> 6][3 + d] = hi[d]; }
    __syncthreads();
    if (threadIdx.x < 6u) {
        double v = sh[0][threadIdx.x];
        for (int w = 1; w < 4; ++w) {
            const double o = sh[w][threadIdx.x];
            if (threadIdx.x < 3u ? o < v : o > v) v = o;
        }
        partial[6 * blockIdx.x + threadIdx.x] = v;
    }
}

// Iterator::reduce(Float::max) over the radii (:781-783): NaN-ignoring max; per-workgroup partials (NaN where a workgroup saw
// only NaNs or nothing)
__global__ void __launch_bounds__(256) fmax64_kernel(const double *__restrict__ v, uint32_t n, double *__restrict__ partial) {
    double m = __builtin_nan("");
    for (uint32_t k = blockIdx.x * 256u + threadIdx.x; k < n; k += gridDim.x * 256u) m = fmax(m, v[k]);
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off, 64));
    __shared__ double sh[4];
    if ((threadIdx.x & 63u) == 0u) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
}

// search_plan (:217-269): element t of the plan in the reference's loop order - x outer, z inner, the 14 masks, and for two
// sets (pair.0, pair.1) then (pair.1, pair.0) - as (first cell, second cell, wrap | tri << 8), or first == 0xFFFFFFFF
struct Entry64 {
    uint32_t ca, cb, flags;
};
__device__ __forceinline__ Entry64 plan_entry(const GridP &G, uint64_t t, uint32_t mult) {
    const uint32_t half = (uint32_t)(t % mult);
    const uint64_t tm = t / mult;
    const uint32_t m = (uint32_t)(tm % 14u);
    const uint64_t cs = tm / 14u;
    const uint32_t z = (uint32_t)(cs % G.dims[2]), y = (uint32_t)((cs / G.dims[2]) % G.dims[1]), x = (uint32_t)(cs / ((uint64_t)G.dims[2] * G.dims[1]));
    uint32_t cc[2][3] = {{x + MASKS64[m][0], y + MASKS64[m][1], z + MASKS64[m][2]}, {x + MASKS64[m][3], y + MASKS64[m][4], z + MASKS64[m][5]}};
    uint32_t wrap = 0, wrapped[2] = {0u, 0u};             // dims in which the first / the second listed cell went round
    for (int i = 0; i < 2; ++i)
        for (int d = 0; d < 3; ++d)
            if (cc[i][d] == G.dims[d]) {
                if ((G.pbc >> d) & 1u) { cc[i][d] = 0; wrap |= 1u << d; wrapped[i] |= 1u << d; }
                else return Entry64{0xFFFFFFFFu, 0u, 0u};                                    // :241-244
            }
    const uint32_t i1 = cc[0][0] + cc[0][1] * G.dims[0] + cc[0][2] * G.dims[0] * G.dims[1];
    const uint32_t i2 = cc[1][0] + cc[1][1] * G.dims[0] + cc[1][2] * G.dims[0] * G.dims[1];
    // bits 12-14: the dims in which the entry's SECOND (column) cell is the one that went round
    if (mult == 1u) return Entry64{i1, i2, wrap | (i1 == i2 ? 0x100u : 0u) | (wrapped[1] << 12)};   // :432-517
    // (pair.0, pair.1) then (pair.1, pair.0): rows are always atoms of the FIRST set (:686-693)
    return half ? Entry64{i2, i1, wrap | (wrapped[0] << 12)} : Entry64{i1, i2, wrap | (wrapped[1] << 12)};
}

__global__ void __launch_bounds__(256) plan64_kernel(GridP G, uint64_t ntasks, uint32_t mult, const uint32_t *__restrict__ startA,
                                                     const uint32_t *__restrict__ startB, uint32_t *__restrict__ task_ns) {
    const uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (t > ntasks) return;
    uint32_t ns = 0;
    if (t < ntasks) {
        const Entry64 e = plan_entry(G, t, mult);
        if (e.ca != 0xFFFFFFFFu) {
            const uint32_t n1 = startA[e.ca + 1] - startA[e.ca], n2 = startB[e.cb + 1] - startB[e.cb];
            if (n1 && n2) ns = (n1 + 63u) / 64u;
        }
    }
    task_ns[t] = ns;                                   // task_ns[ntasks] = 0: its scan is the number of slots
}

__global__ void __launch_bounds__(256) slots64_kernel(GridP G, uint64_t ntasks, uint32_t mult, const uint32_t *__restrict__ startA,
                                                      const uint32_t *__restrict__ startB, const uint32_t *__restrict__ task_first,
                                                      Slot64 *__restrict__ slots) {
    const uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (t >= ntasks) return;
    const uint32_t first = task_first[t], ns = task_first[t + 1] - first;
    if (!ns) return;
    const Entry64 e = plan_entry(G, t, mult);
    const uint32_t n1 = startA[e.ca + 1] - startA[e.ca], n2 = startB[e.cb + 1] - startB[e.cb];
    for (uint32_t q = 0; q < ns; ++q) {
        Slot64 s{};
        s.a0 = startA[e.ca];
        s.i0 = q * 64u;
        s.rows = n1 - s.i0 < 64u ? n1 - s.i0 : 64u;
        s.b0 = startB[e.cb];
        s.n2 = n2;
        s.flags = e.flags;
        s.cb = e.cb;
        slots[first + q] = s;
    }
}

struct Params64 {
    const double *posA, *posB, *vdwA, *vdwB;
    const unsigned long long *idA, *idB;
    const BoxD *box;
    const Slot64 *slots;
    uint32_t nslots;
    int kind;
    int use_box;
    double cutoff2;
    // entries across the periodic boundary: classify with the plain distance to the second cell's adjacent image and decide
    // with PeriodicBox::distance_squared only inside [band_lo, band_hi] * cutoff^2 (see molar_hip_search_count_f64)
    int approx_wrapped;
    double band_lo, band_hi;
    const double *aabbB;      // per cell of the second grid: lo[3], hi[3]
    double prune_limit2;
};

// squared distance of one candidate: plain (:488) or PeriodicBox::distance_squared for an entry that wrapped (:485-486)
__device__ __forceinline__ double pair_d2(const Params64 &P, uint32_t wrap, D3 a, D3 b) {
    const D3 v = b - a;                                                                      // p2 - p1
    if (P.use_box && wrap) return norm2(shortest_vector(*P.box, v, wrap));
    return norm2(v);
}

__device__ __forceinline__ double lane_bcast(double v, uint32_t r) {      // lane r's value in every lane (r uniform)
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), (int)r), hi = __builtin_amdgcn_readlane(__double2hiint(v), (int)r);
    return __hiloint2double(hi, lo);
}

// One 64-row slot whose second cell fits NCH chunks of 64 atoms: the second cell's atoms stay in registers (one per lane and
// chunk), the slot's rows are loaded once (one per lane) and handed round with v_readlane - no memory access in the row loop.
// Same predicate, same order of operations and of results as the generic loop below.
// Output queue of a wave in LDS (fill pass): hits are pushed in order as (i, j, d2) and written 64 at a time - one 512-byte
// block per column, naturally aligned after the slot's first flush (which only goes up to the next 64-entry boundary of the
// output) - with the square roots taken for 64 lanes at once.
struct Fifo64 {
    unsigned long long *fi, *fj;
    double *fd;
    uint32_t head, tail, quota;
};
constexpr uint32_t FIFO64_CAP = 128;

__device__ __forceinline__ void fifo64_flush(Fifo64 &F, uint32_t count, uint32_t lane, unsigned long long off,
                                             unsigned long long *__restrict__ out_i, unsigned long long *__restrict__ out_j,
                                             double *__restrict__ out_d) {
    if (lane < count) {
        const uint32_t sl = (F.head + lane) & (FIFO64_CAP - 1u);
        const unsigned long long at = off + F.head + lane;
        __builtin_nontemporal_store(F.fi[sl], &out_i[at]);
        __builtin_nontemporal_store(F.fj[sl], &out_j[at]);
        __builtin_nontemporal_store(sqrt(F.fd[sl]), &out_d[at]);                  // d2.sqrt() (:448)
    }
    F.head += count;
}

template <bool FILL, int KIND, int NCH>
__device__ __forceinline__ uint32_t run64(const Params64 &P, const Slot64 &S, uint32_t lane, unsigned long long off,
                                          unsigned long long *__restrict__ out_i, unsigned long long *__restrict__ out_j,
                                          double *__restrict__ out_d, Fifo64 F) {
    const uint32_t wrap = S.flags & 7u;
    const bool tri = (S.flags >> 8) & 1u;
    constexpr bool VDW = KIND == MOLAR_HIP_SEARCH_DOUBLE_VDW, WITHIN = KIND == MOLAR_HIP_SEARCH_WITHIN;
    double bx[NCH], by[NCH], bz[NCH], bv[NCH];
    unsigned long long bid[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const uint32_t j = (uint32_t)k * 64u + lane;
        bx[k] = by[k] = bz[k] = bv[k] = 0.0;
        bid[k] = 0ull;
        if (j < S.n2) {
            const size_t rb = (size_t)S.b0 + j;
            bx[k] = P.posB[3 * rb]; by[k] = P.posB[3 * rb + 1]; bz[k] = P.posB[3 * rb + 2];
            if (VDW) bv[k] = P.vdwB[rb];
            if (FILL && !WITHIN) bid[k] = P.idB[rb];
        }
    }
    F.head = F.tail = 0u;
    F.quota = 64u - ((uint32_t)off & 63u);
    // b + S is the image of the second cell next to the first one (second cell went round: + box vector, first cell: -)
    const bool approx = P.approx_wrapped && P.use_box && wrap != 0u && !(P.box->nshift != 0 && wrap == MOLAR_HIP_PBC_FULL);
    double cx[NCH], cy[NCH], cz[NCH];
    double Sx = 0.0, Sy = 0.0, Sz = 0.0;
    if (approx) {
        for (int d = 0; d < 3; ++d) {
            if (!((wrap >> d) & 1u)) continue;
            const double sgn = ((S.flags >> (12 + d)) & 1u) ? 1.0 : -1.0;
            Sx += sgn * P.box->m[3 * d];
            Sy += sgn * P.box->m[3 * d + 1];
            Sz += sgn * P.box->m[3 * d + 2];
        }
#pragma unroll
        for (int k = 0; k < NCH; ++k) { cx[k] = bx[k] + Sx; cy[k] = by[k] + Sy; cz[k] = bz[k] + Sz; }
    }
    double ax = 0.0, ay = 0.0, az = 0.0, av = 0.0;
    unsigned long long aid = 0ull;
    if (lane < S.rows) {
        const size_t ra = (size_t)S.a0 + S.i0 + lane;
        ax = P.posA[3 * ra]; ay = P.posA[3 * ra + 1]; az = P.posA[3 * ra + 2];
        if (VDW) av = P.vdwA[ra];
        if (FILL) aid = P.idA[ra];
    }
    // Rows that cannot have a hit.  Every position of the second cell lies inside the cell's bounding box [lo, hi], and each
    // f64 operation of d2 = ((dx*dx)+(dy*dy))+(dz*dz) is monotone in |dx|, |dy|, |dz|: the same expression on the distances to
    // the box is a lower bound of every d2 of the row IN f64 ARITHMETIC (vdW: against the largest pair cutoff).  Entries
    // classified by the adjacent image: against the image box with the margin of the band; other wrapped entries: no pruning.
    unsigned long long live;
    {
        bool need = lane < S.rows;
        if (!tri && (wrap == 0u || !P.use_box || approx)) {
            const double *bb = P.aabbB + 6 * (size_t)S.cb;
            double px = ax, py = ay, pz = az, lim = P.cutoff2;
            if (approx) {
                px = ax - Sx;
                py = ay - Sy;
                pz = az - Sz;
                lim = P.prune_limit2;
            }
            const double ex = fmax(fmax(bb[0] - px, px - bb[3]), 0.0), ey = fmax(fmax(bb[1] - py, py - bb[4]), 0.0),
                         ez = fmax(fmax(bb[2] - pz, pz - bb[5]), 0.0);
            need = need && !((ex * ex + ey * ey) + ez * ez > lim);
        }
        live = __builtin_amdgcn_ballot_w64(need);
    }
    uint32_t acc = 0, total = 0;
    while (live) {
        const uint32_t r = (uint32_t)__builtin_ctzll(live);
        live &= live - 1ull;
        const D3 a = D3{lane_bcast(ax, r), lane_bcast(ay, r), lane_bcast(az, r)};
        const double vdw_a = VDW ? lane_bcast(av, r) : 0.0;
        const uint32_t i = S.i0 + r;
        unsigned long long idr = 0ull;
        if (FILL && !WITHIN)
            idr = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(aid >> 32), (int)r) << 32) |
                  (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)aid, (int)r);
        bool any = false;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            if (tri && (uint32_t)k * 64u + 63u <= i) continue;                  // the whole chunk has j <= i (:443)
            const uint32_t j = (uint32_t)k * 64u + lane;
            bool hit = j < S.n2 && !(tri && j <= i);
            double cut2 = P.cutoff2;
            if (VDW) {
                const double cut = (vdw_a + bv[k]) + 2.220446049250313e-16;    // :392, :423
                cut2 = cut * cut;
            }
            double d2;
            if (approx) {
                const double ex = cx[k] - a.x, ey = cy[k] - a.y, ez = cz[k] - a.z;
                const double q2 = (ex * ex + ey * ey) + ez * ez;
                const bool sure = q2 < cut2 * P.band_lo, maybe = hit && q2 <= cut2 * P.band_hi;
                d2 = q2;
                // hits carry the reference's own distance; candidates inside the band are decided by it
                if (__builtin_amdgcn_ballot_w64(maybe && (FILL || !sure))) {
                    if (maybe && (FILL || !sure)) d2 = pair_d2(P, wrap, a, D3{bx[k], by[k], bz[k]});
                }
                hit = maybe && (sure || d2 <= cut2);
            } else {
                d2 = pair_d2(P, wrap, a, D3{bx[k], by[k], bz[k]});
                hit = hit && d2 <= cut2;
            }
            if (WITHIN) {
                any = any || hit;
            } else if (!FILL) {
                acc += hit ? 1u : 0u;
            } else {
                const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
                if (!m) continue;
                if (hit) {
                    const uint32_t sl = (F.tail + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))) & (FIFO64_CAP - 1u);
                    F.fi[sl] = idr;
                    F.fj[sl] = bid[k];
                    F.fd[sl] = d2;
                }
                F.tail += (uint32_t)__popcll(m);
                if (F.tail - F.head >= 64u) {
                    __builtin_amdgcn_wave_barrier();
                    do {
                        fifo64_flush(F, F.quota, lane, off, out_i, out_j, out_d);
                        F.quota = 64u;
                    } while (F.tail - F.head >= 64u);
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        if (WITHIN && __builtin_amdgcn_ballot_w64(any)) {                       // the row's first hit only (:287-290)
            if (FILL && lane == 0) out_i[off] = P.idA[(size_t)S.a0 + S.i0 + r];
            off += 1;
            total += 1;
        }
    }
    if (!WITHIN && !FILL) {
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        total = acc;
    }
    if (FILL && !WITHIN && F.tail != F.head) {
        __builtin_amdgcn_wave_barrier();
        fifo64_flush(F, F.tail - F.head, lane, off, out_i, out_j, out_d);
    }
    return total;
}

template <bool FILL, int KIND>
__global__ void __launch_bounds__(64) pair64_kernel(Params64 P, uint32_t *__restrict__ slot_cnt,
                                                    const unsigned long long *__restrict__ slot_base,
                                                    unsigned long long *__restrict__ out_i, unsigned long long *__restrict__ out_j,
                                                    double *__restrict__ out_d) {
    const uint32_t s = blockIdx.y * gridDim.x + blockIdx.x;
    if (s >= P.nslots) return;
    const Slot64 S = P.slots[s];
    const uint32_t lane = threadIdx.x;
    const uint32_t wrap = S.flags & 7u;
    const bool tri = (S.flags >> 8) & 1u;
    unsigned long long off = FILL ? slot_base[s] : 0ull;
    const uint32_t nch = (S.n2 + 63u) >> 6;
    constexpr uint32_t QN = (FILL && KIND != MOLAR_HIP_SEARCH_WITHIN) ? FIFO64_CAP : 1u;
    __shared__ unsigned long long q_i[QN], q_j[QN];
    __shared__ double q_d[QN];
    if (nch <= 4u) {
        uint32_t t;
        Fifo64 F{q_i, q_j, q_d, 0u, 0u, 64u};
        switch (nch) {
            case 1: t = run64<FILL, KIND, 1>(P, S, lane, off, out_i, out_j, out_d, F); break;
            case 2: t = run64<FILL, KIND, 2>(P, S, lane, off, out_i, out_j, out_d, F); break;
            case 3: t = run64<FILL, KIND, 3>(P, S, lane, off, out_i, out_j, out_d, F); break;
            default: t = run64<FILL, KIND, 4>(P, S, lane, off, out_i, out_j, out_d, F); break;
        }
        if (!FILL && lane == 0) slot_cnt[s] = t;
        return;
    }
    // second cells of more than 256 atoms: chunk by chunk from memory
    // rows are atoms of the first set's grid, columns atoms of the second set's (the same grid for SINGLE)
    const double *pa = P.posA, *pb = P.posB;
    const unsigned long long *ia = P.idA, *ib = P.idB;
    const double *va = P.vdwA, *vb = P.vdwB;
    uint32_t total = 0;
    for (uint32_t r = 0; r < S.rows; ++r) {
        const uint32_t ra = S.a0 + S.i0 + r;
        const D3 a = D3{pa[3 * ra], pa[3 * ra + 1], pa[3 * ra + 2]};
        const double vdw_a = (KIND == MOLAR_HIP_SEARCH_DOUBLE_VDW) ? va[ra] : 0.0;
        bool found = false;                                       // WITHIN: first hit of the row only (:287-290)
        for (uint32_t j0 = 0; j0 < S.n2 && !found; j0 += 64u) {
            const uint32_t j = j0 + lane;
            bool hit = false;
            double d2 = 0.0;
            if (j < S.n2 && !(tri && j <= S.i0 + r)) {            // same cell: j in i+1..n (:443)
                const uint32_t rb = S.b0 + j;
                d2 = pair_d2(P, wrap, a, D3{pb[3 * rb], pb[3 * rb + 1], pb[3 * rb + 2]});
                if (KIND == MOLAR_HIP_SEARCH_DOUBLE_VDW) {
                    const double cut = (vdw_a + vb[rb]) + 2.220446049250313e-16;        // :392, :423
                    hit = d2 <= cut * cut;
                } else {
                    hit = d2 <= P.cutoff2;
                }
            }
            const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
            if (!m) continue;
            if (KIND == MOLAR_HIP_SEARCH_WITHIN) {
                if (FILL && lane == 0) out_i[off] = ia[ra];
                off += 1;
                total += 1;
                found = true;
                continue;
            }
            const uint32_t cnt = (uint32_t)__popcll(m);
            if (FILL && hit) {
                const unsigned long long at = off + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                out_i[at] = ia[ra];
                out_j[at] = ib[S.b0 + j];
                out_d[at] = sqrt(d2);                             // d2.sqrt() (:448)
            }
            off += cnt;
            total += cnt;
        }
    }
    if (!FILL && lane == 0) slot_cnt[s] = total;
}

template <bool FILL>
void launch_pair64(int kind, dim3 grid, hipStream_t stream, const Params64 &P, uint32_t *slot_cnt, const unsigned long long *slot_base,
                   unsigned long long *out_i, unsigned long long *out_j, double *out_d) {
    switch (kind) {
        case MOLAR_HIP_SEARCH_SINGLE: hipLaunchKernelGGL((pair64_kernel<FILL, MOLAR_HIP_SEARCH_SINGLE>), grid, dim3(64), 0, stream, P, slot_cnt, slot_base, out_i, out_j, out_d); break;
        case MOLAR_HIP_SEARCH_DOUBLE: hipLaunchKernelGGL((pair64_kernel<FILL, MOLAR_HIP_SEARCH_DOUBLE>), grid, dim3(64), 0, stream, P, slot_cnt, slot_base, out_i, out_j, out_d); break;
        case MOLAR_HIP_SEARCH_WITHIN: hipLaunchKernelGGL((pair64_kernel<FILL, MOLAR_HIP_SEARCH_WITHIN>), grid, dim3(64), 0, stream, P, slot_cnt, slot_base, out_i, out_j, out_d); break;
        default: hipLaunchKernelGGL((pair64_kernel<FILL, MOLAR_HIP_SEARCH_DOUBLE_VDW>), grid, dim3(64), 0, stream, P, slot_cnt, slot_base, out_i, out_j, out_d); break;
    }
}

inline dim3 grid_of(uint32_t n) {
    const unsigned gx = n < (1u << 20) ? (n ? n : 1u) : (1u << 20);
    return dim3(gx, (n + gx - 1u) / gx);
}

}  // namespace

namespace mh {
void search64_release(molar_hip_ctx *c) {
    if (!c->s64) return;
    molar_hip_search64_state &Z = *c->s64;
    for (DevBuf *b : {&Z.posA, &Z.idA, &Z.vdwA, &Z.posB, &Z.idB, &Z.vdwB, &Z.slots, &Z.slot_cnt, &Z.slot_base, &Z.box, &Z.out_i, &Z.out_j, &Z.out_d,
                      &Z.in_xyz[0], &Z.in_xyz[1], &Z.in_idx[0], &Z.in_idx[1], &Z.in_vdw[0], &Z.in_vdw[1], &Z.key_in, &Z.key_out, &Z.val_in,
                      &Z.val_out, &Z.pos3, &Z.aabbA, &Z.aabbB, &Z.startA, &Z.startB, &Z.task_ns, &Z.task_first, &Z.cub_tmp, &Z.partial, &Z.flags})
        b->release();
    delete c->s64;
    c->s64 = nullptr;
}
}  // namespace mh

extern "C" {

// one input array on the device: used in place if it is device memory, copied to `stage` otherwise
static int to_device(molar_hip_ctx *c, const void *src, size_t bytes, DevBuf &stage, const void **out) {
    if (!src || !bytes) {
        *out = nullptr;
        return 0;
    }
    if (is_device_ptr(src)) {
        *out = src;
        return 0;
    }
    MH_TRY(stage.reserve(bytes));
    MH_HIP(hipMemcpyAsync(stage.p, src, bytes, hipMemcpyHostToDevice, c->stream));
    *out = stage.p;
    return 0;
}

static int host_copy(molar_hip_ctx *c, const double *src, size_t count, double *dst) {
    if (is_device_ptr(src)) MH_HIP(hipMemcpy(dst, src, count * 8, hipMemcpyDeviceToHost));
    else std::memcpy(dst, src, count * 8);
    return 0;
}

// grid of one set on the device: items in the reference's push order (pos / id / vdw), CSR starts per cell
static int build_grid64(molar_hip_ctx *c, molar_hip_search64_state &Z, const SetIn &S, const GridP &G, DevBuf &pos, DevBuf &id, DevBuf &vdw,
                        DevBuf &start, int *err_dev) {
    const uint32_t n = S.nsel;
    MH_TRY(start.reserve(((size_t)G.ncells + 1) * 4));
    MH_TRY(pos.reserve((size_t)(n ? n : 1) * 24));
    MH_TRY(id.reserve((size_t)(n ? n : 1) * 8));
    if (S.vdw) MH_TRY(vdw.reserve((size_t)(n ? n : 1) * 8));
    if (n == 0) {
        MH_HIP(hipMemsetAsync(start.p, 0, ((size_t)G.ncells + 1) * 4, c->stream));
        return 0;
    }
    MH_TRY(Z.key_in.reserve((size_t)n * 4));
    MH_TRY(Z.key_out.reserve((size_t)n * 4));
    MH_TRY(Z.val_in.reserve((size_t)n * 4));
    MH_TRY(Z.val_out.reserve((size_t)n * 4));
    MH_TRY(Z.pos3.reserve((size_t)n * 24));
    const unsigned nb = (n + 255u) / 256u;
    hipLaunchKernelGGL(assign64_kernel, dim3(nb), dim3(256), 0, c->stream, S, G, Z.key_in.as<uint32_t>(), Z.val_in.as<uint32_t>(),
                       Z.pos3.as<double>(), err_dev);
    int end_bit = 1;
    while (end_bit < 32 && (2ull * G.ncells) >> end_bit) ++end_bit;            // keys 0 .. 2 * ncells
    MH_TRY(device_sort_pairs_u32(c, Z.cub_tmp, Z.key_in.as<uint32_t>(), Z.key_out.as<uint32_t>(), Z.val_in.as<uint32_t>(), Z.val_out.as<uint32_t>(),
                                 n, end_bit));
    hipLaunchKernelGGL(gather64_kernel, dim3(nb), dim3(256), 0, c->stream, S, n, G.ncells, Z.key_out.as<uint32_t>(), Z.val_out.as<uint32_t>(),
                       Z.pos3.as<double>(), pos.as<double>(), id.as<unsigned long long>(), S.vdw ? vdw.as<double>() : nullptr);
    hipLaunchKernelGGL(cellstart64_kernel, dim3((G.ncells + 1u + 255u) / 256u), dim3(256), 0, c->stream, Z.key_out.as<uint32_t>(), n, G.ncells,
                       start.as<uint32_t>());
    MH_HIP(hipGetLastError());
    return 0;
}

int molar_hip_search_count_f64(molar_hip_ctx *c, const molar_hip_search_desc_f64 *q, uint64_t *out_count) {
    if (!c || !q) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "search_count_f64: null argument");
    MH_HIP(hipSetDevice(c->device));
    if (!c->s64) c->s64 = new molar_hip_search64_state;
    molar_hip_search64_state &Z = *c->s64;
    Z.have = false;
    const int kind = q->kind;
    if (kind < MOLAR_HIP_SEARCH_SINGLE || kind > MOLAR_HIP_SEARCH_DOUBLE_VDW)
        return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "search_count_f64: unknown kind %d", kind);
    const bool two = kind != MOLAR_HIP_SEARCH_SINGLE, vdw = kind == MOLAR_HIP_SEARCH_DOUBLE_VDW;
    if (vdw && (!q->vdw1 || !q->vdw2)) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "vdw search: radii pointer is null");
    if (!q->xyz1 || (two && !q->xyz2)) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "search_f64: xyz pointer is null");
    // ---- the two sets on the device (in place if they are device memory)
    SetIn S[2] = {};
    const double *hvdw[2] = {nullptr, nullptr};            // radii the host can read without a copy
    for (int w = 0; w < (two ? 2 : 1); ++w) {
        const double *xyz = w ? q->xyz2 : q->xyz1, *vd = vdw ? (w ? q->vdw2 : q->vdw1) : nullptr;
        const uint64_t *idx = w ? q->idx2 : q->idx1;
        const size_t natoms = w ? q->natoms2 : q->natoms1, nidx = w ? q->n2 : q->n1;
        const size_t nsel = idx ? nidx : natoms;
        if (nsel >= 0x7FFFFFFFull) return fail(MOLAR_HIP_ERR_TOO_LARGE, "search_f64: %zu atoms exceed the 2^31 limit", nsel);
        const void *dx, *di, *dv;
        MH_TRY(to_device(c, xyz, natoms * 24, Z.in_xyz[w], &dx));
        MH_TRY(to_device(c, idx, idx ? nsel * 8 : 0, Z.in_idx[w], &di));
        MH_TRY(to_device(c, vd, vd ? nsel * 8 : 0, Z.in_vdw[w], &dv));
        S[w] = SetIn{static_cast<const double *>(dx), static_cast<const unsigned long long *>(di), static_cast<const double *>(dv),
                     (uint64_t)natoms, (uint32_t)nsel, (q->ids_local || vdw) ? 1 : 0};
        if (vd && !is_device_ptr(vd)) hvdw[w] = vd;
    }
    BoxD box{};
    const bool use_box = q->box9 != nullptr;
    if (use_box) {
        double hb[9];
        MH_TRY(host_copy(c, q->box9, 9, hb));
        MH_TRY(box64_from_matrix(hb, &box));
    }
    Z.kind = kind;
    Z.use_box = use_box;
    Z.pbc = use_box ? q->pbc : 0;
    Z.total = 0;
    Z.nslots = 0;
    Z.dims[0] = Z.dims[1] = Z.dims[2] = 1;
    constexpr unsigned PARTS = 256;                        // workgroups of the reductions whose result the host needs
    MH_TRY(Z.partial.reserve((size_t)PARTS * 6 * 8 * 2));
    MH_TRY(Z.flags.reserve(64));
    MH_HIP(hipMemsetAsync(Z.flags.p, 0, 64, c->stream));
    int *err_dev = Z.flags.as<int>();
    std::vector<double> hpart((size_t)PARTS * 6 * 2);
    double cutoff = q->cutoff;
    if (vdw) {
        if (S[0].nsel == 0 || S[1].nsel == 0) {        // the reference unwrap()s an empty max: nothing to report here
            Z.have = true;
            if (out_count) *out_count = 0;
            return MOLAR_HIP_OK;
        }
        double mx[2];
        for (int w = 0; w < 2; ++w) {                    // Iterator::reduce(Float::max): NaN-ignoring max (:781-783)
            if (hvdw[w]) {
                double m = hvdw[w][0];
                for (size_t k = 1; k < S[w].nsel; ++k) m = std::fmax(m, hvdw[w][k]);
                mx[w] = m;
                continue;
            }
            const unsigned nb = std::min<unsigned>(PARTS, (S[w].nsel + 255u) / 256u);
            hipLaunchKernelGGL(fmax64_kernel, dim3(nb), dim3(256), 0, c->stream, S[w].vdw, S[w].nsel, Z.partial.as<double>());
            MH_HIP(hipMemcpyAsync(hpart.data(), Z.partial.p, (size_t)nb * 8, hipMemcpyDeviceToHost, c->stream));
            MH_HIP(hipStreamSynchronize(c->stream));
            double m = hpart[0];
            for (unsigned b = 1; b < nb; ++b) m = std::fmax(m, hpart[b]);
            mx[w] = m;
        }
        cutoff = (mx[0] + mx[1]) + 2.220446049250313e-16;
    }
    if (!(cutoff > 0.0)) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "search_f64: cutoff must be positive (got %g)", cutoff);
    Z.cutoff = cutoff;
    double lower[3] = {0, 0, 0}, upper[3] = {0, 0, 0}, ext[3];
    if (use_box) {
        ext[0] = (box.m[0] + box.m[3]) + box.m[6];                                          // get_lab_extents (:369-375)
        ext[1] = (box.m[1] + box.m[4]) + box.m[7];
        ext[2] = (box.m[2] + box.m[5]) + box.m[8];
    } else {
        if (kind == MOLAR_HIP_SEARCH_WITHIN) {
            if (!q->lower3 || !q->upper3) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "non-periodic within search needs lower3/upper3");
            MH_TRY(host_copy(c, q->lower3, 3, lower));
            MH_TRY(host_copy(c, q->upper3, 3, upper));
        } else {
            // compute_min_max (:602-616, seeded with zeros) + compute_bounding_box_single / _double (:618-646)
            double lo[2][3] = {}, hi[2][3] = {};
            for (int w = 0; w < (two ? 2 : 1); ++w) {
                const unsigned nb = std::max(1u, std::min<unsigned>(PARTS, (S[w].nsel + 255u) / 256u));
                hipLaunchKernelGGL(minmax64_kernel, dim3(nb), dim3(256), 0, c->stream, S[w], Z.partial.as<double>(), err_dev);
                MH_HIP(hipMemcpyAsync(hpart.data(), Z.partial.p, (size_t)nb * 48, hipMemcpyDeviceToHost, c->stream));
                MH_HIP(hipStreamSynchronize(c->stream));
                for (unsigned b = 0; b < nb; ++b)
                    for (int d = 0; d < 3; ++d) {
                        if (hpart[6 * b + d] < lo[w][d]) lo[w][d] = hpart[6 * b + d];
                        if (hpart[6 * b + 3 + d] > hi[w][d]) hi[w][d] = hpart[6 * b + 3 + d];
                    }
            }
            for (int d = 0; d < 3; ++d) {
                lower[d] = lo[0][d];
                upper[d] = hi[0][d];
                if (two) {
                    lower[d] = lo[0][d] < lo[1][d] ? lo[0][d] : lo[1][d];
                    upper[d] = hi[0][d] > hi[1][d] ? hi[0][d] : hi[1][d];
                }
                lower[d] += (-cutoff - 2.220446049250313e-16);
                upper[d] += (cutoff + 2.220446049250313e-16);
            }
        }
        for (int d = 0; d < 3; ++d) ext[d] = upper[d] - lower[d];
    }
    uint64_t dims[3];
    double ncell_d = 1.0;
    for (int d = 0; d < 3; ++d) {                                                            // :103-110
        dims[d] = std::max<uint64_t>(as_usize(std::floor(ext[d] / cutoff)), 1);
        ncell_d *= (double)dims[d];
    }
    const uint32_t mult = two ? 2u : 1u;
    if (!(ncell_d * 14.0 * mult < 2.0e9))
        return fail(MOLAR_HIP_ERR_TOO_LARGE, "search_f64: grid of %.3g cells (the plan of this path holds one word per entry)", ncell_d);
    std::memcpy(Z.dims, dims, sizeof dims);
    GridP G{};
    G.box = box;
    G.use_box = use_box ? 1 : 0;
    G.pbc = Z.pbc;
    for (int d = 0; d < 3; ++d) { G.dims[d] = (uint32_t)dims[d]; G.lower[d] = lower[d]; G.upper[d] = upper[d]; }
    G.ncells = (uint32_t)(dims[0] * dims[1] * dims[2]);
    // ---- grids (populate / populate_pbc) and the plan, on the device
    MH_TRY(build_grid64(c, Z, S[0], G, Z.posA, Z.idA, Z.vdwA, Z.startA, err_dev));
    if (two) MH_TRY(build_grid64(c, Z, S[1], G, Z.posB, Z.idB, Z.vdwB, Z.startB, err_dev));
    const uint32_t *startA = Z.startA.as<uint32_t>(), *startB = two ? Z.startB.as<uint32_t>() : startA;
    {   // bounding boxes of the second grid's cells
        DevBuf &bb = two ? Z.aabbB : Z.aabbA;
        MH_TRY(bb.reserve((size_t)G.ncells * 48));
        hipLaunchKernelGGL(aabb64_kernel, dim3((G.ncells + 3u) / 4u), dim3(256), 0, c->stream, startB, G.ncells,
                           (two ? Z.posB : Z.posA).as<double>(), bb.as<double>());
    }
    const uint64_t ntasks = (uint64_t)G.ncells * 14ull * mult;
    MH_TRY(Z.task_ns.reserve((ntasks + 1) * 4));
    MH_TRY(Z.task_first.reserve((ntasks + 1) * 4));
    hipLaunchKernelGGL(plan64_kernel, dim3((unsigned)((ntasks + 1 + 255) / 256)), dim3(256), 0, c->stream, G, ntasks, mult, startA, startB,
                       Z.task_ns.as<uint32_t>());
    MH_TRY(device_exclusive_sum_u32(c, Z.cub_tmp, Z.task_ns.as<uint32_t>(), Z.task_first.as<uint32_t>(), ntasks + 1));
    struct { uint32_t nslots; int err; } hs = {0, 0};
    MH_HIP(hipMemcpyAsync(&hs.nslots, Z.task_first.as<uint32_t>() + ntasks, 4, hipMemcpyDeviceToHost, c->stream));
    MH_HIP(hipMemcpyAsync(&hs.err, err_dev, 4, hipMemcpyDeviceToHost, c->stream));
    MH_HIP(hipStreamSynchronize(c->stream));
    if (hs.err) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "search_f64: selection index out of range");
    if (hs.nslots >= 0x7FFFFF00u) return fail(MOLAR_HIP_ERR_TOO_LARGE, "search_f64: plan too large");
    Z.nslots = hs.nslots;
    Z.total = 0;
    if (Z.nslots == 0) {
        Z.have = true;
        if (out_count) *out_count = 0;
        return MOLAR_HIP_OK;
    }
    MH_TRY(Z.slots.reserve((size_t)Z.nslots * sizeof(Slot64)));
    hipLaunchKernelGGL(slots64_kernel, dim3((unsigned)((ntasks + 255) / 256)), dim3(256), 0, c->stream, G, ntasks, mult, startA, startB,
                       Z.task_first.as<uint32_t>(), Z.slots.as<Slot64>());
    // ---- count, offsets
    MH_TRY(Z.box.reserve(sizeof box));
    MH_HIP(hipMemcpyAsync(Z.box.p, &box, sizeof box, hipMemcpyHostToDevice, c->stream));
    MH_TRY(Z.slot_cnt.reserve(((size_t)Z.nslots + 1) * 4));
    MH_TRY(Z.slot_base.reserve(((size_t)Z.nslots + 1) * 8));
    MH_HIP(hipMemsetAsync(Z.slot_cnt.as<uint32_t>() + Z.nslots, 0, 4, c->stream));
    const DevBuf &pB = two ? Z.posB : Z.posA, &iB = two ? Z.idB : Z.idA, &vB = two ? Z.vdwB : Z.vdwA;
    // Entries across the periodic boundary (search.hip, make_params, derives the same bound for f32): the reference evaluates
    // v = p2 - p1, f = inv v, f -= round(f), s = M f; the kernels classify with (b + S) - a.  With u = 2^-53, L the largest
    // |coordinate| the box allows and kappa = || |M| |M^-1| ||_inf the two difference vectors disagree by at most
    // e = (4 kappa + 9) u L per component, d2 near cutoff^2 by 2 sqrt(3) e / rc relative to cutoff^2.  Outside a band of
    // 1e-9 + four times that around cutoff^2 the plain distance to the adjacent image decides; inside it, and for every hit's
    // distance, PeriodicBox::distance_squared itself.  Needs >= 4 cells along every periodic dimension (round(f_d) = +-1 for
    // every pair of a wrapped entry that can be within the cutoff).
    Z.approx = false;
    Z.band_lo = Z.band_hi = 1.0;
    if (use_box) {
        bool ok = true;
        double lmax = 0.0, lsum = 0.0, kappa = 0.0;
        for (int d = 0; d < 3; ++d) {
            if (((Z.pbc >> d) & 1u) && dims[d] < 4u) ok = false;
            lmax = std::fmax(lmax, std::fabs(ext[d]));
            double row = 0.0;
            for (int k = 0; k < 3; ++k) {
                lmax = std::fmax(lmax, std::fabs(box.m[3 * k + d]));
                row += std::fabs(box.m[3 * k + d]);
            }
            lsum = std::fmax(lsum, row);
        }
        lmax = std::fmax(lmax, lsum);
        for (int i = 0; i < 3; ++i) {
            double row = 0.0;
            for (int j = 0; j < 3; ++j)
                for (int k = 0; k < 3; ++k) row += std::fabs(box.m[3 * k + i]) * std::fabs(box.inv[3 * j + k]);
            kappa = std::fmax(kappa, row);
        }
        const double e = (4.0 * kappa + 9.0) * 1.1102230246251565e-16 * lmax;
        const double rel = 1.0e-9 + 4.0 * 2.0 * 1.7320508075688772 * e / cutoff;
        if (ok && !vdw && std::isfinite(kappa) && rel < 1.0e-3) {        // (vdW: the cutoff differs from pair to pair - always exact)
            Z.approx = true;
            Z.band_lo = 1.0 - rel;
            Z.band_hi = 1.0 + rel;
            const double lim = cutoff + (1.0e-9 + 4.0 * 1.7320508075688772 * e);      // row pruning against the image box: the same margin
            Z.prune_limit2 = lim * lim;
        }
    }
    Params64 P{Z.posA.as<double>(), pB.as<double>(), Z.vdwA.as<double>(), vB.as<double>(),
               Z.idA.as<unsigned long long>(), iB.as<unsigned long long>(), Z.box.as<BoxD>(), Z.slots.as<Slot64>(), Z.nslots,
               kind, use_box ? 1 : 0, cutoff * cutoff, Z.approx ? 1 : 0, Z.band_lo, Z.band_hi, (two ? Z.aabbB : Z.aabbA).as<double>(), Z.prune_limit2};
    launch_pair64<false>(kind, grid_of(Z.nslots), c->stream, P, Z.slot_cnt.as<uint32_t>(), nullptr, nullptr, nullptr, nullptr);
    MH_HIP(hipGetLastError());
    MH_TRY(device_exclusive_sum_u32_u64(c, Z.cub_tmp, Z.slot_cnt.as<uint32_t>(), Z.slot_base.as<unsigned long long>(), (size_t)Z.nslots + 1));
    unsigned long long run = 0;
    MH_HIP(hipMemcpyAsync(&run, Z.slot_base.as<unsigned long long>() + Z.nslots, 8, hipMemcpyDeviceToHost, c->stream));
    MH_HIP(hipStreamSynchronize(c->stream));
    Z.total = run;
    Z.have = true;
    if (out_count) *out_count = run;
    return MOLAR_HIP_OK;
}

static int fill64(molar_hip_ctx *c, uint64_t *oi, uint64_t *oj, double *od, bool within) {
    if (!c || !c->s64 || !c->s64->have) return fail(MOLAR_HIP_ERR_NO_SEARCH, "no cached f64 search: call molar_hip_search_count_f64 first");
    molar_hip_search64_state &Z = *c->s64;
    if ((Z.kind == MOLAR_HIP_SEARCH_WITHIN) != within)
        return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, within ? "fill_ids_f64 is for within searches" : "within search yields ids: use molar_hip_search_fill_ids_f64");
    if (Z.total == 0 || Z.nslots == 0) return MOLAR_HIP_OK;
    MH_HIP(hipSetDevice(c->device));
    const size_t n = (size_t)Z.total;
    auto dev = [&](void *user, DevBuf &own, size_t bytes, void **out) -> int {
        if (user && is_device_ptr(user)) { *out = user; return 0; }
        MH_TRY(own.reserve(bytes));
        *out = own.p;
        return 0;
    };
    void *di, *dj = nullptr, *dd = nullptr;
    MH_TRY(dev(oi, Z.out_i, n * 8, &di));
    if (!within) {
        MH_TRY(dev(oj, Z.out_j, n * 8, &dj));
        MH_TRY(dev(od, Z.out_d, n * 8, &dd));
    }
    const bool two = Z.kind != MOLAR_HIP_SEARCH_SINGLE;
    const DevBuf &pB = two ? Z.posB : Z.posA, &iB = two ? Z.idB : Z.idA, &vB = two ? Z.vdwB : Z.vdwA;
    Params64 P{Z.posA.as<double>(), pB.as<double>(), Z.vdwA.as<double>(), vB.as<double>(),
               Z.idA.as<unsigned long long>(), iB.as<unsigned long long>(), Z.box.as<BoxD>(), Z.slots.as<Slot64>(), Z.nslots,
               Z.kind, Z.use_box ? 1 : 0, Z.cutoff * Z.cutoff, Z.approx ? 1 : 0, Z.band_lo, Z.band_hi, (two ? Z.aabbB : Z.aabbA).as<double>(), Z.prune_limit2};
    launch_pair64<true>(Z.kind, grid_of(Z.nslots), c->stream, P, nullptr, Z.slot_base.as<unsigned long long>(),
                        static_cast<unsigned long long *>(di), static_cast<unsigned long long *>(dj), static_cast<double *>(dd));
    MH_HIP(hipGetLastError());
    std::vector<RingJob> jobs;
    if (oi && di != oi) jobs.push_back(RingJob{di, n * 8, RING_COPY, oi, nullptr});
    if (oj && dj != oj) jobs.push_back(RingJob{dj, n * 8, RING_COPY, oj, nullptr});
    if (od && dd != od) jobs.push_back(RingJob{dd, n * 8, RING_COPY, od, nullptr});
    if (n * 8 * jobs.size() >= (24u << 20)) {
        MH_TRY(ring_to_host(c, jobs));              // large results: pinned ring + host threads (hoststream.hpp)
    } else {
        for (const RingJob &J : jobs) MH_HIP(hipMemcpyAsync(J.dst0, J.src, J.bytes, hipMemcpyDeviceToHost, c->stream));
    }
    MH_HIP(hipStreamSynchronize(c->stream));
    return MOLAR_HIP_OK;
}

int molar_hip_search_fill_f64(molar_hip_ctx *c, uint64_t *i, uint64_t *j, double *dist) { return fill64(c, i, j, dist, false); }

int molar_hip_search_fill_ids_f64(molar_hip_ctx *c, uint64_t *ids) { return fill64(c, ids, nullptr, nullptr, true); }

int molar_hip_search_grid_dims_f64(molar_hip_ctx *c, uint64_t dims[3]) {
    if (!c || !c->s64 || !c->s64->have || !dims) return fail(MOLAR_HIP_ERR_NO_SEARCH, "no cached f64 search");
    for (int d = 0; d < 3; ++d) dims[d] = c->s64->dims[d];
    return MOLAR_HIP_OK;
}

}  // extern "C"
