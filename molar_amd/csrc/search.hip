// search.hip — MolAR's cell-list distance search (molar/src/distance_search.rs) for gfx950.
//
// Pipeline per search (all on ctx->stream, no host round trip except the scalar result count
// and, for non-periodic drivers, the bounding box):
//   bin      one thread per atom: fractional coords -> cell, wrap/drop decision      (:120-199)
//   scan     cell counts -> cell_start
//   scatter  unordered placement into the cell's segment
//   place    rank inside the segment by (wrapped, input index) -> reference cell order
//            (in-box atoms in input order, then wrapped atoms, :180,:203-209); writes the
//            cell-sorted float4 {x,y,z,id} array the pair kernels read
//   count    one 64-lane wave per plan entry (x,y,z outer..inner, 14 masks, :217-269): lanes hold
//            the atoms of the second cell in registers, the first cell's atoms arrive through
//            scalar loads; v_cmp masks are popcounted -> hits per task
//   scan     task counts -> 64-bit output offsets (reference order = plan order, :949-953)
//   fill     same traversal; hits are compacted with mbcnt prefix ranks into a per-wave LDS
//            FIFO (i-major, j-minor = the reference's inner loop order) and flushed 64 at a
//            time as fully coalesced (u32,u32) + f32 stores
//
// All f32 arithmetic follows boxmath.hpp (reference operation order, no FMA contraction).
#include <cstdlib>

#include "boxmath.hpp"
#include "common.hpp"
#include "hoststream.hpp"
#include <algorithm>

#include "pair_kernels.hpp"
#include "stages.hpp"

namespace mh {
// pair_small.hip: count / fill kernels for frames of small cells, 64 / lanes_per_slot slots per wave
void launch_pair_small(int mode, int lanes_per_slot, hipStream_t stream, const pairk::SearchParams *dP, const pairk::SlotDesc *slot_desc,
                       uint32_t nslots, uint32_t *slot_cnt, const unsigned long long *slot_base, uint2 *pairs, float *dist);
}
using namespace mh;
using namespace mh::pairk;

namespace {

constexpr uint32_t DROPPED = 0xFFFFFFFFu;

// ================================================================= grid build

struct BinParams {
    const float *xyz;
    const uint64_t *idx;
    uint32_t n;
    uint32_t dx, dy, dz;
    uint32_t pbc;
    uint32_t use_box;
    float lower[3], upper[3];
    molar_hip_box box;
};

struct CellOfAtom {
    uint32_t key;   // cell<<1 | wrapped, or DROPPED
    V3 pos;         // position stored in the grid (original or wrapped image)
};

// Grid::populate_pbc (distance_search.rs:144-210) / Grid::populate (:120-142) for one atom.
__device__ __forceinline__ CellOfAtom classify(const BinParams &P, V3 p) {
    CellOfAtom r;
    r.pos = p;
    if (P.use_box) {
        V3 rel = mat_vec(P.box.inv, p);
        float rl[3] = {rel.x, rel.y, rel.z};
        bool wrap = false;
        for (int d = 0; d < 3; ++d) {
            if (rl[d] < 0.0f || rl[d] >= 1.0f) {
                if (!((P.pbc >> d) & 1u)) {
                    r.key = DROPPED;
                    return r;
                }
                wrap = true;
                break;
            }
        }
        const uint32_t dims[3] = {P.dx, P.dy, P.dz};
        uint32_t loc[3];
        if (!wrap) {
            for (int d = 0; d < 3; ++d) loc[d] = floor_to_cell(rl[d] * (float)dims[d], dims[d]);
        } else {
            for (int d = 0; d < 3; ++d) {
                if ((P.pbc >> d) & 1u) {
                    rl[d] = fract_rs(rl[d]);
                    if (rl[d] < 0.0f) rl[d] = 1.0f + rl[d];
                }
                loc[d] = floor_to_cell(rl[d] * (float)dims[d], dims[d]);
            }
            r.pos = mat_vec(P.box.m, v3(rl[0], rl[1], rl[2]));
        }
        r.key = ((loc[0] + loc[1] * P.dx + loc[2] * P.dx * P.dy) << 1) | (wrap ? 1u : 0u);
        return r;
    }
    const float pp[3] = {p.x, p.y, p.z};
    const uint32_t dims[3] = {P.dx, P.dy, P.dz};
    uint32_t loc[3];
    for (int d = 0; d < 3; ++d) {
        const float dim_sz = P.upper[d] - P.lower[d];
        const float f = __builtin_floorf((float)dims[d] * (pp[d] - P.lower[d]) / dim_sz);
        if (f != f) {               // NaN as isize == 0
            loc[d] = 0;
            continue;
        }
        if (f < 0.0f || f >= (float)dims[d]) {
            r.key = DROPPED;
            return r;
        }
        loc[d] = (uint32_t)f;
    }
    r.key = (loc[0] + loc[1] * P.dx + loc[2] * P.dx * P.dy) << 1;
    return r;
}

__device__ __forceinline__ V3 load_pos(const float *xyz, uint64_t a) {
    const float *q = xyz + 3 * a;
    return v3(q[0], q[1], q[2]);
}

// `counters` holds one counter per cell at stride 1 << pad_shift words: with a few thousand cells and hundreds of
// atoms per cell, 32 neighbouring counters in one 128-byte line serialise in L2; one counter per line does not.
// (Round 4: the kernel takes 44 us alone on the 1M-atom frame and 8.7 us with the atomic taken out - it is bound by the
// chip's rate of returning atomics, ~28 G/s, not by queues on single addresses: one set of counters per XCD, chosen by
// HW_REG_XCC_ID and incremented at workgroup scope, left it at 41 us.  The compiler emits the same instruction -
// global_atomic_add ... sc0 - for workgroup and agent scope on gfx950.)
__device__ __forceinline__ void bin_body(const BinParams &P, uint32_t *__restrict__ key, uint32_t *__restrict__ arrival,
                                         uint32_t *__restrict__ counters, uint32_t pad_shift) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= P.n) return;
    const uint64_t a = P.idx ? P.idx[k] : (uint64_t)k;
    const CellOfAtom c = classify(P, load_pos(P.xyz, a));
    key[k] = c.key;
    // arrival order inside the cell (arbitrary): lets scatter place the atom without a second atomic
    if (c.key != DROPPED) arrival[k] = atomicAdd(&counters[(size_t)(c.key >> 1) << pad_shift], 1u);
}
__global__ void __launch_bounds__(256) bin_kernel(BinParams P, uint32_t *__restrict__ key,
                                                  uint32_t *__restrict__ arrival, uint32_t *__restrict__ counters,
                                                  uint32_t pad_shift) {
    bin_body(P, key, arrival, counters, pad_shift);
}

// The grids of several frames in one set of launches (molar_hip_search_histogram_frames): what the kernels of ONE frame take, as a
// record in device memory; blockIdx.y picks the frame.  The frames of a batch have the same number of atoms and cells, so the
// launch shapes and the scalar arguments are shared.
struct GridFrame {
    BinParams P;
    uint32_t *key, *cursor, *counters, *cell_count, *cnt_pad, *tmp_key;
    float4 *sorted, *aabb, *perm, *chunk_aabb, *cell_org;
    uint4 *h16;
};
__global__ void __launch_bounds__(256) bin_frames_kernel(const GridFrame *__restrict__ G, uint32_t pad_shift) {
    const GridFrame &g = G[blockIdx.y];
    bin_body(g.P, g.key, g.cursor, g.counters, pad_shift);
}

// The same with the counters privatised (round 4): a workgroup of 256 threads takes a tile of 256 * BIN_PER_THREAD consecutive atoms,
// counts them per cell in LDS (returning LDS atomics: the atom's rank inside the tile's share of its cell), reserves each touched
// cell's share with ONE returning global atomic and adds the base to the ranks.  A tile of 8192 atoms touches ~3400 of the headline
// frame's 3825 cells: 0.41 M global atomics instead of 1 M.  (Arrival order stays arbitrary - place_order_kernel ranks by input
// index.)  For grids whose counters fit in LDS.
constexpr uint32_t BIN_TILE_MAX_CELLS = 12288;      // 48 KB of LDS counters
template <uint32_t BIN_PER_THREAD>
__device__ __forceinline__ void bin_tile_body(const BinParams &P, uint32_t *__restrict__ key, uint32_t *__restrict__ arrival,
                                              uint32_t *__restrict__ counters, uint32_t pad_shift, uint32_t ncells) {
    extern __shared__ uint32_t bin_cnt[];
    for (uint32_t c = threadIdx.x; c < ncells; c += 256u) bin_cnt[c] = 0u;
    __syncthreads();
    const uint32_t k0 = blockIdx.x * (256u * BIN_PER_THREAD) + threadIdx.x;
    uint32_t ky[BIN_PER_THREAD], lr[BIN_PER_THREAD];
#pragma unroll
    for (uint32_t u = 0; u < BIN_PER_THREAD; ++u) {
        const uint32_t k = k0 + u * 256u;
        ky[u] = DROPPED;
        lr[u] = 0u;
        if (k < P.n) {
            const uint64_t a = P.idx ? P.idx[k] : (uint64_t)k;
            ky[u] = classify(P, load_pos(P.xyz, a)).key;
            key[k] = ky[u];
            if (ky[u] != DROPPED) lr[u] = atomicAdd(&bin_cnt[ky[u] >> 1], 1u);
        }
    }
    __syncthreads();
    for (uint32_t c = threadIdx.x; c < ncells; c += 256u) {
        const uint32_t m = bin_cnt[c];
        if (m) bin_cnt[c] = atomicAdd(&counters[(size_t)c << pad_shift], m);
    }
    __syncthreads();
#pragma unroll
    for (uint32_t u = 0; u < BIN_PER_THREAD; ++u) {
        const uint32_t k = k0 + u * 256u;
        if (k < P.n && ky[u] != DROPPED) arrival[k] = bin_cnt[ky[u] >> 1] + lr[u];
    }
}
template <uint32_t BIN_PER_THREAD>
__global__ void __launch_bounds__(256) bin_tile_kernel(BinParams P, uint32_t *__restrict__ key, uint32_t *__restrict__ arrival,
                                                       uint32_t *__restrict__ counters, uint32_t pad_shift, uint32_t ncells) {
    bin_tile_body<BIN_PER_THREAD>(P, key, arrival, counters, pad_shift, ncells);
}
template <uint32_t BIN_PER_THREAD>
__global__ void __launch_bounds__(256) bin_tile_frames_kernel(const GridFrame *__restrict__ G, uint32_t pad_shift, uint32_t ncells) {
    const GridFrame &g = G[blockIdx.y];
    bin_tile_body<BIN_PER_THREAD>(g.P, g.key, g.cursor, g.counters, pad_shift, ncells);
}

// one launch instead of hipMemsetAsync, which splits an unaligned range into up to three fill kernels (~5 us each)
__global__ void __launch_bounds__(256) zero2_kernel(uint32_t *__restrict__ a, size_t na, uint32_t *__restrict__ b, size_t nb) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < na + nb; i += (size_t)gridDim.x * blockDim.x) {
        if (i < na) a[i] = 0u;
        else b[i - na] = 0u;
    }
}

__global__ void __launch_bounds__(256) add_u64_kernel(unsigned long long *__restrict__ dst, const unsigned long long *__restrict__ src,
                                                      size_t n) {
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i < n) dst[i] += src[i];
}

__global__ void __launch_bounds__(256) unpad_kernel(uint32_t ncells, const uint32_t *__restrict__ padded, uint32_t pad_shift,
                                                    uint32_t *__restrict__ cell_count) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < ncells) cell_count[c] = padded[(size_t)c << pad_shift];
}
// (frames: zero the counters of every frame / copy the padded counters out)
__global__ void __launch_bounds__(256) zero_frames_kernel(const GridFrame *__restrict__ G, size_t na, size_t nb) {
    const GridFrame &g = G[blockIdx.y];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < na + nb; i += (size_t)gridDim.x * blockDim.x) {
        if (i < na) g.cell_count[i] = 0u;
        else g.cnt_pad[i - na] = 0u;
    }
}
// Padded counters -> cell starts of every frame (one workgroup per frame): reads a frame's ncells counters, leaves them ZERO for the
// next group that bins into this generation (so nothing has to be zeroed in front of the binning kernel again), and writes the
// exclusive scan - cell_count[0 .. ncells], the last entry the number of kept atoms.  unpad + scan + zero of the single-frame
// path in one launch.
__global__ void __launch_bounds__(256) unpad_scan_frames_kernel(const GridFrame *__restrict__ G, uint32_t ncells, uint32_t pad_shift) {
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t carry_s;
    const GridFrame &g = G[blockIdx.x];
    if (threadIdx.x == 0) carry_s = 0u;
    __syncthreads();
    for (uint32_t start = 0; start <= ncells; start += 1024u) {
        const uint32_t base = start + threadIdx.x * 4u;
        uint32_t item[4], sum = 0;
        for (int q = 0; q < 4; ++q) {
            const uint32_t cidx = base + q;
            item[q] = 0u;
            if (cidx < ncells) {
                item[q] = g.cnt_pad[(size_t)cidx << pad_shift];
                g.cnt_pad[(size_t)cidx << pad_shift] = 0u;
            }
            sum += item[q];
        }
        uint32_t inc = sum;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(inc, off, 64);
            if ((int)(threadIdx.x & 63u) >= off) inc += o;
        }
        if ((threadIdx.x & 63u) == 63u) wsum[threadIdx.x >> 6] = inc;
        __syncthreads();
        uint32_t pre = carry_s;
        for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) pre += wsum[w];
        uint32_t run = pre + inc - sum;
        for (int q = 0; q < 4; ++q) {
            if (base + q <= ncells) g.cell_count[base + q] = run;
            run += item[q];
        }
        __syncthreads();
        if (threadIdx.x == 255u) carry_s = pre + inc;
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) scatter_kernel(uint32_t n, const uint32_t *__restrict__ key,
                                                      const uint32_t *__restrict__ cell_start,
                                                      const uint32_t *__restrict__ arrival,
                                                      uint32_t *__restrict__ tmp_key) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t ky = key[k];
    if (ky == DROPPED) return;
    const uint32_t cell = ky >> 1;
    const uint32_t pos = cell_start[cell] + arrival[k];
    tmp_key[pos] = ((ky & 1u) << 31) | k;     // sort key inside the cell: in-box first, then input order
}
__global__ void __launch_bounds__(256) scatter_frames_kernel(const GridFrame *__restrict__ G, uint32_t n) {
    const GridFrame &g = G[blockIdx.y];
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t ky = g.key[k];
    if (ky == DROPPED) return;
    g.tmp_key[g.cell_count[ky >> 1] + g.cursor[k]] = ((ky & 1u) << 31) | k;
}

// Grids of large cells (more than 384 atoms per cell on average: cutoffs that are large against the box) order their atoms by a
// stable device sort of (cell << 1 | wrapped, atom number) instead of ranking every atom against its whole cell (quadratic in the
// cell population: with one wave per 64 atoms of a cell, 100 us for 64 cells of 1560 atoms, 1 ms for 660 of them): sort_prep_kernel makes the keys
// (dropped atoms behind everything), sort_tmpkey_kernel turns the sorted order into the tmp_key entries place_order_kernel reads,
// which then takes an entry's position as its rank.
__global__ void __launch_bounds__(256) sort_prep_kernel(uint32_t n, uint32_t ncells, const uint32_t *__restrict__ key, uint32_t *__restrict__ k2,
                                                        uint32_t *__restrict__ v) {
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= n) return;
    const uint32_t ky = key[k];
    k2[k] = ky == DROPPED ? 2u * ncells : ky;
    v[k] = k;
}
__global__ void __launch_bounds__(256) sort_tmpkey_kernel(uint32_t n, uint32_t ncells, const uint32_t *__restrict__ k_sorted,
                                                          const uint32_t *__restrict__ v_sorted, uint32_t *__restrict__ tmp_key) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= n) return;
    const uint32_t ky = k_sorted[t];
    if (ky < 2u * ncells) tmp_key[t] = ((ky & 1u) << 31) | v_sorted[t];      // kept atoms come first, cell by cell: t is the final position
}

// One wave per cell does everything that depends on the cell's content:
//  * PLACE: rank of every atom inside the cell segment by (wrapped, input index) - the reference's push order:
//    in-box atoms in input order, then wrapped atoms (distance_search.rs:180,203-209) - and its record
//    sorted[s + rank] = {x, y, z, id}.  Cells of <= 512 atoms (everything the register-resident pair path takes)
//    rank against the keys held in LDS; larger cells loop over global memory.
//  * the axis-aligned bounding box of the stored positions (aabb[2c] = lo, aabb[2c+1] = hi; lets a row of the pair
//    kernels prove "no atom of the other cell can be within the cutoff");
//  * for cells of <= 512 atoms a SPATIAL order of their atoms for the count pass:
//      perm[cell_start[c] + m]  = {x, y, z, position inside the cell} of the m-th atom in Morton order (3 bits per
//                                 axis of the cell's bounding box; LDS counting sort, order inside a key irrelevant;
//                                 a second copy of the coordinates, so the count pass loads them without a gather),
//      chunk_aabb[2*u], [2*u+1] = bounding box of the 64 atoms of Morton chunk k, u = (cell_start[c] >> 6) + c + k
//                                 (distinct for all chunks of all cells: a cell owns floor(n/64)+1 >= ceil(n/64) slots).
//    Counting does not depend on the order in which candidates are visited, so the count pass walks compact
//    chunks and skips (row, chunk) pairs by bounding box; the fill pass keeps the reference's order.
constexpr uint32_t ORDER_MAX = 512;   // = KREG * 64 of the pair kernels

// The PLACE step and the bounding box alone, 16 lanes per cell: grids of ~1e5 cells of a few atoms (vdW cutoffs, `within` with a
// short range) that want no spatial order.  A wave per cell spends its time starting up: 98 736 waves took 68 us for a set of
// 950k atoms and 40 us for one of 50k.  Keys are ranked straight from global memory (a cell's segment is one or two lines).
constexpr uint32_t OCC_WORDS = 16;      // the occupied-cell count of a grid, spread over this many words behind its cell starts
__global__ void __launch_bounds__(256) place_small_kernel(BinParams P, uint32_t ncells, int ids_local,
                                                          const uint32_t *__restrict__ cell_start,
                                                          const uint32_t *__restrict__ tmp_key, const float *__restrict__ vdw,
                                                          float4 *__restrict__ sorted, float *__restrict__ sorted_vdw,
                                                          float4 *__restrict__ aabb, float4 *__restrict__ cell_org,
                                                          uint32_t *__restrict__ occupied) {
    const uint32_t sub = threadIdx.x & 15u;
    const uint32_t c = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const bool live = c < ncells;                               // (whole 16-lane groups: the shuffles below stay inside a group)
    const uint32_t s = live ? cell_start[c] : 0u, e = live ? cell_start[c + 1] : 0u, n = e - s;
    if (occupied && sub == 0 && n) atomicAdd(occupied + (c & (OCC_WORDS - 1u)), 1u);      // cells that hold atoms (small_cell_lanes)
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t t = sub; t < n; t += 16u) {
        const uint32_t mine = tmp_key[s + t];
        uint32_t rank = 0;
        for (uint32_t q = s; q < e; ++q) rank += tmp_key[q] < mine ? 1u : 0u;
        const uint32_t k = mine & 0x7FFFFFFFu;
        const uint64_t a = P.idx ? P.idx[k] : (uint64_t)k;
        const CellOfAtom ca = classify(P, load_pos(P.xyz, a));   // same arithmetic as bin_kernel
        const uint32_t id = ids_local ? k : (uint32_t)a;
        sorted[s + rank] = make_float4(ca.pos.x, ca.pos.y, ca.pos.z, __uint_as_float(id));
        if (vdw) sorted_vdw[s + rank] = vdw[k];
        lo[0] = fminf(lo[0], ca.pos.x); hi[0] = fmaxf(hi[0], ca.pos.x);
        lo[1] = fminf(lo[1], ca.pos.y); hi[1] = fmaxf(hi[1], ca.pos.y);
        lo[2] = fminf(lo[2], ca.pos.z); hi[2] = fmaxf(hi[2], ca.pos.z);
    }
    for (int d = 0; d < 3; ++d)
        for (int off = 8; off > 0; off >>= 1) {
            lo[d] = fminf(lo[d], __shfl_xor(lo[d], off, 64));
            hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], off, 64));
        }
    if (live && sub == 0) {
        aabb[2 * c] = make_float4(lo[0], lo[1], lo[2], 0.f);
        aabb[2 * c + 1] = make_float4(hi[0], hi[1], hi[2], 0.f);
        const float org[3] = {0.5f * (lo[0] + hi[0]), 0.5f * (lo[1] + hi[1]), 0.5f * (lo[2] + hi[2])};
        const float ex = fmaxf(hi[0] - org[0], org[0] - lo[0]), ey = fmaxf(hi[1] - org[1], org[1] - lo[1]),
                    ez = fmaxf(hi[2] - org[2], org[2] - lo[2]);
        cell_org[c] = make_float4(org[0], org[1], org[2], 1.0001f * sqrtf((ex * ex + ey * ey) + ez * ez));
    }
}
__device__ __forceinline__ void place_order_body(const BinParams &P, uint32_t ncells, int ids_local,
                                                          const uint32_t *__restrict__ cell_start,
                                                          const uint32_t *__restrict__ tmp_key, const float *__restrict__ vdw,
                                                          float4 *__restrict__ sorted, float *__restrict__ sorted_vdw,
                                                          float4 *__restrict__ aabb, float4 *__restrict__ perm,
                                                          float4 *__restrict__ chunk_aabb, uint4 *__restrict__ h16,
                                                          float4 *__restrict__ cell_org, int want_order, int presorted) {
    // One wave per workgroup (5 KB of LDS): the grid of the NEXT frame is built on the side stream while the fill pass of
    // the frame in flight holds every wave slot of the chip with one-wave workgroups.  A freed slot takes a one-wave
    // workgroup of either queue; a four-wave workgroup needs four free slots on ONE compute unit at the same moment and
    // starves until the fill pass has nothing left to dispatch (round 3: the grid ended 5 us after the fill pass and the
    // next frame's plan waited for it).
    __shared__ uint32_t keys_s[ORDER_MAX];        // sort keys of the cell; reused as the Morton histogram
    __shared__ uint32_t kr_s[ORDER_MAX];
    __shared__ uint16_t perm_s[ORDER_MAX];
    // (the placed records are read back from `sorted` - written by this wave - rather than kept in LDS)
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t c = blockIdx.x;
    if (c >= ncells) return;
    const uint32_t s = cell_start[c], e = cell_start[c + 1], n = e - s;
    const bool small = n <= ORDER_MAX;
    uint32_t *keys = keys_s, *kr = kr_s;
    uint16_t *pl = perm_s;
    const float4 *pos = sorted + s;               // valid once the placement loop's stores are visible (fence below)
    if (small) {
        for (uint32_t t = lane; t < n; t += 64u) keys[t] = tmp_key[s + t];
        __builtin_amdgcn_wave_barrier();
    }
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t t = lane; t < n; t += 64u) {
        const uint32_t mine = small ? keys[t] : tmp_key[s + t];
        uint32_t rank = 0;
        if (presorted) {
            rank = t;                                           // tmp_key is already in the reference's order (sort_tmpkey_kernel)
        } else if (small) {
            const uint4 *k4 = reinterpret_cast<const uint4 *>(keys);
            uint32_t q = 0;
            for (; q + 4u <= n; q += 4u) {                      // wave-uniform address: one broadcast ds_read_b128
                const uint4 v = k4[q >> 2];
                rank += (v.x < mine ? 1u : 0u) + (v.y < mine ? 1u : 0u) + (v.z < mine ? 1u : 0u) + (v.w < mine ? 1u : 0u);
            }
            for (; q < n; ++q) rank += keys[q] < mine ? 1u : 0u;
        } else {
#pragma unroll 8
            for (uint32_t q = s; q < e; ++q) rank += tmp_key[q] < mine ? 1u : 0u;
        }
        const uint32_t k = mine & 0x7FFFFFFFu;
        const uint64_t a = P.idx ? P.idx[k] : (uint64_t)k;
        const CellOfAtom ca = classify(P, load_pos(P.xyz, a));   // same arithmetic as bin_kernel
        const uint32_t id = ids_local ? k : (uint32_t)a;
        const float4 rec = make_float4(ca.pos.x, ca.pos.y, ca.pos.z, __uint_as_float(id));
        sorted[s + rank] = rec;
        if (vdw) sorted_vdw[s + rank] = vdw[k];
        // cells of more than ORDER_MAX atoms get no Morton order: their "spatial" copy is the reference's order itself (the fused
        // histogram walks it in blocks of ORDER_MAX atoms, hist_kernel), their chunk boxes below the box of the whole cell
        if (!small && want_order) perm[s + rank] = make_float4(rec.x, rec.y, rec.z, __uint_as_float(rank));
        lo[0] = fminf(lo[0], rec.x); hi[0] = fmaxf(hi[0], rec.x);
        lo[1] = fminf(lo[1], rec.y); hi[1] = fmaxf(hi[1], rec.y);
        lo[2] = fminf(lo[2], rec.z); hi[2] = fmaxf(hi[2], rec.z);
    }
    for (int d = 0; d < 3; ++d)
        for (int off = 32; off > 0; off >>= 1) {
            lo[d] = fminf(lo[d], __shfl_xor(lo[d], off, 64));
            hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], off, 64));
        }
    if (lane == 0) {
        aabb[2 * c] = make_float4(lo[0], lo[1], lo[2], 0.f);
        aabb[2 * c + 1] = make_float4(hi[0], hi[1], hi[2], 0.f);
    }
    // origin of the cell for the matrix-core count pass (pair_kernels.hpp, run_count_mfma): centre of the bounding box;
    // .w bounds the distance of any atom of the cell from it (NaN / empty cells give a non-finite bound: that pass
    // then leaves the cell to the exact path)
    const float org[3] = {0.5f * (lo[0] + hi[0]), 0.5f * (lo[1] + hi[1]), 0.5f * (lo[2] + hi[2])};
    if (lane == 0) {
        const float ex = fmaxf(hi[0] - org[0], org[0] - lo[0]), ey = fmaxf(hi[1] - org[1], org[1] - lo[1]),
                    ez = fmaxf(hi[2] - org[2], org[2] - lo[2]);
        cell_org[c] = make_float4(org[0], org[1], org[2], 1.0001f * sqrtf((ex * ex + ey * ey) + ez * ez));
    }
    // (want_order: the spatial order, the chunk boxes and the f16 records serve the count pass and the fused histogram of the
    // fixed-cutoff kinds only - the vdW and `within` searches, grids of 1e5 cells of a few atoms, stop here)
    if (!small && want_order) {
        const uint32_t ub = (s >> 6) + c;
        for (uint32_t k = lane; k * 64u < n; k += 64u) {
            chunk_aabb[2 * (ub + k)] = make_float4(lo[0], lo[1], lo[2], 0.f);
            chunk_aabb[2 * (ub + k) + 1] = make_float4(hi[0], hi[1], hi[2], 0.f);
        }
    }
    if (n == 0 || !small || !want_order) return;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");   // the records placed above are re-read by other lanes of this wave (same CU: no L2 write-back needed)
    __builtin_amdgcn_wave_barrier();
    uint32_t *hist = keys;                          // the sort keys are no longer needed
    for (uint32_t b = lane; b < ORDER_MAX; b += 64u) hist[b] = 0u;
    __builtin_amdgcn_wave_barrier();
    float sc[3];
    for (int d = 0; d < 3; ++d) sc[d] = hi[d] > lo[d] ? 8.0f / (hi[d] - lo[d]) : 0.0f;
    auto cell3 = [&](float v, int d) -> uint32_t {
        const float f = (v - lo[d]) * sc[d];
        return f >= 7.0f ? 7u : (f > 0.0f ? (uint32_t)f : 0u);      // NaN -> 0
    };
    for (uint32_t t = lane; t < n; t += 64u) {
        const float4 p = pos[t];
        const uint32_t x = cell3(p.x, 0), y = cell3(p.y, 1), z = cell3(p.z, 2);
        uint32_t key = 0;
        for (int b = 0; b < 3; ++b) key |= (((x >> b) & 1u) << (3 * b)) | (((y >> b) & 1u) << (3 * b + 1)) | (((z >> b) & 1u) << (3 * b + 2));
        const uint32_t rank = atomicAdd(&hist[key], 1u);
        kr[t] = key | (rank << 16);
    }
    __builtin_amdgcn_wave_barrier();
    {   // exclusive prefix over the 512 bins: 8 consecutive bins per lane
        uint32_t loc[8], sum = 0;
        for (int b = 0; b < 8; ++b) { loc[b] = hist[lane * 8u + b]; sum += loc[b]; }
        uint32_t inc = sum;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(inc, off, 64);
            if ((int)lane >= off) inc += o;
        }
        uint32_t run = inc - sum;
        for (int b = 0; b < 8; ++b) { hist[lane * 8u + b] = run; run += loc[b]; }
    }
    __builtin_amdgcn_wave_barrier();
    for (uint32_t t = lane; t < n; t += 64u) {
        const uint32_t v = kr[t];
        pl[hist[v & 0xFFFFu] + (v >> 16)] = (uint16_t)t;
    }
    __builtin_amdgcn_wave_barrier();
    const uint32_t ubase = (s >> 6) + c;
    for (uint32_t k = 0; k * 64u < n; ++k) {
        const uint32_t m = k * 64u + lane;
        float l3[3] = {INFINITY, INFINITY, INFINITY}, h3[3] = {-INFINITY, -INFINITY, -INFINITY};
        if (m < n) {
            const uint32_t t = pl[m];
            const float4 p = pos[t];
            perm[s + m] = make_float4(p.x, p.y, p.z, __uint_as_float(t));
            {   // f16 hi/lo split of the position relative to the cell origin and of its squared norm (22 bits each)
                const float r[3] = {p.x - org[0], p.y - org[1], p.z - org[2]};
                _Float16 h[3], l[3];
                float eff[3];
                for (int d = 0; d < 3; ++d) {
                    h[d] = (_Float16)r[d];
                    l[d] = (_Float16)(r[d] - (float)h[d]);
                    eff[d] = (float)h[d] + (float)l[d];
                }
                const float nb = (eff[0] * eff[0] + eff[1] * eff[1]) + eff[2] * eff[2];
                const _Float16 nh = (_Float16)nb, nl = (_Float16)(nb - (float)nh);
                auto pk = [](_Float16 a, _Float16 b) -> uint32_t {
                    return (uint32_t)__builtin_bit_cast(unsigned short, a) | ((uint32_t)__builtin_bit_cast(unsigned short, b) << 16);
                };
                // an atom with a NaN / infinite coordinate has a non-finite |b|^2: it pairs with nothing in the reference (its d2 is
                // NaN or inf) but its accumulators would be NaNs of either sign - it gets the record of "an atom past the end"
                // (|b|^2 = 65504, never a hit) here, once, instead of a test per tile in every slot of the count pass
                const bool finite = (__builtin_bit_cast(unsigned short, nh) & 0x7C00u) != 0x7C00u;
                h16[s + t] = finite ? make_uint4(pk(h[0], h[1]), pk(h[2], l[0]), pk(l[1], l[2]), pk(nh, nl))      // in the REFERENCE's cell order
                                    : make_uint4(0u, 0u, 0u, 0x00007BFFu);
            }
            l3[0] = h3[0] = p.x; l3[1] = h3[1] = p.y; l3[2] = h3[2] = p.z;
        }
        for (int d = 0; d < 3; ++d)
            for (int off = 32; off > 0; off >>= 1) {
                l3[d] = fminf(l3[d], __shfl_xor(l3[d], off, 64));
                h3[d] = fmaxf(h3[d], __shfl_xor(h3[d], off, 64));
            }
        if (lane == 0) {
            chunk_aabb[2 * (ubase + k)] = make_float4(l3[0], l3[1], l3[2], 0.f);
            chunk_aabb[2 * (ubase + k) + 1] = make_float4(h3[0], h3[1], h3[2], 0.f);
        }
    }
}
__global__ void __launch_bounds__(64) place_order_kernel(BinParams P, uint32_t ncells, int ids_local,
                                                          const uint32_t *__restrict__ cell_start,
                                                          const uint32_t *__restrict__ tmp_key, const float *__restrict__ vdw,
                                                          float4 *__restrict__ sorted, float *__restrict__ sorted_vdw,
                                                          float4 *__restrict__ aabb, float4 *__restrict__ perm,
                                                          float4 *__restrict__ chunk_aabb, uint4 *__restrict__ h16,
                                                          float4 *__restrict__ cell_org, int want_order, int presorted,
                                                          uint32_t *__restrict__ occupied) {
    if (occupied && threadIdx.x == 0 && blockIdx.x < ncells && cell_start[blockIdx.x + 1] != cell_start[blockIdx.x])
        atomicAdd(occupied + (blockIdx.x & (OCC_WORDS - 1u)), 1u);
    place_order_body(P, ncells, ids_local, cell_start, tmp_key, vdw, sorted, sorted_vdw, aabb, perm, chunk_aabb, h16, cell_org, want_order, presorted);
}
__global__ void __launch_bounds__(64) place_order_frames_kernel(const GridFrame *__restrict__ G, uint32_t ncells, int ids_local) {
    const GridFrame &g = G[blockIdx.y];
    place_order_body(g.P, ncells, ids_local, g.cell_count, g.tmp_key, nullptr, g.sorted, nullptr, g.aabb, g.perm, g.chunk_aabb, g.h16, g.cell_org,
                     /*want_order=*/1, /*presorted=*/0);
}


// ================================================================= scans (exclusive, n elements)

constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = 256 * SCAN_ITEMS;

template <class T>
__device__ __forceinline__ T block_exclusive_scan(T v, T *total) {
    __shared__ T wave_sums[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (int)(blockDim.x >> 6);
    T inc = v;
    for (int off = 1; off < 64; off <<= 1) {
        T o = __shfl_up(inc, off, 64);
        if (lane >= off) inc += o;
    }
    if (lane == 63) wave_sums[wave] = inc;
    __syncthreads();
    T base = 0, tot = 0;
    for (int w = 0; w < nw; ++w) {
        if (w < wave) base += wave_sums[w];
        tot += wave_sums[w];
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

template <class TIn, class TOut>
__global__ void __launch_bounds__(256) scan_tile_kernel(const TIn *in, TOut *out, TOut *block_sums, uint64_t n) {
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
    TOut item[SCAN_ITEMS];
    TOut sum = 0;
    for (int q = 0; q < SCAN_ITEMS; ++q) {
        item[q] = base + q < n ? (TOut)in[base + q] : (TOut)0;
        sum += item[q];
    }
    TOut tot;
    TOut run = block_exclusive_scan<TOut>(sum, &tot);
    for (int q = 0; q < SCAN_ITEMS; ++q) {
        if (base + q < n) out[base + q] = run;
        run += item[q];
    }
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

// Single-launch exclusive scan (decoupled look-back, Merrill & Garland): tiles take tickets in launch order, publish
// their aggregate, and resolve their prefix by walking back over the descriptors of earlier tiles.  A descriptor is
// one 64-bit word: flag in the top two bits (0 not ready, 1 aggregate, 2 inclusive prefix), value below (< 2^62).
// Scans TWO arrays of the same length at once when in2 != nullptr (task slot counts and task hit-history units).
// `state` holds 1 ticket word + 2 descriptors per tile and must be zero on entry.
constexpr unsigned long long LB_MASK = (1ull << 62) - 1ull;
constexpr uint64_t FPLAN_MAX_TASKS = 1ull << 22;                          // plan_tiles_kernel / plan_slots_kernel: plans up to this many entries (+ terminator)
constexpr uint32_t PLAN_GROUP = 128;                                       // plans of more than 1024 tiles: tile totals are summed in groups of this many first (plan_groups_kernel)
__device__ __forceinline__ unsigned long long lb_resolve(unsigned long long *desc, uint32_t tile, unsigned long long tot) {
    if (tile == 0) {
        __hip_atomic_store(&desc[0], (2ull << 62) | tot, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        return 0ull;
    }
    __hip_atomic_store(&desc[tile], (1ull << 62) | tot, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long run = 0;
    for (uint32_t t = tile - 1u;; --t) {
        unsigned long long d;
        do {
            d = __hip_atomic_load(&desc[t], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        } while ((d >> 62) == 0ull);
        run += d & LB_MASK;
        if ((d >> 62) == 2ull) break;
    }
    __hip_atomic_store(&desc[tile], (2ull << 62) | (run + tot), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    return run;
}

template <class TIn1, class TOut1, class TIn2, class TOut2>
__global__ void __launch_bounds__(256) scan_lookback_kernel(const TIn1 *in1, TOut1 *out1, const TIn2 *in2, TOut2 *out2, uint64_t n,
                                                            unsigned long long *state, uint32_t ntiles) {
    __shared__ uint32_t tile_s;
    __shared__ unsigned long long pre_s[2];
    if (threadIdx.x == 0) tile_s = atomicAdd(reinterpret_cast<uint32_t *>(state), 1u);
    __syncthreads();
    const uint32_t tile = tile_s;
    unsigned long long *d1 = state + 1, *d2 = state + 1 + ntiles;
    const uint64_t base = (uint64_t)tile * (blockDim.x * SCAN_ITEMS) + (uint64_t)threadIdx.x * SCAN_ITEMS;
    TOut1 a[SCAN_ITEMS];
    TOut2 b[SCAN_ITEMS];
    TOut1 sa = 0;
    TOut2 sb = 0;
    for (int q = 0; q < SCAN_ITEMS; ++q) {
        a[q] = base + q < n ? (TOut1)in1[base + q] : (TOut1)0;
        sa += a[q];
        b[q] = (in2 && base + q < n) ? (TOut2)in2[base + q] : (TOut2)0;
        sb += b[q];
    }
    TOut1 ta;
    TOut1 ra = block_exclusive_scan<TOut1>(sa, &ta);
    TOut2 tb = 0, rb = 0;
    if (in2) rb = block_exclusive_scan<TOut2>(sb, &tb);
    if (threadIdx.x == 0) pre_s[0] = lb_resolve(d1, tile, (unsigned long long)ta);
    if (threadIdx.x == (blockDim.x > 64u ? 64u : 1u) && in2) pre_s[1] = lb_resolve(d2, tile, (unsigned long long)tb);
    __syncthreads();
    ra += (TOut1)pre_s[0];
    if (in2) rb += (TOut2)pre_s[1];
    for (int q = 0; q < SCAN_ITEMS; ++q) {
        if (base + q < n) {
            out1[base + q] = ra;
            if (in2) out2[base + q] = rb;
        }
        ra += a[q];
        rb += b[q];
    }
}

// state words needed by scan_lookback for n elements (sized for the one-wave tiles of the side stream)
inline size_t lookback_state_words(uint64_t n) { return 1 + 2 * (size_t)((n + 64 * SCAN_ITEMS - 1) / (64 * SCAN_ITEMS)); }

template <class TIn1, class TOut1, class TIn2, class TOut2>
int scan_lookback(molar_hip_ctx *c, const TIn1 *in1, TOut1 *out1, const TIn2 *in2, TOut2 *out2, uint64_t n,
                  unsigned long long *zeroed_state) {
    if (n == 0) return 0;
    const uint32_t block = c->on_side ? 64u : 256u;           // one-wave workgroups on the side stream (place_order_kernel)
    const uint32_t ntiles = (uint32_t)((n + block * SCAN_ITEMS - 1) / (block * SCAN_ITEMS));
    hipLaunchKernelGGL((scan_lookback_kernel<TIn1, TOut1, TIn2, TOut2>), dim3(ntiles), dim3(block), 0, c->stream, in1, out1, in2, out2, n,
                       zeroed_state, ntiles);
    MH_HIP(hipGetLastError());
    return 0;
}

// The plan of a search in TWO launches without a chain between workgroups (round 5; it was plan_kernel -> scan_lookback_kernel ->
// slotmap_kernel, 5 + 13 + 13 us on the critical stream of every frame): plan_tiles_kernel decodes 256 plan entries per workgroup,
// scans their slot counts (and hit-history units) inside the tile and leaves the tile totals; in plan_slots_kernel workgroup b sums
// the b totals in front of it (a few hundred words from L2, as slot_offsets_kernel does), finishes its entries' offsets and writes
// the records of their slots.  (One launch with a decoupled look-back between the tiles was built first: correct, and 56 us - the
// tiles' chain waits for whichever workgroup the dispatcher starts last, and the next frame's grid build on the high-priority
// side stream starts at that very moment.)  plan_tiles_kernel also zeroes the slot counters and writes the parameter block, as
// plan_kernel does; plan_slots_kernel also blanks the records between the real slot count and the host's bound (spread over all
// workgroups: one workgroup alone took 45 us for 5e4 of them).
template <int KIND>
__global__ void __launch_bounds__(256) plan_tiles_kernel(SearchParams P, uint32_t *__restrict__ local_first, TaskDesc *__restrict__ task_desc,
                                                         unsigned long long *__restrict__ local_moff, uint32_t fast_kind,
                                                         uint32_t *__restrict__ slot_cnt, uint64_t nslot_cnt,
                                                         unsigned long long *__restrict__ tile_tot, SearchParams *__restrict__ params_dst) {
    const uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    for (uint64_t w = t; w < nslot_cnt; w += (uint64_t)gridDim.x * 256u) slot_cnt[w] = 0u;   // the count kernel writes the slots that exist
    if (params_dst && blockIdx.x == 0) {
        const uint32_t *src = (const uint32_t *)__builtin_amdgcn_kernarg_segment_ptr();     // P is this kernel's first argument
        uint32_t *d = (uint32_t *)params_dst;
        for (uint32_t w = threadIdx.x; w < (uint32_t)(sizeof(SearchParams) / 4u); w += 256u) d[w] = src[w];
    }
    uint32_t nb = 0;
    unsigned long long mu = 0;
    if (t < P.ntasks) {
        const Task T = decode_task<KIND, false>(P, t);
        nb = T.valid ? (T.n1 + T.rps - 1u) / T.rps : 0u;
        const uint32_t nch = (T.n2 + 63u) >> 6;
        mu = (fast_kind && P.use_box && T.wrap != 0u && nch <= (uint32_t)KREG) ? (unsigned long long)nb * 2u * nch : 0ull;   // as plan_kernel
        TaskDesc d;
        d.a0 = T.a0; d.n1 = T.n1; d.b0 = T.b0; d.n2 = T.n2; d.cb = T.cb;
        d.flags = T.wrap | (T.tri ? 0x100u : 0u) | (T.valid ? 0x200u : 0u) | (T.wrap_b << 12) | (T.rps << 16);
        d.pad0 = nb;                              // slots of the entry (plan_slots_kernel)
        d.pad1 = 0u;
        task_desc[t] = d;
    }
    uint32_t tot_nb;
    const uint32_t first = block_exclusive_scan<uint32_t>(nb, &tot_nb);
    unsigned long long tot_mu = 0ull, m0 = 0ull;
    if (fast_kind) m0 = block_exclusive_scan<unsigned long long>(mu, &tot_mu);
    if (t <= P.ntasks) {                         // offsets inside the tile (t == ntasks: the terminator)
        local_first[t] = first;
        if (fast_kind) local_moff[t] = m0;
    }
    if (threadIdx.x == 0) {
        tile_tot[2 * blockIdx.x] = tot_nb;
        tile_tot[2 * blockIdx.x + 1] = tot_mu;
    }
}

// Plans of more than 1024 tiles (2^18 entries: a 1M-atom frame below 0.72 nm): every workgroup of plan_slots_kernel summing every
// tile total in front of it would read ntiles^2 / 2 words (2.3e6 entries: 36 000 workgroups x 9 000 tiles), so the totals of
// groups of PLAN_GROUP tiles are formed first - one more launch of ntiles / 128 single-wave workgroups - and a workgroup adds the
// groups in front of its own and the tiles of its own group in front of its tile.
static __global__ void __launch_bounds__(64) plan_groups_kernel(uint32_t ntiles, const unsigned long long *__restrict__ tile_tot,
                                                                unsigned long long *__restrict__ group_tot) {
    unsigned long long n = 0, m = 0;
    for (uint32_t k = threadIdx.x; k < PLAN_GROUP; k += 64u) {
        const uint32_t b = blockIdx.x * PLAN_GROUP + k;
        if (b < ntiles) { n += tile_tot[2 * b]; m += tile_tot[2 * b + 1]; }
    }
    for (int off = 32; off > 0; off >>= 1) {
        n += __shfl_xor(n, off, 64);
        m += __shfl_xor(m, off, 64);
    }
    if (threadIdx.x == 0) { group_tot[2 * blockIdx.x] = n; group_tot[2 * blockIdx.x + 1] = m; }
}

constexpr uint32_t PLAN_SUB = 64;          // plan entries per workgroup of plan_slots_kernel (a quarter of a tile)
static __global__ void __launch_bounds__(256) plan_slots_kernel(uint64_t ntasks, uint32_t ntiles, const uint32_t *__restrict__ local_first,
                                                                const unsigned long long *__restrict__ local_moff,
                                                                uint32_t *__restrict__ task_first, const TaskDesc *__restrict__ task_desc,
                                                                unsigned long long *__restrict__ task_moff,      // NULL: no hit history
                                                                const unsigned long long *__restrict__ tile_tot,
                                                                SlotDesc *__restrict__ slot_desc, uint64_t nslots_bound,
                                                                unsigned long long *__restrict__ sizes_host,    // pinned, or NULL
                                                                const unsigned long long *__restrict__ group_tot) {  // NULL: at most 1024 tiles
    __shared__ unsigned long long part[3][4];
    const uint32_t tile = blockIdx.x / (256u / PLAN_SUB), sub = blockIdx.x % (256u / PLAN_SUB);
    unsigned long long pn = 0, pm = 0, all_n = 0;           // slots / hit-history units in front of this tile, slots of the whole plan
    if (!group_tot) {
        for (uint32_t b = threadIdx.x; b < ntiles; b += 256u) {
            const unsigned long long n = tile_tot[2 * b];
            all_n += n;
            if (b < tile) {
                pn += n;
                pm += tile_tot[2 * b + 1];
            }
        }
    } else {
        const uint32_t ngroups = (ntiles + PLAN_GROUP - 1u) / PLAN_GROUP, g = tile / PLAN_GROUP;
        for (uint32_t b = threadIdx.x; b < ngroups; b += 256u) {
            const unsigned long long n = group_tot[2 * b];
            all_n += n;
            if (b < g) {
                pn += n;
                pm += group_tot[2 * b + 1];
            }
        }
        for (uint32_t b = g * PLAN_GROUP + threadIdx.x; b < tile; b += 256u) {
            pn += tile_tot[2 * b];
            pm += tile_tot[2 * b + 1];
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        pn += __shfl_xor(pn, off, 64);
        pm += __shfl_xor(pm, off, 64);
        all_n += __shfl_xor(all_n, off, 64);
    }
    if ((threadIdx.x & 63u) == 0u) { part[0][threadIdx.x >> 6] = pn; part[1][threadIdx.x >> 6] = pm; part[2][threadIdx.x >> 6] = all_n; }
    __syncthreads();
    const uint32_t base_n = (uint32_t)((part[0][0] + part[0][1]) + (part[0][2] + part[0][3]));
    const unsigned long long base_m = (part[1][0] + part[1][1]) + (part[1][2] + part[1][3]);
    const unsigned long long total = (part[2][0] + part[2][1]) + (part[2][2] + part[2][3]);
    // this workgroup's entries in LDS: offsets inside the tile, descriptors, first hit-history unit
    __shared__ uint32_t lfirst[PLAN_SUB + 1];
    __shared__ TaskDesc ldesc[PLAN_SUB];
    __shared__ unsigned long long lm0[PLAN_SUB];
    const uint32_t tile_n = (uint32_t)tile_tot[2 * tile];
    if (threadIdx.x <= PLAN_SUB) {
        const uint64_t t = (uint64_t)tile * 256u + sub * PLAN_SUB + threadIdx.x;        // (thread PLAN_SUB: the next workgroup's first entry)
        const bool inside = threadIdx.x < PLAN_SUB || sub + 1u < 256u / PLAN_SUB;     // ... which lies in this tile
        uint32_t local = tile_n;                 // past the terminator / past the tile: the tile's end
        unsigned long long m0 = ~0ull >> 1;
        TaskDesc d;
        d.a0 = d.n1 = d.b0 = d.n2 = d.cb = d.flags = d.pad0 = d.pad1 = 0u;
        if (inside && t <= ntasks) {
            local = local_first[t];
            if (threadIdx.x < PLAN_SUB) {
                if (task_moff) m0 = base_m + local_moff[t];
                if (t < ntasks) d = task_desc[t];
            }
        }
        lfirst[threadIdx.x] = local;
        if (threadIdx.x < PLAN_SUB) {
            ldesc[threadIdx.x] = d;
            lm0[threadIdx.x] = m0;
        }
    }
    __syncthreads();
    if (threadIdx.x < PLAN_SUB) {
        const uint64_t t = (uint64_t)tile * 256u + sub * PLAN_SUB + threadIdx.x;
        if (t <= ntasks) {
            task_first[t] = base_n + lfirst[threadIdx.x];
            if (task_moff) task_moff[t] = lm0[threadIdx.x];
            if (t == ntasks && sizes_host) {
                sizes_host[1] = task_moff ? lm0[threadIdx.x] : 0ull;      // hit-history units of this search
                sizes_host[2] = base_n + lfirst[threadIdx.x];             // its slots
            }
        }
    }
    // One record per slot (as slotmap_kernel), written as the workgroup's contiguous run of 16-byte words: word i belongs to
    // slot s0 + i / 3 of the tile, whose entry is the last one that starts at or before it.
    const uint32_t s0 = lfirst[0], s1 = lfirst[PLAN_SUB];
    uint4 *out = reinterpret_cast<uint4 *>(slot_desc) + 3 * ((size_t)base_n + s0);
    for (uint32_t i = threadIdx.x; i < 3u * (s1 - s0); i += 256u) {
        const uint32_t sl = s0 + i / 3u, part3 = i % 3u;
        uint32_t lo = 0u, hi = PLAN_SUB;         // largest k with lfirst[k] <= sl
        while (hi - lo > 1u) {
            const uint32_t mid = (lo + hi) >> 1;
            if (lfirst[mid] <= sl) lo = mid;
            else hi = mid;
        }
        const TaskDesc d = ldesc[lo];
        const uint32_t q = sl - lfirst[lo], rps = d.flags >> 16, nch = (d.n2 + 63u) >> 6;
        uint4 w;
        if (part3 == 0u) w = make_uint4(d.a0, d.n1, d.b0, d.n2);
        else if (part3 == 1u) w = make_uint4(d.cb, d.flags, q * rps, 0u);
        else {
            const unsigned long long mo = task_moff ? lm0[lo] + (unsigned long long)q * 2u * nch : lm0[lo];
            w = make_uint4((uint32_t)mo, (uint32_t)(mo >> 32), 0u, 0u);
        }
        out[i] = w;
    }
    // slots between the real count and the host's bound: waves launched for them leave at once (every workgroup blanks its share)
    if (total <= nslots_bound) {
        uint4 *blank = reinterpret_cast<uint4 *>(slot_desc) + 3 * (size_t)total;
        const uint64_t nw = 3ull * (nslots_bound + 1ull - total);
        for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < nw; i += (uint64_t)gridDim.x * 256u)
            blank[i] = (i % 3ull == 2ull) ? make_uint4(0xFFFFFFFFu, 0x7FFFFFFFu, 0u, 0u) : make_uint4(0u, 0u, 0u, 0u);
    }
}

template <class T>
__global__ void __launch_bounds__(256) scan_sums_kernel(T *sums, uint64_t nb) {
    __shared__ T carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (uint64_t start = 0; start < nb; start += 256) {
        const uint64_t i = start + threadIdx.x;
        const T v = i < nb ? sums[i] : (T)0;
        T tot;
        const T ex = block_exclusive_scan<T>(v, &tot);
        const T carry = carry_s;
        if (i < nb) sums[i] = carry + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + tot;
        __syncthreads();
    }
}

template <class T>
__global__ void __launch_bounds__(256) scan_add_kernel(T *out, const T *sums, uint64_t n) {
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
    const T add = sums[blockIdx.x];
    for (int q = 0; q < SCAN_ITEMS; ++q)
        if (base + q < n) out[base + q] += add;
}

// small n: one workgroup walks the array in 1024-element steps (a single launch instead of three)
template <class TIn, class TOut>
__global__ void __launch_bounds__(256) scan_small_kernel(const TIn *in, TOut *out, uint32_t n) {
    __shared__ TOut carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t start = 0; start < n; start += 1024u) {
        const uint32_t base = start + threadIdx.x * 4u;
        TOut item[4], sum = 0;
        for (int q = 0; q < 4; ++q) {
            item[q] = base + q < n ? (TOut)in[base + q] : (TOut)0;
            sum += item[q];
        }
        TOut tot;
        TOut run = carry_s + block_exclusive_scan<TOut>(sum, &tot);
        for (int q = 0; q < 4; ++q) {
            if (base + q < n) out[base + q] = run;
            run += item[q];
        }
        __syncthreads();
        if (threadIdx.x == 0) carry_s += tot;
        __syncthreads();
    }
}

// the same with ONE wave (512 elements per step): a one-wave workgroup finds a wave slot on a chip that another stream's
// kernel keeps full (grid build on the side stream, see place_order_kernel)
template <class TIn, class TOut>
__global__ void __launch_bounds__(64) scan_small_wave_kernel(const TIn *in, TOut *out, uint32_t n) {
    TOut carry = 0;
    for (uint32_t start = 0; start < n; start += 512u) {
        const uint32_t base = start + threadIdx.x * 8u;
        TOut item[8], sum = 0;
        for (int q = 0; q < 8; ++q) {
            item[q] = base + q < n ? (TOut)in[base + q] : (TOut)0;
            sum += item[q];
        }
        TOut inc = sum;
        for (int off = 1; off < 64; off <<= 1) {
            const TOut o = __shfl_up(inc, off, 64);
            if ((int)threadIdx.x >= off) inc += o;
        }
        TOut run = carry + inc - sum;
        for (int q = 0; q < 8; ++q) {
            if (base + q < n) out[base + q] = run;
            run += item[q];
        }
        carry += __shfl(inc, 63, 64);
    }
}
// cell counts -> cell starts of every frame of a batch, in place: one wave per frame (blockIdx.x)
__global__ void __launch_bounds__(64) scan_frames_kernel(const GridFrame *__restrict__ G, uint32_t n) {
    uint32_t *a = G[blockIdx.x].cell_count;
    uint32_t carry = 0;
    for (uint32_t start = 0; start < n; start += 512u) {
        const uint32_t base = start + threadIdx.x * 8u;
        uint32_t item[8], sum = 0;
        for (int q = 0; q < 8; ++q) {
            item[q] = base + q < n ? a[base + q] : 0u;
            sum += item[q];
        }
        uint32_t inc = sum;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(inc, off, 64);
            if ((int)threadIdx.x >= off) inc += o;
        }
        uint32_t run = carry + inc - sum;
        for (int q = 0; q < 8; ++q) {
            if (base + q < n) a[base + q] = run;
            run += item[q];
        }
        carry += __shfl(inc, 63, 64);
    }
}

template <class TIn, class TOut>
int exclusive_scan(molar_hip_ctx *c, const TIn *in, TOut *out, uint64_t n) {
    if (n == 0) return 0;
    if (n <= 8192ull) {
        if (c->on_side) hipLaunchKernelGGL((scan_small_wave_kernel<TIn, TOut>), dim3(1), dim3(64), 0, c->stream, in, out, (uint32_t)n);
        else
        hipLaunchKernelGGL((scan_small_kernel<TIn, TOut>), dim3(1), dim3(256), 0, c->stream, in, out, (uint32_t)n);
        MH_HIP(hipGetLastError());
        return 0;
    }
    const uint64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    mh::DevBuf &tmp = c->on_side ? c->scan_tmp_side : c->scan_tmp;   // the two streams may scan at the same time
    MH_TRY(tmp.reserve(nb * sizeof(TOut)));
    TOut *sums = tmp.as<TOut>();
    hipLaunchKernelGGL((scan_tile_kernel<TIn, TOut>), dim3((unsigned)nb), dim3(256), 0, c->stream, in, out, sums, n);
    if (nb > 1) {
        hipLaunchKernelGGL((scan_sums_kernel<TOut>), dim3(1), dim3(256), 0, c->stream, sums, nb);
        hipLaunchKernelGGL((scan_add_kernel<TOut>), dim3((unsigned)nb), dim3(256), 0, c->stream, out, sums, n);
    }
    MH_HIP(hipGetLastError());
    return 0;
}

// Output offsets of the slots in two launches instead of the general scan's three (tile scan, scan of the block sums,
// add - the last one alone took 39 us per frame in the round-3 trace): tile_sums_kernel adds up the counts of every tile
// of 256 slots; in slot_offsets_kernel workgroup b owns tile b, takes as its base the sum of the b tile totals in front
// of it (read from L2 by 256 threads: ~1100 tiles = 9 KB on the headline frame), and one block scan over the tile's own
// 256 counts gives every slot its offset.  With `sizes_host` the grand total goes straight to the host's pinned block
// (no copy command behind the fill pass).  (The tile totals as 64-bit atomics of the count pass itself - one launch
// instead of two - cost that pass 70-85 us: 256 device-scope atomics per address from eight XCDs; measured, removed.)
// Work grows with the square of the tile count: plans above SLOT_SCAN_MAX_TILES take exclusive_scan.
constexpr uint64_t SLOT_SCAN_MAX_TILES = 4096;
__global__ void __launch_bounds__(256) tile_sums_kernel(const uint32_t *__restrict__ slot_cnt, unsigned long long *__restrict__ tile_sum,
                                                        uint64_t n) {
    __shared__ unsigned long long part[4];
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    unsigned long long v = i < n ? (unsigned long long)slot_cnt[i] : 0ull;
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63u) == 0u) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

__global__ void __launch_bounds__(256) slot_offsets_kernel(const uint32_t *__restrict__ slot_cnt,
                                                           const unsigned long long *__restrict__ tile_sum,
                                                           unsigned long long *__restrict__ slot_base, uint64_t n,
                                                           unsigned long long *__restrict__ sizes_host,
                                                           const uint32_t *__restrict__ occ0, const uint32_t *__restrict__ occ1) {
    __shared__ unsigned long long part[4];
    unsigned long long pre = 0;
    for (uint32_t t = threadIdx.x; t < blockIdx.x; t += 256u) pre += tile_sum[t];
    for (int off = 32; off > 0; off >>= 1) pre += __shfl_xor(pre, off, 64);
    if ((threadIdx.x & 63u) == 0u) part[threadIdx.x >> 6] = pre;
    __syncthreads();
    const unsigned long long base = (part[0] + part[1]) + (part[2] + part[3]);
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    const unsigned long long v = i < n ? (unsigned long long)slot_cnt[i] : 0ull;
    unsigned long long tot;
    const unsigned long long ex = block_exclusive_scan<unsigned long long>(v, &tot);
    if (i < n) slot_base[i] = base + ex;
    if (i + 1 == n && sizes_host) {
        sizes_host[0] = base + ex;      // slot_cnt[n - 1] is the terminator: its offset is the grand total
        // the occupied cells of the two grids (counted by their placement kernels), for the host's choice of kernels next time
        uint32_t o0 = 0u, o1 = 0u;
        for (uint32_t k = 0; k < OCC_WORDS; ++k) {
            o0 += occ0 ? occ0[k] : 0u;
            o1 += occ1 ? occ1[k] : 0u;
        }
        sizes_host[3] = ((unsigned long long)o1 << 32) | o0;
    }
}

// slot counts -> output offsets (n = slots + 1 terminator), the grand total also to `sizes_host` (pinned, may be null)
int scan_slot_counts(molar_hip_ctx *c, unsigned long long *sizes_host) {
    const uint64_t n = c->nslots_bound + 1;
    const uint64_t ntiles = (n + 255) / 256;
    const uint32_t *occ[2] = {nullptr, nullptr};
    if (sizes_host) {
        const size_t ncells = (size_t)c->dims[0] * c->dims[1] * c->dims[2];
        for (int s = 0; s < (c->kind == MOLAR_HIP_SEARCH_SINGLE ? 1 : 2); ++s)
            if (c->set[s].cell_count.cap >= (ncells + 1 + OCC_WORDS) * 4) occ[s] = c->set[s].cell_count.as<uint32_t>() + ncells + 1;
    }
    if (ntiles <= SLOT_SCAN_MAX_TILES && !c->env_no_tile_sum) {
        MH_TRY(c->tile_sum.reserve(ntiles * 8));
        hipLaunchKernelGGL(tile_sums_kernel, dim3((unsigned)ntiles), dim3(256), 0, c->stream, c->slot_cnt.as<uint32_t>(),
                           c->tile_sum.as<unsigned long long>(), n);
        hipLaunchKernelGGL(slot_offsets_kernel, dim3((unsigned)ntiles), dim3(256), 0, c->stream, c->slot_cnt.as<uint32_t>(),
                           c->tile_sum.as<unsigned long long>(), c->slot_base.as<unsigned long long>(), n, sizes_host, occ[0], occ[1]);
        MH_HIP(hipGetLastError());
        return 0;
    }
    MH_TRY((exclusive_scan<uint32_t, unsigned long long>(c, c->slot_cnt.as<uint32_t>(), c->slot_base.as<unsigned long long>(), n)));
    if (sizes_host)
        MH_HIP(hipMemcpyAsync(sizes_host, c->slot_base.as<unsigned long long>() + c->nslots_bound, 8, hipMemcpyDefault, c->stream));
    return 0;
}

// ================================================================= bounding box / max reductions

__device__ __forceinline__ uint32_t f2ord(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
inline float ord2f(uint32_t o) {
    const uint32_t u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
inline uint32_t f2ord_host(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// compute_min_max (distance_search.rs:602-616): mm[0..2] = min, mm[3..5] = max, as ordered uints,
// pre-seeded by the host with 0.0 (the reference seeds with zeros, so the box contains the origin).
__global__ void __launch_bounds__(256) minmax_kernel(const float *__restrict__ xyz, const uint64_t *__restrict__ idx,
                                                     uint32_t n, uint32_t *__restrict__ mm) {
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t k = blockIdx.x * 256u + threadIdx.x; k < n; k += gridDim.x * 256u) {
        const uint64_t a = idx ? idx[k] : (uint64_t)k;
        const float *q = xyz + 3 * a;
        for (int d = 0; d < 3; ++d) {
            const float v = q[d];
            if (v < lo[d]) lo[d] = v;      // NaN never compares true, as in the reference
            if (v > hi[d]) hi[d] = v;
        }
    }
    for (int d = 0; d < 3; ++d) {
        for (int off = 32; off > 0; off >>= 1) {
            lo[d] = fminf(lo[d], __shfl_xor(lo[d], off, 64));
            hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], off, 64));
        }
    }
    if ((threadIdx.x & 63) == 0) {
        for (int d = 0; d < 3; ++d) {
            if (lo[d] != INFINITY) atomicMin(&mm[d], f2ord(lo[d]));
            if (hi[d] != -INFINITY) atomicMax(&mm[3 + d], f2ord(hi[d]));
        }
    }
}

// vdw.iter().cloned().reduce(Float::max) (distance_search.rs:781-782) of both sets in one launch (blockIdx.y: the set), one
// atomic per workgroup (it was one per wave of up to 1024 workgroups per set: 4096 atomics on one word, 18 us per set)
__global__ void __launch_bounds__(256) fmax_kernel(const float *__restrict__ v1, uint32_t n1, const float *__restrict__ v2, uint32_t n2,
                                                   uint32_t *__restrict__ out) {
    const float *v = blockIdx.y ? v2 : v1;
    const uint32_t n = blockIdx.y ? n2 : n1;
    float m = -INFINITY;
    bool any = false;
    for (uint32_t k = blockIdx.x * 256u + threadIdx.x; k < n; k += gridDim.x * 256u) {
        const float x = v[k];
        if (x == x) {
            m = fmaxf(m, x);
            any = true;
        }
    }
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    const bool wave_any = __ballot(any) != 0ull;   // by the whole wave: inside the branch below only lane 0 would vote
    __shared__ float wm[4];
    __shared__ uint32_t wa[4];
    if ((threadIdx.x & 63) == 0) {
        wm[threadIdx.x >> 6] = m;
        wa[threadIdx.x >> 6] = wave_any ? 1u : 0u;
    }
    __syncthreads();
    if (threadIdx.x == 0 && (wa[0] | wa[1] | wa[2] | wa[3])) {
        float r = -INFINITY;
        for (int w = 0; w < 4; ++w)
            if (wa[w]) r = fmaxf(r, wm[w]);
        atomicMax(out + blockIdx.y, f2ord(r));
    }
}

// u32 (i,j) pairs -> separate usize arrays (Vec<(usize,usize,Float)> split by field)
__global__ void __launch_bounds__(256) widen_pairs_kernel(const uint2 *__restrict__ pairs, uint64_t n,
                                                          unsigned long long *__restrict__ oi,
                                                          unsigned long long *__restrict__ oj) {
    for (uint64_t k = (uint64_t)blockIdx.x * 256u + threadIdx.x; k < n; k += (uint64_t)gridDim.x * 256u) {
        const uint2 p = pairs[k];
        if (oi) oi[k] = p.x;
        if (oj) oj[k] = p.y;
    }
}

__global__ void __launch_bounds__(256) widen_ids_kernel(const uint32_t *__restrict__ ids, uint64_t n,
                                                        unsigned long long *__restrict__ out) {
    for (uint64_t k = (uint64_t)blockIdx.x * 256u + threadIdx.x; k < n; k += (uint64_t)gridDim.x * 256u)
        out[k] = ids[k];
}

// The parameter block of the pair kernels travels as a kernel argument and is written to device memory by the
// GPU itself (read straight from the kernarg segment: no by-value struct indexing, no scratch copy).
__global__ void upload_params_kernel(SearchParams P, SearchParams *dst) {
    const uint32_t *src = (const uint32_t *)__builtin_amdgcn_kernarg_segment_ptr();
    uint32_t *d = (uint32_t *)dst;
    for (uint32_t t = threadIdx.x; t < (uint32_t)(sizeof(SearchParams) / 4u); t += blockDim.x) d[t] = src[t];
}

// ================================================================= host orchestration

// Grid::from_cutoff_and_extents (distance_search.rs:103-110)
int dims_from_extents(molar_hip_ctx *c, float cutoff, const float ext[3]) {
    double cells = 1.0;
    for (int d = 0; d < 3; ++d) {
        const float q = std::floor(ext[d] / cutoff);
        uint64_t s = (q > 0.0f) ? (q >= 4.0e9f ? 4000000000ull : (uint64_t)q) : 0ull;   // `as usize` saturates
        if (s < 1) s = 1;
        c->dims[d] = (uint32_t)s;
        cells *= (double)s;
    }
    if (cells > 1.0e8)
        return fail(MOLAR_HIP_ERR_TOO_LARGE, "search grid %u x %u x %u has too many cells (cutoff %g)", c->dims[0],
                    c->dims[1], c->dims[2], (double)cutoff);
    return 0;
}

// Frames of small cells (contact / hydrogen-bond cutoffs) of the fixed-cutoff kinds go to pair_small.hip: 16 lanes per slot up to 13
// atoms per cell on average, 32 up to 19, 0 = the regular kernels (1M atoms: 0.30 / 0.35 / 0.40 / 0.45 / 0.50 nm take 1.30 / 1.00 /
// 0.89 / 1.01 / 0.99 ms per frame against 2.91 / 2.14 / 1.66 / 1.30 / 1.07; at 0.55 nm - 23 atoms per cell - the regular kernels win).
// Decided from the sets' sizes and the grid alone: the grid build (no spatial order for these) and both passes agree on it.
int small_cell_lanes(const molar_hip_ctx *c) {
#ifdef MH_NO_SMALL_CELLS
    return 0;           // (A/B builds: every frame through the regular kernels)
#endif
    if (c->kind != MOLAR_HIP_SEARCH_SINGLE && c->kind != MOLAR_HIP_SEARCH_DOUBLE) return 0;
    const uint64_t ncells = (uint64_t)c->dims[0] * c->dims[1] * c->dims[2];
    // atoms per OCCUPIED cell where the last search of this shape has told (prepare_search latches occ_use), per cell otherwise
    int lanes = 16;
    const int nsets = c->kind == MOLAR_HIP_SEARCH_SINGLE ? 1 : 2;
    for (int s = 0; s < nsets; ++s) {
        const uint64_t cells = c->occ_use[s] ? std::min<uint64_t>(c->occ_use[s], ncells) : ncells;
        const uint64_t n = c->set[s].n;
        const int l = n <= 13ull * cells ? 16 : (n <= 19ull * cells ? 32 : 0);
        lanes = (l == 0 || lanes == 0) ? 0 : std::max(lanes, l);
    }
    return lanes;
}

// the shape a grid's occupancy is remembered for
static unsigned long long occ_key_of(const molar_hip_ctx *c) {
    unsigned long long h = 1469598103934665603ull;
    auto mix = [&](unsigned long long v) { h = (h ^ v) * 1099511628211ull; };
    mix((unsigned long long)c->kind + 1u); mix(c->set[0].n); mix(c->kind == MOLAR_HIP_SEARCH_SINGLE ? 0u : c->set[1].n);
    mix(c->dims[0]); mix(c->dims[1]); mix(c->dims[2]); mix(c->use_box ? 1u : 2u);
    return h ? h : 1ull;
}
static void occ_latch(molar_hip_ctx *c) {
    const bool known = c->occ_valid && c->occ_key == occ_key_of(c);
    c->occ_use[0] = known ? c->occ_cells[0] : 0u;
    c->occ_use[1] = known ? c->occ_cells[1] : 0u;
}
static void occ_note(molar_hip_ctx *c, unsigned long long key, unsigned long long packed) {
    if (!key || !(uint32_t)packed) return;          // (a search that did not report: the hint stays as it is)
    c->occ_key = key;
    c->occ_cells[0] = (uint32_t)packed;
    c->occ_cells[1] = (uint32_t)(packed >> 32);
    c->occ_valid = true;
}

// Does the count pass of this search record hit history for the fill pass to replay?  The fixed-cutoff kinds do - except on
// frames of small cells: pair_small.hip evaluates in both passes and never touches the buffer, so its plan accounts for none
// (a plan that did made the first resident frame of such a trajectory grow the buffer and run count and fill again for nothing).
static bool records_hit_history(const molar_hip_ctx *c) {
    return (c->kind == MOLAR_HIP_SEARCH_SINGLE || c->kind == MOLAR_HIP_SEARCH_DOUBLE) && small_cell_lanes(c) == 0;
}

// what the grid kernels take of the request (the context's cached search) and of a set
static BinParams grid_bin_params(const molar_hip_ctx *c, const GridSet &S) {
    BinParams P{};
    P.xyz = S.d_xyz;
    P.idx = S.d_idx;
    P.n = S.n;
    P.dx = c->dims[0];
    P.dy = c->dims[1];
    P.dz = c->dims[2];
    P.pbc = c->pbc;
    P.use_box = c->use_box ? 1u : 0u;
    for (int d = 0; d < 3; ++d) {
        P.lower[d] = c->lower[d];
        P.upper[d] = c->upper[d];
    }
    P.box = c->box;
    return P;
}

// one counter per 128-byte line while binning, as long as that stays small (<= 64 MB) and cells are crowded
static uint32_t grid_pad_shift(uint32_t n, uint32_t ncells) { return (n && ncells <= (1u << 19) && (uint64_t)n >= 8ull * ncells) ? 5u : 0u; }

static int grid_reserve(GridSet &S, uint32_t ncells) {
    MH_TRY(S.key.reserve((size_t)(S.n ? S.n : 1) * 4));
    MH_TRY(S.cell_count.reserve((size_t)(ncells + 1 + OCC_WORDS) * 4));      // cell starts, then the occupied-cell counters
    MH_TRY(S.cursor.reserve((size_t)(S.n ? S.n : 1) * 4));   // arrival order of each atom in its cell
    MH_TRY(S.tmp_key.reserve((size_t)(S.n ? S.n : 1) * 4));
    MH_TRY(S.sorted.reserve((size_t)(S.n ? S.n : 1) * sizeof(float4)));
    if (S.d_vdw) MH_TRY(S.sorted_vdw.reserve((size_t)(S.n ? S.n : 1) * 4));
    MH_TRY(S.aabb.reserve((size_t)ncells * 2 * sizeof(float4)));
    MH_TRY(S.perm.reserve((size_t)(S.n ? S.n : 1) * 16));
    MH_TRY(S.chunk_aabb.reserve(((size_t)S.n / 64 + ncells + 1) * 2 * sizeof(float4)));
    MH_TRY(S.h16.reserve((size_t)(S.n ? S.n : 1) * 16));
    MH_TRY(S.cell_org.reserve((size_t)(ncells + 1) * 16));
    const uint32_t pad_shift = grid_pad_shift(S.n, ncells);
    if (pad_shift) MH_TRY(S.cnt_pad.reserve(((size_t)ncells << pad_shift) * 4));
    return 0;
}

int build_grid(molar_hip_ctx *c, GridSet &S, int ids_local) {
    Prof prof(c, 0);
    const uint32_t ncells = c->dims[0] * c->dims[1] * c->dims[2];
    const BinParams P = grid_bin_params(c, S);
    MH_TRY(S.key.reserve((size_t)(S.n ? S.n : 1) * 4));
    MH_TRY(S.cell_count.reserve((size_t)(ncells + 1 + OCC_WORDS) * 4));      // cell starts, then the occupied-cell counters
    MH_TRY(S.cursor.reserve((size_t)(S.n ? S.n : 1) * 4));   // arrival order of each atom in its cell
    MH_TRY(S.tmp_key.reserve((size_t)(S.n ? S.n : 1) * 4));
    MH_TRY(S.sorted.reserve((size_t)(S.n ? S.n : 1) * sizeof(float4)));
    if (S.d_vdw) MH_TRY(S.sorted_vdw.reserve((size_t)(S.n ? S.n : 1) * 4));
    MH_TRY(S.aabb.reserve((size_t)ncells * 2 * sizeof(float4)));
    MH_TRY(S.perm.reserve((size_t)(S.n ? S.n : 1) * 16));
    MH_TRY(S.chunk_aabb.reserve(((size_t)S.n / 64 + ncells + 1) * 2 * sizeof(float4)));
    MH_TRY(S.h16.reserve((size_t)(S.n ? S.n : 1) * 16));
    MH_TRY(S.cell_org.reserve((size_t)(ncells + 1) * 16));
    // one counter per 128-byte line while that stays small (<= 64 MB) and cells are crowded
    const uint32_t pad_shift = grid_pad_shift(S.n, ncells);
    const size_t npad = pad_shift ? ((size_t)ncells << pad_shift) : 0;
    if (pad_shift) MH_TRY(S.cnt_pad.reserve(npad * 4));
    {
        const size_t nz = (size_t)ncells + 1 + npad;
        // workgroup size: one wave on the side stream (see place_order_kernel), four otherwise
        const unsigned bs = c->on_side ? 64u : 256u;
        const unsigned zb = (unsigned)std::min<size_t>((nz + bs - 1) / bs, 2048u * (256u / bs));
        hipLaunchKernelGGL(zero2_kernel, dim3(zb), dim3(bs), 0, c->stream, S.cell_count.as<uint32_t>(), (size_t)ncells + 1 + OCC_WORDS,
                           S.cnt_pad.as<uint32_t>(), npad);
    }
    if (S.n) {
        const unsigned bs = c->on_side ? 64u : 256u;
        const unsigned nb = (S.n + bs - 1u) / bs;
        uint32_t *counters = pad_shift ? S.cnt_pad.as<uint32_t>() : S.cell_count.as<uint32_t>();
        // (on the side stream, beside the pair kernels of the frame in flight: there the privatised form wins - 692-695 against
        // 699-706 frames/s in three alternations; alone it is 10 us SLOWER on the 1M-atom frame - 123 workgroups - so searches
        // that build their grid on the main stream keep one atomic per atom)
        const bool tile_ok = !c->env_no_bin_tile && c->on_side && ncells <= BIN_TILE_MAX_CELLS && (uint64_t)S.n >= 16ull * ncells;
        // (tiles of 8192 atoms for the 1M-atom frame; at 250k atoms 31 of those are too few - the grid's span grew from 0.12 to
        // 0.35 ms - so smaller frames take tiles of 2048)
        if (tile_ok && S.n >= (1u << 19))
            hipLaunchKernelGGL(bin_tile_kernel<32>, dim3((S.n + 256u * 32u - 1u) / (256u * 32u)), dim3(256), (size_t)ncells * 4, c->stream,
                               P, S.key.as<uint32_t>(), S.cursor.as<uint32_t>(), counters, pad_shift, ncells);
        else if (tile_ok && S.n >= (1u << 17))
            hipLaunchKernelGGL(bin_tile_kernel<8>, dim3((S.n + 256u * 8u - 1u) / (256u * 8u)), dim3(256), (size_t)ncells * 4, c->stream,
                               P, S.key.as<uint32_t>(), S.cursor.as<uint32_t>(), counters, pad_shift, ncells);
        else
        hipLaunchKernelGGL(bin_kernel, dim3(nb), dim3(bs), 0, c->stream, P, S.key.as<uint32_t>(),
                           S.cursor.as<uint32_t>(), counters, pad_shift);
        if (pad_shift)
            hipLaunchKernelGGL(unpad_kernel, dim3((ncells + bs - 1u) / bs), dim3(bs), 0, c->stream, ncells, counters, pad_shift,
                               S.cell_count.as<uint32_t>());
        MH_TRY((exclusive_scan<uint32_t, uint32_t>(c, S.cell_count.as<uint32_t>(), S.cell_count.as<uint32_t>(),
                                                   (uint64_t)ncells + 1)));
        // cells of more than 384 atoms on average: the order comes from a stable sort (sort_prep_kernel)
        const bool by_sort = (uint64_t)S.n > 384ull * ncells;
        if (by_sort) {
            MH_TRY(S.sort_buf.reserve((size_t)S.n * 16));
            uint32_t *k_in = S.sort_buf.as<uint32_t>(), *v_in = k_in + S.n, *k_out = v_in + S.n, *v_out = k_out + S.n;
            hipLaunchKernelGGL(sort_prep_kernel, dim3((S.n + 255u) / 256u), dim3(256), 0, c->stream, S.n, ncells, S.key.as<uint32_t>(), k_in, v_in);
            int end_bit = 1;
            while (end_bit < 32 && (2ull * ncells) >> end_bit) ++end_bit;
            MH_TRY(device_sort_pairs_u32(c, c->on_side ? c->sort_tmp_side : c->sort_tmp, k_in, k_out, v_in, v_out, S.n, end_bit));
            hipLaunchKernelGGL(sort_tmpkey_kernel, dim3((S.n + 255u) / 256u), dim3(256), 0, c->stream, S.n, ncells, k_out, v_out, S.tmp_key.as<uint32_t>());
        } else
        hipLaunchKernelGGL(scatter_kernel, dim3(nb), dim3(bs), 0, c->stream, S.n, S.key.as<uint32_t>(),
                           S.cell_count.as<uint32_t>(), S.cursor.as<uint32_t>(), S.tmp_key.as<uint32_t>());
        // (the spatial order serves the regular count pass and the fused histogram; pair_small.hip reads the placed records only)
        const int want_order = ((c->kind == MOLAR_HIP_SEARCH_SINGLE || c->kind == MOLAR_HIP_SEARCH_DOUBLE) &&
                                (c->hist_plan_now || small_cell_lanes(c) == 0)) ? 1 : 0;
        if (!want_order && !by_sort && (uint64_t)S.n < 16ull * ncells)
            hipLaunchKernelGGL(place_small_kernel, dim3((unsigned)(((uint64_t)ncells * 16u + bs - 1u) / bs)), dim3(bs), 0, c->stream, P, ncells,
                               ids_local, S.cell_count.as<uint32_t>(), S.tmp_key.as<uint32_t>(), S.d_vdw, S.sorted.as<float4>(),
                               S.d_vdw ? S.sorted_vdw.as<float>() : nullptr, S.aabb.as<float4>(), S.cell_org.as<float4>(),
                               S.cell_count.as<uint32_t>() + ncells + 1);
        else
        hipLaunchKernelGGL(place_order_kernel, dim3(ncells), dim3(64), 0, c->stream, P, ncells, ids_local,
                           S.cell_count.as<uint32_t>(), S.tmp_key.as<uint32_t>(), S.d_vdw, S.sorted.as<float4>(),
                           S.d_vdw ? S.sorted_vdw.as<float>() : nullptr, S.aabb.as<float4>(), S.perm.as<float4>(),
                           S.chunk_aabb.as<float4>(), S.h16.as<uint4>(), S.cell_org.as<float4>(), want_order, by_sort ? 1 : 0,
                           S.cell_count.as<uint32_t>() + ncells + 1);
        MH_HIP(hipGetLastError());
    }
    return 0;
}

int stage_set(molar_hip_ctx *c, GridSet &S, const float *xyz, size_t natoms, const uint64_t *idx, size_t n,
              const float *vdw, bool want_vdw) {
    if (!xyz) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "search: xyz pointer is null");
    const size_t nsel = idx ? n : natoms;
    if (nsel >= 0x7FFFFFFFull || natoms >= 0xFFFFFFFFull)
        return fail(MOLAR_HIP_ERR_TOO_LARGE, "search: %zu atoms exceed the 2^31 limit of the 32-bit device ids", nsel);
    S.n = (uint32_t)nsel;
    MH_TRY(to_device(c, xyz, natoms * 3, S.xyz_stage, &S.d_xyz));
    MH_TRY(to_device(c, idx, idx ? n : 0, S.idx_stage, &S.d_idx));
    S.d_vdw = nullptr;
    if (want_vdw) {
        if (!vdw) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "vdw search: radii pointer is null");
        MH_TRY(to_device(c, vdw, nsel, S.vdw_stage, &S.d_vdw));
    }
    return 0;
}

SearchParams make_params(molar_hip_ctx *c) {
    SearchParams P{};
    const bool two = c->kind != MOLAR_HIP_SEARCH_SINGLE;
    P.sa = c->set[0].sorted.as<float4>();
    P.csa = c->set[0].cell_count.as<uint32_t>();
    P.sb = two ? c->set[1].sorted.as<float4>() : P.sa;
    P.csb = two ? c->set[1].cell_count.as<uint32_t>() : P.csa;
    P.vdwa = c->set[0].sorted_vdw.as<float>();
    P.vdwb = c->set[1].sorted_vdw.as<float>();
    P.aabb_b = two ? c->set[1].aabb.as<float4>() : c->set[0].aabb.as<float4>();
    P.perm_b = two ? c->set[1].perm.as<float4>() : c->set[0].perm.as<float4>();
    P.chunk_aabb_b = two ? c->set[1].chunk_aabb.as<float4>() : c->set[0].chunk_aabb.as<float4>();
    P.h16_b = two ? c->set[1].h16.as<uint4>() : c->set[0].h16.as<uint4>();
    P.cell_org_b = two ? c->set[1].cell_org.as<float4>() : c->set[0].cell_org.as<float4>();
    P.mfma_count = c->env_no_mfma ? 0u : (c->env_no_mfma_wrapped ? 1u : 3u);
    P.task_desc = c->task_desc.as<TaskDesc>();
    P.maskbuf = c->maskbuf.as<uint32_t>();
    P.task_moff = c->task_moff.as<unsigned long long>();
    P.mask_cap_units = c->maskbuf.cap / 256u;
    P.out_cap = ~0ull;
    P.dx = c->dims[0];
    P.dy = c->dims[1];
    P.dz = c->dims[2];
    P.pbc = c->use_box ? c->pbc : 0u;
    P.use_box = c->use_box ? 1u : 0u;
    P.cutoff2 = c->cutoff * c->cutoff;
    P.ntasks = c->ntasks;
    P.nblocks = (uint32_t)c->nslots_bound;      // count / fill: one wave per workgroup (launch_pairs adjusts the histogram mode)
    P.box = c->box;
    for (int k = 0; k < 96; ++k) P.shifts4[k] = (c->use_box && k < 3 * c->box.nshift) ? c->box.shifts[k] : 0.0f;
    P.hist_nbins = 0u;
    P.hist_min = P.hist_max = 0.f;
    P.hist_bins = nullptr;
    P.hist_total = nullptr;
    P.hist_nslots = nullptr;
    // zero pattern shared by the matrix and its inverse -> which products a wrapped pair may skip
    P.wrap_kind = 3u;   // WK_GENERAL
    if (c->use_box) {
        const float *m = c->box.m, *iv = c->box.inv;
        auto z = [&](int k) { return m[k] == 0.0f && iv[k] == 0.0f; };
        const bool lower_zero = z(1) && z(2) && z(5);            // (1,0) (2,0) (2,1)
        const bool upper_zero = z(3) && z(6) && z(7);            // (0,1) (0,2) (1,2)
        if (lower_zero && upper_zero) P.wrap_kind = 1u;          // WK_DIAG
        else if (lower_zero) P.wrap_kind = 2u;                   // WK_UPPER (GROMACS-style boxes)
    }
    // image-box pruning of wrapped entries argues with n_d = round(f_d) in {-1,0,1}; keep the exhaustive
    // path on degenerate grids (a periodic dimension with fewer than 3 cells): they are tiny anyway
    P.prune_wrapped = 0u;
    if (c->use_box) {
        bool ok = true;
        for (int d = 0; d < 3; ++d)
            if (((c->pbc >> d) & 1u) && c->dims[d] < 3u) ok = false;
        P.prune_wrapped = ok ? 1u : 0u;
    }
    // Wrapped entries: classify with the plain distance to the image cell (b + S - a), decide exactly inside a band
    // around cutoff^2, and prune rows against the image box with a margin.  How far can the two evaluations of a
    // wrapped difference vector disagree?  u = 2^-24, L = largest |coordinate| the box allows (lab extents, box
    // vectors), kappa = || |M| |M^-1| ||_inf (1 for a rectangular box, grows with shear):
    //   reference (periodic_box.rs:291-297): v = p2 - p1 (error <= u L per component), f = inv v (3 products, 2 sums:
    //     <= 4u sum|inv||v| per component), s = M f (the error of f amplified by |M|: <= 4u kappa L, plus 4u L of
    //     its own roundings)                                            => <= (4 kappa + 5) u L per component
    //   approximate: S summed from box columns (<= 2u L), b + S (u L), - a (u L)   => <= 4u L per component
    // so a component of the difference vector differs by at most e = (4 kappa + 9) u L, a distance by sqrt(3) e, and
    // d2 near cutoff^2 by 2 sqrt(3) rc e, i.e. by 2 sqrt(3) (4 kappa + 9) u L/rc relative to cutoff^2.  The band is
    // 2e-4 plus FOUR times that; the pruning margin is 1e-3 nm plus four times sqrt(3) e.  Both therefore scale with
    // the box size AND its shear; boxes for which the band would exceed 5 % of cutoff^2 or the margin 5 % of the
    // cutoff take the exact path for every wrapped candidate.
    P.approx_wrapped = 0u;
    P.band_lo = P.band_hi = P.cutoff2;
    float margin = 1.0e-3f;
    if (c->use_box) {
        bool ok = (uint64_t)c->set[0].n + (uint64_t)c->set[1].n < (1ull << 26);   // (row<<26 | position) packing
        float ext[3], lmax = 0.f;
        molar_hip_box_lab_extents(&c->box, ext);
        for (int d = 0; d < 3; ++d) {
            if (((c->pbc >> d) & 1u) && c->dims[d] < 4u) ok = false;   // round(f_d) must be +-1 for wrapped pairs
            lmax = std::fmax(lmax, std::fabs(ext[d]));
            for (int k = 0; k < 3; ++k) lmax = std::fmax(lmax, std::fabs(c->box.m[3 * k + d]));
        }
        // atoms sit inside the cell along periodic dims, so a coordinate is bounded by the sum of the |box vectors|
        float lsum = 0.f;
        for (int d = 0; d < 3; ++d) {
            float row = 0.f;
            for (int k = 0; k < 3; ++k) row += std::fabs(c->box.m[3 * k + d]);
            lsum = std::fmax(lsum, row);
        }
        lmax = std::fmax(lmax, lsum);
        double kappa = 0.0;
        for (int i = 0; i < 3; ++i) {
            double row = 0.0;
            for (int j = 0; j < 3; ++j) {
                double e = 0.0;
                for (int k = 0; k < 3; ++k) e += std::fabs((double)c->box.m[3 * k + i]) * std::fabs((double)c->box.inv[3 * j + k]);
                row += e;
            }
            kappa = std::fmax(kappa, row);
        }
        const double u = 5.9604645e-8;
        const double e = (4.0 * kappa + 9.0) * u * (double)lmax;                   // per-component disagreement bound
        const double rel = 2.0e-4 + 4.0 * 2.0 * 1.7320508 * e / (double)c->cutoff;
        margin = (float)(1.0e-3 + 4.0 * 1.7320508 * e);
        if (!(margin < 0.05f * c->cutoff) || !std::isfinite(kappa)) {
            P.prune_wrapped = 0u;
            ok = false;
        }
        if (ok && rel < 0.05) {
            P.approx_wrapped = 1u;
            P.band_lo = P.cutoff2 * (float)(1.0 - rel);
            P.band_hi = P.cutoff2 * (float)(1.0 + rel);
        }
    }
    const float lim = c->cutoff + margin;
    P.prune_limit2 = lim * lim;
#ifdef MOLAR_HIP_DEBUG_KNOBS
    P.debug_skip = c->env_debug_skip;
    P.dbg = nullptr;
#endif
    return P;
}

template <bool FILL>
int launch_pairs(molar_hip_ctx *c, uint2 *pairs, float *dist, uint32_t *ids, uint32_t hist_nbins = 0, float hmin = 0.f,
                 float hmax = 0.f, unsigned long long *hist_bins = nullptr, unsigned long long out_cap = ~0ull,
                 bool params_resident = false, unsigned long long *hist_total = nullptr) {
    // the fill pass writes two (i, j) pairs / two distances per lane and instruction (fifo_flush_wide): naturally aligned
    // dwordx4 / dwordx2 stores relative to these bases (include/molar_hip.h states the requirement for caller-owned outputs)
    if (FILL && (((uintptr_t)pairs & 15u) || ((uintptr_t)dist & 7u)))
        return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "search fill: device outputs must be aligned to 16 bytes (pairs) and 8 bytes (distances)");
    Prof prof(c, FILL ? 3 : 1);
    SearchParams P = make_params(c);
    if (P.nblocks == 0) return 0;
    P.out_cap = out_cap;
    P.hist_nbins = hist_nbins;
    P.hist_min = hmin;
    P.hist_max = hmax;
    P.hist_bins = hist_bins;
    P.hist_total = hist_total;
    size_t dyn_lds = 0;
    P.hist_lean = 0u;
    P.hist_big = 0u;
    P.hist_edges = nullptr;
    P.hist_nslots = nullptr;
    P.hist_scale = 0.f;
    if (hist_nbins) {
        dyn_lds = (size_t)hist_nbins * 4;
        const uint32_t wpb = (uint32_t)waves_per_block(MODE_HIST);
        P.nblocks = (P.nblocks + wpb - 1u) / wpb;
        const uint32_t cap = (uint32_t)c->num_cus * 8u;
        if (P.nblocks > cap) P.nblocks = cap;
        // plain, same-cell and band-classified wrapped slots go to the lean kernel (8 waves per SIMD), the rest to
        // pair_kernel<MODE_HIST>; both add into the same bins
        P.hist_lean = (c->kind == MOLAR_HIP_SEARCH_SINGLE || c->kind == MOLAR_HIP_SEARCH_DOUBLE) ? 1u : 0u;
        {   // cells of more than KREG * 64 atoms are the rule (the grids' own criterion for ordering them by a sort, build_grid)
            const uint64_t ncells = (uint64_t)c->dims[0] * c->dims[1] * c->dims[2];
            const uint64_t nb_set = c->kind == MOLAR_HIP_SEARCH_SINGLE ? c->set[0].n : c->set[1].n;
            P.hist_big = (P.hist_lean && nb_set > 384ull * ncells) ? 1u : 0u;
        }
        if (P.hist_lean && c->edges_nbins == hist_nbins && c->edges_min == hmin && c->edges_max == hmax) {
            P.hist_edges = c->hist_edges.as<float>();
            P.hist_scale = (float)hist_nbins / (hmax - hmin);
        }
#ifdef MOLAR_HIP_DEBUG_KNOBS
        // 16 words per wave of hist_kernel; MOLAR_HIP_DEBUG_LAUNCH=n: only the n-th histogram launch of the context records
        // (a launch in the middle of a queued sequence, with the next frame's grid being built beside it)
        ++c->dbg_launches;
        if (!c->dbg.reserve((size_t)c->num_cus * 32u * 128u) && (c->env_debug_launch == 0 || c->env_debug_launch == c->dbg_launches))
            P.dbg = c->dbg.as<unsigned long long>();
#endif
    }
    MH_TRY(c->params.reserve(sizeof(SearchParams)));
    // params_resident: the block uploaded for the previous pass of this search is still valid for this one
    // (a kernel argument, not a copy from pageable memory: the latter makes the host wait for the stream to drain,
    // which would serialise the begin/end pipelining of molar_hip_search_resident_begin)
    // (params_fresh: the plan kernel of this search has already written exactly this block - resident searches)
    // (compared with the capacity the plan kernel was given: resident_enqueue has reset plan_out_cap by now - comparing with
    // that made every resident search upload the block again, 8.6 us on the critical stream of every frame, until round 4)
    const bool plan_wrote = c->params_fresh && !FILL && !hist_nbins && out_cap == c->params_fresh_cap;
    c->params_fresh = false;
    if (!params_resident && !plan_wrote && !P.hist_lean)
        hipLaunchKernelGGL(upload_params_kernel, dim3(1), dim3(256), 0, c->stream, P, c->params.as<SearchParams>());
    const SearchParams *dP = c->params.as<SearchParams>();
    const SlotDesc *tf = c->slot_desc.as<SlotDesc>();
    uint32_t st = (uint32_t)c->nslots_bound;
    if (c->slot_launch && !hist_nbins && c->slot_launch < st) P.nblocks = st = c->slot_launch;      // resident searches, see common.hpp
    uint32_t *sc = c->slot_cnt.as<uint32_t>();
    auto *sb = c->slot_base.as<unsigned long long>();
    const int mode = !FILL ? MODE_COUNT : (hist_nbins ? MODE_HIST : MODE_FILL);
    if (P.hist_lean) {
        // fused histogram of the fixed-cutoff kinds: one-kernel plan (which also writes the parameter block), the lean kernel on
        // its list, the generic kernel on the list of what the lean one cannot do (triclinic corner entries, oversized cells)
        if (!c->hist_queue.p) {         // slot queues and list counters: zeroed once, every launch leaves what the next one needs zeroed
            MH_TRY(c->hist_queue.reserve(hist_queue_words() * 4));
            MH_HIP(hipMemsetAsync(c->hist_queue.p, 0, hist_queue_words() * 4, c->stream));
        }
        uint32_t *queue = c->hist_queue.as<uint32_t>();
        const int lslot = (int)(c->hist_frames++ & 3u);
        P.hist_nslots = hist_list_count(queue, lslot, 1);
        launch_hist_plan(c->kind, c->stream, P, c->params.as<SearchParams>(), c->slot_desc.as<SlotDesc>(), c->slot_desc_rest.as<SlotDesc>(), queue, lslot);
        launch_hist_lean(c->kind, (unsigned)c->num_cus, dyn_lds, c->stream, dP, tf, st, queue, lslot, P.hist_big != 0u);
        tf = c->slot_desc_rest.as<SlotDesc>();
    }
    // frames of large cells (more than 448 atoms per cell of the second set on average, i.e. cells above 512 are common): the
    // instances with 128 registers per lane and up to 16 chunks of the second cell resident
    if (!hist_nbins && !ids && (c->kind == MOLAR_HIP_SEARCH_SINGLE || c->kind == MOLAR_HIP_SEARCH_DOUBLE)) {
        // (per occupied cell where the last search of this shape has counted them, like small_cell_lanes: a slab in a mostly empty box)
        const int sb_set = c->kind == MOLAR_HIP_SEARCH_SINGLE ? 0 : 1;
        const uint64_t all_cells = (uint64_t)c->dims[0] * c->dims[1] * c->dims[2];
        const uint64_t ncells = c->occ_use[sb_set] ? std::min<uint64_t>(c->occ_use[sb_set], all_cells) : all_cells;
        const uint64_t nb_set = c->set[sb_set].n;
        if (nb_set > 1000ull * ncells) {       // cells above 1024 atoms are the rule: 256 registers per lane, up to 2048 atoms resident (pair_k7.hip)
            launch_pair_huge(c->kind, mode, P.nblocks, c->stream, dP, tf, st, sc, sb, pairs, dist);
            MH_HIP(hipGetLastError());
            return 0;
        }
        if (nb_set > 448ull * ncells) {
            launch_pair_wide(c->kind, mode, P.nblocks, c->stream, dP, tf, st, sc, sb, pairs, dist);
            MH_HIP(hipGetLastError());
            return 0;
        }
    }
    // frames of small cells (contact / hydrogen-bond cutoffs): several slots per wave, pair_small.hip
    if (!hist_nbins && !ids) {
        const int lanes = small_cell_lanes(c);
        if (lanes) {
            launch_pair_small(mode, lanes, c->stream, dP, tf, st, sc, sb, pairs, dist);
            MH_HIP(hipGetLastError());
            return 0;
        }
    }
    switch (c->kind) {
        case MOLAR_HIP_SEARCH_SINGLE: launch_pair_single(mode, P.nblocks, dyn_lds, c->stream, dP, tf, st, sc, sb, pairs, dist, ids); break;
        case MOLAR_HIP_SEARCH_DOUBLE: launch_pair_double(mode, P.nblocks, dyn_lds, c->stream, dP, tf, st, sc, sb, pairs, dist, ids); break;
        case MOLAR_HIP_SEARCH_WITHIN: launch_pair_within(mode, P.nblocks, dyn_lds, c->stream, dP, tf, st, sc, sb, pairs, dist, ids); break;
        default: launch_pair_vdw(mode, P.nblocks, dyn_lds, c->stream, dP, tf, st, sc, sb, pairs, dist, ids); break;
    }
    MH_HIP(hipGetLastError());
    return 0;
}

// Exact bin edges of Histogram1D::add_one (molar_membrane/src/stats.rs:29-35) in terms of the SQUARED distance:
// edges[b], b = 0..nbins, is the smallest non-negative float x with floor(n * (sqrt(x) - min) / (max - min)) >= b, found
// by bisection over the bit patterns of the non-negative floats with the formula itself (IEEE sqrt / divide, no
// contraction - the same operations the kernels and the reference perform).  The bin of x is then the largest b with
// edges[b] <= x.  Returns false when the formula is not monotone-finite (max <= min, non-finite bounds): callers then
// evaluate the formula per value.
bool histogram_edges(float hmin, float hmax, size_t nbins, float *edges) {
    if (!(hmax > hmin) || !std::isfinite(hmin) || !std::isfinite(hmax) || nbins == 0) return false;
    const float hn = (float)nbins, range = hmax - hmin;
    auto bin_of = [&](uint32_t bits) -> float {
        float x;
        std::memcpy(&x, &bits, 4);
        volatile float d = std::sqrt(x);
        volatile float t = d - hmin;
        volatile float u = hn * t;
        volatile float v = u / range;
        return std::floor(v);
    };
    for (size_t b = 0; b <= nbins; ++b) {
        uint32_t lo = 0u, hi = 0x7F800000u;        // +0.0 .. +inf; bin_of(+inf) = +inf >= b
        if (bin_of(lo) >= (float)b) {
            hi = lo;
        } else {
            while (hi - lo > 1u) {                 // invariant: bin_of(lo) < b <= bin_of(hi)
                const uint32_t mid = lo + (hi - lo) / 2u;
                if (bin_of(mid) >= (float)b) hi = mid;
                else lo = mid;
            }
        }
        std::memcpy(&edges[b], &hi, 4);
    }
    return true;
}

int ensure_hist_edges(molar_hip_ctx *c, float hmin, float hmax, size_t nbins) {
    if (c->edges_nbins == nbins && c->edges_min == hmin && c->edges_max == hmax) return 0;
    c->edges_nbins = 0;
    std::vector<float> e(nbins + 1);
    if (!histogram_edges(hmin, hmax, nbins, e.data())) return 0;      // formula path
    // The kernel's first guess of a bin - (v_sqrt_f32(d2) - min) * n / (max - min) - is off by at most 2 ulp(d) / binwidth
    // + n * 2^-23 bins, and hist_add corrects it by ONE step against the table.  That is exact while a bin spans at least
    // 8 ulp of the largest distance of the range (0.26 bins of error); with narrower bins (a tiny [min, max] far from 0,
    // thousands of bins: neighbouring edges even coincide) the table is declined and the kernel evaluates the formula.
    {
        const float dmax = std::max(std::fabs(hmin), std::fabs(hmax));
        const float ulp = std::nextafter(dmax, INFINITY) - dmax;
        if (!((hmax - hmin) / (float)nbins >= 8.0f * ulp)) return 0;
        for (size_t b = 0; b < nbins; ++b)
            if (!(e[b] < e[b + 1]) && std::isfinite(e[b + 1]) && e[b + 1] != 0.0f) return 0;   // (bins below d = 0 share the edge 0)
    }
    MH_TRY(c->hist_edges.reserve((nbins + 1) * 4));
    MH_TRY(ensure_pinned(c, (nbins + 1) * 4));
    MH_HIP(hipStreamSynchronize(c->stream));       // a kernel of an earlier call may still read the old table / the staging area
    std::memcpy(c->h_pinned, e.data(), (nbins + 1) * 4);
    MH_HIP(hipMemcpyAsync(c->hist_edges.p, c->h_pinned, (nbins + 1) * 4, hipMemcpyHostToDevice, c->stream));
    MH_HIP(hipStreamSynchronize(c->stream));
    c->edges_min = hmin;
    c->edges_max = hmax;
    c->edges_nbins = nbins;
    return 0;
}

int read_back(molar_hip_ctx *c, void *dst_host, const void *src_dev, size_t bytes) {
    MH_TRY(ensure_pinned(c, bytes));
    MH_HIP(hipMemcpyAsync(c->h_pinned, src_dev, bytes, hipMemcpyDeviceToHost, c->stream));
    MH_HIP(hipStreamSynchronize(c->stream));
    std::memcpy(dst_host, c->h_pinned, bytes);
    return 0;
}

// zero-seeded bounding box of one set, accumulated into c->scan_tmp-independent 6-word buffer
int bbox_accumulate(molar_hip_ctx *c, uint32_t *d_mm, const GridSet &S) {
    if (!S.n) return 0;
    unsigned nb = (S.n + 255u) / 256u;
    if (nb > 2048u) nb = 2048u;
    hipLaunchKernelGGL(minmax_kernel, dim3(nb), dim3(256), 0, c->stream, S.d_xyz, S.d_idx, S.n, d_mm);
    MH_HIP(hipGetLastError());
    return 0;
}

// max of each of two device arrays with ONE read-back (the vdW cutoff, distance_search.rs:781-783)
int device_fmax2(molar_hip_ctx *c, const float *d_v1, uint32_t n1, const float *d_v2, uint32_t n2, float *out1, float *out2) {
    MH_TRY(c->hist.reserve(64));
    uint32_t *d = c->hist.as<uint32_t>();
    MH_HIP(hipMemsetAsync(d, 0, 8, c->stream));          // 0 is below every ordered float
    unsigned nb = ((n1 > n2 ? n1 : n2) + 2047u) / 2048u;
    if (nb > 128u) nb = 128u;
    if (nb == 0u) nb = 1u;
    hipLaunchKernelGGL(fmax_kernel, dim3(nb, 2), dim3(256), 0, c->stream, d_v1, n1, d_v2, n2, d);
    MH_HIP(hipGetLastError());
    uint32_t o[2];
    MH_TRY(read_back(c, o, d, 8));
    *out1 = ord2f(o[0]);
    *out2 = ord2f(o[1]);
    return 0;
}

// Second half of the plan, for the kernels that walk slots (count / fill / histogram): slot index of every plan entry and
// (fast kinds) its first hit-history unit by one single-pass scan over both, then one record per slot.
int enqueue_plan_slots(molar_hip_ctx *c) {
    const uint32_t fast_kind = records_hit_history(c) ? 1u : 0u;
    const unsigned pbs = c->on_side ? 64u : 256u;          // one-wave workgroups on the side stream (place_order_kernel)
    if (c->ntasks + 1 > (1ull << 18)) {
        // sparse giant plans (a vdW search of 1M atoms: 4e6 entries): the single-pass scan's chain of tiles - each waits for the
        // one before it - took 0.3 ms there; three launches of the tile scan do not chain (in place: a thread reads its items first)
        MH_TRY((exclusive_scan<uint32_t, uint32_t>(c, c->task_nb.as<uint32_t>(), c->task_nb.as<uint32_t>(), c->ntasks + 1)));
        if (fast_kind) MH_TRY((exclusive_scan<uint32_t, unsigned long long>(c, c->task_mu.as<uint32_t>(), c->task_moff.as<unsigned long long>(), c->ntasks + 1)));
    } else
    MH_TRY((scan_lookback<uint32_t, uint32_t, uint32_t, unsigned long long>(
        c, c->task_nb.as<uint32_t>(), c->task_nb.as<uint32_t>(), fast_kind ? c->task_mu.as<uint32_t>() : nullptr,
        c->task_moff.as<unsigned long long>(), c->ntasks + 1, c->scan_state.as<unsigned long long>())));
    // one record per slot (threads past the tasks blank the slots between the real count and the bound)
    const unsigned nbs = (unsigned)((c->ntasks + c->nslots_bound + 1 + pbs - 1) / pbs);
    hipLaunchKernelGGL(slotmap_kernel, dim3(nbs), dim3(pbs), 0, c->stream, c->ntasks, c->task_nb.as<uint32_t>(),
                       c->task_desc.as<TaskDesc>(), fast_kind ? c->task_moff.as<unsigned long long>() : nullptr,
                       c->slot_desc.as<SlotDesc>(), c->nslots_bound, c->sizes_dev);
    MH_HIP(hipGetLastError());
    return 0;
}

// Host-synchronous searches (molar_hip_search_count -> fill calls): what the plan came to, in one small read-back.
//  * the hit-history units of the fast kinds: the buffer is sized exactly;
//  * the number of slots: `nslots_bound` sizes every launch that walks slots (one workgroup per slot in the count and fill
//    passes), and the bound counts 14 / 28 plan entries per cell whether or not both cells hold atoms.  Where one set is
//    small (a solute in its solvent, the vdW overlap search of command_solvate.rs, `within` in its stream form) the plan is
//    mostly empty - 3.2e6 workgroups launched for 1.6e5 slots cost 0.8 ms per pass in bare launches - so the launches of this
//    search shrink to the slots that exist.  (Slot counters and records were prepared for the bound: a superset.)
int size_plan(molar_hip_ctx *c) {
    const bool fast_kind = records_hit_history(c);
    if (!fast_kind && c->nslots_bound <= 65536ull) return 0;       // nothing to size, too few launches to save: no round trip
    MH_TRY(ensure_pinned(c, 64));
    unsigned long long *h = reinterpret_cast<unsigned long long *>(c->h_pinned);
    h[0] = h[1] = 0ull;
    if (fast_kind) MH_HIP(hipMemcpyAsync(&h[0], c->task_moff.as<unsigned long long>() + c->ntasks, 8, hipMemcpyDeviceToHost, c->stream));
    MH_HIP(hipMemcpyAsync(&h[1], c->task_nb.as<uint32_t>() + c->ntasks, 4, hipMemcpyDeviceToHost, c->stream));      // task_nb is scanned in place: the total
    MH_HIP(hipStreamSynchronize(c->stream));
    if (fast_kind) {
        c->mask_units = h[0];
        MH_TRY(c->maskbuf.reserve((size_t)h[0] * 256u + 256u));
    }
    const uint64_t real = (uint32_t)h[1];
    if (real < c->nslots_bound) c->nslots_bound = real;
    return 0;
}

int prepare_search(molar_hip_ctx *c, const molar_hip_search_desc *q, bool size_masks = true) {
    if (!c || !q) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "search: null argument");
    if (q->kind < 0 || q->kind > 3) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "search: unknown kind %d", q->kind);
    MH_HIP(hipSetDevice(c->device));
    c->have_search = false;
    c->kind = q->kind;
    const bool two = q->kind != MOLAR_HIP_SEARCH_SINGLE;
    const bool vdw = q->kind == MOLAR_HIP_SEARCH_DOUBLE_VDW;
    c->use_box = q->box9 != nullptr;
    c->pbc = q->pbc & 7u;
    if (c->use_box) MH_TRY(molar_hip_box_from_matrix(q->box9, &c->box));
    // molar_hip_within_hold: consecutive `within` requests against the SAME first set (same pointers and sizes, same box, same
    // grid) reuse its staged coordinates and its grid; anything else that stages or bins set 0 ends the hold's validity
    const bool may_hold = c->within_hold && q->kind == MOLAR_HIP_SEARCH_WITHIN && c->skip_plan;
    // A first set in HOST memory is staged, so the hold would serve a copy: a fingerprint of the caller's array (its two ends and
    // 512 atoms spread over it) tells an array updated in place, or a new one that the allocator put at the old address, from the
    // one that was staged.  Device memory is read in place by the grid build; there the caller's promise is all there is.
    uint64_t fp = 0;
    if (may_hold && q->xyz1 && q->natoms1 && !is_device_ptr(q->xyz1)) {
        fp = 0xcbf29ce484222325ull;
        auto mix = [&](size_t atom) {
            uint32_t w[3];
            std::memcpy(w, q->xyz1 + 3 * atom, 12);
            for (int k = 0; k < 3; ++k) fp = (fp ^ w[k]) * 0x100000001b3ull;
        };
        const size_t n = q->natoms1, step = n > 512 ? n / 512 : 1;
        for (size_t a = 0; a < n && a < 64; ++a) mix(a);
        for (size_t a = 0; a < n; a += step) mix(a);
        for (size_t a = n > 64 ? n - 64 : 0; a < n; ++a) mix(a);
        fp |= 1ull;
    }
    bool reuse0 = may_hold && c->hold_valid && c->hold_fp == fp && c->hold_set == c->set && c->hold_xyz == q->xyz1 && c->hold_natoms == q->natoms1 &&
                  c->hold_idx == q->idx1 && c->hold_n == q->n1 && c->hold_ids_local == q->ids_local && c->hold_use_box == c->use_box &&
                  c->hold_pbc == c->pbc && (!c->use_box || std::memcmp(&c->hold_box, &c->box, sizeof c->box) == 0);
    if (!reuse0) {
        c->hold_valid = false;
        MH_TRY(stage_set(c, c->set[0], q->xyz1, q->natoms1, q->idx1, q->n1, q->vdw1, vdw));
    }
    if (two) MH_TRY(stage_set(c, c->set[1], q->xyz2, q->natoms2, q->idx2, q->n2, q->vdw2, vdw));
    else c->set[1].n = 0;

    float cutoff = q->cutoff;
    if (vdw) {
        // cutoff = max(vdw1) + max(vdw2) + EPSILON (:781-783); the reference panics on empty input
        if (c->set[0].n == 0 || c->set[1].n == 0) {
            c->dims[0] = c->dims[1] = c->dims[2] = 1;
            c->ntasks = 0;
            c->total = 0;
            c->have_search = true;
            return 0;
        }
        float m1, m2;
        MH_TRY(device_fmax2(c, c->set[0].d_vdw, c->set[0].n, c->set[1].d_vdw, c->set[1].n, &m1, &m2));
        cutoff = (m1 + m2) + F32_EPS;
    }
    if (!(cutoff > 0.0f)) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "search: cutoff must be positive (got %g)", (double)cutoff);
    c->cutoff = cutoff;

    float ext[3];
    if (c->use_box) {
        molar_hip_box_lab_extents(&c->box, ext);                       // Grid::from_cutoff_and_box :116-118
    } else {
        if (q->kind == MOLAR_HIP_SEARCH_WITHIN) {
            if (!q->lower3 || !q->upper3)
                return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "non-periodic within search needs lower3/upper3");
            for (int d = 0; d < 3; ++d) {
                c->lower[d] = q->lower3[d];
                c->upper[d] = q->upper3[d];
            }
        } else {
            // compute_bounding_box_single/double (:618-646), min/max seeded with zeros (:602-616)
            MH_TRY(c->hist.reserve(64));
            uint32_t *d_mm = c->hist.as<uint32_t>();
            uint32_t seed[6];
            for (int d = 0; d < 6; ++d) seed[d] = f2ord_host(0.0f);
            MH_HIP(hipMemcpyAsync(d_mm, seed, sizeof seed, hipMemcpyHostToDevice, c->stream));
            MH_TRY(bbox_accumulate(c, d_mm, c->set[0]));
            if (two) MH_TRY(bbox_accumulate(c, d_mm, c->set[1]));
            uint32_t mm[6];
            MH_TRY(read_back(c, mm, d_mm, sizeof mm));
            for (int d = 0; d < 3; ++d) {
                c->lower[d] = ord2f(mm[d]) + (-cutoff - F32_EPS);
                c->upper[d] = ord2f(mm[3 + d]) + (cutoff + F32_EPS);
            }
        }
        for (int d = 0; d < 3; ++d) ext[d] = c->upper[d] - c->lower[d];   // from_cutoff_and_min_max :112-114
    }
    MH_TRY(dims_from_extents(c, cutoff, ext));
    if (reuse0) {          // the same cells? (the cutoff and, without a box, the caller's bounds decide the grid)
        bool same = c->hold_dims[0] == c->dims[0] && c->hold_dims[1] == c->dims[1] && c->hold_dims[2] == c->dims[2];
        for (int d = 0; d < 3 && !c->use_box; ++d) same = same && c->hold_lower[d] == c->lower[d] && c->hold_upper[d] == c->upper[d];
        if (!same) {
            reuse0 = false;
            c->hold_valid = false;
            MH_TRY(stage_set(c, c->set[0], q->xyz1, q->natoms1, q->idx1, q->n1, q->vdw1, vdw));
        }
    }
    occ_latch(c);        // kind, set sizes and grid dimensions stand: what small_cell_lanes() answers for this search from here on
    const uint64_t ncells = (uint64_t)c->dims[0] * c->dims[1] * c->dims[2];
    c->ntasks = ncells * 14ull * (two ? 2ull : 1ull);
    // every set-1 cell is the first cell of at most 14 (two grids: 28) tasks, so
    // sum_t ceil(n1(t)/64) <= mult*N1/64 + ntasks
    c->nslots_bound = (two ? 28ull : 14ull) * (((uint64_t)c->set[0].n + 63ull) / 64ull) + c->ntasks;
    // entries wrapping in all three dims of a triclinic box use 2-row (<= 1024 rows) or 8-row slots: <= 28 tasks of <= 512 slots
    c->nslots_bound += 28ull * 512ull;
    // (the one-kernel plan of the fused histogram cuts same-cell entries into 32-row slots: one such entry per cell)
    if (c->hist_plan_now) c->nslots_bound += ((uint64_t)c->set[0].n + 31ull) / 32ull + ncells;
    if (c->ntasks >= 0xFFFFFFF0ull || c->nslots_bound >= 0xFFFFFFF0ull || c->ntasks + c->nslots_bound >= 0xFFFFFF00ull)
        return fail(MOLAR_HIP_ERR_TOO_LARGE, "search plan too large (%llu entries)", (unsigned long long)c->ntasks);
    MH_TRY(c->task_nb.reserve((c->ntasks + 1) * 4));
    MH_TRY(c->task_desc.reserve((c->ntasks + 1) * sizeof(TaskDesc)));
    MH_TRY(c->slot_desc.reserve((c->nslots_bound + 1) * sizeof(SlotDesc)));
    if (c->hist_plan_now) MH_TRY(c->slot_desc_rest.reserve((c->nslots_bound + 1) * sizeof(SlotDesc)));
    MH_TRY(c->slot_cnt.reserve((c->nslots_bound + 1) * 4));
    MH_TRY(c->slot_base.reserve((c->nslots_bound + 1) * 8));
    MH_TRY(c->params.reserve(sizeof(SearchParams)));
    const size_t st_tasks = lookback_state_words(c->ntasks + 1), st_slots = lookback_state_words(c->nslots_bound + 1);
    MH_TRY(c->scan_state.reserve((st_tasks + st_slots) * 8));
    MH_TRY(c->task_mu.reserve((c->ntasks + 1) * 4));
    MH_TRY(c->task_moff.reserve((c->ntasks + 1) * 8));
    const uint32_t fast_kind = records_hit_history(c) ? 1u : 0u;
    auto enqueue_plan = [&]() -> int {
        Prof prof(c, 0);
        SearchParams P = make_params(c);
        // Resident searches (no host round trip between plan and count: the hit-history buffer keeps its size): the plan
        // kernel also writes the parameter block the count and fill passes read, output capacity included.
        SearchParams *params_dst = nullptr;
        c->params_fresh = false;
        if (!size_masks) {
            P.out_cap = c->plan_out_cap;
            params_dst = c->params.as<SearchParams>();
            c->params_fresh = true;
            c->params_fresh_cap = P.out_cap;
        }
        if (!c->on_side && c->ntasks + 1 <= FPLAN_MAX_TASKS) {
            // main stream, a plan of at most 2^22 entries: plan and tile scan, then offsets and slot records - two launches (three
            // above 2^18 entries, plan_groups_kernel), no chain
            const uint32_t ntiles = (uint32_t)((c->ntasks + 1 + 255) / 256);
            const uint32_t ngroups = ntiles > 1024u ? (ntiles + PLAN_GROUP - 1u) / PLAN_GROUP : 0u;
            MH_TRY(c->fplan_tiles.reserve((size_t)ntiles * 16 + (c->ntasks + 1) * 8 + (size_t)ngroups * 16));
            unsigned long long *moff = fast_kind ? c->task_moff.as<unsigned long long>() : nullptr;
            unsigned long long *tt = c->fplan_tiles.as<unsigned long long>();
            unsigned long long *lmoff = tt + 2 * (size_t)ntiles;           // offsets inside the tiles: hit-history units ...
            uint32_t *lfirst = c->task_mu.as<uint32_t>();                  // ... and slots
            if (c->kind == MOLAR_HIP_SEARCH_SINGLE)
                hipLaunchKernelGGL((plan_tiles_kernel<MOLAR_HIP_SEARCH_SINGLE>), dim3(ntiles), dim3(256), 0, c->stream, P, lfirst,
                                   c->task_desc.as<TaskDesc>(), lmoff, fast_kind, c->slot_cnt.as<uint32_t>(), c->nslots_bound + 1, tt, params_dst);
            else          // the three two-grid kinds decode tasks identically
                hipLaunchKernelGGL((plan_tiles_kernel<MOLAR_HIP_SEARCH_DOUBLE>), dim3(ntiles), dim3(256), 0, c->stream, P, lfirst,
                                   c->task_desc.as<TaskDesc>(), lmoff, fast_kind, c->slot_cnt.as<uint32_t>(), c->nslots_bound + 1, tt, params_dst);
            unsigned long long *gt = ngroups ? lmoff + (c->ntasks + 1) : nullptr;
            if (ngroups) hipLaunchKernelGGL(plan_groups_kernel, dim3(ngroups), dim3(64), 0, c->stream, ntiles, tt, gt);
            hipLaunchKernelGGL(plan_slots_kernel, dim3(ntiles * (256u / PLAN_SUB)), dim3(256), 0, c->stream, c->ntasks, ntiles, lfirst, lmoff, c->task_nb.as<uint32_t>(),
                               c->task_desc.as<TaskDesc>(), moff, tt, c->slot_desc.as<SlotDesc>(), c->nslots_bound, c->sizes_dev, gt);
            MH_HIP(hipGetLastError());
            return 0;
        }
        // ntasks + 1 threads (the last one writes the scan terminators); the same grid zeroes the slot counters and the
        // descriptors of the two look-back scans of this search
        const uint64_t nplan = std::max<uint64_t>(std::max<uint64_t>(c->ntasks + 1, c->nslots_bound + 1), st_tasks + st_slots);
        const unsigned pbs = c->on_side ? 64u : 256u;          // one-wave workgroups on the side stream (place_order_kernel)
        const unsigned nb = (unsigned)((nplan + pbs - 1) / pbs);
        switch (c->kind) {
            case MOLAR_HIP_SEARCH_SINGLE:
                hipLaunchKernelGGL((plan_kernel<MOLAR_HIP_SEARCH_SINGLE>), dim3(nb), dim3(pbs), 0, c->stream, P, c->task_nb.as<uint32_t>(), c->task_desc.as<TaskDesc>(),
                                   c->task_mu.as<uint32_t>(), fast_kind, c->slot_cnt.as<uint32_t>(), c->nslots_bound + 1,
                                   c->scan_state.as<unsigned long long>(), (uint64_t)(st_tasks + st_slots), params_dst);
                break;
            default:   // the three two-grid kinds decode tasks identically
                hipLaunchKernelGGL((plan_kernel<MOLAR_HIP_SEARCH_DOUBLE>), dim3(nb), dim3(pbs), 0, c->stream, P, c->task_nb.as<uint32_t>(), c->task_desc.as<TaskDesc>(),
                                   c->task_mu.as<uint32_t>(), fast_kind, c->slot_cnt.as<uint32_t>(), c->nslots_bound + 1,
                                   c->scan_state.as<unsigned long long>(), (uint64_t)(st_tasks + st_slots), params_dst);
                break;
        }
        MH_HIP(hipGetLastError());
        return enqueue_plan_slots(c);
    };
    // Pipelined search on a context that owns its stream, inputs already in device memory: the grid build goes to the
    // side stream.  It touches only this generation's GridSet (last read by the search two frames back, which has been
    // ended) and its own scan scratch, so it needs to wait for nothing and overlaps the pair kernels of the frame in
    // front of it; the plan and the pair kernels of THIS search wait for it on the main stream.
    const bool side = c->want_side && c->own_stream && c->use_box && !vdw && c->set[0].d_xyz == q->xyz1 &&
                      c->set[0].d_idx == q->idx1 && (!two || (c->set[1].d_xyz == q->xyz2 && c->set[1].d_idx == q->idx2));
    if (side) {
        if (!c->side_stream) {
            int lo = 0, hi = 0;
            MH_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
            MH_HIP(hipStreamCreateWithPriority(&c->side_stream, hipStreamNonBlocking, hi));
            MH_HIP(hipEventCreateWithFlags(&c->grid_done, hipEventDisableTiming));
        }
        if (c->side_wait) MH_HIP(hipStreamWaitEvent(c->side_stream, c->side_wait, 0));
        if (c->side_wait2) MH_HIP(hipStreamWaitEvent(c->side_stream, c->side_wait2, 0));
        hipStream_t main_stream = c->stream;
        c->stream = c->side_stream;
        c->on_side = true;
        int rc = reuse0 ? 0 : build_grid(c, c->set[0], q->ids_local || vdw);
        if (!rc && two) rc = build_grid(c, c->set[1], q->ids_local || vdw);
        hipError_t e = rc ? hipSuccess : hipEventRecord(c->grid_done, c->side_stream);
        c->stream = main_stream;
        c->on_side = false;
        MH_TRY(rc);
        MH_HIP(e);
        // The plan and the pair kernels of this search need the grid.  A wait command on the main stream costs that stream
        // ~25 us in front of the plan kernel even when the grid has long been finished (cross-queue barrier packet); the
        // host waiting for the side stream instead costs nothing there: the frame in flight still has its fill pass
        // (~1 ms) in front of this search, and what follows is enqueued well inside that time.
        if (c->env_host_grid_wait) MH_HIP(hipEventSynchronize(c->grid_done));
        else MH_HIP(hipStreamWaitEvent(c->stream, c->grid_done, 0));
    } else {
        if (!reuse0) MH_TRY(build_grid(c, c->set[0], q->ids_local || vdw));
        if (two) MH_TRY(build_grid(c, c->set[1], q->ids_local || vdw));
    }
    if (may_hold && !reuse0) {        // this grid of set 0 is the one later `within` requests may reuse
        c->hold_valid = true;
        c->hold_fp = fp;
        c->hold_set = c->set;
        c->hold_xyz = q->xyz1; c->hold_natoms = q->natoms1; c->hold_idx = q->idx1; c->hold_n = q->n1;
        c->hold_ids_local = q->ids_local; c->hold_use_box = c->use_box; c->hold_pbc = c->pbc; c->hold_box = c->box;
        for (int d = 0; d < 3; ++d) { c->hold_dims[d] = c->dims[d]; c->hold_lower[d] = c->lower[d]; c->hold_upper[d] = c->upper[d]; }
    }

    // (Round 4: the plan kernels - 32 us of small launches - on the side stream behind the grid, with a second generation
    // of their buffers, one-wave workgroups: bit-identical, not faster; the fill pass of the frame in flight is stretched by
    // more than the main stream gains, 1.03 -> 1.06-1.12 ms.  Whatever shares the chip with the fill pass costs it at least
    // its own stand-alone duration.)
    if (!c->skip_plan) MH_TRY(enqueue_plan());
    // hit-history buffer of the count -> fill pair: sized exactly (one small read-back; the fused histogram
    // mode does not use it, but sizing it here keeps a later count/fill on the same cached search valid)
    c->mask_units = 0;
    if (size_masks && !c->skip_plan) MH_TRY(size_plan(c));
    return 0;
}

int finish_count(molar_hip_ctx *c) {
    Prof *prof = new Prof(c, 2);
    // (the look-back scan is slower here: ~140 chained tiles take 43 us against 14 us for the three-kernel scan)
    // the grand total (and the grids' occupied cells, for the next search of this shape) straight into pinned memory where the
    // block is visible to the device; else one read-back of the total
    void *sizes_dev = nullptr;
    if (ensure_pinned(c, 64) == 0 && hipHostGetDevicePointer(&sizes_dev, c->h_pinned, 0) == hipSuccess && sizes_dev) {
        std::memset(c->h_pinned, 0, 32);
    } else {
        (void)hipGetLastError();
        sizes_dev = nullptr;
    }
    int rc = scan_slot_counts(c, (unsigned long long *)sizes_dev);
    delete prof;
    MH_TRY(rc);
    unsigned long long tot = 0;
    if (sizes_dev) {
        MH_HIP(hipStreamSynchronize(c->stream));
        unsigned long long occ = 0;
        std::memcpy(&tot, c->h_pinned, 8);
        std::memcpy(&occ, (const char *)c->h_pinned + 24, 8);
        occ_note(c, occ_key_of(c), occ);
    } else {
        MH_TRY(read_back(c, &tot, c->slot_base.as<unsigned long long>() + c->nslots_bound, 8));
    }
    c->total = tot;
    c->have_search = true;
    return 0;
}

// ================================================================= SearchConnectivity on the device (connectivity.rs:19-35)
// `for (i, j) in pairs { conn[i].push(j); conn[j].push(i) }` as CSR: degrees by atomics, an exclusive scan, every entry
// dropped into its list in arrival order, then moved to its place - the number of entries of the same list that come from
// earlier pairs (a pair feeds a list at most once per side; the entry of its i side precedes that of its j side when i == j
// never happens: the single searches emit i != j).  Lists are a handful of bonded neighbours long, so the ranking reads its
// own list; the result is the reference's push order exactly.
// SearchConnectivity::from_iter (connectivity.rs:19-35): pair p pushes j onto i's list (entry 2p), then i onto j's (entry 2p + 1).
// A list in push order is the entries of its row in entry order: a STABLE sort of the entries by row (devsort.hip).  (Until the end
// of round 5: atomics into buckets, then every entry ranked against its whole list - quadratic in the list length, 10 ms of an
// 11 ms call for 25k atoms at rc 1.0 nm, 420 entries per list.)
__global__ void __launch_bounds__(256) conn_entries_kernel(const uint2 *__restrict__ pairs, unsigned long long npairs, uint32_t *__restrict__ row,
                                                           uint32_t *__restrict__ nb) {
    const unsigned long long p = (unsigned long long)blockIdx.x * 256u + threadIdx.x;
    if (p >= npairs) return;
    const uint2 ij = pairs[p];
    reinterpret_cast<uint2 *>(row)[p] = make_uint2(ij.x, ij.y);
    reinterpret_cast<uint2 *>(nb)[p] = make_uint2(ij.y, ij.x);
}

// off[r] = first sorted entry with row >= r, r = 0 .. nrows (off[nrows] = number of entries)
__global__ void __launch_bounds__(256) conn_offsets_kernel(const uint32_t *__restrict__ row_sorted, unsigned long long nent, uint32_t nrows,
                                                           unsigned long long *__restrict__ off) {
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r > nrows) return;
    unsigned long long lo = 0, hi = nent;
    while (lo < hi) {
        const unsigned long long mid = (lo + hi) >> 1;
        if (row_sorted[mid] < r) lo = mid + 1;
        else hi = mid;
    }
    off[r] = lo;
}

__global__ void __launch_bounds__(256) conn_widen_kernel(const uint32_t *__restrict__ nb_sorted, unsigned long long nent, unsigned long long *__restrict__ neigh) {
    const unsigned long long e = (unsigned long long)blockIdx.x * 256u + threadIdx.x;
    if (e < nent) neigh[e] = nb_sorted[e];
}

// ================================================================= `within` as a set (selection/ast.rs:589-631)
//
// What a caller of distance_search_within(_pbc) keeps is the SET of first-set atoms with a second-set atom in range: the
// raw stream (one id per plan entry in which the atom has a hit, :271-322,519-598) goes through SortedSet::from_unsorted
// (selection_expr.rs:112).  A set needs no stream, no offsets and no second pass - and an atom that has been found needs
// no further candidates, in ANY of its up to 28 plan entries:
//  * within_partners_kernel inverts the plan (same decode_task as every other kernel, so the same wrap / drop rules and
//    the same incomplete neighbourhoods of sheared boxes): per first-set cell the list of (second-set cell, wrap dims)
//    it meets, from both halves of every entry;
//  * within_flags_kernel: one wave per (first-set cell, share of its 64-row blocks) walks that list with a mask of the
//    rows still looking; per partner the rows that cannot reach its bounding box are left out (plain entries: the
//    exact f32 lower bound of run_fast), the partner's atoms are loaded 64 at a time, every remaining row is fetched
//    with one LDS broadcast and tested with the reference's arithmetic (PeriodicBox::distance_squared over the entry's
//    wrap dims for wrapped entries); a row leaves the mask at its first hit, the wave leaves the list when the mask is
//    empty; found rows set flags[id];
//  * ids are indices into a sorted selection (or 0..n), so the flagged positions in ascending order ARE the sorted,
//    de-duplicated result: a count per 2048-flag tile, a scan of the tile counts and one compaction pass.
constexpr uint32_t WITHIN_MAX_PART = 28;
// every (mask, half) pair of the plan has exactly one first-set cell as its origin: a cell is the first cell of at most
// 14 masks x 2 halves entries.  A plan with more masks or another stencil must grow the lists with it.
static_assert(WITHIN_MAX_PART == 2u * sizeof(pairk::MASKS) / sizeof(pairk::MASKS[0]), "partner lists are sized for the plan's stencil");

__global__ void __launch_bounds__(256) within_partners_kernel(const SearchParams *__restrict__ Pp, uint32_t *__restrict__ part_cnt,
                                                              uint32_t *__restrict__ part) {
    const SearchParams &P = *Pp;
    const uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (t >= P.ntasks) return;
    const Task T = decode_task<MOLAR_HIP_SEARCH_WITHIN, false>(P, t);
    if (!T.valid) return;
    const uint32_t s = atomicAdd(&part_cnt[T.ca], 1u);
    if (s < WITHIN_MAX_PART) part[(size_t)T.ca * WITHIN_MAX_PART + s] = T.cb | (T.wrap << 28);
}

// `within` against a SMALL second set (a ligand, a few residues: selection/ast.rs:589-631 on the shapes of
// molar/benches/within_size_bench.rs): the plan is walked from the second set's side.  One wave per (second-set atom, one of
// the 28 (mask, half) combinations); the wave of the FIRST atom of each occupied second-set cell finds the one plan entry
// that has this cell as its second cell through that combination (decode_task on the candidate entry index: the
// reference's own wrap / drop rules decide, nothing is re-derived), and tests the rows of the entry's first cell, 64 at a
// time, lanes = rows, against the cell's few atoms with the reference's arithmetic (distance_squared over the entry's wrap
// dims for wrapped entries).  A row that finds a partner sets its flag byte with an atomic OR on the containing word; the
// lane that turned it on appends the id to a list - the result, unsorted, with its length in memory.  No partner lists over
// all cells, no pass over the whole plan, no flag scan: two launches and one read-back for `within 0.8 of <20 atoms>`.
__global__ void __launch_bounds__(64) within_small_kernel(const SearchParams *__restrict__ Pp, uint32_t n2, uint32_t ncells, uint32_t by_cell,
                                                          uint32_t *__restrict__ flags32, uint32_t *__restrict__ list,
                                                          uint32_t *__restrict__ list_n) {
    const SearchParams &P = *Pp;
    const uint32_t lane = threadIdx.x;
    const uint32_t j = blockIdx.x / 28u, combo = blockIdx.x % 28u;
    uint32_t cell;
    if (by_cell) {                  // fewer cells than second-set atoms (large cutoffs): one wave per (cell, combination)
        cell = j;
        if (cell >= ncells || P.csb[cell + 1] == P.csb[cell]) return;
    } else {
        if (j >= n2) return;
        // the cell of sorted second-set atom j: the last cell whose start is <= j (binary search over cell_start)
        uint32_t lo = 0u, hi = ncells;
        while (hi - lo > 1u) {
            const uint32_t mid = (lo + hi) >> 1;
            if (P.csb[mid] <= j) lo = mid;
            else hi = mid;
        }
        cell = lo;
        if (P.csb[cell] != j) return;                               // not the first atom of its cell
    }
    const uint32_t m = combo >> 1, half = combo & 1u;
    const uint32_t cz = cell / (P.dx * P.dy), cy = (cell / P.dx) % P.dy, cx = cell % P.dx;
    const uint32_t cc[3] = {cx, cy, cz}, dims[3] = {P.dx, P.dy, P.dz};
    uint32_t home[3];
    for (int d = 0; d < 3; ++d) {
        const uint32_t off = half ? MASKS[m][d] : MASKS[m][3 + d];   // offset of the entry's SECOND cell from its home cell
        if (cc[d] >= off) home[d] = cc[d] - off;
        else if ((P.pbc >> d) & 1u) home[d] = dims[d] - 1u;          // reached across the periodic boundary
        else return;
    }
    const uint64_t cidx = ((uint64_t)home[0] * P.dy + home[1]) * P.dz + home[2];          // x outer, z inner
    const uint64_t t = (cidx * 14ull + m) * 2ull + half;
    if (t >= P.ntasks) return;
    const Task T = decode_task<MOLAR_HIP_SEARCH_WITHIN, false>(P, t);
    if (!T.valid || T.cb != cell) return;
    const float cutoff2 = P.cutoff2;
    const bool wrapped = P.use_box && T.wrap != 0u;
    float4 lo4 = make_float4(0.f, 0.f, 0.f, 0.f), hi4 = lo4;
    if (!wrapped) {
        lo4 = gload4(P.aabb_b, 2 * T.cb);
        hi4 = gload4(P.aabb_b, 2 * T.cb + 1);
    }
    // (blockIdx.y: a share of the first cell's 64-row blocks - large cutoffs mean cells of thousands of atoms)
    for (uint32_t i0 = blockIdx.y * 64u; i0 < T.n1; i0 += gridDim.y * 64u) {
        const bool have = i0 + lane < T.n1;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (have) a = gload4(P.sa, T.a0 + i0 + lane);
        bool look = have;
        if (!wrapped) look = look && !(aabb_d2(a.x, a.y, a.z, lo4.x, lo4.y, lo4.z, hi4.x, hi4.y, hi4.z) > cutoff2);
        if (__builtin_amdgcn_ballot_w64(look) == 0ull) continue;
        bool found = false;
        for (uint32_t k0 = 0; k0 < T.n2; k0 += 64u) {
            // 64 partner atoms at a time into the lanes, handed round with v_readlane (one load per 64 partners, not one each)
            float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k0 + lane < T.n2) q = gload4(P.sb, T.b0 + k0 + lane);
            const uint32_t cnt = T.n2 - k0 < 64u ? T.n2 - k0 : 64u;
            for (uint32_t kk = 0; kk < cnt; ++kk) {
                const float bx = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(q.x), kk));
                const float by = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(q.y), kk));
                const float bz = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(q.z), kk));
                const float dx = bx - a.x, dy = by - a.y, dz = bz - a.z;                   // p2 - p1
                const float d2 = wrapped ? wrapped_d2_exact<true>(P, T.wrap, dx, dy, dz)    // :306-321
                                         : (dx * dx + dy * dy) + dz * dz;                  // :281-292
                found = found || (look && d2 <= cutoff2);
                if ((kk & 7u) == 7u && __builtin_amdgcn_ballot_w64(look && !found) == 0ull) break;      // every row has its partner (:289, :318)
            }
            if (__builtin_amdgcn_ballot_w64(look && !found) == 0ull) break;
        }
        if (found) {
            const uint32_t id = __float_as_uint(a.w);
            const uint32_t bit = 1u << (8u * (id & 3u));
            const uint32_t old = atomicOr(&flags32[id >> 2], bit);
            if (!(old & bit)) list[atomicAdd(list_n, 1u)] = id;
        }
    }
}

// clears the flags a small-path call set (through its list) instead of a memset over every flag
__global__ void __launch_bounds__(256) within_clear_kernel(const uint32_t *__restrict__ list, uint32_t n, uint8_t *__restrict__ flags) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) flags[list[i]] = 0u;
}

__global__ void __launch_bounds__(64) within_flags_kernel(const SearchParams *__restrict__ Pp, const uint32_t *__restrict__ part_cnt,
                                                          const uint32_t *__restrict__ part, uint8_t *__restrict__ flags,
                                                          uint32_t nsplit) {
    __shared__ float4 la[64];
    const SearchParams &P = *Pp;
    const uint32_t lane = threadIdx.x, ca = blockIdx.x, split = blockIdx.y;
    const uint32_t a0 = P.csa[ca], n1 = P.csa[ca + 1] - a0;
    uint32_t np = part_cnt[ca];
    if (n1 == 0u || np == 0u) return;
    if (np > WITHIN_MAX_PART) np = WITHIN_MAX_PART;
    const float cutoff2 = P.cutoff2;
    for (uint32_t i0 = split * 64u; i0 < n1; i0 += nsplit * 64u) {
        const uint32_t rows = n1 - i0 < 64u ? n1 - i0 : 64u;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane < rows) a = gload4(P.sa, a0 + i0 + lane);
        __builtin_amdgcn_wave_barrier();
        la[lane] = a;
        __builtin_amdgcn_wave_barrier();
        const unsigned long long full = rows == 64u ? ~0ull : ((1ull << rows) - 1ull);
        unsigned long long live = full;                      // rows still looking for a partner atom
        for (uint32_t pi = 0; pi < np && live; ++pi) {
            const uint32_t e = __builtin_amdgcn_readfirstlane(part[(size_t)ca * WITHIN_MAX_PART + pi]);
            const uint32_t cb = e & 0x0FFFFFFFu, wrap = e >> 28;
            const uint32_t b0 = P.csb[cb], n2 = P.csb[cb + 1] - b0;
            unsigned long long cand = live;
            if (!wrap) {       // rows that cannot reach the partner's box: the f32 lower bound of every d2 of the row (run_fast)
                const float4 lo = gload4(P.aabb_b, 2 * cb), hi = gload4(P.aabb_b, 2 * cb + 1);
                cand &= __builtin_amdgcn_ballot_w64(!(aabb_d2(a.x, a.y, a.z, lo.x, lo.y, lo.z, hi.x, hi.y, hi.z) > cutoff2));
            }
            for (uint32_t j0 = 0; j0 < n2 && cand; j0 += 64u) {
                const bool inb = j0 + lane < n2;
                float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
                if (inb) q = gload4(P.sb, b0 + j0 + lane);
                unsigned long long rm = cand;
                while (rm) {
                    const uint32_t r = (uint32_t)__builtin_ctzll(rm);
                    rm &= rm - 1ull;
                    const float4 p = lload4(la, r);                                  // one broadcast ds_read per row
                    const float dx = q.x - p.x, dy = q.y - p.y, dz = q.z - p.z;      // p2 - p1
                    const float d2 = wrap ? wrapped_d2_exact<true>(P, wrap, dx, dy, dz)    // :306-321 (distance_squared over the entry's dims)
                                          : (dx * dx + dy * dy) + dz * dz;          // :281-292
                    if (__builtin_amdgcn_ballot_w64(inb && d2 <= cutoff2)) {         // `break` at the first hit (:289, :318)
                        live &= ~(1ull << r);
                        cand &= ~(1ull << r);
                    }
                }
            }
        }
        const unsigned long long found = full & ~live;
        if ((found >> lane) & 1ull) flags[__float_as_uint(a.w)] = 1u;
    }
}

// flags -> count per tile of 2048 (8 flags per thread)
__global__ void __launch_bounds__(256) flag_tile_count_kernel(const uint8_t *__restrict__ flags, uint64_t n, uint32_t *__restrict__ tile_cnt) {
    __shared__ uint32_t part[4];
    const uint64_t i = ((uint64_t)blockIdx.x * 256u + threadIdx.x) * 8u;
    uint32_t v = 0;
    for (uint32_t k = 0; k < 8u; ++k) v += (i + k < n && flags[i + k]) ? 1u : 0u;
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63u) == 0u) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

// positions of the set flags, ascending, as u64
__global__ void __launch_bounds__(256) flag_compact_kernel(const uint8_t *__restrict__ flags, uint64_t n, const unsigned long long *__restrict__ tile_off,
                                                           unsigned long long *__restrict__ out) {
    const uint64_t i = ((uint64_t)blockIdx.x * 256u + threadIdx.x) * 8u;
    uint32_t f[8], v = 0;
    for (uint32_t k = 0; k < 8u; ++k) {
        f[k] = (i + k < n && flags[i + k]) ? 1u : 0u;
        v += f[k];
    }
    uint32_t tot;
    uint32_t at = block_exclusive_scan<uint32_t>(v, &tot);
    unsigned long long o = tile_off[blockIdx.x] + at;
    for (uint32_t k = 0; k < 8u; ++k)
        if (f[k]) out[o++] = i + k;
}

// the pinned ring is worth its host threads for results of some size that go to memory the runtime cannot DMA into
bool ring_pays(size_t link_bytes, const void *a, const void *b = nullptr, const void *d = nullptr) {
    if (link_bytes < (24u << 20)) return false;
    for (const void *p : {a, b, d})
        if (p && is_pinned_host(p)) return false;
    return true;
}

}  // namespace

extern "C" {

int molar_hip_search_count(molar_hip_ctx *c, const molar_hip_search_desc *q, uint64_t *out_count) {
    MH_TRY(prepare_search(c, q));
    if (c->have_search) {   // degenerate (empty vdw input)
        if (out_count) *out_count = 0;
        return MOLAR_HIP_OK;
    }
    MH_TRY(launch_pairs<false>(c, nullptr, nullptr, nullptr));
    MH_TRY(finish_count(c));
    if (out_count) *out_count = c->total;
    return MOLAR_HIP_OK;
}

int molar_hip_search_grid_dims(molar_hip_ctx *c, uint64_t dims[3]) {
    if (!c || !c->have_search) return fail(MOLAR_HIP_ERR_NO_SEARCH, "no cached search: call molar_hip_search_count first");
    for (int d = 0; d < 3; ++d) dims[d] = c->dims[d];
    return MOLAR_HIP_OK;
}

int molar_hip_search_cell_kernels(molar_hip_ctx *c, int32_t *lanes, uint64_t occupied_cells[2]) {
    if (!c || !lanes) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "search_cell_kernels: null argument");
    *lanes = small_cell_lanes(c);
    if (*lanes == 0 && (c->kind == MOLAR_HIP_SEARCH_SINGLE || c->kind == MOLAR_HIP_SEARCH_DOUBLE)) {      // (launch_pairs' rule for large cells)
        const int sb_set = c->kind == MOLAR_HIP_SEARCH_SINGLE ? 0 : 1;
        const uint64_t all_cells = (uint64_t)c->dims[0] * c->dims[1] * c->dims[2];
        const uint64_t ncells = c->occ_use[sb_set] ? std::min<uint64_t>(c->occ_use[sb_set], all_cells) : all_cells;
        if (c->set[sb_set].n > 1000ull * ncells) *lanes = -2;
        else if (c->set[sb_set].n > 448ull * ncells) *lanes = -1;
    }
    if (occupied_cells) {
        occupied_cells[0] = c->occ_use[0];
        occupied_cells[1] = c->occ_use[1];
    }
    return MOLAR_HIP_OK;
}

static int fill_common(molar_hip_ctx *c, uint2 *d_pairs, float *d_dist, uint32_t *d_ids) {
    if (!c || !c->have_search) return fail(MOLAR_HIP_ERR_NO_SEARCH, "no cached search: call molar_hip_search_count first");
    MH_HIP(hipSetDevice(c->device));
    if (c->total == 0 || c->ntasks == 0) return 0;
    // The fill pass stores two results per lane (16-byte / 8-byte stores relative to the output bases).  A caller's device view
    // that is not aligned like that (an offset slice of a larger tensor) is filled through the context's own buffers and copied
    // device to device - correct at the cost of one more pass over the result, instead of an error.
    uint2 *p = d_pairs;
    float *d = d_dist;
    const bool via_p = p && ((uintptr_t)p & 15u) != 0, via_d = d && ((uintptr_t)d & 7u) != 0;
    if (via_p) {
        MH_TRY(c->out_pairs.reserve((size_t)c->total * 8));
        p = c->out_pairs.as<uint2>();
    }
    if (via_d) {
        MH_TRY(c->out_dist.reserve((size_t)c->total * 4));
        d = c->out_dist.as<float>();
    }
    MH_TRY(launch_pairs<true>(c, p, d, d_ids));
    if (via_p) MH_HIP(hipMemcpyAsync(d_pairs, p, (size_t)c->total * 8, hipMemcpyDeviceToDevice, c->stream));
    if (via_d) MH_HIP(hipMemcpyAsync(d_dist, d, (size_t)c->total * 4, hipMemcpyDeviceToDevice, c->stream));
    return 0;
}

int molar_hip_search_fill(molar_hip_ctx *c, uint32_t *pairs, float *dist) {
    if (!c || !c->have_search) return fail(MOLAR_HIP_ERR_NO_SEARCH, "no cached search: call molar_hip_search_count first");
    if (c->kind == MOLAR_HIP_SEARCH_WITHIN)
        return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "within search yields ids: use molar_hip_search_fill_ids");
    const bool pd = is_device_ptr(pairs), dd = is_device_ptr(dist);
    uint2 *dp = nullptr;
    float *ddist = nullptr;
    if (pairs) {
        if (pd) dp = reinterpret_cast<uint2 *>(pairs);
        else {
            MH_TRY(c->out_pairs.reserve((size_t)(c->total ? c->total : 1) * 8));
            dp = c->out_pairs.as<uint2>();
        }
    }
    if (dist) {
        if (dd) ddist = dist;
        else {
            MH_TRY(c->out_dist.reserve((size_t)(c->total ? c->total : 1) * 4));
            ddist = c->out_dist.as<float>();
        }
    }
    MH_TRY(fill_common(c, dp, ddist, nullptr));
    const bool hp = pairs && !pd && c->total, hd = dist && !dd && c->total;
    if (ring_pays(c->total * ((hp ? 8u : 0u) + (hd ? 4u : 0u)), hp ? pairs : nullptr, hd ? dist : nullptr)) {
        // large result into pageable memory: pinned ring + host threads (hoststream.hpp) instead of the runtime's bounce buffer
        std::vector<RingJob> jobs;
        if (hp) jobs.push_back(RingJob{dp, (size_t)c->total * 8, RING_COPY, pairs, nullptr});
        if (hd) jobs.push_back(RingJob{ddist, (size_t)c->total * 4, RING_COPY, dist, nullptr});
        return ring_to_host(c, jobs);
    }
    if (hp) MH_HIP(hipMemcpyAsync(pairs, dp, c->total * 8, hipMemcpyDeviceToHost, c->stream));
    if (hd) MH_HIP(hipMemcpyAsync(dist, ddist, c->total * 4, hipMemcpyDeviceToHost, c->stream));
    if ((pairs && !pd) || (dist && !dd)) MH_HIP(hipStreamSynchronize(c->stream));
    return MOLAR_HIP_OK;
}

int molar_hip_search_fill_device(molar_hip_ctx *c, const uint32_t **d_pairs, const float **d_dist) {
    if (!c || !c->have_search) return fail(MOLAR_HIP_ERR_NO_SEARCH, "no cached search: call molar_hip_search_count first");
    if (c->kind == MOLAR_HIP_SEARCH_WITHIN)
        return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "within search yields ids: use molar_hip_search_fill_ids");
    MH_TRY(c->out_pairs.reserve((size_t)(c->total ? c->total : 1) * 8));
    MH_TRY(c->out_dist.reserve((size_t)(c->total ? c->total : 1) * 4));
    MH_TRY(fill_common(c, c->out_pairs.as<uint2>(), c->out_dist.as<float>(), nullptr));
    if (d_pairs) *d_pairs = c->out_pairs.as<uint32_t>();
    if (d_dist) *d_dist = c->out_dist.as<float>();
    return MOLAR_HIP_OK;
}

// Enqueue one whole resident search (grid, plan, count, offset scan, fill into outP/outD against their present
// capacity) and the async read-back of its two sizes into `sizes` (pinned, 16 bytes).  No host wait.
// (struct ResidentLaunch: stages.hpp)

static int resident_enqueue(molar_hip_ctx *c, const molar_hip_search_desc *q, mh::DevBuf &outP, mh::DevBuf &outD,
                            void *sizes, ResidentLaunch *L) {
    if (q && q->kind == MOLAR_HIP_SEARCH_WITHIN)
        return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "within search yields ids: use molar_hip_search_count + fill_ids");
    const unsigned long long a = outP.cap / 8u, b = outD.cap / 4u;
    const unsigned long long cap0 = a < b ? a : b;
    // `sizes` is pinned host memory: the kernels write the two sizes there themselves (slotmap_kernel the hit-history
    // units, slot_offsets_kernel the grand total) - no copy commands behind the fill pass
    void *sizes_dev = nullptr;
    if (hipHostGetDevicePointer(&sizes_dev, sizes, 0) != hipSuccess) {
        (void)hipGetLastError();
        sizes_dev = nullptr;
    }
    c->plan_out_cap = cap0;
    c->sizes_dev = (unsigned long long *)sizes_dev;
    const int prc = prepare_search(c, q, /*size_masks=*/false);
    c->plan_out_cap = ~0ull;
    c->sizes_dev = nullptr;
    MH_TRY(prc);
    ++c->search_serial;
    *L = ResidentLaunch{};
    L->degenerate = c->have_search;
    if (L->degenerate) return 0;
    const bool fast_kind = records_hit_history(c);
    L->maskcap0 = c->maskbuf.cap / 256u;
    L->mask_units_dev = fast_kind ? c->task_moff.as<unsigned long long>() + c->ntasks : nullptr;
    // slots the two passes are launched over: what the plan of the search before came to, with a margin (common.hpp, trim_real)
    L->launched = c->nslots_bound;
    L->ntasks = c->ntasks;
    if (c->trim_real && c->trim_ntasks == c->ntasks && c->trim_kind == c->kind) {
        const unsigned long long want = c->trim_real + c->trim_real / 32u + 512ull;
        if (want < L->launched) L->launched = want;
    }
    c->slot_launch = (uint32_t)L->launched;
    struct SlotLaunchReset {
        molar_hip_ctx *c;
        ~SlotLaunchReset() { c->slot_launch = 0u; }
    } slot_launch_reset{c};
    Prof frame_span(c, 5);       // (profile mode 2 only: count + offsets + fill between ONE pair of events)
    // one parameter block serves both passes: the count pass ignores the output capacity
    MH_TRY(launch_pairs<false>(c, nullptr, nullptr, nullptr, 0, 0.f, 0.f, nullptr, cap0));
    if (c->record_count_done) {
        MH_HIP(hipEventRecord(c->count_done, c->stream));
        c->count_done_set = true;
    }
    {
        Prof prof(c, 2);
        MH_TRY(scan_slot_counts(c, (unsigned long long *)sizes_dev));
    }
    if (cap0) MH_TRY(launch_pairs<true>(c, outP.as<uint2>(), c->resident_no_dist ? nullptr : outD.as<float>(), nullptr, 0, 0.f, 0.f, nullptr, cap0,
                                        /*params_resident=*/true));
    if (!sizes_dev) {
        MH_HIP(hipMemcpyAsync(sizes, c->slot_base.as<unsigned long long>() + c->nslots_bound, 8, hipMemcpyDeviceToHost, c->stream));
        if (fast_kind)
            MH_HIP(hipMemcpyAsync((char *)sizes + 8, c->task_moff.as<unsigned long long>() + c->ntasks, 8, hipMemcpyDeviceToHost, c->stream));
        MH_HIP(hipMemcpyAsync((char *)sizes + 16, c->task_nb.as<uint32_t>() + c->ntasks, 4, hipMemcpyDeviceToHost, c->stream));   // (little-endian: the upper half was zeroed)
    }
    L->cap0 = cap0;
    return 0;
}

// The sizes of a resident search have been read: what its plan came to is the next search's launch size - and was this search
// itself launched over enough slots?  If not (the frame before had fewer slots than this one less the margin), the next
// enqueue of this plan launches the bound again.
static bool resident_covered(molar_hip_ctx *c, const void *sizes, unsigned long long launched, unsigned long long ntasks, int kind) {
    unsigned long long res[3] = {0, 0, 0};
    std::memcpy(res, sizes, 24);
    if (res[2] > launched) {
        c->trim_real = 0;
        return false;
    }
    c->trim_real = res[2];
    c->trim_ntasks = ntasks;
    c->trim_kind = kind;
    return true;
}

// With `sizes` delivered and the search still the context's cached one: grow what was too small and repeat the
// affected passes (first frame of a trajectory; later frames are the same size +- noise).
static int resident_settle(molar_hip_ctx *c, mh::DevBuf &outP, mh::DevBuf &outD, const void *sizes, const ResidentLaunch &L) {
    const bool fast_kind = c->kind == MOLAR_HIP_SEARCH_SINGLE || c->kind == MOLAR_HIP_SEARCH_DOUBLE;
    unsigned long long res[2] = {0, 0};
    std::memcpy(res, sizes, 16);
    c->total = res[0];
    c->mask_units = fast_kind ? res[1] : 0;
    c->have_search = true;
    bool refill = L.cap0 == 0 && c->total != 0;
    if (c->mask_units > L.maskcap0) {        // hit bits did not fit: grow, record them, fill again
        if (c->mask_units > c->maskbuf.cap / 256u)
            MH_TRY(c->maskbuf.reserve((size_t)(c->mask_units + c->mask_units / 4u) * 256u + 256u));
        MH_TRY(launch_pairs<false>(c, nullptr, nullptr, nullptr));
        refill = true;
    }
    if (c->total > L.cap0) {
        MH_TRY(outP.reserve((size_t)(c->total + c->total / 16u) * 8));
        MH_TRY(outD.reserve((size_t)(c->total + c->total / 16u) * 4));
        refill = true;
    }
    if (refill && c->total) {
        MH_TRY(launch_pairs<true>(c, outP.as<uint2>(), c->resident_no_dist ? nullptr : outD.as<float>(), nullptr));
        MH_HIP(hipStreamSynchronize(c->stream));   // like the common case, the result is complete when the call returns
    }
    return 0;
}

// One whole resident search, complete when the call returns (the stream has been waited for).
static int resident_run(molar_hip_ctx *c, const molar_hip_search_desc *q, mh::DevBuf &outP, mh::DevBuf &outD) {
    MH_TRY(ensure_pinned(c, 64));
    ResidentLaunch L;
    std::memset(c->h_pinned, 0, 32);
    MH_TRY(resident_enqueue(c, q, outP, outD, c->h_pinned, &L));
    if (L.degenerate) return 0;
    MH_HIP(hipStreamSynchronize(c->stream));
    if (!resident_covered(c, c->h_pinned, L.launched, L.ntasks, c->kind)) {       // too few slots launched: once more, over the bound
        std::memset(c->h_pinned, 0, 32);
        MH_TRY(resident_enqueue(c, q, outP, outD, c->h_pinned, &L));
        MH_HIP(hipStreamSynchronize(c->stream));
        (void)resident_covered(c, c->h_pinned, L.launched, L.ntasks, c->kind);
    }
    {
        unsigned long long occ = 0;
        std::memcpy(&occ, (const char *)c->h_pinned + 24, 8);
        occ_note(c, occ_key_of(c), occ);
    }
    return resident_settle(c, outP, outD, c->h_pinned, L);
}

}  // extern "C"

// stages.hpp: the resident search for callers that chain device work behind it
int mh::search_resident_enqueue(molar_hip_ctx *c, const molar_hip_search_desc *q, void *sizes_pinned, ResidentLaunch *L,
                                const unsigned long long **total_dev, const uint32_t **pairs_dev) {
    std::memset(sizes_pinned, 0, 24);
    {   // the chained stages (the membrane's patches) read the (i, j) plane only; their passes launch the bound (searches of 1e4
        // markers: nothing to trim, and nobody reads the plan's size back in between)
        const bool keep = c->resident_no_dist;
        const unsigned long long keep_trim = c->trim_real;
        c->resident_no_dist = true;
        c->trim_real = 0;
        const int rc = resident_enqueue(c, q, c->out_pairs, c->out_dist, sizes_pinned, L);
        c->resident_no_dist = keep;
        c->trim_real = keep_trim;
        MH_TRY(rc);
    }
    c->have_search = false;              // the sizes are not known to the host: not a cached search for the fill calls
    *total_dev = L->degenerate ? nullptr : c->slot_base.as<unsigned long long>() + c->nslots_bound;
    *pairs_dev = c->out_pairs.as<uint32_t>();
    return 0;
}

int mh::search_resident_fits(molar_hip_ctx *c, const void *sizes_pinned, const ResidentLaunch &L, bool *fits) {
    unsigned long long res[2] = {0, 0};
    std::memcpy(res, sizes_pinned, 16);
    *fits = true;
    if (L.degenerate) return 0;
    if (res[1] > L.maskcap0) {
        MH_TRY(c->maskbuf.reserve((size_t)(res[1] + res[1] / 4u) * 256u + 256u));
        *fits = false;
    }
    if (res[0] > L.cap0) {
        MH_TRY(c->out_pairs.reserve((size_t)(res[0] + res[0] / 16u) * 8));
        MH_TRY(c->out_dist.reserve((size_t)(res[0] + res[0] / 16u) * 4));
        *fits = false;
    }
    return 0;
}

extern "C" {

int molar_hip_search_resident(molar_hip_ctx *c, const molar_hip_search_desc *q, uint64_t *out_count,
                              const uint32_t **d_pairs, const float **d_dist) {
    if (!c) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "search: null argument");
    // Count, scan and fill are enqueued back to back against the capacities left by earlier frames; one
    // read-back tells whether the hit-bit buffer or the result buffers were too small.
    MH_TRY(resident_run(c, q, c->out_pairs, c->out_dist));
    if (out_count) *out_count = c->total;
    if (d_pairs) *d_pairs = c->out_pairs.as<uint32_t>();
    if (d_dist) *d_dist = c->resident_no_dist ? nullptr : c->out_dist.as<float>();
    return MOLAR_HIP_OK;
}

// DistanceSearchOutput for (usize, usize) (distance_search.rs:14-20): consumers of the pair list that never look at the
// distances - SearchConnectivity, the membrane's patches - ask for the (i, j) plane only.  The fill pass then takes no square
// roots and writes 8 instead of 12 bytes per result; the distance pointers of the resident calls come back NULL.
int molar_hip_search_resident_planes(molar_hip_ctx *c, int want_dist) {
    if (!c) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "null context");
    if (c->tickets[0].pending || c->tickets[1].pending)
        return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "resident_planes: pipelined searches are in flight");
    c->resident_no_dist = !want_dist;
    return MOLAR_HIP_OK;
}

int molar_hip_search_resident_begin(molar_hip_ctx *c, const molar_hip_search_desc *q, int32_t *ticket) {
    if (!c || !q || !ticket) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "search: null argument");
    const int slot = c->next_ticket & 1;
    molar_hip_ctx::Ticket &T = c->tickets[slot];
    if (T.pending)
        return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "both result sets are in flight: call molar_hip_search_resident_end first");
    MH_HIP(hipSetDevice(c->device));
    if (!c->h_sizes) {
        MH_HIP(hipHostMalloc(&c->h_sizes, 64, hipHostMallocDefault));      // 32 bytes per ticket
        std::memset(c->h_sizes, 0, 64);
    }
    if (!T.done) MH_HIP(hipEventCreateWithFlags(&T.done, hipEventDisableTiming));
    T.desc = *q;
    ResidentLaunch L;
    c->set = c->set_store[slot];         // this ticket's grid generation (the other one may still be read by the frame in flight)
    c->want_side = !c->env_no_side;
    c->side_wait = c->gen_free[slot];    // an asynchronous histogram call may have been the last reader of this generation
    // The grid of this frame is built on the side stream beside the kernels of the frame in flight, in one-wave workgroups
    // (place_order_kernel) so that it gets wave slots while they run.  Nothing about that is free: whatever shares the chip
    // with a pair kernel stretches it by about its own stand-alone duration (round 4: 115 us of grid kernels cost the fill
    // pass 40-60 us when they ran under it, the count pass 35-50 us when they run under that; frames/s the same in six
    // alternations).  The grid starts as soon as its generation is free - it then runs under the COUNT pass of the frame in
    // flight, whose workgroups leave one wave slot per SIMD open, and the fill pass has the chip and the write path to
    // itself.  MOLAR_HIP_GRID_LATE holds it back until that count pass has ended (the round-3 order).
    c->side_wait2 = (c->count_done_set && c->env_grid_late) ? c->count_done : nullptr;
    if (c->env_grid_late && !c->count_done) MH_HIP(hipEventCreateWithFlags(&c->count_done, hipEventDisableTiming));
    c->record_count_done = c->env_grid_late;
    std::memset((char *)c->h_sizes + 32 * slot + 24, 0, 8);      // (occupied cells: zero = not reported)
    const int erc = resident_enqueue(c, q, c->out_pairs_set[slot], c->out_dist_set[slot], (char *)c->h_sizes + 32 * slot, &L);
    c->record_count_done = false;
    c->want_side = false;
    c->side_wait = nullptr;
    c->side_wait2 = nullptr;
    MH_TRY(erc);
    MH_HIP(hipEventRecord(T.done, c->stream));
    T.cap0 = L.cap0;
    T.maskcap0 = L.maskcap0;
    T.launched = L.launched;
    T.ntasks = L.ntasks;
    T.kind = q->kind;
    T.occ_key = occ_key_of(c);
    T.degenerate = L.degenerate;
    T.serial = c->search_serial;
    T.pending = true;
    c->next_ticket ^= 1;
    *ticket = slot;
    return MOLAR_HIP_OK;
}

int molar_hip_search_resident_end(molar_hip_ctx *c, int32_t ticket, uint64_t *out_count, const uint32_t **d_pairs,
                                  const float **d_dist) {
    if (!c || ticket < 0 || ticket > 1 || !c->tickets[ticket].pending)
        return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "search: no pipelined search with ticket %d", (int)ticket);
    molar_hip_ctx::Ticket &T = c->tickets[ticket];
    mh::DevBuf &outP = c->out_pairs_set[ticket];
    mh::DevBuf &outD = c->out_dist_set[ticket];
    T.pending = false;
    MH_HIP(hipSetDevice(c->device));
    uint64_t total = 0;
    if (!T.degenerate) {
        MH_HIP(hipEventSynchronize(T.done));
        const void *sizes = (const char *)c->h_sizes + 32 * ticket;
        unsigned long long res[2], occ = 0;
        std::memcpy(res, sizes, 16);
        std::memcpy(&occ, (const char *)sizes + 24, 8);
        occ_note(c, T.occ_key, occ);
        const bool fast_kind = T.desc.kind == MOLAR_HIP_SEARCH_SINGLE || T.desc.kind == MOLAR_HIP_SEARCH_DOUBLE;
        total = res[0];
        const bool covered = resident_covered(c, sizes, T.launched, T.ntasks, T.kind);
        if (!covered || res[0] > T.cap0 || (fast_kind && res[1] > T.maskcap0)) {
            // A buffer was too small (first frames of a trajectory).  Let everything in flight finish - a younger
            // search owns the context's intermediate buffers by now, its results sit in the other result set - then
            // grow and repeat: the affected passes if this is still the context's cached search, else the frame.
            MH_HIP(hipStreamSynchronize(c->stream));
            ResidentLaunch L;
            L.cap0 = T.cap0;
            L.maskcap0 = T.maskcap0;
            if (T.serial != c->search_serial || !covered) {      // (not covered: the counts themselves are short - the whole frame again)
                if (fast_kind && res[1] > c->maskbuf.cap / 256u)
                    MH_TRY(c->maskbuf.reserve((size_t)(res[1] + res[1] / 4u) * 256u + 256u));
                if (res[0] > T.cap0) {
                    MH_TRY(outP.reserve((size_t)(res[0] + res[0] / 16u) * 8));
                    MH_TRY(outD.reserve((size_t)(res[0] + res[0] / 16u) * 4));
                }
                MH_TRY(ensure_pinned(c, 64));
                std::memset(c->h_pinned, 0, 32);
                MH_TRY(resident_enqueue(c, &T.desc, outP, outD, c->h_pinned, &L));
                MH_HIP(hipStreamSynchronize(c->stream));
                sizes = c->h_pinned;
                (void)resident_covered(c, sizes, L.launched, L.ntasks, T.kind);      // (launched over the bound: covered)
            }
            MH_TRY(resident_settle(c, outP, outD, sizes, L));
            MH_HIP(hipStreamSynchronize(c->stream));
            total = c->total;
        }
    }
    if (out_count) *out_count = total;
    if (d_pairs) *d_pairs = outP.as<uint32_t>();
    if (d_dist) *d_dist = c->resident_no_dist ? nullptr : outD.as<float>();
    return MOLAR_HIP_OK;
}

int molar_hip_search_fill_usize(molar_hip_ctx *c, uint64_t *oi, uint64_t *oj, float *dist) {
    if (!c || !c->have_search) return fail(MOLAR_HIP_ERR_NO_SEARCH, "no cached search: call molar_hip_search_count first");
    if (c->kind == MOLAR_HIP_SEARCH_WITHIN)
        return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "within search yields ids: use molar_hip_search_fill_ids");
    const size_t n = (size_t)c->total;
    MH_TRY(c->out_pairs.reserve((n ? n : 1) * 8));
    const bool dd = is_device_ptr(dist);
    float *ddist = nullptr;
    if (dist) {
        if (dd) ddist = dist;
        else {
            MH_TRY(c->out_dist.reserve((n ? n : 1) * 4));
            ddist = c->out_dist.as<float>();
        }
    }
    MH_TRY(fill_common(c, c->out_pairs.as<uint2>(), ddist, nullptr));
    if (n == 0) return MOLAR_HIP_OK;
    const bool di = is_device_ptr(oi), dj = is_device_ptr(oj);
    if (!di && !dj && !(dist && dd) && ring_pays(n * 12u, oi, oj, dist)) {
        // the usual Rust caller: (usize, usize, Float) columns in ordinary memory.  The link carries the 8-byte (u32, u32)
        // records and the f32 distances; the widening to u64 happens on the host threads that empty the pinned ring.
        std::vector<RingJob> jobs;
        if (oi || oj) jobs.push_back(RingJob{c->out_pairs.p, n * 8, RING_PAIRS_TO_U64, oi, oj});
        if (dist) jobs.push_back(RingJob{ddist, n * 4, RING_COPY, dist, nullptr});
        return ring_to_host(c, jobs);
    }
    unsigned long long *wi = nullptr, *wj = nullptr;
    if (oi) {
        if (di) wi = reinterpret_cast<unsigned long long *>(oi);
        else {
            MH_TRY(c->wide_i.reserve(n * 8));
            wi = c->wide_i.as<unsigned long long>();
        }
    }
    if (oj) {
        if (dj) wj = reinterpret_cast<unsigned long long *>(oj);
        else {
            MH_TRY(c->wide_j.reserve(n * 8));
            wj = c->wide_j.as<unsigned long long>();
        }
    }
    unsigned nb = (unsigned)((n + 255) / 256);
    if (nb > 8192u) nb = 8192u;
    hipLaunchKernelGGL(widen_pairs_kernel, dim3(nb), dim3(256), 0, c->stream, c->out_pairs.as<uint2>(), (uint64_t)n, wi, wj);
    MH_HIP(hipGetLastError());
    if (oi && !di) MH_HIP(hipMemcpyAsync(oi, wi, n * 8, hipMemcpyDeviceToHost, c->stream));
    if (oj && !dj) MH_HIP(hipMemcpyAsync(oj, wj, n * 8, hipMemcpyDeviceToHost, c->stream));
    if (dist && !dd) MH_HIP(hipMemcpyAsync(dist, ddist, n * 4, hipMemcpyDeviceToHost, c->stream));
    MH_HIP(hipStreamSynchronize(c->stream));
    return MOLAR_HIP_OK;
}

int molar_hip_search_fill_ids(molar_hip_ctx *c, uint64_t *ids) {
    if (!c || !c->have_search) return fail(MOLAR_HIP_ERR_NO_SEARCH, "no cached search: call molar_hip_search_count first");
    if (c->kind != MOLAR_HIP_SEARCH_WITHIN)
        return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "fill_ids is for within searches");
    const size_t n = (size_t)c->total;
    if (n == 0 || !ids) return MOLAR_HIP_OK;
    MH_TRY(c->out_ids.reserve(n * 4));
    MH_TRY(fill_common(c, nullptr, nullptr, c->out_ids.as<uint32_t>()));
    const bool dv = is_device_ptr(ids);
    if (!dv && ring_pays(n * 4u, ids)) {
        std::vector<RingJob> jobs{RingJob{c->out_ids.p, n * 4, RING_U32_TO_U64, ids, nullptr}};
        return ring_to_host(c, jobs);
    }
    unsigned long long *w;
    if (dv) w = reinterpret_cast<unsigned long long *>(ids);
    else {
        MH_TRY(c->wide_i.reserve(n * 8));
        w = c->wide_i.as<unsigned long long>();
    }
    unsigned nb = (unsigned)((n + 255) / 256);
    if (nb > 8192u) nb = 8192u;
    hipLaunchKernelGGL(widen_ids_kernel, dim3(nb), dim3(256), 0, c->stream, c->out_ids.as<uint32_t>(), (uint64_t)n, w);
    MH_HIP(hipGetLastError());
    if (!dv) {
        MH_HIP(hipMemcpyAsync(ids, w, n * 8, hipMemcpyDeviceToHost, c->stream));
        MH_HIP(hipStreamSynchronize(c->stream));
    }
    return MOLAR_HIP_OK;
}

#ifdef MOLAR_HIP_DEBUG_KNOBS
// debug builds only (not part of include/molar_hip.h): the per-wave time accounting of the LAST hist_kernel launch
int molar_hip_debug_fetch(molar_hip_ctx *c, void *dst, size_t bytes) {
    if (!c || !dst || bytes > c->dbg.cap) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "debug_fetch: bad argument");
    MH_HIP(hipStreamSynchronize(c->stream));
    MH_HIP(hipMemcpy(dst, c->dbg.p, bytes, hipMemcpyDeviceToHost));
    return MOLAR_HIP_OK;
}
#endif

int molar_hip_histogram_edges(float hmin, float hmax, size_t nbins, float *edges) {
    if (!edges) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "histogram_edges: null argument");
    if (!histogram_edges(hmin, hmax, nbins, edges))
        return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "histogram_edges: needs finite min < max and nbins > 0");
    return MOLAR_HIP_OK;
}

int molar_hip_search_histogram(molar_hip_ctx *c, const molar_hip_search_desc *q, float hmin, float hmax, size_t nbins,
                               uint64_t *bins, uint64_t *out_count) {
    if (!c || !q || !bins) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "search_histogram: null argument");
    if (q->kind == MOLAR_HIP_SEARCH_WITHIN)
        return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "search_histogram: a within search has no distances");
    if (nbins == 0 || nbins > 8192) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "search_histogram: nbins must be in 1..8192");
    // Device-resident bins and no count wanted: the call waits for nothing, so a loop over frames queues them back to
    // back.  Such calls alternate between the two grid generations and build their grid on the side stream, under the
    // histogram kernel of the frame before (each generation's last reader is marked by an event on the main stream).
    const bool async = is_device_ptr(bins) && !out_count && !c->tickets[0].pending && !c->tickets[1].pending;
    int gen = 0;
    if (async) {
        MH_HIP(hipSetDevice(c->device));
        gen = (c->hist_gen ^= 1);
        c->set = c->set_store[gen];
        c->side_wait = c->gen_free[gen];
        c->want_side = !c->env_no_side;
    }
    // the fixed-cutoff kinds plan inside launch_pairs (hist_plan_kernel): prepare_search builds the grid only
    const bool hist_plan = q->kind == MOLAR_HIP_SEARCH_SINGLE || q->kind == MOLAR_HIP_SEARCH_DOUBLE;
    c->skip_plan = hist_plan;
    c->hist_plan_now = hist_plan;
    const int prc = prepare_search(c, q, /*size_masks=*/false);     // the fused pass records no hit bits
    c->skip_plan = false;
    c->hist_plan_now = false;
    c->want_side = false;
    c->side_wait = nullptr;
    MH_TRY(prc);
    if (out_count) *out_count = 0;
    if (c->have_search) return MOLAR_HIP_OK;     // degenerate (empty vdw input)
    // single pass: no counts, no offsets - every emitted distance goes straight into the histogram
    MH_TRY(ensure_hist_edges(c, hmin, hmax, nbins));
    if (async) {
        // bins in device memory and no count wanted: the kernels add straight into the caller's bins (integer atomics:
        // the same sums) - no scratch histogram to zero before and to add after, two launches less per frame
        MH_TRY(launch_pairs<true>(c, nullptr, nullptr, nullptr, (uint32_t)nbins, hmin, hmax, reinterpret_cast<unsigned long long *>(bins)));
        if (!c->gen_free[gen]) MH_HIP(hipEventCreateWithFlags(&c->gen_free[gen], hipEventDisableTiming));
        MH_HIP(hipEventRecord(c->gen_free[gen], c->stream));
        return MOLAR_HIP_OK;
    }
    MH_TRY(c->hist.reserve((nbins + 1) * 8));
    hipLaunchKernelGGL(zero2_kernel, dim3(1), dim3(256), 0, c->stream, c->hist.as<uint32_t>(), (nbins + 1) * 2, (uint32_t *)nullptr,
                       (size_t)0);
    MH_TRY(launch_pairs<true>(c, nullptr, nullptr, nullptr, (uint32_t)nbins, hmin, hmax, c->hist.as<unsigned long long>(), ~0ull, false,
                              c->hist.as<unsigned long long>() + nbins));
    if (is_device_ptr(bins)) {
        hipLaunchKernelGGL(add_u64_kernel, dim3((unsigned)((nbins + 255) / 256)), dim3(256), 0, c->stream,
                           reinterpret_cast<unsigned long long *>(bins), c->hist.as<unsigned long long>(), nbins);
        MH_HIP(hipGetLastError());
        if (out_count) {
            unsigned long long tot = 0;
            MH_TRY(read_back(c, &tot, c->hist.as<unsigned long long>() + nbins, 8));
            *out_count = tot;
        }
        return MOLAR_HIP_OK;
    }
    std::vector<unsigned long long> h(nbins + 1);
    MH_TRY(read_back(c, h.data(), c->hist.p, (nbins + 1) * 8));
    for (size_t b = 0; b < nbins; ++b) bins[b] += h[b];
    if (out_count) *out_count = h[nbins];
    return MOLAR_HIP_OK;
}

// Several frames of one trajectory through the fused histogram in ONE set of launches (BASELINE config 4: the frames of a
// trajectory are independent, and bins do not care in which order - or of which frame - pairs are found).  What a frame costs
// beside its arithmetic is all per LAUNCH: the persistent kernel's idle tail (a fifth of its span), the plan, the generic kernel
// for the triclinic corner entries, and the grid of the next frame, which the persistent kernel keeps off the chip until that
// tail.  A group of up to HIST_BATCH frames shares them:
//   side stream : parameter blocks + grid records of the group (one copy from pinned memory), then the grids of ALL its frames by
//                 the frame-indexed kernels (bin, unpad + scan, scatter, place + order), then all plans in one launch - slot
//                 records of every frame in the same two lists, the frame's number in the record - and the generic kernel
//                 over the joint rest list;
//   main stream : one hist_kernel over the joint list.
// Two generations of everything the side stream writes; the side stream starts on a generation when the launch two before has
// ended (gen_free, the event the asynchronous single-frame calls use as well: the two forms may be mixed on a context).
// Returns 1 when the group does not qualify (the caller then walks it frame by frame), 0 when enqueued.
constexpr int HIST_BATCH = MH_HIST_BATCH;
static_assert(HIST_BATCH <= (int)(sizeof(((molar_hip_ctx *)nullptr)->hb_sets[0]) / sizeof(((molar_hip_ctx *)nullptr)->hb_sets[0][0])), "hb_sets holds a group");

static int hist_frames_group(molar_hip_ctx *c, const molar_hip_search_desc *q, size_t first, int W, size_t stride1, size_t stride2,
                             const float *boxes9, float hmin, float hmax, size_t nbins, unsigned long long *bins) {
    // ---- what every frame of the group comes to on the host: box, grid dims.  One shape for all of them, or no batch.
    const float cutoff = q->cutoff;
    if (!(cutoff > 0.0f)) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "search: cutoff must be positive (got %g)", (double)cutoff);
    const bool two = q->kind == MOLAR_HIP_SEARCH_DOUBLE;
    const int nsets = two ? 2 : 1;
    const size_t nsel[2] = {q->idx1 ? q->n1 : q->natoms1, two ? (q->idx2 ? q->n2 : q->natoms2) : 0};
    for (int s = 0; s < nsets; ++s)
        if (nsel[s] == 0 || nsel[s] >= 0x7FFFFFFFull || (s ? q->natoms2 : q->natoms1) >= 0xFFFFFFFFull) return 1;
    molar_hip_box boxes[HIST_BATCH];
    uint32_t dims[3] = {0, 0, 0};
    for (int f = 0; f < W; ++f) {
        const float *b9 = boxes9 ? boxes9 + 9 * (first + f) : q->box9;
        MH_TRY(molar_hip_box_from_matrix(b9, &boxes[f]));
        float ext[3];
        molar_hip_box_lab_extents(&boxes[f], ext);
        c->box = boxes[f];
        MH_TRY(dims_from_extents(c, cutoff, ext));
        if (f == 0) for (int d = 0; d < 3; ++d) dims[d] = c->dims[d];
        else if (dims[0] != c->dims[0] || dims[1] != c->dims[1] || dims[2] != c->dims[2]) return 1;
    }
    const uint64_t ncells64 = (uint64_t)dims[0] * dims[1] * dims[2];
    if (ncells64 + 1 > 32768ull) return 1;                                   // one-wave scan per frame
    for (int s = 0; s < nsets; ++s)
        if ((uint64_t)nsel[s] > 384ull * ncells64) return 1;               // big cells are ordered by a device sort
    const uint32_t ncells = (uint32_t)ncells64;
    const uint32_t n[2] = {(uint32_t)nsel[0], (uint32_t)nsel[1]};
    c->kind = q->kind;
    c->use_box = true;
    c->pbc = q->pbc & 7u;
    c->cutoff = cutoff;
    c->have_search = false;
    c->hold_valid = false;
    c->ntasks = ncells64 * 14ull * (two ? 2ull : 1ull);
    // slots of one frame (prepare_search) incl. the 32-row slots of the same-cell entries
    const uint64_t bound1 = (two ? 28ull : 14ull) * (((uint64_t)n[0] + 63ull) / 64ull) + c->ntasks + 28ull * 512ull + ((uint64_t)n[0] + 31ull) / 32ull + ncells64;
    const uint64_t bound = bound1 * (uint64_t)W;
    if (bound >= 0xFFFFFFF0ull) return 1;
    const int gen = (c->hist_gen ^= 1);
    MH_TRY(c->hb_lean[gen].reserve((bound1 * HIST_BATCH + 1) * sizeof(SlotDesc)));
    MH_TRY(c->hb_rest[gen].reserve((bound1 * HIST_BATCH + 1) * sizeof(SlotDesc)));
    const size_t blk_bytes = HIST_BATCH * (sizeof(SearchParams) + 2 * sizeof(GridFrame));
    MH_TRY(c->hb_blocks[gen].reserve(blk_bytes));
    if (!c->hb_pin) {
        MH_HIP(hipHostMalloc(&c->hb_pin, 4 * blk_bytes, hipHostMallocDefault));
        for (auto &e : c->hb_pin_ev) MH_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    if (!c->side_stream) {
        int lo = 0, hi = 0;
        MH_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
        MH_HIP(hipStreamCreateWithPriority(&c->side_stream, hipStreamNonBlocking, hi));
        MH_HIP(hipEventCreateWithFlags(&c->grid_done, hipEventDisableTiming));
    }
    MH_TRY(ensure_hist_edges(c, hmin, hmax, nbins));
    if (!c->hist_queue.p) {
        MH_TRY(c->hist_queue.reserve(hist_queue_words() * 4));
        MH_HIP(hipMemsetAsync(c->hist_queue.p, 0, hist_queue_words() * 4, c->stream));
        MH_HIP(hipStreamSynchronize(c->stream));          // the side stream's plan is the first to touch the counters
    }
    uint32_t *queue = c->hist_queue.as<uint32_t>();
    const int lslot = (int)(c->hist_frames++ & 3u);
    // ---- the group's records in the pinned slot: parameter blocks, then grid records (first sets, second sets)
    const int pslot = c->hb_pin_next++ & 3;
    if (c->hb_pin_used[pslot]) MH_HIP(hipEventSynchronize(c->hb_pin_ev[pslot]));      // its copy of four groups ago has long run
    char *pin = (char *)c->hb_pin + (size_t)pslot * blk_bytes;
    SearchParams *hP = reinterpret_cast<SearchParams *>(pin);
    GridFrame *hG = reinterpret_cast<GridFrame *>(pin + HIST_BATCH * sizeof(SearchParams));
    SearchParams *dP = c->hb_blocks[gen].as<SearchParams>();
    GridFrame *dG = reinterpret_cast<GridFrame *>((char *)c->hb_blocks[gen].p + HIST_BATCH * sizeof(SearchParams));
    mh::GridSet *keep_set = c->set;
    const uint32_t pad_shift[2] = {grid_pad_shift(n[0], ncells), two ? grid_pad_shift(n[1], ncells) : 0u};
    size_t dyn_lds = (size_t)nbins * 4;
    bool big = false;
    for (int f = 0; f < W; ++f) {
        mh::GridSet *SS = c->hb_sets[gen][f];
        for (int s = 0; s < nsets; ++s) {
            mh::GridSet &S = SS[s];
            S.n = n[s];
            S.d_xyz = s ? q->xyz2 + (first + f) * stride2 : q->xyz1 + (first + f) * stride1;
            S.d_idx = s ? q->idx2 : q->idx1;
            S.d_vdw = nullptr;
            const int rrc = grid_reserve(S, ncells);
            if (rrc) { c->set = keep_set; return rrc; }
        }
        c->box = boxes[f];
        c->set = SS;                              // (SINGLE: set[1] is never read)
        c->nslots_bound = bound;
        SearchParams P = make_params(c);
        P.nblocks = 0;
        P.out_cap = ~0ull;
        P.hist_nbins = (uint32_t)nbins;
        P.hist_min = hmin;
        P.hist_max = hmax;
        P.hist_bins = bins;
        P.hist_total = nullptr;
        P.hist_lean = 1u;
        P.hist_big = 0u;                          // (mean population <= 384: checked above)
        P.hist_edges = nullptr;
        P.hist_scale = 0.f;
        if (c->edges_nbins == nbins && c->edges_min == hmin && c->edges_max == hmax) {
            P.hist_edges = c->hist_edges.as<float>();
            P.hist_scale = (float)nbins / (hmax - hmin);
        }
        P.hist_nslots = hist_list_count(queue, lslot, 1);
        hP[f] = P;
        for (int s = 0; s < nsets; ++s) {
            mh::GridSet &S = SS[s];
            GridFrame g{};
            g.P = grid_bin_params(c, S);
            g.key = S.key.as<uint32_t>();
            g.cursor = S.cursor.as<uint32_t>();
            g.cell_count = S.cell_count.as<uint32_t>();
            g.cnt_pad = S.cnt_pad.as<uint32_t>();
            g.counters = pad_shift[s] ? g.cnt_pad : g.cell_count;
            g.tmp_key = S.tmp_key.as<uint32_t>();
            g.sorted = S.sorted.as<float4>();
            g.aabb = S.aabb.as<float4>();
            g.perm = S.perm.as<float4>();
            g.chunk_aabb = S.chunk_aabb.as<float4>();
            g.cell_org = S.cell_org.as<float4>();
            g.h16 = S.h16.as<uint4>();
            hG[s * HIST_BATCH + f] = g;
        }
    }
    c->set = keep_set;
    // ---- side stream: records, grids, plans
    hipStream_t ss = c->side_stream;
    if (c->gen_free[gen]) MH_HIP(hipStreamWaitEvent(ss, c->gen_free[gen], 0));
    MH_HIP(hipMemcpyAsync(c->hb_blocks[gen].p, pin, blk_bytes, hipMemcpyHostToDevice, ss));
    MH_HIP(hipEventRecord(c->hb_pin_ev[pslot], ss));
    c->hb_pin_used[pslot] = true;
    {
        const unsigned fw = (unsigned)W;
        for (int s = 0; s < nsets; ++s) {        // the grids of the group's first sets, then of its second sets
            const GridFrame *G = dG + s * HIST_BATCH;
            const uint32_t ns = n[s], ps = pad_shift[s];
            const size_t na = (size_t)ncells + 1, npad = ps ? ((size_t)ncells << ps) : 0;
            const unsigned zb = (unsigned)std::min<size_t>((na + npad + 255) / 256, 512u);
            // (padded counters: unpad_scan_frames_kernel leaves them zero; a generation's sets are zeroed when they are new)
            bool need_zero = !ps;
            for (int f = 0; f < W; ++f) {
                if (!c->hb_zeroed[gen][f][s] || c->hb_zeroed_cells[gen][f][s] != ncells) need_zero = true;
                c->hb_zeroed[gen][f][s] = ps != 0;
                c->hb_zeroed_cells[gen][f][s] = ncells;
            }
            if (need_zero) hipLaunchKernelGGL(zero_frames_kernel, dim3(zb, fw), dim3(256), 0, ss, G, na, npad);
            const bool tile_ok = ncells <= BIN_TILE_MAX_CELLS && (uint64_t)ns >= 16ull * ncells;
            if (tile_ok && ns >= (1u << 19))
                hipLaunchKernelGGL(bin_tile_frames_kernel<32>, dim3((ns + 256u * 32u - 1u) / (256u * 32u), fw), dim3(256), (size_t)ncells * 4, ss, G, ps, ncells);
            else if (tile_ok && ns >= (1u << 17))
                hipLaunchKernelGGL(bin_tile_frames_kernel<8>, dim3((ns + 256u * 8u - 1u) / (256u * 8u), fw), dim3(256), (size_t)ncells * 4, ss, G, ps, ncells);
            else
                hipLaunchKernelGGL(bin_frames_kernel, dim3((ns + 255u) / 256u, fw), dim3(256), 0, ss, G, ps);
            if (ps) hipLaunchKernelGGL(unpad_scan_frames_kernel, dim3(fw), dim3(256), 0, ss, G, ncells, ps);
            else hipLaunchKernelGGL(scan_frames_kernel, dim3(fw), dim3(64), 0, ss, G, ncells + 1u);
            hipLaunchKernelGGL(scatter_frames_kernel, dim3((ns + 255u) / 256u, fw), dim3(256), 0, ss, G, ns);
            hipLaunchKernelGGL(place_order_frames_kernel, dim3(ncells, fw), dim3(64), 0, ss, G, ncells, q->ids_local);
        }
        launch_hist_plan_frames(q->kind, ss, dP, fw, c->ntasks, c->hb_lean[gen].as<SlotDesc>(), c->hb_rest[gen].as<SlotDesc>(), queue, lslot);
        // The generic kernel (the triclinic corner entries of every frame of the group: short latency-bound slots) runs HERE, behind
        // its plan on the side stream, not behind the persistent kernel on the main stream: there it met the next group's placement
        // kernel - 10^4 one-wave workgroups of 5 KB of LDS each, from the stream of higher priority - and waited for it to drain
        // (30 us alone, 450 us in every second group).  It adds into the same bins with the same integer atomics.
        uint32_t nblk = ((uint32_t)bound + (uint32_t)waves_per_block(MODE_HIST) - 1u) / (uint32_t)waves_per_block(MODE_HIST);
        const uint32_t cap = (uint32_t)c->num_cus * 8u;
        if (nblk > cap) nblk = cap;
        if (two)
            launch_pair_double(MODE_HIST, nblk, dyn_lds, ss, dP, c->hb_rest[gen].as<SlotDesc>(), (uint32_t)bound, c->slot_cnt.as<uint32_t>(),
                               c->slot_base.as<unsigned long long>(), nullptr, nullptr, nullptr);
        else
            launch_pair_single(MODE_HIST, nblk, dyn_lds, ss, dP, c->hb_rest[gen].as<SlotDesc>(), (uint32_t)bound, c->slot_cnt.as<uint32_t>(),
                               c->slot_base.as<unsigned long long>(), nullptr, nullptr, nullptr);
        MH_HIP(hipGetLastError());
    }
    MH_HIP(hipEventRecord(c->grid_done, ss));
    // ---- main stream: the persistent kernel over the joint list
    MH_HIP(hipStreamWaitEvent(c->stream, c->grid_done, 0));
    {
        Prof prof(c, 3);
        launch_hist_lean(q->kind, (unsigned)c->num_cus, dyn_lds, c->stream, dP, c->hb_lean[gen].as<SlotDesc>(), (uint32_t)bound, queue, lslot, big);
        MH_HIP(hipGetLastError());
    }
    if (!c->gen_free[gen]) MH_HIP(hipEventCreateWithFlags(&c->gen_free[gen], hipEventDisableTiming));
    MH_HIP(hipEventRecord(c->gen_free[gen], c->stream));
    return 0;
}

int molar_hip_search_histogram_frames(molar_hip_ctx *c, const molar_hip_search_desc *q, size_t nframes, size_t xyz1_stride, size_t xyz2_stride,
                                      const float *boxes9, float hmin, float hmax, size_t nbins, uint64_t *bins) {
    if (!c || !q || !bins) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "search_histogram_frames: null argument");
    if (q->kind == MOLAR_HIP_SEARCH_WITHIN)
        return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "search_histogram: a within search has no distances");
    if (nbins == 0 || nbins > 8192) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "search_histogram: nbins must be in 1..8192");
    if (nframes == 0) return MOLAR_HIP_OK;
    if (!q->xyz1) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "search: xyz pointer is null");
    MH_HIP(hipSetDevice(c->device));
    // the batched form: fixed-cutoff kinds, periodic, everything the kernels read already in device memory, a context with its own streams
    const bool two = q->kind == MOLAR_HIP_SEARCH_DOUBLE;
    if (two && !q->xyz2) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "search: xyz pointer is null");
    const bool batch = (q->kind == MOLAR_HIP_SEARCH_SINGLE || two) && (q->box9 || boxes9) && c->own_stream && !c->env_no_side && is_device_ptr(bins) &&
                       is_device_ptr(q->xyz1) && (!q->idx1 || is_device_ptr(q->idx1)) &&
                       (!two || (is_device_ptr(q->xyz2) && (!q->idx2 || is_device_ptr(q->idx2)))) && !c->tickets[0].pending && !c->tickets[1].pending;
    size_t f = 0;
    while (f < nframes) {
        const int W = (int)std::min<size_t>(HIST_BATCH, nframes - f);
        int rc = 1;
        if (batch && W >= 2) {
            rc = hist_frames_group(c, q, f, W, xyz1_stride, xyz2_stride, boxes9, hmin, hmax, nbins, reinterpret_cast<unsigned long long *>(bins));
            if (rc < 0 || rc > 1) return rc;
        }
        if (rc == 1) {               // frame by frame (the form molar_hip_search_histogram documents), same sums
            for (int k = 0; k < W; ++k) {
                molar_hip_search_desc qf = *q;
                qf.xyz1 = q->xyz1 + (f + k) * xyz1_stride;
                if (q->xyz2) qf.xyz2 = q->xyz2 + (f + k) * xyz2_stride;
                if (boxes9) qf.box9 = boxes9 + 9 * (f + k);
                MH_TRY(molar_hip_search_histogram(c, &qf, hmin, hmax, nbins, bins, nullptr));
            }
        }
        f += (size_t)W;
    }
    return MOLAR_HIP_OK;
}

int molar_hip_within_count(molar_hip_ctx *c, const molar_hip_search_desc *q, uint64_t *out_count) {
    if (!c || !q) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "within: null argument");
    if (q->kind != MOLAR_HIP_SEARCH_WITHIN) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "within: the request must be of kind MOLAR_HIP_SEARCH_WITHIN");
    c->have_within = false;
    c->skip_plan = true;                          // no slots, no offsets: this path walks cells
    const int prc = prepare_search(c, q, /*size_masks=*/false);
    c->skip_plan = false;
    MH_TRY(prc);
    c->have_search = false;                       // not a cached search for the fill calls of the stream form
    const uint64_t nflags = q->ids_local ? c->set[0].n : (q->idx1 ? q->natoms1 : c->set[0].n);
    c->within_nflags = nflags;
    c->within_total = 0;
    if (out_count) *out_count = 0;
    if (c->set[0].n == 0 || c->set[1].n == 0 || nflags == 0) {
        c->have_within = true;
        return MOLAR_HIP_OK;
    }
    const uint32_t ncells = c->dims[0] * c->dims[1] * c->dims[2];
    const uint64_t ntiles = (nflags + 2047) / 2048;
    const size_t flag_bytes = (nflags + 3) & ~(size_t)3;
    const bool fresh_flags = c->w_flags.cap < flag_bytes;
    MH_TRY(c->w_flags.reserve(flag_bytes));
    MH_TRY(c->w_tile_cnt.reserve((ntiles + 1) * 4));
    MH_TRY(c->w_tile_off.reserve((ntiles + 1) * 8));
    MH_TRY(c->params.reserve(sizeof(SearchParams)));
    const SearchParams P = make_params(c);
    hipLaunchKernelGGL(upload_params_kernel, dim3(1), dim3(256), 0, c->stream, P, c->params.as<SearchParams>());
    // flags: all zero between calls.  A small-path call leaves only the flags of its list set: those are cleared through the list.
    if (fresh_flags || c->w_flags_all_dirty) MH_HIP(hipMemsetAsync(c->w_flags.p, 0, flag_bytes, c->stream));
    else if (c->w_list_dirty)
        hipLaunchKernelGGL(within_clear_kernel, dim3((unsigned)((c->w_list_dirty + 255) / 256)), dim3(256), 0, c->stream, c->w_list.as<uint32_t>() + 1,
                           (uint32_t)c->w_list_dirty, c->w_flags.as<uint8_t>());
    c->w_flags_all_dirty = false;
    c->w_list_dirty = 0;
    c->w_small = (uint64_t)c->set[1].n * 28ull <= (1ull << 17);           // <= 4681 second-set atoms: one wave per (atom, combination)
    if (c->w_small) {
        MH_TRY(c->w_list.reserve((nflags + 1) * 4));
        MH_HIP(hipMemsetAsync(c->w_list.p, 0, 4, c->stream));               // word 0: the list's length, then the ids
        // cells of many 64-row blocks (large cutoffs) get several waves per (cell, combination)
        uint32_t wsplit = (uint32_t)(((uint64_t)c->set[0].n / 64u) / ncells) + 1u;
        if (wsplit > 64u) wsplit = 64u;
        const uint32_t by_cell = ncells < c->set[1].n ? 1u : 0u;
        hipLaunchKernelGGL(within_small_kernel, dim3((by_cell ? ncells : c->set[1].n) * 28u, wsplit), dim3(64), 0, c->stream, c->params.as<SearchParams>(), c->set[1].n, ncells, by_cell,
                           c->w_flags.as<uint32_t>(), c->w_list.as<uint32_t>() + 1, c->w_list.as<uint32_t>());
        MH_HIP(hipGetLastError());
        uint32_t tot = 0;
        MH_TRY(read_back(c, &tot, c->w_list.p, 4));
        c->within_total = tot;
        c->w_list_dirty = tot;
        c->have_within = true;
        if (out_count) *out_count = tot;
        return MOLAR_HIP_OK;
    }
    c->w_flags_all_dirty = true;
    MH_TRY(c->w_part_cnt.reserve((size_t)ncells * 4));
    MH_TRY(c->w_part.reserve((size_t)ncells * WITHIN_MAX_PART * 4));
    MH_HIP(hipMemsetAsync(c->w_part_cnt.p, 0, (size_t)ncells * 4, c->stream));
    MH_HIP(hipMemsetAsync(c->w_tile_cnt.p, 0, (ntiles + 1) * 4, c->stream));
    hipLaunchKernelGGL(within_partners_kernel, dim3((unsigned)((c->ntasks + 255) / 256)), dim3(256), 0, c->stream, c->params.as<SearchParams>(),
                       c->w_part_cnt.as<uint32_t>(), c->w_part.as<uint32_t>());
    // big cells get several waves (a share of their 64-row blocks each)
    uint32_t nsplit = (uint32_t)(((uint64_t)c->set[0].n / 64u) / ncells) + 1u;
    if (nsplit > 64u) nsplit = 64u;
    hipLaunchKernelGGL(within_flags_kernel, dim3(ncells, nsplit), dim3(64), 0, c->stream, c->params.as<SearchParams>(),
                       c->w_part_cnt.as<uint32_t>(), c->w_part.as<uint32_t>(), c->w_flags.as<uint8_t>(), nsplit);
    hipLaunchKernelGGL(flag_tile_count_kernel, dim3((unsigned)ntiles), dim3(256), 0, c->stream, c->w_flags.as<uint8_t>(), nflags,
                       c->w_tile_cnt.as<uint32_t>());
    MH_HIP(hipGetLastError());
    MH_TRY((exclusive_scan<uint32_t, unsigned long long>(c, c->w_tile_cnt.as<uint32_t>(), c->w_tile_off.as<unsigned long long>(), ntiles + 1)));
    unsigned long long tot = 0;
    MH_TRY(read_back(c, &tot, c->w_tile_off.as<unsigned long long>() + ntiles, 8));
    c->within_total = tot;
    c->have_within = true;
    if (out_count) *out_count = tot;
    return MOLAR_HIP_OK;
}

// The first set of consecutive `within` requests stays the same while a selection expression is evaluated against one frame
// (`within 0.3 of resid 1`, `within 0.8 of resid 1`, ...: within_size_bench.rs:13-47 asks 1600 times): with the hold on, a
// request that names the same first set (same pointers and sizes, same box and periodicity) and comes to the same grid
// reuses the staged coordinates and the grid of the one before.  The CALLER promises that the first set's coordinates do not
// change while the hold is on (in Rust: for as long as it holds the `&State`); any other search on the context ends the reuse.
int molar_hip_within_hold(molar_hip_ctx *c, int on) {
    if (!c) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "null context");
    c->within_hold = on != 0;
    if (!on) c->hold_valid = false;
    return MOLAR_HIP_OK;
}

int molar_hip_within_fill(molar_hip_ctx *c, uint64_t *ids) {
    if (!c || !c->have_within) return fail(MOLAR_HIP_ERR_NO_SEARCH, "no cached within set: call molar_hip_within_count first");
    if (c->within_total == 0) return MOLAR_HIP_OK;
    if (!ids) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "within_fill: null output");
    MH_HIP(hipSetDevice(c->device));
    const bool dev = is_device_ptr(ids);
    // (the list of a small second set with a LARGE cutoff is long - 1.9e4 ids for `within 2.5 of 210 atoms` in a 100k-atom box: sorting it
    // on the host took 0.25 ms of a 0.68 ms call; from 4096 ids on, the flags are compacted on the device like the large path's)
    if (c->w_small && !dev && c->within_total <= 4096) {           // the list IS the set: bring it over, sort it (SortedSet::from_unsorted, selection_expr.rs:112), widen it
        std::vector<uint32_t> l((size_t)c->within_total);
        MH_TRY(read_back(c, l.data(), c->w_list.as<uint32_t>() + 1, l.size() * 4));
        std::sort(l.begin(), l.end());
        for (size_t k = 0; k < l.size(); ++k) ids[k] = l[k];
        return MOLAR_HIP_OK;
    }
    unsigned long long *dst = reinterpret_cast<unsigned long long *>(ids);
    if (c->w_small) {                   // device output: compact the flags (they hold the same set)
        const uint64_t ntiles = (c->within_nflags + 2047) / 2048;
        MH_HIP(hipMemsetAsync(c->w_tile_cnt.p, 0, (ntiles + 1) * 4, c->stream));
        hipLaunchKernelGGL(flag_tile_count_kernel, dim3((unsigned)ntiles), dim3(256), 0, c->stream, c->w_flags.as<uint8_t>(), c->within_nflags,
                           c->w_tile_cnt.as<uint32_t>());
        MH_TRY((exclusive_scan<uint32_t, unsigned long long>(c, c->w_tile_cnt.as<uint32_t>(), c->w_tile_off.as<unsigned long long>(), ntiles + 1)));
    }
    if (!dev) {
        MH_TRY(c->wide_i.reserve((size_t)c->within_total * 8));
        dst = c->wide_i.as<unsigned long long>();
    }
    const uint64_t ntiles = (c->within_nflags + 2047) / 2048;
    hipLaunchKernelGGL(flag_compact_kernel, dim3((unsigned)ntiles), dim3(256), 0, c->stream, c->w_flags.as<uint8_t>(), c->within_nflags,
                       c->w_tile_off.as<unsigned long long>(), dst);
    MH_HIP(hipGetLastError());
    if (!dev) {
        MH_HIP(hipMemcpyAsync(ids, dst, (size_t)c->within_total * 8, hipMemcpyDeviceToHost, c->stream));
        MH_HIP(hipStreamSynchronize(c->stream));
    }
    return MOLAR_HIP_OK;
}

// SearchConnectivity (connectivity.rs:8-60) of a single-selection search, built on the device from the resident pair list:
// CSR over the id range of the request (local ids: the selection's length; global ids: natoms), lists in the reference's push
// order.  Count-then-fill: the first call runs the search and builds the CSR in context-owned device memory and returns the
// number of entries (2 x pairs); the second copies offsets (rows + 1) and neighbours to host or device memory.
int molar_hip_search_connectivity(molar_hip_ctx *c, const molar_hip_search_desc *q, uint64_t *out_rows, uint64_t *out_entries) {
    if (!c || !q) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "search_connectivity: null argument");
    if (q->kind != MOLAR_HIP_SEARCH_SINGLE)
        return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "search_connectivity: ids of one selection index the lists - the request must be of kind MOLAR_HIP_SEARCH_SINGLE");
    c->have_conn = false;
    uint64_t npairs = 0;
    const uint32_t *d_pairs = nullptr;
    {   // the lists hold ids only: the (i, j) plane alone (DistanceSearchOutput of (usize, usize), distance_search.rs:14-20)
        const bool keep = c->resident_no_dist;
        c->resident_no_dist = true;
        const int rc = molar_hip_search_resident(c, q, &npairs, &d_pairs, nullptr);
        c->resident_no_dist = keep;
        MH_TRY(rc);
    }
    const size_t nsel = q->idx1 ? q->n1 : q->natoms1;
    const uint64_t nrows = q->ids_local ? nsel : (q->idx1 ? q->natoms1 : nsel);
    if (nrows >= 0xFFFFFFF0ull) return fail(MOLAR_HIP_ERR_TOO_LARGE, "search_connectivity: too many rows");
    const uint64_t nent = 2ull * npairs;
    if (nent >= 0x7FFFFFFFull) return fail(MOLAR_HIP_ERR_TOO_LARGE, "search_connectivity: %llu list entries", (unsigned long long)nent);
    MH_TRY(c->conn_off.reserve((nrows + 1) * 8));
    MH_TRY(c->conn_ent.reserve((nent ? nent : 1) * 16));           // four u32 arrays: rows and neighbours, unsorted and sorted
    MH_TRY(c->conn_neigh.reserve((nent ? nent : 1) * 8));
    uint32_t *row_in = c->conn_ent.as<uint32_t>(), *nb_in = row_in + nent, *row_out = nb_in + nent, *nb_out = row_out + nent;
    const unsigned nbP = (unsigned)((npairs + 255) / 256), nbE = (unsigned)((nent + 255) / 256);
    const uint2 *pairs = reinterpret_cast<const uint2 *>(d_pairs);
    if (nbP) {
        hipLaunchKernelGGL(conn_entries_kernel, dim3(nbP), dim3(256), 0, c->stream, pairs, (unsigned long long)npairs, row_in, nb_in);
        int end_bit = 1;
        while (end_bit < 32 && (nrows >> end_bit)) ++end_bit;
        MH_TRY(device_sort_pairs_u32(c, c->conn_deg, row_in, row_out, nb_in, nb_out, (size_t)nent, end_bit));
        hipLaunchKernelGGL(conn_widen_kernel, dim3(nbE), dim3(256), 0, c->stream, nb_out, (unsigned long long)nent, c->conn_neigh.as<unsigned long long>());
    }
    hipLaunchKernelGGL(conn_offsets_kernel, dim3((unsigned)((nrows + 1 + 255) / 256)), dim3(256), 0, c->stream, row_out, (unsigned long long)nent,
                       (uint32_t)nrows, c->conn_off.as<unsigned long long>());
    MH_HIP(hipGetLastError());
    c->conn_rows = nrows;
    c->conn_entries = nent;
    c->have_conn = true;
    if (out_rows) *out_rows = nrows;
    if (out_entries) *out_entries = nent;
    return MOLAR_HIP_OK;
}

int molar_hip_search_connectivity_fill(molar_hip_ctx *c, uint64_t *offsets, uint64_t *neigh) {
    if (!c || !c->have_conn) return fail(MOLAR_HIP_ERR_NO_SEARCH, "no cached connectivity: call molar_hip_search_connectivity first");
    MH_HIP(hipSetDevice(c->device));
    if (offsets) MH_HIP(hipMemcpyAsync(offsets, c->conn_off.p, (c->conn_rows + 1) * 8, hipMemcpyDefault, c->stream));
    if (neigh && c->conn_entries) {
        // large lists into ordinary host memory: the pinned ring and its host threads (hoststream.hpp), as the pair lists travel
        // (25k atoms at rc 1.0 nm: 83 MB of neighbours, 11 ms through the runtime's pageable copy)
        if (!is_device_ptr(neigh) && ring_pays(c->conn_entries * 8, neigh)) {
            MH_HIP(hipStreamSynchronize(c->stream));
            std::vector<RingJob> jobs;
            jobs.push_back(RingJob{c->conn_neigh.p, (size_t)c->conn_entries * 8, RING_COPY, neigh, nullptr});
            return ring_to_host(c, jobs);
        }
        MH_HIP(hipMemcpyAsync(neigh, c->conn_neigh.p, c->conn_entries * 8, hipMemcpyDefault, c->stream));
    }
    MH_HIP(hipStreamSynchronize(c->stream));
    return MOLAR_HIP_OK;
}

// Modify::unwrap_connectivity_dim (molar/src/modify.rs:72-131).  The neighbour search - the heavy part - runs on the GPU
// (distance_search_single_pbc over the selection with LOCAL ids under full PBC, :77-78); the adjacency lists in push
// order (SearchConnectivity::from_iter, connectivity.rs:19-35: for (i, j) in pair order conn[i].push(j), conn[j].push(i))
// are built on the device from the resident pair list (molar_hip_search_connectivity), and the reference's stack walk (:84-128) - serial by nature: every
// atom is pulled to the closest image of the atom it was REACHED FROM, whose position the walk may just have changed -
// runs on the host over that CSR with the same f32 arithmetic (boxmath.hpp's closest_image, the one the kernels use).
// Quirks kept: the atom a component starts from (0, then the lowest unused index) is not a member of the selection
// the component returns (:97-98,111-113); a component of one atom returns no selection; members are emitted as the
// reference's `select(&sel_vec)` makes them, sorted.
int molar_hip_unwrap_connectivity(molar_hip_ctx *c, float *xyz, size_t natoms, const uint64_t *idx, size_t n, const float *box9,
                                  float cutoff, uint8_t pbc_dims, uint64_t *group_offsets, uint64_t *group_ids, size_t *ngroups) {
    if (!c || !xyz) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "unwrap_connectivity: null argument");
    if (!box9) return fail(MOLAR_HIP_ERR_NO_PBC, "no periodic box");                               // require_box (:76)
    const size_t nsel = idx ? n : natoms;
    if (nsel == 0) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "unwrap_connectivity of an empty selection");
    if (nsel >= 0x7FFFFFFFull) return fail(MOLAR_HIP_ERR_TOO_LARGE, "unwrap_connectivity: %zu atoms exceed the 32-bit local ids", nsel);
    MH_HIP(hipSetDevice(c->device));
    molar_hip_box b;
    MH_TRY(molar_hip_box_from_matrix(box9, &b));
    // ---- the search, local ids (0..len), full PBC whatever `dims` (:77-78)
    molar_hip_search_desc q{};
    q.kind = MOLAR_HIP_SEARCH_SINGLE;
    q.cutoff = cutoff;
    q.xyz1 = xyz; q.natoms1 = natoms; q.idx1 = idx; q.n1 = n;
    q.ids_local = 1;
    q.box9 = box9;
    q.pbc = MOLAR_HIP_PBC_FULL;
    // ---- adjacency in push order: SearchConnectivity on the device, only the CSR comes to the host
    uint64_t nrows = 0, nent = 0;
    MH_TRY(molar_hip_search_connectivity(c, &q, &nrows, &nent));
    std::vector<uint64_t> off(nsel + 1, 0), adj((size_t)(nent ? nent : 1));
    MH_TRY(molar_hip_search_connectivity_fill(c, off.data(), adj.data()));
    // ---- the coordinates of the frame on the host
    const bool dev = is_device_ptr(xyz);
    std::vector<float> hostcopy;
    float *h = xyz;
    if (dev) {
        hostcopy.resize(natoms * 3);
        MH_HIP(hipMemcpyAsync(hostcopy.data(), xyz, natoms * 12, hipMemcpyDeviceToHost, c->stream));
        MH_HIP(hipStreamSynchronize(c->stream));
        h = hostcopy.data();
    }
    std::vector<uint64_t> hidx;
    if (idx && is_device_ptr(idx)) {      // the walk reads the selection on the host
        hidx.resize(n);
        MH_HIP(hipMemcpy(hidx.data(), idx, n * 8, hipMemcpyDeviceToHost));
    }
    const uint64_t *ix = hidx.empty() ? idx : hidx.data();
    auto pos = [&](size_t k) -> float * { return h + 3 * (ix ? ix[k] : (uint64_t)k); };
    // ---- the walk (:80-128)
    std::vector<uint8_t> used(nsel, 0);
    std::vector<uint32_t> todo, sel_vec;
    todo.reserve(1024);
    size_t ng = 0, nids = 0, first_unused = 0;
    if (group_offsets) group_offsets[0] = 0;
    auto emit = [&]() {
        if (sel_vec.empty()) return;
        std::sort(sel_vec.begin(), sel_vec.end());
        if (group_ids) for (uint32_t v : sel_vec) group_ids[nids++] = v;
        else nids += sel_vec.size();
        ++ng;
        if (group_offsets) group_offsets[ng] = nids;
        sel_vec.clear();
    };
    todo.push_back(0);
    used[0] = 1;
    const uint32_t dims = pbc_dims & 7u;
    for (;;) {
        while (!todo.empty()) {
            const uint32_t cc = todo.back();
            todo.pop_back();
            const float *pc = pos(cc);
            const V3 p0 = v3(pc[0], pc[1], pc[2]);
            for (uint64_t e = off[cc]; e < off[cc + 1]; ++e) {
                const uint32_t ind = (uint32_t)adj[e];
                if (used[ind]) continue;
                float *pp = pos(ind);
                const V3 r = closest_image(b, v3(pp[0], pp[1], pp[2]), p0, dims);
                pp[0] = r.x; pp[1] = r.y; pp[2] = r.z;
                todo.push_back(ind);
                used[ind] = 1;
                sel_vec.push_back(ind);
            }
        }
        while (first_unused < nsel && used[first_unused]) ++first_unused;       // used.iter().find_position(false)
        if (first_unused == nsel) {
            emit();
            break;
        }
        todo.push_back((uint32_t)first_unused);
        used[first_unused] = 1;
        emit();
    }
    if (ngroups) *ngroups = ng;
    if (dev) {
        MH_HIP(hipMemcpyAsync(xyz, hostcopy.data(), natoms * 12, hipMemcpyHostToDevice, c->stream));
        MH_HIP(hipStreamSynchronize(c->stream));
    }
    return MOLAR_HIP_OK;
}

}  // extern "C"


