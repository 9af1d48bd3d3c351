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

using namespace mh;

namespace {

constexpr int WAVES_PER_BLOCK = 4;
constexpr int BLOCK = 64 * WAVES_PER_BLOCK;
constexpr uint32_t DROPPED = 0xFFFFFFFFu;
constexpr int KREG = 8;            // B-cell chunks (of 64 atoms) a lane keeps in registers
constexpr int FIFO_CAP = 128;      // per-wave LDS FIFO entries (flush threshold 64, push <= 64)
constexpr float F32_EPS = 1.1920929e-07f;

// distance_search.rs:39-60
__constant__ uint8_t MASKS[14][6] = {
    {0, 0, 0, 0, 0, 0},
    {0, 0, 0, 1, 0, 0}, {0, 0, 0, 0, 1, 0}, {0, 0, 0, 0, 0, 1},
    {0, 0, 0, 1, 1, 0}, {0, 0, 0, 1, 0, 1}, {0, 0, 0, 0, 1, 1},
    {0, 0, 0, 1, 1, 1},
    {1, 0, 0, 0, 1, 0}, {1, 0, 0, 0, 0, 1}, {0, 1, 0, 0, 0, 1},
    {1, 1, 0, 0, 0, 1}, {1, 0, 1, 0, 1, 0}, {0, 1, 1, 1, 0, 0},
};

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

__global__ void __launch_bounds__(256) bin_kernel(BinParams P, uint32_t *__restrict__ key,
                                                  uint32_t *__restrict__ arrival, uint32_t *__restrict__ cell_count) {
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= P.n) return;
    const uint64_t a = P.idx ? P.idx[k] : (uint64_t)k;
    const CellOfAtom c = classify(P, load_pos(P.xyz, a));
    key[k] = c.key;
    // arrival order inside the cell (arbitrary): lets scatter place the atom without a second atomic
    if (c.key != DROPPED) arrival[k] = atomicAdd(&cell_count[c.key >> 1], 1u);
}

__global__ void __launch_bounds__(256) scatter_kernel(uint32_t n, const uint32_t *__restrict__ key,
                                                      const uint32_t *__restrict__ cell_start,
                                                      const uint32_t *__restrict__ arrival,
                                                      uint32_t *__restrict__ tmp_key, uint32_t *__restrict__ tmp_cell) {
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= n) return;
    const uint32_t ky = key[k];
    if (ky == DROPPED) return;
    const uint32_t cell = ky >> 1;
    const uint32_t pos = cell_start[cell] + arrival[k];
    tmp_key[pos] = ((ky & 1u) << 31) | k;     // sort key inside the cell: in-box first, then input order
    tmp_cell[pos] = cell;
}

__global__ void __launch_bounds__(256) place_kernel(BinParams P, uint32_t ncells, int ids_local,
                                                    const uint32_t *__restrict__ cell_start,
                                                    const uint32_t *__restrict__ tmp_key,
                                                    const uint32_t *__restrict__ tmp_cell,
                                                    const float *__restrict__ vdw, float4 *__restrict__ sorted,
                                                    float *__restrict__ sorted_vdw) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= cell_start[ncells]) return;
    const uint32_t mine = tmp_key[t];
    const uint32_t cell = tmp_cell[t];
    const uint32_t s = cell_start[cell], e = cell_start[cell + 1];
    uint32_t rank = 0;
#pragma unroll 8
    for (uint32_t q = s; q < e; ++q) rank += tmp_key[q] < mine ? 1u : 0u;   // 8 loads in flight per thread
    const uint32_t k = mine & 0x7FFFFFFFu;
    const uint64_t a = P.idx ? P.idx[k] : (uint64_t)k;
    const CellOfAtom c = classify(P, load_pos(P.xyz, a));   // same arithmetic as bin_kernel
    const uint32_t id = ids_local ? k : (uint32_t)a;
    sorted[s + rank] = make_float4(c.pos.x, c.pos.y, c.pos.z, __uint_as_float(id));
    if (vdw) sorted_vdw[s + rank] = vdw[k];
}

// Axis-aligned bounding box of the positions stored in each cell (lab frame): aabb[2c] = lo,
// aabb[2c+1] = hi.  One wave per cell.  Lets a row of the pair kernels prove "no atom of the other
// cell can be within the cutoff" and skip its candidates (see run_plain).
__global__ void __launch_bounds__(256) cell_aabb_kernel(uint32_t ncells, const uint32_t *__restrict__ cell_start,
                                                        const float4 *__restrict__ sorted, float4 *__restrict__ aabb) {
    const uint32_t c = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (c >= ncells) return;
    const uint32_t s = cell_start[c], e = cell_start[c + 1];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t q = s + lane; q < e; q += 64u) {
        const float4 p = sorted[q];
        lo[0] = fminf(lo[0], p.x); hi[0] = fmaxf(hi[0], p.x);
        lo[1] = fminf(lo[1], p.y); hi[1] = fmaxf(hi[1], p.y);
        lo[2] = fminf(lo[2], p.z); hi[2] = fmaxf(hi[2], p.z);
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
}

// ================================================================= scans (exclusive, n elements)

constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = 256 * SCAN_ITEMS;

template <class T>
__device__ __forceinline__ T block_exclusive_scan(T v, T *total) {
    __shared__ T wave_sums[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    T inc = v;
    for (int off = 1; off < 64; off <<= 1) {
        T o = __shfl_up(inc, off, 64);
        if (lane >= off) inc += o;
    }
    if (lane == 63) wave_sums[wave] = inc;
    __syncthreads();
    T base = 0, tot = 0;
    for (int w = 0; w < 4; ++w) {
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

template <class TIn, class TOut>
int exclusive_scan(molar_hip_ctx *c, const TIn *in, TOut *out, uint64_t n) {
    if (n == 0) return 0;
    if (n <= 8192ull) {
        hipLaunchKernelGGL((scan_small_kernel<TIn, TOut>), dim3(1), dim3(256), 0, c->stream, in, out, (uint32_t)n);
        MH_HIP(hipGetLastError());
        return 0;
    }
    const uint64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    MH_TRY(c->scan_tmp.reserve(nb * sizeof(TOut)));
    TOut *sums = c->scan_tmp.as<TOut>();
    hipLaunchKernelGGL((scan_tile_kernel<TIn, TOut>), dim3((unsigned)nb), dim3(256), 0, c->stream, in, out, sums, n);
    if (nb > 1) {
        hipLaunchKernelGGL((scan_sums_kernel<TOut>), dim3(1), dim3(256), 0, c->stream, sums, nb);
        hipLaunchKernelGGL((scan_add_kernel<TOut>), dim3((unsigned)nb), dim3(256), 0, c->stream, out, sums, n);
    }
    MH_HIP(hipGetLastError());
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

// vdw.iter().cloned().reduce(Float::max) (distance_search.rs:781-782)
__global__ void __launch_bounds__(256) fmax_kernel(const float *__restrict__ v, uint32_t n, uint32_t *__restrict__ out) {
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
    const bool wave_any = __ballot(any) != 0ull;
    if ((threadIdx.x & 63) == 0 && wave_any) atomicMax(out, f2ord(m));
}

// ================================================================= pair kernels

struct SearchParams {
    const float4 *sa;        // cell-sorted atoms of set 1
    const float4 *sb;        // cell-sorted atoms of set 2 (== sa for SINGLE)
    const uint32_t *csa;     // cell_start of set 1
    const uint32_t *csb;     // cell_start of set 2
    const float *vdwa;
    const float *vdwb;
    const float4 *aabb_b;    // per-cell bounding boxes of set 2 (== set 1 for SINGLE)
    uint32_t dx, dy, dz;
    uint32_t pbc;            // PbcDims of the plan (0 for the non-periodic drivers)
    uint32_t use_box;
    uint32_t nblocks;        // launch grid
    uint32_t wrap_kind;      // WK_* of the box matrix (zero pattern of m and inv)
    uint32_t prune_wrapped;  // wrapped entries may use the image-box row pruning (all periodic dims >= 3 cells)
    float prune_limit2;      // (cutoff + margin)^2 for that pruning
    uint32_t debug_skip;     // profiling aid (env MOLAR_HIP_DEBUG_SKIP): bit0 plain, bit1 wrapped, bit2 triangular slots do nothing
    float cutoff2;
    uint64_t ntasks;
    molar_hip_box box;
};

struct Task {
    uint32_t a0, n1, b0, n2;
    uint32_t rps;             // rows of the first cell per slot: 64, or 8 for entries that run the
                              // triclinic candidate loop (~50x the arithmetic per candidate)
    uint32_t cb;              // second cell (for its bounding box)
    uint32_t wrap;
    bool tri;
    bool valid;
};

// search_plan (distance_search.rs:217-269).  Task index == position in the reference's plan
// enumeration (x outer, z inner, 14 masks; for two-grid searches each entry is two tasks:
// (c1,c2) then (c2,c1), :686-693).  Entries the reference filters out because a cell is empty
// produce zero results here, so no compaction of the plan is needed.
template <int KIND, bool UNIFORM>
__device__ __forceinline__ Task decode_task(const SearchParams &P, uint64_t t) {
    Task T;
    T.valid = false;
    T.tri = false;
    T.a0 = T.b0 = T.n1 = T.n2 = 0;
    T.cb = 0;
    T.rps = 64u;
    T.wrap = 0;
    uint32_t half = 0;
    uint64_t e = t;
    if (KIND != MOLAR_HIP_SEARCH_SINGLE) {
        half = (uint32_t)(t & 1ull);
        e = t >> 1;
    }
    const uint32_t m = (uint32_t)(e % 14ull);
    const uint64_t cidx = e / 14ull;
    const uint32_t z = (uint32_t)(cidx % P.dz);
    const uint64_t r = cidx / P.dz;
    const uint32_t y = (uint32_t)(r % P.dy);
    const uint32_t x = (uint32_t)(r / P.dy);
    const uint32_t dims[3] = {P.dx, P.dy, P.dz};
    uint32_t c[2][3] = {{x + MASKS[m][0], y + MASKS[m][1], z + MASKS[m][2]},
                        {x + MASKS[m][3], y + MASKS[m][4], z + MASKS[m][5]}};
    uint32_t wrap = 0;
    for (int i = 0; i < 2; ++i)
        for (int d = 0; d < 3; ++d)
            if (c[i][d] == dims[d]) {
                if ((P.pbc >> d) & 1u) {
                    c[i][d] = 0;
                    wrap |= 1u << d;
                } else {
                    return T;   // non-periodic dimension: entry dropped (:241-244)
                }
            }
    const uint32_t i1 = c[0][0] + c[0][1] * P.dx + c[0][2] * P.dx * P.dy;
    const uint32_t i2 = c[1][0] + c[1][1] * P.dx + c[1][2] * P.dx * P.dy;
    uint32_t ca = i1, cb = i2;
    if (KIND == MOLAR_HIP_SEARCH_SINGLE) {
        T.tri = (i1 == i2);
    } else if (half) {
        ca = i2;
        cb = i1;
    }
    T.a0 = P.csa[ca];
    T.n1 = P.csa[ca + 1] - T.a0;
    T.b0 = P.csb[cb];
    T.n2 = P.csb[cb + 1] - T.b0;
    T.wrap = wrap;
    // only the entries of the single home cell (dx-1,dy-1,dz-1) can wrap in all three dims: <= 28 tasks
    T.rps = (P.use_box && wrap == MOLAR_HIP_PBC_FULL && P.box.nshift != 0 && T.n1 <= 4096u) ? 8u : 64u;
    T.cb = cb;
    if (UNIFORM) {
        T.cb = __builtin_amdgcn_readfirstlane(T.cb);
        T.rps = __builtin_amdgcn_readfirstlane(T.rps);   // one task per wave: keep the descriptor in SGPRs
        T.a0 = __builtin_amdgcn_readfirstlane(T.a0);
        T.n1 = __builtin_amdgcn_readfirstlane(T.n1);
        T.b0 = __builtin_amdgcn_readfirstlane(T.b0);
        T.n2 = __builtin_amdgcn_readfirstlane(T.n2);
        T.wrap = __builtin_amdgcn_readfirstlane(T.wrap);
    }
    T.valid = T.n1 > 0 && T.n2 > 0 && !(T.tri && T.n1 < 2);
    return T;
}

typedef float v2f __attribute__((ext_vector_type(2)));

// How a wrapped cell pair evaluates PeriodicBox::distance_squared (periodic_box.rs:286-318,379-381).
// The matrix kinds only skip products with entries that are exactly 0.0 in BOTH the box matrix
// and its inverse: x + (0*y) == x in IEEE arithmetic (up to the sign of a zero result, which the
// final sum of squares cannot see), so every kind yields the same d2 bits as the general form.
enum { WK_NONE = 0, WK_DIAG = 1, WK_UPPER = 2, WK_GENERAL = 3 };

__device__ __forceinline__ v2f round_away2(v2f f) { return v2f{__builtin_roundf(f.x), __builtin_roundf(f.y)}; }

// squared distances from the broadcast atom p (SGPR operands) to TWO second-cell atoms per lane
// (one of each 64-chunk of a chunk pair): packed f32 math, two candidates per VALU instruction.
//   plain   : |p2-p1|^2 = ((dx*dx)+(dy*dy))+(dz*dz)                                   (:488)
//   wrapped : f = inv*v; f[d] -= round(f[d]) for the entry's wrap dims; s = M*f;
//             triclinic candidate loop only if shifts exist and all three dims wrap (:304)
struct BoxRegs {
    float I[9], M[9];   // inverse and matrix, column-major, held in SGPRs (statically indexed only)
    int nshift;
};

template <int WK>
__device__ __forceinline__ v2f pair_d2x2(const SearchParams &P, const BoxRegs &B, uint32_t wrap, float px, float py,
                                         float pz, v2f qx, v2f qy, v2f qz) {
    const v2f vx = qx - px, vy = qy - py, vz = qz - pz;
    if (WK == WK_NONE) return (vx * vx + vy * vy) + vz * vz;
    const float *I = B.I, *M = B.M;
    v2f fx, fy, fz;
    if (WK == WK_DIAG) {
        fx = I[0] * vx;
        fy = I[4] * vy;
        fz = I[8] * vz;
    } else if (WK == WK_UPPER) {
        fx = (I[0] * vx + I[3] * vy) + I[6] * vz;
        fy = I[4] * vy + I[7] * vz;
        fz = I[8] * vz;
    } else {
        fx = (I[0] * vx + I[3] * vy) + I[6] * vz;
        fy = (I[1] * vx + I[4] * vy) + I[7] * vz;
        fz = (I[2] * vx + I[5] * vy) + I[8] * vz;
    }
    if (wrap & 1u) fx -= round_away2(fx);
    if (wrap & 2u) fy -= round_away2(fy);
    if (wrap & 4u) fz -= round_away2(fz);
    v2f sx, sy, sz;
    if (WK == WK_DIAG) {
        sx = M[0] * fx;
        sy = M[4] * fy;
        sz = M[8] * fz;
    } else if (WK == WK_UPPER) {
        sx = (M[0] * fx + M[3] * fy) + M[6] * fz;
        sy = M[4] * fy + M[7] * fz;
        sz = M[8] * fz;
    } else {
        sx = (M[0] * fx + M[3] * fy) + M[6] * fz;
        sy = (M[1] * fx + M[4] * fy) + M[7] * fz;
        sz = (M[2] * fx + M[5] * fy) + M[8] * fz;
    }
    v2f best2 = (sx * sx + sy * sy) + sz * sz;
    if (WK != WK_DIAG && B.nshift != 0 && wrap == MOLAR_HIP_PBC_FULL) {
        for (int k = 0; k < B.nshift; ++k) {
            const v2f cx = sx + P.box.shifts[3 * k], cy = sy + P.box.shifts[3 * k + 1], cz = sz + P.box.shifts[3 * k + 2];
            const v2f n2 = (cx * cx + cy * cy) + cz * cz;
            // `best` itself is only needed through its norm: cand = start + s is always formed from
            // `start`, not from the running best (:310), so tracking best2 is enough
            best2.x = n2.x < best2.x ? n2.x : best2.x;
            best2.y = n2.y < best2.y ? n2.y : best2.y;
        }
    }
    return best2;
}

// Per-wave output FIFO of the fill pass.
struct Fifo {
    uint32_t *fi, *fj, *fd;     // LDS, FIFO_CAP entries each
    uint32_t head, tail;        // monotonically increasing, wave-uniform
    uint64_t base;              // output offset of this task
    uint2 *pairs;
    float *dist;
    uint32_t *ids;              // WITHIN output
};

template <int KIND>
__device__ __forceinline__ void fifo_flush(Fifo &F, uint32_t count, uint32_t lane) {
    if (lane < count) {
        const uint32_t s = (F.head + lane) & (FIFO_CAP - 1);
        const uint64_t pos = F.base + F.head + lane;
        if (KIND == MOLAR_HIP_SEARCH_WITHIN) {
            F.ids[pos] = F.fi[s];
        } else {
            if (F.pairs) F.pairs[pos] = make_uint2(F.fi[s], F.fj[s]);
            // d2.sqrt() (:448): llvm.sqrt.f32 without fpmath metadata = IEEE correctly rounded
            if (F.dist) F.dist[pos] = __builtin_sqrtf(__uint_as_float(F.fd[s]));
        }
    }
    F.head += count;
}

// One task = one ordered block of the reference's output:
//   search_cell_pair_single(_pbc) :432-517, _double(_pbc) :324-373, _vdw(_pbc) :375-430,
//   _within(_pbc) :271-322.
// j (second cell): one atom per lane per 64-chunk, NCH chunks resident in registers as packed
// pairs (NCH = 0: cells larger than KREG*64 atoms are re-read from memory per row).
// i (first cell): 64 atoms at a time are loaded one-per-lane and each row's atom is broadcast
// with v_readlane into SGPRs, so the distance arithmetic takes scalar operands and no per-row
// memory access sits on the critical path.
// RUNTIME_NCH: the chunk count is checked at run time (triangular tasks, which skip chunks).
template <int KIND, bool FILL, int WK, bool TRI, int NCH, bool RUNTIME_NCH>
__device__ __forceinline__ uint32_t run_task(const SearchParams &P, const Task &T, uint32_t i0, Fifo &F, uint32_t lane) {
    constexpr bool VDW = KIND == MOLAR_HIP_SEARCH_DOUBLE_VDW;
    constexpr bool WITHIN = KIND == MOLAR_HIP_SEARCH_WITHIN;
    constexpr bool STREAM = NCH == 0;
    constexpr int NPAIR = STREAM ? 1 : (NCH + 1) / 2;
    uint32_t total = 0;
    BoxRegs B;
    B.nshift = 0;
    if (WK != WK_NONE) {
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            B.I[k] = P.box.inv[k];
            B.M[k] = P.box.m[k];
        }
        B.nshift = P.box.nshift;
    }
    const float cutoff2 = P.cutoff2;

    v2f bx[NPAIR], by[NPAIR], bz[NPAIR], bv[NPAIR];
    uint32_t bid[2 * NPAIR];
    auto load_b = [&](uint32_t jj, float &x, float &y, float &z, float &v, uint32_t &id) {
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
        float rv = 0.f;
        if (jj < T.n2) {
            q = P.sb[T.b0 + jj];
            if (VDW) rv = P.vdwb[T.b0 + jj];
        }
        x = q.x; y = q.y; z = q.z; v = rv; id = __float_as_uint(q.w);
    };
    if (!STREAM) {
#pragma unroll
        for (int h = 0; h < NPAIR; ++h) {
            float x0, y0, z0, v0, x1, y1, z1, v1;
            load_b((uint32_t)(2 * h) * 64u + lane, x0, y0, z0, v0, bid[2 * h]);
            load_b((uint32_t)(2 * h + 1) * 64u + lane, x1, y1, z1, v1, bid[2 * h + 1]);
            bx[h] = v2f{x0, x1}; by[h] = v2f{y0, y1}; bz[h] = v2f{z0, z1}; bv[h] = v2f{v0, v1};
        }
    }
    const uint32_t nchunks = (T.n2 + 63u) >> 6;

    {   // one slot = rows [i0, i0+64) of the first cell
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        float ra = 0.f;
        if (i0 + lane < T.n1) {
            a = P.sa[T.a0 + i0 + lane];
            if (VDW) ra = P.vdwa[T.a0 + i0 + lane];
        }
        const uint32_t rows = T.n1 - i0 < T.rps ? T.n1 - i0 : T.rps;
        for (uint32_t r = 0; r < rows; ++r) {
            const uint32_t i = i0 + r;
            const float px = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a.x), r));
            const float py = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a.y), r));
            const float pz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a.z), r));
            const uint32_t id_i = (uint32_t)__builtin_amdgcn_readlane(__float_as_int(a.w), r);
            float r1 = 0.f;
            if (VDW) r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ra), r));
            bool found = false;   // WITHIN

            // consume the hits of one 64-chunk, in j order
            auto emit = [&](bool hit, float d2, uint32_t qid) {
                const unsigned long long mask = __builtin_amdgcn_ballot_w64(hit);
                if (WITHIN) {
                    if (mask) found = true;
                    return;
                }
                const uint32_t cnt = (uint32_t)__popcll(mask);
                if (!FILL) {
                    total += cnt;
                    return;
                }
                if (cnt) {
                    if (hit) {
                        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                                        __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                        const uint32_t s = (F.tail + rank) & (FIFO_CAP - 1);
                        F.fi[s] = id_i;
                        F.fj[s] = qid;
                        F.fd[s] = __float_as_uint(d2);
                    }
                    F.tail += cnt;
                    total += cnt;
                    if (F.tail - F.head >= 64u) {
                        __builtin_amdgcn_wave_barrier();
                        fifo_flush<KIND>(F, 64u, lane);
                    }
                }
            };
            // two chunks (c0 = first chunk index) against row i; use0/use1: chunk is live
            auto pair_body = [&](uint32_t c0, bool use0, bool use1, bool rag0, bool rag1, v2f qx, v2f qy, v2f qz, v2f qv,
                                 uint32_t id0, uint32_t id1) {
                const v2f d2 = pair_d2x2<WK>(P, B, T.wrap, px, py, pz, qx, qy, qz);
                bool h0, h1;
                if (VDW) {
                    const v2f cut = (r1 + qv) + F32_EPS;                   // :392, :423
                    const v2f c2 = cut * cut;
                    h0 = d2.x <= c2.x;
                    h1 = d2.y <= c2.y;
                } else {
                    h0 = d2.x <= cutoff2;
                    h1 = d2.y <= cutoff2;
                }
                const uint32_t j0 = c0 * 64u + lane, j1 = j0 + 64u;
                if (rag0) h0 = h0 && (j0 < T.n2);                          // only the last chunk is ragged
                if (rag1) h1 = h1 && (j1 < T.n2);
                if (TRI) {                                                 // j in i+1..n (:443, :482)
                    h0 = h0 && (j0 > i);
                    h1 = h1 && (j1 > i);
                }
                if (use0 && !(WITHIN && found)) emit(h0, d2.x, id0);
                if (use1 && !(WITHIN && found)) emit(h1, d2.y, id1);      // `break` at the first hit (:289, :318)
            };

            if (!STREAM) {
#pragma unroll
                for (int h = 0; h < NPAIR; ++h) {
                    const uint32_t c0 = 2u * (uint32_t)h, c1 = c0 + 1u;
                    bool use0 = true, use1 = (int)c1 < NCH;
                    if (RUNTIME_NCH) {
                        use0 = c0 < nchunks;
                        use1 = c1 < nchunks;
                    }
                    if (TRI) {                                             // whole chunk has j <= i
                        use0 = use0 && !(c0 * 64u + 63u <= i);
                        use1 = use1 && !(c1 * 64u + 63u <= i);
                    }
                    if (WITHIN && found) break;
                    if (!use0 && !use1) continue;
                    const bool rag0 = RUNTIME_NCH ? (c0 + 1u == nchunks) : ((int)c0 == NCH - 1);
                    const bool rag1 = RUNTIME_NCH ? (c1 + 1u == nchunks) : ((int)c1 == NCH - 1);
                    pair_body(c0, use0, use1, rag0, rag1, bx[h], by[h], bz[h], bv[h], bid[2 * h], bid[2 * h + 1]);
                }
            } else {
                for (uint32_t c0 = 0; c0 < nchunks; c0 += 2u) {
                    bool use0 = true, use1 = c0 + 1u < nchunks;
                    if (TRI) {
                        use0 = !(c0 * 64u + 63u <= i);
                        use1 = use1 && !((c0 + 1u) * 64u + 63u <= i);
                    }
                    if (WITHIN && found) break;
                    if (!use0 && !use1) continue;
                    float x0, y0, z0, v0, x1, y1, z1, v1;
                    uint32_t id0, id1;
                    load_b(c0 * 64u + lane, x0, y0, z0, v0, id0);
                    load_b((c0 + 1u) * 64u + lane, x1, y1, z1, v1, id1);
                    pair_body(c0, use0, use1, true, true, v2f{x0, x1}, v2f{y0, y1}, v2f{z0, z1}, v2f{v0, v1}, id0, id1);
                }
            }

            if (WITHIN && found) {
                if (FILL) {
                    if (lane == 0) F.fi[F.tail & (FIFO_CAP - 1)] = id_i;
                    F.tail += 1;
                    if (F.tail - F.head >= 64u) {
                        __builtin_amdgcn_wave_barrier();
                        fifo_flush<KIND>(F, 64u, lane);
                    }
                }
                total += 1;
            }
        }
    }
    if (FILL && F.tail != F.head) {
        __builtin_amdgcn_wave_barrier();
        fifo_flush<KIND>(F, F.tail - F.head, lane);
    }
    return total;
}

// Box matrices copied into VGPRs: an SGPR source operand halves the issue rate of f32 VALU ops on
// gfx950 (4.4 vs 2.4-2.7 cycles per wave instruction, profiles/microbench/valu_rate.hip).
struct BoxV {
    float I[9], M[9];
};

__device__ __forceinline__ float to_vgpr(float s) {
    float v;
    asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(s));
    return v;
}

// PeriodicBox::distance_squared for ONE candidate (periodic_box.rs:286-318, 379-381), plain f32 ops.
// The grid stores every atom inside the primary cell along periodic dimensions (populate_pbc wraps
// them, distance_search.rs:183-196), so |f[d]| < 1.5 for a wrapped dimension and f32::round reduces
// to "copysign(1, f) if |f| >= 0.5 else 0" - the same value, three instructions instead of six.
template <int WK>
__device__ __forceinline__ float wrapped_d2(const SearchParams &P, const BoxV &B, uint32_t wrap, int nshift, float vx,
                                            float vy, float vz) {
    float fx, fy, fz;
    if (WK == WK_DIAG) {
        fx = B.I[0] * vx; fy = B.I[4] * vy; fz = B.I[8] * vz;
    } else if (WK == WK_UPPER) {
        fx = (B.I[0] * vx + B.I[3] * vy) + B.I[6] * vz;
        fy = B.I[4] * vy + B.I[7] * vz;
        fz = B.I[8] * vz;
    } else {
        fx = (B.I[0] * vx + B.I[3] * vy) + B.I[6] * vz;
        fy = (B.I[1] * vx + B.I[4] * vy) + B.I[7] * vz;
        fz = (B.I[2] * vx + B.I[5] * vy) + B.I[8] * vz;
    }
    if (wrap & 1u) fx -= (fabsf(fx) >= 0.5f ? copysignf(1.0f, fx) : 0.0f);
    if (wrap & 2u) fy -= (fabsf(fy) >= 0.5f ? copysignf(1.0f, fy) : 0.0f);
    if (wrap & 4u) fz -= (fabsf(fz) >= 0.5f ? copysignf(1.0f, fz) : 0.0f);
    float sx, sy, sz;
    if (WK == WK_DIAG) {
        sx = B.M[0] * fx; sy = B.M[4] * fy; sz = B.M[8] * fz;
    } else if (WK == WK_UPPER) {
        sx = (B.M[0] * fx + B.M[3] * fy) + B.M[6] * fz;
        sy = B.M[4] * fy + B.M[7] * fz;
        sz = B.M[8] * fz;
    } else {
        sx = (B.M[0] * fx + B.M[3] * fy) + B.M[6] * fz;
        sy = (B.M[1] * fx + B.M[4] * fy) + B.M[7] * fz;
        sz = (B.M[2] * fx + B.M[5] * fy) + B.M[8] * fz;
    }
    float best2 = (sx * sx + sy * sy) + sz * sz;
    if (WK != WK_DIAG && nshift != 0 && wrap == MOLAR_HIP_PBC_FULL) {   // triclinic candidates (:304-317)
        for (int k = 0; k < nshift; ++k) {
            const float cx = sx + P.box.shifts[3 * k], cy = sy + P.box.shifts[3 * k + 1], cz = sz + P.box.shifts[3 * k + 2];
            const float n2 = (cx * cx + cy * cy) + cz * cz;
            best2 = n2 < best2 ? n2 : best2;
        }
    }
    return best2;
}

// squared distance from a point to an axis-aligned box, same f32 expression as a pair distance
__device__ __forceinline__ float aabb_d2(float ax, float ay, float az, float lx, float ly, float lz, float hx, float hy,
                                         float hz) {
    const float ex = fmaxf(fmaxf(lx - ax, ax - hx), 0.f);
    const float ey = fmaxf(fmaxf(ly - ay, ay - hy), 0.f);
    const float ez = fmaxf(fmaxf(lz - az, az - hz), 0.f);
    return (ex * ex + ey * ey) + ez * ez;
}

// Fast path for the bulk of the work: fixed-cutoff, non-triangular cell pairs whose second cell
// fits in registers (NCH <= 8 chunks of 64), plain (WK_NONE) or wrapped (WK_DIAG/UPPER/GENERAL).
//  * the slot's 64 first-cell atoms are staged in LDS and each row is fetched with ONE broadcast
//    ds_read_b128, so the arithmetic runs on VGPR operands only and no v_readlane sits in the row
//    loop;
//  * the count pass never leaves the VALU: hits are added per lane through the carry of the
//    compare and reduced across the wave once per slot (a v_cmp -> s_bcnt1 -> s_add chain costs
//    ~14 cycles per chunk because of the VALU->SALU hazard);
//  * lanes past the end of the second cell hold a coordinate so large that d2 overflows to +inf
//    (or NaN) and the compare fails by itself - no separate validity mask;
//  * rows that provably cannot have a hit are skipped (see `live` below).
template <int KIND, bool FILL, int WK, int NCH, bool TRI>
__device__ __forceinline__ uint32_t run_fast(const SearchParams &P, const Task &T, uint32_t i0, Fifo &F, float4 *la,
                                             uint32_t lane) {
    float bx[NCH], by[NCH], bz[NCH];
    uint32_t bid[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const uint32_t jj = (uint32_t)k * 64u + lane;
        float4 q = make_float4(3.0e38f, 3.0e38f, 3.0e38f, 0.f);
        if (jj < T.n2) q = P.sb[T.b0 + jj];
        bx[k] = q.x; by[k] = q.y; bz[k] = q.z; bid[k] = __float_as_uint(q.w);
    }
    BoxV B;
    int nshift = 0;
    if (WK != WK_NONE) {
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            B.I[k] = to_vgpr(P.box.inv[k]);
            B.M[k] = to_vgpr(P.box.m[k]);
        }
        nshift = P.box.nshift;
    }
    const float cutoff2 = P.cutoff2;
    const uint32_t rows = __builtin_amdgcn_readfirstlane(T.n1 - i0 < T.rps ? T.n1 - i0 : T.rps);
    unsigned long long live;   // rows of this slot that can have a hit at all
    {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane < rows) a = P.sa[T.a0 + i0 + lane];
        la[lane] = a;
        const float4 lo = P.aabb_b[2 * T.cb], hi = P.aabb_b[2 * T.cb + 1];
        bool need = true;
        if (TRI) {
            // same cell (i < j triangle, :439-451): the atom lies inside its own cell's box, nothing to prune
        } else if (WK == WK_NONE) {
            // Exact row pruning.  Every B position lies inside the cell's bounding box [lo,hi], and each
            // f32 operation of d2 = ((dx*dx)+(dy*dy))+(dz*dz) is monotone in |dx|,|dy|,|dz|, so the same
            // expression on the box distances is a lower bound of every d2 of the row IN f32 ARITHMETIC:
            // if it already exceeds cutoff2 the reference finds no hit in this row either.
            need = !(aabb_d2(a.x, a.y, a.z, lo.x, lo.y, lo.z, hi.x, hi.y, hi.z) > cutoff2);
        } else if (P.prune_wrapped && !(nshift != 0 && T.wrap == MOLAR_HIP_PBC_FULL)) {
            // Conservative row pruning for wrapped entries.  In exact arithmetic the reference's vector
            // is (b - a) - sum_d n_d*col_d with n_d = round(f_d) in {-1,0,1} for the wrapped dims (atoms
            // are stored inside the primary cell).  The row is skipped only if for EVERY such image the
            // box distance exceeds cutoff + margin; the margin (1e-3 nm, >100x the f32 evaluation error
            // of inv*v / M*f at MD box sizes) absorbs the difference between the reference's f32
            // arithmetic and this geometric bound.
            const float lim = P.prune_limit2;
            need = false;
            for (int nx = -1; nx <= 1; ++nx) {
                if (!(T.wrap & 1u) && nx != 0) continue;
                for (int ny = -1; ny <= 1; ++ny) {
                    if (!(T.wrap & 2u) && ny != 0) continue;
                    for (int nz = -1; nz <= 1; ++nz) {
                        if (!(T.wrap & 4u) && nz != 0) continue;
                        const float fx = (float)nx, fy = (float)ny, fz = (float)nz;
                        const float tx = (fx * P.box.m[0] + fy * P.box.m[3]) + fz * P.box.m[6];
                        const float ty = (fx * P.box.m[1] + fy * P.box.m[4]) + fz * P.box.m[7];
                        const float tz = (fx * P.box.m[2] + fy * P.box.m[5]) + fz * P.box.m[8];
                        // (b - n*cols) - a  ==  b - (a + n*cols): shift the point instead of the box
                        const float e2 = aabb_d2(a.x + tx, a.y + ty, a.z + tz, lo.x, lo.y, lo.z, hi.x, hi.y, hi.z);
                        need = need || !(e2 > lim);
                    }
                }
            }
        }
        live = __builtin_amdgcn_ballot_w64(lane < rows && need);
    }
    __builtin_amdgcn_wave_barrier();
    uint32_t acc = 0;      // per-lane hit counter (count pass)
    uint32_t total = 0;
    while (live) {
        const uint32_t r = (uint32_t)__builtin_ctzll(live);
        live &= live - 1ull;
        const float4 p = la[r];                      // one broadcast ds_read per row
        const uint32_t id_i = __float_as_uint(p.w);
        const uint32_t i = i0 + r;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            if (TRI && (uint32_t)k * 64u + 63u <= i) continue;                   // whole chunk has j <= i
            const float dx = bx[k] - p.x, dy = by[k] - p.y, dz = bz[k] - p.z;     // p2 - p1
            float d2;
            if (WK == WK_NONE) d2 = (dx * dx + dy * dy) + dz * dz;               // |p2-p1|^2 (:446, :460)
            else d2 = wrapped_d2<WK>(P, B, T.wrap, nshift, dx, dy, dz);          // (:485-486)
            if (TRI && (uint32_t)k * 64u <= i)                                   // diagonal chunk: j in i+1..n (:443)
                d2 = ((uint32_t)k * 64u + lane > i) ? d2 : INFINITY;
            if (!FILL) {
                // acc += (d2 <= cutoff2): the compare's carry is added per lane, no SALU involved
                asm volatile("v_cmp_ge_f32 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc"
                             : "+v"(acc)
                             : "v"(d2), "s"(cutoff2)
                             : "vcc");
            } else {
                const bool hit = d2 <= cutoff2;
                const unsigned long long mask = __builtin_amdgcn_ballot_w64(hit);
                if (mask) {
                    const uint32_t cnt = (uint32_t)__popcll(mask);
                    if (hit) {
                        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                                        __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                        const uint32_t s = (F.tail + rank) & (FIFO_CAP - 1);
                        F.fi[s] = id_i;
                        F.fj[s] = bid[k];
                        F.fd[s] = __float_as_uint(d2);
                    }
                    F.tail += cnt;
                    total += cnt;
                    if (F.tail - F.head >= 64u) {
                        __builtin_amdgcn_wave_barrier();
                        fifo_flush<KIND>(F, 64u, lane);
                    }
                }
            }
        }
    }
    if (!FILL) {
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        total = acc;
    } else if (F.tail != F.head) {
        __builtin_amdgcn_wave_barrier();
        fifo_flush<KIND>(F, F.tail - F.head, lane);
    }
    return total;
}

// chunk-count dispatch: for the single-set search the non-triangular tasks get a fully unrolled,
// branch-free row body per chunk count; everything else checks the chunk count at run time
template <int KIND, bool FILL, int WK>
__device__ __forceinline__ uint32_t run_task_nch(const SearchParams &P, const Task &T, uint32_t i0, Fifo &F, float4 *la,
                                                 uint32_t lane) {
    const uint32_t nchunks = (T.n2 + 63u) >> 6;
    if ((KIND == MOLAR_HIP_SEARCH_SINGLE || KIND == MOLAR_HIP_SEARCH_DOUBLE) && !T.tri && nchunks <= (uint32_t)KREG) {
        switch (nchunks) {
            case 1: return run_fast<KIND, FILL, WK, 1, false>(P, T, i0, F, la, lane);
            case 2: return run_fast<KIND, FILL, WK, 2, false>(P, T, i0, F, la, lane);
            case 3: return run_fast<KIND, FILL, WK, 3, false>(P, T, i0, F, la, lane);
            case 4: return run_fast<KIND, FILL, WK, 4, false>(P, T, i0, F, la, lane);
            case 5: return run_fast<KIND, FILL, WK, 5, false>(P, T, i0, F, la, lane);
            case 6: return run_fast<KIND, FILL, WK, 6, false>(P, T, i0, F, la, lane);
            case 7: return run_fast<KIND, FILL, WK, 7, false>(P, T, i0, F, la, lane);
            default: return run_fast<KIND, FILL, WK, 8, false>(P, T, i0, F, la, lane);
        }
    }
    if (KIND == MOLAR_HIP_SEARCH_SINGLE && WK == WK_NONE && T.tri && nchunks <= (uint32_t)KREG) {
        switch (nchunks) {
            case 1: return run_fast<KIND, FILL, WK_NONE, 1, true>(P, T, i0, F, la, lane);
            case 2: return run_fast<KIND, FILL, WK_NONE, 2, true>(P, T, i0, F, la, lane);
            case 3: return run_fast<KIND, FILL, WK_NONE, 3, true>(P, T, i0, F, la, lane);
            case 4: return run_fast<KIND, FILL, WK_NONE, 4, true>(P, T, i0, F, la, lane);
            case 5: return run_fast<KIND, FILL, WK_NONE, 5, true>(P, T, i0, F, la, lane);
            case 6: return run_fast<KIND, FILL, WK_NONE, 6, true>(P, T, i0, F, la, lane);
            case 7: return run_fast<KIND, FILL, WK_NONE, 7, true>(P, T, i0, F, la, lane);
            default: return run_fast<KIND, FILL, WK_NONE, 8, true>(P, T, i0, F, la, lane);
        }
    }
    if (nchunks > (uint32_t)KREG) {
        if (KIND == MOLAR_HIP_SEARCH_SINGLE && T.tri) return run_task<KIND, FILL, WK, true, 0, false>(P, T, i0, F, lane);
        return run_task<KIND, FILL, WK, false, 0, false>(P, T, i0, F, lane);
    }
    if (KIND == MOLAR_HIP_SEARCH_SINGLE && T.tri) return run_task<KIND, FILL, WK, true, KREG, true>(P, T, i0, F, lane);
    return run_task<KIND, FILL, WK, false, KREG, true>(P, T, i0, F, lane);
}

// Slots.  A plan entry ("task") is cut into blocks of 64 rows of its first cell; one wave processes
// one slot.  This bounds the work of a wave (the corner entries that run the triclinic candidate
// loop are ~50x a plain entry) and gives small systems enough waves to fill the chip.  Slots are
// numbered in plan order, then row order, so an exclusive scan of the per-slot counts is the
// reference's output order.
template <int KIND>
__global__ void __launch_bounds__(256) plan_kernel(SearchParams P, uint32_t *__restrict__ task_nb) {
    const uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (t >= P.ntasks) return;
    const Task T = decode_task<KIND, false>(P, t);
    task_nb[t] = T.valid ? (T.n1 + T.rps - 1u) / T.rps : 0u;
}

__global__ void __launch_bounds__(256) slotmap_kernel(uint64_t ntasks, const uint32_t *__restrict__ task_first,
                                                      uint32_t *__restrict__ slot_task) {
    const uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (t >= ntasks) return;
    const uint32_t s0 = task_first[t], s1 = task_first[t + 1];
    for (uint32_t s = s0; s < s1; ++s) slot_task[s] = (uint32_t)t;
}

template <int KIND, bool FILL>
__global__ void __launch_bounds__(BLOCK) pair_kernel(const SearchParams *__restrict__ Pp,
                                                     const uint32_t *__restrict__ task_first,
                                                     const uint32_t *__restrict__ slot_task,
                                                     uint32_t *__restrict__ slot_cnt,
                                                     const unsigned long long *__restrict__ slot_base,
                                                     uint2 *__restrict__ out_pairs, float *__restrict__ out_dist,
                                                     uint32_t *__restrict__ out_ids) {
    __shared__ uint32_t lds[WAVES_PER_BLOCK][3][FIFO_CAP];
    __shared__ float4 lds_a[WAVES_PER_BLOCK][64];
    // The parameter block lives in device memory: a by-value struct this large, indexed dynamically
    // (box.shifts[k]), gets copied to scratch by the compiler and drags every field into VGPRs.
    const SearchParams &P = *Pp;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t nslots = task_first[P.ntasks];
    const uint32_t w = blockIdx.x * WAVES_PER_BLOCK + wave;
    if (w >= nslots) return;
    // Blocks are handed out in launch order: walk the plan BACKWARDS so the cells at the far x edge,
    // whose entries wrap (several times the arithmetic per candidate), start first and the cheap
    // entries fill the tail; consecutive blocks land on different XCDs, which spreads that band
    // over the whole chip.
    const uint32_t slot = nslots - 1u - w;
    const uint32_t t = slot_task[slot];
    const Task T = decode_task<KIND, true>(P, t);
    const uint32_t i0 = (slot - task_first[t]) * T.rps;
    Fifo F;
    F.fi = lds[wave][0];
    F.fj = lds[wave][1];
    F.fd = lds[wave][2];
    F.head = F.tail = 0;
    F.pairs = out_pairs;
    F.dist = out_dist;
    F.ids = out_ids;
    F.base = 0;
    if (FILL) {
        F.base = slot_base[slot];
        if (slot_base[slot + 1] == F.base) return;     // nothing to emit: skip the traversal
    }
    uint32_t total = 0;
    const uint32_t wk = (P.use_box && T.wrap != 0) ? P.wrap_kind : (uint32_t)WK_NONE;
    if (P.debug_skip) {
        const uint32_t kind_bit = T.tri ? 4u : (wk != WK_NONE ? 2u : 1u);
        if (P.debug_skip & kind_bit) return;
    }
    switch (wk) {
        case WK_NONE: total = run_task_nch<KIND, FILL, WK_NONE>(P, T, i0, F, lds_a[wave], lane); break;
        case WK_DIAG: total = run_task_nch<KIND, FILL, WK_DIAG>(P, T, i0, F, lds_a[wave], lane); break;
        case WK_UPPER: total = run_task_nch<KIND, FILL, WK_UPPER>(P, T, i0, F, lds_a[wave], lane); break;
        default: total = run_task_nch<KIND, FILL, WK_GENERAL>(P, T, i0, F, lds_a[wave], lane); break;
    }
    if (!FILL && lane == 0) slot_cnt[slot] = total;
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

int build_grid(molar_hip_ctx *c, GridSet &S, int ids_local) {
    Prof prof(c, 0);
    const uint32_t ncells = c->dims[0] * c->dims[1] * c->dims[2];
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
    MH_TRY(S.key.reserve((size_t)(S.n ? S.n : 1) * 4));
    MH_TRY(S.cell_count.reserve((size_t)(ncells + 1) * 4));
    MH_TRY(S.cursor.reserve((size_t)(S.n ? S.n : 1) * 4));   // arrival order of each atom in its cell
    MH_TRY(S.tmp_key.reserve((size_t)(S.n ? S.n : 1) * 4));
    MH_TRY(S.tmp_cell.reserve((size_t)(S.n ? S.n : 1) * 4));
    MH_TRY(S.sorted.reserve((size_t)(S.n ? S.n : 1) * sizeof(float4)));
    if (S.d_vdw) MH_TRY(S.sorted_vdw.reserve((size_t)(S.n ? S.n : 1) * 4));
    MH_TRY(S.aabb.reserve((size_t)ncells * 2 * sizeof(float4)));
    MH_HIP(hipMemsetAsync(S.cell_count.p, 0, (size_t)(ncells + 1) * 4, c->stream));
    if (S.n) {
        const unsigned nb = (S.n + 255u) / 256u;
        hipLaunchKernelGGL(bin_kernel, dim3(nb), dim3(256), 0, c->stream, P, S.key.as<uint32_t>(),
                           S.cursor.as<uint32_t>(), S.cell_count.as<uint32_t>());
        MH_TRY((exclusive_scan<uint32_t, uint32_t>(c, S.cell_count.as<uint32_t>(), S.cell_count.as<uint32_t>(),
                                                   (uint64_t)ncells + 1)));
        hipLaunchKernelGGL(scatter_kernel, dim3(nb), dim3(256), 0, c->stream, S.n, S.key.as<uint32_t>(),
                           S.cell_count.as<uint32_t>(), S.cursor.as<uint32_t>(), S.tmp_key.as<uint32_t>(),
                           S.tmp_cell.as<uint32_t>());
        hipLaunchKernelGGL(place_kernel, dim3(nb), dim3(256), 0, c->stream, P, ncells, ids_local,
                           S.cell_count.as<uint32_t>(), S.tmp_key.as<uint32_t>(), S.tmp_cell.as<uint32_t>(), S.d_vdw,
                           S.sorted.as<float4>(), S.d_vdw ? S.sorted_vdw.as<float>() : nullptr);
        hipLaunchKernelGGL(cell_aabb_kernel, dim3((ncells + 3u) / 4u), dim3(256), 0, c->stream, ncells,
                           S.cell_count.as<uint32_t>(), S.sorted.as<float4>(), S.aabb.as<float4>());
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
    P.dx = c->dims[0];
    P.dy = c->dims[1];
    P.dz = c->dims[2];
    P.pbc = c->use_box ? c->pbc : 0u;
    P.use_box = c->use_box ? 1u : 0u;
    P.cutoff2 = c->cutoff * c->cutoff;
    P.ntasks = c->ntasks;
    P.nblocks = (uint32_t)((c->nslots_bound + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK);
    P.box = c->box;
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
    const float lim = c->cutoff + 1.0e-3f;
    P.prune_limit2 = lim * lim;
    const char *dbg = std::getenv("MOLAR_HIP_DEBUG_SKIP");
    P.debug_skip = dbg ? (uint32_t)std::atoi(dbg) : 0u;
    return P;
}

template <bool FILL>
int launch_pairs(molar_hip_ctx *c, uint2 *pairs, float *dist, uint32_t *ids) {
    Prof prof(c, FILL ? 3 : 1);
    const SearchParams P = make_params(c);
    if (P.nblocks == 0) return 0;
    MH_TRY(c->params.reserve(sizeof(SearchParams)));
    MH_HIP(hipMemcpyAsync(c->params.p, &P, sizeof(SearchParams), hipMemcpyHostToDevice, c->stream));
    const SearchParams *dP = c->params.as<SearchParams>();
    const uint32_t *tf = c->task_nb.as<uint32_t>();
    const uint32_t *st = c->slot_task.as<uint32_t>();
    uint32_t *sc = c->slot_cnt.as<uint32_t>();
    auto *sb = c->slot_base.as<unsigned long long>();
    const dim3 g(P.nblocks), b(BLOCK);
    switch (c->kind) {
        case MOLAR_HIP_SEARCH_SINGLE:
            hipLaunchKernelGGL((pair_kernel<MOLAR_HIP_SEARCH_SINGLE, FILL>), g, b, 0, c->stream, dP, tf, st, sc, sb, pairs, dist, ids);
            break;
        case MOLAR_HIP_SEARCH_DOUBLE:
            hipLaunchKernelGGL((pair_kernel<MOLAR_HIP_SEARCH_DOUBLE, FILL>), g, b, 0, c->stream, dP, tf, st, sc, sb, pairs, dist, ids);
            break;
        case MOLAR_HIP_SEARCH_WITHIN:
            hipLaunchKernelGGL((pair_kernel<MOLAR_HIP_SEARCH_WITHIN, FILL>), g, b, 0, c->stream, dP, tf, st, sc, sb, pairs, dist, ids);
            break;
        default:
            hipLaunchKernelGGL((pair_kernel<MOLAR_HIP_SEARCH_DOUBLE_VDW, FILL>), g, b, 0, c->stream, dP, tf, st, sc, sb, pairs, dist, ids);
            break;
    }
    MH_HIP(hipGetLastError());
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

int device_fmax(molar_hip_ctx *c, const float *d_v, uint32_t n, float *out) {
    MH_TRY(c->hist.reserve(64));
    uint32_t *d = c->hist.as<uint32_t>();
    const uint32_t seed = 0u;   // below every ordered float
    MH_HIP(hipMemcpyAsync(d, &seed, 4, hipMemcpyHostToDevice, c->stream));
    unsigned nb = (n + 255u) / 256u;
    if (nb > 1024u) nb = 1024u;
    hipLaunchKernelGGL(fmax_kernel, dim3(nb), dim3(256), 0, c->stream, d_v, n, d);
    MH_HIP(hipGetLastError());
    uint32_t o;
    MH_TRY(read_back(c, &o, d, 4));
    *out = ord2f(o);
    return 0;
}

int prepare_search(molar_hip_ctx *c, const molar_hip_search_desc *q) {
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
    MH_TRY(stage_set(c, c->set[0], q->xyz1, q->natoms1, q->idx1, q->n1, q->vdw1, vdw));
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
        MH_TRY(device_fmax(c, c->set[0].d_vdw, c->set[0].n, &m1));
        MH_TRY(device_fmax(c, c->set[1].d_vdw, c->set[1].n, &m2));
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
    MH_TRY(build_grid(c, c->set[0], q->ids_local || vdw));
    if (two) MH_TRY(build_grid(c, c->set[1], q->ids_local || vdw));

    const uint64_t ncells = (uint64_t)c->dims[0] * c->dims[1] * c->dims[2];
    c->ntasks = ncells * 14ull * (two ? 2ull : 1ull);
    // every set-1 cell is the first cell of at most 14 (two grids: 28) tasks, so
    // sum_t ceil(n1(t)/64) <= mult*N1/64 + ntasks
    c->nslots_bound = (two ? 28ull : 14ull) * (((uint64_t)c->set[0].n + 63ull) / 64ull) + c->ntasks;
    // entries wrapping in all three dims of a triclinic box use 8-row slots: <= 28 tasks of <= 4096 rows
    c->nslots_bound += 28ull * 512ull;
    if (c->ntasks >= 0xFFFFFFF0ull || c->nslots_bound >= 0xFFFFFFF0ull)
        return fail(MOLAR_HIP_ERR_TOO_LARGE, "search plan too large (%llu entries)", (unsigned long long)c->ntasks);
    MH_TRY(c->task_nb.reserve((c->ntasks + 1) * 4));
    MH_TRY(c->slot_task.reserve((c->nslots_bound + 1) * 4));
    MH_TRY(c->slot_cnt.reserve((c->nslots_bound + 1) * 4));
    MH_TRY(c->slot_base.reserve((c->nslots_bound + 1) * 8));
    {
        Prof prof(c, 0);
        MH_HIP(hipMemsetAsync(c->task_nb.p, 0, (c->ntasks + 1) * 4, c->stream));
        MH_HIP(hipMemsetAsync(c->slot_cnt.p, 0, (c->nslots_bound + 1) * 4, c->stream));
        const SearchParams P = make_params(c);
        const unsigned nb = (unsigned)((c->ntasks + 255) / 256);
        switch (c->kind) {
            case MOLAR_HIP_SEARCH_SINGLE:
                hipLaunchKernelGGL((plan_kernel<MOLAR_HIP_SEARCH_SINGLE>), dim3(nb), dim3(256), 0, c->stream, P, c->task_nb.as<uint32_t>());
                break;
            default:   // the three two-grid kinds decode tasks identically
                hipLaunchKernelGGL((plan_kernel<MOLAR_HIP_SEARCH_DOUBLE>), dim3(nb), dim3(256), 0, c->stream, P, c->task_nb.as<uint32_t>());
                break;
        }
        MH_TRY((exclusive_scan<uint32_t, uint32_t>(c, c->task_nb.as<uint32_t>(), c->task_nb.as<uint32_t>(), c->ntasks + 1)));
        hipLaunchKernelGGL(slotmap_kernel, dim3(nb), dim3(256), 0, c->stream, c->ntasks, c->task_nb.as<uint32_t>(),
                           c->slot_task.as<uint32_t>());
        MH_HIP(hipGetLastError());
    }
    return 0;
}

int finish_count(molar_hip_ctx *c) {
    Prof *prof = new Prof(c, 2);
    int rc = (exclusive_scan<uint32_t, unsigned long long>(c, c->slot_cnt.as<uint32_t>(),
                                                           c->slot_base.as<unsigned long long>(), c->nslots_bound + 1));
    delete prof;
    MH_TRY(rc);
    unsigned long long tot = 0;
    MH_TRY(read_back(c, &tot, c->slot_base.as<unsigned long long>() + c->nslots_bound, 8));
    c->total = tot;
    c->have_search = true;
    return 0;
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

static int fill_common(molar_hip_ctx *c, uint2 *d_pairs, float *d_dist, uint32_t *d_ids) {
    if (!c || !c->have_search) return fail(MOLAR_HIP_ERR_NO_SEARCH, "no cached search: call molar_hip_search_count first");
    MH_HIP(hipSetDevice(c->device));
    if (c->total == 0 || c->ntasks == 0) return 0;
    return launch_pairs<true>(c, d_pairs, d_dist, d_ids);
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
    if (pairs && !pd && c->total) MH_HIP(hipMemcpyAsync(pairs, dp, c->total * 8, hipMemcpyDeviceToHost, c->stream));
    if (dist && !dd && c->total) MH_HIP(hipMemcpyAsync(dist, ddist, c->total * 4, hipMemcpyDeviceToHost, c->stream));
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

int molar_hip_search_histogram(molar_hip_ctx *c, const molar_hip_search_desc *q, float hmin, float hmax, size_t nbins,
                               uint64_t *bins, uint64_t *out_count) {
    (void)c; (void)q; (void)hmin; (void)hmax; (void)nbins; (void)bins; (void)out_count;
    return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "molar_hip_search_histogram: not implemented yet");
}

}  // extern "C"
