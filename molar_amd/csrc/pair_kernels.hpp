// pair_kernels.hpp - the cell-pair kernels of the distance search (device code) and their launcher.
// Included by search.hip (parameter structs, plan kernels) and by one translation unit per search
// kind (pair_k0..3.hip), so the heavy template instantiations compile in parallel.
#pragma once

#include "boxmath.hpp"
#include "common.hpp"

namespace mh {
namespace pairk {

enum { MODE_COUNT = 0, MODE_FILL = 1, MODE_HIST = 2 };

// (The fill kernel is compiled for 8 waves per SIMD - 64 VGPRs, the few spills fall outside the row loop: -5 % against
// 7 waves; 32 one-wave workgroups hold 144 KB of the CU's 160 KB LDS; re-measured in round 6: 7 / 6 waves lose 1 / 4 % of the
// headline.  The count kernel keeps a cell's B records in registers for the matrix cores (run_count_mfma): 8 waves (64
// registers) 0.41 ms at 0.6 nm, 7 waves (72 registers, 70 spilled, 95 scalars spilled) 0.30, SIX waves (80 registers, 4 and 46
// spilled) 0.29 - and 0.385 -> 0.365 ms on the headline frame, 5 waves 0.32: since round 6 it runs six.)
// Waves per workgroup.  Count / fill: ONE wave per workgroup - slots differ a lot in work, and a workgroup's
// resources are only released when its slowest wave ends (measured: 4 -> 1 waves gives +7 % frames/s).  The fused
// histogram keeps 4: every workgroup owns an LDS histogram that it flushes with atomics at the end.
// (round 4, count pass with 2 / 4 waves per workgroup - half / a quarter of the 2.9e5 workgroups, whose bare launch takes
// 69 us: 0.42 / 0.45 ms against 0.40, three alternations on one box)
#ifndef MH_COUNT_WPE
#define MH_COUNT_WPE 6
#endif
#ifndef MH_FILL_WPE
#define MH_FILL_WPE 8
#endif
#ifndef MH_WPB
#define MH_WPB 1
#endif
constexpr int waves_per_block(int mode) { return mode == MODE_HIST ? 4 : MH_WPB; }
constexpr int KREG = 8;            // B-cell chunks (of 64 atoms) a lane keeps in registers
constexpr int FIFO_CAP = 128;      // per-wave LDS FIFO entries (flush threshold 64, push <= 64)
#ifndef MOLAR_HIP_NO_WIDE_FLUSH
#define MOLAR_HIP_WIDE_FLUSH 1
#endif
constexpr int FIFO_WIDE = 256;     // ... of the plain entries' queue in the fill pass: flushed 128 at a time (fifo_flush_wide)
constexpr uint32_t XCD_RUN = 128;  // consecutive slots an XCD takes at a time (pair_kernel)
constexpr float F32_EPS = 1.1920929e-07f;
constexpr bool MASKED_COUNT_SORTED = true;   // count pass of plain / same-cell entries walks the spatial order

// distance_search.rs:39-60
static __constant__ uint8_t MASKS[14][6] = {
    {0, 0, 0, 0, 0, 0},
    {0, 0, 0, 1, 0, 0}, {0, 0, 0, 0, 1, 0}, {0, 0, 0, 0, 0, 1},
    {0, 0, 0, 1, 1, 0}, {0, 0, 0, 1, 0, 1}, {0, 0, 0, 0, 1, 1},
    {0, 0, 0, 1, 1, 1},
    {1, 0, 0, 0, 1, 0}, {1, 0, 0, 0, 0, 1}, {0, 1, 0, 0, 0, 1},
    {1, 1, 0, 0, 0, 1}, {1, 0, 1, 0, 1, 0}, {0, 1, 1, 1, 0, 0},
};

// ================================================================= pair kernels

// Pointers read out of the parameter block in memory have no address space the compiler can see, so plain
// dereferences become FLAT accesses (aperture check, both wait counters).  These helpers state the space:
// global_load with an SGPR base, ds_read for LDS.
typedef float v4f_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 gload4(const float4 *p, size_t i) {
    const v4f_t v = ((const __attribute__((address_space(1))) v4f_t *)p)[i];
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint32_t gload_u32(const uint32_t *p, size_t i) {
    return ((const __attribute__((address_space(1))) uint32_t *)p)[i];
}
__device__ __forceinline__ void gstore_u32(uint32_t *p, size_t i, uint32_t v) {
    ((__attribute__((address_space(1))) uint32_t *)p)[i] = v;
}
// a wave-uniform float pinned to an SGPR (see HistFifo: uniform values that live long are not left in VGPRs)
__device__ __forceinline__ float uniform_f32(float v) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); }
__device__ __forceinline__ float4 lload4(const float4 *p, uint32_t i) {
    const v4f_t v = ((const __attribute__((address_space(3))) v4f_t *)p)[i];
    return make_float4(v.x, v.y, v.z, v.w);
}

struct SearchParams {
    const float4 *sa;        // cell-sorted atoms of set 1
    const float4 *sb;        // cell-sorted atoms of set 2 (== sa for SINGLE)
    const uint32_t *csa;     // cell_start of set 1
    const uint32_t *csb;     // cell_start of set 2
    const float *vdwa;
    const float *vdwb;
    const float4 *aabb_b;    // per-cell bounding boxes of set 2 (== set 1 for SINGLE)
    const float4 *perm_b;    // set 2: atoms in Morton order inside each cell, {x,y,z,position in the cell} (cells of <= 512 atoms), see place_order_kernel
    const float4 *chunk_aabb_b;   // set 2: bounding boxes of the 64-atom Morton chunks, slot (cell_start >> 6) + cell + k
    const uint4 *h16_b;      // set 2, in the order of sb: 8 x f16 {hi xyz, lo xyz, |.|^2 hi, lo} relative to the cell origin
    const float4 *cell_org_b; // set 2: per cell {origin, bound on |position - origin|}
    uint32_t mfma_count;     // count pass on the matrix cores: bit 0 plain / same-cell entries (run_count_mfma), bit 1 wrapped entries (run_count_mfma_wrapped)
    const struct TaskDesc *task_desc;   // per plan entry, written by plan_kernel
    uint32_t *maskbuf;       // fast-path slots: hit bits found by the count pass, replayed by the fill pass
    const unsigned long long *task_moff;   // per task: first 64-word unit of its slots in maskbuf (a slot owns 2*nch units)
    unsigned long long mask_cap_units;     // units maskbuf can hold; a slot beyond it records / replays nothing (the host
                                           // grows the buffer and repeats the pass - single-sync resident path)
    unsigned long long out_cap;            // result entries the output buffers can hold; slots beyond it write nothing
    uint32_t dx, dy, dz;
    uint32_t pbc;            // PbcDims of the plan (0 for the non-periodic drivers)
    uint32_t use_box;
    uint32_t nblocks;        // launch grid
    uint32_t wrap_kind;      // WK_* of the box matrix (zero pattern of m and inv)
    uint32_t prune_wrapped;  // wrapped entries may use the image-box row pruning (all periodic dims >= 3 cells)
    float prune_limit2;      // (cutoff + margin)^2 for that pruning
    uint32_t approx_wrapped; // wrapped entries may be classified with the plain distance to the image cell
    float band_lo, band_hi;  // band around cutoff^2 inside which the exact formula decides
    uint32_t hist_nbins;     // != 0: the fill traversal feeds a histogram instead of writing pairs
    float hist_min, hist_max;
    unsigned long long *hist_bins;    // [nbins]
    unsigned long long *hist_total;   // number of distances binned or out of range (NULL: not wanted)
    const float *hist_edges; // f32[nbins + 1], see histogram_edges() in search.hip; NULL: hist_kernel evaluates the formula per hit
    float hist_scale;        // nbins / (max - min), for the first guess of the bin
    uint32_t hist_lean;      // histogram mode: hist_kernel takes the slots hist_lean_slot() accepts, pair_kernel<MODE_HIST> the rest
    const uint32_t *hist_nslots;   // histogram mode with the one-kernel plan (hist_plan_kernel): pair_kernel<MODE_HIST> walks a list of
                                   // exactly this many records, all of them its own (NULL: the slot records of the regular plan)
#ifdef MOLAR_HIP_DEBUG_KNOBS
    uint32_t debug_skip;     // profiling aid (env MOLAR_HIP_DEBUG_SKIP): bit0 plain, bit1 wrapped, bit2 triangular slots do nothing
    unsigned long long *dbg; // per-wave time accounting of hist_kernel (8 words per wave), molar_hip_debug_fetch
#endif
    float cutoff2;
    uint64_t ntasks;
    molar_hip_box box;
    // box.shifts padded with zero vectors to 32 entries: the candidate loops of the triclinic corner entries read FOUR lattice
    // shifts per trip (three 16-byte scalar loads, one wait) instead of one dependent scalar load per image; a zero shift gives
    // the start vector itself, which never beats the running minimum (periodic_box.rs:304-317)
    alignas(16) float shifts4[96];
    uint32_t hist_big;       // histogram mode: cells of more than KREG * 64 atoms are the rule (mean population above 384): the lean kernel's BIG
                             // instance (at the end: the fields in front of it keep their places)
};

// what plan_kernel stores per plan entry and the pair kernels read back with one 32-byte load
struct TaskDesc {
    uint32_t a0, n1, b0, n2, cb, flags, pad0, pad1;   // flags: wrap | tri<<8 | valid<<9 | wrap_b<<12 | rps<<16
};

// what slotmap_kernel stores per slot: everything a wave needs to start on its slot, so that the chain of dependent
// loads in front of the first atom is kernel arguments -> this record -> atoms (it used to be parameter block ->
// slot count -> slot's task -> task descriptor -> atoms).  Slots in [number of slots, host-side bound] have flags == 0.
struct SlotDesc {
    uint32_t a0, n1, b0, n2;          // as TaskDesc
    uint32_t cb, flags, i0, frame;    // i0: first row of the slot inside the first cell; frame: which parameter block of the launch the slot
                                      // belongs to (fused histogram over several frames, molar_hip_search_histogram_frames; 0 everywhere else)
    unsigned long long moff, pad1;    // first 64-word unit of the slot in maskbuf
};
static_assert(sizeof(SlotDesc) == 48, "SlotDesc is read with three 16-byte loads");

struct Task {
    uint32_t a0, n1, b0, n2;
    uint32_t rps;             // rows of the first cell per slot: 64, or 8 for entries that run the
                              // triclinic candidate loop (~50x the arithmetic per candidate)
    uint32_t cb;              // second cell (for its bounding box)
    uint32_t ca;              // first cell (decode_task only; the slot records do not carry it)
    uint32_t wrap_b;          // dims in which the SECOND cell is the one that wrapped (subset of wrap)
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
    T.ca = 0;
    T.wrap_b = 0;
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
    uint32_t wrap = 0, wrap_c2 = 0;
    for (int i = 0; i < 2; ++i)
        for (int d = 0; d < 3; ++d)
            if (c[i][d] == dims[d]) {
                if ((P.pbc >> d) & 1u) {
                    c[i][d] = 0;
                    wrap |= 1u << d;
                    if (i == 1) wrap_c2 |= 1u << d;
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
    // second cell of the task: c2, or c1 for the swapped half of a two-grid entry
    T.wrap_b = (KIND != MOLAR_HIP_SEARCH_SINGLE && half) ? (wrap & ~wrap_c2) : wrap_c2;
    // only the entries of the single home cell (dx-1,dy-1,dz-1) can wrap in all three dims: <= 28 tasks
    // (their candidate loop reads the lattice shifts from memory, one dependent scalar load per candidate image: a slot
    // is latency-bound, so these few tasks are cut into many short slots - 2 rows, or 8 for very large cells)
    T.rps = (P.use_box && wrap == MOLAR_HIP_PBC_FULL && P.box.nshift != 0 && T.n1 <= 4096u) ? (T.n1 <= 1024u ? 2u : 8u) : 64u;
    T.cb = cb;
    T.ca = ca;
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

template <int WK, bool WIDE4 = false>
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
        // (WIDE4: four lattice shifts per trip - three 16-byte scalar loads and one wait instead of a dependent scalar load per image.
        // Takes the histogram's generic kernel from 27 to 17 us on the C4 frame (hist total 0.288 -> 0.280 ms); in the count pass the twelve extra scalar registers
        // cost more than the loads save (0.40 -> 0.43 ms on the headline frame), so count and fill keep one image per trip.)
        if (WIDE4) {
            for (int k = 0; k < B.nshift; k += 4) {
                float sh[12];
#pragma unroll
                for (int q = 0; q < 12; ++q) sh[q] = P.shifts4[3 * k + q];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const v2f cx = sx + sh[3 * q], cy = sy + sh[3 * q + 1], cz = sz + sh[3 * q + 2];
                    const v2f n2 = (cx * cx + cy * cy) + cz * cz;
                    best2.x = n2.x < best2.x ? n2.x : best2.x;
                    best2.y = n2.y < best2.y ? n2.y : best2.y;
                }
            }
        } else {
            for (int k = 0; k < B.nshift; ++k) {
                const v2f cx = sx + P.box.shifts[3 * k], cy = sy + P.box.shifts[3 * k + 1], cz = sz + P.box.shifts[3 * k + 2];
                const v2f n2 = (cx * cx + cy * cy) + cz * cz;
                // `best` itself is only needed through its norm: cand = start + s is always formed from
                // `start`, not from the running best (:310), so tracking best2 is enough
                best2.x = n2.x < best2.x ? n2.x : best2.x;
                best2.y = n2.y < best2.y ? n2.y : best2.y;
            }
        }
    }
    return best2;
}

// Per-wave output FIFO of the fill pass.
struct Fifo {
    uint32_t *fi, *fj, *fd;     // LDS, FIFO_CAP entries each
    uint32_t head, tail;        // monotonically increasing, wave-uniform
    uint32_t quota;             // entries the next full flush writes: 64 - (base % 64) for the first flush of a slot, so
                                // that every later flush is one naturally aligned 512-byte / 256-byte block; then 64
    uint64_t base;              // output offset of this slot
    uint32_t room;              // entries this slot owns in the output (what the count pass found): a flush never writes past
                                // them, whatever the two passes may disagree on - the neighbours' results and the buffer's end
                                // are out of reach of a slot
    bool has_pairs, has_dist;   // which of the two planes the caller wants (wave-uniform)
    uint2 *pairs;
    float *dist;
    uint32_t *ids;              // WITHIN output
    uint32_t *hist;             // consumer-fused mode: workgroup histogram in LDS (NULL otherwise)
    float hmin, hmax, hn;
    uint32_t recompute;         // 0: entries carry ids + d2;  1 / 2: entries are (row<<26 | sorted position) only and ids,
                                // d2 are rebuilt at flush time with the exact wrapped (1) or the plain (2) formula
    const float4 *la;           // the slot's first-cell atoms in LDS
    float4 *fq_store;           // LDS backing of fq (fill kernels)
    float4 *fq;                 // replay mode: FIFO of the hits' second atoms {x,y,z,id}; fd then holds the row
    uint4 *lh;                  // count pass: LDS staging of the slot's matrix-core row records (128 x 16 B), or NULL
    uint32_t wrap;
};

template <bool WIDE4 = false>
__device__ __forceinline__ float wrapped_d2_exact(const SearchParams &P, uint32_t wrap, float vx, float vy, float vz);

// One queued hit, resolved at flush time: ids and squared distance
struct Hit {
    uint32_t i, j;
    float d2;
};
__device__ __forceinline__ Hit fifo_hit(const SearchParams &P, const Fifo &F, uint32_t s) {
    const uint32_t w = F.fd[s];
    if (F.recompute == 0u) return Hit{F.fi[s], F.fj[s], __uint_as_float(w)};
    float4 a, b;
    if (F.fq) {                 // replay: two LDS reads
        a = lload4(F.la, w);
        b = lload4(F.fq, s);
    } else {
        a = lload4(F.la, w >> 26);
        b = gload4(P.sb, w & 0x3FFFFFFu);
    }
    const float dx = b.x - a.x, dy = b.y - a.y, dz = b.z - a.z;               // p2 - p1
    const float d2 = F.recompute == 1u ? wrapped_d2_exact(P, F.wrap, dx, dy, dz) : (dx * dx + dy * dy) + dz * dz;
    return Hit{__float_as_uint(a.w), __float_as_uint(b.w), d2};
}

template <int KIND, int CAP = FIFO_CAP>
__device__ __forceinline__ void fifo_flush(const SearchParams &P, Fifo &F, uint32_t count, uint32_t lane) {
    if (lane < count && (F.hist || F.head + lane < F.room)) {
        const uint32_t s = (F.head + lane) & (CAP - 1);
        // F.pairs / F.dist / F.ids are per-lane pointers to entry (slot base + lane): the flush adds the FIFO head, and
        // the kernel's output pointers need not stay in (spilled) SGPRs
        uint32_t pos = F.head;
#ifdef MOLAR_HIP_DEBUG_KNOBS
        if (P.debug_skip & 32u) pos &= 63u;
#endif
        if (KIND == MOLAR_HIP_SEARCH_WITHIN) {
            __builtin_nontemporal_store(F.fi[s], &F.ids[pos]);
        } else if (F.hist) {
            // Histogram1D::add_one (molar_membrane/src/stats.rs:29-35) on d = sqrt(d2):
            //   b = (n as Float * (val - min) / (max - min)).floor() as isize;  if 0 <= b < n: bins[b] += 1
            const float d = __builtin_sqrtf(fifo_hit(P, F, s).d2);
            float fb = __builtin_floorf(F.hn * (d - F.hmin) / (F.hmax - F.hmin));
            if (fb != fb) fb = 0.0f;                                    // NaN as isize == 0
            if (fb >= 0.0f && fb < F.hn) atomicAdd(&F.hist[(uint32_t)fb], 1u);
        } else {
            const Hit h = fifo_hit(P, F, s);
            // The result stream is written once and read by nobody on this chip before the kernel ends: marked non-temporal
            // (global_store ... nt) it does not take L2 lines from the second cell's records.  Fill pass 1.05 -> 0.98 ms,
            // 655-660 -> 680-695 frames/s on one box, alternating runs; "sc0 nt", "sc1 nt", "sc0 sc1 nt" the same, "sc1"
            // alone slower than plain stores (profiles/r03_store_policy_ab.txt).
#ifdef MOLAR_HIP_DEBUG_KNOBS
            // knock-outs of the store path (timing only, wrong results; profiles/r04_store_knockout.txt): bit 4 no stores
            // (everything else is computed), bit 5 every flush lands on the slot's first 64 entries
            const float dist = __builtin_sqrtf(h.d2);
            const bool wr = (P.debug_skip & 16u) ? dist == -1.0f : true;
            if (F.has_pairs && wr) __builtin_nontemporal_store(((unsigned long long)h.j << 32) | h.i, reinterpret_cast<unsigned long long *>(&F.pairs[pos]));
            if (F.has_dist && wr) __builtin_nontemporal_store(dist, &F.dist[pos]);
#else
            if (F.has_pairs) __builtin_nontemporal_store(((unsigned long long)h.j << 32) | h.i, reinterpret_cast<unsigned long long *>(&F.pairs[pos]));
            // d2.sqrt() (:448): llvm.sqrt.f32 without fpmath metadata = IEEE correctly rounded
            if (F.has_dist) __builtin_nontemporal_store(__builtin_sqrtf(h.d2), &F.dist[pos]);
#endif
        }
    }
    F.head += count;
}

// >= 64 entries are queued: write them out in 64-entry blocks.  A slot's output starts wherever the slots before it
// end, so its first flush only goes up to the next 64-entry boundary of the output arrays; from then on every
// flush is a naturally aligned block (unaligned 512-byte stores straddle five cache lines instead of four and
// reach 3.9 instead of 4.6 TB/s on this chip, profiles/microbench/store_shapes_mi355x.txt).
template <int KIND>
__device__ __forceinline__ void fifo_drain(const SearchParams &P, Fifo &F, uint32_t lane) {
    __builtin_amdgcn_wave_barrier();
    do {
        fifo_flush<KIND>(P, F, F.quota, lane);
        F.quota = 64u;
    } while (F.tail - F.head >= 64u);
}

// The plain (and same-cell) entries of the fill pass queue up to FIFO_WIDE hits and write 128 at a time: lane l takes entries
// 2 l and 2 l + 1 - one 16-byte store of two (i, j) pairs and one 8-byte store of two distances per lane, 1 KB + 512 B
// contiguous per wave instruction instead of 512 B + 256 B.  The slot's first flush goes, one entry per lane in rounds of 64
// (fifo_flush), up to the next 128-entry boundary of the output; every later one is a naturally aligned block.
template <int KIND>
__device__ __forceinline__ void fifo_flush_wide(const SearchParams &P, Fifo &F, uint32_t lane) {
    typedef uint32_t u4_t __attribute__((ext_vector_type(4)));
    typedef float f2_t __attribute__((ext_vector_type(2)));
    const uint32_t e = F.head + 2u * lane;
    if (e + 1u < F.room) {
        const uint32_t s0 = e & (FIFO_WIDE - 1), s1 = (e + 1u) & (FIFO_WIDE - 1);
        const u4_t pr = {F.fi[s0], F.fj[s0], F.fi[s1], F.fj[s1]};
        const f2_t ds = {__builtin_sqrtf(__uint_as_float(F.fd[s0])), __builtin_sqrtf(__uint_as_float(F.fd[s1]))};   // d2.sqrt() (:448)
        // F.pairs / F.dist point at entry (slot base + lane): + head + lane is entry head + 2 lane
        uint32_t at = F.head + lane;
        bool wr = true;
#ifdef MOLAR_HIP_DEBUG_KNOBS
        if (P.debug_skip & 16u) wr = ds.x == -1.0f;       // knock-outs, see fifo_flush
        if (P.debug_skip & 32u) at = (F.head & 127u) + lane;
#endif
        if (F.has_pairs && wr) __builtin_nontemporal_store(pr, reinterpret_cast<u4_t *>(&F.pairs[at]));
        if (F.has_dist && wr) __builtin_nontemporal_store(ds, reinterpret_cast<f2_t *>(&F.dist[at]));
    } else if (e < F.room) {         // (the count pass gave this slot an odd number of entries and the fill pass more: never, but never past them)
        const uint32_t s0 = e & (FIFO_WIDE - 1);
        if (F.has_pairs) __builtin_nontemporal_store(((unsigned long long)F.fj[s0] << 32) | F.fi[s0], reinterpret_cast<unsigned long long *>(&F.pairs[F.head + lane]));
        if (F.has_dist) __builtin_nontemporal_store(__builtin_sqrtf(__uint_as_float(F.fd[s0])), &F.dist[F.head + lane]);
    }
    F.head += 128u;
}

// >= 128 entries are queued (plain entries of the fill pass)
template <int KIND>
__device__ __forceinline__ void fifo_drain_wide(const SearchParams &P, Fifo &F, uint32_t lane) {
    __builtin_amdgcn_wave_barrier();
    do {
        if (F.hist) {                    // consumer-fused mode on the generic kernel: the same queue, binned 64 at a time
            fifo_flush<KIND, FIFO_WIDE>(P, F, 64u, lane);
            fifo_flush<KIND, FIFO_WIDE>(P, F, 64u, lane);
        } else if (F.quota != 128u) {    // the slot's first flush: up to the next 128-entry boundary of the output, one entry per lane
            uint32_t left = F.quota;
            while (left) {
                const uint32_t c = left < 64u ? left : 64u;
                fifo_flush<KIND, FIFO_WIDE>(P, F, c, lane);
                left -= c;
            }
            F.quota = 128u;
        } else {
            fifo_flush_wide<KIND>(P, F, lane);
        }
    } while (F.tail - F.head >= 128u);
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
template <int KIND, bool FILL, int WK, bool TRI, int NCH, bool RUNTIME_NCH, bool WIDE4 = false>
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
                    if (F.tail - F.head >= 64u) fifo_drain<KIND>(P, F, lane);
                }
            };
            // two chunks (c0 = first chunk index) against row i; use0/use1: chunk is live
            auto pair_body = [&](uint32_t c0, bool use0, bool use1, bool rag0, bool rag1, v2f qx, v2f qy, v2f qz, v2f qv,
                                 uint32_t id0, uint32_t id1) {
                const v2f d2 = pair_d2x2<WK, WIDE4>(P, B, T.wrap, px, py, pz, qx, qy, qz);
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
                    if (F.tail - F.head >= 64u) fifo_drain<KIND>(P, F, lane);
                }
                total += 1;
            }
        }
    }
    if (FILL && F.tail != F.head) {
        __builtin_amdgcn_wave_barrier();
        fifo_flush<KIND>(P, F, F.tail - F.head, lane);
    }
    return total;
}

// Exact PeriodicBox::distance_squared for ONE candidate (periodic_box.rs:286-318, 379-381) in the general
// matrix form; the kernel library is built without FMA contraction, and products with matrix entries
// that are exactly zero do not change an IEEE sum (up to the sign of a zero the squares cannot see), so
// this equals the reference for diagonal, triangular and full boxes alike.
// The grid stores every atom inside the primary cell along periodic dimensions (populate_pbc wraps
// them, distance_search.rs:183-196), so |f[d]| < 1.5 for a wrapped dimension and f32::round reduces
// to "copysign(1, f) if |f| >= 0.5 else 0" - the same value, three instructions instead of six.
template <bool WIDE4>
__device__ __forceinline__ float wrapped_d2_exact(const SearchParams &P, uint32_t wrap, float vx, float vy, float vz) {
    const float *I = P.box.inv, *M = P.box.m;
    float fx = (I[0] * vx + I[3] * vy) + I[6] * vz;
    float fy = (I[1] * vx + I[4] * vy) + I[7] * vz;
    float fz = (I[2] * vx + I[5] * vy) + I[8] * vz;
    if (wrap & 1u) fx -= (fabsf(fx) >= 0.5f ? copysignf(1.0f, fx) : 0.0f);
    if (wrap & 2u) fy -= (fabsf(fy) >= 0.5f ? copysignf(1.0f, fy) : 0.0f);
    if (wrap & 4u) fz -= (fabsf(fz) >= 0.5f ? copysignf(1.0f, fz) : 0.0f);
    const float sx = (M[0] * fx + M[3] * fy) + M[6] * fz;
    const float sy = (M[1] * fx + M[4] * fy) + M[7] * fz;
    const float sz = (M[2] * fx + M[5] * fy) + M[8] * fz;
    float best2 = (sx * sx + sy * sy) + sz * sz;
    if (P.box.nshift != 0 && wrap == MOLAR_HIP_PBC_FULL) {   // triclinic candidates (:304-317)
        if (WIDE4) {
            for (int k = 0; k < P.box.nshift; k += 4) {          // four images per trip (SearchParams::shifts4, see pair_d2x2)
                float sh[12];
#pragma unroll
                for (int q = 0; q < 12; ++q) sh[q] = P.shifts4[3 * k + q];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float cx = sx + sh[3 * q], cy = sy + sh[3 * q + 1], cz = sz + sh[3 * q + 2];
                    const float n2 = (cx * cx + cy * cy) + cz * cz;
                    best2 = n2 < best2 ? n2 : best2;
                }
            }
        } else {
            for (int k = 0; k < P.box.nshift; ++k) {
                const float cx = sx + P.box.shifts[3 * k], cy = sy + P.box.shifts[3 * k + 1], cz = sz + P.box.shifts[3 * k + 2];
                const float n2 = (cx * cx + cy * cy) + cz * cz;
                best2 = n2 < best2 ? n2 : best2;
            }
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

// Fast path for the bulk of the work: fixed-cutoff cell pairs whose second cell fits in registers
// (NCH <= 8 chunks of 64): plain, triangular (same cell) or wrapped.
//  * the slot's first-cell atoms are staged in LDS and each row is fetched with ONE broadcast
//    ds_read_b128, so the arithmetic runs on VGPR operands only (an SGPR source halves the issue rate
//    of f32 VALU ops on gfx950, profiles/microbench/valu_rate.hip) and no v_readlane sits in the loop;
//  * the count pass never leaves the VALU: hits are added per lane through the carry of the compare
//    and reduced across the wave once per slot (v_cmp -> s_bcnt1 -> s_add costs ~14 cycles per chunk);
//  * lanes past the end of the second cell hold a coordinate so large that d2 overflows and the
//    compare fails by itself - no separate validity mask;
//  * rows that provably cannot have a hit are skipped (`live`);
//  * WRAPPED entries (a cell pair across the periodic boundary) need PeriodicBox::distance_squared:
//    inv*v, round, M*v - 4x the arithmetic of a plain pair.  Only the DECISION d2 <= cutoff^2 and the
//    d2 of actual hits must equal the reference's, so the row loop classifies candidates with the
//    plain distance to the image of the second cell (b + S, S = the lattice vector of the wrap):
//    below the band [lo,hi] around cutoff^2 it is a hit, above it a miss, and only candidates INSIDE the
//    band (a fraction of a percent of the chunks) are evaluated with the exact formula.  Hits carry
//    (row, atom position) through the FIFO and their exact d2 is recomputed densely at flush time.
//    The band is >10x the worst-case disagreement between the two evaluations (make_params()).
// Count pass of plain and same-cell entries.  A count does not depend on the order in which candidates are
// visited, so the second cell is walked in the SPATIAL order prepared by cell_order_kernel: its 64-atom chunks are
// compact, and a (row, chunk) pair is skipped when the row's atom is farther than the cutoff from the chunk's
// bounding box - the same exact f32 lower-bound argument as the row pruning of run_fast.  In the reference's
// order a chunk spans the whole cell and nothing could be skipped; here ~43 % of the candidate evaluations of
// the 13 neighbour entries go away.  Row-outer loop (one LDS broadcast per row), a scalar test per chunk,
// VALU-only counting as in run_fast.
template <int KIND, int NCH, bool TRI>
__device__ __forceinline__ uint32_t run_count_sorted(const SearchParams &P, const Task &T, uint32_t i0, float4 *la,
                                                     uint32_t lane) {
    const float cutoff2 = P.cutoff2;
    const uint32_t rows = __builtin_amdgcn_readfirstlane(T.n1 - i0 < T.rps ? T.n1 - i0 : T.rps);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane < rows) a = gload4(P.sa, T.a0 + i0 + lane);
    la[lane] = a;
    float bx[NCH], by[NCH], bz[NCH];
    uint32_t bpos[NCH];          // position of the atom in the reference's cell order (same-cell entries: j > i)
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const uint32_t jj = (uint32_t)k * 64u + lane;
        float4 q = make_float4(3.0e38f, 3.0e38f, 3.0e38f, 0.f);
        uint32_t pj = 0u;
        if (jj < T.n2) {
            q = gload4(P.perm_b, T.b0 + jj);
            pj = __float_as_uint(q.w);
        }
        bx[k] = q.x; by[k] = q.y; bz[k] = q.z; bpos[k] = pj;
    }
    const uint32_t ubase = (T.b0 >> 6) + T.cb;
    unsigned long long livek[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const float4 lo = gload4(P.chunk_aabb_b, 2u * (ubase + k)), hi = gload4(P.chunk_aabb_b, 2u * (ubase + k) + 1u);
        const bool need = lane < rows && !(aabb_d2(a.x, a.y, a.z, lo.x, lo.y, lo.z, hi.x, hi.y, hi.z) > cutoff2);
        livek[k] = __builtin_amdgcn_ballot_w64(need);
    }
    __builtin_amdgcn_wave_barrier();
    uint32_t acc = 0;
    unsigned long long live = 0ull;
#pragma unroll
    for (int k = 0; k < NCH; ++k) live |= livek[k];
    while (live) {
        const uint32_t r = (uint32_t)__builtin_ctzll(live);
        live &= ~(1ull << r);
        const float4 p = lload4(la, r);              // one broadcast ds_read per row
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            if (!((livek[k] >> r) & 1ull)) continue;                              // wave-uniform: scalar branch
            const float dx = bx[k] - p.x, dy = by[k] - p.y, dz = bz[k] - p.z;     // p2 - p1
            float d2 = (dx * dx + dy * dy) + dz * dz;                            // |p2-p1|^2 (:446, :460)
            if (TRI) d2 = (bpos[k] > i0 + r) ? d2 : INFINITY;                    // same cell: j in i+1..n (:443)
            asm volatile("v_cmp_ge_f32 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(acc) : "v"(d2), "s"(cutoff2) : "vcc");
        }
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    return acc;
}

// ================================================================= count pass on the matrix cores
// Count pass of plain and same-cell entries (no periodic image, second cell <= 320 atoms).  |p2 - p1|^2 - cutoff^2
// of a 32 x 32 block of (row, atom) pairs is ONE v_mfma_f32_32x32x16_f16: with both positions taken relative to the
// second cell's origin O and split into f16 hi + lo parts (22 significant bits; subnormal parts are multiplied exactly,
// profiles/microbench/mfma_f16_denormals_mi355x.txt),
//     |b - a|^2 - c = sum_k A[row][k] * B[k][col],
//     A = (-2 ah, -2 ah, |a|^2 - c (hi, lo) | -2 al, -2 al, 1, 1),   B = (bh, bl, 1, 1 | bh, bl, |b|^2 (hi, lo))
// (f16 products are exact in the f32 accumulator).  The B records are prepared once per frame by place_order_kernel in
// the reference's cell order (a same-cell entry's j > i is then a comparison of block coordinates: blocks below the
// diagonal are skipped, blocks on it masked); the A records of the slot's 64 rows are built in the prologue and redistributed
// through LDS.  Only the DECISION d2 <= cutoff^2 has to equal the reference's f32 evaluation: the sign of an accumulator
// decides when its magnitude exceeds E, a bound on every difference between the two evaluations, with R = Ra + Rb the
// bounds on |a - O| (reduced over the slot's rows) and |b - O| (stored with the cell), c = cutoff^2, in units of 2^-22:
//     positions relative to O in f32 (2^-24 each) and their 22-bit split, times 2 |b - a| <= 2 R          2.6 R^2
//     the two squared norms in f32 (4 roundings) and their hi + lo split                                  2 R^2 + 2 c
//     16 f32 additions inside the instruction, partial sums <= R^2 + c                                    4 (R^2 + c)
//     the reference's own roundings (three differences, three squares, two sums: 5 * 2^-24 relative)      1.25 R^2
// together < 9.9 R^2 + 6 c; E = 2^-22 * (16 R^2 + 8 c).  Blocks with an accumulator inside (-E, E) - 3-4 % of them - are
// recounted with the exact f32 formula after the loop; slots whose bound is not small against c, or not finite, are left
// to run_count_sorted.  Per block: 16 v_alignbit (sign bits), 8 v_min3 (magnitudes), a popcount - 28 VALU instructions
// for 1024 candidates instead of 160 (90 after run_count_sorted's bounding-box skips).  All memory a slot needs (rows,
// origin, <= 10 B records per lane) is requested at once: with one block column prefetched the slot was a chain of ten
// exposed latencies and the kernel no faster than before.  The count kernel runs 6 waves per SIMD (80 registers) for it.
typedef _Float16 v8h_t __attribute__((ext_vector_type(8)));
typedef float v16f_t __attribute__((ext_vector_type(16)));

__device__ __forceinline__ uint32_t pack_h2(_Float16 a, _Float16 b) {
    return (uint32_t)__builtin_bit_cast(unsigned short, a) | ((uint32_t)__builtin_bit_cast(unsigned short, b) << 16);
}

constexpr int MFMA_TILES = 10;     // block columns (32 atoms) per second cell the matrix-core count keeps in registers: cells of <= 320 atoms

// E of the derivation above plus the ABSOLUTE limits of f16 the relative terms do not see: a lo part below 2^-14 is a
// subnormal f16 with a quantum of 2^-24, so each of the six relative coordinates carries up to 2^-25 of its own
// whatever its size (times 2 |b - a| <= 4 R in d2: < 2^-22 * 1.8 R), and so does each of the two norm splits
// (2 * 2^-25).  With them the bound is 2^-22 * (16 R^2 + 8 c + 4 R + 1); at MD scales the addition is a few percent,
// below a cutoff of ~0.05 it is what decides (a CPU emulation of the records showed 6e-8 of error at every small cutoff).
__device__ __forceinline__ float mfma_error_bound(float R, float c) {
    return 2.3841858e-07f * ((16.0f * R * R + 8.0f * c) + (4.0f * R + 1.0f));          // 2^-22 * (...)
}
// The matrix-core count is used for a slot only when the bound is small against the cutoff (few undecided blocks) and
// every f16 operand stays finite: |a|^2 - c and |b|^2 are at most R^2 + c in size, and the padding value 65504 ("never a
// hit") must remain the largest term, so c + R^2 is kept below 3e4.
__device__ __forceinline__ bool mfma_bound_usable(float R, float E, float c) {
    return R < 64.0f && E < 0.02f * c && (c + R * R) < 3.0e4f;
}

template <int KIND, bool TRI>
__device__ __forceinline__ uint32_t run_count_mfma(const SearchParams &P, const Task &T, uint32_t i0, float4 *la, uint4 *lh,
                                                   uint32_t lane, bool &done) {
    typedef uint32_t u4_t __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) u4_t lds_u4;
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    typedef __attribute__((address_space(1))) u4_t glb_u4;
    const float cutoff2 = P.cutoff2;
    const uint32_t rows = __builtin_amdgcn_readfirstlane(T.n1 - i0 < T.rps ? T.n1 - i0 : T.rps);
    const uint32_t kh = lane >> 5, cl = lane & 31u;
    const uint32_t nct = (T.n2 + 31u) >> 5;
    // Everything the slot reads from memory is requested at once - its rows, the second cell's origin and ALL B
    // records: a slot is a chain of memory latencies, not of arithmetic (one block column prefetched: 0.50 ms for the
    // count kernel, exactly what 8 waves x (2 + 9 exposed latencies) per slot predict).
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane < rows) a = gload4(P.sa, T.a0 + i0 + lane);
    const float4 org = gload4(P.cell_org_b, T.cb);
    // Rows that cannot have a hit (run_fast's exact pruning against the second cell's bounding box: the same f32
    // expression on the point-to-box distances bounds every d2 of the row from below) are left out, and the rows that
    // remain are COMPACTED in front of the instruction - a count does not care in which order its rows are visited.  With
    // at most 32 live rows the slot needs one row block instead of two: most slots of the corner entries (~2/3 of their
    // rows pruned) and a third of the edge entries'.  Same-cell entries keep their rows in place (the j > i mask below
    // argues with positions, and a row inside its own cell's box is never pruned).
    bool need = lane < rows;
    if (!TRI) {
        const float4 blo = gload4(P.aabb_b, 2 * T.cb), bhi = gload4(P.aabb_b, 2 * T.cb + 1);
        need = need && !(aabb_d2(a.x, a.y, a.z, blo.x, blo.y, blo.z, bhi.x, bhi.y, bhi.z) > cutoff2);
    }
    const unsigned long long live = __builtin_amdgcn_ballot_w64(need);
    const uint32_t nlive = TRI ? rows : (uint32_t)__popcll(live);
    if (!TRI && nlive == 0u) return 0u;
    const uint32_t rank = TRI ? lane : __builtin_amdgcn_mbcnt_hi((uint32_t)(live >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)live, 0u));
    u4_t bq[MFMA_TILES];
#pragma unroll
    for (int t = 0; t < MFMA_TILES; ++t) {
        const uint32_t col = (uint32_t)t * 32u + cl;
        // (a full block column - a wave-uniform test - is loaded as it is; only the cell's last, ragged one pays the per-lane
        // selects.  Non-finite atoms carry the "never a hit" record since the grid build, place_order_kernel.)
        if (32u * (uint32_t)(t + 1) <= T.n2) {
            bq[t] = ((const glb_u4 *)P.h16_b)[T.b0 + col];
        } else {
            bq[t] = u4_t{0u, 0u, 0u, 0x00007BFFu};                      // atom past the end: |b|^2 = 65504, never a hit
            if (col < T.n2) bq[t] = ((const glb_u4 *)P.h16_b)[T.b0 + col];
        }
    }
    if (TRI || need) la[rank] = a;                  // f32 rows (compacted), for the exact decision inside the band
    const float r0 = a.x - org.x, r1 = a.y - org.y, r2 = a.z - org.z;
    float ra2 = need ? (r0 * r0 + r1 * r1) + r2 * r2 : 0.0f;
    const bool fin = ra2 == ra2;
    for (int off = 32; off > 0; off >>= 1) ra2 = fmaxf(ra2, __shfl_xor(ra2, off, 64));
    const float R = 1.0001f * __builtin_sqrtf(ra2) + org.w;
    // (R and E live across the whole block loop: pinned to SGPRs, see HistFifo)
    const float E = uniform_f32(mfma_error_bound(R, cutoff2));
    if (__builtin_amdgcn_ballot_w64(!fin) != 0ull || !mfma_bound_usable(R, E, cutoff2)) {
        done = false;                               // not finite / not small: the exact path takes the slot
        return 0u;
    }
    {   // A records of this lane's row: k = 0..7 and k = 8..15, stored at the row's rank among the live rows
        u4_t k0 = {0u, 0u, 0u, 0x00007BFFu}, k1 = {0u, 0u, 0u, 0x3C003C00u};      // row past the end: +65504
        if (!TRI) {
            ((lds_u4 *)lh)[2u * lane] = k0;
            ((lds_u4 *)lh)[2u * lane + 1u] = k1;
            __builtin_amdgcn_wave_barrier();
        }
        if (need) {
            const _Float16 h0 = (_Float16)r0, h1 = (_Float16)r1, h2 = (_Float16)r2;
            const _Float16 l0 = (_Float16)(r0 - (float)h0), l1 = (_Float16)(r1 - (float)h1), l2 = (_Float16)(r2 - (float)h2);
            const float e0 = (float)h0 + (float)l0, e1 = (float)h1 + (float)l1, e2 = (float)h2 + (float)l2;
            const float na = ((e0 * e0 + e1 * e1) + e2 * e2) - cutoff2;
            const _Float16 nh = (_Float16)na, nl = (_Float16)(na - (float)nh);
            const _Float16 m2 = (_Float16)-2.0f;
            const _Float16 g0 = m2 * h0, g1 = m2 * h1, g2 = m2 * h2, s0 = m2 * l0, s1 = m2 * l1, s2 = m2 * l2;
            k0 = u4_t{pack_h2(g0, g1), pack_h2(g2, g0), pack_h2(g1, g2), pack_h2(nh, nl)};
            k1 = u4_t{pack_h2(s0, s1), pack_h2(s2, s0), pack_h2(s1, s2), 0x3C003C00u};
        }
        if (TRI || need) {
            ((lds_u4 *)lh)[2u * rank] = k0;
            ((lds_u4 *)lh)[2u * rank + 1u] = k1;
        }
    }
    __builtin_amdgcn_wave_barrier();
    const u4_t a0q = ((const lds_u4 *)lh)[2u * cl + kh], a1q = ((const lds_u4 *)lh)[2u * (32u + cl) + kh];
    const v8h_t A0 = __builtin_bit_cast(v8h_t, a0q), A1 = __builtin_bit_cast(v8h_t, a1q);
    __builtin_amdgcn_wave_barrier();
    // blocks with an accumulator inside (-E, E) are not counted here: their ids go to a list (the A staging area is
    // free again) and they are recounted with the exact formula after the loop (2-3 % of the blocks)
    lds_u32 *todo = (lds_u32 *)lh;
    uint32_t ntodo = 0;
    uint32_t cnt = 0;
#pragma unroll
    for (int t = 0; t < MFMA_TILES; ++t) {
        if ((uint32_t)t < nct) {
            u4_t bt = bq[t];
            if (kh == 0u) bt.w = 0x3C003C00u;                           // k = 6, 7 of the first half: (1, 1)
            const v8h_t B = __builtin_bit_cast(v8h_t, bt);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                // same-cell entries (j > i, :443; B records are in the reference's order, so atom `col` has position col):
                // blocks below the diagonal hold no pair, blocks above it all of theirs, blocks on it are masked
                bool diag = false;
                if (32u * (uint32_t)rt >= nlive) continue;                                   // no live row in this block
                if (TRI) {
                    const uint32_t row0 = i0 + 32u * (uint32_t)rt;
                    if (32u * (uint32_t)t + 31u <= row0) continue;                       // every j <= every i
                    diag = 32u * (uint32_t)t <= row0 + 31u;
                }
                v16f_t acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(rt == 0 ? A0 : A1, B, acc, 0, 0, 0);
                // accumulator i of lane (kh, cl): row 32 rt + 8 (i / 4) + 4 kh + i % 4, atom 32 t + cl
                if (TRI && diag) {
                    const int tv = (int)(32u * (uint32_t)t + cl) - (int)(i0 + 32u * (uint32_t)rt + 4u * kh);
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[i] = (tv > 8 * (i / 4) + (i % 4)) ? acc[i] : 1.0e30f;
                }
                uint32_t h = 0u;
                float m = INFINITY;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    h = __builtin_amdgcn_alignbit(h, __float_as_uint(acc[i]), 31);       // h = 2 h + sign
                    m = __builtin_fminf(m, __builtin_fabsf(acc[i]));
                }
                if (__builtin_amdgcn_ballot_w64(m < E) == 0ull) {
                    cnt += (uint32_t)__popc(h);
                } else {
                    if (lane == 0) todo[ntodo] = (uint32_t)(2 * t + rt);
                    ++ntodo;
                }
            }
        }
    }
    if (ntodo) {
        __builtin_amdgcn_wave_barrier();
        // exact recount of the listed blocks: lanes = the block's 32 atoms, each half of the wave takes 16 of its 32 rows
        for (uint32_t q = 0; q < ntodo; ++q) {
            const uint32_t id = __builtin_amdgcn_readfirstlane(todo[q]);
            const uint32_t ct = id >> 1, rt = id & 1u;
            const uint32_t col = ct * 32u + cl;
            float4 b = make_float4(3.0e38f, 3.0e38f, 3.0e38f, 0.f);
            if (col < T.n2) b = gload4(P.sb, T.b0 + col);
            for (uint32_t r = 0; r < 16u; ++r) {
                const uint32_t row = 32u * rt + 16u * kh + r;
                const float4 p = lload4(la, row);
                const float dx = b.x - p.x, dy = b.y - p.y, dz = b.z - p.z;     // p2 - p1
                const float d2 = (dx * dx + dy * dy) + dz * dz;                // |p2-p1|^2 (:446, :460)
                cnt += (row < nlive && (!TRI || col > i0 + row) && d2 <= cutoff2) ? 1u : 0u;
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
    return cnt;
}

// Count pass of band-classified WRAPPED entries on the matrix cores.  The vector path (run_fast<.., WRAPPED, .., MASKED>)
// evaluates every candidate of these entries in f32 (plain distance to the image of the second cell, two band compares,
// hit history by carry-add): 6 % of the pairs, but as many VALU instructions as the whole matrix-core count of the other
// 94 % (run as separate kernels: 134 us against 267 us).  Here the same 32 x 32 blocks go through
// v_mfma_f32_32x32x16_f16 with the row side shifted by the wrap's lattice vector (a - S, so that the B records of the
// second cell serve unchanged), and the result is written in EXACTLY the hit-history format the fill pass replays
// (word (g, k) of lane l = the bits of atom 64 k + l for the g-th group of 32 LIVE rows, first row in bit 31):
//  * the rows that survive run_fast's image-box pruning are compacted in front of the instruction (their records go to
//    consecutive LDS slots by rank), so a row block IS a history group and pruned rows cost nothing - half of these slots
//    need one row block instead of two;
//  * a candidate is decided by the sign of its accumulator when |acc| > band + E: band = rel * cutoff^2 of make_params (the
//    argument of run_fast: outside it the plain distance to the image decides like PeriodicBox::distance_squared), E the
//    matrix-core error bound, plus 4 eta rc for measuring from fl(a - S) instead of to fl(b + S) (eta = 2^-24 (2 L + 2 rc));
//  * candidates inside that range (one or two per slot) are queued and decided with the exact formula
//    (periodic_box.rs:286-318), all at once; their bits are OR-ed into the stored words afterwards;
//  * the two 16-row halves of a column's hit bits live in lanes l and l + 32: one exchange per (chunk, row block) and a
//    nibble interleave put them into row order.
// Returns false (nothing written) when the slot has to take the vector path: bound too wide, not finite.
template <int KIND>
__device__ __forceinline__ bool run_count_mfma_wrapped(const SearchParams &P, const Task &T, uint32_t i0, float4 *la, uint4 *lh,
                                                       uint32_t lane, uint32_t *mwords, uint32_t &count_out) {
    typedef uint32_t u4_t __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) u4_t lds_u4;
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    typedef __attribute__((address_space(1))) u4_t glb_u4;
    const float cutoff2 = P.cutoff2;
    const uint32_t rows = __builtin_amdgcn_readfirstlane(T.n1 - i0 < 64u ? T.n1 - i0 : 64u);
    const uint32_t kh = lane >> 5, cl = lane & 31u;
    const uint32_t nct = (T.n2 + 31u) >> 5, nch = (T.n2 + 63u) >> 6;
    float Sx = 0.f, Sy = 0.f, Sz = 0.f;      // b + S is the image of the second cell next to the first one (run_fast)
    for (int d = 0; d < 3; ++d) {
        if (!((T.wrap >> d) & 1u)) continue;
        const float sgn = ((T.wrap_b >> d) & 1u) ? 1.0f : -1.0f;
        Sx += sgn * P.box.m[3 * d];
        Sy += sgn * P.box.m[3 * d + 1];
        Sz += sgn * P.box.m[3 * d + 2];
    }
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane < rows) a = gload4(P.sa, T.a0 + i0 + lane);
    const float4 org = gload4(P.cell_org_b, T.cb);
    const float4 blo = gload4(P.aabb_b, 2 * T.cb), bhi = gload4(P.aabb_b, 2 * T.cb + 1);
    // the live rows, exactly as run_fast finds them in both passes
    const bool need = lane < rows && !(aabb_d2(a.x - Sx, a.y - Sy, a.z - Sz, blo.x, blo.y, blo.z, bhi.x, bhi.y, bhi.z) > P.prune_limit2);
    const unsigned long long live = __builtin_amdgcn_ballot_w64(need);
    const uint32_t nlive = (uint32_t)__popcll(live);
    if (nlive == 0u) {
        count_out = 0u;
        return true;
    }
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(live >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)live, 0u));
    const float r0 = (a.x - Sx) - org.x, r1 = (a.y - Sy) - org.y, r2 = (a.z - Sz) - org.z;
    float ra2 = need ? (r0 * r0 + r1 * r1) + r2 * r2 : 0.0f;
    float big = need ? fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fabsf(a.z)) : 0.0f;
    const bool fin = ra2 == ra2;
    for (int off = 32; off > 0; off >>= 1) {
        ra2 = fmaxf(ra2, __shfl_xor(ra2, off, 64));
        big = fmaxf(big, __shfl_xor(big, off, 64));
    }
    const float R = 1.0001f * __builtin_sqrtf(ra2) + org.w;
    const float rc = __builtin_sqrtf(P.band_hi);
    const float L = (fmaxf(fmaxf(fabsf(org.x), fabsf(org.y)), fabsf(org.z)) + R) + (fmaxf(fmaxf(fabsf(Sx), fabsf(Sy)), fabsf(Sz)) + big);
    const float Em = mfma_error_bound(R, cutoff2);
    // |acc| beyond this: decided by the sign.  (band_hi - cutoff^2 = rel * cutoff^2 is the half width of run_fast's band)
    const float E = uniform_f32((P.band_hi - cutoff2) * 1.0001f + Em + 4.0f * (5.9604645e-08f * (2.0f * L + 2.0f * rc)) * rc);
    if (__builtin_amdgcn_ballot_w64(!fin || !(E == E) || !mfma_bound_usable(R, Em, cutoff2) || !(E < 0.06f * cutoff2)) != 0ull) return false;
    // B records of this lane's column of every block column, requested at once (see run_count_mfma)
    u4_t bq[MFMA_TILES];
#pragma unroll
    for (int t = 0; t < MFMA_TILES; ++t) {
        const uint32_t col = (uint32_t)t * 32u + cl;
        // (a full block column - a wave-uniform test - is loaded as it is; only the cell's last, ragged one pays the per-lane
        // selects.  Non-finite atoms carry the "never a hit" record since the grid build, place_order_kernel.)
        if (32u * (uint32_t)(t + 1) <= T.n2) {
            bq[t] = ((const glb_u4 *)P.h16_b)[T.b0 + col];
        } else {
            bq[t] = u4_t{0u, 0u, 0u, 0x00007BFFu};                      // atom past the end: |b|^2 = 65504, never a hit
            if (col < T.n2) bq[t] = ((const glb_u4 *)P.h16_b)[T.b0 + col];
        }
    }
    {   // row records by RANK among the live rows; everything behind them is "a row past the end"
        ((lds_u4 *)lh)[2u * lane] = u4_t{0u, 0u, 0u, 0x00007BFFu};
        ((lds_u4 *)lh)[2u * lane + 1u] = u4_t{0u, 0u, 0u, 0x3C003C00u};
        __builtin_amdgcn_wave_barrier();
        if (need) {
            const _Float16 h0 = (_Float16)r0, h1 = (_Float16)r1, h2 = (_Float16)r2;
            const _Float16 l0 = (_Float16)(r0 - (float)h0), l1 = (_Float16)(r1 - (float)h1), l2 = (_Float16)(r2 - (float)h2);
            const float e0 = (float)h0 + (float)l0, e1 = (float)h1 + (float)l1, e2 = (float)h2 + (float)l2;
            const float na = ((e0 * e0 + e1 * e1) + e2 * e2) - cutoff2;
            const _Float16 nh = (_Float16)na, nl = (_Float16)(na - (float)nh);
            const _Float16 m2 = (_Float16)-2.0f;
            const _Float16 g0 = m2 * h0, g1 = m2 * h1, g2 = m2 * h2, s0 = m2 * l0, s1 = m2 * l1, s2 = m2 * l2;
            ((lds_u4 *)lh)[2u * rank] = u4_t{pack_h2(g0, g1), pack_h2(g2, g0), pack_h2(g1, g2), pack_h2(nh, nl)};
            ((lds_u4 *)lh)[2u * rank + 1u] = u4_t{pack_h2(s0, s1), pack_h2(s2, s0), pack_h2(s1, s2), 0x3C003C00u};
            la[rank] = a;                            // the live rows, compacted, for the exact decisions
        }
    }
    __builtin_amdgcn_wave_barrier();
    const u4_t a0q = ((const lds_u4 *)lh)[2u * cl + kh], a1q = ((const lds_u4 *)lh)[2u * (32u + cl) + kh];
    const v8h_t A0 = __builtin_bit_cast(v8h_t, a0q), A1 = __builtin_bit_cast(v8h_t, a1q);
    __builtin_amdgcn_wave_barrier();
    lds_u32 *todo = (lds_u32 *)lh;              // candidates inside the range: (live row << 16 | atom), decided exactly below
    constexpr uint32_t TODO_CAP = 512u;
    uint32_t ntodo = 0, cnt = 0;
    const uint32_t ngroups = nlive > 32u ? 2u : 1u;
    // hit word of one block: bit 15 - i = accumulator i = live row 32 rt + 8 (i / 4) + 4 kh + i % 4
    auto block = [&](int t, int rt) __attribute__((always_inline)) -> uint32_t {
        u4_t bt = bq[t];
        if (kh == 0u) bt.w = 0x3C003C00u;                               // k = 6, 7 of the first half: (1, 1)
        const v16f_t zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const v16f_t acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(rt == 0 ? A0 : A1, __builtin_bit_cast(v8h_t, bt), zero, 0, 0, 0);
        uint32_t h = 0u;
        float m = INFINITY;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            h = __builtin_amdgcn_alignbit(h, __float_as_uint(acc[i]), 31);
            m = __builtin_fminf(m, __builtin_fabsf(acc[i]));
        }
        if (__builtin_amdgcn_ballot_w64(m < E) != 0ull) {
            uint32_t bm = 0u;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const bool ib = __builtin_fabsf(acc[i]) < E;
                bm = (bm << 1) | (ib ? 1u : 0u);
                const unsigned long long mk = __builtin_amdgcn_ballot_w64(ib);
                if (mk) {
                    const uint32_t at = ntodo + __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
                    const uint32_t rl = 32u * (uint32_t)rt + 8u * (uint32_t)(i / 4) + 4u * kh + (uint32_t)(i % 4);
                    if (ib && at < TODO_CAP) todo[at] = (rl << 16) | (32u * (uint32_t)t + cl);
                    ntodo += (uint32_t)__popcll(mk);
                }
            }
            h &= ~bm;
        }
        return h;
    };
    // nibbles of a 16-bit word to the low halves of the four bytes
    auto spread = [](uint32_t x) -> uint32_t {
        x = (x | (x << 8)) & 0x00FF00FFu;
        return (x | (x << 4)) & 0x0F0F0F0Fu;
    };
#pragma unroll
    for (int k = 0; k < MFMA_TILES / 2; ++k) {
        if ((uint32_t)k < nch) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                if ((uint32_t)rt >= ngroups) continue;
                const uint32_t hx = block(2 * k, rt);
                const uint32_t hy = (uint32_t)(2 * k + 1) < nct ? block(2 * k + 1, rt) : 0u;
                // lane l < 32 owns atom 64 k + l (block 2k): it holds that column's rows 4 kh' = 0 half and gets the other
                // half from lane l + 32; lane l >= 32 owns atom 64 k + l (block 2k + 1): the other way round
                const uint32_t got = (uint32_t)__shfl_xor((int)(kh ? hx : hy), 32, 64);
                const uint32_t half0 = kh ? got : hx, half1 = kh ? hy : got;
                const uint32_t word = (spread(half0) << 4) | spread(half1);
                cnt += (uint32_t)__popc(word);
                gstore_u32(mwords, ((uint32_t)rt * nch + (uint32_t)k) * 64u + lane, word);
            }
        }
    }
    if (ntodo > TODO_CAP) return false;       // (a slot full of pairs at the cutoff: the vector path, which rewrites every word)
    if (ntodo) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the words are in memory before single bits are OR-ed into them
        __builtin_amdgcn_wave_barrier();
        for (uint32_t q0 = 0; q0 < ntodo; q0 += 64u) {
            bool hx = false;
            uint32_t rl = 0u, col = 0u;
            if (q0 + lane < ntodo) {
                const uint32_t e = todo[q0 + lane];
                rl = e >> 16;
                col = e & 0xFFFFu;
                if (rl < nlive && col < T.n2) {
                    const float4 b = gload4(P.sb, T.b0 + col);
                    const float4 p = lload4(la, rl);
                    hx = wrapped_d2_exact(P, T.wrap, b.x - p.x, b.y - p.y, b.z - p.z) <= cutoff2;      // p2 - p1 (:485-486)
                }
            }
            if (hx) atomicOr(mwords + ((rl >> 5) * nch + (col >> 6)) * 64u + (col & 63u), 0x80000000u >> (rl & 31u));
            const unsigned long long mx = __builtin_amdgcn_ballot_w64(hx);
            if (lane == 0) cnt += (uint32_t)__popcll(mx);
        }
    }
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
    count_out = cnt;
    return true;
}

template <int KIND, bool FILL, bool WRAPPED, int NCH, bool TRI, bool MASKED>
__device__ __forceinline__ uint32_t run_fast(const SearchParams &P, const Task &T, uint32_t i0, Fifo &F, float4 *la,
                                             uint32_t lane, uint32_t *mwords) {
    static_assert(!(WRAPPED && TRI), "a triangular (same-cell) entry never wraps on the fast path");
    if constexpr (!FILL && !WRAPPED && MASKED_COUNT_SORTED) return run_count_sorted<KIND, NCH, TRI>(P, T, i0, la, lane);
    // approximate classification allowed?  (needs n_d = round(f_d) = +-1 for every wrapped pair: >= 4 cells
    // per periodic dimension; the corner entries of triclinic boxes run the candidate loop: always exact)
    const bool approx = WRAPPED && P.approx_wrapped != 0u && !(P.box.nshift != 0 && T.wrap == MOLAR_HIP_PBC_FULL);
    float Sx = 0.f, Sy = 0.f, Sz = 0.f;      // b + S is the image of the second cell next to the first one
    if (WRAPPED && approx) {
        for (int d = 0; d < 3; ++d) {
            if (!((T.wrap >> d) & 1u)) continue;
            const float sgn = ((T.wrap_b >> d) & 1u) ? 1.0f : -1.0f;   // second cell wrapped: +col, first cell: -col
            Sx += sgn * P.box.m[3 * d];
            Sy += sgn * P.box.m[3 * d + 1];
            Sz += sgn * P.box.m[3 * d + 2];
        }
    }
    constexpr bool REPLAY = FILL && MASKED;   // the fill pass replays the hit bits of the count pass: no distances
    float bx[NCH], by[NCH], bz[NCH];
    uint32_t bid[NCH];
    if (!REPLAY) {
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const uint32_t jj = (uint32_t)k * 64u + lane;
            float4 q = make_float4(3.0e38f, 3.0e38f, 3.0e38f, 0.f);
            if (jj < T.n2) {
                q = gload4(P.sb, T.b0 + jj);
                if (WRAPPED) { q.x += Sx; q.y += Sy; q.z += Sz; }       // S == 0 when the slot is evaluated exactly
            }
            bx[k] = q.x; by[k] = q.y; bz[k] = q.z; bid[k] = __float_as_uint(q.w);
        }
    }
    const float cutoff2 = P.cutoff2;
    const float band_lo = P.band_lo, band_hi = P.band_hi;
    const uint32_t rows = __builtin_amdgcn_readfirstlane(T.n1 - i0 < T.rps ? T.n1 - i0 : T.rps);
    unsigned long long live;   // rows of this slot that can have a hit at all (same value in both passes)
    {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane < rows) a = gload4(P.sa, T.a0 + i0 + lane);
        la[lane] = a;
        const float4 lo = gload4(P.aabb_b, 2 * T.cb), hi = gload4(P.aabb_b, 2 * T.cb + 1);
        bool need = true;
        if (TRI) {
            // same cell (i < j triangle, :439-451): the atom lies inside its own cell's box, nothing to prune
        } else if (!WRAPPED) {
            // Exact row pruning.  Every B position lies inside the cell's bounding box [lo,hi], and each
            // f32 operation of d2 = ((dx*dx)+(dy*dy))+(dz*dz) is monotone in |dx|,|dy|,|dz|, so the same
            // expression on the box distances is a lower bound of every d2 of the row IN f32 ARITHMETIC:
            // if it already exceeds cutoff2 the reference finds no hit in this row either.
            need = !(aabb_d2(a.x, a.y, a.z, lo.x, lo.y, lo.z, hi.x, hi.y, hi.z) > cutoff2);
        } else if (approx) {
            // Conservative pruning against the image box [lo+S, hi+S] with a 1e-3 nm margin (>100x the f32
            // evaluation error of the reference's inv*v / M*f at MD box sizes).
            need = !(aabb_d2(a.x - Sx, a.y - Sy, a.z - Sz, lo.x, lo.y, lo.z, hi.x, hi.y, hi.z) > P.prune_limit2);
        }
        live = __builtin_amdgcn_ballot_w64(lane < rows && need);
    }
    __builtin_amdgcn_wave_barrier();
    uint32_t total = 0;

    if (REPLAY) {
        if (!mwords) return 0u;                // the bits of this slot were not recorded (buffer too small)
        // ---- fill pass over recorded hit bits.  Word (g, k) of lane l holds, MSB first, the hit bits of
        // (live row, atom k*64+l) for the g-th group of 32 live rows.  A queued hit is (row, sorted position);
        // ids and the exact d2 are rebuilt densely at flush time (fifo_hit).
        F.recompute = WRAPPED ? 1u : 2u;
        F.la = la;
        F.wrap = T.wrap;
        F.fq = F.fq_store;
        float4 q[NCH];                         // second-cell atoms of this lane, unshifted
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const uint32_t jj = (uint32_t)k * 64u + lane;
            q[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (jj < T.n2) q[k] = gload4(P.sb, T.b0 + jj);
        }
        uint32_t w[NCH];
        uint32_t nrow = 0;
        while (live) {
            if ((nrow & 31u) == 0u) {
#pragma unroll
                for (int k = 0; k < NCH; ++k) w[k] = gload_u32(mwords, ((nrow >> 5) * NCH + k) * 64u + lane);
            }
            ++nrow;
            const uint32_t r = (uint32_t)__builtin_ctzll(live);
            live &= ~(1ull << r);
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                const bool hit = (int32_t)w[k] < 0;
                w[k] += w[k];
                const unsigned long long mask = __builtin_amdgcn_ballot_w64(hit);
                if (mask) {
                    const uint32_t cnt = (uint32_t)__popcll(mask);
                    if (hit) {
                        const uint32_t s = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                                     __builtin_amdgcn_mbcnt_lo((uint32_t)mask, F.tail)) &
                                           (FIFO_CAP - 1);
                        F.fq[s] = q[k];
                        F.fd[s] = r;
                    }
                    F.tail += cnt;
                    total += cnt;
                    if (F.tail - F.head >= 64u) fifo_drain<KIND>(P, F, lane);
                }
            }
        }
        if (F.tail != F.head) {
            __builtin_amdgcn_wave_barrier();
            fifo_flush<KIND>(P, F, F.tail - F.head, lane);
        }
        F.recompute = 0u;
        F.fq = nullptr;
        return total;
    }

    // plain and same-cell entries of the fill pass: queue of FIFO_WIDE, 128 entries per flush (fifo_flush_wide)
#ifdef MOLAR_HIP_WIDE_FLUSH
    constexpr bool WIDE = FILL && !WRAPPED && KIND != MOLAR_HIP_SEARCH_WITHIN;
#else
    constexpr bool WIDE = false;
#endif
    if (FILL) {
        F.recompute = (WRAPPED && approx) ? 1u : 0u;
        F.la = la;
        F.wrap = T.wrap;
        if (WIDE && !F.hist) F.quota = 128u - ((uint32_t)F.base & 127u);
    }
    uint32_t acc = 0;       // count pass.  !MASKED: per-lane hit counter (wrapped+approx: hits that are certain)
    uint32_t acc_hi = 0;    //              wrapped+approx: candidates at or below the upper edge of the band
    uint32_t hw[NCH];       // MASKED: per chunk, the hit bits of the rows done so far
    uint32_t su = 0, mu = 0;   // MASKED wrapped+approx: per-row summaries, one bit per chunk (certain / at or below the band)
    uint32_t nrow = 0;
    if (MASKED) {
#pragma unroll
        for (int k = 0; k < NCH; ++k) hw[k] = 0u;
    }
    // store the history words of a finished group (n rows, left-aligned) and count its hits
    auto store_group = [&](uint32_t g, uint32_t n) {
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            acc += (uint32_t)__popc(hw[k]);
            if (mwords) gstore_u32(mwords, (g * NCH + k) * 64u + lane, hw[k] << (32u - n));
            hw[k] = 0u;
        }
    };
    // exact d2 of (row atom p, atom jj of the second cell), second cell re-read unshifted
    auto exact_d2 = [&](const float4 &p, uint32_t jj) -> float {
        float4 q = make_float4(3.0e38f, 3.0e38f, 3.0e38f, 0.f);
        if (jj < T.n2) q = gload4(P.sb, T.b0 + jj);
        return wrapped_d2_exact(P, T.wrap, q.x - p.x, q.y - p.y, q.z - p.z);
    };
    while (live) {
        const uint32_t r = (uint32_t)__builtin_ctzll(live);
        live &= ~(1ull << r);
        const float4 p = lload4(la, r);              // one broadcast ds_read per row
        const uint32_t id_i = __float_as_uint(p.w);
        const uint32_t i = i0 + r;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            if (TRI && (uint32_t)k * 64u + 63u <= i) {                           // whole chunk has j <= i
                if (MASKED && !FILL) hw[k] += hw[k];                             // ... its bit of this row is 0
                continue;
            }
            const uint32_t jj = (uint32_t)k * 64u + lane;
            const float dx = bx[k] - p.x, dy = by[k] - p.y, dz = bz[k] - p.z;     // p2 - p1 (image of p2 if WRAPPED)
            float d2 = (dx * dx + dy * dy) + dz * dz;                            // |p2-p1|^2 (:446, :460)
            if (WRAPPED && !approx) d2 = wrapped_d2_exact<FILL && !MASKED>(P, T.wrap, dx, dy, dz); // S == 0: dx is the raw difference (:485-486)
            if (TRI && (uint32_t)k * 64u <= i) d2 = (jj > i) ? d2 : INFINITY;    // diagonal chunk: j in i+1..n (:443)
            if (!FILL) {
                if (MASKED) {
                    // h = 2*h + (d2 <= cutoff2): the compare's carry shifts into the per-lane history, VALU only
                    if (WRAPPED && approx) {
                        // sure = d2 < band_lo goes into the chunk's history AND into the row summary `su`;
                        // maybe = d2 <= band_hi into the row summary `mu`: the row has an undecided candidate
                        // iff the two summaries differ in their low NCH bits
                        unsigned long long scr;
                        asm volatile("v_cmp_gt_f32 vcc, %4, %3\n\t"
                                     "v_addc_co_u32 %0, %2, %0, %0, vcc\n\t"
                                     "v_addc_co_u32 %1, %2, %1, %1, vcc"
                                     : "+v"(hw[k]), "+v"(su), "=&s"(scr) : "v"(d2), "s"(band_lo) : "vcc");
                        asm volatile("v_cmp_ge_f32 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(mu) : "v"(d2), "s"(band_hi) : "vcc");
                    } else {
                        asm volatile("v_cmp_ge_f32 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(hw[k]) : "v"(d2), "s"(cutoff2) : "vcc");
                    }
                } else if (WRAPPED && approx) {
                    asm volatile("v_cmp_gt_f32 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(acc) : "v"(d2), "s"(band_lo) : "vcc");
                    asm volatile("v_cmp_ge_f32 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(acc_hi) : "v"(d2), "s"(band_hi) : "vcc");
                } else {
                    // acc += (d2 <= cutoff2): the compare's carry is added per lane, no SALU involved
                    asm volatile("v_cmp_ge_f32 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(acc) : "v"(d2), "s"(cutoff2) : "vcc");
                }
            } else {
                bool hit;
                uint32_t third = __float_as_uint(d2);
                if (WRAPPED && approx) {
                    const bool sure = d2 < band_lo, maybe = d2 <= band_hi;
                    hit = sure;
                    if (__builtin_amdgcn_ballot_w64(maybe && !sure)) hit = sure || (maybe && exact_d2(p, jj) <= cutoff2);
                    third = (r << 26) | (T.b0 + jj);          // exact d2 is recomputed at flush time
                } else {
                    hit = d2 <= cutoff2;
                }
                const unsigned long long mask = __builtin_amdgcn_ballot_w64(hit);
                if (mask) {
                    const uint32_t cnt = (uint32_t)__popcll(mask);
                    if (hit) {
                        // FIFO slot = tail + rank among the hit lanes (mbcnt accumulates onto tail)
                        // (rank first, tail added with the shift: v_add_lshl_u32 takes the SGPR, no v_mov of the tail)
                        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                        const uint32_t off = ((rank + F.tail) << 2) & (((WIDE ? FIFO_WIDE : FIFO_CAP) - 1u) << 2);
                        typedef __attribute__((address_space(3))) uint32_t lds_u32;
                        *(lds_u32 *)((__attribute__((address_space(3))) char *)F.fi + off) = id_i;
                        *(lds_u32 *)((__attribute__((address_space(3))) char *)F.fj + off) = bid[k];
                        *(lds_u32 *)((__attribute__((address_space(3))) char *)F.fd + off) = third;
                    }
                    F.tail += cnt;
                    total += cnt;
                    if (WIDE) {
                        if (F.tail - F.head >= 128u) fifo_drain_wide<KIND>(P, F, lane);
                    } else if (F.tail - F.head >= 64u) fifo_drain<KIND>(P, F, lane);
                }
            }
        }
        if (!FILL && WRAPPED && approx) {
            // some candidate of this row fell inside the band: settle those (and only those) exactly
            bool any_amb;
            if (MASKED) {
                any_amb = __builtin_amdgcn_ballot_w64(((su ^ mu) & ((1u << NCH) - 1u)) != 0u) != 0ull;
            } else {
                any_amb = __builtin_amdgcn_ballot_w64(acc != acc_hi) != 0ull;
            }
            if (any_amb) {
#pragma unroll
                for (int k = 0; k < NCH; ++k) {
                    const float dx = bx[k] - p.x, dy = by[k] - p.y, dz = bz[k] - p.z;
                    const float d2 = (dx * dx + dy * dy) + dz * dz;
                    const bool amb = d2 <= band_hi && !(d2 < band_lo);
                    if (__builtin_amdgcn_ballot_w64(amb)) {
                        if (amb && exact_d2(p, (uint32_t)k * 64u + lane) <= cutoff2) {
                            if (MASKED) hw[k] |= 1u;
                            else acc += 1u;
                        }
                    }
                }
                acc_hi = acc;
            }
        }
        if (!FILL && MASKED) {
            if ((++nrow & 31u) == 0u) store_group((nrow >> 5) - 1u, 32u);
        }
    }
    if (!FILL) {
        if (MASKED && (nrow & 31u) != 0u) store_group(nrow >> 5, nrow & 31u);
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        total = acc;
    } else if (F.tail != F.head) {
        __builtin_amdgcn_wave_barrier();
        if (WIDE) {                      // < 128 entries left (or a slot of fewer than 128 altogether): rounds of 64
            while (F.tail != F.head) {
                const uint32_t c = F.tail - F.head < 64u ? F.tail - F.head : 64u;
                fifo_flush<KIND, FIFO_WIDE>(P, F, c, lane);
            }
        } else {
            fifo_flush<KIND>(P, F, F.tail - F.head, lane);
        }
    }
    if (FILL) F.recompute = 0u;
    return total;
}

// chunk-count dispatch: for the single-set search the non-triangular tasks get a fully unrolled,
// branch-free row body per chunk count; everything else checks the chunk count at run time
// KMAX: chunks of the second cell a slot may keep in registers.  KREG for the regular instances (64 VGPRs at 8 waves per SIMD);
// the instances for frames of large cells (pair_kernel<KIND, MODE, WPE = 4>: 128 VGPRs) take cells of up to KMAX_WIDE * 64 atoms
// through the same row loops - plain, same-cell and, evaluated in both passes (no hit history: the plan records none beyond
// KREG chunks), band-classified wrapped entries - instead of streaming the second cell from memory for every row.
constexpr int KMAX_WIDE = 16;
constexpr int KMAX_HUGE = 32;      // ... of the instances with 3 (or 2) waves per SIMD: 168 (256) registers per lane
template <int KIND, bool FILL, int WK, bool MASKED, int KMAX = KREG>
__device__ __forceinline__ uint32_t run_task_nch(const SearchParams &P, const Task &T, uint32_t i0, Fifo &F, float4 *la,
                                                 uint32_t lane, uint32_t *mwords) {
    const uint32_t nchunks = (T.n2 + 63u) >> 6;
    if constexpr (KMAX > KMAX_WIDE) {
        // the 168-register instances (pair_kernel<KIND, MODE, WPE = 3>: frames whose cells hold more than 1024 atoms as a rule): up to
        // KMAX_HUGE chunks (2048 atoms) of the second cell resident, the chunk count rounded up to 16 / 20 / 24 / 28 / 32 (a chunk past
        // the end of the cell holds "never a hit" coordinates: wasted lanes, no wrong results) - five sizes instead of 24 keep the
        // build within minutes
        if ((KIND == MOLAR_HIP_SEARCH_SINGLE || KIND == MOLAR_HIP_SEARCH_DOUBLE) && nchunks > (uint32_t)KREG && nchunks <= (uint32_t)KMAX) {
            constexpr bool WR = WK != WK_NONE;
            const bool tri = KIND == MOLAR_HIP_SEARCH_SINGLE && WK == WK_NONE && T.tri;
            if (!T.tri || tri) {
#define MH_HUGE_CASE(N)                                                                                              \
    if (nchunks <= (uint32_t)(N)) {                                                                                  \
        if (tri) {                                                                                                   \
            if constexpr (KIND == MOLAR_HIP_SEARCH_SINGLE && WK == WK_NONE)                                          \
                return run_fast<KIND, FILL, false, N, true, false>(P, T, i0, F, la, lane, nullptr);                  \
        }                                                                                                            \
        return run_fast<KIND, FILL, WR, N, false, false>(P, T, i0, F, la, lane, nullptr);                            \
    }
                MH_HUGE_CASE(16) MH_HUGE_CASE(20) MH_HUGE_CASE(24) MH_HUGE_CASE(28) MH_HUGE_CASE(32)
#undef MH_HUGE_CASE
            }
        }
    } else if constexpr (KMAX > KREG) {
        if ((KIND == MOLAR_HIP_SEARCH_SINGLE || KIND == MOLAR_HIP_SEARCH_DOUBLE) && nchunks > (uint32_t)KREG && nchunks <= (uint32_t)KMAX) {
            constexpr bool WR = WK != WK_NONE;
            const bool tri = KIND == MOLAR_HIP_SEARCH_SINGLE && WK == WK_NONE && T.tri;
            if (!T.tri || tri) {
#define MH_WIDE_CASE(N)                                                                                              \
    case N:                                                                                                          \
        if (tri) {                                                                                                   \
            if constexpr (KIND == MOLAR_HIP_SEARCH_SINGLE && WK == WK_NONE)                                          \
                return run_fast<KIND, FILL, false, N, true, false>(P, T, i0, F, la, lane, nullptr);                  \
        }                                                                                                            \
        return run_fast<KIND, FILL, WR, N, false, false>(P, T, i0, F, la, lane, nullptr);
                switch (nchunks) {
                    MH_WIDE_CASE(9) MH_WIDE_CASE(10) MH_WIDE_CASE(11) MH_WIDE_CASE(12)
                    MH_WIDE_CASE(13) MH_WIDE_CASE(14) MH_WIDE_CASE(15)
                    default: break;
                }
                if (tri) {
                    if constexpr (KIND == MOLAR_HIP_SEARCH_SINGLE && WK == WK_NONE)
                        return run_fast<KIND, FILL, false, 16, true, false>(P, T, i0, F, la, lane, nullptr);
                }
                return run_fast<KIND, FILL, WR, 16, false, false>(P, T, i0, F, la, lane, nullptr);
#undef MH_WIDE_CASE
            }
        }
    }
    if constexpr (!FILL && WK == WK_NONE && (KIND == MOLAR_HIP_SEARCH_SINGLE || KIND == MOLAR_HIP_SEARCH_DOUBLE)) {
        // plain and same-cell entries with a second cell of <= 320 atoms: the count goes to the matrix cores unless the
        // slot's error bound is too wide
        if ((P.mfma_count & 1u) && F.lh && T.n2 <= 32u * (uint32_t)MFMA_TILES) {
            bool done = true;
            const uint32_t cm = (KIND == MOLAR_HIP_SEARCH_SINGLE && T.tri) ? run_count_mfma<KIND, true>(P, T, i0, la, F.lh, lane, done)
                                                                          : run_count_mfma<KIND, false>(P, T, i0, la, F.lh, lane, done);
            if (done) return cm;
        }
    }
    if constexpr (!FILL && MASKED && WK != WK_NONE && (KIND == MOLAR_HIP_SEARCH_SINGLE || KIND == MOLAR_HIP_SEARCH_DOUBLE)) {
        // band-classified wrapped entries with a second cell of <= 320 atoms: matrix-core count that writes the hit history
        // the fill pass replays (same live rows, same format); declined slots fall through to the vector path
        if ((P.mfma_count & 2u) && F.lh && mwords && !T.tri && T.rps == 64u && T.n2 <= 32u * (uint32_t)MFMA_TILES && P.approx_wrapped != 0u &&
            !(P.box.nshift != 0 && T.wrap == MOLAR_HIP_PBC_FULL)) {
            uint32_t cw = 0u;
            if (run_count_mfma_wrapped<KIND>(P, T, i0, la, F.lh, lane, mwords, cw)) return cw;
            __builtin_amdgcn_wave_barrier();
        }
    }
    if ((KIND == MOLAR_HIP_SEARCH_SINGLE || KIND == MOLAR_HIP_SEARCH_DOUBLE) && !T.tri && nchunks <= (uint32_t)KREG) {
        constexpr bool WR = WK != WK_NONE;
        switch (nchunks) {
            case 1: return run_fast<KIND, FILL, WR, 1, false, MASKED && WR>(P, T, i0, F, la, lane, mwords);
            case 2: return run_fast<KIND, FILL, WR, 2, false, MASKED && WR>(P, T, i0, F, la, lane, mwords);
            case 3: return run_fast<KIND, FILL, WR, 3, false, MASKED && WR>(P, T, i0, F, la, lane, mwords);
            case 4: return run_fast<KIND, FILL, WR, 4, false, MASKED && WR>(P, T, i0, F, la, lane, mwords);
            case 5: return run_fast<KIND, FILL, WR, 5, false, MASKED && WR>(P, T, i0, F, la, lane, mwords);
            case 6: return run_fast<KIND, FILL, WR, 6, false, MASKED && WR>(P, T, i0, F, la, lane, mwords);
            case 7: return run_fast<KIND, FILL, WR, 7, false, MASKED && WR>(P, T, i0, F, la, lane, mwords);
            default: return run_fast<KIND, FILL, WR, 8, false, MASKED && WR>(P, T, i0, F, la, lane, mwords);
        }
    }
    if (KIND == MOLAR_HIP_SEARCH_SINGLE && WK == WK_NONE && T.tri && nchunks <= (uint32_t)KREG) {
        switch (nchunks) {
            case 1: return run_fast<KIND, FILL, false, 1, true, false>(P, T, i0, F, la, lane, mwords);
            case 2: return run_fast<KIND, FILL, false, 2, true, false>(P, T, i0, F, la, lane, mwords);
            case 3: return run_fast<KIND, FILL, false, 3, true, false>(P, T, i0, F, la, lane, mwords);
            case 4: return run_fast<KIND, FILL, false, 4, true, false>(P, T, i0, F, la, lane, mwords);
            case 5: return run_fast<KIND, FILL, false, 5, true, false>(P, T, i0, F, la, lane, mwords);
            case 6: return run_fast<KIND, FILL, false, 6, true, false>(P, T, i0, F, la, lane, mwords);
            case 7: return run_fast<KIND, FILL, false, 7, true, false>(P, T, i0, F, la, lane, mwords);
            default: return run_fast<KIND, FILL, false, 8, true, false>(P, T, i0, F, la, lane, mwords);
        }
    }
    if (nchunks > (uint32_t)KREG) {
        if (KIND == MOLAR_HIP_SEARCH_SINGLE && T.tri) return run_task<KIND, FILL, WK, true, 0, false, FILL && !MASKED>(P, T, i0, F, lane);
        return run_task<KIND, FILL, WK, false, 0, false, FILL && !MASKED>(P, T, i0, F, lane);
    }
    if (KIND == MOLAR_HIP_SEARCH_SINGLE && T.tri) return run_task<KIND, FILL, WK, true, KREG, true, FILL && !MASKED>(P, T, i0, F, lane);
    return run_task<KIND, FILL, WK, false, KREG, true, FILL && !MASKED>(P, T, i0, F, lane);
}

// Slots.  A plan entry ("task") is cut into blocks of 64 rows of its first cell; one wave processes
// one slot.  This bounds the work of a wave (the corner entries that run the triclinic candidate
// loop are ~50x a plain entry) and gives small systems enough waves to fill the chip.  Slots are
// numbered in plan order, then row order, so an exclusive scan of the per-slot counts is the
// reference's output order.
template <int KIND>
__global__ void __launch_bounds__(256) plan_kernel(SearchParams P, uint32_t *__restrict__ task_nb,
                                                   TaskDesc *__restrict__ task_desc, uint32_t *__restrict__ task_mu,
                                                   uint32_t fast_kind, uint32_t *__restrict__ slot_cnt, uint64_t nslot_cnt,
                                                   unsigned long long *__restrict__ scan_state, uint64_t nstate,
                                                   SearchParams *__restrict__ params_dst) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < nslot_cnt) slot_cnt[t] = 0u;     // the count kernel writes the slots that exist; the scan runs over the bound
    if (t < nstate) scan_state[t] = 0ull;    // ticket + tile descriptors of this search's look-back scans
    if (params_dst && blockIdx.x == 0) {
        // the parameter block of the pair kernels, straight from this kernel's own argument segment (P is its first
        // argument): one launch less in front of the count pass than a kernel of its own
        const uint32_t *src = (const uint32_t *)__builtin_amdgcn_kernarg_segment_ptr();
        uint32_t *d = (uint32_t *)params_dst;
        for (uint32_t w = threadIdx.x; w < (uint32_t)(sizeof(SearchParams) / 4u); w += blockDim.x) d[w] = src[w];
    }
    if (t == P.ntasks) {                 // terminators of the two exclusive scans (the grid covers ntasks + 1)
        task_nb[t] = 0u;
        task_mu[t] = 0u;
    }
    if (t >= P.ntasks) return;
    const Task T = decode_task<KIND, false>(P, t);
    const uint32_t nb = T.valid ? (T.n1 + T.rps - 1u) / T.rps : 0u;
    task_nb[t] = nb;
    // hit-history units (64 words) of the task's slots: two 32-row groups x nch chunks per slot, for the WRAPPED
    // tasks run_task_nch sends down the register-resident fast path (same condition as there).  Recording the hit
    // bits in the count pass and replaying them in the fill pass pays where a candidate is expensive (the wrapped
    // distance); for plain and same-cell entries re-evaluating d2 is cheaper than the replay (measured).
    const uint32_t nch = (T.n2 + 63u) >> 6;
    task_mu[t] = (fast_kind && P.use_box && T.wrap != 0u && nch <= (uint32_t)KREG) ? nb * 2u * nch : 0u;
    TaskDesc d;
    d.a0 = T.a0; d.n1 = T.n1; d.b0 = T.b0; d.n2 = T.n2; d.cb = T.cb;
    d.flags = T.wrap | (T.tri ? 0x100u : 0u) | (T.valid ? 0x200u : 0u) | (T.wrap_b << 12) | (T.rps << 16);
    d.pad0 = d.pad1 = 0;
    task_desc[t] = d;
}

static __global__ void __launch_bounds__(256) slotmap_kernel(uint64_t ntasks, const uint32_t *__restrict__ task_first,
                                                      const TaskDesc *__restrict__ task_desc,
                                                      const unsigned long long *__restrict__ task_moff,   // NULL: no hit history
                                                      SlotDesc *__restrict__ slot_desc, uint64_t nslots_bound,
                                                      unsigned long long *__restrict__ sizes_host) {   // pinned, or NULL
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0 && sizes_host) {
        sizes_host[1] = task_moff ? task_moff[ntasks] : 0ull;   // hit-history units of this search
        sizes_host[2] = task_first[ntasks];                     // its slots
    }
    if (t >= ntasks) {
        // slots between the real count and the host's bound: waves launched for them leave at once
        const uint64_t s = (uint64_t)task_first[ntasks] + (t - ntasks);
        if (s <= nslots_bound) {
            SlotDesc z;
            z.a0 = z.n1 = z.b0 = z.n2 = z.cb = z.flags = z.i0 = z.frame = 0u;
            z.moff = ~0ull >> 1;
            z.pad1 = 0ull;
            slot_desc[s] = z;
        }
        return;
    }
    const uint32_t s0 = task_first[t], s1 = task_first[t + 1];
    if (s1 == s0) return;
    const TaskDesc d = task_desc[t];
    const uint32_t rps = d.flags >> 16, nch = (d.n2 + 63u) >> 6;
    const unsigned long long m0 = task_moff ? task_moff[t] : (~0ull >> 1);
    for (uint32_t s = s0; s < s1; ++s) {
        SlotDesc o;
        o.a0 = d.a0; o.n1 = d.n1; o.b0 = d.b0; o.n2 = d.n2;
        o.cb = d.cb; o.flags = d.flags; o.i0 = (s - s0) * rps; o.frame = 0u;
        o.moff = task_moff ? m0 + (unsigned long long)(s - s0) * 2u * nch : m0;
        o.pad1 = 0ull;
        slot_desc[s] = o;
    }
}

// fused histogram: the slots hist_kernel (below) takes; pair_kernel<MODE_HIST> leaves them alone
template <int KIND>
__device__ __forceinline__ bool hist_lean_slot(const SearchParams &P, uint32_t flags, uint32_t n2) {
    if (KIND != MOLAR_HIP_SEARCH_SINGLE && KIND != MOLAR_HIP_SEARCH_DOUBLE) return false;
    // second cells of more than KREG * 64 atoms: hist_kernel<KIND, BIG> walks them in blocks of that size where such cells are the
    // rule (hist_big); the odd oversized cell of an ordinary frame stays with the generic kernel
    if (((n2 + 63u) >> 6) > (uint32_t)KREG && !P.hist_big) return false;
    const uint32_t wrap = flags & 7u;
    if (!(P.use_box && wrap != 0u)) return true;                    // plain or same-cell entry
    if (flags & 0x100u) return false;                               // a cell paired with its own periodic image
    return P.approx_wrapped != 0u && !(P.box.nshift != 0 && wrap == MOLAR_HIP_PBC_FULL);
}

// WPE: waves per SIMD of a second instance of the count / fill kernels for frames whose cells hold more than 448 atoms on
// average (launch_pair_wide, pair_k5.hip): with 128 registers per lane a slot keeps up to 16 chunks (1024 atoms) of the second
// cell resident (run_task_nch, KMAX) - the regular instances stream cells above 512 atoms from memory for every row (1M atoms
// in the sheared box at rc 1.6 nm, 636 atoms per cell: 65 M pairs per ms against 220-240 up to 1.4 nm).  0: the usual budget.
template <int KIND, int MODE, int WPE = 0>
__global__ void __launch_bounds__(64 * waves_per_block(MODE))
__attribute__((amdgpu_waves_per_eu(WPE ? WPE : (MODE == MODE_HIST ? 4 : (MODE == MODE_COUNT ? MH_COUNT_WPE : MH_FILL_WPE))))) pair_kernel(const SearchParams *__restrict__ Pp,
                                                     const SlotDesc *__restrict__ slot_desc,
                                                     const uint32_t nslots_arg,  // the host's bound: slots past the real count are empty
                                                     uint32_t *__restrict__ slot_cnt,
                                                     const unsigned long long *__restrict__ slot_base,
                                                     uint2 *__restrict__ out_pairs, float *__restrict__ out_dist,
                                                     uint32_t *__restrict__ out_ids) {
    constexpr int WAVES_PER_BLOCK = waves_per_block(MODE), BLOCK = 64 * WAVES_PER_BLOCK;
    // fill pass: three planes of FIFO_WIDE entries; the replay queue of the wrapped entries (FIFO_CAP x float4: the hits' second
    // atoms, with the row numbers in plane 2) lies over planes 0 and 1, which a replaying slot does not use
    constexpr int PLANE = MODE != MODE_COUNT ? FIFO_WIDE : FIFO_CAP;
    static_assert(FIFO_CAP * 16 <= 2 * FIFO_WIDE * 4, "the replay queue fits in two planes");
    __shared__ __attribute__((aligned(16))) uint32_t lds[WAVES_PER_BLOCK][3][PLANE];
    __shared__ float4 lds_a[WAVES_PER_BLOCK][64];
    __shared__ uint4 lds_h[MODE == MODE_COUNT ? WAVES_PER_BLOCK : 1][MODE == MODE_COUNT ? 128 : 1];       // matrix-core row records of the count pass
    constexpr bool FILL = MODE != MODE_COUNT;
    extern __shared__ uint32_t lds_hist[];     // histogram mode only (hist_nbins counters)
    // The parameter block lives in device memory: a by-value struct this large, indexed dynamically
    // (box.shifts[k]), gets copied to scratch by the compiler and drags every field into VGPRs.
    // (histogram mode over several frames: Pp is an array, one block per frame; the slot record says which.  The histogram's own
    // fields - bins, range, list counter - are the same in all of them and are read from the first.)
    const SearchParams &P0 = *Pp;
#ifdef MH_SLOT_LOOP
    uint32_t lane_var = threadIdx.x & 63u;
    const uint32_t &lane = lane_var;
#else
    const uint32_t lane = threadIdx.x & 63u;
#endif
    uint32_t nslots = nslots_arg;
    if (MODE == MODE_HIST && P0.hist_nslots) {        // its own list, written by hist_plan_kernel: the count sits in memory
        const uint32_t real = __builtin_amdgcn_readfirstlane(P0.hist_nslots[0]);
        nslots = real < nslots_arg ? real : nslots_arg;
    }
    // one-wave workgroups (count / fill): the wave index is the constant 0, so every LDS address is an immediate
    const uint32_t wave = WAVES_PER_BLOCK == 1 ? 0u : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr bool hist = MODE == MODE_HIST;
    if (hist) {
        for (uint32_t b = threadIdx.x; b < P0.hist_nbins; b += BLOCK) lds_hist[b] = 0u;
        __syncthreads();
    }
    unsigned long long wave_total = 0;
    auto process_slot = [&](uint32_t w) __attribute__((always_inline)) {
        // Blocks are handed out in launch order: walk the plan BACKWARDS so the cells at the far x edge,
        // whose entries wrap (several times the arithmetic per candidate), start first and the cheap
        // entries fill the tail; consecutive blocks land on different XCDs, which spreads that band
        // over the whole chip.
        // Workgroup b runs on XCD b % 8, each with its own L2.  Dealing the slots out one by one gives every XCD every
        // cell; in runs of XCD_RUN consecutive slots an XCD meets a second cell's records (and a first cell's rows) again
        // while they are in its L2, and the heavy band at the far x edge is still spread over all eight.  Count pass
        // 0.53 -> 0.50 ms; runs of 16 / 32 / 64: less, 256 / 512 / 1024: no gain, 8192: the XCDs finish far apart (-7 %).
        uint32_t wr = w;
        {
            const uint32_t x = w & 7u, q = w >> 3;
            const uint32_t full = (nslots / (8u * XCD_RUN)) * (8u * XCD_RUN);      // the part that divides evenly; the tail keeps its order
            if (w < full) wr = ((q / XCD_RUN) * 8u + x) * XCD_RUN + (q % XCD_RUN);
        }
        const uint32_t slot = nslots - 1u - wr;
        Task T;   // record prepared by slotmap_kernel: one dependent load between the kernel arguments and the atoms
        uint32_t i0;
        unsigned long long moff;
        uint32_t frame = 0u;
        {
            const uint4 lo = reinterpret_cast<const uint4 *>(slot_desc + slot)[0];
            const uint4 hi = reinterpret_cast<const uint4 *>(slot_desc + slot)[1];
            const uint2 mo = reinterpret_cast<const uint2 *>(slot_desc + slot)[4];
            const uint32_t fl = __builtin_amdgcn_readfirstlane(hi.y);
            if (!(fl & 0x200u)) return;       // past the last slot
            if (MODE == MODE_HIST) frame = __builtin_amdgcn_readfirstlane(hi.w);
            T.a0 = __builtin_amdgcn_readfirstlane(lo.x);
            T.n1 = __builtin_amdgcn_readfirstlane(lo.y);
            T.b0 = __builtin_amdgcn_readfirstlane(lo.z);
            T.n2 = __builtin_amdgcn_readfirstlane(lo.w);
            T.cb = __builtin_amdgcn_readfirstlane(hi.x);
            i0 = __builtin_amdgcn_readfirstlane(hi.z);
            moff = ((unsigned long long)__builtin_amdgcn_readfirstlane(mo.y) << 32) | __builtin_amdgcn_readfirstlane(mo.x);
            T.wrap = fl & 7u;
            T.tri = (fl & 0x100u) != 0u;
            T.valid = true;
            T.wrap_b = (fl >> 12) & 7u;
            T.rps = fl >> 16;
        }
        const SearchParams &P = MODE == MODE_HIST ? Pp[frame] : P0;
        Fifo F;
        F.fi = lds[wave][0];
        F.fj = lds[wave][1];
        F.fd = lds[wave][2];
        F.head = F.tail = 0;
        F.quota = 64u;
        F.has_pairs = out_pairs != nullptr;
        F.has_dist = out_dist != nullptr;
        F.pairs = nullptr;
        F.dist = nullptr;
        F.ids = nullptr;
        F.base = 0;
        F.room = 0xFFFFFFFFu;
        F.hist = hist ? lds_hist : nullptr;
        F.recompute = 0u;
        F.la = lds_a[wave];
        F.fq = nullptr;
        F.fq_store = MODE == MODE_FILL ? reinterpret_cast<float4 *>(lds[wave][0]) : nullptr;
        F.lh = MODE == MODE_COUNT ? lds_h[wave] : nullptr;
        F.wrap = 0;
        F.hmin = P.hist_min;
        F.hmax = P.hist_max;
        F.hn = (float)P.hist_nbins;
        if (FILL && !hist) {
            F.base = slot_base[slot];
            F.quota = 64u - ((uint32_t)F.base & 63u);
            const unsigned long long end = slot_base[slot + 1];
            if (end == F.base || end > P.out_cap) return;  // nothing to emit / no room (the host grows and repeats)
            F.room = end - F.base < 0xFFFFFFFFull ? (uint32_t)(end - F.base) : 0xFFFFFFFFu;
            if (out_pairs) F.pairs = out_pairs + F.base + lane;
            if (out_dist) F.dist = out_dist + F.base + lane;
            if (out_ids) F.ids = out_ids + F.base + lane;
        }
        uint32_t total = 0;
        const uint32_t wk = (P.use_box && T.wrap != 0) ? P.wrap_kind : (uint32_t)WK_NONE;
        if (hist && P.hist_lean && !P.hist_nslots && hist_lean_slot<KIND>(P, (T.tri ? 0x100u : 0u) | T.wrap, T.n2)) return;   // hist_kernel's
#ifdef MOLAR_HIP_DEBUG_KNOBS
        if (P.debug_skip) {     // not in release builds: tools/dbg_skip.sh builds with -DMOLAR_HIP_DEBUG_KNOBS
            const uint32_t kind_bit = T.tri ? 4u : (wk != WK_NONE ? (T.rps == 8u ? 8u : 2u) : 1u);   // 8: triclinic corner entries
            if (P.debug_skip & kind_bit) return;
        }
#endif
        // count/fill pair: the count pass records the hit bits of the fast-path slots, the fill pass replays them
        constexpr bool MASKED = MODE != MODE_HIST;
        uint32_t *mwords = nullptr;
        if (MASKED) {
            const uint32_t nch = (T.n2 + 63u) >> 6;
            if (moff + 2u * nch <= P.mask_cap_units) mwords = P.maskbuf + moff * 64u;
        }
        constexpr int KMAX = (WPE == 2 || WPE == 3) ? KMAX_HUGE : (WPE ? KMAX_WIDE : KREG);
        switch (wk) {
            case WK_NONE: total = run_task_nch<KIND, FILL, WK_NONE, MASKED, KMAX>(P, T, i0, F, lds_a[wave], lane, mwords); break;
            case WK_DIAG: total = run_task_nch<KIND, FILL, WK_DIAG, MASKED, KMAX>(P, T, i0, F, lds_a[wave], lane, mwords); break;
            case WK_UPPER: total = run_task_nch<KIND, FILL, WK_UPPER, MASKED, KMAX>(P, T, i0, F, lds_a[wave], lane, mwords); break;
            default: total = run_task_nch<KIND, FILL, WK_GENERAL, MASKED, KMAX>(P, T, i0, F, lds_a[wave], lane, mwords); break;
        }
        if (!FILL && lane == 0) slot_cnt[slot] = total;
        wave_total += total;
    };
    const uint32_t w0 = (blockIdx.y * gridDim.x + blockIdx.x) * WAVES_PER_BLOCK + wave;
    if (!hist) {
        // COUNT / FILL: one wave per slot (nothing is live across slots -> fewer registers, more waves)
        // (round 5: two / four consecutive slots per one-wave workgroup in the count pass - half / a quarter of the 2.9e5
        // launches - took 1.21 / 0.93 ms against 0.41: the slot body in a loop loses its register allocation; not kept)
#ifdef MH_SLOT_LOOP
        // (experiment: MH_SLOT_LOOP consecutive slots per wave)
#pragma unroll 1
        for (uint32_t g = 0; g < (uint32_t)MH_SLOT_LOOP; ++g) {
            const uint32_t w = w0 + g * (gridDim.x * gridDim.y * WAVES_PER_BLOCK);      // (strided: a workgroup's slots keep its XCD)
            if (w < nslots) process_slot(w);
            asm volatile("" : "+v"(lane_var));      // nothing derived from the lane id survives an iteration
        }
#else
        if (w0 < nslots) process_slot(w0);
#endif
    } else {
        // histogram mode: capped grid, strided slots, so each workgroup flushes its LDS histogram once
        for (uint32_t w = w0; w < nslots; w += gridDim.x * gridDim.y * WAVES_PER_BLOCK) process_slot(w);
    }
    if (hist) {
        __syncthreads();
        for (uint32_t b = threadIdx.x; b < P0.hist_nbins; b += BLOCK) {
            const uint32_t v = lds_hist[b];
            if (v) atomicAdd(&P0.hist_bins[b], (unsigned long long)v);
        }
        if (lane == 0 && wave_total && P0.hist_total) atomicAdd(P0.hist_total, wave_total);
    }
}


// HIP limits gridDim.x * blockDim.x to 2^32 threads: sparse giant grids (10^8 plan entries, one 64-lane workgroup per
// slot) spill into grid.y; the kernels linearise (x fastest)
inline dim3 pair_grid(unsigned nblocks) {
    const unsigned gx = nblocks < (1u << 24) ? (nblocks ? nblocks : 1u) : (1u << 24);
    return dim3(gx, (nblocks + gx - 1u) / gx);
}

// one launch of the pair kernel for a search kind / mode
template <int KIND, int MODE, int WPE = 0>
inline void launch_pair_kernel(unsigned nblocks, size_t dyn_lds, hipStream_t stream, const SearchParams *dP,
                               const SlotDesc *slot_desc, uint32_t nslots, uint32_t *slot_cnt,
                               const unsigned long long *slot_base, uint2 *pairs, float *dist, uint32_t *ids) {
    // (count / fill: `nblocks` counts slots = waves; the histogram mode passes workgroups)
    if (MODE != MODE_HIST) nblocks = (nblocks + (unsigned)waves_per_block(MODE) - 1u) / (unsigned)waves_per_block(MODE);
#ifdef MH_SLOT_LOOP
    if (MODE != MODE_HIST) nblocks = (nblocks + (unsigned)MH_SLOT_LOOP - 1u) / (unsigned)MH_SLOT_LOOP;
#endif
    hipLaunchKernelGGL((pair_kernel<KIND, MODE, WPE>), pair_grid(nblocks), dim3(64 * waves_per_block(MODE)), dyn_lds, stream, dP, slot_desc, nslots,
                       slot_cnt, slot_base, pairs, dist, ids);
}

}  // namespace pairk

// defined in pair_k0.hip .. pair_k3.hip (one search kind each)
// (pair_k4.hip, hist_kernels.hpp) the fused histogram of the fixed-cutoff kinds: the one-kernel plan (two slot lists: the lean
// kernel's and the generic kernel's) and the lean kernel.  queue: hist_queue_words() words, zero before the first launch;
// parity: alternates between consecutive frames of a context (which pair of list counters this frame uses)
// lslot: which of the four pairs of list counters this launch uses - consecutive launches of a context take consecutive ones, and
// a launch leaves pair (lslot + 2) & 3 zeroed for the launch after the next
void launch_hist_plan(int kind, hipStream_t stream, const pairk::SearchParams &P, pairk::SearchParams *params_dst, pairk::SlotDesc *lean,
                      pairk::SlotDesc *rest, uint32_t *queue, int lslot);
// the plans of `nframes` frames in one launch: parameter blocks params[0 .. nframes) already in device memory
void launch_hist_plan_frames(int kind, hipStream_t stream, const pairk::SearchParams *params, unsigned nframes, uint64_t ntasks_max,
                             pairk::SlotDesc *lean, pairk::SlotDesc *rest, uint32_t *queue, int lslot);
void launch_hist_lean(int kind, unsigned num_cus, size_t dyn_lds, hipStream_t stream, const pairk::SearchParams *dP,
                      const pairk::SlotDesc *slot_desc, uint32_t nslots_bound, uint32_t *queue, int lslot, bool big);
size_t hist_queue_words();
const uint32_t *hist_list_count(const uint32_t *queue, int lslot, int which);      // which: 0 lean, 1 rest
// (pair_k5.hip) count / fill of the fixed-cutoff kinds with 4 waves per SIMD (128 VGPRs): frames of large cells, see pair_kernel
void launch_pair_wide(int kind, int mode, unsigned nblocks, hipStream_t stream, const pairk::SearchParams *dP,
                      const pairk::SlotDesc *slot_desc, uint32_t nslots, uint32_t *slot_cnt,
                      const unsigned long long *slot_base, uint2 *pairs, float *dist);
// (pair_k7.hip) the same with 3 waves per SIMD (168 VGPRs), up to 32 chunks of the second cell resident: frames of cells of more
// than 1024 atoms (single-selection and two-selection kinds)
void launch_pair_huge(int kind, int mode, unsigned nblocks, hipStream_t stream, const pairk::SearchParams *dP,
                      const pairk::SlotDesc *slot_desc, uint32_t nslots, uint32_t *slot_cnt,
                      const unsigned long long *slot_base, uint2 *pairs, float *dist);
void launch_pair_single(int mode, unsigned nblocks, size_t dyn_lds, hipStream_t stream, const pairk::SearchParams *dP,
                        const pairk::SlotDesc *slot_desc, uint32_t nslots, uint32_t *slot_cnt,
                        const unsigned long long *slot_base, uint2 *pairs, float *dist, uint32_t *ids);
void launch_pair_double(int mode, unsigned nblocks, size_t dyn_lds, hipStream_t stream, const pairk::SearchParams *dP,
                        const pairk::SlotDesc *slot_desc, uint32_t nslots, uint32_t *slot_cnt,
                        const unsigned long long *slot_base, uint2 *pairs, float *dist, uint32_t *ids);
void launch_pair_within(int mode, unsigned nblocks, size_t dyn_lds, hipStream_t stream, const pairk::SearchParams *dP,
                        const pairk::SlotDesc *slot_desc, uint32_t nslots, uint32_t *slot_cnt,
                        const unsigned long long *slot_base, uint2 *pairs, float *dist, uint32_t *ids);
void launch_pair_vdw(int mode, unsigned nblocks, size_t dyn_lds, hipStream_t stream, const pairk::SearchParams *dP,
                     const pairk::SlotDesc *slot_desc, uint32_t nslots, uint32_t *slot_cnt,
                     const unsigned long long *slot_base, uint2 *pairs, float *dist, uint32_t *ids);

}  // namespace mh
