// hist_kernels.hpp - the lean kernel of the consumer-fused histogram (molar_hip_search_histogram, BASELINE config 4):
// search + Histogram1D::add_one (molar_membrane/src/stats.rs:29-35) in one pass, no counts, no offsets, no pair list.
// Included by pair_k4.hip only.
//
// It takes the slots that make up nearly all of the work - plain, same-cell and band-classified wrapped entries whose
// second cell fits in registers (hist_lean_slot()); triclinic corner entries, cells > 512 atoms, vdW radii and boxes
// without the band classification stay with pair_kernel<KIND, MODE_HIST>, which skips the slots accepted here.
//
// Bins do not depend on the order in which hits are found, so every entry class walks the second cell in the SPATIAL
// order of the grid (perm_b: the cell's atoms in Morton order, 64-atom chunks with bounding boxes) and a (row, chunk)
// step is skipped when the row's atom is farther than the cutoff from the chunk's box (~43 % of the steps).  A step
// costs 12 vector instructions: 8 of |p2 - p1|^2 in the reference's operation order (distance_search.rs:446,460),
// one compare, two v_mbcnt and one address for the push of the hit lanes' d2 onto a per-wave STACK in LDS; everything
// else of a step (liveness bit, hit count, stack pointer, capacity check) is scalar.  The stack is drained once per
// row, 128 entries at a time, two per lane: bin estimate from v_sqrt_f32, corrected against the exact table of bin
// edges in d2 (histogram_edges(), search.hip), one LDS atomic.  (Round 4's kernel kept the queue's head and tail in
// vector registers and carried a copy of the drain code behind every chunk step: 17 instructions per step and
// 250 KB of code; this one is 12 and a tenth of the code.)
//
// Wrapped entries (a cell pair across the periodic boundary) are classified with the plain distance between the
// second cell's atoms and the row shifted by the wrap's lattice vector (a - S): below the band around cutoff^2 a hit,
// above it a miss, inside it PeriodicBox::distance_squared decides (make_params() derives the band; shifting the row
// instead of the atom has the same 4 u L error bound).  Hits push (row, position in the second cell) onto the same
// stack and their exact distance_squared (periodic_box.rs:286-318) is evaluated densely when the stack is drained
// (second atom gathered from the cell's 6 KB in L2, row from LDS).
//
// LDS is the resource this kernel is careful with: 3 KB per wave (64 rows + 512 stack words), 58 KB per 16-wave
// workgroup with a 1200-bin histogram.  With 75 KB per workgroup (a 3 KB stack per wave) two workgroups still fit a CU
// on paper - and half of them started 100-200 us late in the pipelined run: the one-wave workgroups of the next frame's
// grid build, resident when the kernel starts, leave the first workgroup in the MIDDLE of the CU's 160 KB, and the second
// finds no contiguous 75 KB until the first has ended (profiles/r05_hist_lds_fragmentation.txt).
//
// Work is handed out dynamically: slots sit in per-XCD queues (runs of XCD_RUN consecutive slots, so that an XCD meets
// a second cell again while it is in its L2), four counters per XCD in memory, one returning atomic per slot, issued
// when a wave is free.  (Round 4 gave every workgroup a fixed share of the slots: the workgroups ended between 160 and
// 360 us, 36 % of the kernel's wave-time idle at the end.)
#pragma once

#include "pair_kernels.hpp"

namespace mh {
namespace pairk {

typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) float lds_f32;

#ifndef MH_HIST_WAVES
#define MH_HIST_WAVES 16
#endif
constexpr int HIST_WAVES = MH_HIST_WAVES;      // waves per workgroup (one LDS histogram per workgroup)
#ifndef MH_HIST_WPE
#define MH_HIST_WPE 8          // waves per SIMD the kernel's registers are budgeted for
#endif
#ifndef MH_HIST_CU_WAVES
#define MH_HIST_CU_WAVES 32
#endif
// Resident waves per CU the grid is sized for: all 32.  (24 - two 12-wave workgroups - costs the kernel 2 % alone, and was
// tried so that the next frame's grid, plan and left-over slots could run beside it on the side stream: they do not.  A
// second stream's workgroups get onto this chip beside a persistent kernel only when that kernel holds <= 16 waves per CU,
// whatever registers and LDS it leaves free (profiles/microbench/coresidency.hip, profiles/r05_coresidency_mi355x.txt);
// at 16 waves per CU this kernel is 26 % slower.  So the frames of a trajectory stay one after the other on one stream.)
constexpr int HIST_CU_WAVES = MH_HIST_CU_WAVES;
constexpr uint32_t HQ_WORDS = 512;             // per-wave stack, 32-bit words
constexpr int HQ_ROW_CHUNKS = 6;               // chunks of a row between two looks at the stack: < 128 left over + 6 * 64 pushed fit
static_assert(HQ_WORDS >= 127u + 64u * (uint32_t)HQ_ROW_CHUNKS && 2 * HQ_ROW_CHUNKS >= KREG, "a row is drained at most once in its middle");
constexpr uint32_t HIST_NSUB = 4;              // slot queues per XCD
constexpr uint32_t HIST_LIST_WORD = 32u * (8u * HIST_NSUB + 1u);      // queue words: [queue counters][done][lean, rest counts of list slot 0] ... [of list slot 3]
constexpr uint32_t HIST_TRI_ROWS = 32;         // rows per slot of a same-cell entry: every row meets all chunks, a 64-row slot took 2.7x a plain one

struct HistState {
    lds_u32 *q;            // this wave's stack
    lds_u32 *hist;         // workgroup histogram
    const lds_f32 *edges;  // LDS copy of SearchParams::hist_edges, or NULL: evaluate the formula per hit
    float hmin, hmax, hn, hn1, scale;
    uint32_t nbins;
#ifdef MOLAR_HIP_DEBUG_KNOBS
    mutable unsigned long long t_mark;     // s_memrealtime at the start of the slot's row loop (per-wave time accounting, tools/hist_wave_times.py)
    mutable unsigned long long steps;      // (row, 64-atom chunk) steps evaluated by this wave
#endif
};

__device__ __forceinline__ uint32_t sgpr(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
// LDS byte addresses as numbers: the stack pointer of a wave is ONE scalar (base + 4 * entries), so a push costs the
// vector unit two v_mbcnt and one v_lshl_add and nothing has to be re-derived from a (spilled) base per step
__device__ __forceinline__ uint32_t lds_addr(const lds_u32 *p) { return (uint32_t)(uintptr_t)p; }
__device__ __forceinline__ lds_u32 *lds_ptr(uint32_t a) { return (lds_u32 *)(uintptr_t)a; }

// Histogram1D::add_one (stats.rs:29-35) on d = sqrt(d2):  b = (n as Float * (val - min) / (max - min)).floor() as isize
// The bin is a non-decreasing function of d2 (correctly rounded sqrt, subtraction of and multiplication / division by
// constants, floor), so it is fully described by the smallest d2 that reaches each bin: edges[b], b = 0..n, computed on
// the host with the formula itself.  A cheap estimate of the bin (v_sqrt_f32, one multiply) is at most one bin off
// (ensure_hist_edges() hands the table over only when a bin spans >= 8 ulp of the range's largest distance) and is
// corrected with two comparisons against the exact edges.
__device__ __forceinline__ void hist_add(const HistState &H, float d2) {
    if (H.edges) {
        float est = (__builtin_amdgcn_sqrtf(d2) - H.hmin) * H.scale;
        est = __builtin_fminf(__builtin_fmaxf(est, 0.0f), H.hn1);            // also sends a NaN to 0
        const int b = (int)est;
        const float e0 = H.edges[b], e1 = H.edges[b + 1];
        const int b1 = b + (d2 >= e1 ? 1 : 0) - (d2 < e0 ? 1 : 0);
        if ((uint32_t)b1 < H.nbins) __hip_atomic_fetch_add(H.hist + b1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return;
    }
    const float d = __builtin_sqrtf(d2);
    float fb = __builtin_floorf(H.hn * (d - H.hmin) / (H.hmax - H.hmin));
    if (fb != fb) fb = 0.0f;                                    // NaN as isize == 0
    if (fb >= 0.0f && fb < H.hn) __hip_atomic_fetch_add(H.hist + (uint32_t)fb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// plain stack: entries [base, base + count), count <= 128, two per lane
__device__ __forceinline__ void hist_pop_plain(const HistState &H, uint32_t base, uint32_t count, uint32_t lane) {
    __builtin_amdgcn_wave_barrier();
    const lds_u32 *q = H.q + base;
    const bool a0 = lane < count, a1 = lane + 64u < count;
    float d0 = 0.f, d1 = 0.f;
    if (a0) d0 = __uint_as_float(q[lane]);
    if (a1) d1 = __uint_as_float(q[lane + 64u]);
    if (a0) hist_add(H, d0);
    if (a1) hist_add(H, d1);
    __builtin_amdgcn_wave_barrier();
}

// wrapped stack: entries (row << 16 | position in the second cell's spatial order) [base, base + count), count <= 128.
// Every entry gets the exact PeriodicBox::distance_squared; entries from inside the classification band that turn out to lie
// beyond the cutoff are dropped here - their number is returned.
__device__ __forceinline__ uint32_t hist_pop_wrapped(const SearchParams &P, const HistState &H, const float4 *la, uint32_t wrap, uint32_t b0,
                                                     uint32_t base, uint32_t count, uint32_t lane) {
    __builtin_amdgcn_wave_barrier();
    const lds_u32 *q = H.q + base;
    const float cutoff2 = P.cutoff2;
    const bool a0 = lane < count, a1 = lane + 64u < count;
    uint32_t e0 = 0u, e1 = 0u;
    if (a0) e0 = q[lane];
    if (a1) e1 = q[lane + 64u];
    float4 p0 = make_float4(0.f, 0.f, 0.f, 0.f), p1 = p0;
    if (a0) p0 = gload4(P.perm_b, b0 + (e0 & 0xFFFFu));
    if (a1) p1 = gload4(P.perm_b, b0 + (e1 & 0xFFFFu));
    bool r0 = false, r1 = false;
    if (a0) {
        const float4 a = lload4(la, e0 >> 16);
        const float d2 = wrapped_d2_exact(P, wrap, p0.x - a.x, p0.y - a.y, p0.z - a.z);      // p2 - p1 (distance_search.rs:485-486)
        if (d2 <= cutoff2) hist_add(H, d2);
        else r0 = true;
    }
    if (a1) {
        const float4 a = lload4(la, e1 >> 16);
        const float d2 = wrapped_d2_exact(P, wrap, p1.x - a.x, p1.y - a.y, p1.z - a.z);
        if (d2 <= cutoff2) hist_add(H, d2);
        else r1 = true;
    }
    __builtin_amdgcn_wave_barrier();
    return (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(r0)) + (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(r1));
}

// One slot (<= 64 rows of the first cell against the whole second cell) of a plain or same-cell entry.  `tail_io` = entries
// on the stack, < 128 on entry and on return: what is left over rides along into the wave's next slot.
template <int KIND, int NCH, bool TRI>
__device__ __forceinline__ uint32_t hist_run_plain(const SearchParams &P, const Task &T, uint32_t i0, const HistState &H, uint32_t &tail_io,
                                                   float4 *la, uint32_t lane) {
    const float cutoff2 = P.cutoff2;
    const uint32_t rows = sgpr(T.n1 - i0 < T.rps ? T.n1 - i0 : T.rps);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane < rows) a = gload4(P.sa, T.a0 + i0 + lane);
    la[lane] = a;
    float bx[NCH], by[NCH], bz[NCH];
    uint32_t bpos[TRI ? NCH : 1];      // position in the reference's cell order (same-cell entries: j > i, :443)
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const uint32_t jj = (uint32_t)k * 64u + lane;
        float4 q = make_float4(3.0e38f, 3.0e38f, 3.0e38f, 0.f);      // past the end of the cell: d2 overflows, never a hit
        if (jj < T.n2) q = gload4(P.perm_b, T.b0 + jj);
        bx[k] = q.x; by[k] = q.y; bz[k] = q.z;
        if (TRI) bpos[k] = __float_as_uint(q.w);
    }
    const uint32_t ubase = (T.b0 >> 6) + T.cb;
    unsigned long long livek[NCH], live = 0ull;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const float4 lo = gload4(P.chunk_aabb_b, 2u * (ubase + k)), hi = gload4(P.chunk_aabb_b, 2u * (ubase + k) + 1u);
        // exact: every f32 operation of d2 is monotone in |dx|, |dy|, |dz|
        const bool need = lane < rows && !(aabb_d2(a.x, a.y, a.z, lo.x, lo.y, lo.z, hi.x, hi.y, hi.z) > cutoff2);
        livek[k] = __builtin_amdgcn_ballot_w64(need);
        live |= livek[k];
    }
    __builtin_amdgcn_wave_barrier();
#ifdef MOLAR_HIP_DEBUG_KNOBS
    H.t_mark = __builtin_amdgcn_s_memrealtime();
#endif
    const uint32_t q0 = lds_addr(H.q);
    uint32_t top = sgpr(q0 + 4u * tail_io), total = 0u;      // top: byte address of the first free word
    while (live) {
        const uint32_t r = (uint32_t)__builtin_ctzll(live);
        live &= live - 1ull;
        const float4 p = lload4(la, r);              // one broadcast ds_read per row
        // wait for the row here, once: the chunk bodies sit behind branches, and at their merge points the compiler would
        // otherwise place `s_waitcnt lgkmcnt(0)` in front of every chunk - which also waits for the previous chunk's push
        asm volatile("" ::"v"(p.x), "v"(p.y), "v"(p.z));
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            if (NCH > HQ_ROW_CHUNKS && k == NCH / 2) {            // rows of 7 or 8 chunks: one more look at the stack in the middle
                while (top - q0 >= 4u * 128u) {
                    top -= 4u * 128u;
                    hist_pop_plain(H, (top - q0) >> 2, 128u, lane);
                }
            }
            if (!((livek[k] >> r) & 1ull)) continue;                              // wave-uniform: scalar branch
#ifdef MOLAR_HIP_DEBUG_KNOBS
            H.steps += 1ull;
#endif
            const float dx = bx[k] - p.x, dy = by[k] - p.y, dz = bz[k] - p.z;     // p2 - p1
            const float d2 = (dx * dx + dy * dy) + dz * dz;                      // |p2-p1|^2 (:446, :460)
            bool hit = d2 <= cutoff2;
            if (TRI) hit = hit & (bpos[k] > i0 + r);                              // same cell: j in i+1..n (:443)
            const unsigned long long mask = __builtin_amdgcn_ballot_w64(hit);
            if (mask) {
                if (hit) {
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                    lds_ptr(top)[rank] = __float_as_uint(d2);
                }
                const uint32_t cnt = (uint32_t)__popcll(mask);
                top = sgpr(top + 4u * cnt);
                total += cnt;
            }
        }
        while (top - q0 >= 4u * 128u) {
            top -= 4u * 128u;
            hist_pop_plain(H, (top - q0) >> 2, 128u, lane);
        }
    }
    tail_io = (top - q0) >> 2;
    return total;
}

// One slot of a wrapped entry; the stack is empty on entry and on return (its entries refer to this slot's rows).
template <int KIND, int NCH>
__device__ __forceinline__ uint32_t hist_run_wrapped(const SearchParams &P, const Task &T, uint32_t i0, const HistState &H, float4 *la,
                                                     uint32_t lane) {
    float Sx = 0.f, Sy = 0.f, Sz = 0.f;      // the lattice vector that carries the second cell next to the first
    for (int d = 0; d < 3; ++d) {
        if (!((T.wrap >> d) & 1u)) continue;
        const float sgn = ((T.wrap_b >> d) & 1u) ? 1.0f : -1.0f;   // second cell wrapped: +col, first cell: -col
        Sx += sgn * P.box.m[3 * d];
        Sy += sgn * P.box.m[3 * d + 1];
        Sz += sgn * P.box.m[3 * d + 2];
    }
    const float band_hi = P.band_hi;
    const uint32_t rows = sgpr(T.n1 - i0 < T.rps ? T.n1 - i0 : T.rps);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane < rows) a = gload4(P.sa, T.a0 + i0 + lane);
    la[lane] = a;
    float bx[NCH], by[NCH], bz[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const uint32_t jj = (uint32_t)k * 64u + lane;
        float4 q = make_float4(3.0e38f, 3.0e38f, 3.0e38f, 0.f);
        if (jj < T.n2) q = gload4(P.perm_b, T.b0 + jj);
        bx[k] = q.x; by[k] = q.y; bz[k] = q.z;
    }
    const uint32_t ubase = (T.b0 >> 6) + T.cb;
    unsigned long long livek[NCH], live = 0ull;
    {
        const float ax = a.x - Sx, ay = a.y - Sy, az = a.z - Sz;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const float4 lo = gload4(P.chunk_aabb_b, 2u * (ubase + k)), hi = gload4(P.chunk_aabb_b, 2u * (ubase + k) + 1u);
            // (cutoff + margin)^2: the margin covers what the plain distance to the image may differ from distance_squared by
            const bool need = lane < rows && !(aabb_d2(ax, ay, az, lo.x, lo.y, lo.z, hi.x, hi.y, hi.z) > P.prune_limit2);
            livek[k] = __builtin_amdgcn_ballot_w64(need);
            live |= livek[k];
        }
    }
    __builtin_amdgcn_wave_barrier();
#ifdef MOLAR_HIP_DEBUG_KNOBS
    H.t_mark = __builtin_amdgcn_s_memrealtime();
#endif
    const uint32_t q0 = lds_addr(H.q);
    uint32_t top = q0, total = 0u;               // byte address of the first free word
    while (live) {
        const uint32_t r = (uint32_t)__builtin_ctzll(live);
        live &= live - 1ull;
        const float4 p = lload4(la, r);
        const float px = p.x - Sx, py = p.y - Sy, pz = p.z - Sz;
        const uint32_t rl = (r << 16) | lane;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            if (NCH > HQ_ROW_CHUNKS && k == NCH / 2) {
                while (top - q0 >= 4u * 128u) {
                    top -= 4u * 128u;
                    total -= hist_pop_wrapped(P, H, la, T.wrap, T.b0, (top - q0) >> 2, 128u, lane);
                }
            }
            if (!((livek[k] >> r) & 1ull)) continue;
#ifdef MOLAR_HIP_DEBUG_KNOBS
            H.steps += 1ull;
#endif
            const float dx = bx[k] - px, dy = by[k] - py, dz = bz[k] - pz;
            const float d2 = (dx * dx + dy * dy) + dz * dz;
            // below the band a hit for sure, above it a miss for sure; candidates inside the band are queued like hits and the
            // exact distance_squared of the drain decides about them (hist_pop_wrapped)
            const bool hit = d2 <= band_hi;
            const unsigned long long mask = __builtin_amdgcn_ballot_w64(hit);
            if (mask) {
                if (hit) {
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                    lds_ptr(top)[rank] = rl + (uint32_t)k * 64u;
                }
                const uint32_t cnt = (uint32_t)__popcll(mask);
                top = sgpr(top + 4u * cnt);
                total += cnt;
            }
        }
        while (top - q0 >= 4u * 128u) {
            top -= 4u * 128u;
            total -= hist_pop_wrapped(P, H, la, T.wrap, T.b0, (top - q0) >> 2, 128u, lane);
        }
    }
    if (top != q0) total -= hist_pop_wrapped(P, H, la, T.wrap, T.b0, 0u, (top - q0) >> 2, lane);
    return total;
}

// BIG: frames whose cells hold more than KREG * 64 atoms as a rule (SearchParams::hist_big, set by the host from the mean
// population): a slot's second cell is walked in blocks of that many atoms - bins do not care in which order the pairs of a slot
// are found.  The other instance is the kernel as it was (wrapping its slot body in that loop cost the C4 frame 3-5 %) and
// leaves the odd oversized cell of an ordinary frame to the generic kernel.
template <int KIND, bool BIG>
__global__ void __launch_bounds__(64 * HIST_WAVES) __attribute__((amdgpu_waves_per_eu(MH_HIST_WPE)))
hist_kernel(const SearchParams *__restrict__ Pp, const SlotDesc *__restrict__ slot_desc, uint32_t nslots_bound,
            uint32_t *__restrict__ queue, uint32_t lslot) {
    __shared__ float4 lds_a[HIST_WAVES][64];
    __shared__ uint32_t lds_q[HIST_WAVES][HQ_WORDS];
    extern __shared__ uint32_t lds_hist[];
    // One launch may carry the slots of SEVERAL frames (molar_hip_search_histogram_frames): Pp is then an array with one block per
    // frame and a slot's record says which one is its own (coordinates, box, band).  What belongs to the histogram - bins, range,
    // edges - is the same in all of them and is read from the first.
    const SearchParams &P = *Pp;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = sgpr(threadIdx.x >> 6);
    for (uint32_t b = threadIdx.x; b < P.hist_nbins; b += 64 * HIST_WAVES) lds_hist[b] = 0u;
    float *lds_edges = reinterpret_cast<float *>(lds_hist + P.hist_nbins);       // nbins + 1 floats behind the counters
    if (P.hist_edges)
        for (uint32_t b = threadIdx.x; b <= P.hist_nbins; b += 64 * HIST_WAVES) lds_edges[b] = P.hist_edges[b];
    __syncthreads();
    // Wave-uniform values that live across the whole slot loop are pinned to SGPRs (readfirstlane), never left as "the same
    // value in every lane" of a VGPR: under register pressure the allocator splits such a VGPR's live range with copies, and
    // ROCm 7.2's compiler placed one of them in a join block ahead of the instruction that restores EXEC (found by the
    // fuzzer in round 2, tests/golden/hist_regression_case.npz; molar_amd/build.py audits the ISA for the pattern).
    HistState H;
    H.q = (lds_u32 *)lds_q[wave];
    H.hist = (lds_u32 *)lds_hist;
    H.edges = P.hist_edges ? (const lds_f32 *)lds_edges : nullptr;
    H.scale = P.hist_scale;
    H.hmin = P.hist_min;
    H.hmax = P.hist_max;
    H.hn = uniform_f32((float)P.hist_nbins);
    H.hn1 = uniform_f32((float)P.hist_nbins - 1.0f);
    H.nbins = P.hist_nbins;
    uint32_t tail = 0u;
    unsigned long long wave_total = 0;
#ifdef MOLAR_HIP_DEBUG_KNOBS
    // per-wave time accounting (100 MHz s_memrealtime): start, end, time in front of / inside the row loops, slots by class
    const unsigned long long dbg_t0 = __builtin_amdgcn_s_memrealtime();
    H.steps = 0ull;
    unsigned long long dbg_pre = 0, dbg_rows = 0, dbg_wr = 0, dbg_max = 0, dbg_last = 0, dbg_maxinfo = 0, dbg_maxpre = 0, dbg_maxtk = 0, dbg_maxstart = 0, dbg_first = 0, dbg_t5 = 0, dbg_t6 = 0, dbg_p5 = 0, dbg_p6 = 0;
    uint32_t dbg_n[3] = {0u, 0u, 0u};
#endif
    // The slots that exist (the plan's scan left their number in memory; the host only knows a bound) are walked in reverse
    // plan order - the far x edge, whose entries wrap and cost twice as much, first - in runs of XCD_RUN consecutive slots
    // per XCD (workgroup index mod 8; the grid is a multiple of 8 wide): an XCD meets a second cell's atoms (and a first
    // cell's rows) again while they are in its L2.  Run R of XCD x belongs to queue (x, R mod HIST_NSUB); a wave takes the
    // queue's next slot with one returning atomic - issued while it still works on the slot in hand - so that whoever is
    // free takes what is left, across workgroups.  (One counter for the whole grid serialises: 7*10^4 atomics on one address
    // took 1.1 ms in round 2; here a counter sees ~1800 of them over the kernel's 0.25 ms.)  The last workgroup to leave
    // zeroes the counters for the next launch.
    const uint32_t nlist = queue[HIST_LIST_WORD + 64u * lslot];                     // records hist_plan_kernel wrote into this kernel's list
    const uint32_t nslots = sgpr(nlist < nslots_bound ? nlist : nslots_bound);
    const uint32_t qx = blockIdx.x & 7u, qj = (blockIdx.x >> 3) & (HIST_NSUB - 1u);
    uint32_t *const qctr = queue + 32u * (qx * HIST_NSUB + qj);                      // one counter per 128-byte line
    const uint32_t nruns = (nslots + XCD_RUN - 1u) / XCD_RUN;                        // runs in all; queue (x, j) owns R = (8 k + x) with k mod NSUB == j
    auto take = [&]() -> uint32_t {
        uint32_t t = 0u;
        if (lane == 0) t = __hip_atomic_fetch_add(qctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return t;
    };
    // -DMH_HIST_TICKET_AHEAD=1: the ticket of the NEXT slot is taken while the slot in hand is worked on (one returning atomic in
    // flight per wave), so that its round trip - the first of the four dependent ones in front of a slot's first distance - hides
    // behind the work.  Round 5 measured it 10 % slower on single-frame launches (at the end of a queue every busy wave sits on a
    // slot an idle one could have had); round 6 again on launches over 16 frames, where a queue ends once per group: 0.1993 against
    // 0.1977 ms of kernel per frame in three alternations on one box (tools/ab_rdf.sh) - the ticket costs the kernel its 65th
    // vector register (one spill), and with 8 waves per SIMD its latency was hidden already.  Off.
#ifndef MH_HIST_TICKET_AHEAD
#define MH_HIST_TICKET_AHEAD 0
#endif
    uint32_t ticket_next = MH_HIST_TICKET_AHEAD ? take() : 0u;
    for (;;) {
#ifdef MOLAR_HIP_DEBUG_KNOBS
        const unsigned long long dbg_ta = __builtin_amdgcn_s_memrealtime();
#endif
        const uint32_t s = sgpr(MH_HIST_TICKET_AHEAD ? ticket_next : take());
#ifdef MOLAR_HIP_DEBUG_KNOBS
        const unsigned long long dbg_tb = __builtin_amdgcn_s_memrealtime();      // the ticket has arrived
#endif
        const uint32_t k = (s / XCD_RUN) * HIST_NSUB + qj, run = k * 8u + qx;         // this queue's (s / XCD_RUN)-th run
        if (run >= nruns) break;
        if (MH_HIST_TICKET_AHEAD) ticket_next = take();
        const uint32_t w = run * XCD_RUN + (s % XCD_RUN);
        if (w >= nslots) continue;
        const uint32_t slot = nslots - 1u - w;
        Task T;
        uint32_t i0, fl, frame;
        {
            const uint4 lo = reinterpret_cast<const uint4 *>(slot_desc + slot)[0];
            const uint4 hi = reinterpret_cast<const uint4 *>(slot_desc + slot)[1];
            fl = sgpr(hi.y);
            frame = sgpr(hi.w);
            T.a0 = sgpr(lo.x);
            T.n1 = sgpr(lo.y);
            T.b0 = sgpr(lo.z);
            T.n2 = sgpr(lo.w);
            T.cb = sgpr(hi.x);
            i0 = sgpr(hi.z);
            T.wrap = fl & 7u;
            T.tri = (fl & 0x100u) != 0u;
            T.valid = true;
            T.wrap_b = (fl >> 12) & 7u;
            T.rps = fl >> 16;
        }
        const SearchParams &PF = Pp[frame];          // this slot's frame
#ifdef MOLAR_HIP_DEBUG_KNOBS
        if (P.debug_skip) {
            const uint32_t kind_bit = T.tri ? 4u : ((PF.use_box && T.wrap != 0u) ? 2u : 1u);
            if (P.debug_skip & kind_bit) continue;
        }
#endif
        const uint32_t nchunks = (T.n2 + 63u) >> 6;
        uint32_t total = 0;
#define MH_HIST_CASES(CALL)                  \
    switch (nchunks) {                       \
        case 1: total = CALL(1); break;      \
        case 2: total = CALL(2); break;      \
        case 3: total = CALL(3); break;      \
        case 4: total = CALL(4); break;      \
        case 5: total = CALL(5); break;      \
        case 6: total = CALL(6); break;      \
        case 7: total = CALL(7); break;      \
        default: total = CALL(8); break;     \
    }
        if (!BIG) {
            if (PF.use_box && T.wrap != 0u) {
                if (tail) {                    // plain hits still on the stack: out before entries of the other layout go in
                    hist_pop_plain(H, 0u, tail, lane);
                    tail = 0u;
                }
#define MH_HIST_WRAPPED(N) hist_run_wrapped<KIND, N>(PF, T, i0, H, lds_a[wave], lane)
                MH_HIST_CASES(MH_HIST_WRAPPED)
#undef MH_HIST_WRAPPED
            } else if (KIND == MOLAR_HIP_SEARCH_SINGLE && T.tri) {
#define MH_HIST_TRI(N) hist_run_plain<KIND, N, true>(PF, T, i0, H, tail, lds_a[wave], lane)
                MH_HIST_CASES(MH_HIST_TRI)
#undef MH_HIST_TRI
            } else {
#define MH_HIST_PLAIN(N) hist_run_plain<KIND, N, false>(PF, T, i0, H, tail, lds_a[wave], lane)
                MH_HIST_CASES(MH_HIST_PLAIN)
#undef MH_HIST_PLAIN
            }
#undef MH_HIST_CASES
        } else {
            const uint32_t n2_all = T.n2, b0_all = T.b0;
            for (uint32_t cb0 = 0u; cb0 < n2_all; cb0 += 64u * (uint32_t)KREG) {
                T.b0 = b0_all + cb0;
                T.n2 = n2_all - cb0 < 64u * (uint32_t)KREG ? n2_all - cb0 : 64u * (uint32_t)KREG;
                const uint32_t nchunks_b = (T.n2 + 63u) >> 6;
                uint32_t part = 0;
#define MH_HIST_CASES_B(CALL)                  \
    switch (nchunks_b) {                       \
        case 1: part = CALL(1); break;         \
        case 2: part = CALL(2); break;         \
        case 3: part = CALL(3); break;         \
        case 4: part = CALL(4); break;         \
        case 5: part = CALL(5); break;         \
        case 6: part = CALL(6); break;         \
        case 7: part = CALL(7); break;         \
        default: part = CALL(8); break;        \
    }
                if (PF.use_box && T.wrap != 0u) {
                    if (tail) {                    // plain hits still on the stack: out before entries of the other layout go in
                        hist_pop_plain(H, 0u, tail, lane);
                        tail = 0u;
                    }
#define MH_HIST_WRAPPED_B(N) hist_run_wrapped<KIND, N>(PF, T, i0, H, lds_a[wave], lane)
                    MH_HIST_CASES_B(MH_HIST_WRAPPED_B)
#undef MH_HIST_WRAPPED_B
                } else if (KIND == MOLAR_HIP_SEARCH_SINGLE && T.tri) {
#define MH_HIST_TRI_B(N) hist_run_plain<KIND, N, true>(PF, T, i0, H, tail, lds_a[wave], lane)
                    MH_HIST_CASES_B(MH_HIST_TRI_B)
#undef MH_HIST_TRI_B
                } else {
#define MH_HIST_PLAIN_B(N) hist_run_plain<KIND, N, false>(PF, T, i0, H, tail, lds_a[wave], lane)
                    MH_HIST_CASES_B(MH_HIST_PLAIN_B)
#undef MH_HIST_PLAIN_B
                }
#undef MH_HIST_CASES_B
                total += part;
            }
        }
        wave_total += total;
#ifdef MOLAR_HIP_DEBUG_KNOBS
        {
            const unsigned long long tc = __builtin_amdgcn_s_memrealtime();
            const bool wrapped = PF.use_box && T.wrap != 0u;
            dbg_pre += H.t_mark - dbg_ta;
            dbg_rows += tc - H.t_mark;
            dbg_last = dbg_ta;
            if (!dbg_first) dbg_first = tc - dbg_ta;       // duration of the first slot this wave worked on
            if (T.tri) {        // same-cell slots by chunk count: time (low 40 bits) and number (high bits)
                if (nchunks == 5u) dbg_t5 += (tc - dbg_ta) + (1ull << 40);
                else if (nchunks == 6u) dbg_t6 += (tc - dbg_ta) + (1ull << 40);
            } else if (!wrapped) {
                if (nchunks == 5u) dbg_p5 += (tc - dbg_ta) + (1ull << 40);
                else if (nchunks == 6u) dbg_p6 += (tc - dbg_ta) + (1ull << 40);
            }
            if (tc - dbg_ta > dbg_max) {
                dbg_max = tc - dbg_ta;
                dbg_maxpre = H.t_mark - dbg_ta;
                dbg_maxtk = dbg_tb - dbg_ta;
                dbg_maxstart = dbg_ta;
                dbg_maxinfo = ((unsigned long long)slot << 32) | (fl & 0xFFFFu) | ((unsigned long long)(nchunks & 15u) << 16) | ((unsigned long long)total << 20 & 0xFFF00000ull);
            }
            if (wrapped) dbg_wr += tc - H.t_mark;
            dbg_n[wrapped ? 1 : (T.tri ? 2 : 0)] += 1u;
        }
#endif
    }
    if (tail) hist_pop_plain(H, 0u, tail, lane);
#ifdef MOLAR_HIP_DEBUG_KNOBS
    if (P.dbg && lane == 0) {
        unsigned long long *o = P.dbg + 16ull * (blockIdx.x * HIST_WAVES + wave);
        o[8] = dbg_max;
        o[9] = dbg_last;
        o[10] = dbg_maxinfo;
        o[11] = dbg_maxpre;
        o[12] = dbg_maxtk;
        o[13] = dbg_maxstart;
        o[14] = dbg_first;
        o[15] = dbg_t5;
        o[4] = dbg_t6;          // (replaces the wrapped share)
        o[5] = dbg_p5;
        o[6] = dbg_p6;
        o[0] = dbg_t0;
        o[1] = __builtin_amdgcn_s_memrealtime();
        o[2] = dbg_pre;
        o[3] = dbg_rows;
        o[4] = dbg_wr;
        o[7] = (unsigned long long)__builtin_amdgcn_s_getreg(20 << 0 | 0 << 6 | 31 << 11) | (H.steps << 8);      // HW_REG_XCC_ID | steps
    }
#endif
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < P.hist_nbins; b += 64 * HIST_WAVES) {
        const uint32_t v = lds_hist[b];
        if (v) atomicAdd(&P.hist_bins[b], (unsigned long long)v);
    }
    if (lane == 0 && wave_total && P.hist_total) atomicAdd(P.hist_total, wave_total);
    // every wave of this workgroup is past its last ticket (the barrier above): the last workgroup out resets the queues, and the
    // list counters the launch AFTER THE NEXT will use.  (This launch's own are still read by the generic kernel behind it; the
    // next launch's may be being written already: the plans of a multi-frame launch run on the side stream while the launch
    // before it is still at work.  A plan starts only after the launch two before it has ended - the generation's event.)
    if (threadIdx.x == 0) {
        uint32_t *done = queue + 32u * (8u * HIST_NSUB);
        if (__hip_atomic_fetch_add(done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u) {
            for (uint32_t i = 0; i <= 8u * HIST_NSUB; ++i) __hip_atomic_store(queue + 32u * i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t z = (lslot + 2u) & 3u;
            __hip_atomic_store(queue + HIST_LIST_WORD + 64u * z, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(queue + HIST_LIST_WORD + 64u * z + 32u, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// The plan of a fused-histogram frame in ONE kernel.  The reference's plan order matters to a pair LIST; a histogram only needs
// every (entry, row block) once, so the three launches of the regular plan (entries, scan, slot records) and the parameter upload
// collapse: a thread decodes its plan entry (decode_task: the reference's search_plan, distance_search.rs:217-269), the workgroup
// scans its entries' row-block counts in LDS, reserves its share of the list with ONE atomic and writes the records - into the lean
// kernel's list or, for the entries only the generic kernel can do, into that kernel's own (so neither walks the other's slots).
template <int KIND>
__device__ __forceinline__ void hist_plan_body(const SearchParams &P, uint32_t frame, SlotDesc *__restrict__ lean, SlotDesc *__restrict__ rest,
                                               uint32_t *__restrict__ counts) {
    __shared__ uint32_t sh[2][256];
    __shared__ uint32_t base[2];
    const uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    Task T;
    uint32_t nb = 0u, fl = 0u, rps = 64u;
    bool is_lean = true;
    if (t < P.ntasks) {
        T = decode_task<KIND, false>(P, t);
        if (T.valid) {
            rps = T.rps;
            if (T.tri && rps > HIST_TRI_ROWS) rps = HIST_TRI_ROWS;
            nb = (T.n1 + rps - 1u) / rps;
            fl = T.wrap | (T.tri ? 0x100u : 0u) | 0x200u | (T.wrap_b << 12) | (rps << 16);
            is_lean = P.hist_lean && hist_lean_slot<KIND>(P, fl, T.n2);
        }
    }
    // inclusive scans of the two lists' counts over the workgroup (Hillis-Steele: 8 rounds over 256 entries)
    sh[0][threadIdx.x] = is_lean ? nb : 0u;
    sh[1][threadIdx.x] = is_lean ? 0u : nb;
    __syncthreads();
    for (uint32_t d = 1; d < 256u; d <<= 1) {
        const uint32_t a0 = threadIdx.x >= d ? sh[0][threadIdx.x - d] : 0u, a1 = threadIdx.x >= d ? sh[1][threadIdx.x - d] : 0u;
        __syncthreads();
        sh[0][threadIdx.x] += a0;
        sh[1][threadIdx.x] += a1;
        __syncthreads();
    }
    if (threadIdx.x < 2u) {
        const uint32_t tot = sh[threadIdx.x][255];
        base[threadIdx.x] = tot ? __hip_atomic_fetch_add(counts + 32u * threadIdx.x, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    }
    __syncthreads();
    if (nb == 0u) return;
    const uint32_t l = is_lean ? 0u : 1u;
    SlotDesc *out = (is_lean ? lean : rest) + base[l] + (sh[l][threadIdx.x] - nb);
    SlotDesc o;
    o.a0 = T.a0; o.n1 = T.n1; o.b0 = T.b0; o.n2 = T.n2;
    o.cb = T.cb; o.flags = fl; o.frame = frame;
    o.moff = ~0ull >> 1;          // no hit history in this mode
    o.pad1 = 0ull;
    for (uint32_t k = 0; k < nb; ++k) {
        o.i0 = k * rps;
        out[k] = o;
    }
}

template <int KIND>
__global__ void __launch_bounds__(256) hist_plan_kernel(SearchParams P, SearchParams *__restrict__ params_dst, SlotDesc *__restrict__ lean,
                                                        SlotDesc *__restrict__ rest, uint32_t *__restrict__ counts) {
    if (blockIdx.x == 0) {      // the parameter block the two kernels read, from this kernel's own argument segment (P is its first argument)
        const uint32_t *src = (const uint32_t *)__builtin_amdgcn_kernarg_segment_ptr();
        uint32_t *d = (uint32_t *)params_dst;
        for (uint32_t w = threadIdx.x; w < (uint32_t)(sizeof(SearchParams) / 4u); w += blockDim.x) d[w] = src[w];
    }
    hist_plan_body<KIND>(P, 0u, lean, rest, counts);
}

// The plans of several frames in one launch (blockIdx.y: the frame): the parameter blocks are in device memory already (the grid
// build of the batch put them there), every frame's records go into the same two lists with the frame's number in them.
template <int KIND>
__global__ void __launch_bounds__(256) hist_plan_frames_kernel(const SearchParams *__restrict__ params, SlotDesc *__restrict__ lean,
                                                               SlotDesc *__restrict__ rest, uint32_t *__restrict__ counts) {
    hist_plan_body<KIND>(params[blockIdx.y], blockIdx.y, lean, rest, counts);
}

// words of the queue buffer: slot queues, the `done` word, two pairs of list counters (zeroed once when the buffer is made;
// every launch leaves the queues and the next frame's list counters zeroed)
constexpr size_t HIST_QUEUE_WORDS = HIST_LIST_WORD + 8u * 32u;

template <int KIND>
inline void launch_hist_plan_kernel(hipStream_t stream, const SearchParams &P, SearchParams *params_dst, SlotDesc *lean, SlotDesc *rest,
                                    uint32_t *queue, int lslot) {
    const unsigned nb = (unsigned)((P.ntasks + 255ull) / 256ull);
    hipLaunchKernelGGL((hist_plan_kernel<KIND>), dim3(nb ? nb : 1u), dim3(256), 0, stream, P, params_dst, lean, rest,
                       queue + HIST_LIST_WORD + 64u * (unsigned)lslot);
}

template <int KIND>
inline void launch_hist_plan_frames_kernel(hipStream_t stream, const SearchParams *params, unsigned nframes, uint64_t ntasks_max, SlotDesc *lean,
                                           SlotDesc *rest, uint32_t *queue, int lslot) {
    const unsigned nb = (unsigned)((ntasks_max + 255ull) / 256ull);
    hipLaunchKernelGGL((hist_plan_frames_kernel<KIND>), dim3(nb ? nb : 1u, nframes), dim3(256), 0, stream, params, lean, rest,
                       queue + HIST_LIST_WORD + 64u * (unsigned)lslot);
}

template <int KIND>
inline void launch_hist_kernel(unsigned num_cus, size_t dyn_lds, hipStream_t stream, const SearchParams *dP,
                               const SlotDesc *slot_desc, uint32_t nslots_bound, uint32_t *queue, int lslot, bool big) {
    // persistent workgroups; a multiple of 8 * HIST_NSUB wide so that every queue has the same number of takers
    unsigned nb = num_cus * (HIST_CU_WAVES / HIST_WAVES);
    nb = (nb / (8u * HIST_NSUB)) * (8u * HIST_NSUB);
    if (nb == 0) nb = 8u * HIST_NSUB;
    if (big)
        hipLaunchKernelGGL((hist_kernel<KIND, true>), dim3(nb), dim3(64 * HIST_WAVES), 2 * dyn_lds + 4, stream, dP, slot_desc, nslots_bound, queue, (uint32_t)lslot);
    else
        hipLaunchKernelGGL((hist_kernel<KIND, false>), dim3(nb), dim3(64 * HIST_WAVES), 2 * dyn_lds + 4, stream, dP, slot_desc, nslots_bound, queue, (uint32_t)lslot);   // counters, then the nbins + 1 bin edges
}

}  // namespace pairk
}  // namespace mh
