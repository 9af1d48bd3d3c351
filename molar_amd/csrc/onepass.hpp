// onepass.hpp - the distance search in ONE pass over the candidates (resident searches of the fixed-cutoff kinds).
//
// The two-pass search (pair_kernels.hpp) visits every candidate twice: a count pass so that every slot knows where its
// results go, then a fill pass that evaluates the same distances again and writes them.  Here a plan entry is classified
// ONCE, on the matrix cores, and the classification itself is what the results are expanded from:
//
//  * node = (plan entry, one of OP_NW equal shares of its first cell's rows), one 64-lane wave; the OP_NW nodes of an entry
//    form a workgroup and share the second cell's f32 records in LDS;
//  * classification: |p2 - p1|^2 - cutoff^2 of 32 x 32 blocks by v_mfma_f32_32x32x16_f16 exactly as in run_count_mfma
//    (same records, same error bound E, same exact decision for accumulators inside (-E, E)), but with the operands
//    swapped: the second cell's atoms are the A operand and the rows the B operand, so that the 16 accumulators of a lane
//    belong to ONE ROW.  Sign bits -> 16-bit words, v_permlane32_swap joins the two half-waves, a nibble interleave puts the
//    bits into atom order: lane r then holds, per block column, the 32-bit hit mask of row r over 32 atoms;
//  * the row's hit count is a popcount, the node's a wave scan; the node's place in the output (the reference's order:
//    plan order, then row, then atom, distance_search.rs:432-517,949-953) comes from a decoupled look-back over the nodes
//    in front of it (one 64-bit descriptor per node, 64 predecessors per load) - no count pass, no offset scan;
//  * expansion: every lane walks the set bits of its row (cost proportional to HITS, not candidates) and writes
//    (row, atom position) as 16-bit entries to an LDS staging area at the row's offset; then 64 entries at a time are
//    resolved densely: both atoms from LDS, d2 with the reference's expression ((dx*dx)+(dy*dy))+(dz*dz) (wrapped entries:
//    PeriodicBox::distance_squared), correctly rounded sqrt, one 512-byte and one 256-byte non-temporal store.
//
// Entries the matrix-core classification cannot take (the triclinic corner entries, boxes without the band classification,
// a cell paired with its own image, an error bound too wide) are classified exactly on the vector ALUs into the same row
// masks and share everything behind the classification.  A second cell of more than 320 atoms or a row share beyond 64
// ends the pass at once (status 2): such frames run through the count / fill kernels.
//
// Nothing in HIP promises the dispatch order the look-back relies on (a node waits for nodes of lower index, so those must
// have been started).  Observed on gfx950: workgroups start in index order, round-robin over the XCDs.  A node that polls
// longer than OP_SPIN_LIMIT raises `status`; every polling node sees it and leaves; the host then repeats the frame with
// the two-pass kernels and keeps this context off the one-pass path.
#pragma once

#include <type_traits>

#include "pair_kernels.hpp"

namespace mh {
namespace pairk {

constexpr int OP_NW = 5;                          // nodes (waves) per plan entry = waves per workgroup
constexpr uint32_t OP_KC = 260;                   // nodes per entry of the last home cell when its entries may run the triclinic candidate loop
constexpr uint32_t OP_LB = 32u * MFMA_TILES;      // second-cell atoms a workgroup stages in LDS
constexpr uint32_t OP_STAGE = 2816;               // 16-bit staging entries per wave (>= OP_LB: a row fits)
constexpr uint32_t OP_AREA_WORDS = OP_STAGE / 2;  // the per-wave scratch area in 32-bit words (5632 bytes)
constexpr uint32_t OP_SPIN_LIMIT = 1u << 17;     // polls of one node (each up to 3.4 us apart) before the pass is abandoned
constexpr uint32_t OP_TODO_CAP = 512;
static_assert(OP_KC % OP_NW == 0, "the nodes of a workgroup belong to one plan entry");
static_assert(OP_STAGE >= OP_LB && OP_AREA_WORDS * 4u >= 2048u + MFMA_TILES * 64u * 4u, "scratch area: row records + fix words");

struct OnePassArgs {
    unsigned long long *state;        // per node, zero on entry: 1 << 62 | results of the node (op_lookback)
    unsigned long long *blk;          // per block of 64 nodes, zero on entry: nodes reported << 48 | their results
    unsigned long long *blkp;         // per block, zero on entry: 2 << 62 | results up to the end of the block
    uint32_t nnodes;                  // multiple of OP_NW
    uint32_t nreg;                    // nodes of the entries that have OP_NW nodes each: (ntasks - ncorner) * OP_NW
    uint32_t ntask_reg;               // ntasks - ncorner
    uint32_t xcd_run;                 // consecutive workgroups an XCD takes at a time
    uint2 *pairs;
    float *dist;
    unsigned long long *sizes_host;   // pinned {results, 0, status} or NULL
    unsigned long long *total_dev;    // the number of results, for kernels enqueued behind this one
    uint32_t *status;                 // device word, zero on entry; != 0: the pass gave up (look-back timeout)
    uint32_t dbg;                     // profiling aid (MOLAR_HIP_OP_DBG): 1 no look-back (offset = 1400 n), 2 no stores, 4 classification only
};

#define OP_RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// Publish this node's result count and find the number of results in front of it.  Returns false when the pass has been
// abandoned.  Two levels, because the nodes in flight (5 waves x 1024 SIMDs) publish their counts at about the same time and
// a chain of single descriptors would have every one of them walk back over all the others (measured: 80 windows of 64
// descriptors, ~130 us per node, 7 ms per frame):
//   level 1  state[n] = flag | count, read only by the later nodes of the same block of 64 nodes;
//   level 2  blk[B] += 1 << 48 | count (one atomic per node): a block is COMPLETE when 64 nodes have reported, and its sum is
//            then final; blkp[B] = flag | number of results up to the END of block B, published by the block's last node.
// A node adds the counts in front of it inside its block (one load), then walks back over whole blocks, 64 per load, adding
// the sums of complete blocks until it meets a published prefix.  All words are single 8-byte agent-scope atomics on both
// sides; nothing else has to be ordered with them.
__device__ __forceinline__ bool op_lookback(const OnePassArgs &A, uint32_t n, uint32_t total, uint32_t lane, unsigned long long &base,
                                            const bool publish = true, const bool resolve = true) {
    typedef unsigned long long u64;
    constexpr u64 M48 = (1ull << 48) - 1ull, M62 = (1ull << 62) - 1ull;
    base = 0ull;
    if (A.dbg & 1u) {
        base = 1400ull * n;
        return true;
    }
    const uint32_t B = n >> 6, k = n & 63u;
    if (publish && lane == 0u) {
        __hip_atomic_store(&A.state[n], (1ull << 62) | (u64)total, OP_RLX);
        (void)__hip_atomic_fetch_add(&A.blk[B], (1ull << 48) | (u64)total, OP_RLX);
    }
    if (!resolve) return true;
    uint32_t spins = 0u;
    // a poll that found nothing: wait before the next one, longer every time (thousands of waves polling the same few lines
    // at full rate keep the publishers' stores from getting through)
    auto give_up = [&]() -> bool {
        ++spins;
        // (readfirstlane: the compiler then sees a wave-uniform condition and keeps the loops around this on scalar branches)
        if (spins > OP_SPIN_LIMIT || ((spins & 15u) == 0u && __builtin_amdgcn_readfirstlane(__hip_atomic_load(A.status, OP_RLX)) != 0u)) {
            if (lane == 0u) {
                __hip_atomic_store(A.status, 1u, OP_RLX);
                if (A.sizes_host) A.sizes_host[2] = 1ull;
            }
            return true;
        }
        if (spins < 3u) __builtin_amdgcn_s_sleep(8);
        else if (spins < 6u) __builtin_amdgcn_s_sleep(32);
        else __builtin_amdgcn_s_sleep(127);
        return false;
    };
    u64 run = 0ull;
    const bool instr = (A.dbg & 8u) != 0u;
    const u64 t0 = instr ? wall_clock64() : 0ull;
    if (k != 0u) {                                // the nodes in front of this one inside its block
        u64 d = 1ull << 62;
        for (;;) {
            if (lane < k) d = __hip_atomic_load(&A.state[(n - k) + lane], OP_RLX);
            if (__builtin_amdgcn_ballot_w64((d >> 62) == 0ull) == 0ull) break;
            if (give_up()) return false;
        }
        u64 v = lane < k ? (d & M62) : 0ull;
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        run = v;
    }
    const u64 t1 = instr ? wall_clock64() : 0ull;
    const uint32_t spins1 = spins;
    uint32_t windows = 0u;
    long long pos = (long long)B;                 // blocks [0, pos) are still to be added
    while (pos > 0) {
        ++windows;
        const long long b = pos - 1 - (long long)lane;
        u64 p = 2ull << 62, c = 0ull;             // in front of block 0: a prefix of 0
        unsigned long long pm;
        uint32_t f;
        for (;;) {
            if (b >= 0) {
                p = __hip_atomic_load(&A.blkp[b], OP_RLX);
                c = __hip_atomic_load(&A.blk[b], OP_RLX);
            }
            pm = __builtin_amdgcn_ballot_w64((p >> 62) == 2ull);
            f = pm ? (uint32_t)__builtin_ctzll(pm) : 64u;       // nearest block with a published prefix
            // every block between it and this node has to be complete
            if (__builtin_amdgcn_ballot_w64(lane < f && (c >> 48) != 64ull) == 0ull) break;
            if (give_up()) return false;
        }
        u64 v = lane < f ? (c & M48) : (lane == f ? (p & M62) : 0ull);
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        run += v;
        if (pm) break;
        pos -= 64;
    }
    if (k == 63u && lane == 0u) __hip_atomic_store(&A.blkp[B], (2ull << 62) | (run + (u64)total), OP_RLX);
    if (instr && lane == 0u) {                    // MOLAR_HIP_OP_DBG & 8: where the waiting goes (100 MHz ticks, polls, block windows)
        u64 *w = A.total_dev + 1;
        const u64 t2 = wall_clock64();
        (void)__hip_atomic_fetch_add(&w[0], t1 - t0, OP_RLX);
        (void)__hip_atomic_fetch_add(&w[1], t2 - t1, OP_RLX);
        (void)__hip_atomic_fetch_add(&w[2], (u64)spins1, OP_RLX);
        (void)__hip_atomic_fetch_add(&w[3], (u64)(spins - spins1), OP_RLX);
        (void)__hip_atomic_fetch_add(&w[4], (u64)windows, OP_RLX);
        (void)__hip_atomic_fetch_max(&w[5], t2 - t0, OP_RLX);
    }
    base = run;
    return true;
}

// the last node leaves the grand total where the host and later kernels look for it
__device__ __forceinline__ void op_finish(const OnePassArgs &A, uint32_t n, unsigned long long base, uint32_t total, uint32_t lane) {
    if (n + 1u == A.nnodes && lane == 0u) {
        const unsigned long long tot = base + total;
        if (A.total_dev) *A.total_dev = tot;
        if (A.sizes_host) {
            A.sizes_host[0] = tot;
            A.sizes_host[1] = 0ull;
            if (__hip_atomic_load(A.status, OP_RLX) != 0u) A.sizes_host[2] = 1ull;
        }
    }
}

// nibbles of a 16-bit word to the even nibbles of a 32-bit word
__device__ __forceinline__ uint32_t op_spread(uint32_t x) {
    x = (x | (x << 8)) & 0x00FF00FFu;
    return (x | (x << 4)) & 0x0F0F0F0Fu;
}

// One node.  `la`: 64 x float4 of this wave (its live rows, compacted); `lb`: the second cell's f32 records (workgroup);
// `area`: OP_AREA_WORDS words of this wave.  TRI (a same-cell entry), WRAPPED and MF (the entry may be classified on the
// matrix cores) are wave-uniform run-time flags: one copy of the unrolled block loop, one register allocation.
// Every wave of the workgroup passes exactly ONE workgroup barrier in here (behind the block loop: nobody reads `lb` before).
template <int KIND>
__device__ __forceinline__ void op_node(const SearchParams &P, const OnePassArgs &A, const Task &T, const bool TRI, const bool WRAPPED, const bool MF,
                                        uint32_t i0, uint32_t rows, uint32_t n, float4 *la, float4 *lb, const float4 lbv, const bool lbvalid,
                                        uint32_t *area, uint32_t lane) {
    typedef uint32_t u4_t __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) u4_t lds_u4;
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    typedef __attribute__((address_space(3))) uint16_t lds_u16;
    typedef __attribute__((address_space(1))) u4_t glb_u4;
    const float cutoff2 = P.cutoff2;
    const uint32_t kh = lane >> 5, cl = lane & 31u;
    const uint32_t nct = (T.n2 + 31u) >> 5;
    float Sx = 0.f, Sy = 0.f, Sz = 0.f;      // wrapped entries: b + S is the image of the second cell next to the first one (run_fast)
    if (WRAPPED && MF) {
        for (int d = 0; d < 3; ++d) {
            if (!((T.wrap >> d) & 1u)) continue;
            const float sgn = ((T.wrap_b >> d) & 1u) ? 1.0f : -1.0f;
            Sx += sgn * P.box.m[3 * d];
            Sy += sgn * P.box.m[3 * d + 1];
            Sz += sgn * P.box.m[3 * d + 2];
        }
    }
    // everything the node reads from memory is requested at once (see run_count_mfma)
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane < rows) a = gload4(P.sa, T.a0 + i0 + lane);
    float4 org = make_float4(0.f, 0.f, 0.f, 0.f);
    if (MF) org = gload4(P.cell_org_b, T.cb);
    bool need = lane < rows;
    if (!TRI && (MF || !WRAPPED)) {
        const float4 blo = gload4(P.aabb_b, 2 * T.cb), bhi = gload4(P.aabb_b, 2 * T.cb + 1);
        // rows that cannot have a hit: the exact f32 lower bound of run_fast (plain: S = 0, a - 0 is a), the image-box pruning
        // with its margin (wrapped entries under the band classification)
        const float lim = WRAPPED ? P.prune_limit2 : cutoff2;
        need = need && !(aabb_d2(a.x - Sx, a.y - Sy, a.z - Sz, blo.x, blo.y, blo.z, bhi.x, bhi.y, bhi.z) > lim);
    }
    const unsigned long long live = __builtin_amdgcn_ballot_w64(need);
    const uint32_t nlive = TRI ? rows : (uint32_t)__popcll(live);
    unsigned long long base = 0ull;
    if (nlive == 0u) {
        if (lbvalid) lb[threadIdx.x] = lbv;
        __syncthreads();
        if (!op_lookback(A, n, 0u, lane, base)) return;
        op_finish(A, n, base, 0u, lane);
        return;
    }
    // live rows are compacted (their order is kept); same-cell entries keep every row in place (j > i argues with positions)
    const uint32_t rank = TRI ? lane : __builtin_amdgcn_mbcnt_hi((uint32_t)(live >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)live, 0u));
    u4_t bq[MFMA_TILES];
    if (MF) {
#pragma unroll
        for (int t = 0; t < MFMA_TILES; ++t) {
            const uint32_t col = (uint32_t)t * 32u + cl;
            bq[t] = u4_t{0u, 0u, 0u, 0x00007BFFu};                          // atom past the end: |b|^2 = 65504, never a hit
            if (col < T.n2) bq[t] = ((const glb_u4 *)P.h16_b)[T.b0 + col];  // (non-finite atoms carry that record since the grid build)
            if (kh == 0u) bq[t].w = 0x3C003C00u;                            // k = 6, 7 of the first half: (1, 1)
        }
    }
    // this thread's share of the second cell's f32 records goes to LDS behind the node's own requests (both latencies
    // overlap); the workgroup meets once, after the block loop
    if (lbvalid) lb[threadIdx.x] = lbv;
    if (need) la[rank] = a;
    uint32_t w[MFMA_TILES];                      // lane r: hit mask of live row r over the atoms of block column t
#pragma unroll
    for (int t = 0; t < MFMA_TILES; ++t) w[t] = 0u;
    bool mf = MF;
    uint32_t ntodo = 0u;
    lds_u32 *todo = (lds_u32 *)area;             // candidates inside (-E, E): (live row << 16 | atom), decided exactly below
    if (MF) {
        const float r0 = (a.x - Sx) - org.x, r1 = (a.y - Sy) - org.y, r2 = (a.z - Sz) - org.z;
        float ra2 = need ? (r0 * r0 + r1 * r1) + r2 * r2 : 0.0f;
        float big = (WRAPPED && need) ? fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fabsf(a.z)) : 0.0f;
        const bool fin = ra2 == ra2;
        for (int off = 32; off > 0; off >>= 1) ra2 = fmaxf(ra2, __shfl_xor(ra2, off, 64));
        if (WRAPPED)
            for (int off = 32; off > 0; off >>= 1) big = fmaxf(big, __shfl_xor(big, off, 64));
        const float R = 1.0001f * __builtin_sqrtf(ra2) + org.w;
        const float Em = mfma_error_bound(R, cutoff2);
        float Ev = Em;
        bool usable = mfma_bound_usable(R, Em, cutoff2);
        if (WRAPPED) {
            // as run_count_mfma_wrapped: the band of the approximate classification and the shift's own rounding join the bound
            const float rc = __builtin_sqrtf(P.band_hi);
            const float L = (fmaxf(fmaxf(fabsf(org.x), fabsf(org.y)), fabsf(org.z)) + R) + (fmaxf(fmaxf(fabsf(Sx), fabsf(Sy)), fabsf(Sz)) + big);
            Ev = (P.band_hi - cutoff2) * 1.0001f + Em + 4.0f * (5.9604645e-08f * (2.0f * L + 2.0f * rc)) * rc;
            usable = usable && (Ev == Ev) && (Ev < 0.06f * cutoff2);
        }
        const float E = uniform_f32(Ev);
        mf = __builtin_amdgcn_ballot_w64(!fin || !usable) == 0ull;      // bound too wide / not finite: the exact classification below
        if (mf) {
            uint4 *lh = reinterpret_cast<uint4 *>(area);                // 128 x 16 bytes: the rows' matrix-core records, by rank
            ((lds_u4 *)lh)[2u * lane] = u4_t{0u, 0u, 0u, 0x00007BFFu};  // row past the end: +65504
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
            }
            __builtin_amdgcn_wave_barrier();
            const u4_t q0 = ((const lds_u4 *)lh)[2u * cl + kh], q1 = ((const lds_u4 *)lh)[2u * (32u + cl) + kh];
            const v8h_t R0 = __builtin_bit_cast(v8h_t, q0), R1 = __builtin_bit_cast(v8h_t, q1);
            __builtin_amdgcn_wave_barrier();
            // one block: atoms 32 t .. 32 t + 31 (A operand) against live rows 32 rt .. 32 rt + 31 (B operand).  Accumulator i of
            // lane (kh, cl) belongs to row 32 rt + cl and atom 32 t + 8 (i / 4) + 4 kh + i % 4; bit i of the result is its sign.
            auto block = [&](int t, int rt, const v8h_t &At) __attribute__((always_inline)) -> uint32_t {
                const v16f_t zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                const v16f_t acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(At, rt == 0 ? R0 : R1, zero, 0, 0, 0);
                uint32_t h = 0u;
                float m = INFINITY;
#pragma unroll
                for (int i = 15; i >= 0; --i) {
                    h = __builtin_amdgcn_alignbit(h, __float_as_uint(acc[i]), 31);       // h = 2 h + sign
                    m = __builtin_fminf(m, __builtin_fabsf(acc[i]));
                }
                if (__builtin_amdgcn_ballot_w64(m < E) != 0ull) {
                    uint32_t bm = 0u;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const bool ib = __builtin_fabsf(acc[i]) < E;
                        bm |= ib ? (1u << i) : 0u;
                        const unsigned long long mk = __builtin_amdgcn_ballot_w64(ib);
                        if (mk) {
                            const uint32_t at = ntodo + __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
                            const uint32_t col = 32u * (uint32_t)t + 8u * (uint32_t)(i / 4) + 4u * kh + (uint32_t)(i % 4);
                            if (ib && at < OP_TODO_CAP) todo[at] = ((32u * (uint32_t)rt + cl) << 16) | col;
                            ntodo += (uint32_t)__popcll(mk);
                        }
                    }
                    h &= ~bm;
                }
                return h;
            };
#pragma unroll
            for (int t = 0; t < MFMA_TILES; ++t) {
                if ((uint32_t)t < nct) {
                    const v8h_t At = __builtin_bit_cast(v8h_t, bq[t]);
                    uint32_t H0 = 0u, H1 = 0u;
                    // (same-cell entries, j > i (:443): blocks whose atoms all precede their rows hold no pair, but skipping them
                    // puts a branch around every block; they are 3 % of all blocks and are masked with the rest of the triangle)
                    H0 = block(t, 0, At);
                    if (nlive > 32u) H1 = block(t, 1, At);
                    // lanes 32.. of H0 hold the second atom half of rows 0..31, lanes 0..31 of H1 the first atom half of rows 32..63
                    const auto sw = __builtin_amdgcn_permlane32_swap(H0, H1, false, false);
                    w[t] = op_spread(sw[0]) | (op_spread(sw[1]) << 4);
                }
            }
        }
    }
    __syncthreads();                             // the second cell's f32 records are complete
    if (mf && ntodo > OP_TODO_CAP) mf = false;   // a node full of pairs at the cutoff: decide everything exactly
    if (mf) {
        if (ntodo) {
            lds_u32 *fix = (lds_u32 *)(area + 512);  // behind the todo list: MFMA_TILES x 64 words
#pragma unroll
            for (int t = 0; t < MFMA_TILES; ++t) fix[t * 64 + (int)lane] = 0u;
            __builtin_amdgcn_wave_barrier();
            for (uint32_t q = 0; q < ntodo; q += 64u) {
                if (q + lane < ntodo) {
                    const uint32_t e = todo[q + lane];
                    const uint32_t rl = e >> 16, col = e & 0xFFFFu;
                    if (rl < nlive && col < T.n2) {
                        const float4 p = lload4(la, rl), b = lload4(lb, col);
                        const float dx = b.x - p.x, dy = b.y - p.y, dz = b.z - p.z;               // p2 - p1
                        const float d2 = WRAPPED ? wrapped_d2_exact(P, T.wrap, dx, dy, dz) : (dx * dx + dy * dy) + dz * dz;
                        if (d2 <= cutoff2) atomicOr(area + 512 + (col >> 5) * 64u + rl, 1u << (col & 31u));
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int t = 0; t < MFMA_TILES; ++t) w[t] |= fix[t * 64 + (int)lane];
            __builtin_amdgcn_wave_barrier();
        }
    } else {
        // Exact classification on the vector ALUs (entries the matrix cores cannot take: the triclinic corner entries, boxes
        // without the band classification, a cell paired with its own image, bounds too wide): lanes = the atoms of a
        // 64-chunk, one live row at a time from LDS, the reference's expression; the ballot IS the row's hit mask over the
        // chunk and goes to lane `row`.
#pragma unroll
        for (int c = 0; c < MFMA_TILES / 2; ++c) {
            if ((uint32_t)c * 64u < T.n2) {
                const uint32_t j = (uint32_t)c * 64u + lane;
                float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
                if (j < T.n2) b = lload4(lb, j);
                uint32_t wlo = 0u, whi = 0u;
                for (uint32_t r = 0; r < nlive; ++r) {
                    const float4 p = lload4(la, r);
                    const float dx = b.x - p.x, dy = b.y - p.y, dz = b.z - p.z;                   // p2 - p1
                    const float d2 = WRAPPED ? wrapped_d2_exact(P, T.wrap, dx, dy, dz) : (dx * dx + dy * dy) + dz * dz;
                    const unsigned long long mk = __builtin_amdgcn_ballot_w64(j < T.n2 && d2 <= cutoff2);
                    if (lane == r) {
                        wlo = (uint32_t)mk;
                        whi = (uint32_t)(mk >> 32);
                    }
                }
                w[2 * c] = wlo;
                w[2 * c + 1] = whi;
            }
        }
    }
    uint32_t cnt = 0u;
#pragma unroll
    for (int t = 0; t < MFMA_TILES; ++t) {
        if (TRI) {                               // keep j > i only: atom 32 t + bit against row i0 + lane
            const uint32_t i = i0 + lane, lo = 32u * (uint32_t)t;
            if (i >= lo + 31u) w[t] = 0u;
            else if (i >= lo) w[t] &= ~((2u << (i - lo)) - 1u);
        }
        cnt += (uint32_t)__popc(w[t]);
    }
    // row offsets inside the node
    uint32_t incl = cnt;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off, 64);
        if ((int)lane >= off) incl += o;
    }
    const uint32_t excl = incl - cnt;
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    // The count goes out at once; the offset is asked for only when the first group of rows has been expanded (that needs no
    // offset) - by then the nodes in front have usually reported and nobody polls.
    (void)op_lookback(A, n, total, lane, base, true, false);
    bool resolved = false;
    lds_u16 *stage = (lds_u16 *)area;
    uint32_t r_lo = 0u;
    if (total == 0u || (A.dbg & 4u)) r_lo = nlive;
    while (r_lo < nlive) {
        // rows [r_lo, r_hi): as many as the staging area holds
        const uint32_t g0 = (uint32_t)__builtin_amdgcn_readlane((int)excl, (int)r_lo);
        const unsigned long long fit = __builtin_amdgcn_ballot_w64(lane >= r_lo && lane < nlive && incl - g0 <= OP_STAGE) >> r_lo;
        const uint32_t r_hi = r_lo + (uint32_t)__builtin_ctzll(~fit);
        const uint32_t ng = (uint32_t)__builtin_amdgcn_readlane((int)incl, (int)(r_hi - 1u)) - g0;
        const bool mine = lane >= r_lo && lane < r_hi;
        if (ng) {
            uint32_t pos = excl - g0;
#pragma unroll
            for (int t = 0; t < MFMA_TILES; ++t) {
                if ((uint32_t)t < nct) {
                    uint32_t word = mine ? w[t] : 0u;
                    while (word) {
                        const uint32_t b = (uint32_t)__builtin_ctz(word);
                        word &= word - 1u;
                        stage[pos] = (uint16_t)((lane << 9) | (32u * (uint32_t)t + b));
                        ++pos;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (!resolved) {
                if (!op_lookback(A, n, total, lane, base, false, true)) return;
                op_finish(A, n, base, total, lane);
                resolved = true;
                if (base + total > P.out_cap) return;               // no room (the host grows and repeats)
            }
            // 64 entries at a time; the first step ends at the next 64-entry boundary of the output so that every later
            // one is a naturally aligned 512-byte / 256-byte block (fifo_drain)
            const unsigned long long out0 = base + g0;
            const int32_t lead = (int32_t)((uint32_t)out0 & 63u);
            uint2 *pp = A.pairs + out0;
            float *pd = A.dist + out0;
            for (int32_t p = -lead; p < (int32_t)ng; p += 64) {
                const int32_t k = p + (int32_t)lane;
                if (k >= 0 && k < (int32_t)ng) {
                    const uint32_t e = stage[k];
                    const float4 pa = lload4(la, e >> 9), pb = lload4(lb, e & 511u);
                    const float dx = pb.x - pa.x, dy = pb.y - pa.y, dz = pb.z - pa.z;         // p2 - p1 (:446, :460, :485)
                    const float d2 = WRAPPED ? wrapped_d2_exact(P, T.wrap, dx, dy, dz) : (dx * dx + dy * dy) + dz * dz;
                    if ((A.dbg & 2u) && d2 != -1.0f) continue;
                    __builtin_nontemporal_store(((unsigned long long)__float_as_uint(pb.w) << 32) | __float_as_uint(pa.w),
                                                reinterpret_cast<unsigned long long *>(pp + k));
                    __builtin_nontemporal_store(__builtin_sqrtf(d2), pd + k);                 // d2.sqrt() (:448)
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        r_lo = r_hi;
    }
    if (!resolved) {                             // no results: the node still passes its prefix on (last node of a block, last node)
        if (!op_lookback(A, n, total, lane, base, false, true)) return;
        op_finish(A, n, base, total, lane);
    }
}

template <int KIND>
__global__ void __launch_bounds__(64 * OP_NW) __attribute__((amdgpu_waves_per_eu(5, 5)))
onepass_kernel(const SearchParams *__restrict__ Pp, const OnePassArgs A) {
    __shared__ float4 lds_b[OP_LB];
    __shared__ float4 lds_a[OP_NW][64];
    __shared__ uint32_t lds_area[OP_NW][OP_AREA_WORDS];
    const SearchParams &P = *Pp;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t nwg = A.nnodes / (uint32_t)OP_NW;
    uint32_t g = blockIdx.y * gridDim.x + blockIdx.x;
    if (g >= nwg) return;
    {   // workgroup b runs on XCD b % 8 with its own L2: an XCD takes runs of consecutive entries (their cells meet again in its
        // L2), in ascending order - a node only ever waits for nodes of lower index
        const uint32_t run = A.xcd_run, x = g & 7u, q = g >> 3;
        const uint32_t full = (nwg / (8u * run)) * (8u * run);
        if (g < full) g = ((q / run) * 8u + x) * run + (q % run);
    }
    const uint32_t n = g * (uint32_t)OP_NW + wave;
    uint32_t t, w, K;
    if (n < A.nreg) {
        t = n / (uint32_t)OP_NW;
        w = n % (uint32_t)OP_NW;
        K = (uint32_t)OP_NW;
    } else {
        const uint32_t m = n - A.nreg;
        t = A.ntask_reg + m / OP_KC;
        w = m % OP_KC;
        K = OP_KC;
    }
    Task T;
    {
        const uint4 lo = reinterpret_cast<const uint4 *>(P.task_desc + t)[0];
        const uint4 hi = reinterpret_cast<const uint4 *>(P.task_desc + t)[1];
        const uint32_t fl = __builtin_amdgcn_readfirstlane(hi.y);
        T.a0 = __builtin_amdgcn_readfirstlane(lo.x);
        T.n1 = __builtin_amdgcn_readfirstlane(lo.y);
        T.b0 = __builtin_amdgcn_readfirstlane(lo.z);
        T.n2 = __builtin_amdgcn_readfirstlane(lo.w);
        T.cb = __builtin_amdgcn_readfirstlane(hi.x);
        T.ca = 0u;
        T.wrap = fl & 7u;
        T.tri = (fl & 0x100u) != 0u;
        T.valid = (fl & 0x200u) != 0u;
        T.wrap_b = (fl >> 12) & 7u;
        T.rps = 64u;
    }
    unsigned long long base = 0ull;
    if (!T.valid) {                              // entry dropped by the reference or without atoms: nothing, but the chain goes on
        if (op_lookback(A, n, 0u, lane, base)) op_finish(A, n, base, 0u, lane);
        return;
    }
    const uint32_t R = (T.n1 + K - 1u) / K;      // rows per node
    if (T.n2 > OP_LB || R > 64u) {
        // a cell too large for this kernel's LDS stage / a row share beyond one wave: the pass is abandoned at once (every
        // polling node sees the status) and the host runs the frame through the count / fill passes
        if (lane == 0u) {
            __hip_atomic_store(A.status, 2u, OP_RLX);
            if (A.sizes_host) A.sizes_host[2] = 2ull;
        }
        return;
    }
    const uint32_t i0 = w * R;
    const uint32_t rows = i0 < T.n1 ? (T.n1 - i0 < R ? T.n1 - i0 : R) : 0u;
    const bool wrapped = P.use_box && T.wrap != 0u;
    const bool tri = KIND == MOLAR_HIP_SEARCH_SINGLE && T.tri;
    // classification on the matrix cores, or exactly on the vector ALUs (op_node)
    const bool mf = (P.mfma_count & 1u) && !(wrapped && (tri || P.approx_wrapped == 0u || (P.box.nshift != 0 && T.wrap == MOLAR_HIP_PBC_FULL)));
    // each thread fetches one of the second cell's f32 records for the workgroup's LDS copy
    static_assert(OP_LB <= 64u * (uint32_t)OP_NW, "one record per thread");
    const bool lbvalid = threadIdx.x < T.n2;
    float4 lbv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lbvalid) lbv = gload4(P.sb, T.b0 + threadIdx.x);
    if (rows == 0u) {
        if (lbvalid) lds_b[threadIdx.x] = lbv;
        __syncthreads();
        if (op_lookback(A, n, 0u, lane, base)) op_finish(A, n, base, 0u, lane);
        return;
    }
    op_node<KIND>(P, A, T, tri, wrapped, mf, i0, rows, n, lds_a[wave], lds_b, lbv, lbvalid, lds_area[wave], lane);
}

template <int KIND>
inline void launch_onepass_kernel(hipStream_t stream, const SearchParams *dP, const OnePassArgs &A) {
    hipLaunchKernelGGL((onepass_kernel<KIND>), pair_grid(A.nnodes / (unsigned)OP_NW), dim3(64 * OP_NW), 0, stream, dP, A);
}

}  // namespace pairk

// defined in pair_k5.hip
void launch_onepass(int kind, hipStream_t stream, const pairk::SearchParams *dP, const pairk::OnePassArgs &A);

}  // namespace mh
