// pair_small.hip - count and fill kernels for frames whose cells hold a few atoms (contact and hydrogen-bond cutoffs: 0.35 nm
// gives 6 atoms per cell at water density, 0.5 nm 17).  pair_kernel gives every slot - 64 rows of one plan entry - a wave of
// its own: at these sizes a slot is a 6 x 6 or 17 x 17 block, the wave's lanes are mostly empty and its life is the chain of
// dependent loads in front of the first atom (1M atoms, rc 0.35 nm: 2.3e6 slots, 2.1 ms per frame for 9e6 pairs).  Here a wave
// takes 64 / G consecutive slots per batch, G = 16 or 32 lanes each: lane = candidate of the second cell; the rows of the first
// cell sit one per lane as well and are broadcast inside the slot's lanes (ds_bpermute, no LDS memory); every distance by the
// exact formula of the reference in both passes (no hit history, no matrix cores), hits written straight to their places - a
// slot's output is a few dozen entries.  Per launch at 0.35 nm: 1.9e8 vector instructions for 2.3e6 slots (82 per slot of
// 6 x 6 atoms; 37 % of a slot's 16 lanes hold a candidate), SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES 0.21: bound by vector issue on
// sparse lanes, no longer by latency.
//
// Reference: search_cell_pair_single(_pbc) / _double(_pbc), distance_search.rs:324-373,432-517 (same cell: i < j; wrapped
// entries: PeriodicBox::distance_squared; d2 <= cutoff^2; output (id1, id2, sqrt(d2)) row by row, second index ascending).
// Slots, their counts and bases are those of the regular plan (plan_tiles_kernel / plan_kernel, slot_offsets_kernel).
#include "pair_kernels.hpp"

namespace mh {
namespace pairk {

template <bool FILL, int G, int B>
__global__ void __launch_bounds__(256) small_pair_kernel(const SearchParams *__restrict__ Pp, const SlotDesc *__restrict__ slot_desc,
                                                         const uint32_t nslots, uint32_t *__restrict__ slot_cnt,
                                                         const unsigned long long *__restrict__ slot_base,
                                                         uint2 *__restrict__ out_pairs, float *__restrict__ out_dist) {
    constexpr uint32_t NS = 64u / G;                 // slots per wave and batch
    constexpr uint32_t GM = G == 32 ? 0xFFFFFFFFu : ((1u << (G & 31)) - 1u);
    const SearchParams &P = *Pp;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t sub = lane & (G - 1u), g = lane / G;
    // B batches of NS slots per wave: the records of all of them are requested first, then their atoms, then the batches are
    // evaluated one after the other - a wave's life is the chain record -> atoms -> stores, and two chains in flight per wave
    // hide each other
    bool act[B];
    uint32_t slot[B], a0[B], n1[B], b0[B], n2[B], i0[B], flags[B];
#pragma unroll
    for (int b = 0; b < B; ++b) {
        const unsigned long long slot64 = (((unsigned long long)blockIdx.x * 4u + wave) * B + (uint32_t)b) * NS + g;
        act[b] = slot64 < nslots;
        slot[b] = act[b] ? (uint32_t)slot64 : 0u;
        a0[b] = n1[b] = b0[b] = n2[b] = i0[b] = flags[b] = 0u;
        if (act[b]) {
            const uint4 lo = reinterpret_cast<const uint4 *>(slot_desc + slot[b])[0];
            const uint4 hi = reinterpret_cast<const uint4 *>(slot_desc + slot[b])[1];
            a0[b] = lo.x; n1[b] = lo.y; b0[b] = lo.z; n2[b] = lo.w;
            flags[b] = hi.y; i0[b] = hi.z;
        }
    }
    uint32_t rows[B];
    unsigned long long base[B], end[B];
    float4 ra0[B], q0[B];
#pragma unroll
    for (int b = 0; b < B; ++b) {
        act[b] = act[b] && (flags[b] & 0x200u) != 0u;            // past the last slot of the plan: an empty record
        const uint32_t rps = flags[b] >> 16;
        rows[b] = act[b] ? (n1[b] - i0[b] < rps ? n1[b] - i0[b] : rps) : 0u;
        if (!act[b]) n2[b] = 0u;
        base[b] = end[b] = 0;
        if (FILL && act[b]) {
            base[b] = slot_base[slot[b]];
            end[b] = slot_base[slot[b] + 1];
            if (end[b] == base[b] || end[b] > P.out_cap) rows[b] = 0u;       // nothing to emit / no room (the host grows the buffers and repeats)
        }
        ra0[b] = q0[b] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int b = 0; b < B; ++b) {
        if (sub < rows[b]) ra0[b] = gload4(P.sa, (size_t)a0[b] + i0[b] + sub);          // rows 0 .. G-1 of the slot: lane = row
        if (sub < n2[b] && rows[b]) q0[b] = gload4(P.sb, (size_t)b0[b] + sub);          // candidates 0 .. G-1: lane = candidate
    }
    const float cutoff2 = P.cutoff2;
#pragma unroll
    for (int b = 0; b < B; ++b) {
        const uint32_t wrap = flags[b] & 7u;
        const bool tri = (flags[b] & 0x100u) != 0u;
        const bool wrapped = act[b] && P.use_box != 0u && wrap != 0u;
        // wave-uniform trip counts: the ballots below are taken with every lane present
        uint32_t rows_w = 0, n2_w = 0;
        {
            const uint32_t n2_live = rows[b] ? n2[b] : 0u;
#pragma unroll
            for (uint32_t k = 0; k < NS; ++k) {
                const uint32_t rk = (uint32_t)__builtin_amdgcn_readlane((int)rows[b], (int)(k * G));
                const uint32_t nk = (uint32_t)__builtin_amdgcn_readlane((int)n2_live, (int)(k * G));
                rows_w = rk > rows_w ? rk : rows_w;
                n2_w = nk > n2_w ? nk : n2_w;
            }
        }
        if (rows_w == 0u || n2_w == 0u) {
            if (!FILL && act[b] && sub == 0u) slot_cnt[slot[b]] = 0u;
            continue;
        }
        const bool any_wrapped = __builtin_amdgcn_ballot_w64(wrapped && rows[b] != 0u) != 0ull;
        uint32_t cnt = 0;              // count pass: hits of this lane;  fill pass: entries of the slot written so far
        for (uint32_t rb = 0; rb < rows_w; rb += G) {            // blocks of G rows: lane sub holds row rb + sub
            float4 rv = ra0[b];
            if (rb != 0u) {
                rv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (rb + sub < rows[b]) rv = gload4(P.sa, (size_t)a0[b] + i0[b] + rb + sub);
            }
            const uint32_t rcount = rows_w - rb < (uint32_t)G ? rows_w - rb : (uint32_t)G;
            for (uint32_t rr = 0; rr < rcount; ++rr) {
                const int src = (int)((lane & ~(uint32_t)(G - 1)) + rr);
                float4 a;
                a.x = __shfl(rv.x, src, 64); a.y = __shfl(rv.y, src, 64); a.z = __shfl(rv.z, src, 64); a.w = __shfl(rv.w, src, 64);
                const uint32_t r = rb + rr, i = i0[b] + r;
                for (uint32_t c = 0; c < n2_w; c += G) {
                    const uint32_t j = c + sub;
                    float4 q = q0[b];
                    if (c != 0u) {
                        q = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (j < n2[b] && r < rows[b]) q = gload4(P.sb, (size_t)b0[b] + j);
                    }
                    const float dx = q.x - a.x, dy = q.y - a.y, dz = q.z - a.z;          // p2 - p1
                    float d2 = (dx * dx + dy * dy) + dz * dz;
                    if (any_wrapped) {
                        if (wrapped) d2 = wrapped_d2_exact<false>(P, wrap, dx, dy, dz);
                    }
                    const bool hit = r < rows[b] && j < n2[b] && d2 <= cutoff2 && (!tri || j > i);
                    if (!FILL) {
                        cnt += hit ? 1u : 0u;
                    } else {
                        const unsigned long long mask = __builtin_amdgcn_ballot_w64(hit);
                        if (mask == 0ull) continue;
                        const uint32_t gm = (uint32_t)(mask >> (g * G)) & GM;
                        if (hit) {
                            const unsigned long long pos = base[b] + cnt + (uint32_t)__popc(gm & ((1u << sub) - 1u));
                            if (pos < end[b]) {
                                if (out_pairs)
                                    __builtin_nontemporal_store(((unsigned long long)__float_as_uint(q.w) << 32) | __float_as_uint(a.w),
                                                                reinterpret_cast<unsigned long long *>(out_pairs + pos));
                                if (out_dist) __builtin_nontemporal_store(__builtin_sqrtf(d2), out_dist + pos);     // d2.sqrt() (:448)
                            }
                        }
                        cnt += (uint32_t)__popc(gm);
                    }
                }
            }
        }
        if (!FILL) {
#pragma unroll
            for (int off = G / 2; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
            if (act[b] && sub == 0u) slot_cnt[slot[b]] = cnt;
        }
    }
}

}  // namespace pairk

// lanes_per_slot: 16 or 32
void launch_pair_small(int mode, int lanes_per_slot, hipStream_t stream, const pairk::SearchParams *dP, const pairk::SlotDesc *slot_desc,
                       uint32_t nslots, uint32_t *slot_cnt, const unsigned long long *slot_base, uint2 *pairs, float *dist) {
    using namespace pairk;
    if (nslots == 0) return;
    // batches per wave: two in the count pass (253 against 341 us per launch at 1M atoms / 0.35 nm), one in the fill pass (70
    // registers per lane and 374 us with two, 348 with one)
    constexpr int NB_COUNT = 2, NB_FILL = 1;
    const unsigned per_block = 4u * (unsigned)(mode == MODE_COUNT ? NB_COUNT : NB_FILL) * (64u / (unsigned)lanes_per_slot);
    const dim3 grid((nslots + per_block - 1u) / per_block), block(256);
    if (mode == MODE_COUNT) {
        if (lanes_per_slot == 16) hipLaunchKernelGGL((small_pair_kernel<false, 16, NB_COUNT>), grid, block, 0, stream, dP, slot_desc, nslots, slot_cnt, slot_base, pairs, dist);
        else hipLaunchKernelGGL((small_pair_kernel<false, 32, NB_COUNT>), grid, block, 0, stream, dP, slot_desc, nslots, slot_cnt, slot_base, pairs, dist);
    } else {
        if (lanes_per_slot == 16) hipLaunchKernelGGL((small_pair_kernel<true, 16, NB_FILL>), grid, block, 0, stream, dP, slot_desc, nslots, slot_cnt, slot_base, pairs, dist);
        else hipLaunchKernelGGL((small_pair_kernel<true, 32, NB_FILL>), grid, block, 0, stream, dP, slot_desc, nslots, slot_cnt, slot_base, pairs, dist);
    }
}

}  // namespace mh
