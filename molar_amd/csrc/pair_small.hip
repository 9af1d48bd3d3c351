// pair_small.hip - count and fill kernels for frames whose cells hold a few atoms (contact and hydrogen-bond cutoffs: 0.35 nm
// gives 6 atoms per cell at water density, 0.5 nm 17).  pair_kernel gives every slot - 64 rows of one plan entry - a wave of
// its own: at these sizes a slot is a 6 x 6 or 17 x 17 block, the wave's lanes are mostly empty and its life is the chain of
// dependent loads in front of the first atom (1M atoms, rc 0.35 nm: 2.3e6 slots, 2.1 ms per frame for 9e6 pairs).  Here a wave
// takes 64 / G consecutive slots, G = 16 or 32 lanes each: lane = candidate of the second cell, rows of the first cell from
// LDS, every distance by the exact formula of the reference in both passes (no hit history, no matrix cores), hits written
// straight to their places - a slot's output is a few dozen entries.
//
// Reference: search_cell_pair_single(_pbc) / _double(_pbc), distance_search.rs:324-373,432-517 (same cell: i < j; wrapped
// entries: PeriodicBox::distance_squared; d2 <= cutoff^2; output (id1, id2, sqrt(d2)) row by row, second index ascending).
// Slots, their counts and bases are those of the regular plan (plan_tiles_kernel / plan_kernel, slot_offsets_kernel).
#include "pair_kernels.hpp"

namespace mh {
namespace pairk {

template <bool FILL, int G>
__global__ void __launch_bounds__(256) small_pair_kernel(const SearchParams *__restrict__ Pp, const SlotDesc *__restrict__ slot_desc,
                                                         const uint32_t nslots, uint32_t *__restrict__ slot_cnt,
                                                         const unsigned long long *__restrict__ slot_base,
                                                         uint2 *__restrict__ out_pairs, float *__restrict__ out_dist) {
    constexpr uint32_t NS = 64u / G;                 // slots per wave
    constexpr uint32_t GM = G == 32 ? 0xFFFFFFFFu : ((1u << (G & 31)) - 1u);
    __shared__ float4 rows_s[4][NS][64];
    const SearchParams &P = *Pp;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t sub = lane & (G - 1u), g = lane / G;
    const unsigned long long slot64 = ((unsigned long long)blockIdx.x * 4u + wave) * NS + g;
    bool act = slot64 < nslots;
    const uint32_t slot = act ? (uint32_t)slot64 : 0u;
    uint32_t a0 = 0, n1 = 0, b0 = 0, n2 = 0, i0 = 0, flags = 0;
    if (act) {
        const uint4 lo = reinterpret_cast<const uint4 *>(slot_desc + slot)[0];
        const uint4 hi = reinterpret_cast<const uint4 *>(slot_desc + slot)[1];
        a0 = lo.x; n1 = lo.y; b0 = lo.z; n2 = lo.w;
        flags = hi.y; i0 = hi.z;
        act = (flags & 0x200u) != 0u;                // past the last slot of the plan: an empty record
    }
    const uint32_t wrap = flags & 7u, rps = flags >> 16;
    const bool tri = (flags & 0x100u) != 0u;
    const bool wrapped = act && P.use_box != 0u && wrap != 0u;
    uint32_t rows = act ? (n1 - i0 < rps ? n1 - i0 : rps) : 0u;
    if (!act) n2 = 0u;
    unsigned long long base = 0, end = 0;
    if (FILL && act) {
        base = slot_base[slot];
        end = slot_base[slot + 1];
        if (end == base || end > P.out_cap) rows = 0u;       // nothing to emit / no room (the host grows the buffers and repeats)
    }
    float4 *rs = rows_s[wave][g];
    for (uint32_t t = sub; t < rows; t += G) rs[t] = gload4(P.sa, (size_t)a0 + i0 + t);
    float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (sub < n2 && rows) q0 = gload4(P.sb, (size_t)b0 + sub);
    // wave-uniform trip counts: the ballots below are taken with every lane present
    uint32_t rows_w = 0, n2_w = 0;
    {
        const uint32_t n2_live = rows ? n2 : 0u;
#pragma unroll
        for (uint32_t k = 0; k < NS; ++k) {
            const uint32_t rk = (uint32_t)__builtin_amdgcn_readlane((int)rows, (int)(k * G));
            const uint32_t nk = (uint32_t)__builtin_amdgcn_readlane((int)n2_live, (int)(k * G));
            rows_w = rk > rows_w ? rk : rows_w;
            n2_w = nk > n2_w ? nk : n2_w;
        }
    }
    const bool any_wrapped = __builtin_amdgcn_ballot_w64(wrapped && rows != 0u) != 0ull;
    __builtin_amdgcn_wave_barrier();
    const float cutoff2 = P.cutoff2;
    uint32_t cnt = 0;              // count pass: hits of this lane;  fill pass: entries of the slot written so far
    for (uint32_t r = 0; r < rows_w; ++r) {
        const float4 a = rs[r];
        const uint32_t i = i0 + r;
        for (uint32_t c = 0; c < n2_w; c += G) {
            const uint32_t j = c + sub;
            float4 q = q0;
            if (c != 0u) {
                q = make_float4(0.f, 0.f, 0.f, 0.f);
                if (j < n2 && r < rows) q = gload4(P.sb, (size_t)b0 + j);
            }
            const float dx = q.x - a.x, dy = q.y - a.y, dz = q.z - a.z;          // p2 - p1
            float d2 = (dx * dx + dy * dy) + dz * dz;
            if (any_wrapped) {
                if (wrapped) d2 = wrapped_d2_exact<false>(P, wrap, dx, dy, dz);
            }
            const bool hit = r < rows && j < n2 && d2 <= cutoff2 && (!tri || j > i);
            if (!FILL) {
                cnt += hit ? 1u : 0u;
            } else {
                const unsigned long long mask = __builtin_amdgcn_ballot_w64(hit);
                if (mask == 0ull) continue;
                const uint32_t gm = (uint32_t)(mask >> (g * G)) & GM;
                if (hit) {
                    const unsigned long long pos = base + cnt + (uint32_t)__popc(gm & ((1u << sub) - 1u));
                    if (pos < end) {
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
    if (!FILL) {
#pragma unroll
        for (int off = G / 2; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
        if (act && sub == 0u) slot_cnt[slot] = cnt;
    }
}

}  // namespace pairk

// lanes_per_slot: 16 or 32
void launch_pair_small(int mode, int lanes_per_slot, hipStream_t stream, const pairk::SearchParams *dP, const pairk::SlotDesc *slot_desc,
                       uint32_t nslots, uint32_t *slot_cnt, const unsigned long long *slot_base, uint2 *pairs, float *dist) {
    using namespace pairk;
    if (nslots == 0) return;
    const unsigned per_block = 4u * (64u / (unsigned)lanes_per_slot);
    const dim3 grid((nslots + per_block - 1u) / per_block), block(256);
    if (mode == MODE_COUNT) {
        if (lanes_per_slot == 16) hipLaunchKernelGGL((small_pair_kernel<false, 16>), grid, block, 0, stream, dP, slot_desc, nslots, slot_cnt, slot_base, pairs, dist);
        else hipLaunchKernelGGL((small_pair_kernel<false, 32>), grid, block, 0, stream, dP, slot_desc, nslots, slot_cnt, slot_base, pairs, dist);
    } else {
        if (lanes_per_slot == 16) hipLaunchKernelGGL((small_pair_kernel<true, 16>), grid, block, 0, stream, dP, slot_desc, nslots, slot_cnt, slot_base, pairs, dist);
        else hipLaunchKernelGGL((small_pair_kernel<true, 32>), grid, block, 0, stream, dP, slot_desc, nslots, slot_cnt, slot_base, pairs, dist);
    }
}

}  // namespace mh
