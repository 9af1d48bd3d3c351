// pair_k5.hip - the count / fill kernels of the fixed-cutoff kinds once more, with 4 waves per SIMD (128 VGPRs), for frames whose
// cells hold more than ~330 atoms on average; see pair_kernel (pair_kernels.hpp) and launch_pairs (search.hip).
#include "pair_kernels.hpp"

namespace mh {

void launch_pair_wide(int kind, int mode, unsigned nblocks, hipStream_t stream, const pairk::SearchParams *dP,
                      const pairk::SlotDesc *slot_desc, uint32_t nslots, uint32_t *slot_cnt,
                      const unsigned long long *slot_base, uint2 *pairs, float *dist) {
    using namespace pairk;
    constexpr int W = 4;
    if (kind == MOLAR_HIP_SEARCH_SINGLE) {
        if (mode == MODE_COUNT) launch_pair_kernel<MOLAR_HIP_SEARCH_SINGLE, MODE_COUNT, W>(nblocks, 0, stream, dP, slot_desc, nslots, slot_cnt, slot_base, pairs, dist, nullptr);
        else launch_pair_kernel<MOLAR_HIP_SEARCH_SINGLE, MODE_FILL, W>(nblocks, 0, stream, dP, slot_desc, nslots, slot_cnt, slot_base, pairs, dist, nullptr);
    } else {
        if (mode == MODE_COUNT) launch_pair_kernel<MOLAR_HIP_SEARCH_DOUBLE, MODE_COUNT, W>(nblocks, 0, stream, dP, slot_desc, nslots, slot_cnt, slot_base, pairs, dist, nullptr);
        else launch_pair_kernel<MOLAR_HIP_SEARCH_DOUBLE, MODE_FILL, W>(nblocks, 0, stream, dP, slot_desc, nslots, slot_cnt, slot_base, pairs, dist, nullptr);
    }
}

}  // namespace mh
