// pair_k5.hip / pair_k6.hip - the count / fill kernels of the fixed-cutoff kinds once more, with 4 waves per SIMD (128 VGPRs) and up to
// 16 chunks of the second cell resident, for frames whose cells hold more than 448 atoms on average; see pair_kernel
// (pair_kernels.hpp) and launch_pairs (search.hip).  This unit: the single-selection kind and the dispatcher.
#include "pair_kernels.hpp"

namespace mh {

void launch_pair_wide_double(int mode, unsigned nblocks, hipStream_t stream, const pairk::SearchParams *dP, const pairk::SlotDesc *slot_desc,
                             uint32_t nslots, uint32_t *slot_cnt, const unsigned long long *slot_base, uint2 *pairs, float *dist);

void launch_pair_wide(int kind, int mode, unsigned nblocks, hipStream_t stream, const pairk::SearchParams *dP,
                      const pairk::SlotDesc *slot_desc, uint32_t nslots, uint32_t *slot_cnt,
                      const unsigned long long *slot_base, uint2 *pairs, float *dist) {
    using namespace pairk;
    constexpr int W = 4;
    if (kind != MOLAR_HIP_SEARCH_SINGLE) {
        launch_pair_wide_double(mode, nblocks, stream, dP, slot_desc, nslots, slot_cnt, slot_base, pairs, dist);
        return;
    }
    if (mode == MODE_COUNT) launch_pair_kernel<MOLAR_HIP_SEARCH_SINGLE, MODE_COUNT, W>(nblocks, 0, stream, dP, slot_desc, nslots, slot_cnt, slot_base, pairs, dist, nullptr);
    else launch_pair_kernel<MOLAR_HIP_SEARCH_SINGLE, MODE_FILL, W>(nblocks, 0, stream, dP, slot_desc, nslots, slot_cnt, slot_base, pairs, dist, nullptr);
}

}  // namespace mh
