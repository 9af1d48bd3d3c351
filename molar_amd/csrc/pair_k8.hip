// pair_k8.hip - the two-selection kind of pair_k7.hip's instances (a unit of its own: each takes minutes to compile).
#include "pair_kernels.hpp"

namespace mh {

void launch_pair_huge_double(int mode, unsigned nblocks, hipStream_t stream, const pairk::SearchParams *dP, const pairk::SlotDesc *slot_desc,
                             uint32_t nslots, uint32_t *slot_cnt, const unsigned long long *slot_base, uint2 *pairs, float *dist) {
    using namespace pairk;
    constexpr int W = 3;
    if (mode == MODE_COUNT) launch_pair_kernel<MOLAR_HIP_SEARCH_DOUBLE, MODE_COUNT, W>(nblocks, 0, stream, dP, slot_desc, nslots, slot_cnt, slot_base, pairs, dist, nullptr);
    else launch_pair_kernel<MOLAR_HIP_SEARCH_DOUBLE, MODE_FILL, W>(nblocks, 0, stream, dP, slot_desc, nslots, slot_cnt, slot_base, pairs, dist, nullptr);
}

}  // namespace mh
