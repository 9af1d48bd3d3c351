// pair_k5.hip - instantiates the pair kernels of the segmented resident layout (pair_kernel<KIND, MODE_SEG>) for the
// fixed-cutoff search kinds; see pair_kernels.hpp.
#include "pair_kernels.hpp"

namespace mh {

void launch_pair_seg(int kind, unsigned ntasks, hipStream_t stream, const pairk::SearchParams *dP, const pairk::SlotDesc *slot_desc,
                     const uint32_t *task_first, uint32_t *seg_cnt, const unsigned long long *seg_base, uint2 *pairs, float *dist) {
    using namespace pairk;
    if (kind == MOLAR_HIP_SEARCH_SINGLE)
        launch_pair_kernel<MOLAR_HIP_SEARCH_SINGLE, MODE_SEG>(ntasks, 0, stream, dP, slot_desc, ntasks, seg_cnt, seg_base, pairs, dist, nullptr, task_first);
    else
        launch_pair_kernel<MOLAR_HIP_SEARCH_DOUBLE, MODE_SEG>(ntasks, 0, stream, dP, slot_desc, ntasks, seg_cnt, seg_base, pairs, dist, nullptr, task_first);
}

}  // namespace mh
