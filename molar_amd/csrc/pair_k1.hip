// pair_k1.hip - instantiates the pair kernels of ONE search kind (distance_search_double(_pbc)); see pair_kernels.hpp.
#include "pair_kernels.hpp"

namespace mh {

void launch_pair_double(int mode, unsigned nblocks, size_t dyn_lds, hipStream_t stream, const pairk::SearchParams *dP,
                        const pairk::SlotDesc *slot_desc, uint32_t nslots, uint32_t *slot_cnt,
                        const unsigned long long *slot_base, uint2 *pairs, float *dist, uint32_t *ids) {
    using namespace pairk;
    constexpr int KIND = MOLAR_HIP_SEARCH_DOUBLE;
    if (mode == MODE_COUNT)
        launch_pair_kernel<KIND, MODE_COUNT>(nblocks, dyn_lds, stream, dP, slot_desc, nslots, slot_cnt, slot_base, pairs, dist, ids);
    else if (mode == MODE_HIST)
        launch_pair_kernel<KIND, MODE_HIST>(nblocks, dyn_lds, stream, dP, slot_desc, nslots, slot_cnt, slot_base, pairs, dist, ids);
    else
        launch_pair_kernel<KIND, MODE_FILL>(nblocks, dyn_lds, stream, dP, slot_desc, nslots, slot_cnt, slot_base, pairs, dist, ids);
}

}  // namespace mh
