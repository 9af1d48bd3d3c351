// pair_f1.hip - instantiates the single-pass pair kernel (pair_kernel_fused) for distance_search_double(_pbc); see pair_kernels.hpp.
#include "pair_kernels.hpp"

namespace mh {

void launch_fused_double(unsigned nblocks, hipStream_t stream, const pairk::SearchParams *dP, const pairk::SlotDesc *slot_desc,
                         uint32_t nslots, unsigned long long *slot_state, unsigned long long *aux, uint2 *pairs, float *dist) {
    pairk::launch_pair_fused<MOLAR_HIP_SEARCH_DOUBLE>(nblocks, stream, dP, slot_desc, nslots, slot_state, aux, pairs, dist);
}

}  // namespace mh
