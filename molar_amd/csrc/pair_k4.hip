// pair_k4.hip - instantiates the lean fused-histogram kernels (hist_kernel) of the fixed-cutoff search kinds; see hist_kernels.hpp.
#include "hist_kernels.hpp"

namespace mh {

void launch_hist_lean(int kind, unsigned num_cus, size_t dyn_lds, hipStream_t stream, const pairk::SearchParams *dP,
                      const pairk::SlotDesc *slot_desc, uint32_t nslots_bound, const uint32_t *nslots_real, uint32_t *queue) {
    using namespace pairk;
    if (kind == MOLAR_HIP_SEARCH_SINGLE)
        launch_hist_kernel<MOLAR_HIP_SEARCH_SINGLE>(num_cus, dyn_lds, stream, dP, slot_desc, nslots_bound, nslots_real, queue);
    else if (kind == MOLAR_HIP_SEARCH_DOUBLE)
        launch_hist_kernel<MOLAR_HIP_SEARCH_DOUBLE>(num_cus, dyn_lds, stream, dP, slot_desc, nslots_bound, nslots_real, queue);
}

size_t hist_queue_words() { return pairk::HIST_QUEUE_WORDS; }

}  // namespace mh
