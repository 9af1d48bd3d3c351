// pair_k4.hip - instantiates the lean fused-histogram kernels (hist_kernel) of the fixed-cutoff search kinds; see hist_kernels.hpp.
#include "hist_kernels.hpp"

namespace mh {

void launch_hist_plan(int kind, hipStream_t stream, const pairk::SearchParams &P, pairk::SearchParams *params_dst, pairk::SlotDesc *lean,
                      pairk::SlotDesc *rest, uint32_t *queue, int parity) {
    using namespace pairk;
    if (kind == MOLAR_HIP_SEARCH_SINGLE) launch_hist_plan_kernel<MOLAR_HIP_SEARCH_SINGLE>(stream, P, params_dst, lean, rest, queue, parity);
    else launch_hist_plan_kernel<MOLAR_HIP_SEARCH_DOUBLE>(stream, P, params_dst, lean, rest, queue, parity);
}

void launch_hist_lean(int kind, unsigned num_cus, size_t dyn_lds, hipStream_t stream, const pairk::SearchParams *dP,
                      const pairk::SlotDesc *slot_desc, uint32_t nslots_bound, uint32_t *queue, int parity, bool big) {
    using namespace pairk;
    if (kind == MOLAR_HIP_SEARCH_SINGLE)
        launch_hist_kernel<MOLAR_HIP_SEARCH_SINGLE>(num_cus, dyn_lds, stream, dP, slot_desc, nslots_bound, queue, parity, big);
    else if (kind == MOLAR_HIP_SEARCH_DOUBLE)
        launch_hist_kernel<MOLAR_HIP_SEARCH_DOUBLE>(num_cus, dyn_lds, stream, dP, slot_desc, nslots_bound, queue, parity, big);
}

size_t hist_queue_words() { return pairk::HIST_QUEUE_WORDS; }
const uint32_t *hist_list_count(const uint32_t *queue, int parity, int which) { return queue + pairk::HIST_LIST_WORD + 64u * (unsigned)parity + 32u * (unsigned)which; }

}  // namespace mh
