// pair_k4.hip - instantiates the lean fused-histogram kernels (hist_kernel) of the fixed-cutoff search kinds; see hist_kernels.hpp.
#include "hist_kernels.hpp"

namespace mh {

void launch_hist_plan(int kind, hipStream_t stream, const pairk::SearchParams &P, pairk::SearchParams *params_dst, pairk::SlotDesc *lean,
                      pairk::SlotDesc *rest, uint32_t *queue, int lslot) {
    using namespace pairk;
    if (kind == MOLAR_HIP_SEARCH_SINGLE) launch_hist_plan_kernel<MOLAR_HIP_SEARCH_SINGLE>(stream, P, params_dst, lean, rest, queue, lslot);
    else launch_hist_plan_kernel<MOLAR_HIP_SEARCH_DOUBLE>(stream, P, params_dst, lean, rest, queue, lslot);
}

void launch_hist_plan_frames(int kind, hipStream_t stream, const pairk::SearchParams *params, unsigned nframes, uint64_t ntasks_max,
                             pairk::SlotDesc *lean, pairk::SlotDesc *rest, uint32_t *queue, int lslot) {
    using namespace pairk;
    if (kind == MOLAR_HIP_SEARCH_SINGLE) launch_hist_plan_frames_kernel<MOLAR_HIP_SEARCH_SINGLE>(stream, params, nframes, ntasks_max, lean, rest, queue, lslot);
    else launch_hist_plan_frames_kernel<MOLAR_HIP_SEARCH_DOUBLE>(stream, params, nframes, ntasks_max, lean, rest, queue, lslot);
}

void launch_hist_lean(int kind, unsigned num_cus, size_t dyn_lds, hipStream_t stream, const pairk::SearchParams *dP,
                      const pairk::SlotDesc *slot_desc, uint32_t nslots_bound, uint32_t *queue, int lslot, bool big) {
    using namespace pairk;
    if (kind == MOLAR_HIP_SEARCH_SINGLE)
        launch_hist_kernel<MOLAR_HIP_SEARCH_SINGLE>(num_cus, dyn_lds, stream, dP, slot_desc, nslots_bound, queue, lslot, big);
    else if (kind == MOLAR_HIP_SEARCH_DOUBLE)
        launch_hist_kernel<MOLAR_HIP_SEARCH_DOUBLE>(num_cus, dyn_lds, stream, dP, slot_desc, nslots_bound, queue, lslot, big);
}

size_t hist_queue_words() { return pairk::HIST_QUEUE_WORDS; }
const uint32_t *hist_list_count(const uint32_t *queue, int lslot, int which) { return queue + pairk::HIST_LIST_WORD + 64u * (unsigned)lslot + 32u * (unsigned)which; }

}  // namespace mh
