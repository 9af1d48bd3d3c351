// pair_k5.hip - instantiates the one-pass search kernels (onepass.hpp) of the fixed-cutoff kinds.
#include "onepass.hpp"

namespace mh {

void launch_onepass(int kind, hipStream_t stream, const pairk::SearchParams *dP, const pairk::OnePassArgs &A) {
    using namespace pairk;
    if (kind == MOLAR_HIP_SEARCH_SINGLE) launch_onepass_kernel<MOLAR_HIP_SEARCH_SINGLE>(stream, dP, A);
    else launch_onepass_kernel<MOLAR_HIP_SEARCH_DOUBLE>(stream, dP, A);
}

}  // namespace mh
