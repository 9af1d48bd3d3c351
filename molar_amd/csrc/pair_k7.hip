// pair_k7.hip / pair_k8.hip - the count / fill kernels of the fixed-cutoff kinds a third time, with 3 waves per SIMD (168 VGPRs) and up to
// 32 chunks (2048 atoms) of the second cell resident, for frames whose cells hold more than 1000 atoms on average (rc >= 1.9 nm at
// water density; the reference's own sweep goes to 4.2 nm, benches/within_size_bench.rs:13-47).  Until round 6 such cells were
// streamed from memory for every row of the first cell (1M atoms at rc 2.0 nm: 50 M pairs per ms against 170-250 below; now 149,
// 123 at 2.2 nm.  Two waves per SIMD - 256 registers, no spills - gave 102: the row loop wants the third wave more than the
// 28- and 32-chunk variants mind their spills).
// This unit: the single-selection kind and the dispatcher; see pair_kernel / run_task_nch (pair_kernels.hpp), launch_pairs (search.hip).
#include "pair_kernels.hpp"

namespace mh {

void launch_pair_huge_double(int mode, unsigned nblocks, hipStream_t stream, const pairk::SearchParams *dP, const pairk::SlotDesc *slot_desc,
                             uint32_t nslots, uint32_t *slot_cnt, const unsigned long long *slot_base, uint2 *pairs, float *dist);

void launch_pair_huge(int kind, int mode, unsigned nblocks, hipStream_t stream, const pairk::SearchParams *dP,
                      const pairk::SlotDesc *slot_desc, uint32_t nslots, uint32_t *slot_cnt,
                      const unsigned long long *slot_base, uint2 *pairs, float *dist) {
    using namespace pairk;
    constexpr int W = 3;
    if (kind != MOLAR_HIP_SEARCH_SINGLE) {
        launch_pair_huge_double(mode, nblocks, stream, dP, slot_desc, nslots, slot_cnt, slot_base, pairs, dist);
        return;
    }
    if (mode == MODE_COUNT) launch_pair_kernel<MOLAR_HIP_SEARCH_SINGLE, MODE_COUNT, W>(nblocks, 0, stream, dP, slot_desc, nslots, slot_cnt, slot_base, pairs, dist, nullptr);
    else launch_pair_kernel<MOLAR_HIP_SEARCH_SINGLE, MODE_FILL, W>(nblocks, 0, stream, dP, slot_desc, nslots, slot_cnt, slot_base, pairs, dist, nullptr);
}

}  // namespace mh
