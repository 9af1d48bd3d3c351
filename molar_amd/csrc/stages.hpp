// stages.hpp - launch-only forms of stages that other translation units chain on the context's stream without a host
// round trip in between (membrane.hip: one bilayer frame = unwrap -> markers -> marker search -> patches -> normals ->
// smoothing -> order, a single wait at the end).  Every pointer is device memory; nothing here synchronises.
#pragma once

#include "common.hpp"

namespace mh {

// measure.hip: the kernels behind molar_hip_unwrap_simple_batch / _center_batch / _lipid_tail_order.
// `status` is a device int raised (atomicMax) to the MOLAR_HIP_ERR_* code of the first kind of failure.
int enqueue_unwrap_batch(molar_hip_ctx *c, float *xyz, const uint64_t *idx, const uint64_t *off, uint32_t nsel,
                         const molar_hip_box &box, uint32_t pbc);
int enqueue_center_batch(molar_hip_ctx *c, const float *xyz, const uint64_t *idx, const uint64_t *off, uint32_t nsel,
                         const float *mass, float *out, int *status);
int enqueue_lipid_order(molar_hip_ctx *c, const float *xyz, const uint64_t *idx, const uint64_t *toff, uint32_t ntails,
                        int order_type, const float *normals, const uint64_t *noff, const uint8_t *bonds, float *out,
                        int *status);

// search.hip: the resident search (molar_hip_search_resident) without its wait.  Count, offset scan and fill go to the
// context's result buffers against their present capacity; the number of results stays in device memory
// (*total_dev, u64) for kernels enqueued behind it, and the two sizes the host needs to judge the capacities are
// copied to `sizes_pinned` (24 bytes: results, hit-history units, slots of the plan).
struct ResidentLaunch {
    unsigned long long cap0 = 0;      // result capacity the fill pass was launched with (0: the fill was skipped)
    unsigned long long maskcap0 = 0;  // hit-history units the count pass could record
    unsigned long long launched = 0;  // slots the count and fill passes were launched over (the plan's real count must not exceed it)
    unsigned long long ntasks = 0;    // entries of the search's plan
    bool degenerate = false;          // empty vdw input: nothing was enqueued, the result is empty
    // Device address of the hit-history units this search's plan asked for (NULL: the search records none).  When they exceed
    // maskcap0 the fill pass has left the wrapped slots' results UNWRITTEN - whatever the buffer held before: a kernel chained
    // behind the search must look (the host only finds out at its next wait) and treat the list as not there.
    const unsigned long long *mask_units_dev = nullptr;
};
int search_resident_enqueue(molar_hip_ctx *c, const molar_hip_search_desc *q, void *sizes_pinned, ResidentLaunch *L,
                            const unsigned long long **total_dev, const uint32_t **pairs_dev);
// true if the sizes delivered for launch L fitted; otherwise the buffers have been grown and the search must be
// enqueued again
int search_resident_fits(molar_hip_ctx *c, const void *sizes_pinned, const ResidentLaunch &L, bool *fits);

// devsort.hip: stable radix sort of (u32, u32) pairs by the keys' low `end_bit` bits, exclusive prefix sums (rocPRIM)
int device_sort_pairs_u32(molar_hip_ctx *c, DevBuf &tmp, const uint32_t *keys_in, uint32_t *keys_out, const uint32_t *vals_in,
                          uint32_t *vals_out, size_t n, int end_bit);
int device_exclusive_sum_u32(molar_hip_ctx *c, DevBuf &tmp, const uint32_t *in, uint32_t *out, size_t n);
int device_exclusive_sum_u32_u64(molar_hip_ctx *c, DevBuf &tmp, const uint32_t *in, unsigned long long *out, size_t n);

}  // namespace mh
