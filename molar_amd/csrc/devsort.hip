// devsort.hip - the two device-wide primitives this library takes from rocPRIM (through hipCUB) instead of writing them: a
// STABLE radix sort of (u32 key, u32 value) pairs and exclusive prefix sums.  Stability is what the callers need it for: the
// reference's push orders (items of a grid cell in selection order, distance_search.rs:180,203-209; entries of a connectivity
// list in the order of the pair list, connectivity.rs:19-35) are "sort by the container's index, keep the input order inside".
// One translation unit, so that the sort's many kernels are instantiated once.
#include <hipcub/hipcub.hpp>

#include "common.hpp"

namespace mh {

int device_sort_pairs_u32(molar_hip_ctx *c, DevBuf &tmp, const uint32_t *keys_in, uint32_t *keys_out, const uint32_t *vals_in,
                          uint32_t *vals_out, size_t n, int end_bit) {
    if (n == 0) return 0;
    if (n >= 0x7FFFFFFFull) return fail(MOLAR_HIP_ERR_TOO_LARGE, "device sort: %zu items", n);
    size_t bytes = 0;
    MH_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, keys_in, keys_out, vals_in, vals_out, (int)n, 0, end_bit, c->stream));
    MH_TRY(tmp.reserve(bytes ? bytes : 8));
    MH_HIP(hipcub::DeviceRadixSort::SortPairs(tmp.p, bytes, keys_in, keys_out, vals_in, vals_out, (int)n, 0, end_bit, c->stream));
    return 0;
}

int device_exclusive_sum_u32(molar_hip_ctx *c, DevBuf &tmp, const uint32_t *in, uint32_t *out, size_t n) {
    if (n == 0) return 0;
    if (n >= 0x7FFFFFFFull) return fail(MOLAR_HIP_ERR_TOO_LARGE, "device scan: %zu items", n);
    size_t bytes = 0;
    MH_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, in, out, (int)n, c->stream));
    MH_TRY(tmp.reserve(bytes ? bytes : 8));
    MH_HIP(hipcub::DeviceScan::ExclusiveSum(tmp.p, bytes, in, out, (int)n, c->stream));
    return 0;
}

namespace {
struct U32ToU64 {
    __host__ __device__ unsigned long long operator()(uint32_t v) const { return v; }
};
}  // namespace

int device_exclusive_sum_u32_u64(molar_hip_ctx *c, DevBuf &tmp, const uint32_t *in, unsigned long long *out, size_t n) {
    if (n == 0) return 0;
    if (n >= 0x7FFFFFFFull) return fail(MOLAR_HIP_ERR_TOO_LARGE, "device scan: %zu items", n);
    hipcub::TransformInputIterator<unsigned long long, U32ToU64, const uint32_t *> it(in, U32ToU64());
    size_t bytes = 0;
    MH_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, it, out, (int)n, c->stream));
    MH_TRY(tmp.reserve(bytes ? bytes : 8));
    MH_HIP(hipcub::DeviceScan::ExclusiveSum(tmp.p, bytes, it, out, (int)n, c->stream));
    return 0;
}

}  // namespace mh
