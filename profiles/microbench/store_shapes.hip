// store_shapes.hip — which store shape does the pair list want?  Every kernel writes the same 12 B per entry
// (8 B pair plane + 4 B distance plane) from waves that own contiguous segments, like the fill kernel's slots.
//   hipcc --offload-arch=gfx950 -O3 -o store_shapes store_shapes.hip && ./store_shapes
#include <hip/hip_runtime.h>
#include <cstdio>

template <bool NT, class T>
__device__ __forceinline__ void st(T *p, T v) {
    if (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// 64 entries per step: dwordx2 + dword per lane (what fifo_flush does today)
template <bool NT>
__global__ __launch_bounds__(64) void k_seg64(uint2 *__restrict__ p, float *__restrict__ d, size_t n, size_t seg, size_t skew) {
    const size_t lane = threadIdx.x;
    const size_t s = blockIdx.x;
    const size_t base = s * seg + skew;
    for (size_t k = lane; k < seg && base + k < n; k += 64) {
        typedef unsigned u2 __attribute__((ext_vector_type(2)));
        u2 v = {(unsigned)k, (unsigned)lane};
        st<NT>((u2 *)(p + base + k), v);
        st<NT>(&d[base + k], (float)lane);
    }
}
// 128 entries per step: two consecutive entries per lane, dwordx4 + dwordx2
template <bool NT>
__global__ __launch_bounds__(64) void k_seg128(uint2 *__restrict__ p, float *__restrict__ d, size_t n, size_t seg, size_t skew) {
    const size_t lane = threadIdx.x;
    const size_t s = blockIdx.x;
    const size_t base = s * seg + skew;
    for (size_t k = 2 * lane; k + 1 < seg && base + k + 1 < n; k += 128) {
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        typedef float f2 __attribute__((ext_vector_type(2)));
        u4 v = {(unsigned)k, (unsigned)lane, (unsigned)k + 1, (unsigned)lane};
        f2 w = {(float)lane, (float)lane};
        st<NT>((u4 *)(p + base + k), v);
        st<NT>((f2 *)(d + base + k), w);
    }
}

template <class F>
static float time_ms(F f, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < reps; ++r) f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main() {
    const size_t bytes = 4ull << 30;
    char *b;
    hipMalloc(&b, bytes + 4096);
    const size_t npairs = bytes / 12;
    uint2 *p = (uint2 *)b;
    float *d = (float *)(b + npairs * 8);
    // segments as long as a slot's output (~1270 entries on the headline workload) and longer ones
    for (size_t seg : {1280ul, 1270ul, 3600ul, 3617ul}) {
        const unsigned grid = (unsigned)(npairs / seg);
        for (size_t skew : {0ul, 7ul}) {
            const float a0 = time_ms([&] { k_seg64<false><<<grid, 64>>>(p, d, npairs, seg, skew); }, 5);
            const float a1 = time_ms([&] { k_seg64<true><<<grid, 64>>>(p, d, npairs, seg, skew); }, 5);
            const float b0 = time_ms([&] { k_seg128<false><<<grid, 64>>>(p, d, npairs, seg, skew); }, 5);
            const float b1 = time_ms([&] { k_seg128<true><<<grid, 64>>>(p, d, npairs, seg, skew); }, 5);
            const double gb = (double)grid * seg * 12.0 / 1e6;
            printf("seg %5zu skew %zu: 64/step %.0f GB/s  64/step nt %.0f GB/s  128/step %.0f GB/s  128/step nt %.0f GB/s\n", seg, skew,
                   gb / a0, gb / a1, gb / b0, gb / b1);
        }
    }
    return 0;
}
