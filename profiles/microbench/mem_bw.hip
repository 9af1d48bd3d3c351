// mem_bw.hip — HBM ceilings of one MI355X for the access shapes of the pair-fill kernel:
//   copy (read+write), write-only float4, write-only as the fill kernel does it (uint2 pair stream + float
//   distance stream, 64 consecutive entries per wave), read-only.
// hipcc --offload-arch=gfx950 -O3 mem_bw.hip -o mem_bw && ./mem_bw
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void k_copy(const float4 *__restrict__ a, float4 *__restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
__global__ __launch_bounds__(256) void k_write4(float4 *__restrict__ b, size_t n) {
    const float4 v = make_float4(1.f, 2.f, 3.f, (float)threadIdx.x);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = v;
}
__global__ __launch_bounds__(256) void k_write_pairs(uint2 *__restrict__ p, float *__restrict__ d, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        p[i] = make_uint2((unsigned)i, (unsigned)threadIdx.x);
        d[i] = (float)threadIdx.x;
    }
}
// each wave owns a contiguous segment (like a slot's output range), written 64 entries at a time
__global__ __launch_bounds__(256) void k_write_pairs_seg(uint2 *__restrict__ p, float *__restrict__ d, size_t n, size_t seg) {
    const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) / 64, lane = threadIdx.x & 63;
    const size_t nw = (size_t)gridDim.x * 4;
    for (size_t s = wave; s * seg < n; s += nw) {
        const size_t base = s * seg;
        for (size_t k = lane; k < seg && base + k < n; k += 64) {
            p[base + k] = make_uint2((unsigned)k, (unsigned)lane);
            d[base + k] = (float)lane;
        }
    }
}
__global__ __launch_bounds__(256) void k_read(const float4 *__restrict__ a, float *out, size_t n) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float4 v = a[i];
        s += v.x + v.y + v.z + v.w;
    }
    if (s == 123.456f) *out = s;
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
    const size_t bytes = 4ull << 30;            // 4 GiB per buffer ~ one frame's pair list
    float4 *a, *b;
    float *o;
    hipMalloc(&a, bytes);
    hipMalloc(&b, bytes);
    hipMalloc(&o, 4);
    hipMemset(a, 0, bytes);
    const size_t n4 = bytes / 16;
    const size_t npairs = bytes / 12;            // 8 B + 4 B per entry in one 4 GiB buffer
    uint2 *p = (uint2 *)b;
    float *d = (float *)((char *)b + npairs * 8);
    for (int grid : {2048, 8192, 32768}) {
        const float c = time_ms([&] { k_copy<<<grid, 256>>>(a, b, n4); }, 5);
        const float w = time_ms([&] { k_write4<<<grid, 256>>>(b, n4); }, 5);
        const float wp = time_ms([&] { k_write_pairs<<<grid, 256>>>(p, d, npairs); }, 5);
        const float ws = time_ms([&] { k_write_pairs_seg<<<grid, 256>>>(p, d, npairs, 3600); }, 5);
        const float r = time_ms([&] { k_read<<<grid, 256>>>(a, o, n4); }, 5);
        printf("grid %6d: copy %.0f GB/s (r+w)  write f4 %.0f GB/s  write pairs %.0f GB/s  write pairs/segments %.0f GB/s  read %.0f GB/s\n",
               grid, 2.0 * bytes / c / 1e6, bytes / w / 1e6, npairs * 12.0 / wp / 1e6, npairs * 12.0 / ws / 1e6, bytes / r / 1e6);
    }
    return 0;
}
