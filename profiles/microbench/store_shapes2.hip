// store_shapes2.hip — what separates the 6.1 TB/s of a float4 write front from the 4.6-5.0 TB/s of per-wave pair
// segments?  Varies one thing at a time: bytes per lane per store, one plane or two (8 B pair plane + 4 B distance
// plane), a chip-wide write front or one segment per wave, waves per workgroup, ascending or descending segments.
//   hipcc --offload-arch=gfx950 -O3 -o store_shapes2 store_shapes2.hip && ./store_shapes2
#include <hip/hip_runtime.h>
#include <cstdio>

typedef unsigned u2 __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

// ---- one plane, grid-stride write front, T per lane per store
template <class T>
__global__ __launch_bounds__(256) void k_front(T *__restrict__ b, size_t n) {
    T v;
    __builtin_memset(&v, 1, sizeof(T));
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = v;
}
// ---- two planes, write front, E entries per lane per step
template <int E>
__global__ __launch_bounds__(256) void k_front2(unsigned *__restrict__ p, unsigned *__restrict__ d, size_t n) {
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * E; i + E <= n; i += (size_t)gridDim.x * 256 * E) {
        if (E == 1) {
            *(u2 *)(p + 2 * i) = u2{1u, 2u};
            d[i] = 3u;
        } else if (E == 2) {
            *(u4 *)(p + 2 * i) = u4{1u, 2u, 3u, 4u};
            *(u2 *)(d + i) = u2{5u, 6u};
        } else {
            *(u4 *)(p + 2 * i) = u4{1u, 2u, 3u, 4u};
            *(u4 *)(p + 2 * i + 4) = u4{1u, 2u, 3u, 4u};
            *(u4 *)(d + i) = u4{5u, 6u, 7u, 8u};
        }
    }
}
// ---- two planes, one segment of `seg` entries per wave, WAVES waves per workgroup, E entries per lane per step
// REV: workgroup w takes segment nseg-1-w (the fill kernel walks the plan backwards)
template <int E, int WAVES, bool REV>
__global__ __launch_bounds__(64 * WAVES) void k_seg(unsigned *__restrict__ p, unsigned *__restrict__ d, size_t nseg, size_t seg) {
    const size_t lane = threadIdx.x & 63;
    size_t s = (size_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
    if (s >= nseg) return;
    if (REV) s = nseg - 1 - s;
    const size_t base = s * seg;
    for (size_t k = lane * E; k + E <= seg; k += 64 * E) {
        const size_t i = base + k;
        if (E == 1) {
            *(u2 *)(p + 2 * i) = u2{1u, 2u};
            d[i] = 3u;
        } else if (E == 2) {
            *(u4 *)(p + 2 * i) = u4{1u, 2u, 3u, 4u};
            *(u2 *)(d + i) = u2{5u, 6u};
        } else {
            *(u4 *)(p + 2 * i) = u4{1u, 2u, 3u, 4u};
            *(u4 *)(p + 2 * i + 4) = u4{1u, 2u, 3u, 4u};
            *(u4 *)(d + i) = u4{5u, 6u, 7u, 8u};
        }
    }
}
// ---- k_seg<1, 1, false> with data that differs from entry to entry (vary = 1) or not (vary = 0)
__global__ __launch_bounds__(64) void k_seg_data(unsigned *__restrict__ p, unsigned *__restrict__ d, size_t nseg, size_t seg, unsigned vary) {
    const size_t lane = threadIdx.x;
    const size_t s = blockIdx.x;
    const size_t base = s * seg;
    for (size_t k = lane; k < seg; k += 64) {
        const size_t i = base + k;
        const unsigned h = (unsigned)i * 2654435761u * vary;
        *(u2 *)(p + 2 * i) = u2{1u + h, 2u + (h >> 7)};
        d[i] = 3u + (h ^ (h >> 13));
    }
}
// ---- the same with the two planes of a segment written one after the other in bursts of BURST steps
// (pairs of BURST*64 entries, then their distances): fewer switches between the two address streams
template <int BURST, bool REV>
__global__ __launch_bounds__(64) void k_seg_burst(unsigned *__restrict__ p, unsigned *__restrict__ d, size_t nseg, size_t seg) {
    const size_t lane = threadIdx.x;
    size_t s = blockIdx.x;
    if (REV) s = nseg - 1 - s;
    const size_t base = s * seg;
    for (size_t k0 = 0; k0 < seg; k0 += 64 * BURST) {
#pragma unroll
        for (int b = 0; b < BURST; ++b) {
            const size_t k = k0 + b * 64 + lane;
            if (k < seg) *(u2 *)(p + 2 * (base + k)) = u2{1u, 2u};
        }
#pragma unroll
        for (int b = 0; b < BURST; ++b) {
            const size_t k = k0 + b * 64 + lane;
            if (k < seg) d[base + k] = 3u;
        }
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
    char *buf;
    hipMalloc(&buf, bytes + 4096);
    const size_t npairs = bytes / 12 / 1280 * 1280;
    unsigned *p = (unsigned *)buf;
    unsigned *d = (unsigned *)(buf + npairs * 8);
    const double gbp = npairs * 12.0 / 1e6;
    for (int grid : {8192, 32768, 131072}) {
        const float f1 = time_ms([&] { k_front<unsigned><<<grid, 256>>>((unsigned *)buf, bytes / 4); }, 5);
        const float f2 = time_ms([&] { k_front<u2><<<grid, 256>>>((u2 *)buf, bytes / 8); }, 5);
        const float f4 = time_ms([&] { k_front<u4><<<grid, 256>>>((u4 *)buf, bytes / 16); }, 5);
        const float g1 = time_ms([&] { k_front2<1><<<grid, 256>>>(p, d, npairs); }, 5);
        const float g2 = time_ms([&] { k_front2<2><<<grid, 256>>>(p, d, npairs); }, 5);
        const float g4 = time_ms([&] { k_front2<4><<<grid, 256>>>(p, d, npairs); }, 5);
        printf("front grid %6d: one plane b32 %.0f  b64 %.0f  b128 %.0f GB/s | two planes 1/lane %.0f  2/lane %.0f  4/lane %.0f GB/s\n", grid,
               bytes / f1 / 1e6, bytes / f2 / 1e6, bytes / f4 / 1e6, gbp / g1, gbp / g2, gbp / g4);
    }
    for (size_t seg : {1280ul, 5120ul}) {
        const size_t nseg = npairs / seg;
        const float a1 = time_ms([&] { k_seg<1, 1, false><<<(unsigned)nseg, 64>>>(p, d, nseg, seg); }, 5);
        const float a1r = time_ms([&] { k_seg<1, 1, true><<<(unsigned)nseg, 64>>>(p, d, nseg, seg); }, 5);
        const float a2 = time_ms([&] { k_seg<2, 1, false><<<(unsigned)nseg, 64>>>(p, d, nseg, seg); }, 5);
        const float a4 = time_ms([&] { k_seg<4, 1, false><<<(unsigned)nseg, 64>>>(p, d, nseg, seg); }, 5);
        const float w4 = time_ms([&] { k_seg<1, 4, false><<<(unsigned)((nseg + 3) / 4), 256>>>(p, d, nseg, seg); }, 5);
        const float w4e4 = time_ms([&] { k_seg<4, 4, false><<<(unsigned)((nseg + 3) / 4), 256>>>(p, d, nseg, seg); }, 5);
        const float b2 = time_ms([&] { k_seg_burst<2, false><<<(unsigned)nseg, 64>>>(p, d, nseg, seg); }, 5);
        const float b4 = time_ms([&] { k_seg_burst<4, false><<<(unsigned)nseg, 64>>>(p, d, nseg, seg); }, 5);
        const float b4r = time_ms([&] { k_seg_burst<4, true><<<(unsigned)nseg, 64>>>(p, d, nseg, seg); }, 5);
        const float v0 = time_ms([&] { k_seg_data<<<(unsigned)nseg, 64>>>(p, d, nseg, seg, 0u); }, 5);
        const float v1 = time_ms([&] { k_seg_data<<<(unsigned)nseg, 64>>>(p, d, nseg, seg, 1u); }, 5);
        printf("segments of %zu, 1 wave/wg, 1/lane: constant data %.0f  hashed data %.0f GB/s\n", seg, gbp / v0, gbp / v1);
        printf("segments of %zu: 1 wave/wg 1/lane %.0f (descending %.0f)  2/lane %.0f  4/lane %.0f | 4 waves/wg 1/lane %.0f  4/lane %.0f | plane bursts x2 %.0f  x4 %.0f (descending %.0f) GB/s\n",
               seg, gbp / a1, gbp / a1r, gbp / a2, gbp / a4, gbp / w4, gbp / w4e4, gbp / b2, gbp / b4, gbp / b4r);
    }
    return 0;
}
