// coresidency.hip - when does a small kernel on a second (high-priority) stream get onto the chip beside a persistent kernel?
// A: NWG workgroups of WAVES waves, VGPRS registers, LDS bytes of LDS each, spinning for ~T us.  B: one tiny workgroup (256 threads,
// or 64) launched ~20 us after A on another stream; we report when B finished relative to A's start.
//   hipcc --offload-arch=gfx950 -O2 coresidency.hip -o coresidency && ./coresidency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int VGPRS>
__global__ void __launch_bounds__(1024) spin(unsigned long long ticks, unsigned long long *out) {
    extern __shared__ float lds[];
    float acc[VGPRS > 8 ? VGPRS - 8 : 1];
#pragma unroll
    for (int i = 0; i < (int)(sizeof(acc) / 4); ++i) acc[i] = threadIdx.x * 0.5f + i;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
#pragma unroll
        for (int i = 0; i < (int)(sizeof(acc) / 4); ++i) acc[i] = acc[i] * 1.0001f + 0.5f;
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < (int)(sizeof(acc) / 4); ++i) s += acc[i];
    if (s == 12345.f) lds[0] = s;
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t0; out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime(); }
}

__global__ void tiny(unsigned long long *out) {
    if (threadIdx.x == 0) out[0] = __builtin_amdgcn_s_memrealtime();
}

template <int VGPRS>
int run(int nwg, int waves, size_t lds, int bthreads, hipStream_t sa, hipStream_t sb, unsigned long long *dA, unsigned long long *dB) {
    std::vector<unsigned long long> hA(2 * nwg), hB(1);
    CK(hipMemset(dA, 0, 16 * 4096));
    CK(hipMemset(dB, 0, 8));
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL((spin<VGPRS>), dim3(nwg), dim3(64 * waves), lds, sa, 30000ull /* 300 us */, dA);
    // ~20 us later on the other stream
    hipLaunchKernelGGL((spin<16>), dim3(1), dim3(64), 0, sb, 2000ull, dA + 2 * 4000);
    hipLaunchKernelGGL(tiny, dim3(1), dim3(bthreads), 0, sb, dB);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hA.data(), dA, 16 * nwg, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hB.data(), dB, 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull, t1 = 0, late = 0;
    for (int i = 0; i < nwg; ++i) { if (hA[2 * i] < t0) t0 = hA[2 * i]; if (hA[2 * i + 1] > t1) t1 = hA[2 * i + 1]; }
    for (int i = 0; i < nwg; ++i) if (hA[2 * i] - t0 > 5000) ++late;
    printf("A: %4d WGs x %2d waves, %3d VGPRs, %6zu B LDS: A spans %6.1f us, %3llu WGs started > 50 us late; B (%3d threads) ran at %7.1f us after A's start\n",
           nwg, waves, VGPRS, lds, (t1 - t0) * 0.01, late, bthreads, ((long long)hB[0] - (long long)t0) * 0.01);
    return 0;
}

int main() {
    hipStream_t sa, sb;
    int lo, hi;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, hi));
    unsigned long long *dA, *dB;
    CK(hipMalloc(&dA, 16 * 4096 + 64));
    CK(hipMalloc(&dB, 64));
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    printf("%d CUs\n", cus);
    for (int rep = 0; rep < 2; ++rep) {
        run<64>(cus * 2, 16, 58000, 256, sa, sb, dA, dB);     // the round-4 shape: full chip
        run<64>(cus * 2, 12, 46000, 256, sa, sb, dA, dB);     // 24 waves per CU
        run<64>(cus * 2, 12, 46000, 64, sa, sb, dA, dB);
        run<64>(cus * 3, 8, 34000, 256, sa, sb, dA, dB);
        run<64>(cus * 2, 8, 34000, 256, sa, sb, dA, dB);      // 16 waves per CU
        run<64>(cus * 6, 4, 20000, 256, sa, sb, dA, dB);      // 24 waves per CU in 4-wave workgroups
        run<64>(cus * 2 - 32, 12, 46000, 256, sa, sb, dA, dB);
        run<32>(cus * 2, 12, 46000, 256, sa, sb, dA, dB);
        run<64>(cus * 2, 12, 1000, 256, sa, sb, dA, dB);
    }
    return 0;
}
