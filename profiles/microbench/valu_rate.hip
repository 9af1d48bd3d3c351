// valu_rate.hip — issue cost of the VALU instructions the pair kernels are made of (gfx950).
// Each kernel runs ITER iterations of 32 independent instructions of one kind per wave, with
// WAVES_PER_SIMD waves resident per SIMD; prints cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int ITER = 4096;

template <int KIND>
__global__ void __launch_bounds__(256) k(float *out, float s) {
    float a[16];
    v2f p[16];
    for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x * 0.001f + i; p[i] = v2f{a[i], a[i] + 1.f}; }
    unsigned long long acc = 0;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (KIND == 0) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "s"(s));
                if (KIND == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 15]));
                if (KIND == 2) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(a[i]));
                if (KIND == 3) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(p[i]));
                if (KIND == 4) { unsigned long long m; asm volatile("v_cmp_le_f32 %0, %1, %2" : "=s"(m) : "v"(a[i]), "v"(a[(i + 1) & 15])); acc += m; }
                if (KIND == 5) { int v; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(v) : "v"(a[i])); acc += v; }
                if (KIND == 6) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));
                if (KIND == 7) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[i]));
                if (KIND == 8) asm volatile("v_trunc_f32 %0, %0" : "+v"(a[i]));
                if (KIND == 9) asm volatile("v_mbcnt_lo_u32_b32 %0, -1, %0" : "+v"(a[i]));
                if (KIND == 10) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 15]));
                if (KIND == 11) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 15]));
                if (KIND == 12) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "s"(s));
                if (KIND == 13) asm volatile("v_cmp_le_f32 vcc, %0, %1" :: "v"(a[i]), "v"(a[(i + 1) & 15]) : "vcc");
                if (KIND == 14) { unsigned long long m; asm volatile("v_cmp_le_f32 %0, %1, %2" : "=s"(m) : "v"(a[i]), "v"(a[(i + 1) & 15])); asm volatile("" :: "s"(m)); }
                if (KIND == 15) { int v; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(v) : "v"(a[i])); asm volatile("" :: "s"(v)); }
                if (KIND == 16) { unsigned long long m; int c; asm volatile("v_cmp_le_f32 %0, %2, %3\n s_bcnt1_i32_b64 %1, %0" : "=&s"(m), "=s"(c) : "v"(a[i]), "v"(a[(i + 1) & 15]) : "scc"); acc += (unsigned)c; }
                if (KIND == 17) asm volatile("v_subrev_f32 %0, %1, %0" : "+v"(a[i]) : "s"(s));
                if (KIND == 18) asm volatile("v_add_u32 %0, %0, %0" : "+v"(a[i]));
                if (KIND == 19) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(a[(i + 1) & 15]));
            }
    }
    float r = 0;
    for (int i = 0; i < 16; ++i) r += a[i] + p[i].x + p[i].y;
    if (r == 12345.f || acc == 77) out[0] = r;
}

template <int KIND>
int run(const char *name, int waves_per_simd) {
    float *d; CHECK(hipMalloc(&d, 4));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 256 * waves_per_simd;      // 256 threads = 4 waves = 1 wave per SIMD per block
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
    CHECK(hipDeviceSynchronize());
    hipEventRecord(a);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
    hipEventRecord(b); CHECK(hipEventSynchronize(b));
    float ms; hipEventElapsedTime(&ms, a, b);
    const double instr_per_simd = (double)ITER * 32 * waves_per_simd;
    printf("%-16s waves/SIMD=%d  %.3f ms  -> %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name,
           waves_per_simd, ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
    hipFree(d);
    return 0;
}

int main() {
    for (int w : {1, 4, 8}) {
        run<0>("v_add_f32", w); run<1>("v_pk_add_f32", w); run<2>("v_mul_f32", w); run<3>("v_pk_mul_f32", w);
        run<4>("v_cmp_le_f32", w); run<5>("v_readlane_b32", w); run<6>("v_fma_f32", w); run<7>("v_pk_fma_f32", w);
        run<8>("v_trunc_f32", w); run<9>("v_mbcnt_lo", w);
        run<10>("v_add_f32 vv", w); run<11>("v_sub_f32 vv", w); run<12>("v_mul_f32 sv", w); run<13>("v_cmp vcc", w);
        run<14>("v_cmp sgpr", w); run<15>("v_readlane", w); run<16>("v_cmp+s_bcnt", w); run<17>("v_subrev sv", w);
        run<18>("v_add_u32", w); run<19>("v_cndmask", w);
    }
    return 0;
}
