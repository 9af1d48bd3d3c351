#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
__global__ void k(float *o, float aval, float bval) {
    v8h a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.0f; b[i] = (_Float16)0.0f; }
    a[0] = (_Float16)aval;      // k = 0 (lanes < 32) / k = 8 (lanes >= 32)
    b[0] = (_Float16)bval;
    v16f c = {};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) { o[0] = c[0]; o[1] = (float)a[0]; }
}
int main() {
    float *d; hipMalloc(&d, 64);
    const float as[] = {9.5367431640625e-07f /*2^-20 subnormal*/, 3.0517578125e-05f /*2^-15 subnormal*/, 6.103515625e-05f /*2^-14 min normal*/, 5.9604644775390625e-08f /*2^-24 smallest*/};
    for (float av : as) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, av, 1024.0f);
        float h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        printf("a=%.3e (as f16 %.3e) x 1024 x 2 halves -> %.6e (expect %.6e)\n", av, h[1], h[0], 2.0 * h[1] * 1024.0);
    }
    return 0;
}
