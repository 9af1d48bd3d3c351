// Latency of ONE fit call through the C ABI as a compiled caller (MolAR's Rust side) sees it: no Python, no ctypes.
// C3 shape: M = 1e5 of N = 1e6 atoms, the frame, reference, mass and index columns resident in HBM.
//   g++ -std=c++17 -O2 -I include tools/cpp/bench_fit_single.cpp -o /tmp/bench_fit_single -L molar_amd -lmolar_hip
//       -Wl,-rpath,$PWD/molar_amd -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib -D__HIP_PLATFORM_AMD__ -I /opt/rocm/include
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

#include "molar_hip.h"

int main() {
    const size_t n = 1000000, m = 100000;
    std::mt19937 rng(5);
    std::uniform_real_distribution<float> u(0.f, 21.5f), nz(-0.05f, 0.05f), um(1.f, 16.f);
    std::vector<float> ref(3 * n), cur(3 * n), mass(n);
    for (size_t k = 0; k < 3 * n; ++k) {
        ref[k] = u(rng);
        cur[k] = ref[k] + nz(rng);
    }
    for (auto &x : mass) x = um(rng);
    std::vector<uint64_t> idx(m);
    for (size_t k = 0; k < m; ++k) idx[k] = 10 * k;
    float *d_ref, *d_cur, *d_mass;
    uint64_t *d_idx;
    if (hipMalloc((void **)&d_ref, 12 * n) || hipMalloc((void **)&d_cur, 12 * n) || hipMalloc((void **)&d_mass, 4 * n) ||
        hipMalloc((void **)&d_idx, 8 * m))
        return 2;
    hipMemcpy(d_ref, ref.data(), 12 * n, hipMemcpyHostToDevice);
    hipMemcpy(d_cur, cur.data(), 12 * n, hipMemcpyHostToDevice);
    hipMemcpy(d_mass, mass.data(), 4 * n, hipMemcpyHostToDevice);
    hipMemcpy(d_idx, idx.data(), 8 * m, hipMemcpyHostToDevice);
    molar_hip_ctx *c = molar_hip_create(0);
    if (!c) return 3;
    for (int apply = 0; apply < 2; ++apply) {
        float rmsd, R[9], t[3], com[3], gyr;
        for (int w = 0; w < 50; ++w)
            if (molar_hip_fit_rmsd_batch(c, d_cur, 1, n, d_idx, m, d_mass, d_ref, n, d_idx, apply, &rmsd, R, t, com, &gyr)) return 4;
        const int reps = 2000;
        const auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; ++r)
            if (molar_hip_fit_rmsd_batch(c, d_cur, 1, n, d_idx, m, d_mass, d_ref, n, d_idx, apply, &rmsd, R, t, com, &gyr)) return 4;
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
        std::printf("{\"workload\": \"C3 one frame per call from C++ (no ctypes), M=1e5 of N=1e6, resident\", \"apply_transform\": %s, "
                    "\"us_per_call\": %.2f, \"rmsd\": %.6f}\n", apply ? "true" : "false", us, rmsd);
    }
    molar_hip_destroy(c);
    return 0;
}
