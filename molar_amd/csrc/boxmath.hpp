// boxmath.hpp — PeriodicBox arithmetic shared by host API code and gfx950 kernels.
//
// Restates molar/src/periodic_box.rs in f32 with the reference's operation order; the whole
// library is compiled with -ffp-contract=off so no a*b+c here becomes an FMA (Rust never
// contracts).  nalgebra's 3-vector kernels as used by the reference:
//   M*v   : y_r = ((M_r0*v0) + M_r1*v1) + M_r2*v2
//   |v|^2 : ((x*x) + (y*y)) + (z*z)
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>

#include "../../include/molar_hip.h"

namespace mh {

struct V3 {
    float x, y, z;
};

#define MH_HD __host__ __device__ __forceinline__

MH_HD V3 v3(float x, float y, float z) { return V3{x, y, z}; }
MH_HD V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
MH_HD V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
MH_HD float norm2(V3 v) { return (v.x * v.x + v.y * v.y) + v.z * v.z; }

// column-major 3x3: element (r,c) = m[c*3+r]
MH_HD V3 mat_vec(const float *m, V3 v) {
    return V3{(m[0] * v.x + m[3] * v.y) + m[6] * v.z, (m[1] * v.x + m[4] * v.y) + m[7] * v.z,
              (m[2] * v.x + m[5] * v.y) + m[8] * v.z};
}

// Rust f32::round — half away from zero
MH_HD float round_away(float x) {
#ifdef __HIP_DEVICE_COMPILE__
    return __builtin_roundf(x);
#else
    return std::round(x);
#endif
}

// Rust f32::fract = x - trunc(x)
MH_HD float fract_rs(float x) {
#ifdef __HIP_DEVICE_COMPILE__
    return x - __builtin_truncf(x);
#else
    return x - std::trunc(x);
#endif
}

// Rust `f as usize` followed by .clamp(0, hi): saturating cast, NaN -> 0  (distance_search.rs:176-177)
MH_HD uint32_t floor_to_cell(float v, uint32_t dim) {
#ifdef __HIP_DEVICE_COMPILE__
    float f = __builtin_floorf(v);
#else
    float f = std::floor(v);
#endif
    if (!(f > 0.0f)) return 0u;
    if (f >= (float)dim) return dim - 1u;
    uint32_t u = (uint32_t)f;
    return u > dim - 1u ? dim - 1u : u;
}

// periodic_box.rs:286-318.  `pbc` is the PbcDims byte; the triclinic candidate loop runs only for
// a non-empty shift list AND pbc == PBC_FULL (:304).
MH_HD V3 shortest_vector(const molar_hip_box &b, V3 v, uint32_t pbc) {
    V3 f = mat_vec(b.inv, v);
    if (pbc & 1u) f.x -= round_away(f.x);
    if (pbc & 2u) f.y -= round_away(f.y);
    if (pbc & 4u) f.z -= round_away(f.z);
    V3 start = mat_vec(b.m, f);
    if (b.nshift == 0 || pbc != MOLAR_HIP_PBC_FULL) return start;
    V3 best = start;
    float best2 = norm2(start);
    for (int k = 0; k < b.nshift; ++k) {
        V3 cand = start + V3{b.shifts[3 * k], b.shifts[3 * k + 1], b.shifts[3 * k + 2]};
        float n2 = norm2(cand);
        if (n2 < best2) {
            best2 = n2;
            best = cand;
        }
    }
    return best;
}

// periodic_box.rs:322-330
MH_HD V3 closest_image(const molar_hip_box &b, V3 p, V3 target, uint32_t pbc) {
    return target + shortest_vector(b, p - target, pbc);
}

}  // namespace mh
