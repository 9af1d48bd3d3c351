// boxmath64.hpp - PeriodicBox in double precision (MolAR built with its `f64` feature: Float = f64, aliases.rs:10-13),
// shared by the f64 Measure / Modify entries (measure_f64.hip) and the f64 search drivers (search_f64.hip).
#pragma once

#include <cmath>
#include <cstring>

#include "common.hpp"

namespace mh {

// PeriodicBox in f64 (periodic_box.rs:15-23 with Float = f64): the same construction and the same shortest_vector as
// boxmath.hpp / api.hip, every operation in double.
struct BoxD {
    double m[9];       // column-major, columns a, b, c
    double inv[9];     // nalgebra try_inverse
    int32_t nshift;    // tric_corrections.len()
    double shifts[26 * 3];
};
struct D3 {
    double x, y, z;
};
#define MH64_HD __host__ __device__ __forceinline__
MH64_HD D3 operator+(D3 a, D3 b) { return D3{a.x + b.x, a.y + b.y, a.z + b.z}; }
MH64_HD D3 operator-(D3 a, D3 b) { return D3{a.x - b.x, a.y - b.y, a.z - b.z}; }
MH64_HD double norm2(D3 v) { return (v.x * v.x + v.y * v.y) + v.z * v.z; }
MH64_HD D3 mat_vec(const double *m, D3 v) {
    return D3{(m[0] * v.x + m[3] * v.y) + m[6] * v.z, (m[1] * v.x + m[4] * v.y) + m[7] * v.z,
              (m[2] * v.x + m[5] * v.y) + m[8] * v.z};
}
MH64_HD double round_away(double x) {          // Rust f64::round - half away from zero
#ifdef __HIP_DEVICE_COMPILE__
    return __builtin_round(x);
#else
    return std::round(x);
#endif
}
// shortest_vector_dims (periodic_box.rs:286-318)
MH64_HD D3 shortest_vector(const BoxD &b, D3 v, uint32_t pbc) {
    D3 f = mat_vec(b.inv, v);
    if (pbc & 1u) f.x -= round_away(f.x);
    if (pbc & 2u) f.y -= round_away(f.y);
    if (pbc & 4u) f.z -= round_away(f.z);
    const D3 start = mat_vec(b.m, f);
    if (b.nshift == 0 || pbc != MOLAR_HIP_PBC_FULL) return start;
    D3 best = start;
    double best2 = norm2(start);
    for (int k = 0; k < b.nshift; ++k) {
        const D3 cand = start + D3{b.shifts[3 * k], b.shifts[3 * k + 1], b.shifts[3 * k + 2]};
        const double n2 = norm2(cand);
        if (n2 < best2) {
            best2 = n2;
            best = cand;
        }
    }
    return best;
}
// closest_image_dims (:322-330)
MH64_HD D3 closest_image(const BoxD &b, D3 p, D3 target, uint32_t pbc) { return target + shortest_vector(b, p - target, pbc); }

// PeriodicBox::from_matrix (:156-176) + build_tric_corrections (:25-66)
inline int box64_from_matrix(const double *m9, BoxD *out) {
    if (!m9) return fail(MOLAR_HIP_ERR_NO_PBC, "pbc operation without periodic box");
    D3 col[3];
    for (int k = 0; k < 3; ++k) {
        col[k] = D3{m9[3 * k], m9[3 * k + 1], m9[3 * k + 2]};
        if (std::sqrt(norm2(col[k])) == 0.0) return fail(MOLAR_HIP_ERR_ZERO_LENGTH_VECTOR, "zero length box vector");
    }
    std::memcpy(out->m, m9, sizeof out->m);
    {   // nalgebra try_inverse, 3x3 closed form
        const double *m = out->m;
        double *o = out->inv;
        const double a = m[0], d = m[1], g = m[2], b = m[3], e = m[4], h = m[5], c = m[6], f = m[7], i = m[8];
        const double minor_bf = e * i - h * f, minor_af = d * i - g * f, minor_ae = d * h - g * e;
        const double det = (a * minor_bf - b * minor_af) + c * minor_ae;
        if (det == 0.0) return fail(MOLAR_HIP_ERR_INVERSE_FAILED, "box matrix inverse failed");
        o[0] = minor_bf / det;  o[3] = (c * h - i * b) / det;  o[6] = (b * f - e * c) / det;
        o[1] = -minor_af / det; o[4] = (a * i - g * c) / det;  o[7] = (c * d - f * a) / det;
        o[2] = minor_ae / det;  o[5] = (b * g - h * a) / det;  o[8] = (a * e - d * b) / det;
    }
    out->nshift = 0;
    const bool ortho = m9[3] == 0.0 && m9[6] == 0.0 && m9[1] == 0.0 && m9[7] == 0.0 && m9[2] == 0.0 && m9[5] == 0.0;
    if (ortho) return 0;
    const D3 a = col[0], b = col[1], c = col[2], na = D3{-a.x, -a.y, -a.z};
    auto len = [](D3 v) { return std::sqrt(norm2(v)); };
    const double longest = std::fmax(std::fmax(std::fmax(len((a + b) + c), len((a + b) - c)), len((a - b) + c)), len((na + b) + c));
    const double half_diag = 0.5 * longest, two = 2.0 * half_diag, bound2 = two * two;
    for (int i = -1; i <= 1; ++i)
        for (int j = -1; j <= 1; ++j)
            for (int k = -1; k <= 1; ++k) {
                if (!i && !j && !k) continue;
                const double fi = i, fj = j, fk = k;
                const D3 sft = (D3{fi * a.x, fi * a.y, fi * a.z} + D3{fj * b.x, fj * b.y, fj * b.z}) + D3{fk * c.x, fk * c.y, fk * c.z};
                if (norm2(sft) < bound2) {
                    double *dst = out->shifts + 3 * out->nshift++;
                    dst[0] = sft.x; dst[1] = sft.y; dst[2] = sft.z;
                }
            }
    return 0;
}


}  // namespace mh
