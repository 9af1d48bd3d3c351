// linalg3.hpp — tiny dense kernels in double, usable from host code and from one GPU thread.
//  * symmetric Jacobi eigensolver (N = 3 for the inertia tensor, N = 4 for Horn's key matrix)
//  * optimal rotation from a 3x3 mass-weighted covariance via Horn's unit-quaternion method.
//
// The reference obtains the rotation as U diag(1,1,sign det(U V^T)) V^T from nalgebra's SVD
// (molar/src/measure.rs:626-642).  For a non-degenerate covariance that rotation is the unique
// maximiser of tr(R^T cov) over proper rotations, which is exactly what the dominant
// eigenvector of Horn's 4x4 matrix encodes, so both give the same matrix to rounding.
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>

namespace mh {

// Cyclic Jacobi on a symmetric NxN (row-major a[N*N], destroyed).  w = eigenvalues,
// v = eigenvectors as COLUMNS (v[r*N+c]).
template <int N>
__host__ __device__ inline void jacobi_sym(double *a, double *w, double *v) {
    // every loop over matrix indices is unrolled: on the GPU the two small matrices then live in registers (dynamic
    // indexing would put them in scratch memory, ~20 us per call for one lane)
#pragma unroll
    for (int i = 0; i < N * N; ++i) v[i] = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) v[i * N + i] = 1.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0, diag = 0.0;
#pragma unroll
        for (int p = 0; p < N; ++p) {
            diag += a[p * N + p] * a[p * N + p];
#pragma unroll
            for (int q = p + 1; q < N; ++q) off += a[p * N + q] * a[p * N + q];
        }
        // off-diagonal mass below 1e-13 of the diagonal (squared: 1e-26): eigenvectors good to ~1e-13, far below the f32
        // results they feed; every further sweep is ~2 us of dependent f64 divisions and square roots on one GPU lane
        if (off <= 1e-26 * diag || off < 1e-300) break;
#pragma unroll
        for (int p = 0; p < N - 1; ++p)
#pragma unroll
            for (int q = p + 1; q < N; ++q) {
                const double apq = a[p * N + q];
                if (apq == 0.0) continue;
                const double tau = (a[q * N + q] - a[p * N + p]) / (2.0 * apq);
                const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                const double cs = 1.0 / sqrt(1.0 + t * t), sn = t * cs;
#pragma unroll
                for (int k = 0; k < N; ++k) {   // columns p,q
                    const double akp = a[k * N + p], akq = a[k * N + q];
                    a[k * N + p] = cs * akp - sn * akq;
                    a[k * N + q] = sn * akp + cs * akq;
                }
#pragma unroll
                for (int k = 0; k < N; ++k) {   // rows p,q
                    const double apk = a[p * N + k], aqk = a[q * N + k];
                    a[p * N + k] = cs * apk - sn * aqk;
                    a[q * N + k] = sn * apk + cs * aqk;
                }
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    const double vkp = v[k * N + p], vkq = v[k * N + q];
                    v[k * N + p] = cs * vkp - sn * vkq;
                    v[k * N + q] = sn * vkp + cs * vkq;
                }
            }
    }
#pragma unroll
    for (int i = 0; i < N; ++i) w[i] = a[i * N + i];
}

// cov(r,c) = sum m * q2[r] * q1[c]  (measure.rs:621-623), column-major cov[c*3+r].
// Writes R (column-major) with q2 ~ R q1.  Returns false if cov holds a NaN.
__host__ __device__ inline bool rotation_from_cov(const double *cov, double *R) {
    for (int i = 0; i < 9; ++i)
        if (cov[i] != cov[i]) return false;
    // S[a][b] = sum m q1[a] q2[b] = cov(b,a)
    const double Sxx = cov[0 * 3 + 0], Sxy = cov[0 * 3 + 1], Sxz = cov[0 * 3 + 2];
    const double Syx = cov[1 * 3 + 0], Syy = cov[1 * 3 + 1], Syz = cov[1 * 3 + 2];
    const double Szx = cov[2 * 3 + 0], Szy = cov[2 * 3 + 1], Szz = cov[2 * 3 + 2];
    double Nm[16] = {Sxx + Syy + Szz, Syz - Szy,       Szx - Sxz,        Sxy - Syx,
                     Syz - Szy,       Sxx - Syy - Szz, Sxy + Syx,        Szx + Sxz,
                     Szx - Sxz,       Sxy + Syx,       -Sxx + Syy - Szz, Syz + Szy,
                     Sxy - Syx,       Szx + Sxz,       Syz + Szy,        -Sxx - Syy + Szz};
    double w[4], v[16];
    jacobi_sym<4>(Nm, w, v);
    // eigenvector of the largest eigenvalue (first one wins ties); selected with static indices, see jacobi_sym
    double wb = w[0], q0 = v[0], qx = v[4], qy = v[8], qz = v[12];
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (w[i] > wb) {
            wb = w[i];
            q0 = v[0 * 4 + i]; qx = v[1 * 4 + i]; qy = v[2 * 4 + i]; qz = v[3 * 4 + i];
        }
    const double nq = sqrt(q0 * q0 + qx * qx + qy * qy + qz * qz);
    q0 /= nq; qx /= nq; qy /= nq; qz /= nq;
    // column-major R
    R[0] = 1.0 - 2.0 * (qy * qy + qz * qz);
    R[1] = 2.0 * (qx * qy + q0 * qz);
    R[2] = 2.0 * (qx * qz - q0 * qy);
    R[3] = 2.0 * (qx * qy - q0 * qz);
    R[4] = 1.0 - 2.0 * (qx * qx + qz * qz);
    R[5] = 2.0 * (qy * qz + q0 * qx);
    R[6] = 2.0 * (qx * qz + q0 * qy);
    R[7] = 2.0 * (qy * qz - q0 * qx);
    R[8] = 1.0 - 2.0 * (qx * qx + qy * qy);
    return true;
}

}  // namespace mh
