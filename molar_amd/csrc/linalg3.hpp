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
__host__ __device__ inline void jacobi_sym(double *a, double *w, double *v, double tol2 = 1e-26) {
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
        // (tol2: callers with f64 results ask for 1e-34 - one or two sweeps more, the convergence is quadratic)
        if (off <= tol2 * diag || off < 1e-300) break;
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


// 3x3 determinant of the rows (r0,r1,r2) x columns (c0,c1,c2) of a row-major 4x4
__host__ __device__ inline double minor3(const double *a, int r0, int r1, int r2, int c0, int c1, int c2) {
    return a[r0 * 4 + c0] * (a[r1 * 4 + c1] * a[r2 * 4 + c2] - a[r1 * 4 + c2] * a[r2 * 4 + c1]) -
           a[r0 * 4 + c1] * (a[r1 * 4 + c0] * a[r2 * 4 + c2] - a[r1 * 4 + c2] * a[r2 * 4 + c0]) +
           a[r0 * 4 + c2] * (a[r1 * 4 + c0] * a[r2 * 4 + c1] - a[r1 * 4 + c1] * a[r2 * 4 + c0]);
}

// Dominant eigenvector of Horn's symmetric, traceless 4x4 matrix K without sweeping the whole spectrum: the largest root
// of the characteristic polynomial l^4 + c2 l^2 + c1 l + c0 (c2 = -tr(K^2)/2, c1 = -tr(K^3)/3, c0 = det K) by Newton's
// iteration from the Frobenius norm - an upper bound of every eigenvalue, and from above the iteration on a polynomial
// with only real roots descends monotonically onto the largest one - then a null vector of K - l I as the largest
// column of its adjugate.  ~4 us for one GPU lane against ~20 us for the Jacobi sweeps.  Returns false (the caller falls
// back to Jacobi) when the root is not simple to working precision: the residual |(K - l I) q| is checked, so a
// vector that passes IS the eigenvector to 1e-10 relative.
__host__ __device__ inline bool horn_dominant_eigenvector(const double *K, double &q0, double &qx, double &qy, double &qz) {
    double K2[16], f2 = 0.0, t3 = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k) s += K[i * 4 + k] * K[k * 4 + j];
            K2[i * 4 + j] = s;
        }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        f2 += K[i] * K[i];
        t3 += K2[i] * K[i];          // tr(K^3) = sum_ij (K^2)_ij K_ji, K symmetric
    }
    if (!(f2 > 0.0)) return false;
    const double det = K[0] * minor3(K, 1, 2, 3, 1, 2, 3) - K[1] * minor3(K, 1, 2, 3, 0, 2, 3) +
                       K[2] * minor3(K, 1, 2, 3, 0, 1, 3) - K[3] * minor3(K, 1, 2, 3, 0, 1, 2);
    const double c2 = -0.5 * f2, c1 = -t3 / 3.0, c0 = det;
    double l = sqrt(f2);
    bool converged = false;
    for (int it = 0; it < 60; ++it) {
        const double l2 = l * l;
        const double P = (l2 + c2) * l2 + (c1 * l + c0);
        const double dP = (4.0 * l2 + 2.0 * c2) * l + c1;
        if (!(dP > 0.0)) break;                   // at or beyond a multiple root
        const double step = P / dP;
        l -= step;
        if (fabs(step) <= 1e-15 * fabs(l)) {
            converged = true;
            break;
        }
    }
    if (!converged) return false;
    double A[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) A[i] = K[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) A[i * 4 + i] -= l;
    // adjugate of the symmetric A, column by column: adj[i][j] = (-1)^(i+j) * minor(row j, column i removed)
    double best = -1.0;
    q0 = qx = qy = qz = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r0 = j == 0 ? 1 : 0, r1 = j <= 1 ? 2 : 1, r2 = j <= 2 ? 3 : 2;     // rows without j
        const double v0 = minor3(A, r0, r1, r2, 1, 2, 3), v1 = minor3(A, r0, r1, r2, 0, 2, 3);
        const double v2 = minor3(A, r0, r1, r2, 0, 1, 3), v3 = minor3(A, r0, r1, r2, 0, 1, 2);
        const double sg = (j & 1) ? -1.0 : 1.0;
        const double e0 = sg * v0, e1 = -sg * v1, e2 = sg * v2, e3 = -sg * v3;
        const double nn = (e0 * e0 + e1 * e1) + (e2 * e2 + e3 * e3);
        if (nn > best) {
            best = nn;
            q0 = e0; qx = e1; qy = e2; qz = e3;
        }
    }
    if (!(best > 0.0)) return false;
    double res = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double r = ((A[i * 4 + 0] * q0 + A[i * 4 + 1] * qx) + A[i * 4 + 2] * qy) + A[i * 4 + 3] * qz;
        res += r * r;
    }
    return res <= 1e-20 * f2 * best;
}

// cov(r,c) = sum m * q2[r] * q1[c]  (measure.rs:621-623), column-major cov[c*3+r].
// Writes R (column-major) with q2 ~ R q1.  Returns false if cov holds a non-finite entry (NaN or +-inf: the callers
// report MOLAR_HIP_ERR_SVD) or if no unit quaternion comes out of the eigen-solve.
// precise: Jacobi only, swept to f64 working precision (the f64 Measure entries; the Newton path accepts an eigenvector
// with a 1e-10 relative residual, ample for f32 records only).
__host__ __device__ inline bool rotation_from_cov(const double *cov, double *R, bool precise = false) {
    for (int i = 0; i < 9; ++i)
        if (!(fabs(cov[i]) <= 1.7976931348623157e308)) return false;      // NaN or infinity
    // S[a][b] = sum m q1[a] q2[b] = cov(b,a)
    const double Sxx = cov[0 * 3 + 0], Sxy = cov[0 * 3 + 1], Sxz = cov[0 * 3 + 2];
    const double Syx = cov[1 * 3 + 0], Syy = cov[1 * 3 + 1], Syz = cov[1 * 3 + 2];
    const double Szx = cov[2 * 3 + 0], Szy = cov[2 * 3 + 1], Szz = cov[2 * 3 + 2];
    double Nm[16] = {Sxx + Syy + Szz, Syz - Szy,       Szx - Sxz,        Sxy - Syx,
                     Syz - Szy,       Sxx - Syy - Szz, Sxy + Syx,        Szx + Sxz,
                     Szx - Sxz,       Sxy + Syx,       -Sxx + Syy - Szz, Syz + Szy,
                     Sxy - Syx,       Szx + Sxz,       Syz + Szy,        -Sxx - Syy + Szz};
    double q0, qx, qy, qz;
    if (precise || !horn_dominant_eigenvector(Nm, q0, qx, qy, qz)) {
    double w[4], v[16];
    jacobi_sym<4>(Nm, w, v, precise ? 1e-34 : 1e-26);
    // eigenvector of the largest eigenvalue (first one wins ties); selected with static indices, see jacobi_sym
    double wb = w[0];
    q0 = v[0]; qx = v[4]; qy = v[8]; qz = v[12];
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (w[i] > wb) {
            wb = w[i];
            q0 = v[0 * 4 + i]; qx = v[1 * 4 + i]; qy = v[2 * 4 + i]; qz = v[3 * 4 + i];
        }
    }
    const double nq = sqrt(q0 * q0 + qx * qx + qy * qy + qz * qz);
    if (!(nq > 0.0) || !(nq <= 1.7976931348623157e308)) return false;
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
