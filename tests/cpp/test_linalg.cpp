// The rotation solver of the fit path (molar_amd/csrc/linalg3.hpp) on the host: the Newton + adjugate route to the
// dominant eigenvector of Horn's matrix against the Jacobi eigensolver, on well-conditioned, badly scaled, nearly
// rank-one, nearly-identity and degenerate covariances; and the rotation itself against known rotations.
// Built with hipcc (the header is shared with the kernels), runs without a GPU.
#include "../../molar_amd/csrc/linalg3.hpp"

#include <cstdio>
#include <random>

using namespace mh;

static void horn_matrix(const double *cov, double *N) {
    const double Sxx = cov[0], Sxy = cov[1], Sxz = cov[2], Syx = cov[3], Syy = cov[4], Syz = cov[5], Szx = cov[6], Szy = cov[7], Szz = cov[8];
    const double M[16] = {Sxx + Syy + Szz, Syz - Szy, Szx - Sxz, Sxy - Syx, Syz - Szy, Sxx - Syy - Szz, Sxy + Syx, Szx + Sxz,
                          Szx - Sxz, Sxy + Syx, -Sxx + Syy - Szz, Syz + Szy, Sxy - Syx, Szx + Sxz, Syz + Szy, -Sxx - Syy + Szz};
    for (int i = 0; i < 16; ++i) N[i] = M[i];
}

static void rotation_jacobi_only(const double *cov, double *R) {
    double N[16], w[4], v[16];
    horn_matrix(cov, N);
    jacobi_sym<4>(N, w, v);
    int b = 0;
    for (int i = 1; i < 4; ++i)
        if (w[i] > w[b]) b = i;
    double q0 = v[b], qx = v[4 + b], qy = v[8 + b], qz = v[12 + b];
    const double nq = std::sqrt(q0 * q0 + qx * qx + qy * qy + qz * qz);
    q0 /= nq; qx /= nq; qy /= nq; qz /= nq;
    R[0] = 1.0 - 2.0 * (qy * qy + qz * qz); R[1] = 2.0 * (qx * qy + q0 * qz); R[2] = 2.0 * (qx * qz - q0 * qy);
    R[3] = 2.0 * (qx * qy - q0 * qz); R[4] = 1.0 - 2.0 * (qx * qx + qz * qz); R[5] = 2.0 * (qy * qz + q0 * qx);
    R[6] = 2.0 * (qx * qz + q0 * qy); R[7] = 2.0 * (qy * qz - q0 * qx); R[8] = 1.0 - 2.0 * (qx * qx + qy * qy);
}

int main() {
    std::mt19937_64 g(20240607);
    std::normal_distribution<double> n(0, 1);
    int failures = 0, fast = 0, total = 0;
    double worst = 0.0;
    for (int t = 0; t < 100000; ++t) {
        double cov[9];
        const int kind = t % 6;
        for (int i = 0; i < 9; ++i) cov[i] = n(g) * (kind == 1 ? 1e6 : (kind == 5 ? 1e-9 : 1.0));
        if (kind == 2) {             // nearly rank one
            const double a[3] = {n(g), n(g), n(g)}, b[3] = {n(g), n(g), n(g)};
            for (int c = 0; c < 3; ++c)
                for (int r = 0; r < 3; ++r) cov[c * 3 + r] = a[r] * b[c] + 1e-9 * n(g);
        }
        if (kind == 3) {             // nearly identity fit
            for (int i = 0; i < 9; ++i) cov[i] = 1e-6 * n(g);
            cov[0] += std::fabs(n(g)) + 0.1; cov[4] += std::fabs(n(g)) + 0.1; cov[8] += std::fabs(n(g)) + 0.1;
        }
        if (kind == 4) {             // two equal singular values: the maximiser is not unique
            for (int i = 0; i < 9; ++i) cov[i] = 0;
            cov[0] = 1; cov[4] = 1; cov[8] = 0.3 * n(g);
        }
        double R1[9], R2[9], N[16], q0, qx, qy, qz;
        if (!rotation_from_cov(cov, R1)) { ++failures; continue; }
        rotation_jacobi_only(cov, R2);
        horn_matrix(cov, N);
        fast += horn_dominant_eigenvector(N, q0, qx, qy, qz) ? 1 : 0;
        ++total;
        // both are proper rotations
        double ortho = 0.0;
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) {
                double s = 0.0;
                for (int k = 0; k < 3; ++k) s += R1[a * 3 + k] * R1[b * 3 + k];
                ortho = std::fmax(ortho, std::fabs(s - (a == b ? 1.0 : 0.0)));
            }
        if (ortho > 1e-12) ++failures;
        if (kind == 2 || kind == 4) {                 // compare the objective tr(R^T cov), which is unique
            double o1 = 0, o2 = 0, sc = 0;
            for (int i = 0; i < 9; ++i) { o1 += R1[i] * cov[i]; o2 += R2[i] * cov[i]; sc += std::fabs(cov[i]); }
            if (std::fabs(o1 - o2) > 1e-9 * (sc + 1e-300)) ++failures;
        } else {
            double e = 0.0;
            for (int i = 0; i < 9; ++i) e = std::fmax(e, std::fabs(R1[i] - R2[i]));
            worst = std::fmax(worst, e);
            if (e > 1e-9) ++failures;
        }
    }
    // a known rotation is recovered: cov = sum m (R p) p^T
    for (int t = 0; t < 1000; ++t) {
        double ax[3] = {n(g), n(g), n(g)};
        const double na = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
        for (double &a : ax) a /= na;
        const double ang = 3.0 * n(g), c = std::cos(ang), s = std::sin(ang), k = 1 - c;
        const double Rt[9] = {ax[0] * ax[0] * k + c, ax[0] * ax[1] * k + ax[2] * s, ax[0] * ax[2] * k - ax[1] * s,
                              ax[0] * ax[1] * k - ax[2] * s, ax[1] * ax[1] * k + c, ax[1] * ax[2] * k + ax[0] * s,
                              ax[0] * ax[2] * k + ax[1] * s, ax[1] * ax[2] * k - ax[0] * s, ax[2] * ax[2] * k + c};   // column-major
        double cov[9] = {0};
        for (int a = 0; a < 20; ++a) {
            const double p[3] = {n(g), n(g), n(g)}, m = 1.0 + std::fabs(n(g));
            double q[3];
            for (int r = 0; r < 3; ++r) q[r] = Rt[0 * 3 + r] * p[0] + Rt[1 * 3 + r] * p[1] + Rt[2 * 3 + r] * p[2];
            for (int cc = 0; cc < 3; ++cc)
                for (int r = 0; r < 3; ++r) cov[cc * 3 + r] += m * q[r] * p[cc];
        }
        double R[9];
        rotation_from_cov(cov, R);
        for (int i = 0; i < 9; ++i)
            if (std::fabs(R[i] - Rt[i]) > 1e-10) { ++failures; break; }
    }
    std::printf("fast path on %d of %d covariances, worst |dR| vs Jacobi %.2e, %d failure(s)\n", fast, total, worst, failures);
    if (failures == 0 && fast > total / 2) std::printf("all linalg tests passed\n");
    return failures ? 1 : 0;
}
