// measure.hip — MolAR's Measure / Modify numeric methods (molar/src/measure.rs, modify.rs) for gfx950.
//
// Every reduction gathers through the selection index (providers.rs:103-106), forms each
// per-atom term in f32 with the reference's operation order (p - c, shortest_vector, R*p + t)
// and accumulates the terms in f64: per-thread -> wave shuffle -> one partial per workgroup,
// summed in a fixed order by the finalize step, so results are deterministic and agree with the
// reference's serial f32 sums to better than the f32 roundoff those sums carry themselves.
// These passes move 12-44 bytes per atom and a handful of flops: HBM-bound, no MFMA.
#include <algorithm>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "boxmath.hpp"
#include "common.hpp"
#include "stages.hpp"
#include "linalg3.hpp"

using namespace mh;

namespace {

constexpr int RB = 256;   // reduction block

struct Sel {
    const float *xyz;
    const uint64_t *idx;
    const float *mass;   // full-length column, gathered through idx
    uint32_t n;
    size_t frame_stride; // floats between consecutive frames (batched calls), 0 otherwise
};

__device__ __forceinline__ uint64_t atom_of(const Sel &s, uint32_t k) { return s.idx ? s.idx[k] : (uint64_t)k; }
__device__ __forceinline__ V3 pos_of(const Sel &s, uint32_t frame, uint64_t a) {
    const float *q = s.xyz + (size_t)frame * s.frame_stride + 3 * a;
    return v3(q[0], q[1], q[2]);
}

// block-wide sum of NV doubles; lanes 0..NV-1 of wave 0 write partials[(frame*gridDim.x+block)*NV + v]
template <int NV, int BLOCK = RB>
__device__ __forceinline__ void block_reduce_store(double *acc, double *partials) {
    __shared__ double sh[BLOCK / 64][NV];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        double x = acc[v];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
        if (lane == 0) sh[wave][v] = x;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        double s = 0.0;
        for (int w = 0; w < BLOCK / 64; ++w) s += sh[w][threadIdx.x];
        partials[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * NV + threadIdx.x] = s;
    }
}

// ---------------------------------------------------------------- pass kernels

// [0]=sum m, [1..3]=sum p*m, [4..6]=sum p   (center_of_mass :60-75, center_of_geometry :39-47)
__global__ void __launch_bounds__(RB) k_sums(Sel s, double *partials) {
    double acc[7] = {0, 0, 0, 0, 0, 0, 0};
    for (uint32_t k = blockIdx.x * RB + threadIdx.x; k < s.n; k += gridDim.x * RB) {
        const uint64_t a = atom_of(s, k);
        const V3 p = pos_of(s, blockIdx.y, a);
        const float m = s.mass ? s.mass[a] : 1.0f;
        acc[0] += (double)m;
        acc[1] += (double)(p.x * m);
        acc[2] += (double)(p.y * m);
        acc[3] += (double)(p.z * m);
        acc[4] += (double)p.x;
        acc[5] += (double)p.y;
        acc[6] += (double)p.z;
    }
    block_reduce_store<7>(acc, partials);
}

// images relative to the first selected atom (center_of_*_pbc_dims :156-168, :197-220):
// [0]=sum m (k>=1), [1..3]=sum img*m (k>=1), [4..6]=sum img (k>=1)
__global__ void __launch_bounds__(RB) k_sums_pbc(Sel s, molar_hip_box box, uint32_t pbc, double *partials) {
    double acc[7] = {0, 0, 0, 0, 0, 0, 0};
    const V3 p0 = pos_of(s, blockIdx.y, atom_of(s, 0));
    for (uint32_t k = 1 + blockIdx.x * RB + threadIdx.x; k < s.n; k += gridDim.x * RB) {
        const uint64_t a = atom_of(s, k);
        const V3 im = closest_image(box, pos_of(s, blockIdx.y, a), p0, pbc);
        const float m = s.mass ? s.mass[a] : 1.0f;
        acc[0] += (double)m;
        acc[1] += (double)(im.x * m);
        acc[2] += (double)(im.y * m);
        acc[3] += (double)(im.z * m);
        acc[4] += (double)im.x;
        acc[5] += (double)im.y;
        acc[6] += (double)im.z;
    }
    block_reduce_store<7>(acc, partials);
}

// central moments about c (do_gyration :561-570, do_inertia :573-586); d = p - c, or
// shortest_vector(p - c) for the *_pbc variants (:229, :241).
// [0]=sum m, [1]=sum |d|^2 m, [2..4]=T00,T11,T22, [5..7]=T01,T02,T12 (already negated)
__global__ void __launch_bounds__(RB) k_central(Sel s, const float *center, int use_box, molar_hip_box box,
                                                double *partials) {
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const V3 c = v3(center[3 * blockIdx.y], center[3 * blockIdx.y + 1], center[3 * blockIdx.y + 2]);
    for (uint32_t k = blockIdx.x * RB + threadIdx.x; k < s.n; k += gridDim.x * RB) {
        const uint64_t a = atom_of(s, k);
        V3 d = pos_of(s, blockIdx.y, a) - c;
        if (use_box) d = shortest_vector(box, d, MOLAR_HIP_PBC_FULL);
        const float m = s.mass[a];
        acc[0] += (double)m;
        acc[1] += (double)(norm2(d) * m);
        acc[2] += (double)(m * (d.y * d.y + d.z * d.z));
        acc[3] += (double)(m * (d.x * d.x + d.z * d.z));
        acc[4] += (double)(m * (d.x * d.x + d.y * d.y));
        acc[5] -= (double)(m * d.x * d.y);
        acc[6] -= (double)(m * d.x * d.z);
        acc[7] -= (double)(m * d.y * d.z);
    }
    block_reduce_store<8>(acc, partials);
}

// uncentred moments for the non-periodic gyration / inertia in ONE pass (exact products, f64 sums):
// [0]=S m, [1..3]=S m p, [4..9]=S m xx, yy, zz, xy, xz, yz.  The host centres them on the f32 centre of mass:
// S m (p-c)(p-c)^T = S m p p^T - c (S m p)^T - (S m p) c^T + S m c c^T for any c.
__global__ void __launch_bounds__(RB) k_moments(Sel s, double *partials) {
    double acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t k = blockIdx.x * RB + threadIdx.x; k < s.n; k += gridDim.x * RB) {
        const uint64_t a = atom_of(s, k);
        const V3 pf = pos_of(s, blockIdx.y, a);
        const double x = pf.x, y = pf.y, z = pf.z, m = (double)s.mass[a];
        acc[0] += m;
        acc[1] += m * x; acc[2] += m * y; acc[3] += m * z;
        acc[4] += m * (x * x); acc[5] += m * (y * y); acc[6] += m * (z * z);
        acc[7] += m * (x * y); acc[8] += m * (x * z); acc[9] += m * (y * z);
    }
    block_reduce_store<10>(acc, partials);
}

// two selections: [0]=sum |p2-p1|^2 (rmsd :499-501), [1]=sum |p2-p1|^2 m, [2]=sum m (rmsd_mw :548-551)
__global__ void __launch_bounds__(RB) k_rmsd(Sel s1, Sel s2, double *partials) {
    double acc[3] = {0, 0, 0};
    for (uint32_t k = blockIdx.x * RB + threadIdx.x; k < s1.n; k += gridDim.x * RB) {
        const uint64_t a1 = atom_of(s1, k), a2 = atom_of(s2, k);
        const V3 v = pos_of(s2, 0, a2) - pos_of(s1, blockIdx.y, a1);
        const float d2 = norm2(v);
        const float m = s1.mass ? s1.mass[a1] : 1.0f;
        acc[0] += (double)d2;
        acc[1] += (double)(d2 * m);
        acc[2] += (double)m;
    }
    block_reduce_store<3>(acc, partials);
}

// rot_transform accumulation (:619-623): cov(r,c) += (q2[r]*q1[c])*m, q = p - centre, m from sel1.
// centres: c1[frame] and c2 (shared reference).  partial layout column-major cov[c*3+r].
__global__ void __launch_bounds__(RB) k_cov(Sel s1, Sel s2, const float *c1v, const float *c2v, double *partials) {
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const V3 c1 = v3(c1v[3 * blockIdx.y], c1v[3 * blockIdx.y + 1], c1v[3 * blockIdx.y + 2]);
    const V3 c2 = v3(c2v[0], c2v[1], c2v[2]);
    for (uint32_t k = blockIdx.x * RB + threadIdx.x; k < s1.n; k += gridDim.x * RB) {
        const uint64_t a1 = atom_of(s1, k), a2 = atom_of(s2, k);
        const V3 q1 = pos_of(s1, blockIdx.y, a1) - c1;
        const V3 q2 = pos_of(s2, 0, a2) - c2;
        const float m = s1.mass[a1];
        acc[0] += (double)((q2.x * q1.x) * m);
        acc[1] += (double)((q2.y * q1.x) * m);
        acc[2] += (double)((q2.z * q1.x) * m);
        acc[3] += (double)((q2.x * q1.y) * m);
        acc[4] += (double)((q2.y * q1.y) * m);
        acc[5] += (double)((q2.z * q1.y) * m);
        acc[6] += (double)((q2.x * q1.z) * m);
        acc[7] += (double)((q2.y * q1.z) * m);
        acc[8] += (double)((q2.z * q1.z) * m);
    }
    block_reduce_store<9>(acc, partials);
}

// ---------------------------------------------------------------- fused fit pass
// ONE gather pass gives everything the per-frame loop of a fit needs (fit_transform measure.rs:507-522, rot_transform
// :613-643, rmsd :485-504, center_of_mass :60-75, gyration :78-87 of the fitted selection): uncentred sums in f64,
//   mass-weighted (m from sel1, :621)   [0] S m   [1..3] S m p   [4..6] S m q   [7..15] S m q_r p_c (column-major c*3+r)
//                                       [16] S m |p|^2
//   reference's own masses (cm2, :512)  [17] S m2   [18..20] S m2 q
//   unweighted (rmsd is, :499-501)      [21..23] S p   [24..26] S q   [27..35] S q_r p_c   [36] S |p|^2   [37] S |q|^2
// p = atom of the current frame, q = atom of the reference.  The centred covariance, the RMSD after the fit and the
// gyration radius follow from these analytically in the finalizer; products of two f32 are exact in f64 and the sums
// keep ~1e-16 relative error, so the cancellation in "uncentred minus centre terms" (<= 1e6 : 1 for MD boxes) costs nothing
// that the reference's serial f32 sums do not lose a million times over.
constexpr int FS_W = 21, FS_ALL = 38;

template <bool UNW, int NV>
__device__ __forceinline__ void fit_accumulate(double (&acc)[NV], V3 pf, V3 qf, double m, double m2) {
        const double p[3] = {pf.x, pf.y, pf.z}, q[3] = {qf.x, qf.y, qf.z};
        const double pp = (p[0] * p[0] + p[1] * p[1]) + p[2] * p[2];
        acc[0] += m;
        acc[16] += m * pp;
        acc[17] += m2;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            acc[1 + d] += m * p[d];
            acc[4 + d] += m * q[d];
            acc[18 + d] += m2 * q[d];
#pragma unroll
            for (int r = 0; r < 3; ++r) acc[7 + d * 3 + r] += (m * q[r]) * p[d];
        }
        if (UNW) {
            acc[36] += pp;
            acc[37] += (q[0] * q[0] + q[1] * q[1]) + q[2] * q[2];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                acc[21 + d] += p[d];
                acc[24 + d] += q[d];
#pragma unroll
                for (int r = 0; r < 3; ++r) acc[27 + d * 3 + r] += q[r] * p[d];
            }
        }
}

template <bool UNW>
__global__ void __launch_bounds__(RB) k_fit_sums(Sel s1, Sel s2, double *partials) {
    constexpr int BLOCK = RB;
    constexpr int NV = UNW ? FS_ALL : FS_W;
    double acc[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = 0.0;
    // four atoms per trip, every level of the dependent chain (index -> position, mass) issued for all four before any
    // is consumed: a thread's latency is one chain, not one chain per atom
    const uint32_t stride = gridDim.x * BLOCK;
    for (uint32_t k0 = blockIdx.x * BLOCK + threadIdx.x; k0 < s1.n; k0 += 4u * stride) {
        uint64_t a1[4], a2[4];
        bool on[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t k = k0 + (uint32_t)u * stride;
            on[u] = k < s1.n;
            a1[u] = on[u] ? atom_of(s1, k) : 0ull;
            a2[u] = on[u] ? atom_of(s2, k) : 0ull;
        }
        V3 p[4], q[4];
        float m1[4], m2[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            p[u] = pos_of(s1, blockIdx.y, a1[u]);
            q[u] = pos_of(s2, 0, a2[u]);
            m1[u] = s1.mass[a1[u]];
            m2[u] = s2.mass ? s2.mass[a2[u]] : m1[u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (on[u]) fit_accumulate<UNW, NV>(acc, p[u], q[u], (double)m1[u], (double)m2[u]);
    }
    block_reduce_store<NV, BLOCK>(acc, partials);
}

// Batched fits (fit_rmsd_batch): the reference selection, both mass columns and the index do not change from frame to
// frame, so they are gathered ONCE into a dense record per selected atom - {q.x, q.y, q.z, m1} and {atom of the frame,
// m2} - and every frame then reads 24 coalesced bytes per atom plus ONE scattered 12-byte position instead of four
// scattered reads behind two 8-byte indices.  The terms, their order and the trip structure are those of k_fit_sums:
// the sums are bit-identical to the unpacked kernel's.
__global__ void __launch_bounds__(RB) k_fit_pack(Sel s1, Sel s2, float4 *__restrict__ qm, uint2 *__restrict__ am) {
    const uint32_t k = blockIdx.x * RB + threadIdx.x;
    if (k >= s1.n) return;
    const uint64_t a1 = atom_of(s1, k), a2 = atom_of(s2, k);
    const V3 q = pos_of(s2, 0, a2);
    const float m1 = s1.mass[a1];
    const float m2 = s2.mass ? s2.mass[a2] : m1;
    qm[k] = make_float4(q.x, q.y, q.z, m1);
    am[k] = make_uint2((uint32_t)a1, __float_as_uint(m2));
}

template <bool UNW>
__global__ void __launch_bounds__(RB) k_fit_sums_packed(const float *__restrict__ xyz, size_t frame_stride, uint32_t n,
                                                        const float4 *__restrict__ qm, const uint2 *__restrict__ am,
                                                        double *partials) {
    constexpr int BLOCK = RB;
    constexpr int NV = UNW ? FS_ALL : FS_W;
    double acc[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = 0.0;
    const float *__restrict__ frame = xyz + (size_t)blockIdx.y * frame_stride;
    const uint32_t stride = gridDim.x * BLOCK;
    for (uint32_t k0 = blockIdx.x * BLOCK + threadIdx.x; k0 < n; k0 += 4u * stride) {
        uint2 a[4];
        float4 r[4];
        bool on[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t k = k0 + (uint32_t)u * stride;
            on[u] = k < n;
            a[u] = on[u] ? am[k] : make_uint2(0u, 0u);
            r[u] = on[u] ? qm[k] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        V3 p[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float *q = frame + 3 * (size_t)a[u].x;
            p[u] = v3(q[0], q[1], q[2]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (on[u])
                fit_accumulate<UNW, NV>(acc, p[u], v3(r[u].x, r[u].y, r[u].z), (double)r[u].w, (double)__uint_as_float(a[u].y));
    }
    block_reduce_store<NV, BLOCK>(acc, partials);
}

// the record of one fit from its 38 (21 without the unweighted block) sums; one thread
__device__ __forceinline__ void fit_finalize(const double (&S)[FS_ALL], int nv, uint32_t n, int at_origin, float *o) {
    #pragma unroll
    for (int i = 0; i < 17; ++i) o[i] = 0.f;
    if (S[0] == 0.0 || (!at_origin && S[17] == 0.0)) {
        o[17] = __int_as_float(MOLAR_HIP_ERR_ZERO_MASS);
        return;
    }
    // centres as the reference holds them: f32 points (Pos)
    float c1f[3] = {0.f, 0.f, 0.f}, c2f[3] = {0.f, 0.f, 0.f};
    if (!at_origin)
        #pragma unroll
        for (int d = 0; d < 3; ++d) {
            c1f[d] = (float)(S[1 + d] / S[0]);
            c2f[d] = (float)(S[18 + d] / S[17]);
        }
    double cov[9], R[9];
    #pragma unroll
    for (int c = 0; c < 3; ++c)
        #pragma unroll
        for (int r = 0; r < 3; ++r)      // S m (q_r - c2_r)(p_c - c1_c)
            cov[c * 3 + r] = ((S[7 + c * 3 + r] - (double)c2f[r] * S[1 + c]) - S[4 + r] * (double)c1f[c]) +
                             S[0] * (double)c2f[r] * (double)c1f[c];
    if (!rotation_from_cov(cov, R)) {
        o[17] = __int_as_float(MOLAR_HIP_ERR_SVD);
        return;
    }
    float Rf[9];
    #pragma unroll
    for (int i = 0; i < 9; ++i) Rf[i] = (float)R[i];
    const V3 rv = mat_vec(Rf, v3(-c1f[0], -c1f[1], -c1f[2]));
    const float tf[3] = {at_origin ? 0.f : c2f[0] + rv.x, at_origin ? 0.f : c2f[1] + rv.y, at_origin ? 0.f : c2f[2] + rv.z};
    #pragma unroll
    for (int i = 0; i < 9; ++i) o[i] = Rf[i];
    #pragma unroll
    for (int d = 0; d < 3; ++d) o[9 + d] = tf[d];
    o[17] = __int_as_float(0);
    // scalars of the fitted selection
    const double t[3] = {tf[0], tf[1], tf[2]};
    const double cm[3] = {S[1] / S[0], S[2] / S[0], S[3] / S[0]};
    #pragma unroll
    for (int r = 0; r < 3; ++r) o[13 + r] = (float)(((R[0 * 3 + r] * cm[0] + R[1 * 3 + r] * cm[1]) + R[2 * 3 + r] * cm[2]) + t[r]);
    double rg2 = S[16] / S[0] - ((cm[0] * cm[0] + cm[1] * cm[1]) + cm[2] * cm[2]);
    o[16] = (float)sqrt(rg2 > 0.0 ? rg2 : 0.0);
    if (nv == FS_ALL && n) {
        double cross = 0.0, tRp = 0.0, tq = 0.0;
        #pragma unroll
        for (int r = 0; r < 3; ++r) {
            double rsp = 0.0;
            #pragma unroll
            for (int c = 0; c < 3; ++c) {
                cross += R[c * 3 + r] * S[27 + c * 3 + r];
                rsp += R[c * 3 + r] * S[21 + c];
            }
            tRp += t[r] * rsp;
            tq += t[r] * S[24 + r];
        }
        const double ss = ((S[36] + S[37]) + (double)n * ((t[0] * t[0] + t[1] * t[1]) + t[2] * t[2])) + 2.0 * ((tRp - tq) - cross);
        o[12] = (float)sqrt(ss > 0.0 ? ss / (double)n : 0.0);
    }
}

// One 64-lane workgroup per frame: totals the partials in a fixed order, then lane 0 derives
//   out[frame] = { R (9, column-major), t (3), rmsd, com (3), gyration, status }      (18 floats)
// R by Horn's quaternion method (rotation_from_cov); t = cm2 + R (-cm1) in f32 like Translation*rot*Translation (:521).
// RMSD, COM and gyration are those of the FITTED selection (p' = R p + t), from the sums:
//   S |R p + t - q|^2 = S|p|^2 + S|q|^2 + n|t|^2 + 2 t.(R S p) - 2 t.S q - 2 sum_rc R[r][c] S q_r p_c      (R orthonormal)
//   com' = R (S m p / S m) + t,   rg^2 = S m|p|^2 / S m - |S m p / S m|^2                                 (rigid motion)
// (NV is a template parameter: a run-time "v < nv" around each of the 38 loads makes the compiler branch around every
// one and wait for it - 76 dependent L2 round trips, 15 us of the kernel's 17)
template <int NV>
__global__ void __launch_bounds__(64) k_fit_final(const double *partials, uint32_t nblk, uint32_t n, int at_origin,
                                                  float *out, float *out_host /* pinned host copy or NULL */) {
    constexpr int nv = NV;
    const uint32_t f = blockIdx.x, lane = threadIdx.x;
    double S[FS_ALL];
#pragma unroll
    for (int v = 0; v < FS_ALL; ++v) S[v] = 0.0;
    for (uint32_t b = lane; b < nblk; b += 64) {
        const double *row = partials + ((size_t)f * nblk + b) * nv;
#pragma unroll
        for (int v = 0; v < FS_ALL; ++v)
            if (v < nv) S[v] += row[v];
    }
    // butterfly over the lanes, all values per step (independent shuffles pipeline; one value at a time would
    // serialise 38 x 6 dependent cross-lane round trips)
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int v = 0; v < FS_ALL; ++v) S[v] += __shfl_xor(S[v], off, 64);
    }
    if (lane != 0) return;
    fit_finalize(S, nv, n, at_origin, out + 18 * (size_t)f);
    if (out_host) {
#pragma unroll
        for (int i = 0; i < 18; ++i) out_host[18 * (size_t)f + i] = out[18 * (size_t)f + i];
    }
}

// apply_transform (modify.rs:32-36) for every frame of a batch: p <- R p + t in f32, in place
__global__ void __launch_bounds__(RB) k_apply_batch(Sel s1, float *xyz_rw, const float *fit /*[frame][18]*/) {
    const float *R = fit + 18 * (size_t)blockIdx.y;
    if (__float_as_int(R[17]) != 0) return;
    const V3 t = v3(R[9], R[10], R[11]);
    for (uint32_t k = blockIdx.x * RB + threadIdx.x; k < s1.n; k += gridDim.x * RB) {
        const uint64_t a1 = atom_of(s1, k);
        const V3 pn = mat_vec(R, pos_of(s1, blockIdx.y, a1)) + t;
        float *q = xyz_rw + (size_t)blockIdx.y * s1.frame_stride + 3 * a1;
        q[0] = pn.x; q[1] = pn.y; q[2] = pn.z;
    }
}

// unwrap_simple_dim (modify.rs:40-54): p_k <- closest_image(p_k, p_0) for k >= 1
__global__ void __launch_bounds__(RB) k_unwrap(Sel s, float *xyz_rw, molar_hip_box box, uint32_t pbc) {
    const V3 p0 = pos_of(s, 0, atom_of(s, 0));
    for (uint32_t k = 1 + blockIdx.x * RB + threadIdx.x; k < s.n; k += gridDim.x * RB) {
        const uint64_t a = atom_of(s, k);
        const V3 im = closest_image(box, pos_of(s, 0, a), p0, pbc);
        float *q = xyz_rw + 3 * a;
        q[0] = im.x; q[1] = im.y; q[2] = im.z;
    }
}

__device__ __forceinline__ uint32_t f2ord(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// min_max (:22-36), exact: mm[0..2]=min, mm[3..5]=max as order-preserving uints
__global__ void __launch_bounds__(RB) k_minmax(Sel s, uint32_t *mm) {
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t k = blockIdx.x * RB + threadIdx.x; k < s.n; k += gridDim.x * RB) {
        const V3 p = pos_of(s, 0, atom_of(s, k));
        const float q[3] = {p.x, p.y, p.z};
        for (int d = 0; d < 3; ++d) {
            if (q[d] < lo[d]) lo[d] = q[d];
            if (q[d] > hi[d]) hi[d] = q[d];
        }
    }
    for (int d = 0; d < 3; ++d)
        for (int off = 32; off > 0; off >>= 1) {
            lo[d] = fminf(lo[d], __shfl_xor(lo[d], off, 64));
            hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], off, 64));
        }
    if ((threadIdx.x & 63) == 0)
        for (int d = 0; d < 3; ++d) {
            if (lo[d] != INFINITY) atomicMin(&mm[d], f2ord(lo[d]));
            if (hi[d] != -INFINITY) atomicMax(&mm[3 + d], f2ord(hi[d]));
        }
}

// ---------------------------------------------------------------- batched (CSR) selections: one wave each

__global__ void __launch_bounds__(256) k_center_batch(const float *__restrict__ xyz, const uint64_t *__restrict__ idx,
                                                      const uint64_t *__restrict__ off, uint32_t nsel,
                                                      const float *__restrict__ mass, float *__restrict__ out,
                                                      int *__restrict__ status) {
    const uint32_t k = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (k >= nsel) return;
    const uint64_t s = off[k], e = off[k + 1];
    double sm = 0, sx = 0, sy = 0, sz = 0;
    for (uint64_t q = s + lane; q < e; q += 64) {
        const uint64_t a = idx[q];
        const float *p = xyz + 3 * a;
        const float m = mass ? mass[a] : 1.0f;
        sm += (double)m;
        sx += (double)(p[0] * m);
        sy += (double)(p[1] * m);
        sz += (double)(p[2] * m);
    }
    for (int o = 32; o > 0; o >>= 1) {
        sm += __shfl_xor(sm, o, 64);
        sx += __shfl_xor(sx, o, 64);
        sy += __shfl_xor(sy, o, 64);
        sz += __shfl_xor(sz, o, 64);
    }
    if (lane == 0) {
        const double den = mass ? sm : (double)(e - s);
        if (den == 0.0) {
            atomicMax(status, MOLAR_HIP_ERR_ZERO_MASS);
            out[3 * k] = out[3 * k + 1] = out[3 * k + 2] = 0.f;
        } else {
            out[3 * k] = (float)(sx / den);
            out[3 * k + 1] = (float)(sy / den);
            out[3 * k + 2] = (float)(sz / den);
        }
    }
}

__global__ void __launch_bounds__(256) k_unwrap_batch(float *__restrict__ xyz, const uint64_t *__restrict__ idx,
                                                      const uint64_t *__restrict__ off, uint32_t nsel, molar_hip_box box,
                                                      uint32_t pbc) {
    const uint32_t k = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (k >= nsel) return;
    const uint64_t s = off[k], e = off[k + 1];
    if (e - s < 2) return;
    const float *q0 = xyz + 3 * idx[s];
    const V3 p0 = v3(q0[0], q0[1], q0[2]);
    for (uint64_t q = s + 1 + lane; q < e; q += 64) {
        float *p = xyz + 3 * idx[q];
        const V3 im = closest_image(box, v3(p[0], p[1], p[2]), p0, pbc);
        p[0] = im.x; p[1] = im.y; p[2] = im.z;
    }
}


// ---------------------------------------------------------------- CSR-batched Measure: one wave per selection
// What MolAR runs from rayon over a ParSplit (selection/system.rs:193-213, README :656-691): the same Measure method on
// thousands of small sub-selections (residues, lipids, molecules).  Selection k is idx[off[k] .. off[k+1]).

template <int NV>
__device__ __forceinline__ void wave_sum(double (&x)[NV]) {
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int v = 0; v < NV; ++v) x[v] += __shfl_xor(x[v], off, 64);
    }
}

// gyration (measure.rs:78-87) / gyration_pbc (:222-232) per selection.  Non-periodic: one pass of uncentred moments
// (k_moments).  Periodic: centre of mass from the images relative to the selection's first atom with the reference's
// unweighted-first-atom quirk (:197-220), then S m |shortest_vector(p - c)|^2.
__global__ void __launch_bounds__(256) k_gyration_batch(const float *__restrict__ xyz, const uint64_t *__restrict__ idx,
                                                        const uint64_t *__restrict__ off, uint32_t nsel,
                                                        const float *__restrict__ mass, int use_box, molar_hip_box box,
                                                        float *__restrict__ out, int *__restrict__ status) {
    const uint32_t k = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (k >= nsel) return;
    const uint64_t s = off[k], e = off[k + 1];
    if (e == s) {
        if (lane == 0) { atomicMax(status, MOLAR_HIP_ERR_INVALID_ARGUMENT); out[k] = 0.f; }
        return;
    }
    auto P = [&](uint64_t q) { const float *p = xyz + 3 * idx[q]; return v3(p[0], p[1], p[2]); };
    if (!use_box) {
        double u[5] = {0, 0, 0, 0, 0};     // S m, S m p, S m |p|^2
        for (uint64_t q = s + lane; q < e; q += 64) {
            const V3 pf = P(q);
            const double x = pf.x, y = pf.y, z = pf.z, m = (double)mass[idx[q]];
            u[0] += m; u[1] += m * x; u[2] += m * y; u[3] += m * z;
            u[4] += m * ((x * x + y * y) + z * z);
        }
        wave_sum<5>(u);
        if (lane == 0) {
            if (u[0] == 0.0) { atomicMax(status, MOLAR_HIP_ERR_ZERO_MASS); out[k] = 0.f; return; }
            const double cx = (double)(float)(u[1] / u[0]), cy = (double)(float)(u[2] / u[0]), cz = (double)(float)(u[3] / u[0]);
            const double cc = (cx * cx + cy * cy) + cz * cz;
            const double ss = (u[4] - 2.0 * ((cx * u[1] + cy * u[2]) + cz * u[3])) + u[0] * cc;     // S m |p - c|^2
            out[k] = (float)sqrt(ss > 0.0 ? ss / u[0] : 0.0);
        }
        return;
    }
    const V3 p0 = P(s);
    double a[4] = {0, 0, 0, 0};
    for (uint64_t q = s + 1 + lane; q < e; q += 64) {
        const V3 im = closest_image(box, P(q), p0, MOLAR_HIP_PBC_FULL);
        const float m = mass[idx[q]];
        a[0] += (double)m; a[1] += (double)(im.x * m); a[2] += (double)(im.y * m); a[3] += (double)(im.z * m);
    }
    wave_sum<4>(a);
    const double mt = a[0] + (double)mass[idx[s]];
    if (mt == 0.0) {
        if (lane == 0) { atomicMax(status, MOLAR_HIP_ERR_ZERO_MASS); out[k] = 0.f; }
        return;
    }
    // cm = p0 (unweighted) + S img m, divided by m0 + S m (:180-182,:213-218)
    const V3 c = v3((float)(((double)p0.x + a[1]) / mt), (float)(((double)p0.y + a[2]) / mt), (float)(((double)p0.z + a[3]) / mt));
    double g[2] = {0, 0};
    for (uint64_t q = s + lane; q < e; q += 64) {
        const V3 d = shortest_vector(box, P(q) - c, MOLAR_HIP_PBC_FULL);
        const float m = mass[idx[q]];
        g[0] += (double)m;
        g[1] += (double)(norm2(d) * m);
    }
    wave_sum<2>(g);
    if (lane == 0) out[k] = (float)sqrt(g[1] / g[0]);
}

// rmsd (:485-504) / rmsd_mw (:538-558) per selection pair (idx1[off..], idx2[off..] in lock step)
__global__ void __launch_bounds__(256) k_rmsd_batch(const float *__restrict__ xyz1, const float *__restrict__ xyz2,
                                                    const uint64_t *__restrict__ idx1, const uint64_t *__restrict__ idx2,
                                                    const uint64_t *__restrict__ off, uint32_t nsel,
                                                    const float *__restrict__ mass, float *__restrict__ out,
                                                    int *__restrict__ status) {
    const uint32_t k = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (k >= nsel) return;
    const uint64_t s = off[k], e = off[k + 1];
    double r[2] = {0, 0};
    for (uint64_t q = s + lane; q < e; q += 64) {
        const float *p1 = xyz1 + 3 * idx1[q], *p2 = xyz2 + 3 * idx2[q];
        const float d2 = norm2(v3(p2[0], p2[1], p2[2]) - v3(p1[0], p1[1], p1[2]));
        const float m = mass ? mass[idx1[q]] : 1.0f;
        r[0] += mass ? (double)(d2 * m) : (double)d2;
        r[1] += (double)m;
    }
    wave_sum<2>(r);
    if (lane == 0) {
        if (r[1] == 0.0) { atomicMax(status, e == s ? MOLAR_HIP_ERR_INVALID_ARGUMENT : MOLAR_HIP_ERR_ZERO_MASS); out[k] = 0.f; }
        else out[k] = (float)sqrt(r[0] / r[1]);
    }
}

// fit_transform (:507-522) of every selection onto its counterpart, optionally applied in place (modify.rs:32-36), with
// the RMSD / centre of mass / gyration of the fitted selection: out[k] = the 18-float record of k_fit_final
__global__ void __launch_bounds__(256) k_fit_csr(float *__restrict__ xyz1, const float *__restrict__ xyz2,
                                                 const uint64_t *__restrict__ idx1, const uint64_t *__restrict__ idx2,
                                                 const uint64_t *__restrict__ off, uint32_t nsel,
                                                 const float *__restrict__ mass1, const float *__restrict__ mass2, int apply,
                                                 float *__restrict__ out) {
    const uint32_t k = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (k >= nsel) return;
    const uint64_t s = off[k], e = off[k + 1];
    double S[FS_ALL];
#pragma unroll
    for (int v = 0; v < FS_ALL; ++v) S[v] = 0.0;
    for (uint64_t q = s + lane; q < e; q += 64) {
        const uint64_t a1 = idx1[q], a2 = idx2[q];
        const float *p = xyz1 + 3 * a1, *r = xyz2 + 3 * a2;
        const double m = (double)mass1[a1];
        fit_accumulate<true, FS_ALL>(S, v3(p[0], p[1], p[2]), v3(r[0], r[1], r[2]), m, mass2 ? (double)mass2[a2] : m);
    }
    wave_sum<FS_ALL>(S);
    float *o = out + 18 * (size_t)k;
    if (lane == 0) fit_finalize(S, FS_ALL, (uint32_t)(e - s), 0, o);
    if (!apply) return;
    float Rt[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) Rt[i] = __shfl(lane == 0 ? o[i] : 0.f, 0, 64);
    const int st = __shfl(lane == 0 ? __float_as_int(o[17]) : 0, 0, 64);
    if (st != 0) return;
    const V3 t = v3(Rt[9], Rt[10], Rt[11]);
    for (uint64_t q = s + lane; q < e; q += 64) {
        float *p = xyz1 + 3 * idx1[q];
        const V3 pn = mat_vec(Rt, v3(p[0], p[1], p[2])) + t;
        p[0] = pn.x; p[1] = pn.y; p[2] = pn.z;
    }
}

// translate (modify.rs:16-23): p += shift
__global__ void __launch_bounds__(RB) k_translate(Sel s, float *xyz_rw, float sx, float sy, float sz) {
    for (uint32_t k = blockIdx.x * RB + threadIdx.x; k < s.n; k += gridDim.x * RB) {
        float *q = xyz_rw + 3 * atom_of(s, k);
        q[0] += sx; q[1] += sy; q[2] += sz;
    }
}

// ---------------------------------------------------------------- lipid tail order (measure.rs:270-422)

struct F3 {
    float x, y, z;
};
__device__ __forceinline__ F3 f3sub(F3 a, F3 b) { return F3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ float f3dot(F3 a, F3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ float f3norm(F3 a) { return __builtin_sqrtf(f3dot(a, a)); }
__device__ __forceinline__ F3 f3cross(F3 a, F3 b) {
    return F3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ F3 f3unit(F3 a) {
    const float n = f3norm(a);
    return F3{a.x / n, a.y / n, a.z / n};
}
// nalgebra Vector::angle: acos(clamp(a.b / (|a||b|), -1, 1)), 0 if either vector is zero
__device__ __forceinline__ float f3angle(F3 a, F3 b) {
    const float n1 = f3norm(a), n2 = f3norm(b);
    if (n1 == 0.0f || n2 == 0.0f) return 0.0f;
    float c = f3dot(a, b) / (n1 * n2);
    c = c < -1.0f ? -1.0f : (c > 1.0f ? 1.0f : c);
    return acosf(c);
}

// Sixteen lanes per tail, one output per lane and round.  The reference's loop (measure.rs:330-420) walks the bonds in
// order and a double bond at i writes order[i-1] and order[i], so for any bond pattern output k ends up with the value of
// its LAST writer: the double bond at k+1 (its first half) if there is one, else the single-single pair at k or the second
// half of a double bond at k, else 0.  Each lane evaluates only that writer, with the reference's f32 expressions; every
// bond 0..n-3 is checked for a legal position by the lane of the same number.
constexpr uint32_t ORDER_G = 16;
__global__ void __launch_bounds__(256) k_lipid_order(const float *__restrict__ xyz, const uint64_t *__restrict__ idx,
                                                     const uint64_t *__restrict__ toff, uint32_t ntails, int order_type,
                                                     const float *__restrict__ normals, const uint64_t *__restrict__ noff,
                                                     const uint8_t *__restrict__ bonds, float *__restrict__ out,
                                                     int *__restrict__ status) {
    const uint32_t t = blockIdx.x * (256u / ORDER_G) + threadIdx.x / ORDER_G, sub = threadIdx.x % ORDER_G;
    if (t >= ntails) return;
    const uint64_t a0 = toff[t], n0 = noff[t];
    const uint32_t n = (uint32_t)(toff[t + 1] - a0);
    const uint32_t nn = (uint32_t)(noff[t + 1] - n0);
    if (n < 3) {
        if (sub == 0) atomicMax(status, MOLAR_HIP_ERR_LIPID_TAIL_TOO_SHORT);
        return;
    }
    if (nn != 1 && nn != n - 2) {
        if (sub == 0) atomicMax(status, MOLAR_HIP_ERR_LIPID_NORMALS_COUNT);
        return;
    }
    const uint8_t *bo = bonds + (a0 - t);
    float *order = out + (a0 - 2ull * t);
    auto P = [&](uint32_t k) {
        const float *q = xyz + 3 * idx[a0 + k];
        return F3{q[0], q[1], q[2]};
    };
    auto N = [&](uint32_t k) {
        const float *q = normals + 3 * (n0 + (nn == 1 ? 0 : k));
        return F3{q[0], q[1], q[2]};
    };
    const float sqrt3 = __builtin_sqrtf(3.0f), pi = 3.14159265358979323846f;
    for (uint32_t k = sub; k < n - 2; k += ORDER_G) {
        if (order_type == 0) {
            const float c = cosf(f3angle(f3sub(P(k + 2), P(k)), N(k)));
            order[k] = 1.5f * (c * c) - 0.5f;
            continue;
        }
        const bool dbl_here = bo[k] != 1, dbl_next = bo[k + 1] != 1;
        // a double bond at bond 0 has no C(i-1); at the last bond it has no normal for atom i+1 when normals are per
        // bond.  The reference indexes out of range there (usize underflow panic / unchecked read,
        // measure.rs:361-364,385): refuse the tail instead of touching memory outside it.
        const bool bad_here = dbl_here && (k == 0 || (nn != 1 && k + 1 >= nn));
        if (bad_here) atomicMax(status, MOLAR_HIP_ERR_INVALID_ARGUMENT);
        const bool from_next = dbl_next && k + 1 < n - 2;                 // bond n-2 is never visited by the reference's loop
        float v = 0.0f;
        if (!from_next && !dbl_here) {
            if (!dbl_next) {
                const F3 p1 = P(k), p2 = P(k + 1), p3 = P(k + 2);
                const F3 lz = f3unit(f3sub(p3, p1));
                const F3 lx = f3unit(f3cross(f3sub(p1, p2), f3sub(p3, p2)));
                const F3 ly = f3cross(lx, lz);
                const F3 nv = N(k);
                const float cx = cosf(f3angle(lx, nv)), cy = cosf(f3angle(ly, nv));
                const float sxx = 0.5f * (3.0f * (cx * cx) - 1.0f), syy = 0.5f * (3.0f * (cy * cy) - 1.0f);
                v = -(2.0f * sxx + syy) / 3.0f;
            }
        } else if (from_next || !bad_here) {
            const int side = from_next ? 0 : 1;                            // which half of the double bond at i lands on k
            const uint32_t i = from_next ? k + 1 : k;
            const F3 p1 = P(i - 1), p2 = P(i), p3 = P(i + 1), p4 = P(i + 2);
            const float a = side == 0 ? 0.5f * (pi - f3angle(f3sub(p1, p2), f3sub(p3, p2)))
                                      : 0.5f * (pi - f3angle(f3sub(p2, p3), f3sub(p4, p3)));
            const F3 lz = f3unit(f3sub(p3, p2));
            const F3 lx = f3unit(f3cross(side == 0 ? f3sub(p1, p2) : f3sub(p3, p4), lz));
            const F3 ly = f3cross(lx, lz);
            const F3 nv = N(side == 0 ? i : i + 1);
            const float cy = cosf(f3angle(ly, nv)), cz = cosf(f3angle(lz, nv));
            const float szz = 0.5f * (3.0f * (cz * cz) - 1.0f), syy = 0.5f * (3.0f * (cy * cy) - 1.0f);
            const float syz = 1.5f * cy * cz;
            const float sgn = side == 0 ? -1.0f : 1.0f;
            if (order_type == 2) {
                const float ca = cosf(a), sa = sinf(a);
                v = -(((ca * ca) * syy + (sa * sa) * szz) + sgn * (2.0f * ca * sa * syz));
            } else {
                v = -((szz / 4.0f + 3.0f * syy / 4.0f) + sgn * (sqrt3 * syz / 2.0f));
            }
        }
        order[k] = v;
    }
}

// ---------------------------------------------------------------- finalize kernels (one thread per frame)

// generic: total the partials of frame 0 into results[0..nv)
// (64 lanes: lane l adds blocks l, l+64, ...; a butterfly adds the lanes - a fixed order, so results are reproducible)
template <int NV>
__global__ void __launch_bounds__(64) k_fin_sum(const double *partials, uint32_t nblk, double *results) {
    const uint32_t lane = threadIdx.x;
    double x[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) x[v] = 0.0;
    for (uint32_t b = lane; b < nblk; b += 64) {
#pragma unroll
        for (int v = 0; v < NV; ++v) x[v] += partials[(size_t)b * NV + v];     // unconditional: the loads pipeline
    }
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int v = 0; v < NV; ++v) x[v] += __shfl_xor(x[v], off, 64);
    }
#pragma unroll
    for (int v = 0; v < NV; ++v)
        if (lane == 0) results[v] = x[v];
}

// ---------------------------------------------------------------- host helpers

uint32_t blocks_for(const molar_hip_ctx *c, uint32_t n, uint32_t nframes) {
    // >= 8 atoms per thread for batches; a single frame is latency-bound (a thread's gathers are serial round trips to
    // HBM), so it is spread over more workgroups: 4 atoms per thread
    const uint32_t per = nframes >= 8 ? 8u : 4u;
    uint32_t nb = (n + RB * per - 1) / (RB * per);
    if (nb < 1) nb = 1;
    uint32_t cap = nframes >= 64 ? 16u : (uint32_t)c->num_cus * 4u;
    if (nb > cap) nb = cap;
    return nb;
}

int stage_sel(molar_hip_ctx *c, const float *xyz, size_t natoms, const uint64_t *idx, size_t n, const float *mass,
              DevBuf &bx, DevBuf &bi, DevBuf &bm, Sel *out, size_t nframes = 1) {
    if (!xyz) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "measure: xyz pointer is null");
    const size_t nsel = idx ? n : natoms;
    if (nsel >= 0xFFFFFFFFull) return fail(MOLAR_HIP_ERR_TOO_LARGE, "measure: selection too large");
    out->n = (uint32_t)nsel;
    out->frame_stride = nframes > 1 ? natoms * 3 : 0;
    MH_TRY(to_device(c, xyz, natoms * 3 * nframes, bx, &out->xyz));
    MH_TRY(to_device(c, idx, idx ? n : 0, bi, &out->idx));
    out->mass = nullptr;
    if (mass) MH_TRY(to_device(c, mass, natoms, bm, &out->mass));
    return 0;
}

int pull(molar_hip_ctx *c, void *dst, const void *src_dev, size_t bytes) {
    MH_TRY(ensure_pinned(c, bytes));
    MH_HIP(hipMemcpyAsync(c->h_pinned, src_dev, bytes, hipMemcpyDeviceToHost, c->stream));
    MH_HIP(hipStreamSynchronize(c->stream));
    std::memcpy(dst, c->h_pinned, bytes);
    return 0;
}

// run a reduction for one frame and fetch its NV totals
template <class Launch>
int reduce1(molar_hip_ctx *c, uint32_t n, int nv, double *host_out, Launch launch) {
    const uint32_t nb = blocks_for(c, n, 1);
    MH_TRY(c->m_partials.reserve((size_t)nb * nv * 8));
    // the totals are written straight into pinned host memory: the call ends with a stream wait, not a copy
    MH_TRY(ensure_pinned(c, 64 * 8));
    launch(nb, c->m_partials.as<double>());
    double *part = c->m_partials.as<double>(), *res = static_cast<double *>(c->h_pinned);
    switch (nv) {
        case 3: hipLaunchKernelGGL(k_fin_sum<3>, dim3(1), dim3(64), 0, c->stream, part, nb, res); break;
        case 7: hipLaunchKernelGGL(k_fin_sum<7>, dim3(1), dim3(64), 0, c->stream, part, nb, res); break;
        case 8: hipLaunchKernelGGL(k_fin_sum<8>, dim3(1), dim3(64), 0, c->stream, part, nb, res); break;
        case 9: hipLaunchKernelGGL(k_fin_sum<9>, dim3(1), dim3(64), 0, c->stream, part, nb, res); break;
        case 10: hipLaunchKernelGGL(k_fin_sum<10>, dim3(1), dim3(64), 0, c->stream, part, nb, res); break;
        default: return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "reduce1: unsupported width %d", nv);
    }
    MH_HIP(hipGetLastError());
    MH_HIP(hipStreamSynchronize(c->stream));
    std::memcpy(host_out, c->h_pinned, (size_t)nv * 8);
    return 0;
}

int box_or_err(const float *box9, molar_hip_box *b) {
    if (!box9) return fail(MOLAR_HIP_ERR_NO_PBC, "pbc operation without periodic box");
    return molar_hip_box_from_matrix(box9, b);
}

int com_host(molar_hip_ctx *c, const Sel &s, float out[3]) {
    double r[7];
    MH_TRY(reduce1(c, s.n, 7, r, [&](uint32_t nb, double *part) {
        hipLaunchKernelGGL(k_sums, dim3(nb, 1), dim3(RB), 0, c->stream, s, part);
    }));
    if (r[0] == 0.0) return fail(MOLAR_HIP_ERR_ZERO_MASS, "zero mass");
    for (int d = 0; d < 3; ++d) out[d] = (float)(r[1 + d] / r[0]);
    return 0;
}

// center_of_mass_pbc_dims (:197-220): cm starts at the UNWEIGHTED first position, mass at m0
int com_pbc_host(molar_hip_ctx *c, const Sel &s, const molar_hip_box &box, uint8_t pbc, bool weighted, float out[3]) {
    // the periodic centres are taken relative to the selection's FIRST atom (measure.rs:149,180): an empty selection has
    // none (MolAR cannot construct one, sel.rs:13-19)
    if (s.n == 0) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "periodic centre of an empty selection");
    double r[7];
    MH_TRY(reduce1(c, s.n, 7, r, [&](uint32_t nb, double *part) {
        hipLaunchKernelGGL(k_sums_pbc, dim3(nb, 1), dim3(RB), 0, c->stream, s, box, (uint32_t)pbc, part);
    }));
    // first atom: position and mass
    float p0[3], m0 = 1.0f;
    {
        MH_TRY(c->m_out.reserve(64));
        uint64_t a0 = 0;
        if (s.idx) MH_TRY(pull(c, &a0, s.idx, 8));
        MH_TRY(pull(c, p0, s.xyz + 3 * a0, 12));
        if (s.mass) MH_TRY(pull(c, &m0, s.mass + a0, 4));
    }
    if (weighted) {
        const double mass = (double)m0 + r[0];
        if (mass == 0.0) return fail(MOLAR_HIP_ERR_ZERO_MASS, "zero mass");
        for (int d = 0; d < 3; ++d) out[d] = (float)(((double)p0[d] + r[1 + d]) / mass);
    } else {
        for (int d = 0; d < 3; ++d) out[d] = (float)(((double)p0[d] + r[4 + d]) / (double)s.n);
    }
    return 0;
}

int central_host(molar_hip_ctx *c, const Sel &s, const float center[3], const molar_hip_box *box, double r[8]) {
    MH_TRY(c->m_out.reserve(64));
    MH_HIP(hipMemcpyAsync(c->m_out.p, center, 12, hipMemcpyHostToDevice, c->stream));
    molar_hip_box b{};
    if (box) b = *box;
    return reduce1(c, s.n, 8, r, [&](uint32_t nb, double *part) {
        hipLaunchKernelGGL(k_central, dim3(nb, 1), dim3(RB), 0, c->stream, s, c->m_out.as<float>(), box ? 1 : 0, b, part);
    });
}

// non-periodic central moments in one pass: r[] laid out like central_host's result (k_central)
int central_onepass_host(molar_hip_ctx *c, const Sel &s, double r[8]) {
    double u[10];
    MH_TRY(reduce1(c, s.n, 10, u, [&](uint32_t nb, double *part) {
        hipLaunchKernelGGL(k_moments, dim3(nb, 1), dim3(RB), 0, c->stream, s, part);
    }));
    if (u[0] == 0.0) return fail(MOLAR_HIP_ERR_ZERO_MASS, "zero mass");
    const double cx = (double)(float)(u[1] / u[0]), cy = (double)(float)(u[2] / u[0]), cz = (double)(float)(u[3] / u[0]);   // the f32 centre (:82)
    const double cc[3] = {cx, cy, cz}, sp[3] = {u[1], u[2], u[3]};
    auto cen = [&](int a, int b, double raw) { return ((raw - cc[a] * sp[b]) - sp[a] * cc[b]) + u[0] * cc[a] * cc[b]; };
    const double xx = cen(0, 0, u[4]), yy = cen(1, 1, u[5]), zz = cen(2, 2, u[6]);
    const double xy = cen(0, 1, u[7]), xz = cen(0, 2, u[8]), yz = cen(1, 2, u[9]);
    r[0] = u[0];
    r[1] = (xx + yy) + zz;
    r[2] = yy + zz; r[3] = xx + zz; r[4] = xx + yy;
    r[5] = -xy; r[6] = -xz; r[7] = -yz;
    return 0;
}

#define MH_CTX(c)                                                              \
    do {                                                                       \
        if (!(c)) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "null context"); \
        MH_HIP(hipSetDevice((c)->device));                                     \
    } while (0)

}  // namespace

namespace mh {

int enqueue_unwrap_batch(molar_hip_ctx *c, float *xyz, const uint64_t *idx, const uint64_t *off, uint32_t nsel,
                         const molar_hip_box &box, uint32_t pbc) {
    if (nsel == 0) return 0;
    hipLaunchKernelGGL(k_unwrap_batch, dim3((nsel + 3u) / 4u), dim3(256), 0, c->stream, xyz, idx, off, nsel, box, pbc & 7u);
    MH_HIP(hipGetLastError());
    return 0;
}

int enqueue_center_batch(molar_hip_ctx *c, const float *xyz, const uint64_t *idx, const uint64_t *off, uint32_t nsel,
                         const float *mass, float *out, int *status) {
    if (nsel == 0) return 0;
    hipLaunchKernelGGL(k_center_batch, dim3((nsel + 3u) / 4u), dim3(256), 0, c->stream, xyz, idx, off, nsel, mass, out, status);
    MH_HIP(hipGetLastError());
    return 0;
}

int enqueue_lipid_order(molar_hip_ctx *c, const float *xyz, const uint64_t *idx, const uint64_t *toff, uint32_t ntails,
                        int order_type, const float *normals, const uint64_t *noff, const uint8_t *bonds, float *out,
                        int *status) {
    if (ntails == 0) return 0;
    hipLaunchKernelGGL(k_lipid_order, dim3((ntails + 15u) / 16u), dim3(256), 0, c->stream, xyz, idx, toff, ntails, order_type,
                       normals, noff, bonds, out, status);
    MH_HIP(hipGetLastError());
    return 0;
}

}  // namespace mh

extern "C" {

int molar_hip_min_max(molar_hip_ctx *c, const float *xyz, size_t natoms, const uint64_t *idx, size_t n, float lower[3],
                      float upper[3]) {
    MH_CTX(c);
    Sel s;
    MH_TRY(stage_sel(c, xyz, natoms, idx, n, nullptr, c->m_xyz1, c->m_idx1, c->m_mass1, &s));
    MH_TRY(c->m_out.reserve(64));
    uint32_t seed[6];
    const float fmaxv = 3.40282347e+38f;   // Pos::max_value() / min_value() (:23-24)
    for (int d = 0; d < 3; ++d) {
        uint32_t u;
        std::memcpy(&u, &fmaxv, 4);
        seed[d] = u | 0x80000000u;          // ord(+MAX)
        seed[3 + d] = ~(u | 0x80000000u);   // ord(-MAX)
    }
    MH_HIP(hipMemcpyAsync(c->m_out.p, seed, sizeof seed, hipMemcpyHostToDevice, c->stream));
    if (s.n) {
        uint32_t nb = blocks_for(c, s.n, 1);
        hipLaunchKernelGGL(k_minmax, dim3(nb), dim3(RB), 0, c->stream, s, c->m_out.as<uint32_t>());
        MH_HIP(hipGetLastError());
    }
    uint32_t mm[6];
    MH_TRY(pull(c, mm, c->m_out.p, sizeof mm));
    for (int d = 0; d < 6; ++d) {
        const uint32_t o = mm[d];
        const uint32_t u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
        float f;
        std::memcpy(&f, &u, 4);
        (d < 3 ? lower[d] : upper[d - 3]) = f;
    }
    return MOLAR_HIP_OK;
}

int molar_hip_center_of_geometry(molar_hip_ctx *c, const float *xyz, size_t natoms, const uint64_t *idx, size_t n,
                                 float out[3]) {
    MH_CTX(c);
    Sel s;
    MH_TRY(stage_sel(c, xyz, natoms, idx, n, nullptr, c->m_xyz1, c->m_idx1, c->m_mass1, &s));
    double r[7];
    MH_TRY(reduce1(c, s.n, 7, r, [&](uint32_t nb, double *part) {
        hipLaunchKernelGGL(k_sums, dim3(nb, 1), dim3(RB), 0, c->stream, s, part);
    }));
    for (int d = 0; d < 3; ++d) out[d] = (float)(r[4 + d] / (double)s.n);
    return MOLAR_HIP_OK;
}

int molar_hip_center_of_mass(molar_hip_ctx *c, const float *xyz, size_t natoms, const uint64_t *idx, size_t n,
                             const float *mass, float out[3]) {
    MH_CTX(c);
    if (!mass) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "center_of_mass: mass pointer is null");
    Sel s;
    MH_TRY(stage_sel(c, xyz, natoms, idx, n, mass, c->m_xyz1, c->m_idx1, c->m_mass1, &s));
    return com_host(c, s, out);
}

int molar_hip_center_of_geometry_pbc(molar_hip_ctx *c, const float *xyz, size_t natoms, const uint64_t *idx, size_t n,
                                     const float *box9, uint8_t pbc, float out[3]) {
    MH_CTX(c);
    molar_hip_box b;
    MH_TRY(box_or_err(box9, &b));
    Sel s;
    MH_TRY(stage_sel(c, xyz, natoms, idx, n, nullptr, c->m_xyz1, c->m_idx1, c->m_mass1, &s));
    return com_pbc_host(c, s, b, pbc & 7u, false, out);
}

int molar_hip_center_of_mass_pbc(molar_hip_ctx *c, const float *xyz, size_t natoms, const uint64_t *idx, size_t n,
                                 const float *mass, const float *box9, uint8_t pbc, float out[3]) {
    MH_CTX(c);
    if (!mass) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "center_of_mass_pbc: mass pointer is null");
    molar_hip_box b;
    MH_TRY(box_or_err(box9, &b));
    Sel s;
    MH_TRY(stage_sel(c, xyz, natoms, idx, n, mass, c->m_xyz1, c->m_idx1, c->m_mass1, &s));
    return com_pbc_host(c, s, b, pbc & 7u, true, out);
}

int molar_hip_gyration(molar_hip_ctx *c, const float *xyz, size_t natoms, const uint64_t *idx, size_t n,
                       const float *mass, const float *box9, float *out) {
    MH_CTX(c);
    if (!mass) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "gyration: mass pointer is null");
    Sel s;
    MH_TRY(stage_sel(c, xyz, natoms, idx, n, mass, c->m_xyz1, c->m_idx1, c->m_mass1, &s));
    float cm[3];
    molar_hip_box b;
    double r[8];
    if (box9) {
        MH_TRY(molar_hip_box_from_matrix(box9, &b));
        MH_TRY(com_pbc_host(c, s, b, MOLAR_HIP_PBC_FULL, true, cm));   // center_of_mass_pbc (:227)
        MH_TRY(central_host(c, s, cm, &b, r));
    } else {
        MH_TRY(central_onepass_host(c, s, r));                        // centre of mass (:82) and moments in one pass
    }
    *out = (float)std::sqrt(r[1] > 0.0 ? r[1] / r[0] : 0.0);
    return MOLAR_HIP_OK;
}

int molar_hip_inertia(molar_hip_ctx *c, const float *xyz, size_t natoms, const uint64_t *idx, size_t n,
                      const float *mass, const float *box9, float moments[3], float axes9[9], float tensor9[9]) {
    MH_CTX(c);
    if (!mass) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "inertia: mass pointer is null");
    Sel s;
    MH_TRY(stage_sel(c, xyz, natoms, idx, n, mass, c->m_xyz1, c->m_idx1, c->m_mass1, &s));
    float cm[3];
    molar_hip_box b;
    double r[8];
    if (box9) {
        MH_TRY(molar_hip_box_from_matrix(box9, &b));
        MH_TRY(com_pbc_host(c, s, b, MOLAR_HIP_PBC_FULL, true, cm));
        MH_TRY(central_host(c, s, cm, &b, r));
    } else {
        MH_TRY(central_onepass_host(c, s, r));
    }
    // symmetric tensor (:580-589); stored as f32 like the reference's Matrix3f before the eigen solve
    const float T00 = (float)r[2], T11 = (float)r[3], T22 = (float)r[4];
    const float T01 = (float)r[5], T02 = (float)r[6], T12 = (float)r[7];
    if (tensor9) {
        const float t[9] = {T00, T01, T02, T01, T11, T12, T02, T12, T22};
        std::memcpy(tensor9, t, sizeof t);
    }
    double A[9] = {T00, T01, T02, T01, T11, T12, T02, T12, T22}, w[3], V[9];
    jacobi_sym<3>(A, w, V);
    int ord[3] = {0, 1, 2};   // ascending moments (:594-601)
    for (int a = 0; a < 2; ++a)
        for (int q = a + 1; q < 3; ++q)
            if (w[ord[q]] < w[ord[a]]) std::swap(ord[a], ord[q]);
    for (int k = 0; k < 3; ++k) moments[k] = (float)w[ord[k]];
    // col0, col1 normalised, col2 = col0 x col1 (:603-607)
    float e[2][3];
    for (int k = 0; k < 2; ++k) {
        float v[3] = {(float)V[0 * 3 + ord[k]], (float)V[1 * 3 + ord[k]], (float)V[2 * 3 + ord[k]]};
        const float nn = std::sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
        for (int d = 0; d < 3; ++d) e[k][d] = v[d] / nn;
    }
    const float c2[3] = {e[0][1] * e[1][2] - e[0][2] * e[1][1], e[0][2] * e[1][0] - e[0][0] * e[1][2],
                         e[0][0] * e[1][1] - e[0][1] * e[1][0]};
    for (int d = 0; d < 3; ++d) {
        axes9[0 * 3 + d] = e[0][d];
        axes9[1 * 3 + d] = e[1][d];
        axes9[2 * 3 + d] = c2[d];
    }
    return MOLAR_HIP_OK;
}

static int rmsd_common(molar_hip_ctx *c, const float *xyz1, size_t natoms1, const uint64_t *idx1, size_t n1,
                       const float *mass1, const float *xyz2, size_t natoms2, const uint64_t *idx2, size_t n2,
                       bool weighted, float *out) {
    const size_t s1n = idx1 ? n1 : natoms1, s2n = idx2 ? n2 : natoms2;
    if (s1n != s2n) return fail(MOLAR_HIP_ERR_SIZES, "incompatible sizes: %zu and %zu", s1n, s2n);
    Sel s1, s2;
    MH_TRY(stage_sel(c, xyz1, natoms1, idx1, n1, mass1, c->m_xyz1, c->m_idx1, c->m_mass1, &s1));
    MH_TRY(stage_sel(c, xyz2, natoms2, idx2, n2, nullptr, c->m_xyz2, c->m_idx2, c->m_mass2, &s2));
    double r[3];
    MH_TRY(reduce1(c, s1.n, 3, r, [&](uint32_t nb, double *part) {
        hipLaunchKernelGGL(k_rmsd, dim3(nb, 1), dim3(RB), 0, c->stream, s1, s2, part);
    }));
    if (weighted) {
        if (r[2] == 0.0) return fail(MOLAR_HIP_ERR_ZERO_MASS, "zero mass");
        *out = (float)std::sqrt(r[1] / r[2]);
    } else {
        *out = (float)std::sqrt(r[0] / (double)s1.n);
    }
    return MOLAR_HIP_OK;
}

int molar_hip_rmsd(molar_hip_ctx *c, const float *xyz1, size_t natoms1, const uint64_t *idx1, size_t n1,
                   const float *xyz2, size_t natoms2, const uint64_t *idx2, size_t n2, float *out) {
    MH_CTX(c);
    return rmsd_common(c, xyz1, natoms1, idx1, n1, nullptr, xyz2, natoms2, idx2, n2, false, out);
}

int molar_hip_rmsd_mw(molar_hip_ctx *c, const float *xyz1, size_t natoms1, const uint64_t *idx1, size_t n1,
                      const float *mass1, const float *xyz2, size_t natoms2, const uint64_t *idx2, size_t n2,
                      float *out) {
    MH_CTX(c);
    if (!mass1) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "rmsd_mw: mass pointer is null");
    return rmsd_common(c, xyz1, natoms1, idx1, n1, mass1, xyz2, natoms2, idx2, n2, true, out);
}

int molar_hip_fit_transform(molar_hip_ctx *c, const float *xyz1, size_t natoms1, const uint64_t *idx1, size_t n1,
                            const float *mass1, const float *xyz2, size_t natoms2, const uint64_t *idx2, size_t n2,
                            const float *mass2, int at_origin, float R9[9], float t3[3]) {
    MH_CTX(c);
    if (!mass1 || (!at_origin && !mass2)) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "fit_transform: mass pointer is null");
    Sel s1, s2;
    MH_TRY(stage_sel(c, xyz1, natoms1, idx1, n1, mass1, c->m_xyz1, c->m_idx1, c->m_mass1, &s1));
    MH_TRY(stage_sel(c, xyz2, natoms2, idx2, n2, at_origin ? nullptr : mass2, c->m_xyz2, c->m_idx2, c->m_mass2, &s2));
    if (s1.n == s2.n && s1.n != 0) {
        // one gather pass + finalizer (both centres, the centred covariance and Horn's rotation on the device)
        const uint32_t nb = blocks_for(c, s1.n, 1);
        MH_TRY(c->m_partials.reserve((size_t)nb * FS_W * 8));
        MH_TRY(c->m_out.reserve(18 * 4));
        hipLaunchKernelGGL(k_fit_sums<false>, dim3(nb, 1), dim3(RB), 0, c->stream, s1, s2, c->m_partials.as<double>());
        // the finalizer also writes the record into pinned host memory: the call ends with a stream wait, not a copy
        MH_TRY(ensure_pinned(c, 18 * 4));
        hipLaunchKernelGGL(k_fit_final<FS_W>, dim3(1), dim3(64), 0, c->stream, c->m_partials.as<double>(), nb, s1.n,
                           at_origin ? 1 : 0, c->m_out.as<float>(), static_cast<float *>(c->h_pinned));
        MH_HIP(hipGetLastError());
        MH_HIP(hipStreamSynchronize(c->stream));
        float h[18];
        std::memcpy(h, c->h_pinned, sizeof h);
        int st;
        std::memcpy(&st, &h[17], 4);
        if (st) return fail(st, st == MOLAR_HIP_ERR_ZERO_MASS ? "zero mass" : "SVD failed");
        std::memcpy(R9, h, 36);
        std::memcpy(t3, h + 9, 12);
        return MOLAR_HIP_OK;
    }
    float cm[6] = {0, 0, 0, 0, 0, 0};
    if (!at_origin) {
        MH_TRY(com_host(c, s1, cm));        // cm1 (:511)
        MH_TRY(com_host(c, s2, cm + 3));    // cm2 (:512)
    }
    // izip! stops at the shorter selection (:621)
    Sel a = s1, b = s2;
    a.n = b.n = s1.n < s2.n ? s1.n : s2.n;
    MH_TRY(c->m_out.reserve(64));
    MH_HIP(hipMemcpyAsync(c->m_out.p, cm, sizeof cm, hipMemcpyHostToDevice, c->stream));
    double cov[9];
    MH_TRY(reduce1(c, a.n, 9, cov, [&](uint32_t nb, double *part) {
        hipLaunchKernelGGL(k_cov, dim3(nb, 1), dim3(RB), 0, c->stream, a, b, c->m_out.as<float>(),
                           c->m_out.as<float>() + 3, part);
    }));
    double R[9];
    if (!rotation_from_cov(cov, R)) return fail(MOLAR_HIP_ERR_SVD, "SVD failed");
    for (int i = 0; i < 9; ++i) R9[i] = (float)R[i];
    if (at_origin) {
        t3[0] = t3[1] = t3[2] = 0.f;
    } else {
        const V3 rv = mat_vec(R9, v3(-cm[0], -cm[1], -cm[2]));   // Translation(cm2)*rot*Translation(-cm1) (:521)
        t3[0] = cm[3] + rv.x;
        t3[1] = cm[4] + rv.y;
        t3[2] = cm[5] + rv.z;
    }
    return MOLAR_HIP_OK;
}

int molar_hip_center_batch(molar_hip_ctx *c, const float *xyz, size_t natoms, const uint64_t *idx,
                           const uint64_t *offsets, size_t nsel, const float *mass, float *out) {
    MH_CTX(c);
    if (!xyz || !idx || !offsets || !out) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "center_batch: null argument");
    if (nsel == 0) return MOLAR_HIP_OK;
    uint64_t last = 0;
    if (is_device_ptr(offsets)) MH_HIP(hipMemcpy(&last, offsets + nsel, 8, hipMemcpyDeviceToHost));
    else last = offsets[nsel];
    const float *d_xyz, *d_mass = nullptr;
    const uint64_t *d_idx, *d_off;
    MH_TRY(to_device(c, xyz, natoms * 3, c->m_xyz1, &d_xyz));
    MH_TRY(to_device(c, idx, (size_t)last, c->m_idx1, &d_idx));
    MH_TRY(to_device(c, offsets, nsel + 1, c->m_idx2, &d_off));
    if (mass) MH_TRY(to_device(c, mass, natoms, c->m_mass1, &d_mass));
    const bool out_dev = is_device_ptr(out);
    MH_TRY(c->m_out.reserve(nsel * 12 + 16));
    float *d_out = out_dev ? out : c->m_out.as<float>();
    MH_TRY(c->m_results.reserve(64));
    int *status = c->m_results.as<int>();
    MH_HIP(hipMemsetAsync(status, 0, 4, c->stream));
    MH_TRY(enqueue_center_batch(c, d_xyz, d_idx, d_off, (uint32_t)nsel, d_mass, d_out, status));
    int st = 0;
    MH_TRY(pull(c, &st, status, 4));
    if (st) return fail(st, "zero mass");
    if (!out_dev) {
        MH_HIP(hipMemcpyAsync(out, d_out, nsel * 12, hipMemcpyDeviceToHost, c->stream));
        MH_HIP(hipStreamSynchronize(c->stream));
    }
    return MOLAR_HIP_OK;
}

int molar_hip_unwrap_simple_batch(molar_hip_ctx *c, float *xyz, size_t natoms, const uint64_t *idx,
                                  const uint64_t *offsets, size_t nsel, const float *box9, uint8_t pbc) {
    MH_CTX(c);
    if (!xyz || !idx || !offsets) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "unwrap_simple_batch: null argument");
    molar_hip_box b;
    MH_TRY(box_or_err(box9, &b));
    if (nsel == 0) return MOLAR_HIP_OK;
    uint64_t last = 0;
    if (is_device_ptr(offsets)) MH_HIP(hipMemcpy(&last, offsets + nsel, 8, hipMemcpyDeviceToHost));
    else last = offsets[nsel];
    const float *d_xyz;
    const uint64_t *d_idx, *d_off;
    MH_TRY(to_device(c, (const float *)xyz, natoms * 3, c->m_xyz1, &d_xyz));
    MH_TRY(to_device(c, idx, (size_t)last, c->m_idx1, &d_idx));
    MH_TRY(to_device(c, offsets, nsel + 1, c->m_idx2, &d_off));
    MH_TRY(enqueue_unwrap_batch(c, const_cast<float *>(d_xyz), d_idx, d_off, (uint32_t)nsel, b, pbc));
    if (!is_device_ptr(xyz)) MH_HIP(hipMemcpyAsync(xyz, d_xyz, natoms * 12, hipMemcpyDeviceToHost, c->stream));
    MH_HIP(hipStreamSynchronize(c->stream));
    return MOLAR_HIP_OK;
}

// host-only: f32, the reference's loop order (molar_membrane/src/lib.rs:456-505)
int molar_hip_membrane_initial_normals(size_t K, const float *head, const float *tail, const uint64_t *poff,
                                       const uint64_t *pids, const uint8_t *valid, float *normals) {
    if (!head || !tail || !poff || !normals) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "initial_normals: null argument");
    struct V { float x, y, z; };
    auto nrm = [](V a) { return std::sqrt((a.x * a.x + a.y * a.y) + a.z * a.z); };
    auto unit = [&](V a) { const float n = nrm(a); return V{a.x / n, a.y / n, a.z / n}; };
    // nalgebra Vector::angle with the two norms handed in: they are functions of one vector each, so computing
    // them once per lipid instead of once per pair gives the same bits
    const float half_pi = 1.57079632679489661923f;
    // "angle <= FRAC_PI_2" (lib.rs:472-473, 494).  acos is only evaluated when the cosine is within 1e-6 of zero:
    // outside that band the comparison cannot depend on how acos rounds (the spacing of f32 near pi/2 is 1.2e-7)
    auto within_half_pi = [&](V a, float n1, V b, float n2) {
        if (n1 == 0.0f || n2 == 0.0f) return true;               // Vector::angle returns 0
        float cc = ((a.x * b.x + a.y * b.y) + a.z * b.z) / (n1 * n2);
        if (cc > 1.0e-6f) return true;
        if (cc < -1.0e-6f) return false;
        cc = cc < -1.0f ? -1.0f : (cc > 1.0f ? 1.0f : cc);       // NaN falls through to acos like the reference
        return std::acos(cc) <= half_pi;
    };
    std::vector<V> thv(K), nv(K);
    std::vector<float> len(K);
    auto ok = [&](size_t i) { return !valid || valid[i]; };
    for (size_t i = 0; i < K; ++i) {
        thv[i] = V{0, 0, 0};
        nv[i] = V{normals[3 * i], normals[3 * i + 1], normals[3 * i + 2]};
        if (ok(i)) thv[i] = unit(V{head[3 * i] - tail[3 * i], head[3 * i + 1] - tail[3 * i + 1], head[3 * i + 2] - tail[3 * i + 2]});
    }
    for (int pass = 0; pass < 2; ++pass) {
        const std::vector<V> &src = pass == 0 ? thv : nv;   // pass 2 reads normals already updated for l < i
        for (size_t i = 0; i < K; ++i) len[i] = nrm(src[i]);
        for (size_t i = 0; i < K; ++i) {
            if (!ok(i)) continue;
            const V self = src[i];
            const float nself = len[i];
            V sum{0, 0, 0};
            for (uint64_t q = poff[i]; q < poff[i + 1]; ++q) {
                const V o = src[pids[q]];
                if (within_half_pi(o, len[pids[q]], self, nself)) { sum.x += o.x; sum.y += o.y; sum.z += o.z; }
            }
            sum.x += self.x; sum.y += self.y; sum.z += self.z;   // .chain(once(central))
            nv[i] = unit(sum);
            if (pass == 1) len[i] = nrm(nv[i]);                  // src aliases nv in pass 2: keep its norm current
        }
    }
    for (size_t i = 0; i < K; ++i) {
        normals[3 * i] = nv[i].x; normals[3 * i + 1] = nv[i].y; normals[3 * i + 2] = nv[i].z;
    }
    return MOLAR_HIP_OK;
}

// Membrane::compute_patches' list building (molar_membrane/src/lib.rs:548-557): for (i, j) in pair order,
// patch_ids[i].push(j); patch_ids[j].push(i) - as a CSR over lipid ids (two counting passes, host).
int molar_hip_membrane_patches_from_pairs(const uint32_t *pairs, size_t npairs, size_t K, uint64_t *patch_offsets,
                                          uint64_t *patch_ids) {
    if ((!pairs && npairs) || !patch_offsets || (!patch_ids && npairs))
        return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "patches_from_pairs: null argument");
    std::vector<uint64_t> cnt(K + 1, 0);
    for (size_t p = 0; p < 2 * npairs; ++p) {
        if (pairs[p] >= K) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "patches_from_pairs: id %u out of range", pairs[p]);
        cnt[pairs[p] + 1]++;
    }
    for (size_t i = 0; i < K; ++i) cnt[i + 1] += cnt[i];
    std::memcpy(patch_offsets, cnt.data(), (K + 1) * 8);
    for (size_t p = 0; p < npairs; ++p) {
        const uint32_t i = pairs[2 * p], j = pairs[2 * p + 1];
        patch_ids[cnt[i]++] = j;
        patch_ids[cnt[j]++] = i;
    }
    return MOLAR_HIP_OK;
}

int molar_hip_lipid_tail_order(molar_hip_ctx *c, const float *xyz, size_t natoms, const uint64_t *idx,
                               const uint64_t *tail_offsets, size_t ntails, int order_type, const float *normals,
                               const uint64_t *normal_offsets, const uint8_t *bond_orders, float *out) {
    MH_CTX(c);
    if (!xyz || !idx || !tail_offsets || !normals || !normal_offsets || !out)
        return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "lipid_tail_order: null argument");
    if (order_type < 0 || order_type > 2) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "lipid_tail_order: order_type %d", order_type);
    if (!bond_orders && order_type != 0) return fail(MOLAR_HIP_ERR_LIPID_BOND_ORDER_COUNT, "bond orders missing");
    if (ntails == 0) return MOLAR_HIP_OK;
    // the CSR offsets are needed on the host to size the transfers
    std::vector<uint64_t> toff(ntails + 1), noff(ntails + 1);
    if (is_device_ptr(tail_offsets)) MH_HIP(hipMemcpy(toff.data(), tail_offsets, (ntails + 1) * 8, hipMemcpyDeviceToHost));
    else std::memcpy(toff.data(), tail_offsets, (ntails + 1) * 8);
    if (is_device_ptr(normal_offsets)) MH_HIP(hipMemcpy(noff.data(), normal_offsets, (ntails + 1) * 8, hipMemcpyDeviceToHost));
    else std::memcpy(noff.data(), normal_offsets, (ntails + 1) * 8);
    const size_t nidx = toff[ntails], nnorm = noff[ntails];
    if (nidx < 2 * ntails) return fail(MOLAR_HIP_ERR_LIPID_TAIL_TOO_SHORT, "tail should have at least 3 carbons");
    const size_t nout = nidx - 2 * ntails, nbond = nidx - ntails;
    const float *d_xyz, *d_norm;
    const uint64_t *d_idx, *d_toff, *d_noff;
    const uint8_t *d_bo = nullptr;
    MH_TRY(to_device(c, xyz, natoms * 3, c->m_xyz1, &d_xyz));
    MH_TRY(to_device(c, idx, nidx, c->m_idx1, &d_idx));
    MH_TRY(to_device(c, normals, nnorm * 3, c->m_xyz2, &d_norm));
    MH_TRY(to_device(c, toff.data(), ntails + 1, c->m_idx2, &d_toff));
    MH_TRY(to_device(c, noff.data(), ntails + 1, c->m_mass2, &d_noff));
    static const uint8_t dummy = 1;
    if (bond_orders) MH_TRY(to_device(c, bond_orders, nbond, c->m_mass1, &d_bo));
    else MH_TRY(to_device(c, &dummy, 1, c->m_mass1, &d_bo));
    const bool out_dev = is_device_ptr(out);
    MH_TRY(c->m_out.reserve((nout + 4) * 4));
    float *d_out = out_dev ? out : c->m_out.as<float>();
    MH_TRY(c->m_results.reserve(64));
    int *status = c->m_results.as<int>();
    MH_HIP(hipMemsetAsync(status, 0, 4, c->stream));
    MH_TRY(enqueue_lipid_order(c, d_xyz, d_idx, d_toff, (uint32_t)ntails, order_type, d_norm, d_noff, d_bo, d_out, status));
    int st = 0;
    MH_TRY(pull(c, &st, status, 4));
    if (st) return fail(st, "lipid order error (status %d)", st);
    if (!out_dev && nout) {
        MH_HIP(hipMemcpyAsync(out, d_out, nout * 4, hipMemcpyDeviceToHost, c->stream));
        MH_HIP(hipStreamSynchronize(c->stream));
    }
    return MOLAR_HIP_OK;
}

int molar_hip_apply_transform(molar_hip_ctx *c, float *xyz, size_t natoms, const uint64_t *idx, size_t n,
                              const float R9[9], const float t3[3]) {
    MH_CTX(c);
    Sel s;
    MH_TRY(stage_sel(c, xyz, natoms, idx, n, nullptr, c->m_xyz1, c->m_idx1, c->m_mass1, &s));
    float Rt[18] = {};       // the record layout of k_fit_final: R, t, ..., status 0
    std::memcpy(Rt, R9, 36);
    std::memcpy(Rt + 9, t3, 12);
    MH_TRY(c->m_out.reserve(sizeof Rt));
    MH_HIP(hipMemcpyAsync(c->m_out.p, Rt, sizeof Rt, hipMemcpyHostToDevice, c->stream));
    const uint32_t nb = blocks_for(c, s.n, 1);
    hipLaunchKernelGGL(k_apply_batch, dim3(nb, 1), dim3(RB), 0, c->stream, s, const_cast<float *>(s.xyz), c->m_out.as<float>());
    MH_HIP(hipGetLastError());
    if (!is_device_ptr(xyz)) MH_HIP(hipMemcpyAsync(xyz, s.xyz, natoms * 12, hipMemcpyDeviceToHost, c->stream));
    MH_HIP(hipStreamSynchronize(c->stream));
    return MOLAR_HIP_OK;
}

int molar_hip_unwrap_simple(molar_hip_ctx *c, float *xyz, size_t natoms, const uint64_t *idx, size_t n,
                            const float *box9, uint8_t pbc) {
    MH_CTX(c);
    molar_hip_box b;
    MH_TRY(box_or_err(box9, &b));
    if (idx ? n == 0 : natoms == 0) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "unwrap_simple of an empty selection");
    Sel s;
    MH_TRY(stage_sel(c, xyz, natoms, idx, n, nullptr, c->m_xyz1, c->m_idx1, c->m_mass1, &s));
    if (s.n > 1) {
        const uint32_t nb = blocks_for(c, s.n, 1);
        hipLaunchKernelGGL(k_unwrap, dim3(nb), dim3(RB), 0, c->stream, s, const_cast<float *>(s.xyz), b,
                           (uint32_t)(pbc & 7u));
        MH_HIP(hipGetLastError());
    }
    if (!is_device_ptr(xyz)) MH_HIP(hipMemcpyAsync(xyz, s.xyz, natoms * 12, hipMemcpyDeviceToHost, c->stream));
    MH_HIP(hipStreamSynchronize(c->stream));
    return MOLAR_HIP_OK;
}

int molar_hip_fit_rmsd_batch(molar_hip_ctx *c, float *frames, size_t nframes, size_t natoms, const uint64_t *idx,
                             size_t n, const float *mass, const float *ref_xyz, size_t ref_natoms,
                             const uint64_t *ref_idx, int apply, float *rmsd_out, float *R_out, float *t_out,
                             float *com_out, float *gyr_out) {
    MH_CTX(c);
    if (!frames || !mass || !ref_xyz) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "fit_rmsd_batch: null argument");
    if (nframes == 0) return MOLAR_HIP_OK;
    if (nframes > 65535) return fail(MOLAR_HIP_ERR_TOO_LARGE, "fit_rmsd_batch: at most 65535 frames per call");
    const size_t nsel = idx ? n : natoms, nref = ref_idx ? n : ref_natoms;
    if (nsel != nref) return fail(MOLAR_HIP_ERR_SIZES, "incompatible sizes: %zu and %zu", nsel, nref);
    Sel cur, ref;
    MH_TRY(stage_sel(c, frames, natoms, idx, n, mass, c->m_xyz1, c->m_idx1, c->m_mass1, &cur, nframes));
    cur.frame_stride = natoms * 3;
    // masses of the reference selection: fit_transform takes sel2's own masses for cm2 (:512); the
    // per-frame loop fits a selection onto the same atoms of the reference frame, i.e. same column
    MH_TRY(stage_sel(c, ref_xyz, ref_natoms, ref_idx, n, nullptr, c->m_xyz2, c->m_idx2, c->m_mass2, &ref));
    ref.mass = cur.mass;
    const uint32_t F = (uint32_t)nframes;
    const uint32_t nb = blocks_for(c, cur.n, F);
    const size_t part_bytes = (((size_t)nb * F * FS_ALL * 8) + 255) & ~(size_t)255;
    const size_t qm_bytes = (((size_t)cur.n * 16) + 255) & ~(size_t)255;
    MH_TRY(c->m_partials.reserve(part_bytes + qm_bytes + (size_t)cur.n * 8 + 256));
    MH_TRY(c->m_out.reserve((size_t)F * 18 * 4));
    float *o = c->m_out.as<float>();
    double *part = c->m_partials.as<double>();
    float4 *pk_qm = reinterpret_cast<float4 *>(static_cast<char *>(c->m_partials.p) + part_bytes);
    uint2 *pk_am = reinterpret_cast<uint2 *>(static_cast<char *>(c->m_partials.p) + part_bytes + qm_bytes);
    {
        // one gather pass (batches: behind the once-per-call packing of the frame-invariant columns), one finalizer,
        // and the write pass only if the caller wants the frames moved
        Prof prof(c, 4);
        if (F >= 4) {
            // frame-invariant columns gathered once (k_fit_pack), then one scattered read per atom and frame
            hipLaunchKernelGGL(k_fit_pack, dim3((cur.n + RB - 1) / RB), dim3(RB), 0, c->stream, cur, ref, pk_qm, pk_am);
            hipLaunchKernelGGL(k_fit_sums_packed<true>, dim3(nb, F), dim3(RB), 0, c->stream, cur.xyz, cur.frame_stride,
                               cur.n, pk_qm, pk_am, part);
        } else {
            hipLaunchKernelGGL(k_fit_sums<true>, dim3(nb, F), dim3(RB), 0, c->stream, cur, ref, part);
        }
        MH_TRY(ensure_pinned(c, (size_t)F * 18 * 4));
        // (the records also land in pinned host memory: the call ends with a stream wait instead of a device-to-host copy)
        hipLaunchKernelGGL(k_fit_final<FS_ALL>, dim3(F), dim3(64), 0, c->stream, part, nb, cur.n, 0, o,
                           static_cast<float *>(c->h_pinned));
        if (apply)
            hipLaunchKernelGGL(k_apply_batch, dim3(nb, F), dim3(RB), 0, c->stream, cur, const_cast<float *>(cur.xyz), o);
    }
    MH_HIP(hipGetLastError());
    MH_HIP(hipStreamSynchronize(c->stream));
    std::vector<float> h((size_t)F * 18);
    std::memcpy(h.data(), c->h_pinned, h.size() * 4);
    for (uint32_t f = 0; f < F; ++f) {
        int st;
        std::memcpy(&st, &h[18 * (size_t)f + 17], 4);
        if (st) return fail(st, st == MOLAR_HIP_ERR_ZERO_MASS ? "zero mass" : "SVD failed");
    }
    auto emit = [&](float *dst, size_t count, auto getter) -> int {
        if (!dst) return 0;
        std::vector<float> tmp(count);
        for (size_t k = 0; k < count; ++k) tmp[k] = getter(k);
        if (is_device_ptr(dst)) {
            MH_HIP(hipMemcpyAsync(dst, tmp.data(), count * 4, hipMemcpyHostToDevice, c->stream));
            MH_HIP(hipStreamSynchronize(c->stream));      // tmp dies with this scope
        } else {
            std::memcpy(dst, tmp.data(), count * 4);
        }
        return 0;
    };
    const float *r18 = h.data();
    MH_TRY(emit(rmsd_out, F, [&](size_t k) { return r18[18 * k + 12]; }));
    MH_TRY(emit(gyr_out, F, [&](size_t k) { return r18[18 * k + 16]; }));
    MH_TRY(emit(com_out, (size_t)F * 3, [&](size_t k) { return r18[18 * (k / 3) + 13 + (k % 3)]; }));
    MH_TRY(emit(R_out, (size_t)F * 9, [&](size_t k) { return r18[18 * (k / 9) + (k % 9)]; }));
    MH_TRY(emit(t_out, (size_t)F * 3, [&](size_t k) { return r18[18 * (k / 3) + 9 + (k % 3)]; }));
    if (apply && !is_device_ptr(frames))
        MH_HIP(hipMemcpyAsync(frames, cur.xyz, nframes * natoms * 12, hipMemcpyDeviceToHost, c->stream));
    MH_HIP(hipStreamSynchronize(c->stream));
    return MOLAR_HIP_OK;
}


// ---- CSR-batched entries (one wave per selection)

// last offset = number of index entries; offsets may live on either side
static int csr_total(const uint64_t *offsets, size_t nsel, uint64_t *last) {
    if (is_device_ptr(offsets)) MH_HIP(hipMemcpy(last, offsets + nsel, 8, hipMemcpyDeviceToHost));
    else *last = offsets[nsel];
    return 0;
}

static int finish_batch(molar_hip_ctx *c, int *status, float *out, const float *d_out, size_t count, const char *what) {
    MH_HIP(hipGetLastError());
    int st = 0;
    MH_TRY(pull(c, &st, status, 4));
    if (st) return fail(st, "%s: %s", what, st == MOLAR_HIP_ERR_ZERO_MASS ? "zero mass" : "empty selection");
    if (out != d_out && count) {
        MH_HIP(hipMemcpyAsync(out, d_out, count * 4, hipMemcpyDeviceToHost, c->stream));
        MH_HIP(hipStreamSynchronize(c->stream));
    }
    return MOLAR_HIP_OK;
}

int molar_hip_gyration_batch(molar_hip_ctx *c, const float *xyz, size_t natoms, const uint64_t *idx, const uint64_t *offsets,
                             size_t nsel, const float *mass, const float *box9, float *out) {
    MH_CTX(c);
    if (!xyz || !idx || !offsets || !mass || !out) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "gyration_batch: null argument");
    if (nsel == 0) return MOLAR_HIP_OK;
    molar_hip_box b{};
    if (box9) MH_TRY(molar_hip_box_from_matrix(box9, &b));
    uint64_t last = 0;
    MH_TRY(csr_total(offsets, nsel, &last));
    const float *d_xyz, *d_mass;
    const uint64_t *d_idx, *d_off;
    MH_TRY(to_device(c, xyz, natoms * 3, c->m_xyz1, &d_xyz));
    MH_TRY(to_device(c, idx, (size_t)last, c->m_idx1, &d_idx));
    MH_TRY(to_device(c, offsets, nsel + 1, c->m_idx2, &d_off));
    MH_TRY(to_device(c, mass, natoms, c->m_mass1, &d_mass));
    MH_TRY(c->m_out.reserve(nsel * 4 + 16));
    float *d_out = is_device_ptr(out) ? out : c->m_out.as<float>();
    MH_TRY(c->m_results.reserve(64));
    int *status = c->m_results.as<int>();
    MH_HIP(hipMemsetAsync(status, 0, 4, c->stream));
    hipLaunchKernelGGL(k_gyration_batch, dim3((unsigned)((nsel + 3) / 4)), dim3(256), 0, c->stream, d_xyz, d_idx, d_off,
                       (uint32_t)nsel, d_mass, box9 ? 1 : 0, b, d_out, status);
    return finish_batch(c, status, out, d_out, nsel, "gyration_batch");
}

int molar_hip_rmsd_batch(molar_hip_ctx *c, const float *xyz1, size_t natoms1, const uint64_t *idx1, const float *xyz2,
                         size_t natoms2, const uint64_t *idx2, const uint64_t *offsets, size_t nsel, const float *mass1,
                         float *out) {
    MH_CTX(c);
    if (!xyz1 || !xyz2 || !idx1 || !offsets || !out) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "rmsd_batch: null argument");
    if (nsel == 0) return MOLAR_HIP_OK;
    uint64_t last = 0;
    MH_TRY(csr_total(offsets, nsel, &last));
    const float *d1, *d2, *d_mass = nullptr;
    const uint64_t *i1, *i2, *d_off;
    MH_TRY(to_device(c, xyz1, natoms1 * 3, c->m_xyz1, &d1));
    MH_TRY(to_device(c, xyz2, natoms2 * 3, c->m_xyz2, &d2));
    MH_TRY(to_device(c, idx1, (size_t)last, c->m_idx1, &i1));
    if (idx2 && idx2 != idx1) MH_TRY(to_device(c, idx2, (size_t)last, c->m_idx2, &i2));
    else i2 = i1;
    MH_TRY(c->m_partials.reserve((nsel + 1) * 8));
    MH_TRY(to_device(c, offsets, nsel + 1, c->m_partials, &d_off));
    if (mass1) MH_TRY(to_device(c, mass1, natoms1, c->m_mass1, &d_mass));
    MH_TRY(c->m_out.reserve(nsel * 4 + 16));
    float *d_out = is_device_ptr(out) ? out : c->m_out.as<float>();
    MH_TRY(c->m_results.reserve(64));
    int *status = c->m_results.as<int>();
    MH_HIP(hipMemsetAsync(status, 0, 4, c->stream));
    hipLaunchKernelGGL(k_rmsd_batch, dim3((unsigned)((nsel + 3) / 4)), dim3(256), 0, c->stream, d1, d2, i1, i2, d_off,
                       (uint32_t)nsel, d_mass, d_out, status);
    return finish_batch(c, status, out, d_out, nsel, "rmsd_batch");
}

int molar_hip_fit_batch(molar_hip_ctx *c, float *xyz1, size_t natoms1, const uint64_t *idx1, const float *mass1,
                        const float *xyz2, size_t natoms2, const uint64_t *idx2, const float *mass2, const uint64_t *offsets,
                        size_t nsel, int apply, float *R_out, float *t_out, float *rmsd_out, float *com_out, float *gyr_out) {
    MH_CTX(c);
    if (!xyz1 || !xyz2 || !idx1 || !mass1 || !offsets) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "fit_batch: null argument");
    if (nsel == 0) return MOLAR_HIP_OK;
    uint64_t last = 0;
    MH_TRY(csr_total(offsets, nsel, &last));
    const float *d1, *d2, *m1, *m2 = nullptr;
    const uint64_t *i1, *i2, *d_off;
    MH_TRY(to_device(c, (const float *)xyz1, natoms1 * 3, c->m_xyz1, &d1));
    MH_TRY(to_device(c, xyz2, natoms2 * 3, c->m_xyz2, &d2));
    MH_TRY(to_device(c, idx1, (size_t)last, c->m_idx1, &i1));
    if (idx2 && idx2 != idx1) MH_TRY(to_device(c, idx2, (size_t)last, c->m_idx2, &i2));
    else i2 = i1;
    MH_TRY(c->m_partials.reserve((nsel + 1) * 8));
    MH_TRY(to_device(c, offsets, nsel + 1, c->m_partials, &d_off));
    MH_TRY(to_device(c, mass1, natoms1, c->m_mass1, &m1));
    if (mass2 && mass2 != mass1) MH_TRY(to_device(c, mass2, natoms2, c->m_mass2, &m2));
    MH_TRY(c->m_out.reserve(nsel * 18 * 4));
    float *o = c->m_out.as<float>();
    hipLaunchKernelGGL(k_fit_csr, dim3((unsigned)((nsel + 3) / 4)), dim3(256), 0, c->stream, const_cast<float *>(d1), d2, i1, i2,
                       d_off, (uint32_t)nsel, m1, m2, apply ? 1 : 0, o);
    MH_HIP(hipGetLastError());
    std::vector<float> h(nsel * 18);
    MH_TRY(pull(c, h.data(), o, h.size() * 4));
    for (size_t k = 0; k < nsel; ++k) {
        int st;
        std::memcpy(&st, &h[18 * k + 17], 4);
        if (st) return fail(st, "fit_batch, selection %zu: %s", k, st == MOLAR_HIP_ERR_ZERO_MASS ? "zero mass" : "SVD failed");
    }
    auto emit = [&](float *dst, size_t per, size_t first) -> int {
        if (!dst) return 0;
        std::vector<float> tmp(nsel * per);
        for (size_t k = 0; k < nsel; ++k)
            for (size_t q = 0; q < per; ++q) tmp[k * per + q] = h[18 * k + first + q];
        if (is_device_ptr(dst)) {
            MH_HIP(hipMemcpyAsync(dst, tmp.data(), tmp.size() * 4, hipMemcpyHostToDevice, c->stream));
            MH_HIP(hipStreamSynchronize(c->stream));
        } else {
            std::memcpy(dst, tmp.data(), tmp.size() * 4);
        }
        return 0;
    };
    MH_TRY(emit(R_out, 9, 0));
    MH_TRY(emit(t_out, 3, 9));
    MH_TRY(emit(rmsd_out, 1, 12));
    MH_TRY(emit(com_out, 3, 13));
    MH_TRY(emit(gyr_out, 1, 16));
    if (apply && !is_device_ptr(xyz1)) {
        MH_HIP(hipMemcpyAsync(xyz1, d1, natoms1 * 12, hipMemcpyDeviceToHost, c->stream));
        MH_HIP(hipStreamSynchronize(c->stream));
    }
    return MOLAR_HIP_OK;
}

// ---- Modify::translate / rotate (modify.rs:16-30) and Measure::principal_transform(_pbc) (measure.rs:102-109,246-257)

int molar_hip_translate(molar_hip_ctx *c, float *xyz, size_t natoms, const uint64_t *idx, size_t n, const float shift3[3]) {
    MH_CTX(c);
    if (!shift3) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "translate: null shift");
    Sel s;
    MH_TRY(stage_sel(c, xyz, natoms, idx, n, nullptr, c->m_xyz1, c->m_idx1, c->m_mass1, &s));
    if (s.n) {
        hipLaunchKernelGGL(k_translate, dim3(blocks_for(c, s.n, 1)), dim3(RB), 0, c->stream, s, const_cast<float *>(s.xyz),
                           shift3[0], shift3[1], shift3[2]);
        MH_HIP(hipGetLastError());
    }
    if (!is_device_ptr(xyz)) MH_HIP(hipMemcpyAsync(xyz, s.xyz, natoms * 12, hipMemcpyDeviceToHost, c->stream));
    MH_HIP(hipStreamSynchronize(c->stream));
    return MOLAR_HIP_OK;
}

int molar_hip_rotate(molar_hip_ctx *c, float *xyz, size_t natoms, const uint64_t *idx, size_t n, const float unit_axis3[3],
                     float angle) {
    if (!unit_axis3) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "rotate: null axis");
    // nalgebra Rotation3::from_axis_angle (Rodrigues' formula on a unit axis), f32, column-major
    const float ux = unit_axis3[0], uy = unit_axis3[1], uz = unit_axis3[2];
    const float sn = std::sin(angle), cs = std::cos(angle), k = 1.0f - cs;
    const float sqx = ux * ux, sqy = uy * uy, sqz = uz * uz;
    const float R[9] = {sqx + (1.0f - sqx) * cs, ux * uy * k + uz * sn, ux * uz * k - uy * sn,
                        ux * uy * k - uz * sn, sqy + (1.0f - sqy) * cs, uy * uz * k + ux * sn,
                        ux * uz * k + uy * sn, uy * uz * k - ux * sn, sqz + (1.0f - sqz) * cs};
    const float t[3] = {0.f, 0.f, 0.f};
    return molar_hip_apply_transform(c, xyz, natoms, idx, n, R, t);     // p.coords = tr * p.coords (:28)
}

int molar_hip_principal_transform(molar_hip_ctx *c, const float *xyz, size_t natoms, const uint64_t *idx, size_t n,
                                  const float *mass, const float *box9, float R9[9], float t3[3]) {
    MH_CTX(c);
    if (!R9 || !t3) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "principal_transform: null output");
    float mom[3], axes[9], cm[3];
    MH_TRY(molar_hip_inertia(c, xyz, natoms, idx, n, mass, box9, mom, axes, nullptr));
    if (box9) MH_TRY(molar_hip_center_of_mass_pbc(c, xyz, natoms, idx, n, mass, box9, MOLAR_HIP_PBC_FULL, cm));   // (:251)
    else MH_TRY(molar_hip_center_of_mass(c, xyz, natoms, idx, n, mass, cm));                                   // (:106)
    // do_principal_transform (:646-649): Translation(cm) * Rotation(axes^-1) * Translation(-cm);
    // try_inverse_mut leaves a singular matrix untouched (closed-form 3x3 inverse, nalgebra)
    const float *m = axes;     // column-major: m11 m21 m31 | m12 m22 m32 | m13 m23 m33
    const float m11 = m[0], m21 = m[1], m31 = m[2], m12 = m[3], m22 = m[4], m32 = m[5], m13 = m[6], m23 = m[7], m33 = m[8];
    const float mi1 = m22 * m33 - m32 * m23, mi2 = m21 * m33 - m31 * m23, mi3 = m21 * m32 - m31 * m22;
    const float det = (m11 * mi1 - m12 * mi2) + m13 * mi3;
    for (int i = 0; i < 9; ++i) R9[i] = axes[i];
    if (det != 0.0f) {
        const float inv[9] = {mi1 / det, -mi2 / det, mi3 / det,
                              (m13 * m32 - m33 * m12) / det, (m11 * m33 - m31 * m13) / det, (m12 * m31 - m32 * m11) / det,
                              (m12 * m23 - m22 * m13) / det, (m13 * m21 - m23 * m11) / det, (m11 * m22 - m21 * m12) / det};
        for (int i = 0; i < 9; ++i) R9[i] = inv[i];
    }
    const V3 rv = mat_vec(R9, v3(-cm[0], -cm[1], -cm[2]));
    t3[0] = cm[0] + rv.x;
    t3[1] = cm[1] + rv.y;
    t3[2] = cm[2] + rv.z;
    return MOLAR_HIP_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------------
// The per-frame fit of a trajectory whose frames live in HOST memory (a State's coords, state.rs:22-28, handed to the task by
// analysis_task.rs:245-252) at the rate the selection - not the frame - crosses the link.  molar_hip_fit_rmsd_batch on a host
// frame stages all natoms x 12 bytes from pageable memory per call (1.5 k frames/s for the 1M-atom frame of BASELINE config 3,
// whose selection is 100k atoms).  Here the SELECTED atoms are taken out of the frame by a small pool of host threads into pinned
// staging, sent (1.2 MB instead of 12), fitted by the kernels of the batch entry on the packed selection - same terms, same
// order, same block layout: the record is bit-identical to the batch entry's - and, with `apply`, the moved selection comes back
// the same way and is scattered into the caller's frame at _end.  Three frames may be in flight: gather k+1 on the host, copy
// and fit k on the stream, scatter k-1.
struct molar_hip_fit_stream {
    molar_hip_ctx *c = nullptr;
    size_t natoms = 0;
    uint32_t n = 0;
    std::vector<uint64_t> idx;           // the selection (host copy; empty = identity)
    mh::DevBuf ref_pk, mass_pk, mass_ref_pk;   // packed reference selection [n][3], masses of the frame's / the reference's selected atoms [n]
    static constexpr int NSLOT = 3;
    struct Slot {
        float *h_in = nullptr, *h_out = nullptr;   // pinned: packed selection in / moved selection out
        float *h_rec = nullptr;                    // pinned: the 18-float record of k_fit_final
        mh::DevBuf cur, part, out;
        hipEvent_t done = nullptr;
        float *frame = nullptr;                    // the caller's frame (apply: scattered into at _end)
        bool pending = false, apply = false;
    } slot[NSLOT];
    int next = 0;
    // host threads for the gather / scatter of the selection
    std::vector<std::thread> pool;
    std::mutex m;
    std::condition_variable cv, cv_done;
    std::function<void(size_t, size_t)> job;       // [lo, hi) of the selection
    size_t job_serial = 0, job_left = 0;
    bool quit = false;
    void parallel(const std::function<void(size_t, size_t)> &f) {
        if (pool.empty() || n < 4096u) { f(0, n); return; }
        {
            std::lock_guard<std::mutex> lk(m);
            job = f;
            job_left = pool.size();
            ++job_serial;
        }
        cv.notify_all();
        std::unique_lock<std::mutex> lk(m);
        cv_done.wait(lk, [&] { return job_left == 0; });
    }
};

extern "C" {

void molar_hip_fit_stream_destroy(molar_hip_fit_stream *s) {
    if (!s) return;
    {
        std::lock_guard<std::mutex> lk(s->m);
        s->quit = true;
    }
    s->cv.notify_all();
    for (auto &t : s->pool) t.join();
    if (s->c) {
        (void)hipSetDevice(s->c->device);
        (void)hipStreamSynchronize(s->c->stream);
    }
    for (auto &sl : s->slot) {
        if (sl.h_in) (void)hipHostFree(sl.h_in);
        if (sl.h_out) (void)hipHostFree(sl.h_out);
        if (sl.h_rec) (void)hipHostFree(sl.h_rec);
        if (sl.done) (void)hipEventDestroy(sl.done);
        sl.cur.release(); sl.part.release(); sl.out.release();
    }
    s->ref_pk.release();
    s->mass_pk.release();
    s->mass_ref_pk.release();
    delete s;
}

int molar_hip_fit_stream_create(molar_hip_ctx *c, size_t natoms, const uint64_t *idx, size_t n, const float *mass, const float *ref_xyz,
                                size_t ref_natoms, const uint64_t *ref_idx, int host_threads, molar_hip_fit_stream **out) {
    MH_CTX(c);
    if (!out || !mass || !ref_xyz) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "fit_stream_create: null argument");
    *out = nullptr;
    const size_t nsel = idx ? n : natoms, nref = ref_idx ? n : ref_natoms;
    if (nsel != nref) return fail(MOLAR_HIP_ERR_SIZES, "incompatible sizes: %zu and %zu", nsel, nref);
    if (nsel == 0 || nsel >= 0xFFFFFFFFull) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "fit_stream_create: %zu selected atoms", nsel);
    if (is_device_ptr(idx) || is_device_ptr(ref_idx) || is_device_ptr(mass) || is_device_ptr(ref_xyz))
        return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "fit_stream_create: the topology-side arrays are taken from host memory (frames already "
                                                      "in device memory go through molar_hip_fit_rmsd_batch)");
    for (size_t k = 0; idx && k < n; ++k)
        if (idx[k] >= natoms) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "fit_stream_create: index %llu out of range", (unsigned long long)idx[k]);
    for (size_t k = 0; ref_idx && k < n; ++k)
        if (ref_idx[k] >= ref_natoms) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "fit_stream_create: reference index %llu out of range", (unsigned long long)ref_idx[k]);
    auto *s = new molar_hip_fit_stream;
    s->c = c;
    s->natoms = natoms;
    s->n = (uint32_t)nsel;
    if (idx) s->idx.assign(idx, idx + n);
    // the frame-invariant columns, packed once: reference selection and the masses of the frame's selected atoms (the batch
    // entry's convention: the reference centre uses the same column)
    // (the reference centre takes the SAME column through the reference's own index: fit_transform uses sel2's masses, :512)
    if (ref_idx && ref_natoms > natoms) { delete s; return fail(MOLAR_HIP_ERR_SIZES, "fit_stream_create: the mass column has %zu entries, the reference %zu atoms", natoms, ref_natoms); }
    std::vector<float> rp(nsel * 3), mp(nsel), mr(nsel);
    for (size_t k = 0; k < nsel; ++k) {
        const size_t a = idx ? (size_t)idx[k] : k, r = ref_idx ? (size_t)ref_idx[k] : k;
        rp[3 * k] = ref_xyz[3 * r]; rp[3 * k + 1] = ref_xyz[3 * r + 1]; rp[3 * k + 2] = ref_xyz[3 * r + 2];
        mp[k] = mass[a];
        mr[k] = mass[r];
    }
    int rc = s->ref_pk.reserve(nsel * 12);
    if (!rc) rc = s->mass_pk.reserve(nsel * 4);
    if (!rc) rc = s->mass_ref_pk.reserve(nsel * 4);
    hipError_t e = rc ? hipSuccess : hipMemcpy(s->ref_pk.p, rp.data(), nsel * 12, hipMemcpyHostToDevice);
    if (!rc && e == hipSuccess) e = hipMemcpy(s->mass_pk.p, mp.data(), nsel * 4, hipMemcpyHostToDevice);
    if (!rc && e == hipSuccess) e = hipMemcpy(s->mass_ref_pk.p, mr.data(), nsel * 4, hipMemcpyHostToDevice);
    const uint32_t nb = blocks_for(c, s->n, 1);
    for (auto &sl : s->slot) {
        if (rc || e != hipSuccess) break;
        e = hipHostMalloc((void **)&sl.h_in, nsel * 12, hipHostMallocDefault);
        if (e == hipSuccess) e = hipHostMalloc((void **)&sl.h_out, nsel * 12, hipHostMallocDefault);
        if (e == hipSuccess) e = hipHostMalloc((void **)&sl.h_rec, 18 * 4, hipHostMallocDefault);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.done, hipEventDisableTiming);
        if (e == hipSuccess) rc = sl.cur.reserve(nsel * 12);
        if (!rc) rc = sl.part.reserve((size_t)nb * FS_ALL * 8);
        if (!rc) rc = sl.out.reserve(18 * 4);
    }
    if (rc || e != hipSuccess) {
        molar_hip_fit_stream_destroy(s);
        return rc ? rc : fail(MOLAR_HIP_ERR_HIP, "fit_stream_create: %s", hipGetErrorString(e));
    }
    unsigned nt = host_threads > 0 ? (unsigned)host_threads : std::min(8u, std::max(1u, std::thread::hardware_concurrency() / 2u));
    if (nsel < 4096) nt = 0;
    for (unsigned t = 0; t < nt; ++t) {
        s->pool.emplace_back([s, t, nt]() {
            size_t seen = 0;
            for (;;) {
                std::function<void(size_t, size_t)> f;
                {
                    std::unique_lock<std::mutex> lk(s->m);
                    s->cv.wait(lk, [&] { return s->quit || s->job_serial != seen; });
                    if (s->quit) return;
                    seen = s->job_serial;
                    f = s->job;
                }
                const size_t per = ((size_t)s->n + nt - 1) / nt, lo = std::min((size_t)s->n, t * per), hi = std::min((size_t)s->n, lo + per);
                if (lo < hi) f(lo, hi);
                {
                    std::lock_guard<std::mutex> lk(s->m);
                    if (--s->job_left == 0) s->cv_done.notify_all();
                }
            }
        });
    }
    *out = s;
    return MOLAR_HIP_OK;
}

int molar_hip_fit_stream_begin(molar_hip_fit_stream *s, float *xyz, int apply, int32_t *ticket) {
    if (!s || !xyz || !ticket) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "fit_stream_begin: null argument");
    if (is_device_ptr(xyz)) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "fit_stream_begin: the frame is in device memory (use molar_hip_fit_rmsd_batch)");
    molar_hip_ctx *c = s->c;
    MH_CTX(c);
    const int k = s->next;
    auto &sl = s->slot[k];
    if (sl.pending) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "fit_stream_begin: %d frames are in flight: call molar_hip_fit_stream_end first", molar_hip_fit_stream::NSLOT);
    const uint64_t *idx = s->idx.empty() ? nullptr : s->idx.data();
    float *dst = sl.h_in;
    s->parallel([=](size_t lo, size_t hi) {
        if (!idx) { std::memcpy(dst + 3 * lo, xyz + 3 * lo, (hi - lo) * 12); return; }
        for (size_t q = lo; q < hi; ++q) {
            const float *p = xyz + 3 * idx[q];
            dst[3 * q] = p[0]; dst[3 * q + 1] = p[1]; dst[3 * q + 2] = p[2];
        }
    });
    const uint32_t n = s->n, nb = blocks_for(c, n, 1);
    MH_HIP(hipMemcpyAsync(sl.cur.p, sl.h_in, (size_t)n * 12, hipMemcpyHostToDevice, c->stream));
    Sel cur{sl.cur.as<float>(), nullptr, s->mass_pk.as<float>(), n, 0};
    Sel ref{s->ref_pk.as<float>(), nullptr, s->mass_ref_pk.as<float>(), n, 0};
    {
        Prof prof(c, 4);
        hipLaunchKernelGGL(k_fit_sums<true>, dim3(nb, 1), dim3(RB), 0, c->stream, cur, ref, sl.part.as<double>());
        hipLaunchKernelGGL(k_fit_final<FS_ALL>, dim3(1), dim3(64), 0, c->stream, sl.part.as<double>(), nb, n, 0, sl.out.as<float>(), sl.h_rec);
        if (apply) hipLaunchKernelGGL(k_apply_batch, dim3(nb, 1), dim3(RB), 0, c->stream, cur, sl.cur.as<float>(), sl.out.as<float>());
    }
    MH_HIP(hipGetLastError());
    if (apply) MH_HIP(hipMemcpyAsync(sl.h_out, sl.cur.p, (size_t)n * 12, hipMemcpyDeviceToHost, c->stream));
    MH_HIP(hipEventRecord(sl.done, c->stream));
    sl.pending = true;
    sl.apply = apply != 0;
    sl.frame = xyz;
    s->next = (k + 1) % molar_hip_fit_stream::NSLOT;
    *ticket = k;
    return MOLAR_HIP_OK;
}

int molar_hip_fit_stream_end(molar_hip_fit_stream *s, int32_t ticket, float *rmsd, float R9[9], float t3[3], float com3[3], float *gyr) {
    if (!s || ticket < 0 || ticket >= molar_hip_fit_stream::NSLOT || !s->slot[ticket].pending)
        return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "fit_stream_end: no frame in flight with ticket %d", (int)ticket);
    auto &sl = s->slot[ticket];
    MH_CTX(s->c);
    MH_HIP(hipEventSynchronize(sl.done));
    sl.pending = false;
    float h[18];
    std::memcpy(h, sl.h_rec, sizeof h);
    int st;
    std::memcpy(&st, &h[17], 4);
    if (st) return fail(st, st == MOLAR_HIP_ERR_ZERO_MASS ? "zero mass" : "SVD failed");
    if (sl.apply) {          // apply_transform (modify.rs:32-36) on the caller's frame: the moved selection back into its atoms
        const uint64_t *idx = s->idx.empty() ? nullptr : s->idx.data();
        const float *src = sl.h_out;
        float *frame = sl.frame;
        s->parallel([=](size_t lo, size_t hi) {
            if (!idx) { std::memcpy(frame + 3 * lo, src + 3 * lo, (hi - lo) * 12); return; }
            for (size_t q = lo; q < hi; ++q) {
                float *p = frame + 3 * idx[q];
                p[0] = src[3 * q]; p[1] = src[3 * q + 1]; p[2] = src[3 * q + 2];
            }
        });
    }
    if (rmsd) *rmsd = h[12];
    if (gyr) *gyr = h[16];
    if (R9) std::memcpy(R9, h, 36);
    if (t3) std::memcpy(t3, h + 9, 12);
    if (com3) std::memcpy(com3, h + 13, 12);
    return MOLAR_HIP_OK;
}

}  // extern "C"
