// measure_f64.hip — the Measure / Modify methods for MolAR built with its `f64` feature
// (Float = f64: molar/src/aliases.rs:10-13, molar/Cargo.toml:56-60): centres, gyration radius, RMSD, Kabsch fit and
// apply_transform, plus min_max, the inertia tensor with its principal axes and translate, on double-precision
// coordinates and masses.
//
// Structure follows the reference, not the fused f32 path: a centre pass, then a pass over the centred terms
// (gyration :78-87, rot_transform :613-643), every per-atom term formed in f64 in the reference's operation order and
// accumulated per thread -> wave -> workgroup -> fixed-order total, so results agree with the reference's serial f64 sums
// to ~1e-15 relative and do not depend on the launch shape.  These passes move 24-56 bytes per atom: HBM-bound like
// their f32 counterparts (measure.hip); the periodic centres, gyration radius and unwrap_simple carry their own f64
// PeriodicBox (boxmath64.hpp); the f64 search drivers are in search_f64.hip.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "boxmath64.hpp"
#include "common.hpp"
#include "linalg3.hpp"

using namespace mh;

namespace {

constexpr int RB = 256;

struct SelD {
    const double *xyz;
    const uint64_t *idx;
    const double *mass;   // full-length column, gathered through idx
    uint32_t n;
};

__device__ __forceinline__ uint64_t atom_of(const SelD &s, uint32_t k) { return s.idx ? s.idx[k] : (uint64_t)k; }

template <int NV>
__device__ __forceinline__ void block_store(double *acc, double *partials) {
    __shared__ double sh[RB / 64][NV];
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
        for (int w = 0; w < RB / 64; ++w) s += sh[w][threadIdx.x];
        partials[(size_t)blockIdx.x * NV + threadIdx.x] = s;
    }
}

// [0] = sum m, [1..3] = sum p*m, [4..6] = sum p      (center_of_mass :60-75, center_of_geometry :39-47)
__global__ void __launch_bounds__(RB) k64_sums(SelD s, double *partials) {
    double acc[7] = {0, 0, 0, 0, 0, 0, 0};
    for (uint32_t k = blockIdx.x * RB + threadIdx.x; k < s.n; k += gridDim.x * RB) {
        const uint64_t a = atom_of(s, k);
        const double *p = s.xyz + 3 * a;
        const double m = s.mass ? s.mass[a] : 1.0;
        acc[0] += m;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            acc[1 + d] += p[d] * m;
            acc[4 + d] += p[d];
        }
    }
    block_store<7>(acc, partials);
}

// [0] = sum |p - c|^2 * m, [1] = sum m                  (gyration :78-87)
__global__ void __launch_bounds__(RB) k64_central(SelD s, double cx, double cy, double cz, double *partials) {
    double acc[2] = {0, 0};
    for (uint32_t k = blockIdx.x * RB + threadIdx.x; k < s.n; k += gridDim.x * RB) {
        const uint64_t a = atom_of(s, k);
        const double *p = s.xyz + 3 * a;
        const double dx = p[0] - cx, dy = p[1] - cy, dz = p[2] - cz;
        const double m = s.mass[a];
        acc[0] += ((dx * dx + dy * dy) + dz * dz) * m;
        acc[1] += m;
    }
    block_store<2>(acc, partials);
}

// [0] = sum |p - q|^2, [1] = sum |p - q|^2 * m1, [2] = sum m1      (rmsd :485-504, rmsd_mw :538-558)
__global__ void __launch_bounds__(RB) k64_rmsd(SelD s1, SelD s2, double *partials) {
    double acc[3] = {0, 0, 0};
    for (uint32_t k = blockIdx.x * RB + threadIdx.x; k < s1.n; k += gridDim.x * RB) {
        const uint64_t a1 = atom_of(s1, k), a2 = atom_of(s2, k);
        const double *p = s1.xyz + 3 * a1, *q = s2.xyz + 3 * a2;
        const double dx = q[0] - p[0], dy = q[1] - p[1], dz = q[2] - p[2];
        const double d2 = (dx * dx + dy * dy) + dz * dz;
        const double m = s1.mass ? s1.mass[a1] : 1.0;
        acc[0] += d2;
        acc[1] += d2 * m;
        acc[2] += m;
    }
    block_store<3>(acc, partials);
}

// cov[c*3 + r] = sum m1 (q - c2)_r (p - c1)_c            (rot_transform :613-643; layout of rotation_from_cov)
struct Centres {
    double c1[3], c2[3];
};
__global__ void __launch_bounds__(RB) k64_cov(SelD s1, SelD s2, Centres C, double *partials) {
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t k = blockIdx.x * RB + threadIdx.x; k < s1.n; k += gridDim.x * RB) {
        const uint64_t a1 = atom_of(s1, k), a2 = atom_of(s2, k);
        const double *p = s1.xyz + 3 * a1, *q = s2.xyz + 3 * a2;
        const double m = s1.mass[a1];
        const double pc[3] = {p[0] - C.c1[0], p[1] - C.c1[1], p[2] - C.c1[2]};
        const double qc[3] = {q[0] - C.c2[0], q[1] - C.c2[1], q[2] - C.c2[2]};
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int r = 0; r < 3; ++r) acc[c * 3 + r] += (qc[r] * pc[c]) * m;
    }
    block_store<9>(acc, partials);
}

// inertia tensor about c (:577-590): [0..2] = T00, T11, T22, [3..5] = T01, T02, T12 (negated products)
__global__ void __launch_bounds__(RB) k64_inertia(SelD s, double cx, double cy, double cz, double *partials) {
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (uint32_t k = blockIdx.x * RB + threadIdx.x; k < s.n; k += gridDim.x * RB) {
        const uint64_t a = atom_of(s, k);
        const double *p = s.xyz + 3 * a;
        const double x = p[0] - cx, y = p[1] - cy, z = p[2] - cz;
        const double m = s.mass[a];
        acc[0] += m * (y * y + z * z);
        acc[1] += m * (x * x + z * z);
        acc[2] += m * (x * x + y * y);
        acc[3] -= m * x * y;
        acc[4] -= m * x * z;
        acc[5] -= m * y * z;
    }
    block_store<6>(acc, partials);
}

// min_max (:22-36): per-workgroup lower / upper corners, folded by the host (min / max are exact in any order)
__global__ void __launch_bounds__(RB) k64_minmax(SelD s, double *partials) {
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t k = blockIdx.x * RB + threadIdx.x; k < s.n; k += gridDim.x * RB) {
        const double *p = s.xyz + 3 * atom_of(s, k);
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            lo[d] = p[d] < lo[d] ? p[d] : lo[d];
            hi[d] = p[d] > hi[d] ? p[d] : hi[d];
        }
    }
    __shared__ double sh[RB / 64][6];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        for (int off = 32; off > 0; off >>= 1) {
            const double a = __shfl_xor(lo[d], off, 64), b = __shfl_xor(hi[d], off, 64);
            lo[d] = a < lo[d] ? a : lo[d];
            hi[d] = b > hi[d] ? b : hi[d];
        }
        if (lane == 0) {
            sh[wave][d] = lo[d];
            sh[wave][3 + d] = hi[d];
        }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        double v = sh[0][threadIdx.x];
        for (int w = 1; w < RB / 64; ++w) {
            const double o = sh[w][threadIdx.x];
            v = threadIdx.x < 3 ? (o < v ? o : v) : (o > v ? o : v);
        }
        partials[(size_t)blockIdx.x * 6 + threadIdx.x] = v;
    }
}

// translate (modify.rs:16-23)
__global__ void __launch_bounds__(RB) k64_translate(SelD s, double *xyz_rw, double sx, double sy, double sz) {
    for (uint32_t k = blockIdx.x * RB + threadIdx.x; k < s.n; k += gridDim.x * RB) {
        double *p = xyz_rw + 3 * atom_of(s, k);
        p[0] += sx; p[1] += sy; p[2] += sz;
    }
}

// images relative to the first selected atom (center_of_*_pbc_dims :156-168, :197-220):
// [0] = sum m (k >= 1), [1..3] = sum img*m (k >= 1), [4..6] = sum img (k >= 1)
__global__ void __launch_bounds__(RB) k64_sums_pbc(SelD s, const BoxD *box, uint32_t pbc, double *partials) {
    double acc[7] = {0, 0, 0, 0, 0, 0, 0};
    const BoxD &B = *box;
    const double *q0 = s.xyz + 3 * atom_of(s, 0);
    const D3 p0 = D3{q0[0], q0[1], q0[2]};
    for (uint32_t k = 1 + blockIdx.x * RB + threadIdx.x; k < s.n; k += gridDim.x * RB) {
        const uint64_t a = atom_of(s, k);
        const double *q = s.xyz + 3 * a;
        const D3 im = closest_image(B, D3{q[0], q[1], q[2]}, p0, pbc);
        const double m = s.mass ? s.mass[a] : 1.0;
        acc[0] += m;
        acc[1] += im.x * m; acc[2] += im.y * m; acc[3] += im.z * m;
        acc[4] += im.x; acc[5] += im.y; acc[6] += im.z;
    }
    block_store<7>(acc, partials);
}

// gyration_pbc (:222-232): [0] = sum |shortest_vector(p - c)|^2 * m, [1] = sum m
__global__ void __launch_bounds__(RB) k64_central_pbc(SelD s, const BoxD *box, double cx, double cy, double cz, double *partials) {
    double acc[2] = {0, 0};
    const BoxD &B = *box;
    for (uint32_t k = blockIdx.x * RB + threadIdx.x; k < s.n; k += gridDim.x * RB) {
        const uint64_t a = atom_of(s, k);
        const double *p = s.xyz + 3 * a;
        const D3 d = shortest_vector(B, D3{p[0] - cx, p[1] - cy, p[2] - cz}, MOLAR_HIP_PBC_FULL);
        const double m = s.mass[a];
        acc[0] += norm2(d) * m;
        acc[1] += m;
    }
    block_store<2>(acc, partials);
}

// inertia_pbc (:234-244): the tensor of k64_inertia over d = shortest_vector(p - c)
__global__ void __launch_bounds__(RB) k64_inertia_pbc(SelD s, const BoxD *box, double cx, double cy, double cz, double *partials) {
    double acc[6] = {0, 0, 0, 0, 0, 0};
    const BoxD &B = *box;
    for (uint32_t k = blockIdx.x * RB + threadIdx.x; k < s.n; k += gridDim.x * RB) {
        const uint64_t a = atom_of(s, k);
        const double *p = s.xyz + 3 * a;
        const D3 d = shortest_vector(B, D3{p[0] - cx, p[1] - cy, p[2] - cz}, MOLAR_HIP_PBC_FULL);
        const double m = s.mass[a];
        acc[0] += m * (d.y * d.y + d.z * d.z);
        acc[1] += m * (d.x * d.x + d.z * d.z);
        acc[2] += m * (d.x * d.x + d.y * d.y);
        acc[3] -= m * d.x * d.y;
        acc[4] -= m * d.x * d.z;
        acc[5] -= m * d.y * d.z;
    }
    block_store<6>(acc, partials);
}

// unwrap_simple_dim (modify.rs:40-54): every atom becomes its image closest to the first one
__global__ void __launch_bounds__(RB) k64_unwrap(SelD s, double *xyz_rw, const BoxD *box, uint32_t pbc) {
    const BoxD &B = *box;
    const double *q0 = s.xyz + 3 * atom_of(s, 0);
    const D3 p0 = D3{q0[0], q0[1], q0[2]};
    for (uint32_t k = 1 + blockIdx.x * RB + threadIdx.x; k < s.n; k += gridDim.x * RB) {
        const uint64_t a = atom_of(s, k);
        const double *q = s.xyz + 3 * a;
        const D3 im = closest_image(B, D3{q[0], q[1], q[2]}, p0, pbc);
        double *w = xyz_rw + 3 * a;
        w[0] = im.x; w[1] = im.y; w[2] = im.z;
    }
}

// p <- R p + t (modify.rs:32-36), R column-major
struct Iso {
    double R[9], t[3];
};
__global__ void __launch_bounds__(RB) k64_apply(SelD s, double *xyz_rw, Iso T) {
    for (uint32_t k = blockIdx.x * RB + threadIdx.x; k < s.n; k += gridDim.x * RB) {
        double *p = xyz_rw + 3 * atom_of(s, k);
        const double x = p[0], y = p[1], z = p[2];
#pragma unroll
        for (int r = 0; r < 3; ++r) p[r] = ((T.R[r] * x + T.R[3 + r] * y) + T.R[6 + r] * z) + T.t[r];
    }
}

// ---- the per-frame loop (benches/comparison_small.rs:14-25) over a batch of frames: fit every frame's selection onto the
// reference selection, RMSD / centre of mass / gyration of the fitted selection, optionally move the frame.  Three
// gather passes per frame (centre; centred covariance; residual of the fitted positions), each frame's totals and its
// rotation derived on the device, so a batch is six launches whatever its length.
struct BatchD {
    const double *frames;      // [F][natoms][3]
    size_t stride;             // doubles between frames
    const uint64_t *idx;       // selection of the frames (NULL = all atoms)
    const double *mass;        // full-length column
    const double *ref;         // reference coordinates
    const uint64_t *ref_idx;   // its selection (NULL = all atoms)
    uint32_t n;
};
constexpr int REC64 = 20;      // per frame: R[9] (column-major), t[3], rmsd, com[3], gyration, status, c1 scratch is separate

template <int NV>
__device__ __forceinline__ void block_store_y(double *acc, double *partials) {
    block_store<NV>(acc, partials + (size_t)blockIdx.y * gridDim.x * NV);
}

// sum over the partial rows of frame f, result valid in every lane
template <int NV>
__device__ __forceinline__ void frame_total(const double *partials, uint32_t nblk, uint32_t f, double (&S)[NV]) {
#pragma unroll
    for (int v = 0; v < NV; ++v) S[v] = 0.0;
    for (uint32_t b = threadIdx.x; b < nblk; b += 64)
#pragma unroll
        for (int v = 0; v < NV; ++v) S[v] += partials[((size_t)f * nblk + b) * NV + v];
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int v = 0; v < NV; ++v) S[v] += __shfl_xor(S[v], off, 64);
}

__global__ void __launch_bounds__(RB) k64b_sums(BatchD B, double *partials) {
    double acc[4] = {0, 0, 0, 0};
    const double *fr = B.frames + (size_t)blockIdx.y * B.stride;
    for (uint32_t k = blockIdx.x * RB + threadIdx.x; k < B.n; k += gridDim.x * RB) {
        const uint64_t a = B.idx ? B.idx[k] : (uint64_t)k;
        const double *p = fr + 3 * a;
        const double m = B.mass[a];
        acc[0] += m;
#pragma unroll
        for (int d = 0; d < 3; ++d) acc[1 + d] += p[d] * m;
    }
    block_store_y<4>(acc, partials);
}

// centres[f] = {c1.x, c1.y, c1.z, sum m}; status in rec
__global__ void __launch_bounds__(64) k64b_centres(const double *partials, uint32_t nblk, double *centres, double *rec) {
    double S[4];
    frame_total<4>(partials, nblk, blockIdx.x, S);
    if (threadIdx.x != 0) return;
    double *c = centres + 4 * (size_t)blockIdx.x;
    rec[REC64 * (size_t)blockIdx.x + 17] = S[0] == 0.0 ? (double)MOLAR_HIP_ERR_ZERO_MASS : 0.0;
    c[3] = S[0];
#pragma unroll
    for (int d = 0; d < 3; ++d) c[d] = S[0] == 0.0 ? 0.0 : S[1 + d] / S[0];
}

// [0..8] = cov[c*3+r] = sum m (q - c2)_r (p - c1)_c, [9] = sum m |p - c1|^2
__global__ void __launch_bounds__(RB) k64b_cov(BatchD B, const double *centres, Centres C2, double *partials) {
    double acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const double *fr = B.frames + (size_t)blockIdx.y * B.stride;
    const double *c1 = centres + 4 * (size_t)blockIdx.y;
    const double c1x = c1[0], c1y = c1[1], c1z = c1[2];
    for (uint32_t k = blockIdx.x * RB + threadIdx.x; k < B.n; k += gridDim.x * RB) {
        const uint64_t a1 = B.idx ? B.idx[k] : (uint64_t)k, a2 = B.ref_idx ? B.ref_idx[k] : (uint64_t)k;
        const double *p = fr + 3 * a1, *q = B.ref + 3 * a2;
        const double m = B.mass[a1];
        const double pc[3] = {p[0] - c1x, p[1] - c1y, p[2] - c1z};
        const double qc[3] = {q[0] - C2.c2[0], q[1] - C2.c2[1], q[2] - C2.c2[2]};
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int r = 0; r < 3; ++r) acc[c * 3 + r] += (qc[r] * pc[c]) * m;
        acc[9] += ((pc[0] * pc[0] + pc[1] * pc[1]) + pc[2] * pc[2]) * m;
    }
    block_store_y<10>(acc, partials);
}

// rotation, translation, centre of mass and gyration radius of the fitted selection of frame blockIdx.x
// A frame that fails (zero mass, non-finite covariance) raises *any_failed: no frame of the call is moved then (the
// f32 entry behaves the same way), and its record is zero-filled rather than left uninitialised.
__global__ void __launch_bounds__(64) k64b_rot(const double *partials, uint32_t nblk, const double *centres, Centres C2,
                                               double *rec, unsigned int *any_failed) {
    double S[10];
    frame_total<10>(partials, nblk, blockIdx.x, S);
    if (threadIdx.x != 0) return;
    double *o = rec + REC64 * (size_t)blockIdx.x;
    const double *c1 = centres + 4 * (size_t)blockIdx.x;
    double R[9];
    if (o[17] == 0.0 && !rotation_from_cov(S, R, true)) o[17] = (double)MOLAR_HIP_ERR_SVD;
    if (o[17] != 0.0) {
        for (int i = 0; i < 17; ++i) o[i] = 0.0;
        atomicExch(any_failed, 1u);
        return;
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) o[i] = R[i];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        o[9 + r] = C2.c2[r] + (((R[r] * -c1[0]) + (R[3 + r] * -c1[1])) + (R[6 + r] * -c1[2]));     // (:521)
        o[13 + r] = (((R[r] * c1[0]) + (R[3 + r] * c1[1])) + (R[6 + r] * c1[2])) + o[9 + r];       // R cm + t
    }
    o[16] = sqrt(S[9] / c1[3]);                                                                   // rigid motion keeps it
}

// sum |R p + t - q|^2 of the fitted positions; with `apply` the frame's selection is moved (modify.rs:32-36)
__global__ void __launch_bounds__(RB) k64b_resid(BatchD B, const double *rec, int apply, const unsigned int *any_failed, double *frames_rw,
                                                 double *partials) {
    if (*any_failed) apply = 0;
    double acc[1] = {0};
    const double *o = rec + REC64 * (size_t)blockIdx.y;
    double *fr = frames_rw + (size_t)blockIdx.y * B.stride;
    if (o[17] == 0.0) {
        for (uint32_t k = blockIdx.x * RB + threadIdx.x; k < B.n; k += gridDim.x * RB) {
            const uint64_t a1 = B.idx ? B.idx[k] : (uint64_t)k, a2 = B.ref_idx ? B.ref_idx[k] : (uint64_t)k;
            double *p = fr + 3 * a1;
            const double *q = B.ref + 3 * a2;
            const double x = p[0], y = p[1], z = p[2];
            double w[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) w[r] = ((o[r] * x + o[3 + r] * y) + o[6 + r] * z) + o[9 + r];
            const double dx = q[0] - w[0], dy = q[1] - w[1], dz = q[2] - w[2];
            acc[0] += (dx * dx + dy * dy) + dz * dz;
            if (apply) {
                p[0] = w[0]; p[1] = w[1]; p[2] = w[2];
            }
        }
    }
    block_store_y<1>(acc, partials);
}

__global__ void __launch_bounds__(64) k64b_rmsd(const double *partials, uint32_t nblk, uint32_t n, double *rec, double *rec_host) {
    double S[1];
    frame_total<1>(partials, nblk, blockIdx.x, S);
    if (threadIdx.x != 0) return;
    double *o = rec + REC64 * (size_t)blockIdx.x;
    o[12] = sqrt(S[0] / (double)n);
#pragma unroll
    for (int i = 0; i < REC64; ++i) rec_host[REC64 * (size_t)blockIdx.x + i] = o[i];
}

// Measure::lipid_tail_order (measure.rs:270-422) in f64: the f32 kernel of measure.hip with every operation in double
// (one thread per tail, bonds walked sequentially to keep the reference's last-writer-wins behaviour).
struct E3 {
    double x, y, z;
};
__device__ __forceinline__ E3 e3sub(E3 a, E3 b) { return E3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ double e3dot(E3 a, E3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ double e3norm(E3 a) { return sqrt(e3dot(a, a)); }
__device__ __forceinline__ E3 e3cross(E3 a, E3 b) {
    return E3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ E3 e3unit(E3 a) {
    const double n = e3norm(a);
    return E3{a.x / n, a.y / n, a.z / n};
}
// nalgebra Vector::angle: acos(clamp(a.b / (|a||b|), -1, 1)), 0 if either vector is zero
__device__ __forceinline__ double e3angle(E3 a, E3 b) {
    const double n1 = e3norm(a), n2 = e3norm(b);
    if (n1 == 0.0 || n2 == 0.0) return 0.0;
    double c = e3dot(a, b) / (n1 * n2);
    c = c < -1.0 ? -1.0 : (c > 1.0 ? 1.0 : c);
    return acos(c);
}

__global__ void __launch_bounds__(64) k64_lipid_order(const double *__restrict__ xyz, const uint64_t *__restrict__ idx,
                                                    const uint64_t *__restrict__ toff, uint32_t ntails, int order_type,
                                                    const double *__restrict__ normals, const uint64_t *__restrict__ noff,
                                                    const uint8_t *__restrict__ bonds, double *__restrict__ out,
                                                    int *__restrict__ status) {
    const uint32_t t = blockIdx.x * 64u + threadIdx.x;
    if (t >= ntails) return;
    const uint64_t a0 = toff[t];
    const uint32_t n = (uint32_t)(toff[t + 1] - a0);
    const uint32_t nn = (uint32_t)(noff[t + 1] - noff[t]);
    if (n < 3) {
        atomicMax(status, MOLAR_HIP_ERR_LIPID_TAIL_TOO_SHORT);
        return;
    }
    if (nn != 1 && nn != n - 2) {
        atomicMax(status, MOLAR_HIP_ERR_LIPID_NORMALS_COUNT);
        return;
    }
    const uint8_t *bo = bonds + (a0 - t);
    double *order = out + (a0 - 2ull * t);
    auto P = [&](uint32_t k) {
        const double *q = xyz + 3 * idx[a0 + k];
        return E3{q[0], q[1], q[2]};
    };
    auto N = [&](uint32_t k) {
        const double *q = normals + 3 * (noff[t] + (nn == 1 ? 0 : k));
        return E3{q[0], q[1], q[2]};
    };
    for (uint32_t k = 0; k < n - 2; ++k) order[k] = 0.0;
    if (order_type == 0) {
        for (uint32_t at = 1; at + 1 < n; ++at) {
            const double c = cos(e3angle(e3sub(P(at + 1), P(at - 1)), N(at - 1)));
            order[at - 1] = 1.5 * (c * c) - 0.5;
        }
        return;
    }
    const double sqrt3 = sqrt(3.0), pi = 3.14159265358979323846;
    for (uint32_t i = 0; i + 2 < n; ++i) {
        if (bo[i] == 1) {
            if (bo[i + 1] == 1) {
                const E3 p1 = P(i), p2 = P(i + 1), p3 = P(i + 2);
                const E3 lz = e3unit(e3sub(p3, p1));
                const E3 lx = e3unit(e3cross(e3sub(p1, p2), e3sub(p3, p2)));
                const E3 ly = e3cross(lx, lz);
                const E3 nv = N(i);
                const double cx = cos(e3angle(lx, nv)), cy = cos(e3angle(ly, nv));
                const double sxx = 0.5 * (3.0 * (cx * cx) - 1.0), syy = 0.5 * (3.0 * (cy * cy) - 1.0);
                order[i] = -(2.0 * sxx + syy) / 3.0;
            }
        } else {
            // a double bond at bond 0 has no C(i-1); at the last bond it has no normal for atom i+1 when normals are
            // per bond.  The reference indexes out of range there (usize underflow panic / unchecked read,
            // measure.rs:361-364,385): refuse the tail instead of touching memory outside it.
            if (i == 0 || (nn != 1 && i + 1 >= nn)) {
                atomicMax(status, MOLAR_HIP_ERR_INVALID_ARGUMENT);
                return;
            }
            const E3 p1 = P(i - 1), p2 = P(i), p3 = P(i + 1), p4 = P(i + 2);
            const double a1 = 0.5 * (pi - e3angle(e3sub(p1, p2), e3sub(p3, p2)));
            const double a2 = 0.5 * (pi - e3angle(e3sub(p2, p3), e3sub(p4, p3)));
            const E3 lz = e3unit(e3sub(p3, p2));
            for (int side = 0; side < 2; ++side) {
                const E3 lx = e3unit(e3cross(side == 0 ? e3sub(p1, p2) : e3sub(p3, p4), lz));
                const E3 ly = e3cross(lx, lz);
                const E3 nv = N(side == 0 ? i : i + 1);
                const double cy = cos(e3angle(ly, nv)), cz = cos(e3angle(lz, nv));
                const double szz = 0.5 * (3.0 * (cz * cz) - 1.0), syy = 0.5 * (3.0 * (cy * cy) - 1.0);
                const double syz = 1.5 * cy * cz;
                const double a = side == 0 ? a1 : a2, sgn = side == 0 ? -1.0 : 1.0;
                double v;
                if (order_type == 2) {
                    const double ca = cos(a), sa = sin(a);
                    v = -(((ca * ca) * syy + (sa * sa) * szz) + sgn * (2.0 * ca * sa * syz));
                } else {
                    v = -((szz / 4.0 + 3.0 * syy / 4.0) + sgn * (sqrt3 * syz / 2.0));
                }
                order[side == 0 ? i - 1 : i] = v;
            }
        }
    }
}

// totals of the per-workgroup partials in a fixed order, written to pinned host memory
template <int NV>
__global__ void __launch_bounds__(64) k64_total(const double *partials, uint32_t nblk, double *out_host) {
    double S[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) S[v] = 0.0;
    for (uint32_t b = threadIdx.x; b < nblk; b += 64)
#pragma unroll
        for (int v = 0; v < NV; ++v) S[v] += partials[(size_t)b * NV + v];
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int v = 0; v < NV; ++v) S[v] += __shfl_xor(S[v], off, 64);
    if (threadIdx.x == 0)
#pragma unroll
        for (int v = 0; v < NV; ++v) out_host[v] = S[v];
}

uint32_t blocks64(const molar_hip_ctx *c, uint32_t n) {
    uint32_t nb = (n + RB * 4 - 1) / (RB * 4);
    if (nb < 1) nb = 1;
    const uint32_t cap = (uint32_t)c->num_cus * 4u;
    return nb > cap ? cap : nb;
}

int stage64(molar_hip_ctx *c, const double *xyz, size_t natoms, const uint64_t *idx, size_t n, const double *mass,
            DevBuf &bx, DevBuf &bi, DevBuf &bm, SelD *out) {
    if (!xyz) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "measure (f64): xyz pointer is null");
    const size_t nsel = idx ? n : natoms;
    if (nsel >= 0xFFFFFFFFull) return fail(MOLAR_HIP_ERR_TOO_LARGE, "measure (f64): selection too large");
    out->n = (uint32_t)nsel;
    MH_TRY(to_device(c, xyz, natoms * 3, bx, &out->xyz));
    MH_TRY(to_device(c, idx, idx ? n : 0, bi, &out->idx));
    out->mass = nullptr;
    if (mass) MH_TRY(to_device(c, mass, natoms, bm, &out->mass));
    return 0;
}

template <int NV, class Launch>
int reduce64(molar_hip_ctx *c, uint32_t n, double *host_out, Launch launch) {
    const uint32_t nb = blocks64(c, n);
    MH_TRY(c->m_partials.reserve((size_t)nb * NV * 8));
    MH_TRY(ensure_pinned(c, 64 * 8));
    launch(nb, c->m_partials.as<double>());
    hipLaunchKernelGGL(k64_total<NV>, dim3(1), dim3(64), 0, c->stream, c->m_partials.as<double>(), nb,
                       static_cast<double *>(c->h_pinned));
    MH_HIP(hipGetLastError());
    MH_HIP(hipStreamSynchronize(c->stream));
    std::memcpy(host_out, c->h_pinned, (size_t)NV * 8);
    return 0;
}

int com64(molar_hip_ctx *c, const SelD &s, bool weighted, double out[3]) {
    double r[7];
    MH_TRY((reduce64<7>(c, s.n, r, [&](uint32_t nb, double *part) {
        hipLaunchKernelGGL(k64_sums, dim3(nb), dim3(RB), 0, c->stream, s, part);
    })));
    if (weighted) {
        if (r[0] == 0.0) return fail(MOLAR_HIP_ERR_ZERO_MASS, "zero mass");
        for (int d = 0; d < 3; ++d) out[d] = r[1 + d] / r[0];
    } else {
        for (int d = 0; d < 3; ++d) out[d] = r[4 + d] / (double)s.n;      // 0/0 = NaN for an empty selection, as the reference
    }
    return 0;
}

int rmsd64(molar_hip_ctx *c, const double *xyz1, size_t natoms1, const uint64_t *idx1, size_t n1, const double *mass1,
           const double *xyz2, size_t natoms2, const uint64_t *idx2, size_t n2, bool weighted, double *out) {
    const size_t s1n = idx1 ? n1 : natoms1, s2n = idx2 ? n2 : natoms2;
    if (s1n != s2n) return fail(MOLAR_HIP_ERR_SIZES, "incompatible sizes: %zu and %zu", s1n, s2n);
    SelD s1, s2;
    MH_TRY(stage64(c, xyz1, natoms1, idx1, n1, mass1, c->m_xyz1, c->m_idx1, c->m_mass1, &s1));
    MH_TRY(stage64(c, xyz2, natoms2, idx2, n2, nullptr, c->m_xyz2, c->m_idx2, c->m_mass2, &s2));
    double r[3];
    MH_TRY((reduce64<3>(c, s1.n, r, [&](uint32_t nb, double *part) {
        hipLaunchKernelGGL(k64_rmsd, dim3(nb), dim3(RB), 0, c->stream, s1, s2, part);
    })));
    if (weighted) {
        if (r[2] == 0.0) return fail(MOLAR_HIP_ERR_ZERO_MASS, "zero mass");
        *out = std::sqrt(r[1] / r[2]);
    } else {
        *out = std::sqrt(r[0] / (double)s1.n);
    }
    return MOLAR_HIP_OK;
}

}  // namespace

#define MH64_CTX(c)                                                                  \
    do {                                                                             \
        if (!(c)) return ::mh::fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "null context"); \
        MH_HIP(hipSetDevice((c)->device));                                           \
    } while (0)

extern "C" {

int molar_hip_center_of_geometry_f64(molar_hip_ctx *c, const double *xyz, size_t natoms, const uint64_t *idx, size_t n,
                                     double out[3]) {
    MH64_CTX(c);
    SelD s;
    MH_TRY(stage64(c, xyz, natoms, idx, n, nullptr, c->m_xyz1, c->m_idx1, c->m_mass1, &s));
    return com64(c, s, false, out);
}

int molar_hip_center_of_mass_f64(molar_hip_ctx *c, const double *xyz, size_t natoms, const uint64_t *idx, size_t n,
                                 const double *mass, double out[3]) {
    MH64_CTX(c);
    if (!mass) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "center_of_mass_f64: mass pointer is null");
    SelD s;
    MH_TRY(stage64(c, xyz, natoms, idx, n, mass, c->m_xyz1, c->m_idx1, c->m_mass1, &s));
    return com64(c, s, true, out);
}

int molar_hip_gyration_f64(molar_hip_ctx *c, const double *xyz, size_t natoms, const uint64_t *idx, size_t n,
                           const double *mass, double *out) {
    MH64_CTX(c);
    if (!mass) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "gyration_f64: mass pointer is null");
    SelD s;
    MH_TRY(stage64(c, xyz, natoms, idx, n, mass, c->m_xyz1, c->m_idx1, c->m_mass1, &s));
    double cm[3], r[2];
    MH_TRY(com64(c, s, true, cm));                                        // center_of_mass (:82)
    MH_TRY((reduce64<2>(c, s.n, r, [&](uint32_t nb, double *part) {
        hipLaunchKernelGGL(k64_central, dim3(nb), dim3(RB), 0, c->stream, s, cm[0], cm[1], cm[2], part);
    })));
    *out = std::sqrt(r[0] / r[1]);
    return MOLAR_HIP_OK;
}

int molar_hip_rmsd_f64(molar_hip_ctx *c, const double *xyz1, size_t natoms1, const uint64_t *idx1, size_t n1,
                       const double *xyz2, size_t natoms2, const uint64_t *idx2, size_t n2, double *out) {
    MH64_CTX(c);
    return rmsd64(c, xyz1, natoms1, idx1, n1, nullptr, xyz2, natoms2, idx2, n2, false, out);
}

int molar_hip_rmsd_mw_f64(molar_hip_ctx *c, const double *xyz1, size_t natoms1, const uint64_t *idx1, size_t n1,
                          const double *mass1, const double *xyz2, size_t natoms2, const uint64_t *idx2, size_t n2,
                          double *out) {
    MH64_CTX(c);
    if (!mass1) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "rmsd_mw_f64: mass pointer is null");
    return rmsd64(c, xyz1, natoms1, idx1, n1, mass1, xyz2, natoms2, idx2, n2, true, out);
}

int molar_hip_fit_transform_f64(molar_hip_ctx *c, const double *xyz1, size_t natoms1, const uint64_t *idx1, size_t n1,
                                const double *mass1, const double *xyz2, size_t natoms2, const uint64_t *idx2, size_t n2,
                                const double *mass2, int at_origin, double R9[9], double t3[3]) {
    MH64_CTX(c);
    if (!mass1 || (!at_origin && !mass2)) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "fit_transform_f64: mass pointer is null");
    SelD s1, s2;
    MH_TRY(stage64(c, xyz1, natoms1, idx1, n1, mass1, c->m_xyz1, c->m_idx1, c->m_mass1, &s1));
    MH_TRY(stage64(c, xyz2, natoms2, idx2, n2, at_origin ? nullptr : mass2, c->m_xyz2, c->m_idx2, c->m_mass2, &s2));
    Centres C{};
    if (!at_origin) {
        MH_TRY(com64(c, s1, true, C.c1));        // cm1 (:511)
        MH_TRY(com64(c, s2, true, C.c2));        // cm2 with sel2's own masses (:512)
    }
    SelD a = s1, b = s2;
    a.n = b.n = s1.n < s2.n ? s1.n : s2.n;       // izip! stops at the shorter selection (:621)
    double cov[9];
    MH_TRY((reduce64<9>(c, a.n, cov, [&](uint32_t nb, double *part) {
        hipLaunchKernelGGL(k64_cov, dim3(nb), dim3(RB), 0, c->stream, a, b, C, part);
    })));
    double R[9];
    if (!rotation_from_cov(cov, R, /*precise=*/true)) return fail(MOLAR_HIP_ERR_SVD, "SVD failed");
    std::memcpy(R9, R, sizeof R);
    for (int r = 0; r < 3; ++r)                  // Translation(cm2) * rot * Translation(-cm1) (:521)
        t3[r] = at_origin ? 0.0 : C.c2[r] + (((R[r] * -C.c1[0]) + (R[3 + r] * -C.c1[1])) + (R[6 + r] * -C.c1[2]));
    return MOLAR_HIP_OK;
}

int molar_hip_min_max_f64(molar_hip_ctx *c, const double *xyz, size_t natoms, const uint64_t *idx, size_t n, double lower[3],
                          double upper[3]) {
    MH64_CTX(c);
    SelD s;
    MH_TRY(stage64(c, xyz, natoms, idx, n, nullptr, c->m_xyz1, c->m_idx1, c->m_mass1, &s));
    const uint32_t nb = blocks64(c, s.n);
    MH_TRY(c->m_partials.reserve((size_t)nb * 6 * 8));
    MH_TRY(ensure_pinned(c, (size_t)nb * 6 * 8));
    hipLaunchKernelGGL(k64_minmax, dim3(nb), dim3(RB), 0, c->stream, s, c->m_partials.as<double>());
    MH_HIP(hipGetLastError());
    MH_HIP(hipMemcpyAsync(c->h_pinned, c->m_partials.p, (size_t)nb * 6 * 8, hipMemcpyDeviceToHost, c->stream));
    MH_HIP(hipStreamSynchronize(c->stream));
    const double *part = static_cast<const double *>(c->h_pinned);
    for (int d = 0; d < 3; ++d) {
        // the reference starts from +-Float::MAX (:23-24); an empty selection returns those
        double lo = 1.7976931348623157e308, hi = -1.7976931348623157e308;
        for (uint32_t b = 0; b < nb; ++b) {
            lo = part[b * 6 + d] < lo ? part[b * 6 + d] : lo;
            hi = part[b * 6 + 3 + d] > hi ? part[b * 6 + 3 + d] : hi;
        }
        lower[d] = lo;
        upper[d] = hi;
    }
    return MOLAR_HIP_OK;
}

static void inertia64_finish(const double r[6], double moments[3], double axes9[9], double tensor9[9]) {
    const double T[9] = {r[0], r[3], r[4], r[3], r[1], r[5], r[4], r[5], r[2]};
    if (tensor9) std::memcpy(tensor9, T, sizeof T);
    double A[9], w[3], V[9];
    std::memcpy(A, T, sizeof A);
    jacobi_sym<3>(A, w, V, 1e-34);
    int ord[3] = {0, 1, 2};   // ascending moments (:594-601)
    for (int a = 0; a < 2; ++a)
        for (int q = a + 1; q < 3; ++q)
            if (w[ord[q]] < w[ord[a]]) std::swap(ord[a], ord[q]);
    for (int k = 0; k < 3; ++k) moments[k] = w[ord[k]];
    // col0, col1 normalised, col2 = col0 x col1 (:603-607)
    double e[2][3];
    for (int k = 0; k < 2; ++k) {
        const double v[3] = {V[0 * 3 + ord[k]], V[1 * 3 + ord[k]], V[2 * 3 + ord[k]]};
        const double nn = std::sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
        for (int d = 0; d < 3; ++d) e[k][d] = v[d] / nn;
    }
    const double c2[3] = {e[0][1] * e[1][2] - e[0][2] * e[1][1], e[0][2] * e[1][0] - e[0][0] * e[1][2],
                          e[0][0] * e[1][1] - e[0][1] * e[1][0]};
    for (int d = 0; d < 3; ++d) {
        axes9[0 * 3 + d] = e[0][d];
        axes9[1 * 3 + d] = e[1][d];
        axes9[2 * 3 + d] = c2[d];
    }
}

int molar_hip_inertia_f64(molar_hip_ctx *c, const double *xyz, size_t natoms, const uint64_t *idx, size_t n,
                          const double *mass, double moments[3], double axes9[9], double tensor9[9]) {
    MH64_CTX(c);
    if (!mass) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "inertia_f64: mass pointer is null");
    SelD s;
    MH_TRY(stage64(c, xyz, natoms, idx, n, mass, c->m_xyz1, c->m_idx1, c->m_mass1, &s));
    double cm[3], r[6];
    MH_TRY(com64(c, s, true, cm));                                        // center_of_mass (:94)
    MH_TRY((reduce64<6>(c, s.n, r, [&](uint32_t nb, double *part) {
        hipLaunchKernelGGL(k64_inertia, dim3(nb), dim3(RB), 0, c->stream, s, cm[0], cm[1], cm[2], part);
    })));
    inertia64_finish(r, moments, axes9, tensor9);
    return MOLAR_HIP_OK;
}

int molar_hip_translate_f64(molar_hip_ctx *c, double *xyz, size_t natoms, const uint64_t *idx, size_t n, const double shift3[3]) {
    MH64_CTX(c);
    if (!shift3) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "translate_f64: null argument");
    SelD s;
    MH_TRY(stage64(c, xyz, natoms, idx, n, nullptr, c->m_xyz1, c->m_idx1, c->m_mass1, &s));
    if (s.n == 0) return MOLAR_HIP_OK;
    hipLaunchKernelGGL(k64_translate, dim3(blocks64(c, s.n)), dim3(RB), 0, c->stream, s, const_cast<double *>(s.xyz), shift3[0],
                       shift3[1], shift3[2]);
    MH_HIP(hipGetLastError());
    if (!is_device_ptr(xyz)) MH_HIP(hipMemcpyAsync(xyz, s.xyz, natoms * 24, hipMemcpyDeviceToHost, c->stream));
    MH_HIP(hipStreamSynchronize(c->stream));
    return MOLAR_HIP_OK;
}

int molar_hip_fit_rmsd_batch_f64(molar_hip_ctx *c, double *frames, size_t nframes, size_t natoms, const uint64_t *idx, size_t n,
                                 const double *mass, const double *ref_xyz, size_t ref_natoms, const uint64_t *ref_idx,
                                 int apply, double *rmsd_out, double *R_out, double *t_out, double *com_out, double *gyr_out) {
    MH64_CTX(c);
    if (!frames || !mass || !ref_xyz) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "fit_rmsd_batch_f64: null argument");
    if (nframes == 0) return MOLAR_HIP_OK;
    if (nframes > 65535) return fail(MOLAR_HIP_ERR_TOO_LARGE, "fit_rmsd_batch_f64: at most 65535 frames per call");
    const size_t nsel = idx ? n : natoms, nref = ref_idx ? n : ref_natoms;
    if (nsel != nref) return fail(MOLAR_HIP_ERR_SIZES, "incompatible sizes: %zu and %zu", nsel, nref);
    if (nsel == 0 || nsel >= 0xFFFFFFFFull) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "fit_rmsd_batch_f64: selection of %zu atoms", nsel);
    // the reference selection and its centre with the same mass column gathered through ITS index (:512)
    SelD ref;
    BatchD B;
    MH_TRY(stage64(c, ref_xyz, ref_natoms, ref_idx, n, nullptr, c->m_xyz2, c->m_idx2, c->m_mass2, &ref));
    MH_TRY(to_device(c, mass, natoms, c->m_mass1, &B.mass));        // one column, as long as a frame
    ref.mass = B.mass;
    Centres C2{};
    MH_TRY(com64(c, ref, true, C2.c2));
    B.n = (uint32_t)nsel;
    B.stride = natoms * 3;
    B.ref = ref.xyz;
    B.ref_idx = ref.idx;
    MH_TRY(to_device(c, (const double *)frames, natoms * 3 * nframes, c->m_xyz1, &B.frames));
    MH_TRY(to_device(c, idx, idx ? n : 0, c->m_idx1, &B.idx));
    const uint32_t F = (uint32_t)nframes;
    uint32_t nb = (B.n + RB * 8 - 1) / (RB * 8);
    nb = nb < 1 ? 1 : (nb > 64 ? 64 : nb);
    const size_t part_bytes = (size_t)nb * F * 10 * 8;
    MH_TRY(c->m_partials.reserve(part_bytes + (size_t)F * (REC64 + 4) * 8 + 64));
    MH_TRY(ensure_pinned(c, (size_t)F * REC64 * 8));
    double *part = c->m_partials.as<double>();
    double *rec = reinterpret_cast<double *>(static_cast<char *>(c->m_partials.p) + part_bytes);
    double *centres = rec + (size_t)F * REC64;
    unsigned int *any_failed = reinterpret_cast<unsigned int *>(centres + 4 * (size_t)F);     // inside the 64 spare bytes
    MH_HIP(hipMemsetAsync(any_failed, 0, 4, c->stream));
    double *frames_rw = const_cast<double *>(B.frames);
    hipLaunchKernelGGL(k64b_sums, dim3(nb, F), dim3(RB), 0, c->stream, B, part);
    hipLaunchKernelGGL(k64b_centres, dim3(F), dim3(64), 0, c->stream, part, nb, centres, rec);
    hipLaunchKernelGGL(k64b_cov, dim3(nb, F), dim3(RB), 0, c->stream, B, centres, C2, part);
    hipLaunchKernelGGL(k64b_rot, dim3(F), dim3(64), 0, c->stream, part, nb, centres, C2, rec, any_failed);
    hipLaunchKernelGGL(k64b_resid, dim3(nb, F), dim3(RB), 0, c->stream, B, rec, apply ? 1 : 0, any_failed, frames_rw, part);
    hipLaunchKernelGGL(k64b_rmsd, dim3(F), dim3(64), 0, c->stream, part, nb, B.n, rec, static_cast<double *>(c->h_pinned));
    MH_HIP(hipGetLastError());
    MH_HIP(hipStreamSynchronize(c->stream));
    const double *h = static_cast<const double *>(c->h_pinned);
    // statuses first: a failed frame leaves every frame of the call where it was (nothing was moved on the device
    // either - k64b_resid saw any_failed - and nothing is copied back), as molar_hip_fit_rmsd_batch does
    for (uint32_t f = 0; f < F; ++f) {
        const int st = (int)h[REC64 * (size_t)f + 17];
        if (st) return fail(st, st == MOLAR_HIP_ERR_ZERO_MASS ? "zero mass" : "SVD failed");
    }
    if (apply && !is_device_ptr(frames)) {
        MH_HIP(hipMemcpyAsync(frames, B.frames, nframes * natoms * 24, hipMemcpyDeviceToHost, c->stream));
        MH_HIP(hipStreamSynchronize(c->stream));
    }
    auto emit = [&](double *dst, size_t per, size_t at) -> int {
        if (!dst) return 0;
        std::vector<double> tmp((size_t)F * per);
        for (uint32_t f = 0; f < F; ++f)
            for (size_t k = 0; k < per; ++k) tmp[f * per + k] = h[REC64 * (size_t)f + at + k];
        if (is_device_ptr(dst)) {
            MH_HIP(hipMemcpyAsync(dst, tmp.data(), tmp.size() * 8, hipMemcpyHostToDevice, c->stream));
            MH_HIP(hipStreamSynchronize(c->stream));
        } else {
            std::memcpy(dst, tmp.data(), tmp.size() * 8);
        }
        return 0;
    };
    MH_TRY(emit(R_out, 9, 0));
    MH_TRY(emit(t_out, 3, 9));
    MH_TRY(emit(rmsd_out, 1, 12));
    MH_TRY(emit(com_out, 3, 13));
    MH_TRY(emit(gyr_out, 1, 16));
    return MOLAR_HIP_OK;
}

}  // extern "C"

namespace {

// the box of a call, built on the host and copied behind the partials' buffer (its own DevBuf: m_out)
int box64_to_device(molar_hip_ctx *c, const double *box9, const BoxD **d_box) {
    BoxD b;
    MH_TRY(box64_from_matrix(box9, &b));
    MH_TRY(c->m_out.reserve(sizeof(BoxD)));
    MH_HIP(hipMemcpyAsync(c->m_out.p, &b, sizeof b, hipMemcpyHostToDevice, c->stream));
    MH_HIP(hipStreamSynchronize(c->stream));          // `b` lives on this stack frame
    *d_box = c->m_out.as<BoxD>();
    return 0;
}

// center_of_mass_pbc_dims (:197-220) / center_of_geometry_pbc_dims (:156-168): the sums start at the UNWEIGHTED first
// position (and its mass), the other atoms enter as their images closest to it
int com64_pbc(molar_hip_ctx *c, const SelD &s, const BoxD *d_box, uint32_t pbc, bool weighted, double out[3]) {
    if (s.n == 0) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "periodic centre of an empty selection");
    double r[7];
    MH_TRY((reduce64<7>(c, s.n, r, [&](uint32_t nb, double *part) {
        hipLaunchKernelGGL(k64_sums_pbc, dim3(nb), dim3(RB), 0, c->stream, s, d_box, pbc, part);
    })));
    uint64_t a0 = 0;
    double p0[3], m0 = 1.0;
    if (s.idx) MH_HIP(hipMemcpy(&a0, s.idx, 8, hipMemcpyDeviceToHost));
    MH_HIP(hipMemcpy(p0, s.xyz + 3 * a0, 24, hipMemcpyDeviceToHost));
    if (s.mass) MH_HIP(hipMemcpy(&m0, s.mass + a0, 8, hipMemcpyDeviceToHost));
    if (weighted) {
        const double mass = m0 + r[0];
        if (mass == 0.0) return fail(MOLAR_HIP_ERR_ZERO_MASS, "zero mass");
        for (int d = 0; d < 3; ++d) out[d] = (p0[d] + r[1 + d]) / mass;
    } else {
        for (int d = 0; d < 3; ++d) out[d] = (p0[d] + r[4 + d]) / (double)s.n;
    }
    return 0;
}

}  // namespace

extern "C" {

int molar_hip_center_of_geometry_pbc_f64(molar_hip_ctx *c, const double *xyz, size_t natoms, const uint64_t *idx, size_t n,
                                         const double *box9, uint8_t pbc, double out[3]) {
    MH64_CTX(c);
    const BoxD *d_box;
    MH_TRY(box64_to_device(c, box9, &d_box));
    SelD s;
    MH_TRY(stage64(c, xyz, natoms, idx, n, nullptr, c->m_xyz1, c->m_idx1, c->m_mass1, &s));
    return com64_pbc(c, s, d_box, pbc & 7u, false, out);
}

int molar_hip_center_of_mass_pbc_f64(molar_hip_ctx *c, const double *xyz, size_t natoms, const uint64_t *idx, size_t n,
                                     const double *mass, const double *box9, uint8_t pbc, double out[3]) {
    MH64_CTX(c);
    if (!mass) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "center_of_mass_pbc_f64: mass pointer is null");
    const BoxD *d_box;
    MH_TRY(box64_to_device(c, box9, &d_box));
    SelD s;
    MH_TRY(stage64(c, xyz, natoms, idx, n, mass, c->m_xyz1, c->m_idx1, c->m_mass1, &s));
    return com64_pbc(c, s, d_box, pbc & 7u, true, out);
}

int molar_hip_gyration_pbc_f64(molar_hip_ctx *c, const double *xyz, size_t natoms, const uint64_t *idx, size_t n,
                               const double *mass, const double *box9, double *out) {
    MH64_CTX(c);
    if (!mass) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "gyration_pbc_f64: mass pointer is null");
    const BoxD *d_box;
    MH_TRY(box64_to_device(c, box9, &d_box));
    SelD s;
    MH_TRY(stage64(c, xyz, natoms, idx, n, mass, c->m_xyz1, c->m_idx1, c->m_mass1, &s));
    double cm[3], r[2];
    MH_TRY(com64_pbc(c, s, d_box, MOLAR_HIP_PBC_FULL, true, cm));        // center_of_mass_pbc (:227)
    MH_TRY((reduce64<2>(c, s.n, r, [&](uint32_t nb, double *part) {
        hipLaunchKernelGGL(k64_central_pbc, dim3(nb), dim3(RB), 0, c->stream, s, d_box, cm[0], cm[1], cm[2], part);
    })));
    *out = std::sqrt(r[0] / r[1]);
    return MOLAR_HIP_OK;
}

int molar_hip_unwrap_simple_f64(molar_hip_ctx *c, double *xyz, size_t natoms, const uint64_t *idx, size_t n, const double *box9,
                                uint8_t pbc) {
    MH64_CTX(c);
    if (idx ? n == 0 : natoms == 0) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "unwrap_simple_f64 of an empty selection");
    const BoxD *d_box;
    MH_TRY(box64_to_device(c, box9, &d_box));
    SelD s;
    MH_TRY(stage64(c, xyz, natoms, idx, n, nullptr, c->m_xyz1, c->m_idx1, c->m_mass1, &s));
    hipLaunchKernelGGL(k64_unwrap, dim3(blocks64(c, s.n)), dim3(RB), 0, c->stream, s, const_cast<double *>(s.xyz), d_box,
                       (uint32_t)(pbc & 7u));
    MH_HIP(hipGetLastError());
    if (!is_device_ptr(xyz)) MH_HIP(hipMemcpyAsync(xyz, s.xyz, natoms * 24, hipMemcpyDeviceToHost, c->stream));
    MH_HIP(hipStreamSynchronize(c->stream));
    return MOLAR_HIP_OK;
}

int molar_hip_inertia_pbc_f64(molar_hip_ctx *c, const double *xyz, size_t natoms, const uint64_t *idx, size_t n,
                              const double *mass, const double *box9, double moments[3], double axes9[9], double tensor9[9]) {
    MH64_CTX(c);
    if (!mass) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "inertia_pbc_f64: mass pointer is null");
    const BoxD *d_box;
    MH_TRY(box64_to_device(c, box9, &d_box));
    SelD s;
    MH_TRY(stage64(c, xyz, natoms, idx, n, mass, c->m_xyz1, c->m_idx1, c->m_mass1, &s));
    double cm[3], r[6];
    MH_TRY(com64_pbc(c, s, d_box, MOLAR_HIP_PBC_FULL, true, cm));        // center_of_mass_pbc (:239)
    MH_TRY((reduce64<6>(c, s.n, r, [&](uint32_t nb, double *part) {
        hipLaunchKernelGGL(k64_inertia_pbc, dim3(nb), dim3(RB), 0, c->stream, s, d_box, cm[0], cm[1], cm[2], part);
    })));
    inertia64_finish(r, moments, axes9, tensor9);
    return MOLAR_HIP_OK;
}

int molar_hip_apply_transform_f64(molar_hip_ctx *c, double *xyz, size_t natoms, const uint64_t *idx, size_t n,
                                  const double R9[9], const double t3[3]) {
    MH64_CTX(c);
    if (!R9 || !t3) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "apply_transform_f64: null argument");
    SelD s;
    MH_TRY(stage64(c, xyz, natoms, idx, n, nullptr, c->m_xyz1, c->m_idx1, c->m_mass1, &s));
    if (s.n == 0) return MOLAR_HIP_OK;
    Iso T;
    std::memcpy(T.R, R9, sizeof T.R);
    std::memcpy(T.t, t3, sizeof T.t);
    hipLaunchKernelGGL(k64_apply, dim3(blocks64(c, s.n)), dim3(RB), 0, c->stream, s, const_cast<double *>(s.xyz), T);
    MH_HIP(hipGetLastError());
    if (!is_device_ptr(xyz)) MH_HIP(hipMemcpyAsync(xyz, s.xyz, natoms * 24, hipMemcpyDeviceToHost, c->stream));
    MH_HIP(hipStreamSynchronize(c->stream));
    return MOLAR_HIP_OK;
}

int molar_hip_lipid_tail_order_f64(molar_hip_ctx *c, const double *xyz, size_t natoms, const uint64_t *idx,
                                   const uint64_t *tail_offsets, size_t ntails, int order_type, const double *normals,
                                   const uint64_t *normal_offsets, const uint8_t *bond_orders, double *out) {
    MH64_CTX(c);
    if (!xyz || !idx || !tail_offsets || !normals || !normal_offsets || !out)
        return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "lipid_tail_order_f64: null argument");
    if (order_type < 0 || order_type > 2) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "lipid_tail_order_f64: order_type %d", order_type);
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
    const double *d_xyz, *d_norm;
    const uint64_t *d_idx, *d_toff, *d_noff;
    const uint8_t *d_bo = nullptr;
    MH_TRY(to_device(c, xyz, natoms * 3, c->m_xyz1, &d_xyz));
    MH_TRY(to_device(c, idx, nidx, c->m_idx1, &d_idx));
    MH_TRY(to_device(c, normals, nnorm * 3, c->m_xyz2, &d_norm));
    MH_TRY(to_device(c, (const uint64_t *)toff.data(), ntails + 1, c->m_idx2, &d_toff));
    MH_TRY(to_device(c, (const uint64_t *)noff.data(), ntails + 1, c->m_mass2, &d_noff));
    static const uint8_t dummy = 1;
    if (bond_orders) MH_TRY(to_device(c, bond_orders, nbond, c->m_mass1, &d_bo));
    else MH_TRY(to_device(c, &dummy, (size_t)1, c->m_mass1, &d_bo));
    const bool out_dev = is_device_ptr(out);
    MH_TRY(c->m_out.reserve((nout + 4) * 8));
    double *d_out = out_dev ? out : c->m_out.as<double>();
    MH_TRY(c->m_results.reserve(64));
    int *status = c->m_results.as<int>();
    MH_HIP(hipMemsetAsync(status, 0, 4, c->stream));
    hipLaunchKernelGGL(k64_lipid_order, dim3((unsigned)((ntails + 63) / 64)), dim3(64), 0, c->stream, d_xyz, d_idx, d_toff,
                       (uint32_t)ntails, order_type, d_norm, d_noff, d_bo, d_out, status);
    MH_HIP(hipGetLastError());
    int st = 0;
    MH_HIP(hipMemcpyAsync(&st, status, 4, hipMemcpyDeviceToHost, c->stream));
    MH_HIP(hipStreamSynchronize(c->stream));
    if (st) return fail(st, "lipid order error (status %d)", st);
    if (!out_dev && nout) {
        MH_HIP(hipMemcpyAsync(out, d_out, nout * 8, hipMemcpyDeviceToHost, c->stream));
        MH_HIP(hipStreamSynchronize(c->stream));
    }
    return MOLAR_HIP_OK;
}

int molar_hip_rotate_f64(molar_hip_ctx *c, double *xyz, size_t natoms, const uint64_t *idx, size_t n, const double unit_axis3[3],
                         double angle) {
    if (!unit_axis3) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "rotate_f64: null axis");
    // nalgebra Rotation3::from_axis_angle (Rodrigues' formula on a unit axis), column-major
    const double ux = unit_axis3[0], uy = unit_axis3[1], uz = unit_axis3[2];
    const double sn = std::sin(angle), cs = std::cos(angle), k = 1.0 - cs;
    const double sqx = ux * ux, sqy = uy * uy, sqz = uz * uz;
    const double R[9] = {sqx + (1.0 - sqx) * cs, ux * uy * k + uz * sn, ux * uz * k - uy * sn,
                         ux * uy * k - uz * sn, sqy + (1.0 - sqy) * cs, uy * uz * k + ux * sn,
                         ux * uz * k + uy * sn, uy * uz * k - ux * sn, sqz + (1.0 - sqz) * cs};
    const double t[3] = {0.0, 0.0, 0.0};
    return molar_hip_apply_transform_f64(c, xyz, natoms, idx, n, R, t);     // p.coords = tr * p.coords (modify.rs:28)
}

int molar_hip_principal_transform_f64(molar_hip_ctx *c, const double *xyz, size_t natoms, const uint64_t *idx, size_t n,
                                      const double *mass, const double *box9, double R9[9], double t3[3]) {
    MH64_CTX(c);
    if (!R9 || !t3) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "principal_transform_f64: null output");
    double mom[3], axes[9], cm[3];
    if (box9) {
        MH_TRY(molar_hip_inertia_pbc_f64(c, xyz, natoms, idx, n, mass, box9, mom, axes, nullptr));
        MH_TRY(molar_hip_center_of_mass_pbc_f64(c, xyz, natoms, idx, n, mass, box9, MOLAR_HIP_PBC_FULL, cm));   // (:251)
    } else {
        MH_TRY(molar_hip_inertia_f64(c, xyz, natoms, idx, n, mass, mom, axes, nullptr));
        MH_TRY(molar_hip_center_of_mass_f64(c, xyz, natoms, idx, n, mass, cm));                                // (:106)
    }
    // do_principal_transform (:646-649): Translation(cm) * Rotation(axes^-1) * Translation(-cm);
    // try_inverse_mut leaves a singular matrix untouched (closed-form 3x3 inverse, nalgebra)
    const double *m = axes;
    const double m11 = m[0], m21 = m[1], m31 = m[2], m12 = m[3], m22 = m[4], m32 = m[5], m13 = m[6], m23 = m[7], m33 = m[8];
    const double mi1 = m22 * m33 - m32 * m23, mi2 = m21 * m33 - m31 * m23, mi3 = m21 * m32 - m31 * m22;
    const double det = (m11 * mi1 - m12 * mi2) + m13 * mi3;
    for (int i = 0; i < 9; ++i) R9[i] = axes[i];
    if (det != 0.0) {
        const double inv[9] = {mi1 / det, -mi2 / det, mi3 / det,
                               (m13 * m32 - m33 * m12) / det, (m11 * m33 - m31 * m13) / det, (m12 * m31 - m32 * m11) / det,
                               (m12 * m23 - m22 * m13) / det, (m13 * m21 - m23 * m11) / det, (m11 * m22 - m21 * m12) / det};
        for (int i = 0; i < 9; ++i) R9[i] = inv[i];
    }
    for (int r = 0; r < 3; ++r) t3[r] = cm[r] + (((R9[r] * -cm[0]) + (R9[3 + r] * -cm[1])) + (R9[6 + r] * -cm[2]));
    return MOLAR_HIP_OK;
}

}  // extern "C"
