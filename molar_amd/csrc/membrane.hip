// membrane.hip — the per-lipid geometry loop of molar_membrane on gfx950.
//
// One iteration of Membrane::smooth (molar_membrane/src/lib.rs:661-812).  The reference runs the fit as a
// rayon par_iter over lipids and the marker averaging as a serial scatter loop; here
//   k_membrane_fit      one lane per lipid: local frame, patch markers into the frame (PBC shortest vector),
//                       6x6 normal equations + Cholesky, Voronoi cell by half-plane clipping, curvatures,
//                       fitted normal, cell area, fitted patch points, marker moved onto the surface;
//   k_membrane_average  one lane per lipid: gathers the fitted images of its marker from every valid patch
//                       that contains it, in the order the reference's scatter loop adds them (owner lipid
//                       ascending, patch order inside), so the f32 sum is the same sum.
// Per-lipid state of unbounded length (local points, Voronoi vertices) lives in HBM slices owned by the lane;
// the work per lipid is ~30 neighbours, so the kernel is latency bound and tiny next to the neighbour search.
// f32 throughout, in the reference's operation order (nalgebra gemv/cross/normalize, Cholesky::new + solve).
#include <algorithm>
#include <vector>

#include "boxmath.hpp"
#include "common.hpp"

namespace {

using namespace mh;

struct SmoothDev {
    uint32_t K;
    const molar_hip_box *box;
    const float *saved;        // [K][3] markers before the iteration
    float *head;               // [K][3] in/out
    float *normals;            // [K][3] in/out
    uint8_t *valid;            // [K] in/out
    const uint64_t *poff;      // [K+1]
    const uint64_t *pids;      // [E]
    float *coefs, *mean, *gauss, *pcurv, *pdirs, *area;
    uint32_t *nvert;
    uint64_t *neib;            // [E+4K]
    float *voro;               // [E+4K][3]
    float *fitted;             // [E][3]  (holds the local points while the lane works)
    float4 *vwork;             // [E+4K]  Voronoi vertices {x, y, next, id}
    const uint32_t *rev_off;   // [K+1]   transpose of the patch CSR
    const uint32_t *rev_entry; // [E]     flat patch entry
    const uint32_t *rev_owner; // [E]     lipid owning that entry
};

__device__ __forceinline__ V3 cross(V3 a, V3 b) {
    return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// nalgebra try_inverse for 3x3 (closed form, column-major)
__device__ bool inverse3(const float *m, float *o) {
    const float m11 = m[0], m21 = m[1], m31 = m[2], m12 = m[3], m22 = m[4], m32 = m[5], m13 = m[6], m23 = m[7], m33 = m[8];
    const float mi1 = m22 * m33 - m32 * m23;
    const float mi2 = m21 * m33 - m31 * m23;
    const float mi3 = m21 * m32 - m31 * m22;
    const float det = (m11 * mi1 - m12 * mi2) + m13 * mi3;
    if (det == 0.0f) return false;
    o[0] = mi1 / det;
    o[3] = (m13 * m32 - m33 * m12) / det;
    o[6] = (m12 * m23 - m22 * m13) / det;
    o[1] = -mi2 / det;
    o[4] = (m11 * m33 - m31 * m13) / det;
    o[7] = (m13 * m21 - m23 * m11) / det;
    o[2] = mi3 / det;
    o[5] = (m12 * m31 - m32 * m11) / det;
    o[8] = (m11 * m22 - m21 * m12) / det;
    return true;
}

// get_quad_coefs' solver (lib.rs:862): nalgebra Cholesky::new, then L y = b, L^T x = y.  a is column-major 6x6.
__device__ bool cholesky6_solve(float *a, float *b) {
    for (int j = 0; j < 6; ++j) {
        for (int k = 0; k < j; ++k) {
            const float factor = -a[k * 6 + j];
            for (int r = j; r < 6; ++r) a[j * 6 + r] = factor * a[k * 6 + r] + a[j * 6 + r];
        }
        const float diag = a[j * 6 + j];
        if (!(diag > 0.0f)) return false;
        const float denom = __builtin_sqrtf(diag);
        a[j * 6 + j] = denom;
        for (int r = j + 1; r < 6; ++r) a[j * 6 + r] /= denom;
    }
    for (int i = 0; i < 6; ++i) {
        const float coeff = b[i] / a[i * 6 + i];
        b[i] = coeff;
        for (int r = i + 1; r < 6; ++r) b[r] = (-coeff) * a[i * 6 + r] + b[r];
    }
    for (int i = 5; i >= 0; --i) {
        float dot = 0.0f;
        for (int r = i + 1; r < 6; ++r) dot += a[i * 6 + r] * b[r];
        b[i] = (b[i] - dot) / a[i * 6 + i];
    }
    return true;
}

__device__ __forceinline__ float z_surf(float x, float y, const float *c) {   // lib.rs:870-879
    return ((((c[0] * x * x + c[1] * y * y) + c[2] * x * y) + c[3] * x) + c[4] * y) + c[5];
}

// Voronoi vertex in the lane's HBM slice: {x, y, ccw neighbour, id of the point that made the ccw edge}
struct Vert {
    float x, y;
    uint32_t next;
    int32_t id;
};
__device__ __forceinline__ Vert vload(const float4 *w, uint32_t i) {
    const float4 q = w[i];
    return Vert{q.x, q.y, __float_as_uint(q.z), (int32_t)__float_as_uint(q.w)};
}
__device__ __forceinline__ void vstore(float4 *w, uint32_t i, Vert v) {
    w[i] = make_float4(v.x, v.y, __uint_as_float(v.next), __uint_as_float((uint32_t)v.id));
}
__device__ __forceinline__ float vdist(const float4 *w, uint32_t i, float lx, float ly, float r2) {
    const float4 q = w[i];
    return (lx * q.x + ly * q.y) - r2;     // line.pos.dot(pos) - r2  (voronoi_cell.rs:83-85)
}

// VoronoiCell::add_point (voronoi_cell.rs:107-205).  Returns false only where the reference would never
// return (no vertex on the inner side, e.g. NaN input) - the caller then drops the lipid.
__device__ bool voro_add_point(float4 *w, uint32_t &nv, uint32_t &init, float px, float py, int32_t id) {
    const float TOL = 1e-10f;
    const float lx = 0.5f * px, ly = 0.5f * py;
    const float r2 = lx * lx + ly * ly;
    uint32_t cur = init, guard = 0;
    float cur_d = vdist(w, cur, lx, ly, r2);
    while (cur_d >= TOL) {
        cur = __float_as_uint(w[cur].z);
        cur_d = vdist(w, cur, lx, ly, r2);
        if (++guard > nv) return false;
    }
    init = cur;
    uint32_t c1_in, c1_out, c2_in, c2_out;
    float c1_ind, c1_outd, c2_ind, c2_outd;
    for (;;) {
        const uint32_t nx = __float_as_uint(w[cur].z);
        if (nx == init) return true;               // every vertex is inside: nothing to cut
        const float nd = vdist(w, nx, lx, ly, r2);
        if (nd >= TOL) {
            c1_in = cur; c1_ind = cur_d; c1_out = nx; c1_outd = nd;
            cur = nx; cur_d = nd;
            break;
        }
        cur = nx; cur_d = nd;
    }
    guard = 0;
    for (;;) {
        const uint32_t nx = __float_as_uint(w[cur].z);
        const float nd = vdist(w, nx, lx, ly, r2);
        if (nd < TOL) {
            c2_out = cur; c2_outd = cur_d; c2_in = nx; c2_ind = nd;
            break;
        }
        cur = nx; cur_d = nd;
        if (++guard > nv) return false;
    }
    {   // cut #2 (:173-195)
        const Vert o = vload(w, c2_out), in = vload(w, c2_in);
        const float frac = c2_outd / (fabsf(c2_ind) + c2_outd);
        const float x = (1.0f - frac) * o.x + frac * in.x;
        const float y = (1.0f - frac) * o.y + frac * in.y;
        if (c1_out != c2_out) {
            vstore(w, c2_out, Vert{x, y, o.next, o.id});
            Vert f = vload(w, c1_out);
            f.next = c2_out;
            vstore(w, c1_out, f);
        } else {
            vstore(w, nv, Vert{x, y, c2_in, o.id});
            Vert f = vload(w, c1_out);
            f.next = nv;
            vstore(w, c1_out, f);
            nv += 1;
        }
    }
    {   // cut #1 (:197-202)
        const Vert o = vload(w, c1_out), in = vload(w, c1_in);
        const float frac = c1_outd / (fabsf(c1_ind) + c1_outd);
        vstore(w, c1_out, Vert{(1.0f - frac) * o.x + frac * in.x, (1.0f - frac) * o.y + frac * in.y, o.next, id});
    }
    return true;
}

// Eigenpairs of the symmetric 2x2 [[a, b], [b, c]].  nalgebra's symmetric_eigen leaves order and sign
// unspecified; this engine returns descending eigenvalues and eigenvectors whose first non-zero component
// is positive.
__device__ void eig2_sym(float a, float b, float c, float *w, float *v) {
    const float t = 0.5f * (a - c), m = 0.5f * (a + c);
    const float h = __builtin_sqrtf(t * t + b * b);
    w[0] = m + h;
    w[1] = m - h;
    float x, y;
    if (b == 0.0f) {
        x = a >= c ? 1.0f : 0.0f;
        y = a >= c ? 0.0f : 1.0f;
    } else {
        if (t >= 0.0f) { x = t + h; y = b; } else { x = b; y = h - t; }
        const float n = __builtin_sqrtf(x * x + y * y);
        x /= n; y /= n;
        if (x < 0.0f || (x == 0.0f && y < 0.0f)) { x = -x; y = -y; }
    }
    v[0] = x; v[1] = y;
    float x2 = -y, y2 = x;
    if (x2 < 0.0f || (x2 == 0.0f && y2 < 0.0f)) { x2 = -x2; y2 = -y2; }
    v[2] = x2; v[3] = y2;
}

__global__ __launch_bounds__(64) void k_membrane_fit(SmoothDev A) {
    const uint32_t i = blockIdx.x * 64u + threadIdx.x;
    if (i >= A.K || !A.valid[i]) return;
    const uint64_t p0 = A.poff[i];
    const uint32_t np = (uint32_t)(A.poff[i + 1] - p0);
    const uint64_t slot = p0 + 4ull * i;
    const V3 nrm = v3(A.normals[3 * i], A.normals[3 * i + 1], A.normals[3 * i + 2]);
    float to_lab[9], to_local[9];
    {   // get_to_lab_transform (lipid_molecule.rs:190-196)
        const V3 c0 = cross(nrm, v3(1.0f, 0.0f, 0.0f));
        const V3 c1 = cross(nrm, c0);
        to_lab[0] = c0.x; to_lab[1] = c0.y; to_lab[2] = c0.z;
        to_lab[3] = c1.x; to_lab[4] = c1.y; to_lab[5] = c1.z;
        to_lab[6] = -nrm.x; to_lab[7] = -nrm.y; to_lab[8] = -nrm.z;
    }
    if (!inverse3(to_lab, to_local)) { A.valid[i] = 0; return; }
    const V3 c = v3(A.saved[3 * i], A.saved[3 * i + 1], A.saved[3 * i + 2]);
    const molar_hip_box &box = *A.box;
    float *lp = A.fitted + 3 * p0;
    float m[36], cf[6];
    for (int k = 0; k < 36; ++k) m[k] = 0.0f;
    for (int k = 0; k < 6; ++k) cf[k] = 0.0f;
    for (uint32_t q = 0; q < np; ++q) {   // local points + normal equations (lib.rs:685-689, 851-860)
        const uint64_t j = A.pids[p0 + q];
        const V3 s = v3(A.saved[3 * j], A.saved[3 * j + 1], A.saved[3 * j + 2]);
        const V3 l = mat_vec(to_local, shortest_vector(box, s - c, MOLAR_HIP_PBC_FULL));
        lp[3 * q] = l.x; lp[3 * q + 1] = l.y; lp[3 * q + 2] = l.z;
        const float pw[6] = {l.x * l.x, l.y * l.y, l.x * l.y, l.x, l.y, 1.0f};
#pragma unroll
        for (int cc = 0; cc < 6; ++cc)
#pragma unroll
            for (int r = 0; r < 6; ++r) m[cc * 6 + r] += pw[r] * pw[cc];
#pragma unroll
        for (int r = 0; r < 6; ++r) cf[r] += pw[r] * l.z;
    }
    if (!cholesky6_solve(m, cf)) { A.valid[i] = 0; return; }

    float4 *w = A.vwork + slot;
    vstore(w, 0, Vert{-10.0f, -10.0f, 1u, -1});     // VoronoiCell::new(-10, 10, -10, 10)  (voronoi_cell.rs:62-80)
    vstore(w, 1, Vert{10.0f, -10.0f, 2u, -2});
    vstore(w, 2, Vert{10.0f, 10.0f, 3u, -3});
    vstore(w, 3, Vert{-10.0f, 10.0f, 0u, -4});
    uint32_t nv = 4, init = 0;
    for (uint32_t q = 0; q < np; ++q) {
        if (!voro_add_point(w, nv, init, lp[3 * q], lp[3 * q + 1], (int32_t)A.pids[p0 + q])) { A.valid[i] = 0; return; }
    }
    uint32_t n_vert = 0, n_neib = 0;                 // direct neighbours (lib.rs:706-726)
    {
        uint32_t cur = init;
        do {
            const Vert v = vload(w, cur);
            if (v.id >= 0) A.neib[slot + n_neib++] = (uint64_t)v.id;
            ++n_vert;
            cur = v.next;
        } while (cur != init);
    }
    if (n_neib < n_vert) { A.valid[i] = 0; return; }  // a wall vertex survived: open cell
    A.nvert[i] = n_vert;
#pragma unroll
    for (int k = 0; k < 6; ++k) A.coefs[6 * i + k] = cf[k];
    {   // compute_curvature_and_normal (lipid_molecule.rs:134-187)
        const float a = cf[0], b = cf[1], cq = cf[2], d = cf[3], e = cf[4];
        const float E = 1.0f + d * d, F = d * e, G = 1.0f + e * e;
        const float L = 2.0f * a, M = cq, N = 2.0f * b;
        const float Z = E * G - F * F;
        A.gauss[i] = (L * N - M * M) / Z;
        A.mean[i] = 0.5f * ((E * N - 2.0f * F * M) + G * L) / Z;
        const float gl = __builtin_sqrtf((d * d + e * e) + 1.0f);
        const V3 fn = mat_vec(to_lab, v3(d / gl, e / gl, -1.0f / gl));
        A.normals[3 * i] = fn.x; A.normals[3 * i + 1] = fn.y; A.normals[3 * i + 2] = fn.z;
        float ev[2], evec[4];
        eig2_sym((E * L - F * M) / Z, (G * M - F * L) / Z, (G * N - F * M) / Z, ev, evec);
        A.pcurv[2 * i] = ev[0]; A.pcurv[2 * i + 1] = ev[1];
        for (int k = 0; k < 2; ++k) {
            const V3 pd = mat_vec(to_lab, v3(evec[2 * k], evec[2 * k + 1], 0.0f));
            A.pdirs[6 * i + 3 * k] = pd.x; A.pdirs[6 * i + 3 * k + 1] = pd.y; A.pdirs[6 * i + 3 * k + 2] = pd.z;
        }
    }
    {   // cell vertices on the fitted surface, lab frame, still relative to the marker; fan area (lib.rs:731-752)
        uint32_t cur = init;
        V3 first = v3(0, 0, 0), prev = v3(0, 0, 0);
        float ar = 0.0f;
        for (uint32_t k = 0; k < n_vert; ++k) {
            const Vert v = vload(w, cur);
            const V3 p = mat_vec(to_lab, v3(v.x, v.y, z_surf(v.x, v.y, cf)));
            float *dst = A.voro + 3 * (slot + k);
            dst[0] = p.x; dst[1] = p.y; dst[2] = p.z;
            if (k == 0) first = p;
            else ar += 0.5f * __builtin_sqrtf(norm2(cross(prev, p)));
            prev = p;
            cur = v.next;
        }
        ar += 0.5f * __builtin_sqrtf(norm2(cross(prev, first)));
        A.area[i] = ar;
    }
    for (uint32_t q = 0; q < np; ++q) {   // fitted patch points (lib.rs:760-768); overwrites the local point in place
        const float x = lp[3 * q], y = lp[3 * q + 1], z = lp[3 * q + 2];
        const V3 t = mat_vec(to_lab, v3(0.0f, 0.0f, z_surf(x, y, cf) - z));
        const uint64_t j = A.pids[p0 + q];
        lp[3 * q] = A.saved[3 * j] + t.x;
        lp[3 * q + 1] = A.saved[3 * j + 1] + t.y;
        lp[3 * q + 2] = A.saved[3 * j + 2] + t.z;
    }
    if (fabsf(cf[5]) > 0.5f) { A.valid[i] = 0; return; }   // fitted surface too far from the marker (lib.rs:774-777)
    const V3 t = mat_vec(to_lab, v3(0.0f, 0.0f, cf[5]));
    A.head[3 * i] += t.x; A.head[3 * i + 1] += t.y; A.head[3 * i + 2] += t.z;
}

// lib.rs:781-809.  `fitted_head` holds the markers after k_membrane_fit; the average is written to `head`.
__global__ __launch_bounds__(64) void k_membrane_average(SmoothDev A, const float *fitted_head) {
    const uint32_t i = blockIdx.x * 64u + threadIdx.x;
    if (i >= A.K || !A.valid[i]) return;
    float n = 1.0f;
    V3 s = v3(fitted_head[3 * i], fitted_head[3 * i + 1], fitted_head[3 * i + 2]);
    for (uint32_t r = A.rev_off[i]; r < A.rev_off[i + 1]; ++r) {
        if (!A.valid[A.rev_owner[r]]) continue;
        const float *p = A.fitted + 3ull * A.rev_entry[r];
        n += 1.0f;
        s = s + v3(p[0], p[1], p[2]);
    }
    const V3 h = v3(s.x / n, s.y / n, s.z / n);
    A.head[3 * i] = h.x; A.head[3 * i + 1] = h.y; A.head[3 * i + 2] = h.z;
    const uint64_t slot = A.poff[i] + 4ull * i;
    for (uint32_t k = 0; k < A.nvert[i]; ++k) {
        float *v = A.voro + 3 * (slot + k);
        v[0] += h.x; v[1] += h.y; v[2] += h.z;
    }
}

struct Blob {
    size_t size = 0;
    size_t take(size_t bytes) {
        const size_t at = size;
        size += (bytes + 15) & ~size_t(15);
        return at;
    }
};

}  // namespace

extern "C" int molar_hip_membrane_smooth(molar_hip_ctx *c, const molar_hip_membrane_patches *P, const float *box9,
                                         molar_hip_membrane_state *S) {
    if (!c) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "null context");
    MH_HIP(hipSetDevice(c->device));
    if (!P || !S || !box9) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "membrane_smooth: null argument");
    const size_t K = P->nlipids;
    if (K == 0) return MOLAR_HIP_OK;
    if (!P->patch_offsets || !S->head_markers || !S->normals || !S->valid)
        return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "membrane_smooth: null array");
    if (K >= (1ull << 31)) return fail(MOLAR_HIP_ERR_TOO_LARGE, "membrane_smooth: lipid ids must fit i32 (voronoi_cell.rs:17)");
    const size_t E = (size_t)P->patch_offsets[K];
    if (E && !P->patch_ids) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "membrane_smooth: patch_ids missing");
    if (E >= (1ull << 32)) return fail(MOLAR_HIP_ERR_TOO_LARGE, "membrane_smooth: %zu patch entries", E);
    if (P->patch_offsets[0] != 0) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "membrane_smooth: patch_offsets[0] != 0");
    molar_hip_box box;
    MH_TRY(molar_hip_box_from_matrix(box9, &box));
    const size_t slots = E + 4 * K;

    // transpose of the patch CSR: for each lipid, the patch entries that point at it, ordered by
    // (owner lipid, position in the owner's patch) = the order of the reference's scatter loop
    std::vector<uint32_t> rev_off(K + 1, 0), rev_entry(E), rev_owner(E);
    for (size_t i = 0; i < K; ++i) {
        if (P->patch_offsets[i + 1] < P->patch_offsets[i]) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "membrane_smooth: offsets not monotone");
        for (uint64_t q = P->patch_offsets[i]; q < P->patch_offsets[i + 1]; ++q) {
            if (P->patch_ids[q] >= K) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "membrane_smooth: patch id %llu out of range", (unsigned long long)P->patch_ids[q]);
            rev_off[P->patch_ids[q] + 1]++;
        }
    }
    for (size_t i = 0; i < K; ++i) rev_off[i + 1] += rev_off[i];
    {
        std::vector<uint32_t> cursor(rev_off.begin(), rev_off.end() - 1);
        for (size_t i = 0; i < K; ++i)
            for (uint64_t q = P->patch_offsets[i]; q < P->patch_offsets[i + 1]; ++q) {
                const uint32_t at = cursor[P->patch_ids[q]]++;
                rev_entry[at] = (uint32_t)q;
                rev_owner[at] = (uint32_t)i;
            }
    }

    // one blob: [in/out state | inputs | device-only work]
    Blob L;
    const size_t o_head = L.take(K * 12), o_norm = L.take(K * 12), o_valid = L.take(K), o_coefs = L.take(K * 24),
                 o_mean = L.take(K * 4), o_gauss = L.take(K * 4), o_pcurv = L.take(K * 8), o_pdirs = L.take(K * 24),
                 o_area = L.take(K * 4), o_nvert = L.take(K * 4), o_neib = L.take(slots * 8), o_voro = L.take(slots * 12),
                 o_fitted = L.take(E * 12);
    const size_t io_bytes = L.size;
    const size_t o_poff = L.take((K + 1) * 8), o_pids = L.take(E * 8), o_roff = L.take((K + 1) * 4), o_rent = L.take(E * 4),
                 o_rown = L.take(E * 4), o_box = L.take(sizeof box);
    const size_t up_bytes = L.size;
    const size_t o_saved = L.take(K * 12), o_fh = L.take(K * 12), o_vwork = L.take(slots * 16);
    MH_TRY(c->m_partials.reserve(L.size));
    MH_TRY(ensure_pinned(c, up_bytes));
    char *h = (char *)c->h_pinned, *d = c->m_partials.as<char>();
    auto put = [&](size_t off, const void *src, size_t bytes, bool zero_if_null = true) {
        if (src) std::memcpy(h + off, src, bytes);
        else if (zero_if_null) std::memset(h + off, 0, bytes);
    };
    put(o_head, S->head_markers, K * 12); put(o_norm, S->normals, K * 12); put(o_valid, S->valid, K);
    put(o_coefs, S->quad_coefs, K * 24); put(o_mean, S->mean_curv, K * 4); put(o_gauss, S->gauss_curv, K * 4);
    put(o_pcurv, S->princ_curvs, K * 8); put(o_pdirs, S->princ_dirs, K * 24); put(o_area, S->area, K * 4);
    put(o_nvert, S->nvert, K * 4); put(o_neib, S->neib_ids, slots * 8); put(o_voro, S->voro_vertexes, slots * 12);
    put(o_fitted, S->fitted_patch_points, E * 12);
    put(o_poff, P->patch_offsets, (K + 1) * 8); put(o_pids, P->patch_ids, E * 8);
    put(o_roff, rev_off.data(), (K + 1) * 4); put(o_rent, rev_entry.data(), E * 4); put(o_rown, rev_owner.data(), E * 4);
    put(o_box, &box, sizeof box);
    {
        Prof span(c, 4);
        MH_HIP(hipMemcpyAsync(d, h, up_bytes, hipMemcpyHostToDevice, c->stream));
        MH_HIP(hipMemcpyAsync(d + o_saved, d + o_head, K * 12, hipMemcpyDeviceToDevice, c->stream));
        SmoothDev A;
        A.K = (uint32_t)K;
        A.box = (const molar_hip_box *)(d + o_box);
        A.saved = (const float *)(d + o_saved);
        A.head = (float *)(d + o_head); A.normals = (float *)(d + o_norm); A.valid = (uint8_t *)(d + o_valid);
        A.poff = (const uint64_t *)(d + o_poff); A.pids = (const uint64_t *)(d + o_pids);
        A.coefs = (float *)(d + o_coefs); A.mean = (float *)(d + o_mean); A.gauss = (float *)(d + o_gauss);
        A.pcurv = (float *)(d + o_pcurv); A.pdirs = (float *)(d + o_pdirs); A.area = (float *)(d + o_area);
        A.nvert = (uint32_t *)(d + o_nvert); A.neib = (uint64_t *)(d + o_neib); A.voro = (float *)(d + o_voro);
        A.fitted = (float *)(d + o_fitted); A.vwork = (float4 *)(d + o_vwork);
        A.rev_off = (const uint32_t *)(d + o_roff); A.rev_entry = (const uint32_t *)(d + o_rent);
        A.rev_owner = (const uint32_t *)(d + o_rown);
        const uint32_t nb = (uint32_t)((K + 63) / 64);
        hipLaunchKernelGGL(k_membrane_fit, dim3(nb), dim3(64), 0, c->stream, A);
        MH_HIP(hipMemcpyAsync(d + o_fh, d + o_head, K * 12, hipMemcpyDeviceToDevice, c->stream));
        hipLaunchKernelGGL(k_membrane_average, dim3(nb), dim3(64), 0, c->stream, A, (const float *)(d + o_fh));
        MH_HIP(hipGetLastError());
        MH_HIP(hipMemcpyAsync(h, d, io_bytes, hipMemcpyDeviceToHost, c->stream));
    }
    MH_HIP(hipStreamSynchronize(c->stream));
    auto get = [&](void *dst, size_t off, size_t bytes) {
        if (dst) std::memcpy(dst, h + off, bytes);
    };
    get(S->head_markers, o_head, K * 12); get(S->normals, o_norm, K * 12); get(S->valid, o_valid, K);
    get(S->quad_coefs, o_coefs, K * 24); get(S->mean_curv, o_mean, K * 4); get(S->gauss_curv, o_gauss, K * 4);
    get(S->princ_curvs, o_pcurv, K * 8); get(S->princ_dirs, o_pdirs, K * 24); get(S->area, o_area, K * 4);
    get(S->nvert, o_nvert, K * 4); get(S->neib_ids, o_neib, slots * 8); get(S->voro_vertexes, o_voro, slots * 12);
    get(S->fitted_patch_points, o_fitted, E * 12);
    return MOLAR_HIP_OK;
}
