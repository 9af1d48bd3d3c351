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
// molar_hip_membrane_frame_* (second half of this file) chains a whole frame of Membrane::compute on the stream.
// Per-lipid state of unbounded length (local points, Voronoi vertices) lives in HBM slices owned by the lane;
// the work per lipid is ~30 neighbours, so the kernel is latency bound and tiny next to the neighbour search.
// f32 throughout, in the reference's operation order (nalgebra gemv/cross/normalize, Cholesky::new + solve).
#include <algorithm>
#include <vector>

#include <cmath>

#include "boxmath.hpp"
#include "common.hpp"
#include "stages.hpp"

namespace {

using namespace mh;

struct SmoothDev {
    uint32_t K;
    molar_hip_box box;         // by value: read through the scalar cache, never reloaded behind a store
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
    float *fitted;             // [E][3]
    float4 *vwork;             // [E+4K]  Voronoi vertices {x, y, next, id} of the patches too long for LDS
    float4 *pwork;             // [E]     local points {x, y, z, id} of those patches
    const uint32_t *rev_off;   // [K+1]   transpose of the patch CSR
    const uint32_t *rev_entry; // [E]     flat patch entry
    const uint32_t *rev_owner; // [E]     lipid owning that entry
    uint8_t *redo = nullptr;   // [K] or NULL: k_membrane_fit_lanes leaves 1 for the lipids it hands to k_membrane_fit (which then
                               //             takes only those), 0 for the ones it has done
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

// Voronoi vertex {x, y, ccw neighbour, id of the point that made the ccw edge}.  A lane's vertices sit either in the
// workgroup's LDS, vertex-major (element v of lane l at [v * lanes + l]: the cell is a linked list walked with dependent
// loads, ~30 ns a step there against ~500 ns in HBM), or, for patches of more than VORO_LDS - 4 members, in the lane's
// slice of `vwork` in HBM.  `Verts` hides which: base pointer (generic address space) + element stride.
constexpr uint32_t VORO_LDS = 64;        // vertices per lane held in LDS (1 KB per lane)
struct Vert {
    float x, y;
    uint32_t next;
    int32_t id;
};
struct Verts {
    float4 *p;
    uint32_t stride;
    __device__ __forceinline__ float4 &at(uint32_t i) const { return p[(size_t)i * stride]; }
};
__device__ __forceinline__ Vert vload(const Verts &w, uint32_t i) {
    const float4 q = w.at(i);
    return Vert{q.x, q.y, __float_as_uint(q.z), (int32_t)__float_as_uint(q.w)};
}
__device__ __forceinline__ void vstore(const Verts &w, uint32_t i, Vert v) {
    w.at(i) = make_float4(v.x, v.y, __uint_as_float(v.next), __uint_as_float((uint32_t)v.id));
}
__device__ __forceinline__ float vdist(const Verts &w, uint32_t i, float lx, float ly, float r2) {
    const float4 q = w.at(i);
    return (lx * q.x + ly * q.y) - r2;     // line.pos.dot(pos) - r2  (voronoi_cell.rs:83-85)
}

// VoronoiCell::add_point (voronoi_cell.rs:107-205).  Returns false only where the reference would never
// return (no vertex on the inner side, e.g. NaN input) - the caller then drops the lipid.
__device__ bool voro_add_point(const Verts &w, uint32_t &nv, uint32_t &init, float px, float py, int32_t id) {
    const float TOL = 1e-10f;
    const float lx = 0.5f * px, ly = 0.5f * py;
    const float r2 = lx * lx + ly * ly;
    uint32_t cur = init, guard = 0;
    float cur_d = vdist(w, cur, lx, ly, r2);
    while (cur_d >= TOL) {
        cur = __float_as_uint(w.at(cur).z);
        cur_d = vdist(w, cur, lx, ly, r2);
        if (++guard > nv) return false;
    }
    init = cur;
    uint32_t c1_in, c1_out, c2_in, c2_out;
    float c1_ind, c1_outd, c2_ind, c2_outd;
    for (;;) {
        const uint32_t nx = __float_as_uint(w.at(cur).z);
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
        const uint32_t nx = __float_as_uint(w.at(cur).z);
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

// A patch member in the lipid's local frame {x, y, z, id}: in LDS beside the Voronoi vertices (patches of up to PTS_LDS
// members), else in the lane's slice of `pwork`.
constexpr uint32_t PTS_LDS = 60;
constexpr size_t FIT_LDS_BYTES = (size_t)(VORO_LDS + PTS_LDS) * 64 * sizeof(float4);      // 124 KB for a 64-lane workgroup
struct Pts {
    float4 *p;
    uint32_t stride;
    __device__ __forceinline__ float4 &at(uint32_t i) const { return p[(size_t)i * stride]; }
};

// One lane per lipid; every loop over the patch is a chain of dependent steps, so what a lane touches more than once
// (local points, cell vertices) sits in LDS and the gathers of the neighbours' markers are issued four at a time.
__global__ __launch_bounds__(64) void k_membrane_fit(SmoothDev A) {
    extern __shared__ float4 fit_lds[];
    const uint32_t lanes = blockDim.x;               // lipids per workgroup (16, 32 or 64: launch_fit)
    const uint32_t i = blockIdx.x * lanes + threadIdx.x;
    if (i >= A.K || !A.valid[i]) return;
    if (A.redo && !A.redo[i]) return;                // done by k_membrane_fit_lanes
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
    const molar_hip_box &box = A.box;
    const bool in_lds = np + 4u <= VORO_LDS && np <= PTS_LDS;
    const Verts w = in_lds ? Verts{fit_lds + threadIdx.x, lanes} : Verts{A.vwork + slot, 1u};
    const Pts pt = in_lds ? Pts{fit_lds + VORO_LDS * lanes + threadIdx.x, lanes} : Pts{A.pwork + p0, 1u};
    float m[36], cf[6];
    for (int k = 0; k < 36; ++k) m[k] = 0.0f;
    for (int k = 0; k < 6; ++k) cf[k] = 0.0f;
    for (uint32_t q0 = 0; q0 < np; q0 += 4u) {   // local points + normal equations (lib.rs:685-689, 851-860)
        uint32_t jj[4];
        V3 ss[4];
#pragma unroll
        for (uint32_t u = 0; u < 4u; ++u) jj[u] = (uint32_t)A.pids[p0 + (q0 + u < np ? q0 + u : np - 1u)];
#pragma unroll
        for (uint32_t u = 0; u < 4u; ++u) ss[u] = v3(A.saved[3 * jj[u]], A.saved[3 * jj[u] + 1], A.saved[3 * jj[u] + 2]);
#pragma unroll
        for (uint32_t u = 0; u < 4u; ++u) {
            if (q0 + u >= np) break;
            const V3 l = mat_vec(to_local, shortest_vector(box, ss[u] - c, MOLAR_HIP_PBC_FULL));
            pt.at(q0 + u) = make_float4(l.x, l.y, l.z, __uint_as_float(jj[u]));
            const float pw[6] = {l.x * l.x, l.y * l.y, l.x * l.y, l.x, l.y, 1.0f};
#pragma unroll
            for (int cc = 0; cc < 6; ++cc)
#pragma unroll
                for (int r = 0; r < 6; ++r) m[cc * 6 + r] += pw[r] * pw[cc];
#pragma unroll
            for (int r = 0; r < 6; ++r) cf[r] += pw[r] * l.z;
        }
    }
    if (!cholesky6_solve(m, cf)) { A.valid[i] = 0; return; }

    vstore(w, 0, Vert{-10.0f, -10.0f, 1u, -1});     // VoronoiCell::new(-10, 10, -10, 10)  (voronoi_cell.rs:62-80)
    vstore(w, 1, Vert{10.0f, -10.0f, 2u, -2});
    vstore(w, 2, Vert{10.0f, 10.0f, 3u, -3});
    vstore(w, 3, Vert{-10.0f, 10.0f, 0u, -4});
    uint32_t nv = 4, init = 0;
    for (uint32_t q = 0; q < np; ++q) {
        const float4 r = pt.at(q);
        if (!voro_add_point(w, nv, init, r.x, r.y, (int32_t)__float_as_uint(r.w))) { A.valid[i] = 0; return; }
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
    float *fp = A.fitted + 3 * p0;
    for (uint32_t q0 = 0; q0 < np; q0 += 4u) {   // fitted patch points (lib.rs:760-768)
        float4 r[4];
        V3 ss[4];
#pragma unroll
        for (uint32_t u = 0; u < 4u; ++u) r[u] = pt.at(q0 + u < np ? q0 + u : np - 1u);
#pragma unroll
        for (uint32_t u = 0; u < 4u; ++u) {
            const uint32_t j = __float_as_uint(r[u].w);
            ss[u] = v3(A.saved[3 * j], A.saved[3 * j + 1], A.saved[3 * j + 2]);
        }
#pragma unroll
        for (uint32_t u = 0; u < 4u; ++u) {
            if (q0 + u >= np) break;
            const V3 t = mat_vec(to_lab, v3(0.0f, 0.0f, z_surf(r[u].x, r[u].y, cf) - r[u].z));
            fp[3 * (q0 + u)] = ss[u].x + t.x;
            fp[3 * (q0 + u) + 1] = ss[u].y + t.y;
            fp[3 * (q0 + u) + 2] = ss[u].z + t.z;
        }
    }
    if (fabsf(cf[5]) > 0.5f) { A.valid[i] = 0; return; }   // fitted surface too far from the marker (lib.rs:774-777)
    const V3 t = mat_vec(to_lab, v3(0.0f, 0.0f, cf[5]));
    A.head[3 * i] += t.x; A.head[3 * i + 1] += t.y; A.head[3 * i + 2] += t.z;
}

// The same fit with SIXTEEN lanes per lipid (round 6).  One lane per lipid is a serial chain of ~100 us - 40 patch members x
// (gather, 42 multiply-adds into the normal equations), a Cholesky solve, 40 half-plane clips each walking a linked list with
// dependent LDS loads - and a bilayer of 4000 lipids cannot hide it behind other lipids.  Here a lipid's lanes share the work
// WITHOUT changing a single rounding, so every output equals the one-lane kernel's bit for bit:
//   * local points: one patch member per lane (gathers in parallel), kept in LDS;
//   * normal equations: the 36 + 6 accumulators are dealt over the lanes (three each); every accumulator still runs over the
//     members in patch order - the reference's order of additions (lib.rs:851-860) - only 3 instead of 42 per lane and member;
//   * Cholesky (get_quad_coefs, lib.rs:862): every lane gathers the sums and solves, redundantly;
//   * Voronoi cell: the ring of vertices lives one vertex per lane, in ring order from the cell's `init` vertex.  A half-plane
//     (VoronoiCell::add_point, voronoi_cell.rs:107-205) is classified by all lanes at once (one ballot); the reference's walk -
//     first inner vertex from init, first outer one after it (cut #1), next inner one (cut #2) - becomes three bit scans, the
//     two new vertices are the reference's expressions on values fetched by shuffles, and the ring closes up by one shift.
//     Cells of more than 16 vertices at any moment, patches of more than FIT_PTS members and non-finite distances are handed
//     to k_membrane_fit (redo flag): nothing has been written for them by then;
//   * outputs (neighbour ids, cell vertices on the fitted surface, fan area - summed in ring order -, fitted patch points) by
//     the lanes that hold the values.
constexpr uint32_t FIT_G = 16;           // lanes per lipid
constexpr uint32_t FIT_PTS = 96;         // patch members per lipid in LDS

__global__ __launch_bounds__(64) void k_membrane_fit_lanes(SmoothDev A) {
    // a patch member in the lipid's local frame, as the seven factors of its normal-equation terms and its id:
    // {x x, y y, x y, x, y, 1, z, id} - the accumulators pick their two factors by index (an LDS read each, no selects)
    __shared__ float pts_s[64 / FIT_G][FIT_PTS][8];
    const uint32_t lane = threadIdx.x & 63u, g = lane / FIT_G, sub = lane % FIT_G, gl = g * FIT_G;      // gl: first lane of the group
    const uint32_t i = blockIdx.x * (64u / FIT_G) + g;
    if (i >= A.K) return;
    if (!A.valid[i]) {
        if (sub == 0u) A.redo[i] = 0;
        return;
    }
    const uint64_t p0 = A.poff[i];
    const uint32_t np = (uint32_t)(A.poff[i + 1] - p0);
    const uint64_t slot = p0 + 4ull * i;
    if (np > FIT_PTS) {                              // a patch too long for the LDS slice: the one-lane kernel
        if (sub == 0u) A.redo[i] = 1;
        return;
    }
    float (*pts)[8] = pts_s[g];
    const V3 nrm = v3(A.normals[3 * i], A.normals[3 * i + 1], A.normals[3 * i + 2]);
    float to_lab[9], to_local[9];
    {   // get_to_lab_transform (lipid_molecule.rs:190-196)
        const V3 c0 = cross(nrm, v3(1.0f, 0.0f, 0.0f));
        const V3 c1 = cross(nrm, c0);
        to_lab[0] = c0.x; to_lab[1] = c0.y; to_lab[2] = c0.z;
        to_lab[3] = c1.x; to_lab[4] = c1.y; to_lab[5] = c1.z;
        to_lab[6] = -nrm.x; to_lab[7] = -nrm.y; to_lab[8] = -nrm.z;
    }
    if (!inverse3(to_lab, to_local)) {
        if (sub == 0u) { A.valid[i] = 0; A.redo[i] = 0; }
        return;
    }
    const V3 c = v3(A.saved[3 * i], A.saved[3 * i + 1], A.saved[3 * i + 2]);
    const molar_hip_box &box = A.box;
    // Everything from here to the end of the Voronoi loop runs with WAVE-UNIFORM control flow: the four lipids of a wave differ in
    // patch length and in what a half-plane cuts, and a divergent region in the middle of this code would have the compiler
    // split live ranges across its join (ROCm 7.2 places such copies ahead of the EXEC restore: molar_amd/build.py audits the
    // ISA for it and refused the first version of this kernel).  Loops run to the wave's longest patch, a lipid's own length
    // and state select what is kept.
    uint32_t np_w = 0;
#pragma unroll
    for (uint32_t k = 0; k < 64u; k += FIT_G) {
        const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)np, (int)k);      // (lanes that left above read as whatever they held: bounded below)
        np_w = v > np_w && v <= FIT_PTS ? v : np_w;
    }
    // ---- local points (lib.rs:685-689): one member per lane
    for (uint32_t q0 = 0; q0 < np_w; q0 += FIT_G) {
        const uint32_t q = q0 + sub;
        if (q < np) {
            const uint32_t j = (uint32_t)A.pids[p0 + q];
            const V3 ss = v3(A.saved[3 * j], A.saved[3 * j + 1], A.saved[3 * j + 2]);
            const V3 l = mat_vec(to_local, shortest_vector(box, ss - c, MOLAR_HIP_PBC_FULL));
            float *rec = pts[q];
            rec[0] = l.x * l.x; rec[1] = l.y * l.y; rec[2] = l.x * l.y; rec[3] = l.x; rec[4] = l.y; rec[5] = 1.0f; rec[6] = l.z;
            rec[7] = __uint_as_float(j);
        }
    }
    __builtin_amdgcn_wave_barrier();
    // ---- normal equations (lib.rs:851-860): accumulator a = sub + 16 k; a < 36: m[a] (column a / 6, row a % 6), else cf[a - 36]
    float acc[3] = {0.0f, 0.0f, 0.0f};
    uint32_t ar_[3], ac_[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const uint32_t a = sub + FIT_G * (uint32_t)k;
        ar_[k] = a < 36u ? a % 6u : a - 36u;
        ac_[k] = a < 36u ? a / 6u : 6u;
    }
    for (uint32_t q = 0; q < np_w; ++q) {
        const float *rec = pts[q];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float prod = rec[ar_[k]] * rec[ac_[k]];       // pw[r] * pw[c], pw[r] * z
            acc[k] += q < np ? prod : 0.0f;              // (x + 0 == x: members past this lipid's patch change nothing)
        }
    }
    float m[36], cf[6];
#pragma unroll
    for (int a = 0; a < 36; ++a) m[a] = __shfl(acc[a / (int)FIT_G], (int)(gl + (uint32_t)(a % (int)FIT_G)), 64);
#pragma unroll
    for (int a = 36; a < 42; ++a) cf[a - 36] = __shfl(acc[a / (int)FIT_G], (int)(gl + (uint32_t)(a % (int)FIT_G)), 64);
    // state of this lipid: 0 running, 1 handed to the one-lane kernel, 2 dropped (valid = 0)
    uint32_t state = 0u;
    {   // cholesky6_solve without its early exits (same operations in the same order; a failed factorisation runs on into NaNs nobody reads)
        bool ok = true;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
#pragma unroll
            for (int k = 0; k < j; ++k) {
                const float factor = -m[k * 6 + j];
#pragma unroll
                for (int r = j; r < 6; ++r) m[j * 6 + r] = factor * m[k * 6 + r] + m[j * 6 + r];
            }
            const float diag = m[j * 6 + j];
            ok = ok && diag > 0.0f;
            const float denom = __builtin_sqrtf(diag);
            m[j * 6 + j] = denom;
#pragma unroll
            for (int r = j + 1; r < 6; ++r) m[j * 6 + r] /= denom;
        }
#pragma unroll
        for (int r0 = 0; r0 < 6; ++r0) {
            const float coeff = cf[r0] / m[r0 * 6 + r0];
            cf[r0] = coeff;
#pragma unroll
            for (int r = r0 + 1; r < 6; ++r) cf[r] = (-coeff) * m[r0 * 6 + r] + cf[r];
        }
#pragma unroll
        for (int r0 = 5; r0 >= 0; --r0) {
            float dot = 0.0f;
#pragma unroll
            for (int r = r0 + 1; r < 6; ++r) dot += m[r0 * 6 + r] * cf[r];
            cf[r0] = (cf[r0] - dot) / m[r0 * 6 + r0];
        }
        if (!ok) state = 2u;
    }
    // ---- Voronoi cell: vertex `sub` of the ring, counted from the cell's init vertex
    float vx = (sub == 0u || sub == 3u) ? -10.0f : 10.0f;          // VoronoiCell::new(-10, 10, -10, 10)  (voronoi_cell.rs:62-80)
    float vy = sub < 2u ? -10.0f : 10.0f;
    int32_t vid = -(int32_t)sub - 1;
    uint32_t nv = 4u;
    const float TOL = 1e-10f;
    for (uint32_t q = 0; q < np_w; ++q) {
        const bool run = state == 0u && q < np;
        const float px = pts[q][3], py = pts[q][4], pidf = pts[q][7];
        const float lx = 0.5f * px, ly = 0.5f * py;
        const float r2 = lx * lx + ly * ly;
        float d = (lx * vx + ly * vy) - r2;                          // line.pos.dot(pos) - r2  (voronoi_cell.rs:83-85)
        const uint32_t vmask = (1u << nv) - 1u;
        const bool here = sub < nv;
        uint32_t in_mask = (uint32_t)(__builtin_amdgcn_ballot_w64(here && d < TOL) >> gl) & vmask;
        const uint32_t out_any = (uint32_t)(__builtin_amdgcn_ballot_w64(here && d >= TOL) >> gl) & vmask;
        // a distance that is neither (not a number): the one-lane kernel walks it like the reference;  no vertex on the inner
        // side: the reference's walk never ends (its guard): the lipid is dropped
        const bool nan_d = (in_mask | out_any) != vmask, none_in = in_mask == 0u;
        const uint32_t s0 = none_in ? 0u : (uint32_t)__builtin_ctz(in_mask);          // the walk from init stops at the first inner vertex: the new init
        const bool step = run && !nan_d && !none_in;
        {   // the ring turns so that the new init is vertex 0 (s0 == 0: every lane reads itself)
            const uint32_t src = gl + (sub + s0 < nv ? sub + s0 : (sub + s0 - nv) & (FIT_G - 1u));
            const float rx = __shfl(vx, (int)src, 64), ry = __shfl(vy, (int)src, 64), rd = __shfl(d, (int)src, 64);
            const int32_t rid = __shfl(vid, (int)src, 64);
            if (step) { vx = rx; vy = ry; vid = rid; }
            d = rd;
            in_mask = s0 ? ((in_mask >> s0) | (in_mask << (nv - s0))) & vmask : in_mask;
        }
        const uint32_t out_mask = ~in_mask & vmask;
        const bool cut = step && out_mask != 0u;                                         // else: every vertex is inside, nothing to cut
        const uint32_t t1 = out_mask ? (uint32_t)__builtin_ctz(out_mask) : 1u;          // c1_out; c1_in = t1 - 1  (t1 >= 1: vertex 0 is inside)
        const uint32_t t1s = t1 ? t1 : 1u;
        const uint32_t rest = in_mask >> t1s;
        const uint32_t t2 = rest ? t1s + (uint32_t)__builtin_ctz(rest) : nv;            // c2_in (nv: the ring closes on vertex 0); c2_out = t2 - 1
        const uint32_t c2in = t2 >= nv ? 0u : t2;
        const uint32_t l1i = gl + ((t1s - 1u) & (FIT_G - 1u)), l1o = gl + (t1s & (FIT_G - 1u)), l2o = gl + ((t2 - 1u) & (FIT_G - 1u)), l2i = gl + c2in;
        const float x1i = __shfl(vx, (int)l1i, 64), y1i = __shfl(vy, (int)l1i, 64), d1i = __shfl(d, (int)l1i, 64);
        const float x1o = __shfl(vx, (int)l1o, 64), y1o = __shfl(vy, (int)l1o, 64), d1o = __shfl(d, (int)l1o, 64);
        const float x2o = __shfl(vx, (int)l2o, 64), y2o = __shfl(vy, (int)l2o, 64), d2o = __shfl(d, (int)l2o, 64);
        const float x2i = __shfl(vx, (int)l2i, 64), y2i = __shfl(vy, (int)l2i, 64), d2i = __shfl(d, (int)l2i, 64);
        const int32_t id2 = __shfl(vid, (int)l2o, 64);
        const float f2 = d2o / (fabsf(d2i) + d2o);                                      // cut #2 (:173-195)
        const float p2x = (1.0f - f2) * x2o + f2 * x2i, p2y = (1.0f - f2) * y2o + f2 * y2i;
        const float f1 = d1o / (fabsf(d1i) + d1o);                                      // cut #1 (:197-202)
        const float p1x = (1.0f - f1) * x1o + f1 * x1i, p1y = (1.0f - f1) * y1o + f1 * y1i;
        const uint32_t nv_new = t1s + 2u + (nv - (t2 > nv ? nv : t2));
        const bool grow = cut && nv_new > FIT_G;                                         // the cell outgrows the lanes: the one-lane kernel
        // the ring closes up: [0, t1) stay, t1 <- P1 (edge made by this point), t1 + 1 <- P2 (keeps the outer vertex's edge id), then old [t2, nv)
        const uint32_t from = sub >= t1s + 2u ? sub - (t1s + 2u) + t2 : sub;
        const uint32_t srcl = gl + (from & (FIT_G - 1u));
        const float sx = __shfl(vx, (int)srcl, 64), sy = __shfl(vy, (int)srcl, 64);
        const int32_t sid = __shfl(vid, (int)srcl, 64);
        const bool apply = cut && !grow;
        const bool is1 = sub == t1s, is2 = sub == t1s + 1u, tail = sub >= t1s + 2u;
        vx = apply ? (is1 ? p1x : is2 ? p2x : tail ? sx : vx) : vx;
        vy = apply ? (is1 ? p1y : is2 ? p2y : tail ? sy : vy) : vy;
        vid = apply ? (is1 ? (int32_t)__float_as_uint(pidf) : is2 ? id2 : tail ? sid : vid) : vid;
        nv = apply ? nv_new : nv;
        state = run ? (nan_d ? 1u : none_in ? 2u : grow ? 1u : state) : state;
    }
    if (state != 0u) {                               // leaves for good: nothing of this lipid has been written yet
        if (sub == 0u) {
            A.redo[i] = state == 1u ? 1 : 0;
            if (state == 2u) A.valid[i] = 0;
        }
        return;
    }
    if (sub == 0u) A.redo[i] = 0;
    // ---- direct neighbours (lib.rs:706-726): ring order, wall vertices (negative ids) left out
    {
        const bool is_nb = sub < nv && vid >= 0;
        const uint32_t nb_mask = (uint32_t)(__builtin_amdgcn_ballot_w64(is_nb) >> gl) & 0xFFFFu;
        if (is_nb) A.neib[slot + (uint32_t)__popc(nb_mask & ((1u << sub) - 1u))] = (uint64_t)vid;
        if ((uint32_t)__popc(nb_mask) < nv) {        // a wall vertex survived: open cell
            if (sub == 0u) A.valid[i] = 0;
            return;
        }
    }
    if (sub == 0u) {
        A.nvert[i] = nv;
#pragma unroll
        for (int k = 0; k < 6; ++k) A.coefs[6 * i + k] = cf[k];
        // compute_curvature_and_normal (lipid_molecule.rs:134-187)
        const float a = cf[0], b = cf[1], cq = cf[2], dd = cf[3], e = cf[4];
        const float E = 1.0f + dd * dd, F = dd * e, G = 1.0f + e * e;
        const float L = 2.0f * a, M = cq, N = 2.0f * b;
        const float Z = E * G - F * F;
        A.gauss[i] = (L * N - M * M) / Z;
        A.mean[i] = 0.5f * ((E * N - 2.0f * F * M) + G * L) / Z;
        const float glen = __builtin_sqrtf((dd * dd + e * e) + 1.0f);
        const V3 fn = mat_vec(to_lab, v3(dd / glen, e / glen, -1.0f / glen));
        A.normals[3 * i] = fn.x; A.normals[3 * i + 1] = fn.y; A.normals[3 * i + 2] = fn.z;
        float ev[2], evec[4];
        eig2_sym((E * L - F * M) / Z, (G * M - F * L) / Z, (G * N - F * M) / Z, ev, evec);
        A.pcurv[2 * i] = ev[0]; A.pcurv[2 * i + 1] = ev[1];
        for (int k = 0; k < 2; ++k) {
            const V3 pd = mat_vec(to_lab, v3(evec[2 * k], evec[2 * k + 1], 0.0f));
            A.pdirs[6 * i + 3 * k] = pd.x; A.pdirs[6 * i + 3 * k + 1] = pd.y; A.pdirs[6 * i + 3 * k + 2] = pd.z;
        }
    }
    {   // cell vertices on the fitted surface, lab frame, relative to the marker; fan area summed in ring order (lib.rs:731-752)
        V3 p = v3(0.f, 0.f, 0.f);
        if (sub < nv) {
            p = mat_vec(to_lab, v3(vx, vy, z_surf(vx, vy, cf)));
            float *dst = A.voro + 3 * (slot + sub);
            dst[0] = p.x; dst[1] = p.y; dst[2] = p.z;
        }
        const uint32_t prev = gl + (sub == 0u ? nv - 1u : sub - 1u);
        const V3 pp = v3(__shfl(p.x, (int)prev, 64), __shfl(p.y, (int)prev, 64), __shfl(p.z, (int)prev, 64));
        const float term = 0.5f * __builtin_sqrtf(norm2(cross(pp, p)));      // lane 0: the closing triangle (last vertex, first vertex)
        float ar = 0.0f;
        for (uint32_t k = 1; k < nv; ++k) ar += __shfl(term, (int)(gl + k), 64);
        ar += __shfl(term, (int)gl, 64);
        if (sub == 0u) A.area[i] = ar;
    }
    float *fp = A.fitted + 3 * p0;
    for (uint32_t q = sub; q < np; q += FIT_G) {     // fitted patch points (lib.rs:760-768)
        const float rx = pts[q][3], ry = pts[q][4], rz = pts[q][6];
        const uint32_t j = __float_as_uint(pts[q][7]);
        const V3 ss = v3(A.saved[3 * j], A.saved[3 * j + 1], A.saved[3 * j + 2]);
        const V3 t = mat_vec(to_lab, v3(0.0f, 0.0f, z_surf(rx, ry, cf) - rz));
        fp[3 * q] = ss.x + t.x;
        fp[3 * q + 1] = ss.y + t.y;
        fp[3 * q + 2] = ss.z + t.z;
    }
    if (sub != 0u) return;
    if (fabsf(cf[5]) > 0.5f) { A.valid[i] = 0; return; }   // fitted surface too far from the marker (lib.rs:774-777)
    const V3 t = mat_vec(to_lab, v3(0.0f, 0.0f, cf[5]));
    A.head[3 * i] += t.x; A.head[3 * i + 1] += t.y; A.head[3 * i + 2] += t.z;
}

// the fit kernel needs more LDS than a kernel gets by default
int launch_fit(molar_hip_ctx *c, const SmoothDev &A0) {
    SmoothDev A = A0;
    // sixteen lanes per lipid first; what it hands back (cells of more than 16 vertices, patches of more than 96 members) goes
    // through the one-lane kernel below
    A.redo = nullptr;
#ifndef MH_FIT_ONE_LANE          // (-DMH_FIT_ONE_LANE: the one-lane kernel alone, for A/B runs and bit-for-bit comparisons)
    MH_TRY(c->fit_redo.reserve(A.K));
    A.redo = c->fit_redo.as<uint8_t>();
    hipLaunchKernelGGL(k_membrane_fit_lanes, dim3((A.K + 64u / FIT_G - 1u) / (64u / FIT_G)), dim3(64), 0, c->stream, A);
#endif
    static bool ready[64] = {};          // per device: the attribute belongs to the device's copy of the kernel
    const int dev = c->device & 63;
    if (!ready[dev]) {
        MH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_membrane_fit), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FIT_LDS_BYTES));
        ready[dev] = true;
    }
    // A wave costs the same whether 16 or 64 of its lanes hold a lipid, and its time is that of its slowest lane: small
    // bilayers go 16 lipids to a workgroup (4000 lipids: 250 workgroups on 256 compute units instead of 63), large ones fill
    // the waves
    const uint32_t cus = (uint32_t)std::max(c->num_cus, 1);
    const uint32_t lanes = A.K <= 16u * 2u * cus ? 16u : (A.K <= 32u * 2u * cus ? 32u : 64u);
    hipLaunchKernelGGL(k_membrane_fit, dim3((A.K + lanes - 1u) / lanes), dim3(lanes), FIT_LDS_BYTES / 64u * lanes, c->stream, A);
    return 0;
}

// lib.rs:781-809.  `fitted_head` holds the markers after k_membrane_fit; the average is written to `head`.
// Sixteen lanes per lipid: a round gathers sixteen images of the marker at once (owner's flag, entry, point - three dependent
// loads that one lane would walk one after the other), then the group adds them in the order of the reference's scatter
// loop from shuffles, every lane holding the same sum.  Rounds are counted per wave, so the shuffles run with all lanes on.
constexpr uint32_t AVG_G = 16;
__global__ __launch_bounds__(64) void k_membrane_average(SmoothDev A, const float *fitted_head) {
    const uint32_t i = blockIdx.x * (64u / AVG_G) + threadIdx.x / AVG_G, sub = threadIdx.x % AVG_G;
    const bool live = i < A.K && A.valid[i];
    const uint32_t r0 = live ? A.rev_off[i] : 0u, r1 = live ? A.rev_off[i + 1] : 0u;
    uint32_t rounds = (r1 - r0 + AVG_G - 1u) / AVG_G;
    rounds = max(rounds, (uint32_t)__shfl_xor((int)rounds, 16));
    rounds = max(rounds, (uint32_t)__shfl_xor((int)rounds, 32));
    float n = 1.0f;
    V3 s = live ? v3(fitted_head[3 * i], fitted_head[3 * i + 1], fitted_head[3 * i + 2]) : v3(0.0f, 0.0f, 0.0f);
    for (uint32_t q = 0; q < rounds; ++q) {
        const uint32_t r = r0 + q * AVG_G + sub;
        bool ok = false;
        float px = 0.0f, py = 0.0f, pz = 0.0f;
        if (r < r1 && A.valid[A.rev_owner[r]]) {
            const float *p = A.fitted + 3ull * A.rev_entry[r];
            ok = true; px = p[0]; py = p[1]; pz = p[2];
        }
        const uint32_t okmask = (uint32_t)(__ballot(ok) >> (threadIdx.x / AVG_G * AVG_G)) & 0xffffu;
#pragma unroll
        for (uint32_t j = 0; j < AVG_G; ++j) {
            const float x = __shfl(px, (int)j, (int)AVG_G), y = __shfl(py, (int)j, (int)AVG_G), z = __shfl(pz, (int)j, (int)AVG_G);
            const bool take = (okmask >> j) & 1u;
            const V3 t = s + v3(x, y, z);
            n = take ? n + 1.0f : n;
            s.x = take ? t.x : s.x; s.y = take ? t.y : s.y; s.z = take ? t.z : s.z;
        }
    }
    if (!live) return;
    const V3 h = v3(s.x / n, s.y / n, s.z / n);
    if (sub == 0) { A.head[3 * i] = h.x; A.head[3 * i + 1] = h.y; A.head[3 * i + 2] = h.z; }
    const uint64_t slot = A.poff[i] + 4ull * i;
    const uint32_t nv = A.nvert[i];
    for (uint32_t k = sub; k < nv; k += AVG_G) {
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
                 o_rown = L.take(E * 4);
    const size_t up_bytes = L.size;
    const size_t o_saved = L.take(K * 12), o_fh = L.take(K * 12), o_vwork = L.take(slots * 16), o_pwork = L.take(E * 16 + 16);
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
    {
        Prof span(c, 4);
        MH_HIP(hipMemcpyAsync(d, h, up_bytes, hipMemcpyHostToDevice, c->stream));
        MH_HIP(hipMemcpyAsync(d + o_saved, d + o_head, K * 12, hipMemcpyDeviceToDevice, c->stream));
        SmoothDev A;
        A.K = (uint32_t)K;
        A.box = box;
        A.saved = (const float *)(d + o_saved);
        A.head = (float *)(d + o_head); A.normals = (float *)(d + o_norm); A.valid = (uint8_t *)(d + o_valid);
        A.poff = (const uint64_t *)(d + o_poff); A.pids = (const uint64_t *)(d + o_pids);
        A.coefs = (float *)(d + o_coefs); A.mean = (float *)(d + o_mean); A.gauss = (float *)(d + o_gauss);
        A.pcurv = (float *)(d + o_pcurv); A.pdirs = (float *)(d + o_pdirs); A.area = (float *)(d + o_area);
        A.nvert = (uint32_t *)(d + o_nvert); A.neib = (uint64_t *)(d + o_neib); A.voro = (float *)(d + o_voro);
        A.fitted = (float *)(d + o_fitted); A.vwork = (float4 *)(d + o_vwork); A.pwork = (float4 *)(d + o_pwork);
        A.rev_off = (const uint32_t *)(d + o_roff); A.rev_entry = (const uint32_t *)(d + o_rent);
        A.rev_owner = (const uint32_t *)(d + o_rown);
        MH_TRY(launch_fit(c, A));
        MH_HIP(hipMemcpyAsync(d + o_fh, d + o_head, K * 12, hipMemcpyDeviceToDevice, c->stream));
        hipLaunchKernelGGL(k_membrane_average, dim3((uint32_t)((K + 3) / 4)), dim3(64), 0, c->stream, A, (const float *)(d + o_fh));
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


// ================================================================ neighbour shells (host arithmetic)
//
// patches_from_nth_shell and smooth_curvature (molar_membrane/src/lib.rs:562-621) walk the Voronoi neighbour graph the
// smoothing pass left in neib_ids: per valid lipid the set of its direct neighbours, widened (n - 2) times by the
// neighbours of every member (so from n = 3 on a lipid is a member of its own set - the reference's HashSet picks it up on
// the way back, and so does this).  The reference iterates a HashSet, whose order is unspecified; here members are in
// ascending lipid id.  Graph bookkeeping over a few thousand short lists per frame: host loops, like the reference's.

namespace {

// members of lipid i's n-th shell, ascending; `stamp` (K entries, values < 2 * (i + 1) on entry) marks membership
void nth_shell_of(size_t i, size_t n_shells, const uint64_t *slot_off, const uint32_t *nvert, const uint64_t *neib, size_t K,
                  std::vector<uint32_t> &stamp, std::vector<uint32_t> &members, std::vector<uint32_t> &frontier) {
    const uint32_t mark = (uint32_t)i + 1u;
    members.clear();
    auto add_neighbours_of = [&](size_t l) {
        const uint64_t s0 = slot_off[l] + 4ull * l;
        // A member of the growing shell may be a lipid the smoothing pass dropped: its vertex count is whatever an earlier
        // frame (or the caller) left there, and check_shell_args vouches for the valid lipids only.  Never read past the
        // lipid's own slots (patch length + 4): the arrays carry no length across this ABI.
        const uint64_t cap = slot_off[l + 1] - slot_off[l] + 4ull;
        const uint32_t nv = nvert[l] < cap ? nvert[l] : (uint32_t)cap;
        for (uint32_t k = 0; k < nv; ++k) {
            const uint64_t id = neib[s0 + k];
            if (id < K && stamp[id] != mark) {
                stamp[id] = mark;
                members.push_back((uint32_t)id);
            }
        }
    };
    add_neighbours_of(i);
    for (size_t round = 2; round < n_shells; ++round) {
        frontier.assign(members.begin(), members.end());          // `old_neib_list`: the set as it was before this round
        for (uint32_t l : frontier) add_neighbours_of(l);
    }
    std::sort(members.begin(), members.end());
}

int check_shell_args(size_t K, const uint8_t *valid, const uint64_t *patch_offsets, const uint32_t *nvert, const uint64_t *neib_ids,
                     const char *what) {
    if (!valid || !patch_offsets || !nvert || !neib_ids) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "%s: null argument", what);
    if (K >= (1ull << 31)) return fail(MOLAR_HIP_ERR_TOO_LARGE, "%s: too many lipids", what);
    for (size_t i = 0; i < K; ++i) {
        if (patch_offsets[i + 1] < patch_offsets[i]) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "%s: offsets not monotone", what);
        if (valid[i] && nvert[i] > patch_offsets[i + 1] - patch_offsets[i] + 4u)
            return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "%s: lipid %zu has more vertices than its slots hold", what, i);
    }
    return 0;
}

}  // namespace

extern "C" int molar_hip_membrane_nth_shell_patches(size_t K, const uint8_t *valid, const uint64_t *patch_offsets, const uint64_t *patch_ids,
                                                    const uint32_t *nvert, const uint64_t *neib_ids, size_t n_shells,
                                                    uint64_t *out_offsets, uint64_t *out_ids, size_t capacity, size_t *needed) {
    MH_TRY(check_shell_args(K, valid, patch_offsets, nvert, neib_ids, "membrane_nth_shell_patches"));
    if (!out_offsets || !needed) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "membrane_nth_shell_patches: null output");
    if (n_shells < 1) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "membrane_nth_shell_patches: n_shells must be at least 1 (lib.rs:563)");
    if (patch_offsets[K] && !patch_ids) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "membrane_nth_shell_patches: patch_ids missing");
    std::vector<uint32_t> stamp(K, 0u), members, frontier;
    size_t total = 0;
    out_offsets[0] = 0;
    for (size_t i = 0; i < K; ++i) {
        if (valid[i]) {
            // zero-initialised stamps are never equal to a mark (marks are i + 1 >= 1), and marks differ from lipid to lipid
            nth_shell_of(i, n_shells, patch_offsets, nvert, neib_ids, K, stamp, members, frontier);
            if (out_ids && total + members.size() <= capacity)
                for (size_t k = 0; k < members.size(); ++k) out_ids[total + k] = members[k];
            total += members.size();
        } else {                                   // not touched by the reference loop: the lipid keeps the patch it has
            const size_t n = (size_t)(patch_offsets[i + 1] - patch_offsets[i]);
            if (out_ids && total + n <= capacity)
                for (size_t k = 0; k < n; ++k) out_ids[total + k] = patch_ids[patch_offsets[i] + k];
            total += n;
        }
        out_offsets[i + 1] = total;
    }
    *needed = total;
    return MOLAR_HIP_OK;
}

extern "C" int molar_hip_membrane_smooth_curvature(size_t K, const uint8_t *valid, const uint64_t *patch_offsets, const uint32_t *nvert,
                                                   const uint64_t *neib_ids, size_t n_shells, float *mean_curv, float *gauss_curv) {
    MH_TRY(check_shell_args(K, valid, patch_offsets, nvert, neib_ids, "membrane_smooth_curvature"));
    if (!mean_curv || !gauss_curv) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "membrane_smooth_curvature: null argument");
    if (n_shells < 1) return MOLAR_HIP_OK;                         // lib.rs:585-587
    const std::vector<float> mean(mean_curv, mean_curv + K), gauss(gauss_curv, gauss_curv + K);      // the values before smoothing (:589-590)
    std::vector<uint32_t> stamp(K, 0u), members, frontier;
    for (size_t i = 0; i < K; ++i) {
        if (!valid[i]) continue;
        nth_shell_of(i, n_shells, patch_offsets, nvert, neib_ids, K, stamp, members, frontier);
        float m = 0.0f, g = 0.0f;
        uint32_t n_valid = 0;
        for (uint32_t id : members) {
            if (!valid[id]) continue;
            m += mean[id];
            g += gauss[id];
            ++n_valid;
        }
        mean_curv[i] = (mean[i] + m) / (float)(n_valid + 1u);
        gauss_curv[i] = (gauss[i] + g) / (float)(n_valid + 1u);
    }
    return MOLAR_HIP_OK;
}

// ================================================================ one whole frame of Membrane::compute on the stream
//
// Everything between the coordinates of a frame and its per-lipid results, enqueued without a host wait: the number of
// pairs the marker search finds stays in device memory, the kernels behind it are launched over the capacity that
// earlier frames established and read the real sizes from there.  The host looks once, at the end of the frame.

namespace {

struct FrameInfo {                 // device block of a frame, read back with its results
    unsigned long long npairs;     // pairs of the marker search taken into the patches
    unsigned long long E;          // patch entries (2 * npairs)
    int overflow;                  // more pairs than the search buffers, or more entries than the patch arrays, hold
    int st_center, st_order;       // MOLAR_HIP_ERR_* raised by the marker / order kernels
    int changed;                   // the smoothing of this frame dropped a lipid (the valid flags it leaves differ from those it found)
};

// first kernel of a frame's B part: the valid flags as this frame finds them are remembered (or, on a repeat of B after
// its buffers were grown, put back), markers split into their arrays, the search input made
__device__ __forceinline__ float nrm3(float x, float y, float z) { return __builtin_sqrtf((x * x + y * y) + z * z); }

__global__ __launch_bounds__(256) void k_split_markers(uint32_t K, const float *__restrict__ mk, uint8_t *__restrict__ valid,
                                                       uint8_t *__restrict__ valid_prev, int restore, float *__restrict__ head,
                                                       float *__restrict__ mid, float *__restrict__ tail, float *__restrict__ head_search,
                                                       float *__restrict__ thv, float *__restrict__ nrm) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= K) return;
    uint8_t v;
    if (restore) { v = valid_prev[i]; valid[i] = v; }
    else { v = valid[i]; valid_prev[i] = v; }
    const bool ok = v != 0;
    float hd[3], tl[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float h = mk[9 * i + d];
        hd[d] = h;
        tl[d] = mk[9 * i + 6 + d];
        head[3 * i + d] = h;
        mid[3 * i + d] = mk[9 * i + 3 + d];
        tail[3 * i + d] = tl[d];
        // compute_patches searches among the valid lipids only (lib.rs:540-546).  A NaN position pairs with nothing and
        // leaves the order of every other pair alone, so the search keeps its fixed size and the ids stay lipid ids.
        head_search[3 * i + d] = ok ? h : __builtin_nanf("");
    }
    // compute_initial_normals starts from the unit tail-to-head vectors of the valid lipids (lib.rs:456-466); the first-pass
    // normals start from zero
    const float x = hd[0] - tl[0], y = hd[1] - tl[1], z = hd[2] - tl[2];
    const float n = nrm3(x, y, z);
    thv[3 * i] = ok ? x / n : 0.f; thv[3 * i + 1] = ok ? y / n : 0.f; thv[3 * i + 2] = ok ? z / n : 0.f;
    nrm[3 * i] = 0.f; nrm[3 * i + 1] = 0.f; nrm[3 * i + 2] = 0.f;
}

// (mask_units / mask_cap: a search whose hit history did not fit has left the results of its wrapped entries unwritten - stale
// pairs of whatever search used the buffer before, possibly of a larger system: ids beyond this one's arrays.  The list then
// counts as not there, like one that outgrew its buffer; the host grows the history and repeats the stage.)
__global__ __launch_bounds__(256) void k_patch_begin(const unsigned long long *__restrict__ total, unsigned long long cap_pairs,
                                                     unsigned long long cap_entries, const unsigned long long *__restrict__ mask_units,
                                                     unsigned long long mask_cap, FrameInfo *__restrict__ info,
                                                     uint32_t *__restrict__ deg, uint32_t ndeg) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < ndeg) deg[i] = 0u;                       // degrees and cursors of the patch build start from zero
    if (i != 0) return;
    const unsigned long long t = total ? *total : 0ull;
    const bool over = t > cap_pairs || 2ull * t > cap_entries || (mask_units && *mask_units > mask_cap);
    info->npairs = over ? 0ull : t;
    info->E = over ? 0ull : 2ull * t;
    info->overflow = over ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_patch_degree(const uint2 *__restrict__ pairs, const FrameInfo *__restrict__ info,
                                                      uint32_t *__restrict__ deg) {
    const unsigned long long p = (unsigned long long)blockIdx.x * 256u + threadIdx.x;
    if (p >= info->npairs) return;
    const uint2 ij = pairs[p];
    atomicAdd(&deg[ij.x], 1u);
    atomicAdd(&deg[ij.y], 1u);
}

// exclusive scan of deg[0 .. K) into poff[0 .. K] (one workgroup; K is a number of lipids)
__global__ __launch_bounds__(1024) void k_patch_scan(uint32_t K, const uint32_t *__restrict__ deg, uint64_t *__restrict__ poff,
                                                     uint32_t *__restrict__ roff) {
    __shared__ unsigned long long wsum[16];
    __shared__ unsigned long long carry_s;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0ull;
    __syncthreads();
    for (uint32_t base = 0; base < K + 1u; base += 1024u) {
        const uint32_t i = base + threadIdx.x;
        const unsigned long long v = i < K ? deg[i] : 0ull;
        unsigned long long x = v;
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long y = __shfl_up(x, o, 64);
            if ((int)lane >= o) x += y;
        }
        if (lane == 63u) wsum[wave] = x;
        __syncthreads();
        unsigned long long before = carry_s;
        for (uint32_t w = 0; w < wave; ++w) before += wsum[w];
        if (i < K + 1u) {
            poff[i] = before + x - v;
            roff[i] = (uint32_t)(before + x - v);
        }
        __syncthreads();
        if (threadIdx.x == 1023u) carry_s = before + x;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_patch_bucket(const uint2 *__restrict__ pairs, const FrameInfo *__restrict__ info,
                                                      const uint64_t *__restrict__ poff, uint32_t *__restrict__ cursor,
                                                      uint32_t *__restrict__ t_ord, uint32_t *__restrict__ t_oth,
                                                      uint32_t *__restrict__ t_grp) {
    const unsigned long long p = (unsigned long long)blockIdx.x * 256u + threadIdx.x;
    if (p >= info->npairs) return;
    const uint2 ij = pairs[p];
    const uint64_t a = poff[ij.x] + atomicAdd(&cursor[ij.x], 1u);
    t_ord[a] = (uint32_t)p; t_oth[a] = ij.y; t_grp[a] = ij.x;
    const uint64_t b = poff[ij.y] + atomicAdd(&cursor[ij.y], 1u);
    t_ord[b] = (uint32_t)p; t_oth[b] = ij.x; t_grp[b] = ij.y;
}

// patch_ids[i].push(j); patch_ids[j].push(i) in pair order (lib.rs:553-556): an entry's place inside its lipid's list is
// the number of entries of that list that come from earlier pairs (a pair feeds a list at most once: i != j)
__global__ __launch_bounds__(256) void k_patch_rank(const FrameInfo *__restrict__ info, const uint64_t *__restrict__ poff,
                                                    const uint32_t *__restrict__ t_ord, const uint32_t *__restrict__ t_oth,
                                                    const uint32_t *__restrict__ t_grp, uint64_t *__restrict__ pids,
                                                    uint32_t *__restrict__ pids32, uint32_t *__restrict__ owner) {
    const unsigned long long e = (unsigned long long)blockIdx.x * 256u + threadIdx.x;
    if (e >= info->E) return;
    const uint32_t g = t_grp[e], mine = t_ord[e];
    const uint64_t a = poff[g], b = poff[g + 1];
    uint64_t rank = 0;
    for (uint64_t q = a; q < b; ++q) rank += t_ord[q] < mine ? 1u : 0u;
    pids[a + rank] = t_oth[e];
    pids32[a + rank] = t_oth[e];          // what the host pass of the normals reads
    owner[a + rank] = g;
}

// Transpose of the patch lists for the marker averaging (k_membrane_average): for lipid t the entries q that hold t,
// ordered by q = by (owner lipid, position in the owner's list).  The lists are symmetric (a pair pushes both ways), so
// t's transposed list is as long as its own and the entries of owners below i are as many as the ids below i in it.
__global__ __launch_bounds__(256) void k_patch_reverse(const FrameInfo *__restrict__ info, const uint64_t *__restrict__ poff,
                                                       const uint64_t *__restrict__ pids, const uint32_t *__restrict__ owner,
                                                       uint32_t *__restrict__ rev_entry, uint32_t *__restrict__ rev_owner) {
    const unsigned long long q = (unsigned long long)blockIdx.x * 256u + threadIdx.x;
    if (q >= info->E) return;
    const uint32_t i = owner[q];
    const uint64_t t = pids[q];
    const uint64_t ta = poff[t], tb = poff[t + 1];
    uint64_t less = 0, same = 0;
    for (uint64_t e = ta; e < tb; ++e) {
        const uint64_t l = pids[e];
        less += l < i ? 1u : 0u;
        same += l == i ? 1u : 0u;
    }
    if (same > 1)          // the same pair more than once (boxes of fewer than three cells across): keep entry order
        for (uint64_t e = poff[i]; e < q; ++e) less += pids[e] == t ? 1u : 0u;
    rev_entry[ta + less] = (uint32_t)q;
    rev_owner[ta + less] = i;
}

// "angle <= FRAC_PI_2" of nalgebra's Vector::angle (lib.rs:472-473, 494) as a threshold on the cosine: `cos_min` is the
// smallest f32 whose acos rounds to at most pi/2 in the host's libm (found at plan creation), zero vectors give angle 0.
__device__ __forceinline__ bool within_half_pi(float ox, float oy, float oz, float no, float sx, float sy, float sz, float ns,
                                               float cos_min) {
    if (no == 0.0f || ns == 0.0f) return true;
    const float cc = ((ox * sx + oy * sy) + oz * sz) / (no * ns);
    return cc >= cos_min;
}

// One neighbour average of compute_initial_normals for lipid i by 16 lanes: sum, in patch order, of the patch members'
// vectors within 90 degrees of the lipid's own, plus its own, normalised.  `src` holds the vectors.
template <class F>
__device__ __forceinline__ void patch_average16(uint32_t i, uint32_t sub, uint64_t p0, uint32_t np, const uint64_t *__restrict__ pids,
                                                F load, float cos_min, float &rx, float &ry, float &rz) {
    float sx, sy, sz;
    load(i, sx, sy, sz);
    const float ns = nrm3(sx, sy, sz);
    float ax = 0.f, ay = 0.f, az = 0.f;
    for (uint32_t base = 0; base < np; base += 16u) {
        float ox = 0.f, oy = 0.f, oz = 0.f;
        if (base + sub < np) {
            const uint32_t l = (uint32_t)pids[p0 + base + sub];
            float x, y, z;
            load(l, x, y, z);
            if (within_half_pi(x, y, z, nrm3(x, y, z), sx, sy, sz, ns, cos_min)) { ox = x; oy = y; oz = z; }
        }
        // an unselected member adds +0.0: the running sum starts at +0.0 and x + 0.0 == x for every x it can hold
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            ax += __shfl(ox, k, 16);
            ay += __shfl(oy, k, 16);
            az += __shfl(oz, k, 16);
        }
    }
    ax += sx; ay += sy; az += sz;                      // .chain(once(central))
    const float n = nrm3(ax, ay, az);
    rx = ax / n; ry = ay / n; rz = az / n;
}

// first pass (lib.rs:464-484): every lipid from the tail->head vectors, independent of each other
__global__ __launch_bounds__(256) void k_normals_pass1(uint32_t K, const uint8_t *__restrict__ valid, const uint64_t *__restrict__ poff,
                                                       const uint64_t *__restrict__ pids, const float *__restrict__ thv,
                                                       float *__restrict__ nrm, float cos_min) {
    const uint32_t i = blockIdx.x * 16u + (threadIdx.x >> 4), sub = threadIdx.x & 15u;
    if (i >= K || !valid[i]) return;
    const uint64_t p0 = poff[i];
    float x, y, z;
    patch_average16(i, sub, p0, (uint32_t)(poff[i + 1] - p0), pids,
                    [&](uint32_t l, float &a, float &b, float &c) { a = thv[3 * l]; b = thv[3 * l + 1]; c = thv[3 * l + 2]; }, cos_min, x, y, z);
    if (sub == 0) { nrm[3 * i] = x; nrm[3 * i + 1] = y; nrm[3 * i + 2] = z; }
}

// Second pass (lib.rs:486-505): the reference loop overwrites normals in place while it walks the lipids in id order, so
// lipid i sees the new normals of its patch members below i and the old ones of those above: a chain through the lipid
// ids.  With ids laid out along a periodic lattice that chain is K/2 links long, and a link costs a GPU microseconds
// (a dependency-driven kernel, 16 lanes per lipid, normals and flags in LDS: bit-identical, 10.3 ms for 4000 lipids)
// where a CPU core takes 50 ns.  So this pass - and only this one - runs on the host, between two halves of the frame.
// Same arithmetic as molar_hip_membrane_initial_normals (measure.hip).
void normals_pass2_host(size_t K, const uint32_t *poff, const uint32_t *pids, const uint8_t *valid, const float *n1, float *nv,
                        std::vector<float> &len) {
    const float half_pi = 1.57079632679489661923f;
    auto nrm = [](const float *a) { return std::sqrt((a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]); };
    len.resize(K);
    std::memcpy(nv, n1, K * 12);
    for (size_t i = 0; i < K; ++i) len[i] = nrm(nv + 3 * i);
    for (size_t i = 0; i < K; ++i) {
        if (!valid[i]) continue;
        const float sx = nv[3 * i], sy = nv[3 * i + 1], sz = nv[3 * i + 2], ns = len[i];
        float ax = 0.f, ay = 0.f, az = 0.f;
        for (uint32_t q = poff[i]; q < poff[i + 1]; ++q) {
            const uint32_t l = pids[q];
            const float *o = nv + 3 * l;
            const float no = len[l];
            bool in = true;                                    // Vector::angle is 0 for a zero vector
            // |dot| > 2e-6 |o||s| decides without the division: the quotient is then beyond the +-1e-6 band whatever its
            // rounding (the loop is a chain of dependent additions, 4 cycles a member; the division was a sixth of its time)
            const float dot = (o[0] * sx + o[1] * sy) + o[2] * sz, pn = no * ns, band = 2.0e-6f * pn;
            if (dot > band) in = true;
            else if (dot < -band && band > 0.0f) in = false;   // (a band that underflowed to zero decides nothing on this side)
            else if (no != 0.0f && ns != 0.0f) {
                float cc = dot / pn;
                if (cc > 1.0e-6f) in = true;
                else if (cc < -1.0e-6f) in = false;
                else {
                    cc = cc < -1.0f ? -1.0f : (cc > 1.0f ? 1.0f : cc);
                    in = std::acos(cc) <= half_pi;
                }
            }
            if (in) { ax += o[0]; ay += o[1]; az += o[2]; }
        }
        ax += sx; ay += sy; az += sz;
        const float n = std::sqrt((ax * ax + ay * ay) + az * az);
        nv[3 * i] = ax / n; nv[3 * i + 1] = ay / n; nv[3 * i + 2] = az / n;
        len[i] = nrm(nv + 3 * i);
    }
}

// fresh per-lipid state of a frame: the LipidMolecule defaults of Membrane::new (lib.rs:152-177; the rest is zeroed by a
// memset), the working copy of the head markers and the initial normals
__global__ __launch_bounds__(256) void k_state_defaults(uint32_t K, float *__restrict__ mean, float *__restrict__ gauss,
                                                        const float *__restrict__ head, const float *__restrict__ n0_host,
                                                        float *__restrict__ n0, float *__restrict__ s_head, float *__restrict__ s_normals) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= K) return;
    mean[i] = -100.0f;
    gauss[i] = -100.0f;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        s_head[3 * i + d] = head[3 * i + d];
        const float v = n0_host[3 * i + d];          // the host pass's normals, read from its pinned block (no copy ahead)
        n0[3 * i + d] = v;
        s_normals[3 * i + d] = v;
    }
}

// the flags a frame leaves, for its results, and whether its smoothing changed any; and the normal each tail is measured
// against (its lipid's, or the global one) - one launch over max(lipids, tails)
__global__ __launch_bounds__(256) void k_flags_tail_normals(uint32_t K, const uint8_t *__restrict__ before, const uint8_t *__restrict__ after,
                                                            uint8_t *__restrict__ out, int *__restrict__ changed, uint32_t ntails,
                                                            const uint32_t *__restrict__ tail_lipid, const float *__restrict__ normals,
                                                            int use_global, float gx, float gy, float gz, float *__restrict__ tnorm) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < K) {
        out[i] = after[i];
        if (before[i] != after[i]) *changed = 1;
    }
    if (i < ntails) {
        const uint32_t l = tail_lipid[i];
        tnorm[3 * i] = use_global ? gx : normals[3 * l];
        tnorm[3 * i + 1] = use_global ? gy : normals[3 * l + 1];
        tnorm[3 * i + 2] = use_global ? gz : normals[3 * l + 2];
    }
}

// smallest f32 c with acosf(c) <= pi/2 (f32): the comparison compute_initial_normals makes, decided by the host's libm
// like MolAR's f32::acos; acos is monotone, the boundary sits a fraction of an ulp of pi/2 below zero
float half_pi_cos_threshold() {
    const float half_pi = 1.57079632679489661923f;
    auto ok = [&](float c) { return std::acos(c) <= half_pi; };
    float lo = -1.0e-6f, hi = 0.0f;      // !ok(lo), ok(hi)
    if (ok(lo) || !ok(hi)) return 0.0f;  // not a libm this code knows: the plain sign test
    while (std::nextafter(lo, hi) != hi) {
        // bisect on the bit patterns (negative floats: larger pattern = smaller value)
        uint32_t a, b;
        std::memcpy(&a, &lo, 4);
        std::memcpy(&b, &hi, 4);
        if (hi == 0.0f) b = 0x80000000u;
        const uint32_t m = b + (a - b) / 2u;
        float mid;
        std::memcpy(&mid, &m, 4);
        if (mid == lo || mid == hi) break;
        if (ok(mid)) hi = mid; else lo = mid;
    }
    return hi;
}

struct Blob2 {
    size_t size = 0;
    size_t take(size_t bytes) {
        const size_t at = size;
        size += (bytes + 255) & ~size_t(255);
        return at;
    }
};

struct FrameLayout {       // byte offsets inside a frame slot's device blob, for K lipids and room for Ecap patch entries
    size_t info, roff, normals0, valid_prev, pids32;       // (one block, see frame_layout)
    size_t mk, head, mid, tail, head_search, valid_out, thv, poff, pids, owner, rev_entry, rev_owner;
    size_t zero_begin, s_head, s_normals, coefs, pcurv, pdirs, area, nvert, neib, voro, fitted, zero_end, mean, gauss;
    size_t saved, vwork, pwork, tnorm, order, bytes;
};

FrameLayout frame_layout(size_t K, size_t Ecap, size_t ntails, size_t norder) {
    FrameLayout L{};
    Blob2 B;
    const size_t slots = Ecap + 4 * K;
    // [info | roff | normals0 | valid_prev | pids32]: what the host pass of the normals reads, brought over in one copy
    L.info = B.take(sizeof(FrameInfo));
    L.roff = B.take((K + 1) * 4); L.normals0 = B.take(K * 12); L.valid_prev = B.take(K); L.pids32 = B.take(Ecap * 4);
    L.mk = B.take(K * 36); L.head = B.take(K * 12); L.mid = B.take(K * 12); L.tail = B.take(K * 12); L.head_search = B.take(K * 12);
    L.valid_out = B.take(K);
    L.thv = B.take(K * 12);
    L.poff = B.take((K + 1) * 8);
    L.pids = B.take(Ecap * 8); L.owner = B.take(Ecap * 4); L.rev_entry = B.take(Ecap * 4);
    L.rev_owner = B.take(Ecap * 4);
    L.s_head = B.take(K * 12); L.s_normals = B.take(K * 12);
    L.zero_begin = B.size;
    L.coefs = B.take(K * 24); L.pcurv = B.take(K * 8); L.pdirs = B.take(K * 24); L.area = B.take(K * 4); L.nvert = B.take(K * 4);
    L.neib = B.take(slots * 8); L.voro = B.take(slots * 12); L.fitted = B.take(Ecap * 12);
    L.zero_end = B.size;
    L.mean = B.take(K * 4); L.gauss = B.take(K * 4);
    L.saved = B.take(K * 12); L.vwork = B.take(slots * 16); L.pwork = B.take(Ecap * 16 + 16);
    L.tnorm = B.take(ntails * 12); L.order = B.take(norder * 4 + 16);
    L.bytes = B.size;
    return L;
}

}  // namespace

// A frame runs in three pieces on the context's stream:
//   A  unwrap, markers                                     (independent of every other frame)
//   B  marker search, patches, first normals pass          (needs the valid flags the frame before it leaves)
//   -  second normals pass on the host                     (normals_pass2_host)
//   C  smoothing, order                                    (leaves the valid flags for the next frame)
// _begin enqueues A and B; _end runs the host pass (unless done), enqueues C, and waits for it.  With two frames in
// flight the younger frame's B is on the stream ahead of the older frame's C, and its host pass runs while the GPU is
// busy with that C; it has then seen the flags of the frame before, and is repeated if C changed them (FrameInfo::changed).
struct molar_hip_membrane_plan {
    molar_hip_ctx *c = nullptr;
    size_t K = 0, natoms = 0, ntails = 0, nidx_lipid = 0, nidx_marker = 0, nidx_tail = 0, norder = 0;
    float cutoff = 0.f;
    int order_type = 0, max_iter = 1, unwrap = 0, use_global = 0;
    float gn[3] = {0, 0, 1};
    float cos_min = 0.f;
    // constants of the trajectory, device
    DevBuf consts;
    const uint64_t *lipid_idx = nullptr, *lipid_off = nullptr, *marker_idx = nullptr, *marker_off = nullptr, *tail_idx = nullptr,
                   *tail_off = nullptr, *noff = nullptr;
    const float *masses = nullptr;
    const uint32_t *tail_lipid = nullptr;
    const uint8_t *tail_bonds = nullptr;
    // carried from frame to frame, device
    DevBuf valid;                 // [K]
    DevBuf work;                  // deg[K+1] | cursor[K] | t_ord / t_oth / t_grp [Ecap]
    size_t Ecap = 0;              // patch entries the slot blobs and `work` are laid out for
    hipStream_t copy_stream = nullptr;
    void *h_fetch = nullptr;      // pinned staging of molar_hip_membrane_frame_fetch
    size_t h_fetch_cap = 0;
    std::vector<float> len_scratch;
    struct Slot {
        DevBuf blob;
        DevBuf xyz_stage;         // a frame handed over in host memory is staged here
        FrameLayout lay{};
        size_t Ecap = 0;
        void *h = nullptr;        // pinned: 16 bytes of search sizes | FrameInfo at the end of the frame
        void *h_mid = nullptr;    // pinned: what the host pass reads (roff | n1 | valid_prev | pids32) and writes (n2)
        size_t h_mid_cap = 0;
        hipEvent_t mid = nullptr, done = nullptr;
        hipEvent_t back = nullptr;   // the unwrapped frame is back in the caller's host array (frames handed over in host memory)
        bool pending = false, ended = false, b_enqueued = false;
        bool speculative = false;  // B ran on the flags of the frame before the older one, ahead of the older frame's C
        bool passed = false;       // the host pass is done for the B that is on the stream
        unsigned long long serial = 0;
        ResidentLaunch L;
        unsigned long long cap_pairs = 0;
        float *xyz_dev = nullptr, *xyz_host = nullptr;
        float box9[9] = {};
        FrameInfo info{};
    } slot[2];
    int next = 0;
    unsigned long long serial = 0;
};

namespace {

constexpr size_t H_SIZES = 0, H_INFO = 64, H_BYTES = 128;

// the slot's pinned block mirrors the head of the device blob [info .. pids32 + Ecap] at the same offsets; the normals
// the host pass produces follow
size_t mid_bytes(const FrameLayout &L, size_t Ecap) { return L.pids32 + Ecap * 4; }
size_t mid_n2(const FrameLayout &L, size_t Ecap) { return (mid_bytes(L, Ecap) + 255) & ~size_t(255); }

int ensure_capacity(molar_hip_membrane_plan *P, molar_hip_membrane_plan::Slot &S) {
    if (S.Ecap == P->Ecap && S.blob.p) return 0;
    S.lay = frame_layout(P->K, P->Ecap, P->ntails, P->norder);
    MH_TRY(S.blob.reserve(S.lay.bytes));
    S.Ecap = P->Ecap;
    const size_t need = mid_n2(S.lay, P->Ecap) + P->K * 12;
    if (need > S.h_mid_cap) {
        if (S.h_mid) (void)hipHostFree(S.h_mid);
        S.h_mid = nullptr;
        S.h_mid_cap = 0;
        MH_HIP(hipHostMalloc(&S.h_mid, need + need / 8, hipHostMallocDefault));
        S.h_mid_cap = need + need / 8;
    }
    return 0;
}

int ensure_work(molar_hip_membrane_plan *P) {
    const size_t K = P->K;
    return P->work.reserve(((K + 1) + K) * 4 + 3 * P->Ecap * 4 + 1024);
}

// A: unwrap and markers
int enqueue_a(molar_hip_membrane_plan *P, molar_hip_membrane_plan::Slot &S) {
    molar_hip_ctx *c = P->c;
    MH_TRY(ensure_capacity(P, S));
    char *d = S.blob.as<char>();
    const FrameLayout &L = S.lay;
    hipStream_t st = c->stream;
    molar_hip_box box;
    MH_TRY(molar_hip_box_from_matrix(S.box9, &box));
    MH_HIP(hipMemsetAsync(d + L.info, 0, sizeof(FrameInfo), st));
    FrameInfo *info = reinterpret_cast<FrameInfo *>(d + L.info);
    if (P->unwrap) {
        MH_TRY(enqueue_unwrap_batch(c, S.xyz_dev, P->lipid_idx, P->lipid_off, (uint32_t)P->K, box, MOLAR_HIP_PBC_FULL));
        if (S.xyz_host) {    // the caller's frame is unwrapped in place, like Modify::unwrap_simple on the System
            MH_HIP(hipMemcpyAsync(S.xyz_host, S.xyz_dev, P->natoms * 12, hipMemcpyDeviceToHost, st));
            if (!S.back) MH_HIP(hipEventCreateWithFlags(&S.back, hipEventDisableTiming));
            MH_HIP(hipEventRecord(S.back, st));
        }
    }
    MH_TRY(enqueue_center_batch(c, S.xyz_dev, P->marker_idx, P->marker_off, (uint32_t)(3 * P->K), P->masses, (float *)(d + L.mk),
                                &info->st_center));
    return 0;
}

// B: marker search, patch lists, tail->head vectors and the first normals pass; what the host pass needs goes to the
// slot's pinned block.  `restore`: a repeat after the buffers were grown - the valid flags go back to what they were.
int enqueue_b(molar_hip_membrane_plan *P, molar_hip_membrane_plan::Slot &S, bool restore) {
    molar_hip_ctx *c = P->c;
    const size_t K = P->K;
    const uint32_t K32 = (uint32_t)K;
    if (S.Ecap != P->Ecap) {
        // the layout changes with the capacity: carry the results of A (markers, status) and the remembered flags over
        const FrameLayout old = S.lay;
        DevBuf keep;
        MH_TRY(keep.reserve(K * 37 + sizeof(FrameInfo)));
        char *k = keep.as<char>(), *d0 = S.blob.as<char>();
        MH_HIP(hipStreamSynchronize(c->stream));
        MH_HIP(hipMemcpy(k, d0 + old.mk, K * 36, hipMemcpyDeviceToDevice));
        MH_HIP(hipMemcpy(k + K * 36, d0 + old.valid_prev, K, hipMemcpyDeviceToDevice));
        MH_HIP(hipMemcpy(k + K * 37, d0 + old.info, sizeof(FrameInfo), hipMemcpyDeviceToDevice));
        MH_TRY(ensure_capacity(P, S));
        char *d1 = S.blob.as<char>();
        MH_HIP(hipMemcpy(d1 + S.lay.mk, k, K * 36, hipMemcpyDeviceToDevice));
        MH_HIP(hipMemcpy(d1 + S.lay.valid_prev, k + K * 36, K, hipMemcpyDeviceToDevice));
        MH_HIP(hipMemcpy(d1 + S.lay.info, k + K * 37, sizeof(FrameInfo), hipMemcpyDeviceToDevice));
        keep.release();
    }
    MH_TRY(ensure_work(P));
    char *d = S.blob.as<char>();
    const FrameLayout &L = S.lay;
    hipStream_t st = c->stream;
    uint8_t *valid = P->valid.as<uint8_t>();
    FrameInfo *info = reinterpret_cast<FrameInfo *>(d + L.info);
    float *head = (float *)(d + L.head), *tail = (float *)(d + L.tail);
    const uint32_t nbK = (K32 + 255u) / 256u;
    hipLaunchKernelGGL(k_split_markers, dim3(nbK), dim3(256), 0, st, K32, (const float *)(d + L.mk), valid, (uint8_t *)(d + L.valid_prev),
                       restore ? 1 : 0, head, (float *)(d + L.mid), tail, (float *)(d + L.head_search), (float *)(d + L.thv),
                       (float *)(d + L.normals0));
    // ---- compute_patches (lib.rs:539-558)
    molar_hip_search_desc q{};
    q.kind = MOLAR_HIP_SEARCH_SINGLE;
    q.cutoff = P->cutoff;
    q.xyz1 = (const float *)(d + L.head_search);
    q.natoms1 = K;
    q.idx1 = nullptr;
    q.n1 = K;
    q.ids_local = 1;
    q.box9 = S.box9;
    q.pbc = MOLAR_HIP_PBC_FULL;
    const unsigned long long *total_dev = nullptr;
    const uint32_t *pairs_dev = nullptr;
    MH_TRY(search_resident_enqueue(c, &q, (char *)S.h + H_SIZES, &S.L, &total_dev, &pairs_dev));
    S.cap_pairs = S.L.cap0;
    const size_t Ecap = S.Ecap;
    uint32_t *deg = P->work.as<uint32_t>(), *cursor = deg + (K + 1), *t_ord = cursor + K, *t_oth = t_ord + Ecap, *t_grp = t_oth + Ecap;
    const uint32_t ndeg = (uint32_t)((K + 1) + K);
    hipLaunchKernelGGL(k_patch_begin, dim3((ndeg + 255u) / 256u), dim3(256), 0, st, total_dev, S.cap_pairs, (unsigned long long)Ecap,
                       S.L.mask_units_dev, S.L.maskcap0, info, deg, ndeg);
    uint64_t *poff = (uint64_t *)(d + L.poff), *pids = (uint64_t *)(d + L.pids);
    uint32_t *roff = (uint32_t *)(d + L.roff), *pids32 = (uint32_t *)(d + L.pids32), *owner = (uint32_t *)(d + L.owner),
             *rev_entry = (uint32_t *)(d + L.rev_entry), *rev_owner = (uint32_t *)(d + L.rev_owner);
    const size_t pair_room = std::min<size_t>((size_t)S.cap_pairs, Ecap / 2);
    const uint32_t nbP = (uint32_t)((pair_room + 255) / 256), nbE = (uint32_t)((2 * pair_room + 255) / 256);
    const uint2 *pairs = reinterpret_cast<const uint2 *>(pairs_dev);
    if (nbP) hipLaunchKernelGGL(k_patch_degree, dim3(nbP), dim3(256), 0, st, pairs, info, deg);
    hipLaunchKernelGGL(k_patch_scan, dim3(1), dim3(1024), 0, st, K32, deg, poff, roff);
    if (nbP) {
        hipLaunchKernelGGL(k_patch_bucket, dim3(nbP), dim3(256), 0, st, pairs, info, poff, cursor, t_ord, t_oth, t_grp);
        hipLaunchKernelGGL(k_patch_rank, dim3(nbE), dim3(256), 0, st, info, poff, t_ord, t_oth, t_grp, pids, pids32, owner);
        hipLaunchKernelGGL(k_patch_reverse, dim3(nbE), dim3(256), 0, st, info, poff, pids, owner, rev_entry, rev_owner);
    }
    // ---- compute_initial_normals, first pass (lib.rs:456-484)
    float *thv = (float *)(d + L.thv), *n0 = (float *)(d + L.normals0);
    hipLaunchKernelGGL(k_normals_pass1, dim3((K32 + 15u) / 16u), dim3(256), 0, st, K32, valid, poff, pids, thv, n0, P->cos_min);
    MH_HIP(hipGetLastError());
    // ---- to the host, in one copy: status, offsets, first-pass normals, flags, ids
    MH_HIP(hipMemcpyAsync((char *)S.h_mid + L.info, d + L.info, L.pids32 - L.info + std::min(Ecap, 2 * pair_room) * 4, hipMemcpyDeviceToHost, st));
    MH_HIP(hipEventRecord(S.mid, st));
    // the smoothing's per-lipid state starts from zero: cleared here, while the host makes its pass over the normals
    MH_HIP(hipMemsetAsync(d + L.zero_begin, 0, L.zero_end - L.zero_begin, st));
    S.b_enqueued = true;
    S.passed = false;
    return 0;
}

// the host in the middle: wait for B, make sure it fitted (else grow and repeat it - or, with `may_repeat` off, leave
// the frame as it is for a later call), second normals pass
int host_pass(molar_hip_membrane_plan *P, molar_hip_membrane_plan::Slot &S, bool may_repeat) {
    molar_hip_ctx *c = P->c;
    for (int attempt = 0;; ++attempt) {
        MH_HIP(hipEventSynchronize(S.mid));
        std::memcpy(&S.info, (char *)S.h_mid + S.lay.info, sizeof(FrameInfo));
        bool fits = true;
        MH_TRY(search_resident_fits(c, (char *)S.h + H_SIZES, S.L, &fits));
        unsigned long long sizes[2];
        std::memcpy(sizes, (char *)S.h + H_SIZES, 16);
        if (2ull * sizes[0] >= (1ull << 32)) return fail(MOLAR_HIP_ERR_TOO_LARGE, "membrane frame: %llu patch entries (32-bit entry indices)", 2ull * sizes[0]);
        if (2ull * sizes[0] > P->Ecap) {
            P->Ecap = (size_t)(2ull * sizes[0] + sizes[0] / 4u + 1024u);
            fits = false;
        }
        if (fits && !S.info.overflow) break;
        if (!may_repeat) return 0;
        if (attempt >= 3) return fail(MOLAR_HIP_ERR_HIP, "membrane frame: buffers did not settle");
        MH_TRY(enqueue_b(P, S, /*restore=*/true));
    }
    const size_t K = P->K;
    const FrameLayout &L = S.lay;
    char *hm = (char *)S.h_mid;
    normals_pass2_host(K, (const uint32_t *)(hm + L.roff), (const uint32_t *)(hm + L.pids32), (const uint8_t *)(hm + L.valid_prev),
                       (const float *)(hm + L.normals0), (float *)(hm + mid_n2(L, S.Ecap)), P->len_scratch);
    S.passed = true;
    return 0;
}

// C: smoothing on a fresh per-lipid state, order
int enqueue_c(molar_hip_membrane_plan *P, molar_hip_membrane_plan::Slot &S, bool info_to_host = true) {
    molar_hip_ctx *c = P->c;
    const size_t K = P->K;
    const uint32_t K32 = (uint32_t)K;
    char *d = S.blob.as<char>();
    const FrameLayout &L = S.lay;
    hipStream_t st = c->stream;
    uint8_t *valid = P->valid.as<uint8_t>();
    FrameInfo *info = reinterpret_cast<FrameInfo *>(d + L.info);
    float *n0 = (float *)(d + L.normals0);
    const uint32_t nbK = (K32 + 255u) / 256u;
    // ---- smooth (lib.rs:661-812); the per-lipid state was zeroed at the end of B
    hipLaunchKernelGGL(k_state_defaults, dim3(nbK), dim3(256), 0, st, K32, (float *)(d + L.mean), (float *)(d + L.gauss),
                       (const float *)(d + L.head), (const float *)((char *)S.h_mid + mid_n2(L, S.Ecap)), n0, (float *)(d + L.s_head),
                       (float *)(d + L.s_normals));
    SmoothDev A;
    A.K = K32;
    MH_TRY(molar_hip_box_from_matrix(S.box9, &A.box));
    A.head = (float *)(d + L.s_head); A.normals = (float *)(d + L.s_normals); A.valid = valid;
    A.poff = (const uint64_t *)(d + L.poff); A.pids = (const uint64_t *)(d + L.pids);
    A.coefs = (float *)(d + L.coefs); A.mean = (float *)(d + L.mean); A.gauss = (float *)(d + L.gauss);
    A.pcurv = (float *)(d + L.pcurv); A.pdirs = (float *)(d + L.pdirs); A.area = (float *)(d + L.area);
    A.nvert = (uint32_t *)(d + L.nvert); A.neib = (uint64_t *)(d + L.neib); A.voro = (float *)(d + L.voro);
    A.fitted = (float *)(d + L.fitted); A.vwork = (float4 *)(d + L.vwork); A.pwork = (float4 *)(d + L.pwork);
    A.rev_off = (const uint32_t *)(d + L.roff); A.rev_entry = (const uint32_t *)(d + L.rev_entry);
    A.rev_owner = (const uint32_t *)(d + L.rev_owner);
    for (int it = 0; it < P->max_iter; ++it) {
        // the markers before the iteration: the frame's own for the first one (the working copy starts as their image)
        if (it == 0) A.saved = (const float *)(d + L.head);
        else {
            MH_HIP(hipMemcpyAsync(d + L.saved, d + L.s_head, K * 12, hipMemcpyDeviceToDevice, st));
            A.saved = (const float *)(d + L.saved);
        }
        MH_TRY(launch_fit(c, A));
        // (a lane of the averaging kernel reads the fitted marker of its own lipid only, before it overwrites it)
        hipLaunchKernelGGL(k_membrane_average, dim3((uint32_t)((K + 3) / 4)), dim3(64), 0, st, A, (const float *)A.head);
    }
    {
        const uint32_t nt = (uint32_t)P->ntails, nmax = std::max(K32, nt);
        hipLaunchKernelGGL(k_flags_tail_normals, dim3((nmax + 255u) / 256u), dim3(256), 0, st, K32, (const uint8_t *)(d + L.valid_prev), valid,
                           (uint8_t *)(d + L.valid_out), &info->changed, nt, P->tail_lipid, (const float *)(d + L.s_normals), P->use_global,
                           P->gn[0], P->gn[1], P->gn[2], (float *)(d + L.tnorm));
    }
    // ---- compute_order (lib.rs:435-443)
    if (P->ntails) {
        MH_TRY(enqueue_lipid_order(c, S.xyz_dev, P->tail_idx, P->tail_off, (uint32_t)P->ntails, P->order_type, (const float *)(d + L.tnorm),
                                   P->noff, P->tail_bonds, (float *)(d + L.order), &info->st_order));
    }
    MH_HIP(hipGetLastError());
    if (info_to_host) {
        MH_HIP(hipMemcpyAsync((char *)S.h + H_INFO, info, sizeof(FrameInfo), hipMemcpyDeviceToHost, st));
        MH_HIP(hipEventRecord(S.done, st));
    }
    return 0;
}

int check_ticket(molar_hip_membrane_plan *P, int32_t t, bool want_ended) {
    if (!P) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "membrane frame: null plan");
    if (t < 0 || t > 1) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "membrane frame: unknown ticket %d", (int)t);
    if (want_ended && !P->slot[t].ended) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "membrane frame: ticket %d holds no ended frame", (int)t);
    if (!want_ended && !P->slot[t].pending) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "membrane frame: ticket %d is not in flight", (int)t);
    return 0;
}

}  // namespace

extern "C" int molar_hip_membrane_plan_create(molar_hip_ctx *c, const molar_hip_membrane_desc *D, molar_hip_membrane_plan **out) {
    if (!c || !D || !out) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "membrane_plan_create: null argument");
    *out = nullptr;
    MH_HIP(hipSetDevice(c->device));
    const size_t K = D->nlipids;
    if (K == 0) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "membrane_plan_create: no lipids");
    if (K >= (1ull << 30)) return fail(MOLAR_HIP_ERR_TOO_LARGE, "membrane_plan_create: %zu lipids (ids must fit i32, voronoi_cell.rs:17; three markers each are counted in 32 bits)", K);
    if (!D->lipid_idx || !D->lipid_offsets || !D->marker_idx || !D->marker_offsets || !D->masses)
        return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "membrane_plan_create: null index list");
    if (D->ntails && (!D->tail_idx || !D->tail_offsets || !D->tail_lipid))
        return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "membrane_plan_create: null tail list");
    if (D->ntails >= 0xFFFFFFFFull) return fail(MOLAR_HIP_ERR_TOO_LARGE, "membrane_plan_create: too many tails");
    if (D->ntails && !D->tail_bonds && D->order_type != 0) return fail(MOLAR_HIP_ERR_LIPID_BOND_ORDER_COUNT, "bond orders missing");
    if (D->order_type < 0 || D->order_type > 2) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "membrane_plan_create: order_type %d", (int)D->order_type);
    if (D->max_smooth_iter < 1) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "membrane_plan_create: max_smooth_iter %d", (int)D->max_smooth_iter);
    if (!(D->cutoff > 0.0f)) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "membrane_plan_create: cutoff %g", (double)D->cutoff);
    auto check_csr = [&](const uint64_t *idx, const uint64_t *off, size_t n, const char *what) {
        if (off[0] != 0) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "membrane_plan_create: %s offsets do not start at 0", what);
        for (size_t k = 0; k < n; ++k)
            if (off[k + 1] < off[k]) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "membrane_plan_create: %s offsets not monotone", what);
        for (uint64_t q = 0; q < off[n]; ++q)
            if (idx[q] >= D->natoms) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "membrane_plan_create: %s index %llu out of range", what,
                                                 (unsigned long long)idx[q]);
        return 0;
    };
    MH_TRY(check_csr(D->lipid_idx, D->lipid_offsets, K, "lipid"));
    MH_TRY(check_csr(D->marker_idx, D->marker_offsets, 3 * K, "marker"));
    if (D->ntails) {
        MH_TRY(check_csr(D->tail_idx, D->tail_offsets, D->ntails, "tail"));
        for (size_t t = 0; t < D->ntails; ++t) {
            if (D->tail_lipid[t] >= K) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "membrane_plan_create: tail %zu belongs to no lipid", t);
            if (D->tail_offsets[t + 1] - D->tail_offsets[t] < 3) return fail(MOLAR_HIP_ERR_LIPID_TAIL_TOO_SHORT, "tail should have at least 3 carbons");
        }
    }
    auto *P = new molar_hip_membrane_plan();
    P->c = c;
    P->K = K; P->natoms = D->natoms; P->ntails = D->ntails;
    P->nidx_lipid = (size_t)D->lipid_offsets[K]; P->nidx_marker = (size_t)D->marker_offsets[3 * K];
    P->nidx_tail = D->ntails ? (size_t)D->tail_offsets[D->ntails] : 0;
    P->norder = P->nidx_tail - 2 * D->ntails;
    P->cutoff = D->cutoff; P->order_type = D->order_type; P->max_iter = D->max_smooth_iter; P->unwrap = D->unwrap;
    P->use_global = D->use_global_normal;
    std::memcpy(P->gn, D->global_normal, 12);
    P->cos_min = half_pi_cos_threshold();
    int rc = [&]() -> int {
        Blob2 B;
        const size_t o_li = B.take(P->nidx_lipid * 8), o_lo = B.take((K + 1) * 8), o_mi = B.take(P->nidx_marker * 8),
                     o_mo = B.take((3 * K + 1) * 8), o_ms = B.take(D->natoms * 4), o_ti = B.take(P->nidx_tail * 8),
                     o_to = B.take((D->ntails + 1) * 8), o_no = B.take((D->ntails + 1) * 8), o_tl = B.take(D->ntails * 4),
                     o_tb = B.take(P->nidx_tail + 16);
        MH_TRY(P->consts.reserve(B.size));
        std::vector<char> h(B.size, 0);
        std::memcpy(h.data() + o_li, D->lipid_idx, P->nidx_lipid * 8);
        std::memcpy(h.data() + o_lo, D->lipid_offsets, (K + 1) * 8);
        std::memcpy(h.data() + o_mi, D->marker_idx, P->nidx_marker * 8);
        std::memcpy(h.data() + o_mo, D->marker_offsets, (3 * K + 1) * 8);
        std::memcpy(h.data() + o_ms, D->masses, D->natoms * 4);
        if (D->ntails) {
            std::memcpy(h.data() + o_ti, D->tail_idx, P->nidx_tail * 8);
            std::memcpy(h.data() + o_to, D->tail_offsets, (D->ntails + 1) * 8);
            std::memcpy(h.data() + o_tl, D->tail_lipid, D->ntails * 4);
            if (D->tail_bonds) std::memcpy(h.data() + o_tb, D->tail_bonds, P->nidx_tail - D->ntails);
            else std::memset(h.data() + o_tb, 1, P->nidx_tail);
        }
        uint64_t *no = reinterpret_cast<uint64_t *>(h.data() + o_no);
        for (size_t t = 0; t <= D->ntails; ++t) no[t] = t;           // one normal per tail
        MH_HIP(hipMemcpy(P->consts.p, h.data(), B.size, hipMemcpyHostToDevice));
        char *d = P->consts.as<char>();
        P->lipid_idx = (const uint64_t *)(d + o_li); P->lipid_off = (const uint64_t *)(d + o_lo);
        P->marker_idx = (const uint64_t *)(d + o_mi); P->marker_off = (const uint64_t *)(d + o_mo);
        P->masses = (const float *)(d + o_ms);
        P->tail_idx = (const uint64_t *)(d + o_ti); P->tail_off = (const uint64_t *)(d + o_to); P->noff = (const uint64_t *)(d + o_no);
        P->tail_lipid = (const uint32_t *)(d + o_tl); P->tail_bonds = (const uint8_t *)(d + o_tb);
        MH_TRY(P->valid.reserve(K));
        MH_HIP(hipMemset(P->valid.p, 1, K));
        MH_HIP(hipStreamCreateWithFlags(&P->copy_stream, hipStreamNonBlocking));
        for (auto &S : P->slot) {
            MH_HIP(hipHostMalloc(&S.h, H_BYTES, hipHostMallocDefault));
            std::memset(S.h, 0, H_BYTES);
            MH_HIP(hipEventCreateWithFlags(&S.done, hipEventDisableTiming));
            MH_HIP(hipEventCreateWithFlags(&S.mid, hipEventDisableTiming));
        }
        P->Ecap = 64 * K;          // room for 32 patch members per lipid to start with
        return 0;
    }();
    if (rc) {
        molar_hip_membrane_plan_destroy(P);
        return rc;
    }
    *out = P;
    return MOLAR_HIP_OK;
}

extern "C" void molar_hip_membrane_plan_destroy(molar_hip_membrane_plan *P) {
    if (!P) return;
    (void)hipSetDevice(P->c->device);
    (void)hipStreamSynchronize(P->c->stream);
    if (P->copy_stream) {
        (void)hipStreamSynchronize(P->copy_stream);
        (void)hipStreamDestroy(P->copy_stream);
    }
    for (auto &S : P->slot) {
        S.blob.release();
        S.xyz_stage.release();
        if (S.h) (void)hipHostFree(S.h);
        if (S.h_mid) (void)hipHostFree(S.h_mid);
        if (S.done) (void)hipEventDestroy(S.done);
        if (S.mid) (void)hipEventDestroy(S.mid);
        if (S.back) (void)hipEventDestroy(S.back);
    }
    if (P->h_fetch) (void)hipHostFree(P->h_fetch);
    P->consts.release(); P->valid.release(); P->work.release();
    delete P;
}

extern "C" int molar_hip_membrane_plan_set_valid(molar_hip_membrane_plan *P, const uint8_t *valid) {
    if (!P) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "membrane_plan_set_valid: null plan");
    MH_HIP(hipSetDevice(P->c->device));
    for (int t = 0; t < 2; ++t)
        if (P->slot[t].pending) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "membrane_plan_set_valid: frame %d is in flight: end it first", t);
    MH_HIP(hipStreamSynchronize(P->c->stream));
    if (!valid) {
        MH_HIP(hipMemset(P->valid.p, 1, P->K));
    } else {
        std::vector<uint8_t> v(valid, valid + P->K);
        for (auto &b : v) b = b ? 1 : 0;
        MH_HIP(hipMemcpy(P->valid.p, v.data(), P->K, hipMemcpyHostToDevice));
    }
    return MOLAR_HIP_OK;
}

extern "C" int molar_hip_membrane_frame_begin(molar_hip_membrane_plan *P, float *xyz, const float *box9, int32_t *ticket) {
    if (!P || !xyz || !box9 || !ticket) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "membrane_frame_begin: null argument");
    molar_hip_ctx *c = P->c;
    MH_HIP(hipSetDevice(c->device));
    const int t = P->next;
    auto &S = P->slot[t];
    if (S.pending) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "both frames are in flight: call molar_hip_membrane_frame_end first");
    std::memcpy(S.box9, box9, 36);
    if (is_device_ptr(xyz)) {
        S.xyz_dev = xyz;
        S.xyz_host = nullptr;
    } else {
        MH_TRY(S.xyz_stage.reserve(P->natoms * 12));
        MH_HIP(hipMemcpyAsync(S.xyz_stage.p, xyz, P->natoms * 12, hipMemcpyHostToDevice, c->stream));
        S.xyz_dev = S.xyz_stage.as<float>();
        S.xyz_host = xyz;
    }
    S.ended = false;
    S.b_enqueued = false;
    S.serial = ++P->serial;
    MH_TRY(enqueue_a(P, S));
    // B needs the valid flags the older frame's C will leave, and that C is not enqueued yet (it follows the host pass in
    // the older frame's _end).  Smoothing rarely drops a lipid, so B runs now on the flags as they are - the GPU works on
    // it while the host is busy with the older frame - and is repeated behind that C if it did change them.
    S.speculative = P->slot[t ^ 1].pending;
    MH_TRY(enqueue_b(P, S, /*restore=*/false));
    // The header promises that a host frame holds its unwrapped coordinates when this call returns.  Into pageable memory
    // the runtime's copy blocks anyway; into pinned / registered memory it is asynchronous and would land after a caller
    // (the Rust wrapper's `&mut [f32]`) has given up the array - wait for that copy alone, with B already queued behind it.
    if (S.xyz_host && P->unwrap && S.back) MH_HIP(hipEventSynchronize(S.back));
    S.pending = true;
    P->next = t ^ 1;
    *ticket = t;
    return MOLAR_HIP_OK;
}

static void fill_view(molar_hip_membrane_plan *P, const molar_hip_membrane_plan::Slot &S, molar_hip_membrane_view *V) {
    const char *d = S.blob.as<char>();
    const FrameLayout &L = S.lay;
    V->nlipids = P->K; V->patch_entries = (size_t)S.info.E; V->npairs = (size_t)S.info.npairs;
    V->head = (const float *)(d + L.head); V->mid = (const float *)(d + L.mid); V->tail = (const float *)(d + L.tail);
    V->patch_offsets = (const uint64_t *)(d + L.poff); V->patch_ids = (const uint64_t *)(d + L.pids);
    V->initial_normals = (const float *)(d + L.normals0);
    V->valid = (const uint8_t *)(d + L.valid_out);
    V->smoothed_head = (const float *)(d + L.s_head); V->normals = (const float *)(d + L.s_normals);
    V->quad_coefs = (const float *)(d + L.coefs); V->mean_curv = (const float *)(d + L.mean); V->gauss_curv = (const float *)(d + L.gauss);
    V->princ_curvs = (const float *)(d + L.pcurv); V->princ_dirs = (const float *)(d + L.pdirs); V->area = (const float *)(d + L.area);
    V->nvert = (const uint32_t *)(d + L.nvert); V->neib_ids = (const uint64_t *)(d + L.neib); V->voro_vertexes = (const float *)(d + L.voro);
    V->fitted_patch_points = (const float *)(d + L.fitted);
    V->order = (const float *)(d + L.order); V->norder = P->norder;
}

namespace {
// Arrays of a finished frame stored into the pinned block of the fetch, all in one launch.  Sources and places in the block
// are 16-byte aligned (Blob2 / the 64-byte steps of the items); a tail of fewer than 16 bytes goes by bytes.
constexpr uint32_t FETCH_PACK_MAX = 20;
struct FetchPack {
    const char *src[FETCH_PACK_MAX + 1];
    char *dst[FETCH_PACK_MAX + 1];
    size_t bytes[FETCH_PACK_MAX + 1];
    uint32_t n;
};
__global__ __launch_bounds__(256) void k_fetch_pack(FetchPack Q) {
    const size_t tid = (size_t)blockIdx.x * 256u + threadIdx.x, nth = (size_t)gridDim.x * 256u;
    for (uint32_t k = 0; k < Q.n; ++k) {
        const char *s = Q.src[k];
        char *d = Q.dst[k];
        const size_t words = Q.bytes[k] / 16u;
        if ((((uintptr_t)s | (uintptr_t)d) & 15u) == 0) {
            for (size_t w = tid; w < words; w += nth) reinterpret_cast<uint4 *>(d)[w] = reinterpret_cast<const uint4 *>(s)[w];
            for (size_t b = words * 16u + tid; b < Q.bytes[k]; b += nth) d[b] = s[b];
        } else {
            for (size_t b = tid; b < Q.bytes[k]; b += nth) d[b] = s[b];
        }
    }
}

// A fetch in three steps: the list of arrays the caller asked for (and room for them in the pinned block), their way into the
// block on a stream, and - once that stream has been waited for - plain memcpy into the caller's arrays.
struct FetchItem { void *dst; const void *src; size_t bytes, at; };
struct FetchJob {
    std::vector<FetchItem> items;
    size_t total = 0;
};

int fetch_prepare(molar_hip_membrane_plan *P, const molar_hip_membrane_plan::Slot &S, const molar_hip_membrane_out *O, FetchJob &J) {
    molar_hip_membrane_view V;
    fill_view(P, S, &V);
    const size_t K = P->K, E = V.patch_entries, slots = E + 4 * K;
    auto want = [&](void *dst, const void *src, size_t bytes) {
        if (!dst || !bytes) return;
        J.items.push_back(FetchItem{dst, src, bytes, J.total});
        J.total += (bytes + 63) & ~size_t(63);
    };
    want(O->head, V.head, K * 12); want(O->mid, V.mid, K * 12); want(O->tail, V.tail, K * 12);
    want(O->patch_offsets, V.patch_offsets, (K + 1) * 8); want(O->patch_ids, V.patch_ids, E * 8);
    want(O->initial_normals, V.initial_normals, K * 12); want(O->valid, V.valid, K);
    want(O->smoothed_head, V.smoothed_head, K * 12); want(O->normals, V.normals, K * 12);
    want(O->quad_coefs, V.quad_coefs, K * 24); want(O->mean_curv, V.mean_curv, K * 4); want(O->gauss_curv, V.gauss_curv, K * 4);
    want(O->princ_curvs, V.princ_curvs, K * 8); want(O->princ_dirs, V.princ_dirs, K * 24); want(O->area, V.area, K * 4);
    want(O->nvert, V.nvert, K * 4); want(O->neib_ids, V.neib_ids, slots * 8); want(O->voro_vertexes, V.voro_vertexes, slots * 12);
    want(O->fitted_patch_points, V.fitted_patch_points, E * 12); want(O->order, V.order, P->norder * 4);
    if (J.total > P->h_fetch_cap) {
        if (P->h_fetch) (void)hipHostFree(P->h_fetch);
        P->h_fetch = nullptr;
        P->h_fetch_cap = 0;
        MH_HIP(hipHostMalloc(&P->h_fetch, J.total + J.total / 4, hipHostMallocDefault));
        P->h_fetch_cap = J.total + J.total / 4;
    }
    return 0;
}

// one kernel stores the arrays into the pinned block (a device-to-host copy is a launch of its own here, twenty of them cost
// more than the bytes); an array too large for that share goes by a copy
// (`also_src` .. `also_bytes`: one more block for the same launch - the frame's status words on their way to the host)
int fetch_enqueue(molar_hip_membrane_plan *P, const FetchJob &J, hipStream_t st, const void *also_src = nullptr, void *also_dst = nullptr,
                  size_t also_bytes = 0) {
    FetchPack pack{};
    for (const FetchItem &it : J.items) {
        if (it.bytes <= (4u << 20) && pack.n < FETCH_PACK_MAX) {
            pack.src[pack.n] = (const char *)it.src; pack.dst[pack.n] = (char *)P->h_fetch + it.at; pack.bytes[pack.n] = it.bytes;
            ++pack.n;
        } else {
            MH_HIP(hipMemcpyAsync((char *)P->h_fetch + it.at, it.src, it.bytes, hipMemcpyDeviceToHost, st));
        }
    }
    if (also_bytes) {
        pack.src[pack.n] = (const char *)also_src; pack.dst[pack.n] = (char *)also_dst; pack.bytes[pack.n] = also_bytes;
        ++pack.n;
    }
    if (pack.n) {
        hipLaunchKernelGGL(k_fetch_pack, dim3(64), dim3(256), 0, st, pack);
        MH_HIP(hipGetLastError());
    }
    return 0;
}

void fetch_finish(molar_hip_membrane_plan *P, const FetchJob &J) {
    for (const FetchItem &it : J.items) std::memcpy(it.dst, (const char *)P->h_fetch + it.at, it.bytes);
}

int frame_end(molar_hip_membrane_plan *P, int32_t ticket, molar_hip_membrane_view *view, const molar_hip_membrane_out *out) {
    MH_TRY(check_ticket(P, ticket, /*want_ended=*/false));
    MH_HIP(hipSetDevice(P->c->device));
    auto &S = P->slot[ticket];
    // frames end in begin order: the older one first
    auto &O = P->slot[ticket ^ 1];
    if (O.pending && O.serial < S.serial) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "membrane_frame_end: end the older frame (ticket %d) first", ticket ^ 1);
    if (!S.b_enqueued) MH_TRY(enqueue_b(P, S, /*restore=*/false));     // (only after an error left the chain short)
    if (!S.passed) MH_TRY(host_pass(P, S, /*may_repeat=*/true));
    FetchJob J;
    if (out) MH_TRY(fetch_prepare(P, S, out, J));
    const bool packed = !J.items.empty();
    MH_TRY(enqueue_c(P, S, /*info_to_host=*/!packed));
    if (packed) {  // the caller's arrays leave on the frame's own stream, right behind C - no second round trip for them -, and
                   // the frame's status words ride in the same launch
        char *d = S.blob.as<char>();
        MH_TRY(fetch_enqueue(P, J, P->c->stream, d + S.lay.info, (char *)S.h + H_INFO, sizeof(FrameInfo)));
        MH_HIP(hipEventRecord(S.done, P->c->stream));
    }
    // while the GPU smooths this frame: the host pass of the younger one, whose B is already through (it sits ahead of
    // this C on the stream)
    if (O.pending && O.b_enqueued && !O.passed) MH_TRY(host_pass(P, O, /*may_repeat=*/false));
    MH_HIP(hipEventSynchronize(S.done));
    std::memcpy(&S.info, (char *)S.h + H_INFO, sizeof(FrameInfo));
    if (O.pending && (!O.b_enqueued || (O.speculative && S.info.changed))) MH_TRY(enqueue_b(P, O, /*restore=*/false));
    O.speculative = false;
    S.pending = false;
    S.ended = true;
    if (view) fill_view(P, S, view);
    if (out) fetch_finish(P, J);
    if (S.info.st_center) return fail(S.info.st_center, "membrane frame: a marker selection has zero mass");
    if (S.info.st_order) return fail(S.info.st_order, "membrane frame: lipid order error (status %d)", S.info.st_order);
    return MOLAR_HIP_OK;
}
}  // namespace

extern "C" int molar_hip_membrane_frame_end(molar_hip_membrane_plan *P, int32_t ticket, molar_hip_membrane_view *view) {
    return frame_end(P, ticket, view, nullptr);
}

extern "C" int molar_hip_membrane_frame_end_fetch(molar_hip_membrane_plan *P, int32_t ticket, molar_hip_membrane_view *view,
                                                  const molar_hip_membrane_out *out) {
    if (!out) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "membrane_frame_end_fetch: null argument");
    // the caller cannot size the arrays that follow the frame's patch entries before the frame has ended
    if (out->patch_ids || out->neib_ids || out->voro_vertexes || out->fitted_patch_points)
        return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "membrane_frame_end_fetch: per-lipid arrays and order only (patch-sized arrays: _frame_fetch after the view)");
    return frame_end(P, ticket, view, out);
}

extern "C" int molar_hip_membrane_frame_fetch(molar_hip_membrane_plan *P, int32_t ticket, const molar_hip_membrane_out *O) {
    MH_TRY(check_ticket(P, ticket, /*want_ended=*/true));
    if (!O) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "membrane_frame_fetch: null argument");
    MH_HIP(hipSetDevice(P->c->device));
    FetchJob J;
    MH_TRY(fetch_prepare(P, P->slot[ticket], O, J));
    if (J.items.empty()) return MOLAR_HIP_OK;
    // on the copy stream, beside the kernels of the frame in flight
    MH_TRY(fetch_enqueue(P, J, P->copy_stream));
    MH_HIP(hipStreamSynchronize(P->copy_stream));
    fetch_finish(P, J);
    return MOLAR_HIP_OK;
}

