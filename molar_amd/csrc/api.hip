// api.hip — context lifecycle and the host-side PeriodicBox constructors of libmolar_hip.so.
#include "boxmath.hpp"
#include <algorithm>

#include "common.hpp"
#include "hoststream.hpp"
namespace mh { void search64_release(molar_hip_ctx *c); }
#include <cstdlib>

using namespace mh;

// Device-memory ceilings measured in the same process as the benchmark.  Three copy shapes: a plain
// grid-stride float4 copy, four independent 16-byte loads in flight per thread, and the same with non-temporal
// accesses (one CU then keeps 4x the bytes in flight; the guide's 6.3 TB/s figure needs that).  The best one is reported.
typedef float v4f_copy __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_copy_f4(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (size_t)gridDim.x * 256u) dst[i] = src[i];
}
template <bool NT>
__global__ __launch_bounds__(256) void k_copy_f4x4(const v4f_copy *__restrict__ src, v4f_copy *__restrict__ dst, size_t n) {
    // n is a multiple of 1024: every block iteration moves 1024 float4 (16 KiB), four per thread, 256 apart
    for (size_t i = (size_t)blockIdx.x * 1024u + threadIdx.x; i < n; i += (size_t)gridDim.x * 1024u) {
        v4f_copy v0, v1, v2, v3;
        if (NT) {
            v0 = __builtin_nontemporal_load(src + i);
            v1 = __builtin_nontemporal_load(src + i + 256);
            v2 = __builtin_nontemporal_load(src + i + 512);
            v3 = __builtin_nontemporal_load(src + i + 768);
            __builtin_nontemporal_store(v0, dst + i);
            __builtin_nontemporal_store(v1, dst + i + 256);
            __builtin_nontemporal_store(v2, dst + i + 512);
            __builtin_nontemporal_store(v3, dst + i + 768);
        } else {
            v0 = src[i]; v1 = src[i + 256]; v2 = src[i + 512]; v3 = src[i + 768];
            dst[i] = v0; dst[i + 256] = v1; dst[i + 512] = v2; dst[i + 768] = v3;
        }
    }
}
template <bool NT>
__global__ __launch_bounds__(256) void k_write_f4x4(v4f_copy *__restrict__ dst, size_t n) {
    const v4f_copy v = {1.f, 2.f, 3.f, (float)threadIdx.x};
    for (size_t i = (size_t)blockIdx.x * 1024u + threadIdx.x; i < n; i += (size_t)gridDim.x * 1024u) {
        if (NT) {
            __builtin_nontemporal_store(v, dst + i);
            __builtin_nontemporal_store(v, dst + i + 256);
            __builtin_nontemporal_store(v, dst + i + 512);
            __builtin_nontemporal_store(v, dst + i + 768);
        } else {
            dst[i] = v; dst[i + 256] = v; dst[i + 512] = v; dst[i + 768] = v;
        }
    }
}

// mode 0: copy (bytes read + bytes written per second), mode 1: write-only stream
static int bandwidth_probe(molar_hip_ctx *c, size_t bytes, int reps, int mode, float *gbs) {
    if (!c || !gbs || reps < 1 || bytes < 16384) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "bandwidth probe: bad argument");
    MH_HIP(hipSetDevice(c->device));
    const size_t n = (bytes / 16) & ~(size_t)1023;
    float4 *a = nullptr, *b = nullptr;
    MH_HIP(hipMalloc((void **)&a, n * 16));
    if (hipMalloc((void **)&b, n * 16) != hipSuccess) {
        (void)hipFree(a);
        return fail(MOLAR_HIP_ERR_HIP, "bandwidth probe: out of device memory");
    }
    (void)hipMemsetAsync(a, 0, n * 16, c->stream);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 0.f;
    hipError_t err = hipSuccess;
    const unsigned grids[2] = {(unsigned)std::min<size_t>(n / 1024, (size_t)c->num_cus * 8),
                               (unsigned)std::min<size_t>(n / 1024, (size_t)c->num_cus * 32)};
    for (int variant = 0; variant < (mode == 0 ? 6 : 4) && err == hipSuccess; ++variant) {
        const unsigned grid = grids[variant & 1];
        auto launch = [&]() {
            const v4f_copy *s4 = reinterpret_cast<const v4f_copy *>(a);
            v4f_copy *d4 = reinterpret_cast<v4f_copy *>(b);
            if (mode == 0) {
                if (variant < 2) hipLaunchKernelGGL(k_copy_f4, dim3(grid * 4), dim3(256), 0, c->stream, a, b, n);
                else if (variant < 4) hipLaunchKernelGGL(k_copy_f4x4<false>, dim3(grid), dim3(256), 0, c->stream, s4, d4, n);
                else hipLaunchKernelGGL(k_copy_f4x4<true>, dim3(grid), dim3(256), 0, c->stream, s4, d4, n);
            } else {
                if (variant < 2) hipLaunchKernelGGL(k_write_f4x4<false>, dim3(grid), dim3(256), 0, c->stream, d4, n);
                else hipLaunchKernelGGL(k_write_f4x4<true>, dim3(grid), dim3(256), 0, c->stream, d4, n);
            }
        };
        launch();   // warm-up
        (void)hipEventRecord(e0, c->stream);
        for (int r = 0; r < reps; ++r) launch();
        (void)hipEventRecord(e1, c->stream);
        err = hipStreamSynchronize(c->stream);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (err == hipSuccess && ms > 0.f) {
            const float r = (float)((mode == 0 ? 2.0 : 1.0) * (double)(n * 16) * reps / (ms * 1e-3) / 1e9);
            if (r > best) best = r;
        }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(a);
    (void)hipFree(b);
    if (err != hipSuccess || !(best > 0.f)) return fail(MOLAR_HIP_ERR_HIP, "bandwidth probe: %s", hipGetErrorString(err));
    *gbs = best;
    return MOLAR_HIP_OK;
}


extern "C" {

const char *molar_hip_last_error(void) { return last_error().c_str(); }

const char *molar_hip_version(void) { return "molar_hip 0.2 (gfx950)"; }

int molar_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

molar_hip_ctx *molar_hip_create(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) {
        (void)hipGetLastError();
        fail(MOLAR_HIP_ERR_HIP, "molar_hip_create: no HIP device visible (%s)",
             e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
        return nullptr;
    }
    if (device < 0 || device >= n) {
        fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "molar_hip_create: device %d out of range [0,%d)", device, n);
        return nullptr;
    }
    if ((e = hipSetDevice(device)) != hipSuccess) {
        fail(MOLAR_HIP_ERR_HIP, "hipSetDevice(%d): %s", device, hipGetErrorString(e));
        return nullptr;
    }
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) {
        fail(MOLAR_HIP_ERR_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
        return nullptr;
    }
    auto *c = new molar_hip_ctx();
    c->device = device;
    c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) {
        fail(MOLAR_HIP_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
        delete c;
        return nullptr;
    }
    c->own_stream = true;
    // The release library has ONE code path: no environment variable selects a kernel.  Builds with -DMOLAR_HIP_AB_KNOBS
    // (tools/build_variant.sh, A/B runs on one box) read these switches here, once - the per-search paths never call getenv.
#ifdef MOLAR_HIP_AB_KNOBS
    c->env_no_side = std::getenv("MOLAR_HIP_NO_SIDE_STREAM") != nullptr;
    c->env_no_mfma = std::getenv("MOLAR_HIP_NO_MFMA_COUNT") != nullptr;
    c->env_no_mfma_wrapped = std::getenv("MOLAR_HIP_NO_MFMA_WRAPPED") != nullptr;
    c->env_no_tile_sum = std::getenv("MOLAR_HIP_NO_TILE_SUM") != nullptr;
    c->env_host_grid_wait = std::getenv("MOLAR_HIP_HOST_GRID_WAIT") != nullptr;
    c->env_grid_late = std::getenv("MOLAR_HIP_GRID_LATE") != nullptr;
    c->env_no_bin_tile = std::getenv("MOLAR_HIP_NO_BIN_TILE") != nullptr;
#endif
#ifdef MOLAR_HIP_DEBUG_KNOBS
    if (const char *dbg = std::getenv("MOLAR_HIP_DEBUG_SKIP")) c->env_debug_skip = (uint32_t)std::atoi(dbg);
    if (const char *dbg = std::getenv("MOLAR_HIP_DEBUG_LAUNCH")) c->env_debug_launch = (uint32_t)std::atoi(dbg);
#endif
    if (ensure_pinned(c, 1 << 16)) {
        (void)hipStreamDestroy(c->stream);
        delete c;
        return nullptr;
    }
    return c;
}

void molar_hip_destroy(molar_hip_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->side_stream) (void)hipStreamSynchronize(c->side_stream);
    for (auto &gen : c->set_store) for (auto &s : gen) {
        for (DevBuf *b : {&s.xyz_stage, &s.idx_stage, &s.vdw_stage, &s.key, &s.cell_count, &s.cnt_pad, &s.cursor, &s.tmp_key, &s.sort_buf,
                          &s.sorted, &s.sorted_vdw, &s.aabb, &s.perm, &s.chunk_aabb, &s.h16, &s.cell_org})
            b->release();
    }
    if (c->aux) molar_hip_destroy(c->aux);
    for (DevBuf *b : {&c->xh_win[0], &c->xh_win[1], &c->xh_bins, &c->fit_redo}) b->release();
    for (auto &gen : c->hb_sets) for (auto &fr : gen) for (auto &s : fr) {
        for (DevBuf *b : {&s.xyz_stage, &s.idx_stage, &s.vdw_stage, &s.key, &s.cell_count, &s.cnt_pad, &s.cursor, &s.tmp_key, &s.sort_buf,
                          &s.sorted, &s.sorted_vdw, &s.aabb, &s.perm, &s.chunk_aabb, &s.h16, &s.cell_org})
            b->release();
    }
    for (DevBuf *b : {&c->hb_lean[0], &c->hb_lean[1], &c->hb_rest[0], &c->hb_rest[1], &c->hb_blocks[0], &c->hb_blocks[1]}) b->release();
    if (c->hb_pin) (void)hipHostFree(c->hb_pin);
    for (auto e : c->hb_pin_ev)
        if (e) (void)hipEventDestroy(e);
    for (DevBuf *b : {&c->params, &c->task_desc, &c->task_nb, &c->slot_desc, &c->slot_cnt, &c->slot_base, &c->tile_sum, &c->scan_tmp, &c->sort_tmp, &c->sort_tmp_side, &c->scan_tmp_side, &c->scan_state, &c->fplan_tiles, &c->out_pairs_set[0], &c->out_dist_set[0], &c->out_pairs_set[1], &c->out_dist_set[1], &c->out_ids,
                      &c->wide_i, &c->wide_j, &c->hist, &c->hist_queue, &c->slot_desc_rest, &c->hist_edges, &c->dbg, &c->conn_deg, &c->conn_off, &c->conn_ent, &c->conn_neigh, &c->w_flags, &c->w_list, &c->w_part_cnt, &c->w_part, &c->w_tile_cnt, &c->w_tile_off, &c->task_mu, &c->task_moff, &c->maskbuf, &c->m_xyz1, &c->m_xyz2, &c->m_idx1, &c->m_idx2,
                      &c->m_mass1, &c->m_mass2, &c->m_partials, &c->m_results, &c->m_out})
        b->release();
    for (auto &s : c->spans) {
        (void)hipEventDestroy(s.a);
        (void)hipEventDestroy(s.b);
    }
    for (auto e : c->event_pool) (void)hipEventDestroy(e);
    if (c->h_pinned) (void)hipHostFree(c->h_pinned);
    mh::ring_release(c);
    mh::search64_release(c);
    if (c->h_sizes) (void)hipHostFree(c->h_sizes);
    for (auto &t : c->tickets)
        if (t.done) (void)hipEventDestroy(t.done);
    if (c->grid_done) (void)hipEventDestroy(c->grid_done);
    if (c->count_done) (void)hipEventDestroy(c->count_done);
    for (auto e : c->gen_free)
        if (e) (void)hipEventDestroy(e);
    if (c->side_stream) (void)hipStreamDestroy(c->side_stream);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int molar_hip_set_stream(molar_hip_ctx *c, void *hip_stream) {
    if (!c) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "null context");
    MH_HIP(hipSetDevice(c->device));
    MH_HIP(hipStreamSynchronize(c->stream));
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    c->stream = reinterpret_cast<hipStream_t>(hip_stream);
    c->own_stream = false;
    return MOLAR_HIP_OK;
}

int molar_hip_synchronize(molar_hip_ctx *c) {
    if (!c) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "null context");
    MH_HIP(hipStreamSynchronize(c->stream));
    return MOLAR_HIP_OK;
}

int molar_hip_profile_enable(molar_hip_ctx *c, int on) {
    if (!c) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "null context");
    c->profiling = on != 0;
    c->profile_frames = on == 2;
    return MOLAR_HIP_OK;
}

int molar_hip_profile_read(molar_hip_ctx *c, float ms[MOLAR_HIP_PROFILE_CLASSES],
                           uint64_t launches[MOLAR_HIP_PROFILE_CLASSES]) {
    if (!c) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "null context");
    MH_HIP(hipSetDevice(c->device));
    MH_HIP(hipStreamSynchronize(c->stream));
    for (int k = 0; k < MOLAR_HIP_PROFILE_CLASSES; ++k) {
        if (ms) ms[k] = 0.f;
        if (launches) launches[k] = 0;
    }
    for (auto &s : c->spans) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, s.a, s.b) == hipSuccess && s.cls >= 0 && s.cls < MOLAR_HIP_PROFILE_CLASSES) {
            if (ms) ms[s.cls] += t;
            if (launches) launches[s.cls] += 1;
        }
        c->event_pool.push_back(s.a);
        c->event_pool.push_back(s.b);
    }
    c->spans.clear();
    return MOLAR_HIP_OK;
}

int molar_hip_copy_bandwidth(molar_hip_ctx *c, size_t bytes, int reps, float *gbs) { return bandwidth_probe(c, bytes, reps, 0, gbs); }
int molar_hip_write_bandwidth(molar_hip_ctx *c, size_t bytes, int reps, float *gbs) { return bandwidth_probe(c, bytes, reps, 1, gbs); }

// ---------------------------------------------------------------- PeriodicBox constructors (host)

// nalgebra try_inverse, 3x3 closed form (called at periodic_box.rs:167-169)
static bool invert3(const float *m, float *o) {
    const float a = m[0], d = m[1], g = m[2];   // column 0 : (0,0) (1,0) (2,0)
    const float b = m[3], e = m[4], h = m[5];   // column 1
    const float c = m[6], f = m[7], i = m[8];   // column 2
    // row-major names: [a b c; d e f; g h i]
    const float minor_bf = e * i - h * f;
    const float minor_af = d * i - g * f;
    const float minor_ae = d * h - g * e;
    const float det = (a * minor_bf - b * minor_af) + c * minor_ae;
    if (det == 0.0f) return false;
    o[0] = minor_bf / det;             // (0,0)
    o[3] = (c * h - i * b) / det;      // (0,1)
    o[6] = (b * f - e * c) / det;      // (0,2)
    o[1] = -minor_af / det;            // (1,0)
    o[4] = (a * i - g * c) / det;      // (1,1)
    o[7] = (c * d - f * a) / det;      // (1,2)
    o[2] = minor_ae / det;             // (2,0)
    o[5] = (b * g - h * a) / det;      // (2,1)
    o[8] = (a * e - d * b) / det;      // (2,2)
    return true;
}

static float len3(V3 v) { return std::sqrt(norm2(v)); }

int molar_hip_box_from_matrix(const float m9[9], molar_hip_box *out) {
    if (!m9 || !out) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "box_from_matrix: null argument");
    V3 col[3];
    for (int k = 0; k < 3; ++k) {
        col[k] = v3(m9[3 * k], m9[3 * k + 1], m9[3 * k + 2]);
        if (len3(col[k]) == 0.0f) return fail(MOLAR_HIP_ERR_ZERO_LENGTH_VECTOR, "zero length box vector");
    }
    std::memcpy(out->m, m9, sizeof out->m);
    if (!invert3(out->m, out->inv)) return fail(MOLAR_HIP_ERR_INVERSE_FAILED, "box matrix inverse failed");
    out->nshift = 0;
    // build_tric_corrections (periodic_box.rs:25-66)
    const bool ortho = m9[3] == 0.f && m9[6] == 0.f && m9[1] == 0.f && m9[7] == 0.f && m9[2] == 0.f && m9[5] == 0.f;
    if (ortho) return MOLAR_HIP_OK;
    const V3 a = col[0], b = col[1], c = col[2];
    const V3 na = v3(-a.x, -a.y, -a.z);
    float longest = std::fmax(std::fmax(std::fmax(len3((a + b) + c), len3((a + b) - c)), len3((a - b) + c)),
                              len3((na + b) + c));
    const float half_diag = 0.5f * longest;
    const float two = 2.0f * half_diag;
    const float bound2 = two * two;
    for (int i = -1; i <= 1; ++i)
        for (int j = -1; j <= 1; ++j)
            for (int k = -1; k <= 1; ++k) {
                if (!i && !j && !k) continue;
                const float fi = (float)i, fj = (float)j, fk = (float)k;
                V3 s = (v3(fi * a.x, fi * a.y, fi * a.z) + v3(fj * b.x, fj * b.y, fj * b.z)) +
                       v3(fk * c.x, fk * c.y, fk * c.z);
                if (norm2(s) < bound2) {
                    float *dst = out->shifts + 3 * out->nshift++;
                    dst[0] = s.x; dst[1] = s.y; dst[2] = s.z;
                }
            }
    return MOLAR_HIP_OK;
}

int molar_hip_box_from_vectors_angles(float a, float b, float c, float alpha, float beta, float gamma,
                                      molar_hip_box *out) {
    if (a == 0.f || b == 0.f || c == 0.f) return fail(MOLAR_HIP_ERR_ZERO_LENGTH_VECTOR, "zero length box vector");
    if (alpha < 60.f || beta < 60.f || gamma < 60.f) return fail(MOLAR_HIP_ERR_ANGLE_TOO_SMALL, "box angle is <60 deg");
    float m[9] = {0};
    m[0] = a;
    if (alpha != 90.f || beta != 90.f || gamma != 90.f) {
        const float d2r = 3.14159265358979323846f / 180.0f;
        const float cosa = alpha != 90.f ? std::cos(alpha * d2r) : 0.f;
        const float cosb = beta != 90.f ? std::cos(beta * d2r) : 0.f;
        float sing = 1.f, cosg = 0.f;
        if (gamma != 90.f) {
            sing = std::sin(gamma * d2r);
            cosg = std::cos(gamma * d2r);
        }
        m[3] = b * cosg;                        // (0,1)
        m[4] = b * sing;                        // (1,1)
        m[6] = c * cosb;                        // (0,2)
        m[7] = c * (cosa - cosb * cosg) / sing; // (1,2)
        m[8] = std::sqrt(c * c - m[6] * m[6] - m[7] * m[7]);
    } else {
        m[4] = b;
        m[8] = c;
    }
    return molar_hip_box_from_matrix(m, out);
}

void molar_hip_box_shortest_vector(const molar_hip_box *box, const float v[3], uint8_t pbc, float out[3]) {
    V3 r = shortest_vector(*box, v3(v[0], v[1], v[2]), pbc);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

void molar_hip_box_extents(const molar_hip_box *box, float out[3]) {
    for (int c = 0; c < 3; ++c) out[c] = len3(v3(box->m[3 * c], box->m[3 * c + 1], box->m[3 * c + 2]));
}

void molar_hip_box_to_box_coords(const molar_hip_box *box, const float v[3], float out[3]) {
    const V3 r = mat_vec(box->inv, v3(v[0], v[1], v[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

void molar_hip_box_to_lab_coords(const molar_hip_box *box, const float v[3], float out[3]) {
    const V3 r = mat_vec(box->m, v3(v[0], v[1], v[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

int molar_hip_box_is_inside(const molar_hip_box *box, const float p[3]) {
    const V3 v = mat_vec(box->inv, v3(p[0], p[1], p[2]));
    return (v.x < 1.0f && v.y < 1.0f && v.z < 1.0f && v.x >= 0.0f && v.y >= 0.0f && v.z >= 0.0f) ? 1 : 0;
}

void molar_hip_box_wrap_point(const molar_hip_box *box, const float p[3], float out[3]) {
    const V3 f = mat_vec(box->inv, v3(p[0], p[1], p[2]));
    float bv[3] = {f.x, f.y, f.z};
    for (int i = 0; i < 3; ++i) {
        bv[i] = fract_rs(bv[i]);
        if (bv[i] < 0.0f) bv[i] = 1.0f - bv[i];     // sic (periodic_box.rs:414-416)
    }
    const V3 r = mat_vec(box->m, v3(bv[0], bv[1], bv[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

void molar_hip_box_lab_extents(const molar_hip_box *box, float out[3]) {
    const float *m = box->m;
    out[0] = (m[0] + m[3]) + m[6];
    out[1] = (m[1] + m[4]) + m[7];
    out[2] = (m[2] + m[5]) + m[8];
}

}  // extern "C"
