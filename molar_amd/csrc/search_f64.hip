// search_f64.hip - the eight distance_search drivers for MolAR built with its `f64` feature (Float = f64:
// molar/src/aliases.rs:10-13, molar/Cargo.toml:56-60), correct before fast.
//
// The f32 search (search.hip, pair_kernels.hpp) is the tuned product path; an f64 build of MolAR flips `Float` for the
// whole crate including distance_search, so results there are decided by f64 arithmetic: which cell an atom near a cell
// face lands in, which pairs at the cutoff are hits, and the distances themselves.  This file restates the drivers once
// more with every operation in double and NO tuning:
//   * grid dims, populate / populate_pbc (drop rule, wrap, later-dimension quirk, in-box atoms before wrapped atoms)
//     and the 14-mask plan run on the HOST, serially (distance_search.rs:103-269);
//   * one 64-lane wave per 64-row slot of a plan entry evaluates the exact f64 predicate on the device - plain
//     |p2 - p1|^2 or PeriodicBox::distance_squared (periodic_box.rs:286-318) for entries across the periodic boundary -
//     first to count, then, behind an exclusive scan of the slot counts on the host, to write (i, j, sqrt(d2)) at its
//     place of the reference's output order (plan order, then i-major / j-minor, :949-953);
//   * records are 24-byte positions + 8-byte ids; results are (usize, usize, f64) columns or usize ids (within).
// tests/test_gpu_search_f64.py compares all eight drivers bit for bit with the f64 build of the CPU checker.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "boxmath64.hpp"
#include "common.hpp"
#include "hoststream.hpp"

using namespace mh;

struct molar_hip_search64_state {
    int kind = 0;
    uint64_t dims[3] = {1, 1, 1};
    uint64_t total = 0;
    uint32_t nslots = 0;
    bool have = false;
    bool use_box = false;
    uint8_t pbc = 0;
    double cutoff = 0.0;
    DevBuf posA, idA, vdwA, posB, idB, vdwB, slots, slot_cnt, slot_base, box, out_i, out_j, out_d;
};

namespace {

// MASK of distance_search.rs:39-60
const uint8_t MASKS64[14][6] = {
    {0, 0, 0, 0, 0, 0},
    {0, 0, 0, 1, 0, 0}, {0, 0, 0, 0, 1, 0}, {0, 0, 0, 0, 0, 1},
    {0, 0, 0, 1, 1, 0}, {0, 0, 0, 1, 0, 1}, {0, 0, 0, 0, 1, 1},
    {0, 0, 0, 1, 1, 1},
    {1, 0, 0, 0, 1, 0}, {1, 0, 0, 0, 0, 1}, {0, 1, 0, 0, 0, 1},
    {1, 1, 0, 0, 0, 1}, {1, 0, 1, 0, 1, 0}, {0, 1, 1, 1, 0, 0},
};

// Rust `as usize` / `as isize` on f64: saturating, NaN -> 0
inline uint64_t as_usize(double x) {
    if (!(x > 0.0)) return 0;
    if (x >= 18446744073709551615.0) return ~0ull;
    return (uint64_t)x;
}
inline int64_t as_isize(double x) {
    if (x != x) return 0;
    if (x >= 9223372036854775807.0) return INT64_MAX;
    if (x <= -9223372036854775808.0) return INT64_MIN;
    return (int64_t)x;
}

// one set of the search on the host: selected positions in the caller's order, the id each one reports, its radius
struct HostSet {
    std::vector<double> pos;      // 3 * n
    std::vector<uint64_t> id;
    std::vector<double> vdw;      // n or empty
};

// a cell grid in CSR form, items in the reference's push order (:180, :203-209)
struct HostGrid {
    uint64_t dims[3];
    std::vector<uint32_t> start;          // ncells + 1
    std::vector<double> pos;              // 3 * kept (wrapped atoms carry their wrapped image, :196)
    std::vector<uint64_t> id;
    std::vector<double> vdw;
    uint32_t len(size_t c) const { return start[c + 1] - start[c]; }
};

struct Slot64 {                    // what a wave needs for one 64-row slot
    uint32_t a0, rows, i0, b0, n2, flags;     // flags: wrap | tri << 8
    uint32_t pad0, pad1;
};

int fetch_host(const double *src, size_t count, std::vector<double> &tmp, const double **out) {
    if (!src || !count) {
        *out = nullptr;
        return 0;
    }
    if (!is_device_ptr(src)) {
        *out = src;
        return 0;
    }
    tmp.resize(count);
    MH_HIP(hipMemcpy(tmp.data(), src, count * 8, hipMemcpyDeviceToHost));
    *out = tmp.data();
    return 0;
}
int fetch_host_u64(const uint64_t *src, size_t count, std::vector<uint64_t> &tmp, const uint64_t **out) {
    if (!src || !count) {
        *out = nullptr;
        return 0;
    }
    if (!is_device_ptr(src)) {
        *out = src;
        return 0;
    }
    tmp.resize(count);
    MH_HIP(hipMemcpy(tmp.data(), src, count * 8, hipMemcpyDeviceToHost));
    *out = tmp.data();
    return 0;
}

int gather_set(const double *xyz, size_t natoms, const uint64_t *idx, size_t n, const double *vdw, bool ids_local, HostSet &S) {
    if (!xyz) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "search_f64: xyz pointer is null");
    std::vector<double> tx, tv;
    std::vector<uint64_t> ti;
    const double *hx, *hv;
    const uint64_t *hi;
    MH_TRY(fetch_host(xyz, natoms * 3, tx, &hx));
    MH_TRY(fetch_host_u64(idx, idx ? n : 0, ti, &hi));
    const size_t nsel = idx ? n : natoms;
    if (nsel >= 0x7FFFFFFFull) return fail(MOLAR_HIP_ERR_TOO_LARGE, "search_f64: %zu atoms exceed the 2^31 limit", nsel);
    MH_TRY(fetch_host(vdw, vdw ? nsel : 0, tv, &hv));
    S.pos.resize(nsel * 3);
    S.id.resize(nsel);
    for (size_t k = 0; k < nsel; ++k) {
        const uint64_t a = hi ? hi[k] : (uint64_t)k;
        if (a >= natoms) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "search_f64: index %llu out of range", (unsigned long long)a);
        S.pos[3 * k] = hx[3 * a];
        S.pos[3 * k + 1] = hx[3 * a + 1];
        S.pos[3 * k + 2] = hx[3 * a + 2];
        S.id[k] = ids_local ? (uint64_t)k : a;
    }
    if (hv) S.vdw.assign(hv, hv + nsel);
    return 0;
}

// Grid::populate (:120-142) / populate_pbc (:144-210) + the push order, as CSR
void build_grid(const HostSet &S, const uint64_t dims[3], const BoxD *box, uint8_t pbc, const double lower[3], const double upper[3],
                HostGrid &G) {
    const size_t n = S.id.size();
    const size_t ncells = (size_t)(dims[0] * dims[1] * dims[2]);
    std::memcpy(G.dims, dims, sizeof G.dims);
    std::vector<uint32_t> cell(n);
    std::vector<uint8_t> phase(n, 0);
    std::vector<double> p3(S.pos);                 // the position stored with the item (wrapped image for phase 1)
    constexpr uint32_t DROPPED = 0xFFFFFFFFu;
    for (size_t k = 0; k < n; ++k) {
        const double *p = &S.pos[3 * k];
        uint64_t loc[3] = {0, 0, 0};
        if (!box) {
            const double dim_sz[3] = {upper[0] - lower[0], upper[1] - lower[1], upper[2] - lower[2]};
            bool ok = true;
            for (int d = 0; d < 3; ++d) {
                const int64_t nn = as_isize(std::floor((double)dims[d] * (p[d] - lower[d]) / dim_sz[d]));    // :131
                if (nn < 0 || nn >= (int64_t)dims[d]) { ok = false; break; }
                loc[d] = (uint64_t)nn;
            }
            cell[k] = ok ? (uint32_t)(loc[0] + loc[1] * dims[0] + loc[2] * dims[0] * dims[1]) : DROPPED;
            continue;
        }
        D3 rel = mat_vec(box->inv, D3{p[0], p[1], p[2]});                                 // :156
        double r[3] = {rel.x, rel.y, rel.z};
        bool correct = true, drop = false;
        for (int d = 0; d < 3; ++d)                                                         // :161-171
            if (r[d] < 0.0 || r[d] >= 1.0) {
                if (!((pbc >> d) & 1u)) { drop = true; break; }
                correct = false;
                break;
            }
        if (drop) { cell[k] = DROPPED; continue; }
        if (!correct) {                                                                     // :181-199
            for (int d = 0; d < 3; ++d)
                if ((pbc >> d) & 1u) {
                    r[d] = r[d] - std::trunc(r[d]);                                         // fract()
                    if (r[d] < 0.0) r[d] = 1.0 + r[d];
                }
            const D3 w = mat_vec(box->m, D3{r[0], r[1], r[2]});                             // :196
            p3[3 * k] = w.x; p3[3 * k + 1] = w.y; p3[3 * k + 2] = w.z;
            phase[k] = 1;
        }
        for (int d = 0; d < 3; ++d) {
            uint64_t l = as_usize(std::floor(r[d] * (double)dims[d]));                      // :175, :191
            if (l > dims[d] - 1) l = dims[d] - 1;
            loc[d] = l;
        }
        cell[k] = (uint32_t)(loc[0] + loc[1] * dims[0] + loc[2] * dims[0] * dims[1]);
    }
    G.start.assign(ncells + 1, 0u);
    size_t kept = 0;
    for (size_t k = 0; k < n; ++k)
        if (cell[k] != DROPPED) { G.start[cell[k] + 1]++; ++kept; }
    for (size_t c = 0; c < ncells; ++c) G.start[c + 1] += G.start[c];
    G.pos.resize(kept * 3);
    G.id.resize(kept);
    if (!S.vdw.empty()) G.vdw.resize(kept);
    std::vector<uint32_t> cur(G.start.begin(), G.start.end() - 1);
    for (int ph = 0; ph < 2; ++ph)
        for (size_t k = 0; k < n; ++k)
            if (cell[k] != DROPPED && phase[k] == ph) {
                const uint32_t at = cur[cell[k]]++;
                G.pos[3 * at] = p3[3 * k]; G.pos[3 * at + 1] = p3[3 * k + 1]; G.pos[3 * at + 2] = p3[3 * k + 2];
                G.id[at] = S.id[k];
                if (!S.vdw.empty()) G.vdw[at] = S.vdw[k];
            }
}

// compute_min_max (:602-616, seeded with zeros) + compute_bounding_box_single / _double (:618-646)
void bounding_box(double cutoff, const HostSet &A, const HostSet *B, double lower[3], double upper[3]) {
    auto mm = [](const HostSet &S, double lo[3], double hi[3]) {
        for (int d = 0; d < 3; ++d) lo[d] = hi[d] = 0.0;
        for (size_t k = 0; k < S.id.size(); ++k)
            for (int d = 0; d < 3; ++d) {
                const double v = S.pos[3 * k + d];
                if (v < lo[d]) lo[d] = v;
                if (v > hi[d]) hi[d] = v;
            }
    };
    double l1[3], u1[3];
    mm(A, l1, u1);
    for (int d = 0; d < 3; ++d) { lower[d] = l1[d]; upper[d] = u1[d]; }
    if (B) {
        double l2[3], u2[3];
        mm(*B, l2, u2);
        for (int d = 0; d < 3; ++d) {
            lower[d] = l1[d] < l2[d] ? l1[d] : l2[d];
            upper[d] = u1[d] > u2[d] ? u1[d] : u2[d];
        }
    }
    for (int d = 0; d < 3; ++d) {
        lower[d] += (-cutoff - 2.220446049250313e-16);
        upper[d] += (cutoff + 2.220446049250313e-16);
    }
}

struct Params64 {
    const double *posA, *posB, *vdwA, *vdwB;
    const unsigned long long *idA, *idB;
    const BoxD *box;
    const Slot64 *slots;
    uint32_t nslots;
    int kind;
    int use_box;
    double cutoff2;
};

// squared distance of one candidate: plain (:488) or PeriodicBox::distance_squared for an entry that wrapped (:485-486)
__device__ __forceinline__ double pair_d2(const Params64 &P, uint32_t wrap, D3 a, D3 b) {
    const D3 v = b - a;                                                                      // p2 - p1
    if (P.use_box && wrap) return norm2(shortest_vector(*P.box, v, wrap));
    return norm2(v);
}

template <bool FILL>
__global__ void __launch_bounds__(64) pair64_kernel(Params64 P, uint32_t *__restrict__ slot_cnt,
                                                    const unsigned long long *__restrict__ slot_base,
                                                    unsigned long long *__restrict__ out_i, unsigned long long *__restrict__ out_j,
                                                    double *__restrict__ out_d) {
    const uint32_t s = blockIdx.y * gridDim.x + blockIdx.x;
    if (s >= P.nslots) return;
    const Slot64 S = P.slots[s];
    const uint32_t lane = threadIdx.x;
    const uint32_t wrap = S.flags & 7u;
    const bool tri = (S.flags >> 8) & 1u;
    // rows are atoms of the first set's grid, columns atoms of the second set's (the same grid for SINGLE)
    const double *pa = P.posA, *pb = P.posB;
    const unsigned long long *ia = P.idA, *ib = P.idB;
    const double *va = P.vdwA, *vb = P.vdwB;
    unsigned long long off = FILL ? slot_base[s] : 0ull;
    uint32_t total = 0;
    for (uint32_t r = 0; r < S.rows; ++r) {
        const uint32_t ra = S.a0 + S.i0 + r;
        const D3 a = D3{pa[3 * ra], pa[3 * ra + 1], pa[3 * ra + 2]};
        const double vdw_a = (P.kind == MOLAR_HIP_SEARCH_DOUBLE_VDW) ? va[ra] : 0.0;
        bool found = false;                                       // WITHIN: first hit of the row only (:287-290)
        for (uint32_t j0 = 0; j0 < S.n2 && !found; j0 += 64u) {
            const uint32_t j = j0 + lane;
            bool hit = false;
            double d2 = 0.0;
            if (j < S.n2 && !(tri && j <= S.i0 + r)) {            // same cell: j in i+1..n (:443)
                const uint32_t rb = S.b0 + j;
                d2 = pair_d2(P, wrap, a, D3{pb[3 * rb], pb[3 * rb + 1], pb[3 * rb + 2]});
                if (P.kind == MOLAR_HIP_SEARCH_DOUBLE_VDW) {
                    const double cut = (vdw_a + vb[rb]) + 2.220446049250313e-16;        // :392, :423
                    hit = d2 <= cut * cut;
                } else {
                    hit = d2 <= P.cutoff2;
                }
            }
            const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
            if (!m) continue;
            if (P.kind == MOLAR_HIP_SEARCH_WITHIN) {
                if (FILL && lane == 0) out_i[off] = ia[ra];
                off += 1;
                total += 1;
                found = true;
                continue;
            }
            const uint32_t cnt = (uint32_t)__popcll(m);
            if (FILL && hit) {
                const unsigned long long at = off + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                out_i[at] = ia[ra];
                out_j[at] = ib[S.b0 + j];
                out_d[at] = sqrt(d2);                             // d2.sqrt() (:448)
            }
            off += cnt;
            total += cnt;
        }
    }
    if (!FILL && lane == 0) slot_cnt[s] = total;
}

inline dim3 grid_of(uint32_t n) {
    const unsigned gx = n < (1u << 20) ? (n ? n : 1u) : (1u << 20);
    return dim3(gx, (n + gx - 1u) / gx);
}

}  // namespace

namespace mh {
void search64_release(molar_hip_ctx *c) {
    if (!c->s64) return;
    for (DevBuf *b : {&c->s64->posA, &c->s64->idA, &c->s64->vdwA, &c->s64->posB, &c->s64->idB, &c->s64->vdwB, &c->s64->slots,
                      &c->s64->slot_cnt, &c->s64->slot_base, &c->s64->box, &c->s64->out_i, &c->s64->out_j, &c->s64->out_d})
        b->release();
    delete c->s64;
    c->s64 = nullptr;
}
}  // namespace mh

extern "C" {

int molar_hip_search_count_f64(molar_hip_ctx *c, const molar_hip_search_desc_f64 *q, uint64_t *out_count) {
    if (!c || !q) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "search_count_f64: null argument");
    MH_HIP(hipSetDevice(c->device));
    if (!c->s64) c->s64 = new molar_hip_search64_state;
    molar_hip_search64_state &Z = *c->s64;
    Z.have = false;
    const int kind = q->kind;
    if (kind < MOLAR_HIP_SEARCH_SINGLE || kind > MOLAR_HIP_SEARCH_DOUBLE_VDW)
        return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "search_count_f64: unknown kind %d", kind);
    const bool two = kind != MOLAR_HIP_SEARCH_SINGLE, vdw = kind == MOLAR_HIP_SEARCH_DOUBLE_VDW;
    if (vdw && (!q->vdw1 || !q->vdw2)) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "vdw search: radii pointer is null");
    HostSet A, B;
    MH_TRY(gather_set(q->xyz1, q->natoms1, q->idx1, q->n1, vdw ? q->vdw1 : nullptr, q->ids_local || vdw, A));
    if (two) MH_TRY(gather_set(q->xyz2, q->natoms2, q->idx2, q->n2, vdw ? q->vdw2 : nullptr, q->ids_local || vdw, B));
    BoxD box{};
    const bool use_box = q->box9 != nullptr;
    if (use_box) {
        std::vector<double> tb;
        const double *hb;
        MH_TRY(fetch_host(q->box9, 9, tb, &hb));
        MH_TRY(box64_from_matrix(hb, &box));
    }
    Z.kind = kind;
    Z.use_box = use_box;
    Z.pbc = use_box ? q->pbc : 0;
    Z.total = 0;
    Z.nslots = 0;
    Z.dims[0] = Z.dims[1] = Z.dims[2] = 1;
    double cutoff = q->cutoff;
    if (vdw) {
        if (A.id.empty() || B.id.empty()) {        // the reference unwrap()s an empty max: nothing to report here
            Z.have = true;
            if (out_count) *out_count = 0;
            return MOLAR_HIP_OK;
        }
        auto fmax_rust = [](const std::vector<double> &v) {      // Iterator::reduce(Float::max): NaN-ignoring max
            double m = v[0];
            for (size_t k = 1; k < v.size(); ++k) m = std::fmax(m, v[k]);
            return m;
        };
        cutoff = (fmax_rust(A.vdw) + fmax_rust(B.vdw)) + 2.220446049250313e-16;            // :781-783
    }
    if (!(cutoff > 0.0)) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "search_f64: cutoff must be positive (got %g)", cutoff);
    Z.cutoff = cutoff;
    double lower[3] = {0, 0, 0}, upper[3] = {0, 0, 0}, ext[3];
    if (use_box) {
        ext[0] = (box.m[0] + box.m[3]) + box.m[6];                                          // get_lab_extents (:369-375)
        ext[1] = (box.m[1] + box.m[4]) + box.m[7];
        ext[2] = (box.m[2] + box.m[5]) + box.m[8];
    } else {
        if (kind == MOLAR_HIP_SEARCH_WITHIN) {
            if (!q->lower3 || !q->upper3) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "non-periodic within search needs lower3/upper3");
            std::vector<double> tl, tu;
            const double *hl, *hu;
            MH_TRY(fetch_host(q->lower3, 3, tl, &hl));
            MH_TRY(fetch_host(q->upper3, 3, tu, &hu));
            for (int d = 0; d < 3; ++d) { lower[d] = hl[d]; upper[d] = hu[d]; }
        } else {
            bounding_box(cutoff, A, two ? &B : nullptr, lower, upper);
        }
        for (int d = 0; d < 3; ++d) ext[d] = upper[d] - lower[d];
    }
    uint64_t dims[3];
    double ncell_d = 1.0;
    for (int d = 0; d < 3; ++d) {                                                            // :103-110
        dims[d] = std::max<uint64_t>(as_usize(std::floor(ext[d] / cutoff)), 1);
        ncell_d *= (double)dims[d];
    }
    if (!(ncell_d <= 2.0e8)) return fail(MOLAR_HIP_ERR_TOO_LARGE, "search_f64: grid of %.3g cells (this untuned path builds it on the host)", ncell_d);
    std::memcpy(Z.dims, dims, sizeof dims);
    HostGrid G1, G2;
    build_grid(A, dims, use_box ? &box : nullptr, Z.pbc, lower, upper, G1);
    if (two) build_grid(B, dims, use_box ? &box : nullptr, Z.pbc, lower, upper, G2);
    const HostGrid &GB = two ? G2 : G1;

    // ---- search_plan (:217-269) cut into 64-row slots, in plan order then row order
    std::vector<Slot64> slots;
    auto add_task = [&](const HostGrid &ga, uint32_t ca, const HostGrid &gb, uint32_t cb, uint32_t wrap, bool tri) {
        const uint32_t n1 = ga.len(ca), n2 = gb.len(cb);
        if (n1 == 0 || n2 == 0) return;
        for (uint32_t i0 = 0; i0 < n1; i0 += 64u) {
            Slot64 s{};
            s.a0 = ga.start[ca];
            s.rows = std::min<uint32_t>(64u, n1 - i0);
            s.i0 = i0;
            s.b0 = gb.start[cb];
            s.n2 = n2;
            s.flags = wrap | (tri ? 0x100u : 0u);
            slots.push_back(s);
        }
    };
    for (uint64_t x = 0; x < dims[0]; ++x)
        for (uint64_t y = 0; y < dims[1]; ++y)
            for (uint64_t z = 0; z < dims[2]; ++z)
                for (int m = 0; m < 14; ++m) {
                    uint64_t cc[2][3] = {{x + MASKS64[m][0], y + MASKS64[m][1], z + MASKS64[m][2]},
                                         {x + MASKS64[m][3], y + MASKS64[m][4], z + MASKS64[m][5]}};
                    uint32_t wrap = 0;
                    bool skip = false;
                    for (int i = 0; i < 2 && !skip; ++i)
                        for (int d = 0; d < 3; ++d)
                            if (cc[i][d] == dims[d]) {
                                if ((Z.pbc >> d) & 1u) { cc[i][d] = 0; wrap |= 1u << d; }
                                else { skip = true; break; }                               // :241-244
                            }
                    if (skip) continue;
                    const uint32_t i1 = (uint32_t)(cc[0][0] + cc[0][1] * dims[0] + cc[0][2] * dims[0] * dims[1]);
                    const uint32_t i2 = (uint32_t)(cc[1][0] + cc[1][1] * dims[0] + cc[1][2] * dims[0] * dims[1]);
                    if (!two) {
                        add_task(G1, i1, G1, i2, wrap, i1 == i2);                           // :432-517
                    } else {
                        // (pair.0, pair.1) then (pair.1, pair.0): rows are always atoms of the FIRST set (:686-693)
                        add_task(G1, i1, G2, i2, wrap, false);
                        add_task(G1, i2, G2, i1, wrap, false);
                    }
                    if (slots.size() >= 0x7FFFFF00ull) return fail(MOLAR_HIP_ERR_TOO_LARGE, "search_f64: plan too large");
                }
    Z.nslots = (uint32_t)slots.size();
    Z.have = false;           // until the count below has gone through: a failed step must not leave a half-built cached search
    Z.total = 0;
    if (Z.nslots == 0) {
        Z.have = true;
        if (out_count) *out_count = 0;
        return MOLAR_HIP_OK;
    }
    // ---- upload and count
    auto up = [&](DevBuf &buf, const void *src, size_t bytes) -> int {
        MH_TRY(buf.reserve(bytes ? bytes : 8));
        if (bytes) MH_HIP(hipMemcpyAsync(buf.p, src, bytes, hipMemcpyHostToDevice, c->stream));
        return 0;
    };
    MH_TRY(up(Z.posA, G1.pos.data(), G1.pos.size() * 8));
    MH_TRY(up(Z.idA, G1.id.data(), G1.id.size() * 8));
    MH_TRY(up(Z.vdwA, G1.vdw.data(), G1.vdw.size() * 8));
    MH_TRY(up(Z.posB, GB.pos.data(), GB.pos.size() * 8));
    MH_TRY(up(Z.idB, GB.id.data(), GB.id.size() * 8));
    MH_TRY(up(Z.vdwB, GB.vdw.data(), GB.vdw.size() * 8));
    MH_TRY(up(Z.slots, slots.data(), slots.size() * sizeof(Slot64)));
    MH_TRY(up(Z.box, &box, sizeof box));
    MH_TRY(Z.slot_cnt.reserve((size_t)Z.nslots * 4));
    MH_TRY(Z.slot_base.reserve(((size_t)Z.nslots + 1) * 8));
    Params64 P{Z.posA.as<double>(), Z.posB.as<double>(), Z.vdwA.as<double>(), Z.vdwB.as<double>(),
               Z.idA.as<unsigned long long>(), Z.idB.as<unsigned long long>(), Z.box.as<BoxD>(), Z.slots.as<Slot64>(), Z.nslots,
               kind, use_box ? 1 : 0, cutoff * cutoff};
    hipLaunchKernelGGL((pair64_kernel<false>), grid_of(Z.nslots), dim3(64), 0, c->stream, P, Z.slot_cnt.as<uint32_t>(), nullptr, nullptr,
                       nullptr, nullptr);
    MH_HIP(hipGetLastError());
    std::vector<uint32_t> cnt(Z.nslots);
    MH_HIP(hipMemcpyAsync(cnt.data(), Z.slot_cnt.p, (size_t)Z.nslots * 4, hipMemcpyDeviceToHost, c->stream));
    MH_HIP(hipStreamSynchronize(c->stream));        // also: the host vectors uploaded above may go out of scope now
    std::vector<unsigned long long> base((size_t)Z.nslots + 1);
    unsigned long long run = 0;
    for (uint32_t s = 0; s < Z.nslots; ++s) { base[s] = run; run += cnt[s]; }
    base[Z.nslots] = run;
    MH_HIP(hipMemcpyAsync(Z.slot_base.p, base.data(), base.size() * 8, hipMemcpyHostToDevice, c->stream));
    MH_HIP(hipStreamSynchronize(c->stream));
    Z.total = run;
    Z.have = true;
    if (out_count) *out_count = run;
    return MOLAR_HIP_OK;
}

static int fill64(molar_hip_ctx *c, uint64_t *oi, uint64_t *oj, double *od, bool within) {
    if (!c || !c->s64 || !c->s64->have) return fail(MOLAR_HIP_ERR_NO_SEARCH, "no cached f64 search: call molar_hip_search_count_f64 first");
    molar_hip_search64_state &Z = *c->s64;
    if ((Z.kind == MOLAR_HIP_SEARCH_WITHIN) != within)
        return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, within ? "fill_ids_f64 is for within searches" : "within search yields ids: use molar_hip_search_fill_ids_f64");
    if (Z.total == 0 || Z.nslots == 0) return MOLAR_HIP_OK;
    MH_HIP(hipSetDevice(c->device));
    const size_t n = (size_t)Z.total;
    auto dev = [&](void *user, DevBuf &own, size_t bytes, void **out) -> int {
        if (user && is_device_ptr(user)) { *out = user; return 0; }
        MH_TRY(own.reserve(bytes));
        *out = own.p;
        return 0;
    };
    void *di, *dj = nullptr, *dd = nullptr;
    MH_TRY(dev(oi, Z.out_i, n * 8, &di));
    if (!within) {
        MH_TRY(dev(oj, Z.out_j, n * 8, &dj));
        MH_TRY(dev(od, Z.out_d, n * 8, &dd));
    }
    Params64 P{Z.posA.as<double>(), Z.posB.as<double>(), Z.vdwA.as<double>(), Z.vdwB.as<double>(),
               Z.idA.as<unsigned long long>(), Z.idB.as<unsigned long long>(), Z.box.as<BoxD>(), Z.slots.as<Slot64>(), Z.nslots,
               Z.kind, Z.use_box ? 1 : 0, Z.cutoff * Z.cutoff};
    hipLaunchKernelGGL((pair64_kernel<true>), grid_of(Z.nslots), dim3(64), 0, c->stream, P, nullptr, Z.slot_base.as<unsigned long long>(),
                       static_cast<unsigned long long *>(di), static_cast<unsigned long long *>(dj), static_cast<double *>(dd));
    MH_HIP(hipGetLastError());
    std::vector<RingJob> jobs;
    if (oi && di != oi) jobs.push_back(RingJob{di, n * 8, RING_COPY, oi, nullptr});
    if (oj && dj != oj) jobs.push_back(RingJob{dj, n * 8, RING_COPY, oj, nullptr});
    if (od && dd != od) jobs.push_back(RingJob{dd, n * 8, RING_COPY, od, nullptr});
    if (n * 8 * jobs.size() >= (24u << 20)) {
        MH_TRY(ring_to_host(c, jobs));              // large results: pinned ring + host threads (hoststream.hpp)
    } else {
        for (const RingJob &J : jobs) MH_HIP(hipMemcpyAsync(J.dst0, J.src, J.bytes, hipMemcpyDeviceToHost, c->stream));
    }
    MH_HIP(hipStreamSynchronize(c->stream));
    return MOLAR_HIP_OK;
}

int molar_hip_search_fill_f64(molar_hip_ctx *c, uint64_t *i, uint64_t *j, double *dist) { return fill64(c, i, j, dist, false); }

int molar_hip_search_fill_ids_f64(molar_hip_ctx *c, uint64_t *ids) { return fill64(c, ids, nullptr, nullptr, true); }

int molar_hip_search_grid_dims_f64(molar_hip_ctx *c, uint64_t dims[3]) {
    if (!c || !c->s64 || !c->s64->have || !dims) return fail(MOLAR_HIP_ERR_NO_SEARCH, "no cached f64 search");
    for (int d = 0; d < 3; ++d) dims[d] = c->s64->dims[d];
    return MOLAR_HIP_OK;
}

}  // extern "C"
