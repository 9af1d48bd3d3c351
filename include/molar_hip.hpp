// molar_hip.hpp — C++17 host side of the engine, above the C ABI (molar_hip.h).
//
// The reference's host code is Rust; this image has no Rust toolchain, so the compiled-language
// mirror of MolAR's interface for the accelerated path is this header.  Names, argument meaning
// and error behaviour follow the reference (paths under /root/reference/):
//   PbcDims, PBC_FULL/PBC_NONE, PeriodicBox            molar/src/periodic_box.rs:68-435
//   distance_search_{single,double,double_vdw}(_pbc),
//   distance_search_within(_pbc), DistanceSearchOutput molar/src/distance_search.rs:6-26,519-954
//   Measure / Modify methods on a bound selection      molar/src/measure.rs:20-482, modify.rs:15-63
//   rmsd, rmsd_mw, fit_transform(_at_origin)           molar/src/measure.rs:485-558
//   MeasureError / PeriodicBoxError / LipidOrderError  measure.rs:718-762, periodic_box.rs:131-144
//   AnalysisTask, AnalysisContext, TrajAnalysisArgs,
//   process_suffix, run()                              molar/src/analysis_task.rs:12-313
// File formats and the selection language are out of scope: frames come from a FrameSource.
// Header-only; link with -lmolar_hip (or dlopen it and pass the handle — see INTEGRATION.md).
#pragma once

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <optional>
#include <stdexcept>
#include <string>
#include <thread>
#include <tuple>
#include <utility>
#include <vector>

#include "molar_hip.h"

namespace molar {

using Float = float;          // aliases.rs:10-13
using usize = uint64_t;

struct Vector3f {
    Float x = 0, y = 0, z = 0;
    Float &operator[](int d) { return d == 0 ? x : (d == 1 ? y : z); }
    Float operator[](int d) const { return d == 0 ? x : (d == 1 ? y : z); }
};
using Pos = Vector3f;         // Point3<Float>, bit-compatible with xyzxyz... (io/xtc_handler.rs:89-92)
static_assert(sizeof(Pos) == 12, "Pos must be three packed floats");

// column-major 3x3 like nalgebra::Matrix3
struct Matrix3f {
    std::array<Float, 9> m{};
    Float &operator()(int r, int c) { return m[c * 3 + r]; }
    Float operator()(int r, int c) const { return m[c * 3 + r]; }
};

// ---------------------------------------------------------------- errors

struct MolarError : std::runtime_error {
    int code;
    MolarError(int c, const std::string &what) : std::runtime_error(what), code(c) {}
};
struct MeasureError : MolarError {          // measure.rs:732-762
    enum Kind { Sizes = 1, ZeroMass = 2, Svd = 3, Pbc = 4, LipidOrder = 7 };
    using MolarError::MolarError;
};
struct PeriodicBoxError : MolarError {      // periodic_box.rs:131-144
    enum Kind { NoPbc = 4, ZeroLengthVector = 5, InverseFailed = 6, AngleTooSmall = 10 };
    using MolarError::MolarError;
};
struct LipidOrderError : MolarError {       // measure.rs:718-729
    enum Kind { TailTooShort = 7, NormalsCount = 8, BondOrderCount = 9 };
    using MolarError::MolarError;
};

inline void check(int rc) {
    if (rc == MOLAR_HIP_OK) return;
    const std::string msg = molar_hip_last_error();
    if (rc >= 7 && rc <= 9) throw LipidOrderError(rc, msg);
    if (rc == 4 || rc == 5 || rc == 6 || rc == 10) throw PeriodicBoxError(rc, msg);
    if (rc >= 1 && rc <= 3) throw MeasureError(rc, msg);
    throw MolarError(rc, msg);
}

// ---------------------------------------------------------------- PbcDims / PeriodicBox

class PbcDims {                              // periodic_box.rs:68-123
    uint8_t v_ = 0;

   public:
    PbcDims() = default;
    explicit constexpr PbcDims(uint8_t raw) : v_(raw) {}
    static PbcDims make(bool x, bool y, bool z) {   // PbcDims::new
        PbcDims p;
        p.set_dim(0, x); p.set_dim(1, y); p.set_dim(2, z);
        return p;
    }
    void set_dim(size_t n, bool val) {
        if (n > 2) throw std::out_of_range("pbc has only 3 dimentions");
        if (val) v_ |= (uint8_t)(1u << n); else v_ &= (uint8_t)~(1u << n);
    }
    bool get_dim(size_t n) const {
        if (n > 2) throw std::out_of_range("pbc has only 3 dimentions");
        return (v_ & (1u << n)) != 0;
    }
    bool any() const { return (v_ & 7u) != 0; }
    uint8_t raw() const { return v_; }
    bool operator==(const PbcDims &o) const { return v_ == o.v_; }
    bool operator!=(const PbcDims &o) const { return v_ != o.v_; }
};
constexpr PbcDims PBC_FULL{7};               // periodic_box.rs:126
constexpr PbcDims PBC_NONE{0};               // periodic_box.rs:128

class PeriodicBox {                          // periodic_box.rs:15-23,146-435
    molar_hip_box b_{};

   public:
    static PeriodicBox from_matrix(const Matrix3f &m) {
        PeriodicBox p;
        check(molar_hip_box_from_matrix(m.m.data(), &p.b_));
        return p;
    }
    static PeriodicBox from_vectors_angles(Float a, Float b, Float c, Float alpha, Float beta, Float gamma) {
        PeriodicBox p;
        check(molar_hip_box_from_vectors_angles(a, b, c, alpha, beta, gamma, &p.b_));
        return p;
    }
    Matrix3f get_matrix() const {
        Matrix3f m;
        for (int k = 0; k < 9; ++k) m.m[k] = b_.m[k];
        return m;
    }
    const float *colmajor9() const { return b_.m; }
    size_t n_tric_corrections() const { return (size_t)b_.nshift; }
    Vector3f shortest_vector_dims(const Vector3f &v, PbcDims dims) const {
        Vector3f o;
        molar_hip_box_shortest_vector(&b_, &v.x, dims.raw(), &o.x);
        return o;
    }
    Vector3f shortest_vector(const Vector3f &v) const { return shortest_vector_dims(v, PBC_FULL); }
    Pos closest_image_dims(const Pos &p, const Pos &target, PbcDims dims) const {
        const Vector3f s = shortest_vector_dims({p.x - target.x, p.y - target.y, p.z - target.z}, dims);
        return {target.x + s.x, target.y + s.y, target.z + s.z};
    }
    Pos closest_image(const Pos &p, const Pos &target) const { return closest_image_dims(p, target, PBC_FULL); }
    Float distance_squared(const Pos &p1, const Pos &p2, PbcDims dims) const {
        const Vector3f s = shortest_vector_dims({p2.x - p1.x, p2.y - p1.y, p2.z - p1.z}, dims);
        return (s.x * s.x + s.y * s.y) + s.z * s.z;
    }
    Float distance(const Pos &p1, const Pos &p2, PbcDims dims) const { return std::sqrt(distance_squared(p1, p2, dims)); }
    Vector3f get_lab_extents() const {
        Vector3f o;
        molar_hip_box_lab_extents(&b_, &o.x);
        return o;
    }
    Vector3f get_box_extents() const {
        Vector3f o;
        molar_hip_box_extents(&b_, &o.x);
        return o;
    }
    Vector3f to_box_coords(const Vector3f &v) const {
        Vector3f o;
        molar_hip_box_to_box_coords(&b_, &v.x, &o.x);
        return o;
    }
    Vector3f to_lab_coords(const Vector3f &v) const {
        Vector3f o;
        molar_hip_box_to_lab_coords(&b_, &v.x, &o.x);
        return o;
    }
    bool is_inside(const Pos &p) const { return molar_hip_box_is_inside(&b_, &p.x) != 0; }
    Pos wrap_point(const Pos &p) const {
        Pos o;
        molar_hip_box_wrap_point(&b_, &p.x, &o.x);
        return o;
    }
    Vector3f wrap_vec(const Vector3f &v) const { return wrap_point(v); }
    bool is_triclinic() const {
        return b_.m[3] != 0 || b_.m[6] != 0 || b_.m[1] != 0 || b_.m[7] != 0 || b_.m[2] != 0 || b_.m[5] != 0;
    }
};

// ---------------------------------------------------------------- engine handle

class Engine {                               // molar_hip_ctx, one per thread / GPU
    molar_hip_ctx *ctx_;

   public:
    explicit Engine(int device = 0) : ctx_(molar_hip_create(device)) {
        if (!ctx_) throw MolarError(MOLAR_HIP_ERR_HIP, molar_hip_last_error());
    }
    ~Engine() { molar_hip_destroy(ctx_); }
    Engine(const Engine &) = delete;
    Engine &operator=(const Engine &) = delete;
    molar_hip_ctx *ctx() const { return ctx_; }
    static Engine &global() {                // process-wide default, like TprPlugin::get_cached
        static Engine e(0);
        return e;
    }
};

// ---------------------------------------------------------------- data model (only what the path reads)

struct State {                               // state.rs:22-28
    std::vector<Pos> coords;
    std::optional<PeriodicBox> pbox;
    Float time = 0;
    Float get_time() const { return time; }
};
struct Topology {                            // atom_storage.rs: SoA columns
    std::vector<Float> masses;
    std::vector<Float> vdw;
};
struct IsometryMatrix3 {                     // p -> R p + t
    Matrix3f R;
    Vector3f t;
};
enum class OrderType { Sz = 0, Scd = 1, ScdCorr = 2 };   // measure.rs:708-716

class System {                               // selection/system.rs: topology + current state
   public:
    Topology top;
    State state;
    System(Topology t, State s) : top(std::move(t)), state(std::move(s)) {
        if (top.masses.size() != state.coords.size())
            throw MolarError(MOLAR_HIP_ERR_INVALID_ARGUMENT, "topology and state sizes differ");
    }
    size_t len() const { return state.coords.size(); }
    void set_state(State s) {                // system.rs:230-236 (mem::replace, size checked)
        if (s.coords.size() != state.coords.size())
            throw MolarError(MOLAR_HIP_ERR_INVALID_ARGUMENT, "incompatible state size");
        state = std::move(s);
    }
};

// A selection bound to a System: sorted, non-empty index set (sel.rs:10-31) + Measure/Modify.
class SelBound {
    System *sys_;
    std::vector<usize> index_;
    Engine *eng_;

   public:
    SelBound(System &sys, std::vector<usize> index, Engine &eng = Engine::global()) : sys_(&sys), index_(std::move(index)), eng_(&eng) {
        if (index_.empty()) throw MolarError(MOLAR_HIP_ERR_INVALID_ARGUMENT, "selection is empty");   // sel.rs:13-19
    }
    static SelBound all(System &sys, Engine &eng = Engine::global()) {
        std::vector<usize> idx(sys.len());
        for (size_t k = 0; k < idx.size(); ++k) idx[k] = k;
        return SelBound(sys, std::move(idx), eng);
    }
    size_t len() const { return index_.size(); }
    const std::vector<usize> &get_index_slice() const { return index_; }   // IndexSliceProvider
    const float *coords_ptr() const { return &sys_->state.coords[0].x; }   // PosProvider (whole frame)
    float *coords_ptr_mut() { return &sys_->state.coords[0].x; }
    size_t natoms() const { return sys_->len(); }
    const float *masses() const { return sys_->top.masses.data(); }
    const PeriodicBox &require_box() const {                               // BoxProvider::require_box
        if (!sys_->state.pbox) throw PeriodicBoxError(PeriodicBoxError::NoPbc, "pbc operation withon periodic box");
        return *sys_->state.pbox;
    }
    System &system() const { return *sys_; }
    molar_hip_ctx *ctx() const { return eng_->ctx(); }

    // ---- Measure (measure.rs:20-482)
    std::pair<Pos, Pos> min_max() const {
        Pos lo, hi;
        check(molar_hip_min_max(ctx(), coords_ptr(), natoms(), index_.data(), index_.size(), &lo.x, &hi.x));
        return {lo, hi};
    }
    Pos center_of_geometry() const {
        Pos o;
        check(molar_hip_center_of_geometry(ctx(), coords_ptr(), natoms(), index_.data(), index_.size(), &o.x));
        return o;
    }
    Pos center_of_mass() const {
        Pos o;
        check(molar_hip_center_of_mass(ctx(), coords_ptr(), natoms(), index_.data(), index_.size(), masses(), &o.x));
        return o;
    }
    Pos center_of_geometry_pbc_dims(PbcDims dims) const {
        Pos o;
        check(molar_hip_center_of_geometry_pbc(ctx(), coords_ptr(), natoms(), index_.data(), index_.size(),
                                               require_box().colmajor9(), dims.raw(), &o.x));
        return o;
    }
    Pos center_of_geometry_pbc() const { return center_of_geometry_pbc_dims(PBC_FULL); }
    Pos center_of_mass_pbc_dims(PbcDims dims) const {
        Pos o;
        check(molar_hip_center_of_mass_pbc(ctx(), coords_ptr(), natoms(), index_.data(), index_.size(), masses(),
                                           require_box().colmajor9(), dims.raw(), &o.x));
        return o;
    }
    Pos center_of_mass_pbc() const { return center_of_mass_pbc_dims(PBC_FULL); }
    Float gyration() const {
        Float o;
        check(molar_hip_gyration(ctx(), coords_ptr(), natoms(), index_.data(), index_.size(), masses(), nullptr, &o));
        return o;
    }
    Float gyration_pbc() const {
        Float o;
        check(molar_hip_gyration(ctx(), coords_ptr(), natoms(), index_.data(), index_.size(), masses(),
                                 require_box().colmajor9(), &o));
        return o;
    }
    std::pair<Vector3f, Matrix3f> inertia() const {
        Vector3f m; Matrix3f a;
        check(molar_hip_inertia(ctx(), coords_ptr(), natoms(), index_.data(), index_.size(), masses(), nullptr, &m.x,
                                a.m.data(), nullptr));
        return {m, a};
    }
    std::pair<Vector3f, Matrix3f> inertia_pbc() const {
        Vector3f m; Matrix3f a;
        check(molar_hip_inertia(ctx(), coords_ptr(), natoms(), index_.data(), index_.size(), masses(),
                                require_box().colmajor9(), &m.x, a.m.data(), nullptr));
        return {m, a};
    }
    Float rmsd(const SelBound &other) const;
    std::vector<Float> lipid_tail_order(OrderType ot, const std::vector<Vector3f> &normals,
                                        const std::vector<uint8_t> &bond_orders) const {
        // size checks of measure.rs:281-291, same order, same errors
        if (len() < 3) throw LipidOrderError(LipidOrderError::TailTooShort, "tail should have at least 3 carbons");
        if (normals.size() != 1 && normals.size() != len() - 2)
            throw LipidOrderError(LipidOrderError::NormalsCount, "wrong number of normals");
        if (bond_orders.size() != len() - 1)
            throw LipidOrderError(LipidOrderError::BondOrderCount, "wrong number of bond orders");
        const uint64_t toff[2] = {0, len()}, noff[2] = {0, normals.size()};
        std::vector<Float> out(len() - 2);
        check(molar_hip_lipid_tail_order(ctx(), coords_ptr(), natoms(), index_.data(), toff, 1, (int)ot, &normals[0].x, noff,
                                         bond_orders.data(), out.data()));
        return out;
    }

    // ---- Modify (modify.rs:15-63)
    void apply_transform(const IsometryMatrix3 &tr) {
        check(molar_hip_apply_transform(ctx(), coords_ptr_mut(), natoms(), index_.data(), index_.size(), tr.R.m.data(), &tr.t.x));
    }
    void unwrap_simple_dim(PbcDims dims) {
        check(molar_hip_unwrap_simple(ctx(), coords_ptr_mut(), natoms(), index_.data(), index_.size(),
                                      require_box().colmajor9(), dims.raw()));
    }
    void unwrap_simple() { unwrap_simple_dim(PBC_FULL); }
    // Modify::unwrap_connectivity(_dim) (modify.rs:65-131): the connected groups as LOCAL index lists (the reference returns
    // them as selections of this selection)
    std::vector<std::vector<usize>> unwrap_connectivity_dim(Float cutoff, PbcDims dims) {
        const size_t n = index_.size();
        // (an empty index vector may hand out data() == NULL, which the C ABI reads as "all atoms": MolAR has no empty selections)
        if (n == 0) throw MolarError(MOLAR_HIP_ERR_INVALID_ARGUMENT, "unwrap_connectivity: empty selection");
        std::vector<uint64_t> off(n + 1), ids(n);
        size_t ng = 0;
        check(molar_hip_unwrap_connectivity(ctx(), coords_ptr_mut(), natoms(), index_.data(), n, require_box().colmajor9(), cutoff,
                                            dims.raw(), off.data(), ids.data(), &ng));
        std::vector<std::vector<usize>> out(ng);
        for (size_t g = 0; g < ng; ++g) out[g].assign(ids.begin() + off[g], ids.begin() + off[g + 1]);
        return out;
    }
    std::vector<std::vector<usize>> unwrap_connectivity(Float cutoff) { return unwrap_connectivity_dim(cutoff, PBC_FULL); }
    // SearchConnectivity::from_iter(distance_search_single_pbc(cutoff, self, 0..len, box, dims)) (connectivity.rs:19-35, modify.rs:77-78)
    // built on the device: conn[i] = neigh[offsets[i] .. offsets[i + 1]) in the reference's push order, local ids
    struct Connectivity {
        std::vector<uint64_t> offsets, neigh;
        size_t len() const { size_t k = 0; for (size_t i = 0; i + 1 < offsets.size(); ++i) k += offsets[i + 1] > offsets[i]; return k; }   // keys of the reference's map
        std::pair<const uint64_t *, const uint64_t *> get(usize i) const { return {neigh.data() + offsets[i], neigh.data() + offsets[i + 1]}; }
    };
    Connectivity search_connectivity(Float cutoff, PbcDims dims) const {
        molar_hip_search_desc d{};
        d.kind = MOLAR_HIP_SEARCH_SINGLE;
        d.cutoff = cutoff;
        d.xyz1 = coords_ptr(); d.natoms1 = natoms(); d.idx1 = index_.data(); d.n1 = index_.size();
        d.ids_local = 1;
        d.box9 = require_box().colmajor9();
        d.pbc = dims.raw();
        uint64_t rows = 0, ent = 0;
        check(molar_hip_search_connectivity(ctx(), &d, &rows, &ent));
        Connectivity c;
        c.offsets.resize(rows + 1);
        c.neigh.resize(ent);
        check(molar_hip_search_connectivity_fill(ctx(), c.offsets.data(), ent ? c.neigh.data() : nullptr));
        return c;
    }
    void translate(const Vector3f &shift) {                       // modify.rs:16-23
        check(molar_hip_translate(ctx(), coords_ptr_mut(), natoms(), index_.data(), index_.size(), &shift.x));
    }
    void rotate(const Vector3f &unit_axis, Float ang) {          // modify.rs:25-30, Rotation3::from_axis_angle
        check(molar_hip_rotate(ctx(), coords_ptr_mut(), natoms(), index_.data(), index_.size(), &unit_axis.x, ang));
    }

    // ---- principal axes (measure.rs:100-109, 246-257, 646-649): T(cm) * inverse(axes) * T(-cm)
    IsometryMatrix3 principal_transform() const { return principal(nullptr); }
    IsometryMatrix3 principal_transform_pbc() const { return principal(require_box().colmajor9()); }

   private:
    IsometryMatrix3 principal(const Float *box9) const {
        IsometryMatrix3 tr;
        check(molar_hip_principal_transform(ctx(), coords_ptr(), natoms(), index_.data(), index_.size(), masses(), box9,
                                            tr.R.m.data(), &tr.t.x));
        return tr;
    }
};

// ---- free functions of measure.rs:485-558
inline Float rmsd(const SelBound &s1, const SelBound &s2) {
    Float o;
    check(molar_hip_rmsd(s1.ctx(), s1.coords_ptr(), s1.natoms(), s1.get_index_slice().data(), s1.len(), s2.coords_ptr(),
                         s2.natoms(), s2.get_index_slice().data(), s2.len(), &o));
    return o;
}
inline Float SelBound::rmsd(const SelBound &other) const { return molar::rmsd(*this, other); }
inline Float rmsd_mw(const SelBound &s1, const SelBound &s2) {
    Float o;
    check(molar_hip_rmsd_mw(s1.ctx(), s1.coords_ptr(), s1.natoms(), s1.get_index_slice().data(), s1.len(), s1.masses(),
                            s2.coords_ptr(), s2.natoms(), s2.get_index_slice().data(), s2.len(), &o));
    return o;
}
inline IsometryMatrix3 fit_transform(const SelBound &s1, const SelBound &s2) {
    IsometryMatrix3 tr;
    check(molar_hip_fit_transform(s1.ctx(), s1.coords_ptr(), s1.natoms(), s1.get_index_slice().data(), s1.len(), s1.masses(),
                                  s2.coords_ptr(), s2.natoms(), s2.get_index_slice().data(), s2.len(), s2.masses(), 0,
                                  tr.R.m.data(), &tr.t.x));
    return tr;
}
inline IsometryMatrix3 fit_transform_at_origin(const SelBound &s1, const SelBound &s2) {
    IsometryMatrix3 tr;
    check(molar_hip_fit_transform(s1.ctx(), s1.coords_ptr(), s1.natoms(), s1.get_index_slice().data(), s1.len(), s1.masses(),
                                  s2.coords_ptr(), s2.natoms(), s2.get_index_slice().data(), s2.len(), s2.masses(), 1,
                                  tr.R.m.data(), &tr.t.x));
    return tr;
}

// The per-frame fit of an AnalysisTask whose States live in host memory (analysis_task.rs:245-252 hands them over frame by frame),
// at the rate the SELECTION crosses the link: molar_hip_fit_stream_*.  Built on the selection and the reference once;
//     auto t1 = fs.begin(state_k1.coords);  FitRecord r = fs.end(t0);      // up to three frames in flight
// every record is the one fit_transform + apply_transform + rmsd + center_of_mass + gyration give for that frame.
struct FitRecord {
    IsometryMatrix3 tr;          // fit_transform (measure.rs:507-522)
    Float rmsd = 0;              // of the fitted selection against the reference (:485-504)
    Vector3f com;                // center_of_mass of the fitted selection (:60-75)
    Float gyration = 0;          // (:78-87)
};
class FitStream {
    molar_hip_fit_stream *h_ = nullptr;

   public:
    // `sel`: the selection (its index and the topology's masses are taken now; its System supplies natoms); `reference`: the
    // selection fitted onto (coordinates taken now)
    FitStream(const SelBound &sel, const SelBound &reference, int host_threads = 0) {
        const auto i1 = sel.get_index_slice(), i2 = reference.get_index_slice();
        if (i1.size() != i2.size()) throw MolarError(MOLAR_HIP_ERR_SIZES, "incompatible sizes");      // MeasureError::Sizes (measure.rs:489)
        check(molar_hip_fit_stream_create(sel.ctx(), sel.natoms(), i1.data(), i1.size(), sel.masses(), reference.coords_ptr(),
                                          reference.natoms(), i2.data(), host_threads, &h_));
    }
    FitStream(const FitStream &) = delete;
    FitStream &operator=(const FitStream &) = delete;
    ~FitStream() { molar_hip_fit_stream_destroy(h_); }
    // apply: the fitted selection is written into `coords` at end() (Modify::apply_transform, modify.rs:32-36); the vector must stay
    // alive and untouched until then
    int32_t begin(std::vector<Pos> &coords, bool apply = false) {
        int32_t t = -1;
        check(molar_hip_fit_stream_begin(h_, &coords[0].x, apply ? 1 : 0, &t));
        return t;
    }
    FitRecord end(int32_t ticket) {
        FitRecord r;
        check(molar_hip_fit_stream_end(h_, ticket, &r.rmsd, r.tr.R.m.data(), &r.tr.t.x, &r.com.x, &r.gyration));
        return r;
    }
};

// ---------------------------------------------------------------- distance search (distance_search.rs)

// DistanceSearchOutput (:6-26): usize | (usize,usize) | (usize,usize,Float)
template <class T> struct DistanceSearchOutput;
template <> struct DistanceSearchOutput<usize> {
    static usize from_ijd(usize i, usize, Float) { return i; }
};
template <> struct DistanceSearchOutput<std::pair<usize, usize>> {
    static std::pair<usize, usize> from_ijd(usize i, usize j, Float) { return {i, j}; }
};
template <> struct DistanceSearchOutput<std::tuple<usize, usize, Float>> {
    static std::tuple<usize, usize, Float> from_ijd(usize i, usize j, Float d) { return {i, j, d}; }
};

namespace detail {
template <class T>
std::vector<T> run_search(molar_hip_ctx *ctx, const molar_hip_search_desc &d) {
    uint64_t n = 0;
    check(molar_hip_search_count(ctx, &d, &n));
    std::vector<T> out;
    out.reserve(n);
    if (d.kind == MOLAR_HIP_SEARCH_WITHIN) {
        std::vector<uint64_t> ids(n);
        check(molar_hip_search_fill_ids(ctx, ids.data()));
        for (uint64_t k = 0; k < n; ++k) out.push_back(DistanceSearchOutput<T>::from_ijd(ids[k], 0, 0));
        return out;
    }
    std::vector<uint64_t> i(n), j(n);
    std::vector<float> dist(n);
    check(molar_hip_search_fill_usize(ctx, i.data(), j.data(), dist.data()));
    for (uint64_t k = 0; k < n; ++k) out.push_back(DistanceSearchOutput<T>::from_ijd(i[k], j[k], dist[k]));
    return out;
}
inline molar_hip_search_desc desc(int kind, Float cutoff, const SelBound &s1, const SelBound *s2, bool ids_local,
                                  const PeriodicBox *box, PbcDims dims) {
    molar_hip_search_desc d{};
    d.kind = kind;
    d.cutoff = cutoff;
    d.xyz1 = s1.coords_ptr(); d.natoms1 = s1.natoms(); d.idx1 = s1.get_index_slice().data(); d.n1 = s1.len();
    if (s2) { d.xyz2 = s2->coords_ptr(); d.natoms2 = s2->natoms(); d.idx2 = s2->get_index_slice().data(); d.n2 = s2->len(); }
    d.ids_local = ids_local ? 1 : 0;
    d.box9 = box ? box->colmajor9() : nullptr;
    d.pbc = dims.raw();
    return d;
}
}  // namespace detail

// Device-resident result of a search (engine extension, no reference counterpart): `count` pairs as (u32 i, u32 j)
// and f32 distances in engine-owned HIP memory, in the reference's order, valid until the next search on the
// context; one host round trip per call (molar_hip_search_resident).
struct ResidentPairs {
    uint64_t count = 0;
    const uint32_t *pairs = nullptr;   // device pointer, count x 2
    const float *dist = nullptr;       // device pointer, count
};
inline ResidentPairs distance_search_single_pbc_resident(Float cutoff, const SelBound &data, const PeriodicBox &pbox,
                                                         PbcDims pbc_dims, bool ids_local = false) {
    const molar_hip_search_desc d = detail::desc(MOLAR_HIP_SEARCH_SINGLE, cutoff, data, nullptr, ids_local, &pbox, pbc_dims);
    ResidentPairs r;
    check(molar_hip_search_resident(data.ctx(), &d, &r.count, &r.pairs, &r.dist));
    return r;
}
inline ResidentPairs distance_search_double_pbc_resident(Float cutoff, const SelBound &d1, const SelBound &d2,
                                                         const PeriodicBox &pbox, PbcDims pbc_dims, bool ids_local = false) {
    const molar_hip_search_desc d = detail::desc(MOLAR_HIP_SEARCH_DOUBLE, cutoff, d1, &d2, ids_local, &pbox, pbc_dims);
    ResidentPairs r;
    check(molar_hip_search_resident(d1.ctx(), &d, &r.count, &r.pairs, &r.dist));
    return r;
}

// Per-frame loop with two searches in flight (molar_hip_search_resident_begin / _end, engine extension):
//     PairPipeline pipe(ctx);
//     for (frame : trajectory) { if (auto r = pipe.push(rc, sel_of(frame), box, PBC_FULL)) consume(*r); }
//     if (auto r = pipe.finish()) consume(*r);
// push() enqueues the search of its frame and hands back the result of the frame pushed before it, so the host
// work and the result round trip of one frame hide behind the kernels of the other.  A result stays valid until
// the second push after the one that produced it.  The selection's coordinates and index must stay alive and
// unchanged until its result has been handed back (the box matrix is copied).
class PairPipeline {
  public:
    explicit PairPipeline(molar_hip_ctx *ctx) : ctx_(ctx) {}
    PairPipeline(const PairPipeline &) = delete;
    PairPipeline &operator=(const PairPipeline &) = delete;
    ~PairPipeline() {                                   // never leave a ticket pending on the context
        if (pending_) { uint64_t n; const uint32_t *p; const float *d; (void)molar_hip_search_resident_end(ctx_, ticket_, &n, &p, &d); }
    }
    std::optional<ResidentPairs> push(Float cutoff, const SelBound &data, const PeriodicBox &pbox, PbcDims pbc_dims,
                                      bool ids_local = false) {
        const int slot = next_;
        molar_hip_search_desc &d = desc_[slot];
        d = detail::desc(MOLAR_HIP_SEARCH_SINGLE, cutoff, data, nullptr, ids_local, &pbox, pbc_dims);
        std::memcpy(box_[slot], pbox.colmajor9(), sizeof box_[slot]);
        d.box9 = box_[slot];
        int32_t t = -1;
        check(molar_hip_search_resident_begin(ctx_, &d, &t));
        next_ ^= 1;
        std::optional<ResidentPairs> out = finish();
        ticket_ = t;
        pending_ = true;
        return out;
    }
    std::optional<ResidentPairs> finish() {
        if (!pending_) return std::nullopt;
        pending_ = false;
        ResidentPairs r;
        check(molar_hip_search_resident_end(ctx_, ticket_, &r.count, &r.pairs, &r.dist));
        return r;
    }

  private:
    molar_hip_ctx *ctx_;
    molar_hip_search_desc desc_[2]{};
    float box_[2][9]{};
    int next_ = 0;
    int32_t ticket_ = -1;
    bool pending_ = false;
};

// Histogram1D (molar_membrane/src/stats.rs:13-55) with the engine's consumer-fused search as a feed: every distance
// of a search stream goes through add_one's binning rule on the GPU without the pairs ever being written
// (molar_hip_search_histogram - the shape of a radial distribution over a trajectory).  Counts are integers, so frames,
// ranks and the host's add_one sum exactly; bins() gives them as the reference's Float counters (which stop growing
// at 2^24 per bin - the integer counts do not).
class Histogram1D {
  public:
    Histogram1D(Float min, Float max, usize n_bins) : min_(min), max_(max), counts_(n_bins, 0) {}   // stats.rs:20-27
    void add_one(Float val) {                                                                        // stats.rs:29-35
        const Float fb = std::floor(static_cast<Float>(counts_.size()) * (val - min_) / (max_ - min_));
        if (fb != fb) { if (!counts_.empty()) counts_[0] += 1; return; }                             // NaN as isize == 0
        if (fb >= 0 && fb < static_cast<Float>(counts_.size())) counts_[static_cast<size_t>(fb)] += 1;
    }
    // the distance stream of distance_search_single_pbc / _double_pbc (:892-954, :659-754), binned on the GPU
    uint64_t add_distances_single_pbc(Float cutoff, const SelBound &data, const PeriodicBox &pbox, PbcDims pbc_dims) {
        const molar_hip_search_desc d = detail::desc(MOLAR_HIP_SEARCH_SINGLE, cutoff, data, nullptr, false, &pbox, pbc_dims);
        return feed(data.ctx(), d);
    }
    uint64_t add_distances_double_pbc(Float cutoff, const SelBound &d1, const SelBound &d2, const PeriodicBox &pbox,
                                      PbcDims pbc_dims) {
        const molar_hip_search_desc d = detail::desc(MOLAR_HIP_SEARCH_DOUBLE, cutoff, d1, &d2, false, &pbox, pbc_dims);
        return feed(d1.ctx(), d);
    }
    // the same stream over frames [first, first + count) of an XTC trajectory in one call (molar_hip_xtc_histogram): decode on host
    // threads and the fused histogram of 16-frame windows overlap inside it; every frame's own box; `index` empty = all atoms
    void add_distances_trajectory_single_pbc(Engine &eng, const class XtcReader &traj, size_t first, size_t count, Float cutoff,
                                             const std::vector<usize> &index, PbcDims pbc_dims, int decode_threads = 0);
    // ... and between two selections of every frame (distance_search_double_pbc; molar_hip_xtc_histogram_double)
    void add_distances_trajectory_double_pbc(Engine &eng, const class XtcReader &traj, size_t first, size_t count, Float cutoff,
                                             const std::vector<usize> &index1, const std::vector<usize> &index2, PbcDims pbc_dims,
                                             int decode_threads = 0);
    const std::vector<uint64_t> &counts() const { return counts_; }
    // adds another histogram of the same shape bin by bin (the integer reduction at the end of a frame-parallel run:
    // AnalysisTask::run_sharded, or one all_reduce of these counters across ranks)
    void merge(const Histogram1D &o) {
        if (o.counts_.size() != counts_.size() || o.min_ != min_ || o.max_ != max_)
            throw MolarError(MOLAR_HIP_ERR_INVALID_ARGUMENT, "Histogram1D::merge: different binning");
        for (size_t b = 0; b < counts_.size(); ++b) counts_[b] += o.counts_[b];
    }
    std::vector<Float> bins() const { return std::vector<Float>(counts_.begin(), counts_.end()); }
    std::vector<Float> normalized_density() const {                                                  // stats.rs:37-43
        const Float d = (max_ - min_) / static_cast<Float>(counts_.size());
        Float sum = 0;
        for (uint64_t c : counts_) sum += static_cast<Float>(c);
        std::vector<Float> out(counts_.size());
        for (size_t b = 0; b < out.size(); ++b) out[b] = static_cast<Float>(counts_[b]) / (sum * d);
        return out;
    }
    Float bin_center(usize b) const {                                                                // stats.rs:45-55
        const Float d = (max_ - min_) / static_cast<Float>(counts_.size());
        return min_ + static_cast<Float>(b) * d + Float(0.5) * d;
    }

  private:
    uint64_t feed(molar_hip_ctx *ctx, const molar_hip_search_desc &d) {
        uint64_t n = 0;
        check(molar_hip_search_histogram(ctx, &d, min_, max_, counts_.size(), counts_.data(), &n));
        return n;
    }
    Float min_, max_;
    std::vector<uint64_t> counts_;
};

// ids: the reference takes an iterator; the two uses are the selection's own indices
// (sel.iter_index(), ids_local = false) and 0..n (modify.rs:78, ids_local = true).
template <class T>
std::vector<T> distance_search_single(Float cutoff, const SelBound &data, bool ids_local = false) {            // :892-915
    return detail::run_search<T>(data.ctx(), detail::desc(MOLAR_HIP_SEARCH_SINGLE, cutoff, data, nullptr, ids_local, nullptr, PBC_NONE));
}
template <class T>
std::vector<T> distance_search_single_pbc(Float cutoff, const SelBound &data, const PeriodicBox &pbox, PbcDims pbc_dims,
                                          bool ids_local = false) {                                            // :928-954
    return detail::run_search<T>(data.ctx(), detail::desc(MOLAR_HIP_SEARCH_SINGLE, cutoff, data, nullptr, ids_local, &pbox, pbc_dims));
}
template <class T>
std::vector<T> distance_search_double(Float cutoff, const SelBound &d1, const SelBound &d2, bool ids_local = false) {   // :659-698
    return detail::run_search<T>(d1.ctx(), detail::desc(MOLAR_HIP_SEARCH_DOUBLE, cutoff, d1, &d2, ids_local, nullptr, PBC_NONE));
}
template <class T>
std::vector<T> distance_search_double_pbc(Float cutoff, const SelBound &d1, const SelBound &d2, const PeriodicBox &pbox,
                                          PbcDims pbc_dims, bool ids_local = false) {                          // :713-754
    return detail::run_search<T>(d1.ctx(), detail::desc(MOLAR_HIP_SEARCH_DOUBLE, cutoff, d1, &d2, ids_local, &pbox, pbc_dims));
}
template <class T>
std::vector<T> distance_search_double_vdw(const SelBound &d1, const SelBound &d2, const std::vector<Float> &vdw1,
                                          const std::vector<Float> &vdw2) {                                    // :767-814 (local ids)
    auto d = detail::desc(MOLAR_HIP_SEARCH_DOUBLE_VDW, 0, d1, &d2, true, nullptr, PBC_NONE);
    d.vdw1 = vdw1.data(); d.vdw2 = vdw2.data();
    return detail::run_search<T>(d1.ctx(), d);
}
template <class T>
std::vector<T> distance_search_double_vdw_pbc(const SelBound &d1, const SelBound &d2, const std::vector<Float> &vdw1,
                                              const std::vector<Float> &vdw2, const PeriodicBox &pbox, PbcDims pbc_dims) {   // :829-879
    auto d = detail::desc(MOLAR_HIP_SEARCH_DOUBLE_VDW, 0, d1, &d2, true, &pbox, pbc_dims);
    d.vdw1 = vdw1.data(); d.vdw2 = vdw2.data();
    return detail::run_search<T>(d1.ctx(), d);
}
inline std::vector<usize> distance_search_within(Float cutoff, const SelBound &d1, const SelBound &d2, const Vector3f &lower,
                                                 const Vector3f &upper) {                                      // :519-558
    auto d = detail::desc(MOLAR_HIP_SEARCH_WITHIN, cutoff, d1, &d2, false, nullptr, PBC_NONE);
    d.lower3 = &lower.x; d.upper3 = &upper.x;
    return detail::run_search<usize>(d1.ctx(), d);
}
inline std::vector<usize> distance_search_within_pbc(Float cutoff, const SelBound &d1, const SelBound &d2,
                                                     const PeriodicBox &pbox, PbcDims pbc_dims) {              // :560-598
    return detail::run_search<usize>(d1.ctx(), detail::desc(MOLAR_HIP_SEARCH_WITHIN, cutoff, d1, &d2, false, &pbox, pbc_dims));
}

// `within <cutoff> [pbc] of <inner>` as the selection keeps it (LogicalNode::Within, selection/ast.rs:589-631): the sorted,
// de-duplicated ids of the atoms of d1 with an atom of d2 within the cutoff - SortedSet::from_unsorted of the stream the
// two functions above return (selection_expr.rs:112), computed without the stream (molar_hip_within_count / _fill).
// lower / upper: the non-periodic form's extents (ast.rs:597-602); pbox: the periodic form.
inline std::vector<usize> within_set(Float cutoff, const SelBound &d1, const SelBound &d2, const PeriodicBox *pbox, PbcDims pbc_dims,
                                     const Vector3f *lower = nullptr, const Vector3f *upper = nullptr) {
    auto d = detail::desc(MOLAR_HIP_SEARCH_WITHIN, cutoff, d1, &d2, false, pbox, pbc_dims);
    if (lower && upper) { d.lower3 = &lower->x; d.upper3 = &upper->x; }
    uint64_t n = 0;
    check(molar_hip_within_count(d1.ctx(), &d, &n));
    std::vector<uint64_t> ids(n);
    if (n) check(molar_hip_within_fill(d1.ctx(), ids.data()));
    return std::vector<usize>(ids.begin(), ids.end());
}

// ---------------------------------------------------------------- bilayer frames (molar_membrane/src/lib.rs:410-454)
// One frame of Membrane::compute per push, chained on the engine's stream (molar_hip_membrane_frame_begin / _end):
// unwrap -> markers -> patches -> initial normals -> smoothing -> tail order, two frames in flight, the valid flags of
// the lipids carried from frame to frame like LipidMolecule::valid.
//     MembraneFrames mem(eng, desc);
//     for (frame : trajectory) { if (auto v = mem.push(frame.xyz, frame.box)) consume(*v, mem); }
//     if (auto v = mem.finish()) consume(*v, mem);
// push() hands back the view (device addresses, sizes) of the frame pushed before; fetch() copies chosen arrays of that
// frame to host memory.  A view stays valid until the second push after the one that produced it.
class MembraneFrames {
  public:
    MembraneFrames(Engine &eng, const molar_hip_membrane_desc &desc) { check(molar_hip_membrane_plan_create(eng.ctx(), &desc, &plan_)); }
    MembraneFrames(const MembraneFrames &) = delete;
    MembraneFrames &operator=(const MembraneFrames &) = delete;
    ~MembraneFrames() { molar_hip_membrane_plan_destroy(plan_); }      // waits for the frames in flight
    // reset_valid_lipids (lib.rs:269-273) with nullptr, or the caller's flags; no frame may be in flight
    void set_valid(const uint8_t *valid) { check(molar_hip_membrane_plan_set_valid(plan_, valid)); }
    // xyz: float[natoms][3] in device memory (unwrapped in place, read until the frame has been handed back) or host memory
    std::optional<molar_hip_membrane_view> push(float *xyz, const PeriodicBox &pbox) {
        int32_t t = -1;
        check(molar_hip_membrane_frame_begin(plan_, xyz, pbox.colmajor9(), &t));
        std::optional<molar_hip_membrane_view> out = finish();
        ticket_ = t;
        pending_ = true;
        return out;
    }
    std::optional<molar_hip_membrane_view> finish() {
        if (!pending_) return std::nullopt;
        pending_ = false;
        molar_hip_membrane_view v{};
        last_ = ticket_;
        check(molar_hip_membrane_frame_end(plan_, ticket_, &v));
        return v;
    }
    // finish() with the per-lipid arrays of that frame brought to the host in the same wait (molar_hip_membrane_frame_end_fetch:
    // no patch-sized arrays here)
    std::optional<molar_hip_membrane_view> finish(const molar_hip_membrane_out &out) {
        if (!pending_) return std::nullopt;
        pending_ = false;
        molar_hip_membrane_view v{};
        last_ = ticket_;
        check(molar_hip_membrane_frame_end_fetch(plan_, ticket_, &v, &out));
        return v;
    }
    // push() whose handed-back frame comes with its per-lipid arrays, like finish(out)
    std::optional<molar_hip_membrane_view> push(float *xyz, const PeriodicBox &pbox, const molar_hip_membrane_out &out) {
        int32_t t = -1;
        check(molar_hip_membrane_frame_begin(plan_, xyz, pbox.colmajor9(), &t));
        std::optional<molar_hip_membrane_view> v = finish(out);
        ticket_ = t;
        pending_ = true;
        return v;
    }
    // arrays of the frame handed back last (null members of `out` are skipped)
    void fetch(const molar_hip_membrane_out &out) { check(molar_hip_membrane_frame_fetch(plan_, last_, &out)); }

  private:
    molar_hip_membrane_plan *plan_ = nullptr;
    int32_t ticket_ = -1, last_ = -1;
    bool pending_ = false;
};

// ---------------------------------------------------------------- analysis task driver (analysis_task.rs)

struct AnalysisError : MolarError {          // analysis_task.rs:40-75
    enum Kind { ParseFloat = 201, ParseInt = 202, InvalidSuffix = 203, NoFramesConsumed = 204, PreProcess = 205,
                ProcessFrame = 206, PostProcess = 207, Arg = 208, NoTraj = 209 };
    using MolarError::MolarError;
};

struct TrajAnalysisArgs {                    // analysis_task.rs:12-38
    std::vector<std::string> files;
    size_t log = 100;
    std::string begin = "0";
    std::string end = "";
    size_t skip = 1;
    bool use_struct_file = false;
    std::vector<std::string> rest;           // arguments the task itself consumes (A::augment_args)

    static TrajAnalysisArgs parse(const std::vector<std::string> &argv) {
        TrajAnalysisArgs a;
        bool have_files = false;
        auto need = [&](size_t k) -> const std::string & {
            if (k >= argv.size()) throw AnalysisError(AnalysisError::Arg, "argument parsing: missing value for " + argv[k - 1]);
            return argv[k];
        };
        for (size_t k = 0; k < argv.size(); ++k) {
            const std::string &s = argv[k];
            if (s == "-f" || s == "--files") {
                have_files = true;
                while (k + 1 < argv.size() && argv[k + 1].rfind("-", 0) != 0) a.files.push_back(argv[++k]);
            } else if (s == "--log") {
                a.log = std::stoull(need(++k));
            } else if (s == "-b" || s == "--begin") {
                a.begin = need(++k);
            } else if (s == "-e" || s == "--end") {
                a.end = need(++k);
            } else if (s == "--skip") {
                a.skip = std::stoull(need(++k));
                if (a.skip < 1) throw AnalysisError(AnalysisError::Arg, "argument parsing: --skip must be >= 1");
            } else if (s == "--use_struct_file") {
                a.use_struct_file = true;
            } else {
                a.rest.push_back(s);
            }
        }
        if (!have_files || a.files.empty()) throw AnalysisError(AnalysisError::Arg, "argument parsing: -f/--files is required");
        return a;
    }
};

// process_suffix (analysis_task.rs:82-110): "" -> no limit; bare number / "fr" -> frame; ps/ns/us -> time in ps
inline std::pair<std::optional<size_t>, std::optional<Float>> process_suffix(const std::string &in) {
    const auto b = in.find_first_not_of(" \t\n\r");
    if (b == std::string::npos) return {std::nullopt, std::nullopt};
    const std::string s = in.substr(b, in.find_last_not_of(" \t\n\r") - b + 1);
    auto trim = [](std::string t) {
        const auto x = t.find_first_not_of(" \t");
        if (x == std::string::npos) return std::string();
        return t.substr(x, t.find_last_not_of(" \t") - x + 1);
    };
    auto parse_usize = [](const std::string &t, size_t &out) {
        if (t.empty()) return false;
        for (char ch : t)
            if (ch < '0' || ch > '9') return false;
        try { out = std::stoull(t); } catch (...) { return false; }
        return true;
    };
    auto ends_with = [&](const char *suf) { const std::string u(suf); return s.size() >= u.size() && s.compare(s.size() - u.size(), u.size(), u) == 0; };
    size_t fr;
    if (parse_usize(s, fr)) return {fr, std::nullopt};
    if (ends_with("fr")) {
        if (!parse_usize(trim(s.substr(0, s.size() - 2)), fr)) throw AnalysisError(AnalysisError::ParseInt, "invalid digit found in string");
        return {fr, std::nullopt};
    }
    const std::pair<const char *, Float> units[] = {{"ps", 1.0f}, {"ns", 1000.0f}, {"us", 1000000.0f}};
    for (const auto &u : units)
        if (ends_with(u.first)) {
            const std::string num = trim(s.substr(0, s.size() - 2));
            size_t used = 0;
            Float v;
            try { v = std::stof(num, &used); } catch (...) { throw AnalysisError(AnalysisError::ParseFloat, "invalid float literal"); }
            if (used != num.size()) throw AnalysisError(AnalysisError::ParseFloat, "invalid float literal");
            return {std::nullopt, v * u.second};
        }
    throw AnalysisError(AnalysisError::InvalidSuffix, "invalid time suffix, 'fr', 'ps', 'ns', 'us' allowed");
}

// Where frames come from (the reference opens files; IO formats are out of scope here).
struct FrameSource {
    virtual ~FrameSource() = default;
    virtual Topology read_topology(const std::string &structure_file) = 0;
    virtual State read_structure_state(const std::string &structure_file) = 0;
    // Opens trajectory `file`; the returned callable yields frames until it returns nullopt.
    // `skip_to_frame` / `skip_to_time` (io.rs random access) are applied before the first call.
    virtual std::function<std::optional<State>()> open(const std::string &file, std::optional<size_t> skip_to_frame,
                                                       std::optional<Float> skip_to_time) = 0;
    // For the frame-parallel driver (AnalysisTask::run_sharded).  A source that can tell how many frames a trajectory file holds
    // without reading it through (an indexed file: the XTC index of io.rs:691-760) and whose open() may be called from several
    // threads at once lets every worker read its OWN block of frames through its own reader; the defaults keep the one reader.
    virtual std::optional<size_t> frame_count(const std::string & /*file*/) { return std::nullopt; }
    virtual bool concurrent_open() const { return false; }
};

// XTC trajectory through the engine's decoder (molar/src/io/xtc_handler.rs:64-112, 200-229): read_state,
// seek_frame, seek_time; `open_as_source` adapts it to FrameSource::open.
class XtcReader {
    molar_hip_xtc *h_;
    size_t cur_fr_ = 0;

   public:
    explicit XtcReader(const std::string &path) : h_(molar_hip_xtc_open(path.c_str())) {
        if (!h_) throw MolarError(MOLAR_HIP_ERR_IO, molar_hip_last_error());
    }
    XtcReader(const void *data, size_t bytes) : h_(molar_hip_xtc_open_memory(data, bytes)) {
        if (!h_) throw MolarError(MOLAR_HIP_ERR_IO, molar_hip_last_error());
    }
    XtcReader(const XtcReader &) = delete;
    XtcReader &operator=(const XtcReader &) = delete;
    ~XtcReader() { molar_hip_xtc_close(h_); }
    size_t nframes() const { return molar_hip_xtc_nframes(h_); }
    size_t natoms() const { return molar_hip_xtc_natoms(h_); }
    const molar_hip_xtc *handle() const { return h_; }
    size_t current_frame() const { return cur_fr_; }
    void seek_frame(size_t fr) {
        if (fr > nframes()) throw MolarError(MOLAR_HIP_ERR_IO, "seek to frame failed");
        cur_fr_ = fr;
    }
    void seek_time(Float t) { check(molar_hip_xtc_seek_time(h_, t, &cur_fr_)); }
    std::optional<State> read_state() {      // nullopt = FileFormatError::Eof
        if (cur_fr_ >= nframes()) return std::nullopt;
        int32_t nat = 0;
        float time = 0, box9[9];
        check(molar_hip_xtc_frame_info(h_, cur_fr_, &nat, nullptr, &time, box9, nullptr));
        State st;
        st.coords.resize((size_t)nat);
        check(molar_hip_xtc_read(nullptr, h_, cur_fr_, 1, &st.coords[0].x, 1));
        Matrix3f m;
        for (int k = 0; k < 9; ++k) m.m[k] = box9[k];
        st.pbox = PeriodicBox::from_matrix(m);
        st.time = time;
        ++cur_fr_;
        return st;
    }
    // `count` consecutive frames decoded on `nthreads` host threads into xyz (host or device memory)
    void read_frames(size_t first, size_t count, float *xyz, Engine *eng = nullptr, int nthreads = 0) {
        check(molar_hip_xtc_read(eng ? eng->ctx() : nullptr, h_, first, count, xyz, nthreads));
    }
    static std::function<std::optional<State>()> open_as_source(const std::string &file, std::optional<size_t> skip_to_frame,
                                                                std::optional<Float> skip_to_time) {
        auto r = std::make_shared<XtcReader>(file);
        if (skip_to_frame) r->seek_frame(*skip_to_frame);
        if (skip_to_time) r->seek_time(*skip_to_time);
        return [r]() { return r->read_state(); };
    }
};

inline void Histogram1D::add_distances_trajectory_single_pbc(Engine &eng, const XtcReader &traj, size_t first, size_t count, Float cutoff,
                                                             const std::vector<usize> &index, PbcDims pbc_dims, int decode_threads) {
    check(molar_hip_xtc_histogram(eng.ctx(), traj.handle(), first, count, index.empty() ? nullptr : index.data(), index.size(), cutoff,
                                  pbc_dims.raw(), min_, max_, counts_.size(), counts_.data(), decode_threads));
}

inline void Histogram1D::add_distances_trajectory_double_pbc(Engine &eng, const XtcReader &traj, size_t first, size_t count, Float cutoff,
                                                             const std::vector<usize> &index1, const std::vector<usize> &index2,
                                                             PbcDims pbc_dims, int decode_threads) {
    check(molar_hip_xtc_histogram_double(eng.ctx(), traj.handle(), first, count, index1.empty() ? nullptr : index1.data(), index1.size(),
                                         index2.empty() ? nullptr : index2.data(), index2.size(), cutoff, pbc_dims.raw(), min_, max_,
                                         counts_.size(), counts_.data(), decode_threads));
}

// XTC writer: FileFormatHandler::create + write_state (xtc_handler.rs:54-62, 117-168); frames are appended to one file.
class XtcWriter {
    std::FILE *f_;
    Float precision_;
    size_t nframes_ = 0;
    std::vector<uint8_t> buf_;

   public:
    explicit XtcWriter(const std::string &path, Float precision = 1000.0f) : f_(std::fopen(path.c_str(), "wb")), precision_(precision) {
        if (!f_) throw MolarError(MOLAR_HIP_ERR_IO, "cannot create " + path);
    }
    XtcWriter(const XtcWriter &) = delete;
    XtcWriter &operator=(const XtcWriter &) = delete;
    ~XtcWriter() {
        if (f_) std::fclose(f_);
    }
    size_t nframes() const { return nframes_; }
    void write_state(const State &st) {
        static const float zero9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        buf_.resize(96 + 16 * st.coords.size());
        size_t len = 0;
        check(molar_hip_xtc_encode_frame(st.coords.empty() ? nullptr : &st.coords[0].x, st.coords.size(), st.pbox ? st.pbox->colmajor9() : zero9,
                                         (int32_t)nframes_, st.time, precision_, buf_.data(), buf_.size(), &len));
        if (std::fwrite(buf_.data(), 1, len, f_) != len) throw MolarError(MOLAR_HIP_ERR_IO, "failed to write frame");      // XtcHandlerError::WriteFrame
        ++nframes_;
    }
};

// FrameSource over XTC trajectories (FileHandler::open + the state iterator, io.rs:198-271, xtc_handler.rs:64-112) with the
// topology and the structure file's state handed in - structure formats stay in MolAR.  Indexed, so frame_count is known and
// every open() is its own reader: AnalysisTask::run_sharded gives each worker one.
class XtcFrameSource : public FrameSource {
    Topology top_;
    State structure_;

   public:
    XtcFrameSource(Topology top, State structure_state) : top_(std::move(top)), structure_(std::move(structure_state)) {}
    Topology read_topology(const std::string &) override { return top_; }
    State read_structure_state(const std::string &) override { return structure_; }
    std::function<std::optional<State>()> open(const std::string &file, std::optional<size_t> skip_to_frame,
                                               std::optional<Float> skip_to_time) override {
        return XtcReader::open_as_source(file, skip_to_frame, skip_to_time);
    }
    std::optional<size_t> frame_count(const std::string &file) override { return XtcReader(file).nframes(); }
    bool concurrent_open() const override { return true; }
};

template <class A>
struct AnalysisContext {                     // analysis_task.rs:309-313
    System sys;
    size_t consumed_frames = 0;
    A args;
    // additions for the frame-parallel form (run_sharded): the engine context this instance computes on (bind selections
    // with `SelBound(ctx.sys, idx, ctx.eng())`; run() leaves it at the process-wide default) and the position of the
    // current frame among the consumed frames of the whole run (what per-frame results are tagged with)
    Engine *engine = nullptr;
    Engine &eng() const { return engine ? *engine : Engine::global(); }
    size_t frame_index = 0;
};

// AnalysisTask<A> (analysis_task.rs:113-123).  A must be constructible from the unconsumed arguments.
template <class Derived, class A>
class AnalysisTask {
   public:
    // Derived provides:  explicit Derived(AnalysisContext<A>&);   [fn new]
    //                    void process_frame(AnalysisContext<A>&); void post_process(AnalysisContext<A>&);
    //                    static std::string task_name();
    static void run(const std::vector<std::string> &argv, FrameSource &src) {          // :124-280
        const TrajAnalysisArgs traj_args = TrajAnalysisArgs::parse(argv);
        std::unique_ptr<Derived> inst;
        std::unique_ptr<AnalysisContext<A>> context;
        size_t index = 0;
        for_each_consumed(traj_args, src, [&](State &&state, bool from_structure_file) {
            if (!context) {                                                            // :282-306, :253-262
                Topology top = src.read_topology(traj_args.files[0]);
                context.reset(new AnalysisContext<A>{System(std::move(top), std::move(state)), 0, A(traj_args.rest)});
                construct(inst, *context);
            } else {                                                                   // :245-252
                context->sys.set_state(std::move(state));
            }
            context->frame_index = index++;
            process(*inst, *context);
            if (!from_structure_file) context->consumed_frames += 1;                   // :179 counts it in the loop variable only
        });
        finish(inst.get(), context.get());
    }

    // Frame-parallel form of run() for one node with several GPUs: the frames run() would consume (same -b/-e/--skip
    // window logic, same order of reads - the trajectory is still read by ONE thread, the caller's) are dealt in
    // contiguous blocks of `block` frames, round-robin, to one worker thread per entry of `devices`
    // (molar_hip_device_count() of them for a whole node; the same device may be named twice).  Every worker owns an
    // engine context on its device (ctx.eng()), its own System and its own task instance - constructed, like the
    // reference's T::new, on the FIRST consumed frame of the run, which only the worker that owns frame 0 also
    // processes.  There is no exchange while frames are processed (the analysis_task.rs:202-267 loop body runs unchanged
    // per frame); after the last frame the instances are folded into the first one in worker order with
    //     void Derived::merge(Derived &&other);
    // (integer accumulators add; per-frame series carry ctx.frame_index and are put in frame order there) and
    // post_process runs once, on the merged instance, with consumed_frames = the whole run's.
    //
    // One reader PER WORKER where the source allows it (FrameSource::frame_count + concurrent_open, frame-based or absent -b / -e):
    // the frames the run consumes are known up front, every worker takes ONE contiguous block of them and reads it through its
    // own reader, opened at the block's first frame (the index makes the seek free) - no producer thread, no queue.  A single
    // reader decodes ~460 frames/s of 250k atoms per host thread; eight GPUs binning 4 k frames/s each would wait for it.
    // Sources that cannot seek, and time-based windows, keep the single-reader form below (blocks of `block` frames dealt round-robin).
    // devices[w] < 0: the worker gets no engine context (host-only tasks; ctx.eng() then falls back to the process-wide one).
    static void run_sharded(const std::vector<std::string> &argv, FrameSource &src, const std::vector<int> &devices,
                            size_t block = 8) {
        if (devices.empty()) throw AnalysisError(AnalysisError::Arg, "run_sharded: no devices");
        if (block == 0) block = 1;
        const TrajAnalysisArgs traj_args = TrajAnalysisArgs::parse(argv);
        if (run_sharded_own_readers(traj_args, src, devices)) return;
        struct Item { size_t index; State state; };
        struct Worker {
            std::mutex m;
            std::condition_variable cv;
            std::deque<Item> q;
            bool closed = false;
            std::unique_ptr<Engine> engine;
            std::unique_ptr<AnalysisContext<A>> context;
            std::unique_ptr<Derived> inst;
            std::exception_ptr error;
            std::thread th;
        };
        const size_t W = devices.size(), QCAP = 2 * block + 2;
        std::vector<std::unique_ptr<Worker>> workers;
        for (size_t w = 0; w < W; ++w) workers.emplace_back(new Worker);
        std::atomic<bool> abort{false};
        Topology top;                    // read with the first consumed frame, as run() does
        State first;                     // what every instance is constructed on
        auto body = [&](size_t w) {
            Worker &me = *workers[w];
            try {
                if (devices[w] >= 0) me.engine.reset(new Engine(devices[w]));
                for (;;) {
                    Item it;
                    {
                        std::unique_lock<std::mutex> lk(me.m);
                        me.cv.wait(lk, [&] { return !me.q.empty() || me.closed; });
                        if (me.q.empty()) break;
                        it = std::move(me.q.front());
                        me.q.pop_front();
                    }
                    me.cv.notify_all();
                    if (abort.load()) continue;
                    if (!me.context) {
                        me.context.reset(new AnalysisContext<A>{System(top, first), 0, A(traj_args.rest)});
                        me.context->engine = me.engine.get();
                        construct(me.inst, *me.context);
                        if (it.index != 0) me.context->sys.set_state(std::move(it.state));
                    } else {
                        me.context->sys.set_state(std::move(it.state));
                    }
                    me.context->frame_index = it.index;
                    process(*me.inst, *me.context);
                    me.context->consumed_frames += 1;
                }
            } catch (...) {
                me.error = std::current_exception();
                abort.store(true);
                std::lock_guard<std::mutex> lk(me.m);      // unblock a producer waiting for room in this queue
                me.q.clear();
                me.cv.notify_all();
            }
        };
        for (size_t w = 0; w < W; ++w) workers[w]->th = std::thread(body, w);
        size_t index = 0, struct_frames = 0;
        std::exception_ptr producer_error;
        try {
            for_each_consumed(traj_args, src, [&](State &&state, bool from_structure_file) {
                if (abort.load()) return;
                if (index == 0) {
                    top = src.read_topology(traj_args.files[0]);
                    first = state;
                }
                if (from_structure_file) struct_frames += 1;
                Worker &to = *workers[(index / block) % W];
                {
                    std::unique_lock<std::mutex> lk(to.m);
                    to.cv.wait(lk, [&] { return to.q.size() < QCAP || abort.load(); });
                    if (!abort.load()) to.q.push_back(Item{index, std::move(state)});
                }
                to.cv.notify_all();
                index += 1;
            });
        } catch (...) {
            producer_error = std::current_exception();
            abort.store(true);
        }
        for (auto &w : workers) {
            { std::lock_guard<std::mutex> lk(w->m); w->closed = true; }
            w->cv.notify_all();
        }
        for (auto &w : workers) w->th.join();
        if (producer_error) std::rethrow_exception(producer_error);
        for (auto &w : workers)
            if (w->error) std::rethrow_exception(w->error);
        Worker *head = nullptr;
        size_t consumed = 0;
        for (auto &w : workers) {
            if (!w->inst) continue;
            consumed += w->context->consumed_frames;
            if (!head) head = w.get();
            else head->inst->merge(std::move(*w->inst));
        }
        if (head) head->context->consumed_frames = consumed - struct_frames;
        finish(head ? head->inst.get() : nullptr, head ? head->context.get() : nullptr);
    }

    // frames/s the readers of the last run_sharded delivered, summed over workers (each worker: its frames over the time it spent
    // inside its reader) - what a caller prints to see whether the frame supply or the analysis binds
    static double &last_reader_fps() { static double v = 0; return v; }

   private:
    // The per-worker-reader form of run_sharded; false when the run does not qualify (nothing has been read then).
    static bool run_sharded_own_readers(const TrajAnalysisArgs &traj_args, FrameSource &src, const std::vector<int> &devices) {
        if (!src.concurrent_open()) return false;
        if (!traj_args.use_struct_file && traj_args.files.size() < 2) return false;      // (the single-reader form reports NoTraj)
        const auto [begin_frame, begin_time] = process_suffix(traj_args.begin);
        const auto [end_frame, end_time] = process_suffix(traj_args.end);
        if (begin_time || end_time) return false;
        // the frames run() would hand to the task, as (file, frame in the file): global frame g of the files in order is consumed
        // when begin <= g < end and (g - begin) % skip == 0 (analysis_task.rs:205-234; with one file the begin is a seek, :189-198)
        struct Ref { size_t file, frame; };
        std::vector<Ref> refs;
        size_t g = 0;
        const size_t b = begin_frame.value_or(0);
        for (size_t fidx = 1; fidx < traj_args.files.size(); ++fidx) {
            const auto nf = src.frame_count(traj_args.files[fidx]);
            if (!nf) return false;
            for (size_t k = 0; k < *nf; ++k, ++g) {
                if (g < b) continue;
                if (end_frame && g >= *end_frame) break;
                if ((g - b) % traj_args.skip == 0) refs.push_back(Ref{fidx, k});
            }
        }
        const size_t lead = traj_args.use_struct_file ? 1 : 0;       // the structure file's state is consumed first (:168-179)
        const size_t total = refs.size() + lead;
        if (total == 0) throw AnalysisError(AnalysisError::NoFramesConsumed, "no frames consumed");
        // the first consumed frame: every instance is constructed on it (T::new sees the first frame, :282-306)
        Topology top = src.read_topology(traj_args.files[0]);
        State first;
        if (lead) first = src.read_structure_state(traj_args.files[0]);
        else {
            auto next = src.open(traj_args.files[refs[0].file], refs[0].frame, std::nullopt);
            auto st = next();
            if (!st) throw AnalysisError(AnalysisError::NoFramesConsumed, "no frames consumed");
            first = std::move(*st);
        }
        struct Worker {
            std::unique_ptr<Engine> engine;
            std::unique_ptr<AnalysisContext<A>> context;
            std::unique_ptr<Derived> inst;
            std::exception_ptr error;
            std::thread th;
            double read_s = 0;
            size_t frames = 0;
        };
        const size_t W = devices.size(), per = (total + W - 1) / W;
        std::vector<std::unique_ptr<Worker>> workers;
        for (size_t w = 0; w < W; ++w) workers.emplace_back(new Worker);
        std::atomic<bool> abort{false};
        auto body = [&](size_t w) {
            Worker &me = *workers[w];
            const size_t lo = std::min(total, w * per), hi = std::min(total, lo + per);      // consumed indices [lo, hi)
            if (lo == hi) return;
            try {
                if (devices[w] >= 0) me.engine.reset(new Engine(devices[w]));
                me.context.reset(new AnalysisContext<A>{System(top, first), 0, A(traj_args.rest)});
                me.context->engine = me.engine.get();
                construct(me.inst, *me.context);
                std::function<std::optional<State>()> next;
                size_t open_file = 0, next_frame = 0;          // where `next` stands
                for (size_t i = lo; i < hi && !abort.load(); ++i) {
                    if (i >= lead) {
                        const Ref r = refs[i - lead];
                        if (i != 0) {                          // (index 0 is `first`, already in the context)
                            const auto t0 = std::chrono::steady_clock::now();
                            if (!next || open_file != r.file || next_frame != r.frame) {
                                next = src.open(traj_args.files[r.file], r.frame, std::nullopt);
                                open_file = r.file;
                                next_frame = r.frame;
                            }
                            auto st = next();
                            me.read_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                            me.frames += 1;
                            if (!st) throw MolarError(MOLAR_HIP_ERR_IO, "run_sharded: the trajectory ended before frame " + std::to_string(r.frame));
                            next_frame += 1;
                            me.context->sys.set_state(std::move(*st));
                        }
                    }
                    me.context->frame_index = i;
                    process(*me.inst, *me.context);
                    me.context->consumed_frames += 1;
                }
            } catch (...) {
                me.error = std::current_exception();
                abort.store(true);
            }
        };
        for (size_t w = 0; w < W; ++w) workers[w]->th = std::thread(body, w);
        for (auto &w : workers) w->th.join();
        for (auto &w : workers)
            if (w->error) std::rethrow_exception(w->error);
        Worker *head = nullptr;
        size_t consumed = 0;
        double fps = 0;
        for (auto &w : workers) {
            if (w->read_s > 0) fps += (double)w->frames / w->read_s;
            if (!w->inst || w->context->consumed_frames == 0) continue;
            consumed += w->context->consumed_frames;
            if (!head) head = w.get();
            else head->inst->merge(std::move(*w->inst));
        }
        last_reader_fps() = fps;
        if (head) head->context->consumed_frames = consumed - lead;
        finish(head ? head->inst.get() : nullptr, head ? head->context.get() : nullptr);
        return true;
    }
    static void construct(std::unique_ptr<Derived> &inst, AnalysisContext<A> &context) {
        try { inst.reset(new Derived(context)); }
        catch (const AnalysisError &) { throw; }
        catch (const std::exception &e) { throw AnalysisError(AnalysisError::PreProcess, std::string("in task pre_process: ") + e.what()); }
    }
    static void process(Derived &inst, AnalysisContext<A> &context) {
        try { inst.process_frame(context); }
        catch (const std::exception &e) { throw AnalysisError(AnalysisError::ProcessFrame, std::string("in task process_frame: ") + e.what()); }
    }
    static void finish(Derived *inst, AnalysisContext<A> *context) {                    // :270-277
        if (!inst) throw AnalysisError(AnalysisError::NoFramesConsumed, "no frames consumed");
        try { inst->post_process(*context); }
        catch (const std::exception &e) { throw AnalysisError(AnalysisError::PostProcess, std::string("in task post_process: ") + e.what()); }
    }

    // The frame loop of run() (analysis_task.rs:124-267) without the task: calls f(state, from_structure_file) for every
    // frame the reference would hand to the task, in order.
    template <class F>
    static void for_each_consumed(const TrajAnalysisArgs &traj_args, FrameSource &src, F &&f) {
        if (!traj_args.use_struct_file && traj_args.files.size() < 2)
            throw AnalysisError(AnalysisError::NoTraj, "at least one trajectory required if 'use_struct_file' is not set");
        const auto [begin_frame, begin_time] = process_suffix(traj_args.begin);
        const auto [end_frame, end_time] = process_suffix(traj_args.end);
        size_t global_frame = 0, phase = 0;
        const bool random_access_begin = traj_args.files.size() - 1 == 1;
        if (traj_args.use_struct_file) f(src.read_structure_state(traj_args.files[0]), true);   // :168-179
        bool stop = false;
        for (size_t fidx = 1; fidx < traj_args.files.size() && !stop; ++fidx) {         // :184
            std::optional<size_t> sk_fr;
            std::optional<Float> sk_t;
            if (random_access_begin) {                                                  // :189-198
                if (begin_frame) {
                    if (*begin_frame > 0) { sk_fr = begin_frame; global_frame = *begin_frame; }
                } else if (begin_time) {
                    sk_t = begin_time;
                }
            }
            auto next = src.open(traj_args.files[fidx], sk_fr, sk_t);
            while (auto st = next()) {                                                 // :202
                State &state = *st;
                if (!random_access_begin) {                                            // :205-215
                    bool before_begin = false;
                    if (begin_frame) before_begin = global_frame < *begin_frame;
                    else if (begin_time) before_begin = state.get_time() < *begin_time;
                    if (before_begin) { global_frame += 1; continue; }
                }
                if ((end_frame && global_frame >= *end_frame) || (end_time && state.get_time() > *end_time)) {   // :219-223
                    stop = true;
                    break;
                }
                const bool keep = phase % traj_args.skip == 0;                          // :229-234
                phase += 1;
                global_frame += 1;
                if (!keep) continue;
                f(std::move(state), false);                                            // :245-262
            }
        }
    }
};

}  // namespace molar
