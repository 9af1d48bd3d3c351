// GPU parity of the C++ host mirror (include/molar_hip.hpp -> libmolar_hip.so) against the CPU oracle
// (oracle/molar_oracle.h, f32 build).  The tests read like the reference's own usage: bound
// selections, distance_search_*<T>, fit_transform / apply_transform / rmsd, an AnalysisTask.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <tuple>
#include <vector>

#include "molar_hip.hpp"

// the one HIP runtime call this test makes itself (reading a device-resident result back): hipMemcpyDeviceToHost = 2
extern "C" int hipMemcpy(void *dst, const void *src, size_t bytes, int kind);
extern "C" {
#include "molar_oracle.h"
}

using namespace molar;

static int failures = 0;
#define EXPECT(cond)                                                         \
    do {                                                                     \
        if (!(cond)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #cond); ++failures; } \
    } while (0)

static uint64_t rng_state = 20240607ull;
static double urand() {                       // SplitMix64
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (double)(z >> 11) / 9007199254740992.0;
}

static State make_state(size_t n, const Matrix3f &box, double jitter, float time = 0) {
    State s;
    s.coords.resize(n);
    for (size_t k = 0; k < n; ++k) {
        const double f[3] = {urand(), urand(), urand()};
        for (int r = 0; r < 3; ++r)
            s.coords[k][r] = (float)(box(r, 0) * f[0] + box(r, 1) * f[1] + box(r, 2) * f[2] + jitter * (urand() - 0.5));
    }
    s.pbox = PeriodicBox::from_matrix(box);
    s.time = time;
    return s;
}

static bool close_rel(double a, double b, double rel) { return std::fabs(a - b) <= rel * std::fmax(std::fabs(a), std::fabs(b)) + 1e-7; }

static void search_tests() {
    const size_t n = 6000;
    Matrix3f box;                              // triclinic, columns a,b,c (negative shear: reference grid complete)
    box(0, 0) = 3.9f; box(1, 1) = 3.9f; box(2, 2) = 3.9f; box(0, 2) = -0.5f; box(1, 2) = -0.5f;
    System sys(Topology{std::vector<Float>(n, 12.011f), std::vector<Float>(n, 0.17f)}, make_state(n, box, 0.3));
    SelBound all = SelBound::all(sys);
    std::vector<usize> ev, od;
    for (usize k = 0; k < n; ++k) (k % 2 ? od : ev).push_back(k);
    SelBound s_ev(sys, ev), s_od(sys, od);
    const PeriodicBox &pb = all.require_box();
    orc_box ob;
    orc_box_from_matrix(box.m.data(), &ob);

    // distance_search_single_pbc -> Vec<(usize,usize,Float)>: identical elements in identical order
    {
        auto got = distance_search_single_pbc<std::tuple<usize, usize, Float>>(0.6f, all, pb, PBC_FULL);
        std::vector<uint64_t> ids(n);
        for (size_t k = 0; k < n; ++k) ids[k] = k;
        orc_pairs *ref = orc_search_single_pbc(0.6f, all.coords_ptr(), ids.data(), n, &ob, 7, 4);
        EXPECT(got.size() == ref->n && ref->n > 1000);
        bool same = got.size() == ref->n;
        for (size_t k = 0; same && k < ref->n; ++k)
            same = std::get<0>(got[k]) == ref->i[k] && std::get<1>(got[k]) == ref->j[k] && std::get<2>(got[k]) == ref->d[k];
        EXPECT(same);
        // Histogram1D fed by the fused search = add_one over the reference's distance stream (stats.rs:29-35)
        {
            Histogram1D h(0.05f, 0.6f, 220), hh(0.05f, 0.6f, 220);
            const uint64_t nd = h.add_distances_single_pbc(0.6f, all, pb, PBC_FULL);
            std::vector<float> want(220, 0.0f);
            orc_histogram_add(0.05f, 0.6f, 220, ref->d, ref->n, want.data());
            bool hs = nd == ref->n;
            for (size_t b = 0; hs && b < 220; ++b) hs = (float)h.counts()[b] == want[b];
            EXPECT(hs);
            for (size_t k = 0; k < ref->n; ++k) hh.add_one(ref->d[k]);
            EXPECT(hh.counts() == h.counts());
            h.add_distances_single_pbc(0.6f, all, pb, PBC_FULL);              // a second frame adds into the same bins
            bool twice = true;
            for (size_t b = 0; b < 220; ++b) twice = twice && h.counts()[b] == 2 * hh.counts()[b];
            EXPECT(twice);
            const auto dens = h.normalized_density();
            double integral = 0;
            for (size_t b = 0; b < 220; ++b) integral += dens[b] * (0.55 / 220);
            EXPECT(std::fabs(integral - 1.0) < 1e-4);
        }
        orc_pairs_free(ref);
        // (usize,usize) projection and local ids (modify.rs:78)
        auto pr = distance_search_single_pbc<std::pair<usize, usize>>(0.6f, s_ev, pb, PBC_FULL, true);
        std::vector<float> pos_ev(3 * ev.size());
        for (size_t k = 0; k < ev.size(); ++k) std::memcpy(&pos_ev[3 * k], &sys.state.coords[ev[k]].x, 12);
        orc_pairs *rl = orc_search_single_pbc(0.6f, pos_ev.data(), nullptr, ev.size(), &ob, 7, 4);
        bool ok = pr.size() == rl->n;
        for (size_t k = 0; ok && k < rl->n; ++k) ok = pr[k].first == rl->i[k] && pr[k].second == rl->j[k];
        EXPECT(ok);
        orc_pairs_free(rl);
    }
    // distance_search_double (non-periodic) with global ids
    {
        auto got = distance_search_double<std::tuple<usize, usize, Float>>(0.5f, s_ev, s_od);
        std::vector<float> p1(3 * ev.size()), p2(3 * od.size());
        for (size_t k = 0; k < ev.size(); ++k) std::memcpy(&p1[3 * k], &sys.state.coords[ev[k]].x, 12);
        for (size_t k = 0; k < od.size(); ++k) std::memcpy(&p2[3 * k], &sys.state.coords[od[k]].x, 12);
        orc_pairs *ref = orc_search_double(0.5f, p1.data(), ev.data(), ev.size(), p2.data(), od.data(), od.size(), 4);
        bool same = got.size() == ref->n && ref->n > 100;
        for (size_t k = 0; same && k < ref->n; ++k)
            same = std::get<0>(got[k]) == ref->i[k] && std::get<1>(got[k]) == ref->j[k] && std::get<2>(got[k]) == ref->d[k];
        EXPECT(same);
        orc_pairs_free(ref);
        // within_pbc: usize stream with the reference's duplicates
        auto w = distance_search_within_pbc(0.5f, s_ev, s_od, pb, PBC_FULL);
        orc_pairs *rw = orc_search_within_pbc(0.5f, p1.data(), ev.data(), ev.size(), p2.data(), od.data(), od.size(), &ob, 7, 4);
        bool okw = w.size() == rw->n;
        for (size_t k = 0; okw && k < rw->n; ++k) okw = w[k] == rw->i[k];
        EXPECT(okw);
        // ... and the set the selection keeps of it (sorted, de-duplicated: selection_expr.rs:112), computed without the stream
        {
            std::vector<usize> want(rw->i, rw->i + rw->n);
            std::sort(want.begin(), want.end());
            want.erase(std::unique(want.begin(), want.end()), want.end());
            auto ws = within_set(0.5f, s_ev, s_od, &pb, PBC_FULL);
            EXPECT(ws == want && !ws.empty());
        }
        orc_pairs_free(rw);
        // vdw: local ids (:791-792)
        std::vector<Float> v1(ev.size(), 0.17f), v2(od.size(), 0.15f);
        auto gv = distance_search_double_vdw_pbc<std::pair<usize, usize>>(s_ev, s_od, v1, v2, pb, PBC_FULL);
        orc_pairs *rv = orc_search_double_vdw_pbc(p1.data(), ev.size(), p2.data(), od.size(), v1.data(), v2.data(), &ob, 7, 4);
        bool okv = gv.size() == rv->n && rv->n > 10;
        for (size_t k = 0; okv && k < rv->n; ++k) okv = gv[k].first == rv->i[k] && gv[k].second == rv->j[k];
        EXPECT(okv);
        orc_pairs_free(rv);
    }
}

static void measure_tests() {
    const size_t n = 20000;
    Matrix3f box;
    box(0, 0) = 5.8f; box(1, 1) = 5.8f; box(2, 2) = 5.8f; box(0, 2) = -0.8f; box(1, 2) = -0.8f;
    Topology top;
    top.masses.resize(n);
    const float cyc[4] = {1.008f, 12.011f, 14.007f, 15.999f};
    for (size_t k = 0; k < n; ++k) top.masses[k] = cyc[k % 4];
    System ref_sys(top, make_state(n, box, 0.0));
    System cur_sys(top, ref_sys.state);
    // rotate + translate + perturb the current frame
    const double a = 0.8, ca = std::cos(a), sa = std::sin(a);
    for (auto &p : cur_sys.state.coords) {
        const double x = p.x, y = p.y;
        p.x = (float)(ca * x - sa * y + 1.5 + 0.02 * (urand() - 0.5));
        p.y = (float)(sa * x + ca * y - 2.0 + 0.02 * (urand() - 0.5));
        p.z = (float)(p.z + 0.7 + 0.02 * (urand() - 0.5));
    }
    std::vector<usize> idx;
    for (usize k = 0; k < n; k += 10) idx.push_back(k);
    SelBound cur(cur_sys, idx), ref(ref_sys, idx);
    const float *cx = cur.coords_ptr(), *rx = ref.coords_ptr();
    float want3[3], wantf;

    orc_center_of_mass(cx, idx.data(), idx.size(), top.masses.data(), want3);
    const Pos com = cur.center_of_mass();
    EXPECT(close_rel(com.x, want3[0], 3e-4) && close_rel(com.y, want3[1], 3e-4) && close_rel(com.z, want3[2], 3e-4));
    orc_center_of_geometry(cx, idx.data(), idx.size(), want3);
    const Pos cog = cur.center_of_geometry();
    EXPECT(close_rel(cog.x, want3[0], 3e-4) && close_rel(cog.z, want3[2], 3e-4));
    orc_gyration(cx, idx.data(), idx.size(), top.masses.data(), &wantf);
    EXPECT(close_rel(cur.gyration(), wantf, 3e-4));
    orc_rmsd(cx, idx.data(), idx.size(), rx, idx.data(), idx.size(), &wantf);
    EXPECT(close_rel(rmsd(cur, ref), wantf, 3e-4));
    orc_rmsd_mw(cx, idx.data(), idx.size(), top.masses.data(), rx, idx.data(), idx.size(), &wantf);
    EXPECT(close_rel(rmsd_mw(cur, ref), wantf, 3e-4));
    float lo[3], hi[3];
    orc_min_max(cx, idx.data(), idx.size(), lo, hi);
    const auto mm = cur.min_max();
    EXPECT(mm.first.x == lo[0] && mm.first.y == lo[1] && mm.second.z == hi[2]);

    // the align + RMSD loop body of benches/comparison_small.rs:17-24
    float R[9], t[3];
    orc_fit_transform(cx, idx.data(), idx.size(), top.masses.data(), rx, idx.data(), idx.size(), top.masses.data(), R, t);
    const IsometryMatrix3 tr = fit_transform(cur, ref);
    bool okR = true;
    for (int k = 0; k < 9; ++k) okR = okR && std::fabs(tr.R.m[k] - R[k]) < 1e-4;
    EXPECT(okR && std::fabs(tr.t.x - t[0]) < 2e-3 && std::fabs(tr.t.y - t[1]) < 2e-3 && std::fabs(tr.t.z - t[2]) < 2e-3);
    std::vector<float> moved(cx, cx + 3 * n);
    orc_apply_transform(moved.data(), idx.data(), idx.size(), tr.R.m.data(), &tr.t.x);
    cur.apply_transform(tr);
    EXPECT(std::memcmp(moved.data(), cur.coords_ptr(), 12 * n) == 0);           // bit-identical coordinates
    EXPECT(rmsd(cur, ref) < 0.05f);

    // resident result = the host result (one round trip; pairs stay in HBM)
    {
        const auto host = distance_search_single_pbc<std::tuple<usize, usize, Float>>(0.6f, cur, cur.require_box(), PBC_FULL);
        const ResidentPairs rp = distance_search_single_pbc_resident(0.6f, cur, cur.require_box(), PBC_FULL);
        EXPECT(rp.count == host.size() && rp.pairs != nullptr && rp.dist != nullptr);
        std::vector<uint32_t> pr(2 * rp.count);
        std::vector<float> dd(rp.count);
        EXPECT(molar_hip_search_fill(cur.ctx(), pr.data(), dd.data()) == MOLAR_HIP_OK);     // cached search, host copy
        bool same = true;
        for (size_t k = 0; k < host.size(); ++k)
            same = same && pr[2 * k] == std::get<0>(host[k]) && pr[2 * k + 1] == std::get<1>(host[k]) && dd[k] == std::get<2>(host[k]);
        EXPECT(same);
    }

    // the same search through the two-in-flight pipeline: three pushes of the same selection, every result = the host result
    {
        const auto host = distance_search_single_pbc<std::tuple<usize, usize, Float>>(0.6f, cur, cur.require_box(), PBC_FULL);
        PairPipeline pipe(cur.ctx());
        size_t seen = 0;
        auto verify = [&](const ResidentPairs &rp) {
            ++seen;
            EXPECT(rp.count == host.size());
            std::vector<uint32_t> pr(2 * rp.count);
            std::vector<float> dd(rp.count);
            EXPECT(hipMemcpy(pr.data(), rp.pairs, pr.size() * 4, 2) == 0);
            EXPECT(hipMemcpy(dd.data(), rp.dist, dd.size() * 4, 2) == 0);
            bool same = rp.count == host.size();
            for (size_t k = 0; same && k < host.size(); ++k)
                same = pr[2 * k] == std::get<0>(host[k]) && pr[2 * k + 1] == std::get<1>(host[k]) && dd[k] == std::get<2>(host[k]);
            EXPECT(same);
        };
        for (int f = 0; f < 3; ++f)
            if (auto r = pipe.push(0.6f, cur, cur.require_box(), PBC_FULL)) verify(*r);
        if (auto r = pipe.finish()) verify(*r);
        EXPECT(seen == 3 && !pipe.finish());
    }

    // translate / rotate / principal_transform (modify.rs:16-30, measure.rs:100-109)
    {
        const Pos c0 = cur.center_of_mass();
        cur.translate({0.5f, -1.0f, 0.25f});
        const Pos c1 = cur.center_of_mass();
        EXPECT(std::fabs(c1.x - c0.x - 0.5f) < 1e-4 && std::fabs(c1.y - c0.y + 1.0f) < 1e-4 && std::fabs(c1.z - c0.z - 0.25f) < 1e-4);
        const Float g0 = cur.gyration();
        cur.rotate({0.0f, 0.0f, 1.0f}, 1.5707963f);             // (x, y) -> (-y, x)
        const Pos c2 = cur.center_of_mass();
        EXPECT(std::fabs(c2.x + c1.y) < 2e-4 && std::fabs(c2.y - c1.x) < 2e-4 && std::fabs(c2.z - c1.z) < 1e-4);
        EXPECT(std::fabs(cur.gyration() - g0) < 1e-4);
        const IsometryMatrix3 pt = cur.principal_transform();
        cur.apply_transform(pt);
        const auto in = cur.inertia();                          // axes of the aligned cloud = identity up to sign
        EXPECT(in.first.x <= in.first.y && in.first.y <= in.first.z);
        EXPECT(std::fabs(std::fabs(in.second(0, 0)) - 1.0f) < 1e-3 && std::fabs(std::fabs(in.second(1, 1)) - 1.0f) < 1e-3);
        const Pos c3 = cur.center_of_mass();
        EXPECT(std::fabs(c3.x - c2.x) < 2e-4 && std::fabs(c3.y - c2.y) < 2e-4 && std::fabs(c3.z - c2.z) < 2e-4);
    }

    // error mapping (measure.rs:732-762)
    bool threw = false;
    std::vector<usize> shorter(idx.begin(), idx.end() - 1);
    SelBound sh(ref_sys, shorter);
    try { rmsd(cur, sh); } catch (const MeasureError &e) { threw = e.code == MeasureError::Sizes; }
    EXPECT(threw);
    threw = false;
    System nobox(top, State{ref_sys.state.coords, std::nullopt, 0});
    try { SelBound::all(nobox).center_of_mass_pbc(); } catch (const PeriodicBoxError &e) { threw = e.code == PeriodicBoxError::NoPbc; }
    EXPECT(threw);
}

// ---- an AnalysisTask written against the mirror: per-frame fit + RMSD to the first frame
struct Frames : FrameSource {
    size_t n = 5000;
    Matrix3f box;
    Topology top;
    std::vector<State> traj;
    Frames() {
        box(0, 0) = 3.7f; box(1, 1) = 3.7f; box(2, 2) = 3.7f;
        top.masses.assign(n, 12.011f);
        State base = make_state(n, box, 0.0);
        for (int f = 0; f < 6; ++f) {
            State s = base;
            s.time = 10.0f * f;
            for (auto &p : s.coords) { p.x += (float)(0.05 * f + 0.03 * (urand() - 0.5)); p.y += (float)(0.03 * (urand() - 0.5)); }
            traj.push_back(s);
        }
    }
    Topology read_topology(const std::string &) override { return top; }
    State read_structure_state(const std::string &) override { return traj[0]; }
    std::function<std::optional<State>()> open(const std::string &, std::optional<size_t> skip, std::optional<Float>) override {
        auto pos = std::make_shared<size_t>(skip ? *skip : 0);
        return [this, pos]() -> std::optional<State> {
            if (*pos >= traj.size()) return std::nullopt;
            return traj[(*pos)++];
        };
    }
};
struct NoArgs { explicit NoArgs(const std::vector<std::string> &) {} };
static std::vector<float> task_rmsd;
struct AlignTask : AnalysisTask<AlignTask, NoArgs> {
    std::unique_ptr<System> ref;
    explicit AlignTask(AnalysisContext<NoArgs> &ctx) : ref(new System(ctx.sys.top, ctx.sys.state)) {}
    void process_frame(AnalysisContext<NoArgs> &ctx) {
        SelBound cur = SelBound::all(ctx.sys), r = SelBound::all(*ref);
        cur.apply_transform(fit_transform(cur, r));
        task_rmsd.push_back(rmsd(cur, r));
    }
    void post_process(AnalysisContext<NoArgs> &) {}
    static std::string task_name() { return "align"; }
};

static void task_tests() {
    Frames src;
    AlignTask::run({"-f", "top", "traj", "-b", "1", "--skip", "2"}, src);       // frames 1, 3, 5
    EXPECT(task_rmsd.size() == 3);
    const std::vector<size_t> which{1, 3, 5};
    std::vector<float> m(src.top.masses);
    for (size_t q = 0; q < which.size() && q < task_rmsd.size(); ++q) {
        // reference = first processed frame (frame 1), as T::new sees it
        const State &cur = src.traj[which[q]], &ref = src.traj[1];
        float R[9], t[3], want;
        orc_fit_transform(&cur.coords[0].x, nullptr, src.n, m.data(), &ref.coords[0].x, nullptr, src.n, m.data(), R, t);
        std::vector<float> moved(&cur.coords[0].x, &cur.coords[0].x + 3 * src.n);
        orc_apply_transform(moved.data(), nullptr, src.n, R, t);
        orc_rmsd(moved.data(), nullptr, src.n, &ref.coords[0].x, nullptr, src.n, &want);
        EXPECT(std::fabs(task_rmsd[q] - want) < 2e-5f + 3e-4f * want);
    }
}

// FitStream (molar_hip_fit_stream_*): host-memory frames, three in flight - every record equals what the selection methods give
// for that frame (same kernels on the packed selection), and apply moves the frame like apply_transform does
static void fit_stream_tests() {
    Frames src;
    System ref_sys(src.top, src.traj[0]);
    std::vector<usize> idx;
    for (usize k = 0; k < src.n; k += 3) idx.push_back(k);
    SelBound ref(ref_sys, idx);
    FitStream fs(ref, ref, 2);
    std::vector<State> work(src.traj.begin(), src.traj.end());
    std::vector<int32_t> tickets;
    std::vector<FitRecord> got;
    for (size_t f = 0; f < work.size(); ++f) {
        tickets.push_back(fs.begin(work[f].coords, /*apply=*/true));
        if (tickets.size() == 3) { got.push_back(fs.end(tickets.front())); tickets.erase(tickets.begin()); }
    }
    for (int32_t t : tickets) got.push_back(fs.end(t));
    EXPECT(got.size() == work.size());
    for (size_t f = 0; f < work.size() && f < got.size(); ++f) {
        System cur_sys(src.top, src.traj[f]);
        SelBound cur(cur_sys, idx);
        const IsometryMatrix3 tr = fit_transform(cur, ref);
        for (int q = 0; q < 9; ++q) EXPECT(std::fabs(tr.R.m[q] - got[f].tr.R.m[q]) < 2e-6f);
        EXPECT(std::fabs(tr.t.x - got[f].tr.t.x) < 2e-5f && std::fabs(tr.t.y - got[f].tr.t.y) < 2e-5f && std::fabs(tr.t.z - got[f].tr.t.z) < 2e-5f);
        cur.apply_transform(got[f].tr);
        EXPECT(std::fabs(rmsd(cur, ref) - got[f].rmsd) < 1e-5f + 1e-5f * got[f].rmsd);
        for (usize k : idx) {     // the stream moved the same atoms to the same places; the others are untouched
            EXPECT(cur_sys.state.coords[k].x == work[f].coords[k].x && cur_sys.state.coords[k].y == work[f].coords[k].y &&
                   cur_sys.state.coords[k].z == work[f].coords[k].z);
        }
        EXPECT(work[f].coords[1].x == src.traj[f].coords[1].x);
    }
}

// ---- the frame-parallel form: two engine contexts (both on device 0 here; one per GPU on a node) over 8 frames must give
// the integer bins bit for bit and the per-frame series value for value of the serial run (analysis_task.rs:202-267 per
// frame; DESIGN.md section 7: contiguous frame blocks, integer reduction at the end, series in frame order)
struct RdfTask : AnalysisTask<RdfTask, NoArgs> {
    Histogram1D hist{0.0f, 0.9f, 300};
    std::vector<std::pair<size_t, float>> gyr;        // (frame index, gyration radius)
    uint64_t pairs = 0;
    size_t frames_seen_at_post = 0;
    static RdfTask *last;
    explicit RdfTask(AnalysisContext<NoArgs> &) {}
    void process_frame(AnalysisContext<NoArgs> &ctx) {
        SelBound all = SelBound::all(ctx.sys, ctx.eng());
        pairs += hist.add_distances_single_pbc(0.9f, all, *ctx.sys.state.pbox, PBC_FULL);
        gyr.emplace_back(ctx.frame_index, all.gyration());
    }
    void merge(RdfTask &&o) {
        hist.merge(o.hist);
        pairs += o.pairs;
        gyr.insert(gyr.end(), o.gyr.begin(), o.gyr.end());
        std::sort(gyr.begin(), gyr.end());
    }
    void post_process(AnalysisContext<NoArgs> &ctx) {
        frames_seen_at_post = ctx.consumed_frames;
        result_counts = hist.counts();
        result_gyr = gyr;
        result_pairs = pairs;
        result_frames = frames_seen_at_post;
    }
    static std::string task_name() { return "rdf"; }
    static std::vector<uint64_t> result_counts;
    static std::vector<std::pair<size_t, float>> result_gyr;
    static uint64_t result_pairs;
    static size_t result_frames;
};
std::vector<uint64_t> RdfTask::result_counts;
std::vector<std::pair<size_t, float>> RdfTask::result_gyr;
uint64_t RdfTask::result_pairs = 0;
size_t RdfTask::result_frames = 0;

struct ManyFrames : Frames {
    ManyFrames() {
        State base = traj[0];
        traj.clear();
        for (int f = 0; f < 11; ++f) {
            State s = base;
            s.time = 10.0f * f;
            s.pbox = PeriodicBox::from_matrix(box);
            for (auto &p : s.coords) { p.x += (float)(0.04 * (urand() - 0.5)); p.y += (float)(0.04 * (urand() - 0.5)); p.z += (float)(0.04 * (urand() - 0.5)); }
            traj.push_back(s);
        }
    }
};

static void sharded_task_tests() {
    ManyFrames src;
    const std::vector<std::string> argv{"-f", "top", "traj", "-b", "2", "-e", "10"};      // frames 2..9: 8 frames
    RdfTask::run(argv, src);
    const auto counts1 = RdfTask::result_counts;
    const auto gyr1 = RdfTask::result_gyr;
    const uint64_t pairs1 = RdfTask::result_pairs;
    EXPECT(RdfTask::result_frames == 8 && gyr1.size() == 8 && pairs1 > 0);
    for (size_t block : {size_t(1), size_t(3), size_t(8)}) {
        RdfTask::run_sharded(argv, src, {0, 0}, block);
        EXPECT(RdfTask::result_frames == 8);
        EXPECT(RdfTask::result_pairs == pairs1);
        EXPECT(RdfTask::result_counts == counts1);                       // integer bins: bit-identical
        EXPECT(RdfTask::result_gyr.size() == gyr1.size());
        for (size_t k = 0; k < gyr1.size() && k < RdfTask::result_gyr.size(); ++k) {
            EXPECT(RdfTask::result_gyr[k].first == k);                   // frame order restored by merge()
            EXPECT(RdfTask::result_gyr[k].second == gyr1[k].second);     // same kernels, same frames: same floats
        }
    }
    // three workers, one of which never gets a frame (8 frames in blocks of 8 -> worker 0 only)
    RdfTask::run_sharded(argv, src, {0, 0, 0}, 8);
    EXPECT(RdfTask::result_counts == counts1 && RdfTask::result_frames == 8);
    // an error inside a worker's process_frame surfaces in the caller
    struct Boom : AnalysisTask<Boom, NoArgs> {
        explicit Boom(AnalysisContext<NoArgs> &) {}
        void process_frame(AnalysisContext<NoArgs> &ctx) { if (ctx.frame_index == 5) throw std::runtime_error("frame 5"); }
        void merge(Boom &&) {}
        void post_process(AnalysisContext<NoArgs> &) {}
        static std::string task_name() { return "boom"; }
    };
    bool threw = false;
    try { Boom::run_sharded(argv, src, {0, 0}, 2); } catch (const AnalysisError &e) { threw = e.code == AnalysisError::ProcessFrame; }
    EXPECT(threw);
}

// The chained bilayer frame (MembraneFrames over molar_hip_membrane_frame_*) against the stage-by-stage C calls it
// replaces: the same kernels in the same order, so every array has to be bit-identical.
static void membrane_frames_tests() {
    Engine &eng = Engine::global();
    // a small bilayer: 2 x (12 x 12) lipids of 8 beads on a jittered lattice, heads out, tails towards the mid-plane
    const int side = 12, per = 8, K = 2 * side * side;
    const float L = side * 0.8f, Lz = 9.0f;
    const size_t natoms = (size_t)K * per;
    std::vector<float> xyz0(natoms * 3);
    uint32_t seed = 12345u;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return ((seed >> 8) & 0xFFFF) / 65536.0f - 0.5f; };
    for (int k = 0; k < K; ++k) {
        const int leaf = k / (side * side), a = k % (side * side);
        const float sgn = leaf == 0 ? 1.0f : -1.0f;
        const float cx = (a % side + 0.5f + 0.3f * rnd()) * 0.8f, cy = (a / side + 0.5f + 0.3f * rnd()) * 0.8f;
        for (int b = 0; b < per; ++b) {
            float *p = &xyz0[3 * ((size_t)k * per + b)];
            p[0] = cx + 0.05f * rnd(); p[1] = cy + 0.05f * rnd();
            p[2] = Lz / 2 + sgn * (2.0f - 0.25f * b) + 0.03f * rnd();
            for (int d = 0; d < 2; ++d) p[d] = p[d] - L * std::floor(p[d] / L);      // wrap: edge lipids get split
        }
    }
    const float box9[9] = {L, 0, 0, 0, L, 0, 0, 0, Lz};
    std::vector<uint64_t> lipid_idx(natoms), lipid_off(K + 1), marker_idx, marker_off{0}, tail_idx, tail_off{0};
    std::vector<uint32_t> tail_lipid;
    std::vector<float> masses(natoms);
    for (size_t i = 0; i < natoms; ++i) { lipid_idx[i] = i; masses[i] = 12.0f + (i % 3); }
    for (int k = 0; k <= K; ++k) lipid_off[k] = (uint64_t)k * per;
    for (int k = 0; k < K; ++k) {
        const uint64_t f = (uint64_t)k * per;
        for (uint64_t b : {0, 1}) marker_idx.push_back(f + b);
        marker_off.push_back(marker_idx.size());
        for (uint64_t b : {2, 3}) marker_idx.push_back(f + b);
        marker_off.push_back(marker_idx.size());
        for (uint64_t b : {6, 7}) marker_idx.push_back(f + b);
        marker_off.push_back(marker_idx.size());
        for (uint64_t b = 2; b < 8; ++b) tail_idx.push_back(f + b);
        tail_off.push_back(tail_idx.size());
        tail_lipid.push_back((uint32_t)k);
    }
    std::vector<uint8_t> bonds(tail_idx.size() - K, 1);
    molar_hip_membrane_desc D{};
    D.natoms = natoms; D.nlipids = K;
    D.lipid_idx = lipid_idx.data(); D.lipid_offsets = lipid_off.data(); D.marker_idx = marker_idx.data(); D.marker_offsets = marker_off.data();
    D.masses = masses.data(); D.ntails = K; D.tail_idx = tail_idx.data(); D.tail_offsets = tail_off.data(); D.tail_lipid = tail_lipid.data();
    D.tail_bonds = bonds.data(); D.cutoff = 1.6f; D.order_type = 1; D.max_smooth_iter = 1; D.unwrap = 1;
    MembraneFrames mem(eng, D);
    const PeriodicBox pbox = PeriodicBox::from_matrix(Matrix3f{{box9[0], box9[1], box9[2], box9[3], box9[4], box9[5], box9[6], box9[7], box9[8]}});
    std::vector<uint8_t> valid(K, 1);                     // the staged chain keeps its flags on the host
    const size_t norder = tail_idx.size() - 2 * K;
    std::vector<uint64_t> noff(K + 1);
    for (int k = 0; k <= K; ++k) noff[k] = k;
    for (int frame = 0; frame < 3; ++frame) {
        std::vector<float> a(xyz0), b(xyz0);
        for (size_t i = 0; i < a.size(); ++i) { const float j = 0.02f * rnd(); a[i] += j; b[i] += j; }
        // ---- chained
        auto none = mem.push(a.data(), pbox);
        EXPECT(!none.has_value());
        std::vector<float> head(K * 3), tail(K * 3), n0(K * 3), nrm(K * 3), area(K), order(norder), sh(K * 3);
        std::vector<uint64_t> poff(K + 1);
        std::vector<uint8_t> vout(K);
        std::vector<uint32_t> nvert(K);
        molar_hip_membrane_out O{};
        O.head = head.data(); O.tail = tail.data(); O.initial_normals = n0.data(); O.normals = nrm.data(); O.area = area.data();
        O.order = order.data(); O.smoothed_head = sh.data(); O.patch_offsets = poff.data(); O.valid = vout.data(); O.nvert = nvert.data();
        // (the middle frame takes its per-lipid arrays with the end of the frame - molar_hip_membrane_frame_end_fetch -, the others
        // fetch them afterwards: the same bytes either way)
        auto v = frame == 1 ? mem.finish(O) : mem.finish();
        EXPECT(v.has_value() && v->nlipids == (size_t)K);
        const size_t E = v->patch_entries, slots = E + 4 * (size_t)K;
        std::vector<float> fitted(E * 3), voro(slots * 3);
        std::vector<uint64_t> pids(E), neib(slots);
        if (frame == 1) {
            molar_hip_membrane_out Oe{};
            Oe.fitted_patch_points = fitted.data(); Oe.voro_vertexes = voro.data(); Oe.patch_ids = pids.data(); Oe.neib_ids = neib.data();
            mem.fetch(Oe);
            bool refused = false;                 // patch-sized arrays cannot come with the end of a frame
            try { MembraneFrames m2(eng, D); (void)m2.push(a.data(), pbox); (void)m2.finish(Oe); } catch (const MolarError &) { refused = true; }
            EXPECT(refused);
        } else {
            O.fitted_patch_points = fitted.data(); O.voro_vertexes = voro.data(); O.patch_ids = pids.data(); O.neib_ids = neib.data();
            mem.fetch(O);
        }
        // ---- stage by stage
        check(molar_hip_unwrap_simple_batch(eng.ctx(), b.data(), natoms, lipid_idx.data(), lipid_off.data(), K, box9, 7));
        EXPECT(std::memcmp(a.data(), b.data(), a.size() * 4) == 0);
        std::vector<float> mk(K * 9);
        check(molar_hip_center_batch(eng.ctx(), b.data(), natoms, marker_idx.data(), marker_off.data(), 3 * K, masses.data(), mk.data()));
        std::vector<float> h2(K * 3), t2(K * 3);
        for (int k = 0; k < K; ++k)
            for (int d = 0; d < 3; ++d) { h2[3 * k + d] = mk[9 * k + d]; t2[3 * k + d] = mk[9 * k + 6 + d]; }
        EXPECT(std::memcmp(h2.data(), head.data(), h2.size() * 4) == 0 && std::memcmp(t2.data(), tail.data(), t2.size() * 4) == 0);
        std::vector<uint64_t> vidx;
        for (int k = 0; k < K; ++k) if (valid[k]) vidx.push_back(k);
        molar_hip_search_desc q{};
        q.kind = MOLAR_HIP_SEARCH_SINGLE; q.cutoff = D.cutoff; q.xyz1 = h2.data(); q.natoms1 = K; q.idx1 = vidx.data(); q.n1 = vidx.size();
        q.ids_local = 0; q.box9 = box9; q.pbc = 7;
        uint64_t np = 0;
        check(molar_hip_search_count(eng.ctx(), &q, &np));
        std::vector<uint32_t> pairs(2 * np + 2);
        check(molar_hip_search_fill(eng.ctx(), pairs.data(), nullptr));
        EXPECT(2 * np == E && np == v->npairs);
        std::vector<uint64_t> poff2(K + 1), pids2(2 * np + 1);
        check(molar_hip_membrane_patches_from_pairs(pairs.data(), np, K, poff2.data(), pids2.data()));
        EXPECT(poff2 == poff && std::memcmp(pids2.data(), pids.data(), E * 8) == 0);
        std::vector<float> n02(K * 3, 0.f);
        check(molar_hip_membrane_initial_normals(K, h2.data(), t2.data(), poff2.data(), pids2.data(), valid.data(), n02.data()));
        EXPECT(std::memcmp(n02.data(), n0.data(), n0.size() * 4) == 0);
        std::vector<float> sh2(h2), nrm2(n02), area2(K, 0.f), fitted2(std::max<size_t>(E, 1) * 3, 0.f), voro2(slots * 3, 0.f);
        std::vector<uint64_t> neib2(slots, 0);
        std::vector<uint32_t> nvert2(K, 0);
        molar_hip_membrane_patches PP{(size_t)K, poff2.data(), pids2.data()};
        molar_hip_membrane_state S{};
        S.head_markers = sh2.data(); S.normals = nrm2.data(); S.valid = valid.data(); S.area = area2.data(); S.nvert = nvert2.data();
        S.neib_ids = neib2.data(); S.voro_vertexes = voro2.data(); S.fitted_patch_points = fitted2.data();
        check(molar_hip_membrane_smooth(eng.ctx(), &PP, box9, &S));
        EXPECT(std::memcmp(valid.data(), vout.data(), K) == 0);
        EXPECT(std::memcmp(sh2.data(), sh.data(), sh.size() * 4) == 0 && std::memcmp(nrm2.data(), nrm.data(), nrm.size() * 4) == 0);
        EXPECT(std::memcmp(area2.data(), area.data(), K * 4) == 0 && nvert2 == nvert && neib2 == neib);
        EXPECT(std::memcmp(voro2.data(), voro.data(), voro.size() * 4) == 0 && std::memcmp(fitted2.data(), fitted.data(), E * 12) == 0);
        std::vector<float> order2(norder);
        check(molar_hip_lipid_tail_order(eng.ctx(), b.data(), natoms, tail_idx.data(), tail_off.data(), K, 1, nrm2.data(), noff.data(), bonds.data(),
                                         order2.data()));
        EXPECT(std::memcmp(order2.data(), order.data(), norder * 4) == 0);
        size_t nvalid = 0;
        for (auto f : valid) nvalid += f;
        EXPECT(nvalid > (size_t)K / 2);
    }
}

int main() {
    try {
        search_tests();
        measure_tests();
        task_tests();
        fit_stream_tests();
        sharded_task_tests();
        membrane_frames_tests();
    } catch (const std::exception &e) {
        std::printf("exception: %s\n", e.what());
        return 2;
    }
    if (failures) { std::printf("%d failure(s)\n", failures); return 1; }
    std::printf("all host-mirror GPU tests passed\n");
    return 0;
}
