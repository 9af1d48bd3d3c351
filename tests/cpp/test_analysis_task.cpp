// CPU-only tests of the C++ host mirror: process_suffix (the reference's five tests,
// molar/src/analysis_task.rs:329-366, restated value for value) and the frame-window logic of
// AnalysisTask::run (analysis_task.rs:124-280) driven by a mock FrameSource.  No GPU calls.
#include <cstdio>
#include <string>
#include <vector>

#include "molar_hip.hpp"

using namespace molar;

static int failures = 0;
#define EXPECT(cond)                                                         \
    do {                                                                     \
        if (!(cond)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #cond); ++failures; } \
    } while (0)

template <class F>
static bool throws_kind(F f, int kind) {
    try { f(); } catch (const AnalysisError &e) { return e.code == kind; } catch (...) { return false; }
    return false;
}

static void suffix_tests() {
    using R = std::pair<std::optional<size_t>, std::optional<Float>>;
    // suffix_empty_is_no_limit
    EXPECT(process_suffix("") == R(std::nullopt, std::nullopt));
    EXPECT(process_suffix("   ") == R(std::nullopt, std::nullopt));
    // suffix_bare_number_is_frame
    EXPECT(process_suffix("0") == R(0, std::nullopt));
    EXPECT(process_suffix("5") == R(5, std::nullopt));
    EXPECT(process_suffix("42") == R(42, std::nullopt));
    EXPECT(process_suffix("100") == R(100, std::nullopt));
    // suffix_explicit_frame
    EXPECT(process_suffix("5fr") == R(5, std::nullopt));
    EXPECT(process_suffix("100fr") == R(100, std::nullopt));
    // suffix_time_units_convert_to_ps
    EXPECT(process_suffix("5ps") == R(std::nullopt, 5.0f));
    EXPECT(process_suffix("2ns") == R(std::nullopt, 2000.0f));
    EXPECT(process_suffix("1us") == R(std::nullopt, 1000000.0f));
    EXPECT(process_suffix("1.5ns") == R(std::nullopt, 1500.0f));
    // suffix_invalid
    EXPECT(throws_kind([] { process_suffix("5km"); }, AnalysisError::InvalidSuffix));
    bool threw = false;
    try { process_suffix("fr"); } catch (const AnalysisError &) { threw = true; }
    EXPECT(threw);
}

// ---- mock trajectory: file "tA" has frames with times 0,10,20,...; "tB" continues
struct MockSource : FrameSource {
    size_t natoms = 4;
    std::vector<std::pair<std::string, std::vector<Float>>> files{{"tA", {0, 10, 20, 30, 40}}, {"tB", {50, 60, 70}}};
    int topology_reads = 0;
    Topology read_topology(const std::string &) override {
        ++topology_reads;
        return Topology{std::vector<Float>(natoms, 1.0f), {}};
    }
    State read_structure_state(const std::string &) override {
        State s;
        s.coords.assign(natoms, Pos{});
        s.time = -1.0f;
        return s;
    }
    std::function<std::optional<State>()> open(const std::string &file, std::optional<size_t> skip_to_frame,
                                               std::optional<Float> skip_to_time) override {
        const std::vector<Float> *times = nullptr;
        for (auto &f : files)
            if (f.first == file) times = &f.second;
        auto pos = std::make_shared<size_t>(0);
        if (skip_to_frame) *pos = *skip_to_frame;
        if (skip_to_time)
            while (*pos < times->size() && (*times)[*pos] < *skip_to_time) ++*pos;
        const size_t n = natoms;
        return [times, pos, n]() -> std::optional<State> {
            if (!times || *pos >= times->size()) return std::nullopt;
            State s;
            s.coords.assign(n, Pos{});
            s.time = (*times)[(*pos)++];
            return s;
        };
    }
};

struct NoArgs {
    explicit NoArgs(const std::vector<std::string> &) {}
};

static std::vector<Float> seen;
static std::vector<size_t> seen_consumed;
static int post_calls = 0;

struct Recorder : AnalysisTask<Recorder, NoArgs> {
    explicit Recorder(AnalysisContext<NoArgs> &) {}
    void process_frame(AnalysisContext<NoArgs> &ctx) {
        seen.push_back(ctx.sys.state.get_time());
        seen_consumed.push_back(ctx.consumed_frames);
    }
    void post_process(AnalysisContext<NoArgs> &) { ++post_calls; }
    static std::string task_name() { return "recorder"; }
};

static std::vector<Float> run_with(std::vector<std::string> argv, MockSource &src) {
    seen.clear(); seen_consumed.clear(); post_calls = 0;
    Recorder::run(argv, src);
    return seen;
}

static void window_tests() {
    using V = std::vector<Float>;
    MockSource src;
    // single trajectory: random-access begin (:189-198), absolute exclusive end (:219-223)
    EXPECT(run_with({"-f", "top", "tA"}, src) == (V{0, 10, 20, 30, 40}));
    EXPECT(post_calls == 1 && src.topology_reads == 1);
    EXPECT((seen_consumed == std::vector<size_t>{0, 1, 2, 3, 4}));
    EXPECT(run_with({"-f", "top", "tA", "-b", "2"}, src) == (V{20, 30, 40}));
    EXPECT(run_with({"-f", "top", "tA", "-b", "1", "-e", "4"}, src) == (V{10, 20, 30}));
    EXPECT(run_with({"-f", "top", "tA", "-b", "15ps"}, src) == (V{20, 30, 40}));
    EXPECT(run_with({"-f", "top", "tA", "-e", "25ps"}, src) == (V{0, 10, 20}));
    // --skip counts from `begin`, the begin frame is always processed (:225-234)
    EXPECT(run_with({"-f", "top", "tA", "--skip", "2"}, src) == (V{0, 20, 40}));
    EXPECT(run_with({"-f", "top", "tA", "-b", "1", "--skip", "2"}, src) == (V{10, 30}));
    // several files = one continuous stream; begin filtered serially (:205-215), cadence continuous
    EXPECT(run_with({"-f", "top", "tA", "tB"}, src) == (V{0, 10, 20, 30, 40, 50, 60, 70}));
    EXPECT(run_with({"-f", "top", "tA", "tB", "-b", "3", "-e", "7"}, src) == (V{30, 40, 50, 60}));
    EXPECT(run_with({"-f", "top", "tA", "tB", "--skip", "3"}, src) == (V{0, 30, 60}));
    EXPECT(run_with({"-f", "top", "tA", "tB", "-b", "45ps"}, src) == (V{50, 60, 70}));
    // --use_struct_file: the structure's own state is frame one (:168-179)
    EXPECT(run_with({"-f", "top", "tA", "--use_struct_file", "-e", "2"}, src) == (V{-1, 0, 10}));
    EXPECT(run_with({"-f", "top", "--use_struct_file"}, src) == (V{-1}));
    // errors
    EXPECT(throws_kind([&] { run_with({"-f", "top"}, src); }, AnalysisError::NoTraj));                       // :139-141
    EXPECT(throws_kind([&] { run_with({"-f", "top", "tA", "-b", "99"}, src); }, AnalysisError::NoFramesConsumed));   // :275-277
    EXPECT(throws_kind([&] { run_with({"-f", "top", "tA", "-b", "5km"}, src); }, AnalysisError::InvalidSuffix));
    EXPECT(throws_kind([&] { run_with({"--skip", "2"}, src); }, AnalysisError::Arg));
    EXPECT(throws_kind([&] { run_with({"-f", "top", "tA", "--skip", "0"}, src); }, AnalysisError::Arg));
}

static void pbcdims_tests() {
    PbcDims d = PbcDims::make(true, false, true);
    EXPECT(d.get_dim(0) && !d.get_dim(1) && d.get_dim(2) && d.any());
    EXPECT(d != PBC_FULL && PbcDims::make(true, true, true) == PBC_FULL && !PBC_NONE.any());
    bool threw = false;
    try { d.get_dim(3); } catch (const std::out_of_range &) { threw = true; }
    EXPECT(threw);
    // periodic_box.rs:559-575 through the C++ PeriodicBox (host arithmetic of the engine)
    Matrix3f m;
    m(0, 0) = 10; m(0, 1) = 4; m(0, 2) = -4; m(1, 1) = 10; m(2, 2) = 10;
    const PeriodicBox pb = PeriodicBox::from_matrix(m);
    const Float dd = pb.distance({38.9214f, 40.0078f, -34.0795f}, {-26.6187f, 40.8926f, 30.9709f}, PBC_FULL);
    EXPECT(std::fabs(dd - 5.353627f) < 1e-3f);
    EXPECT(pb.is_triclinic() && pb.n_tric_corrections() > 0);
    threw = false;
    try { PeriodicBox::from_vectors_angles(10.0f, 0.2f, 15.0f, 90.0f, 9.0f, 90.0f); } catch (const PeriodicBoxError &e) { threw = e.code == PeriodicBoxError::AngleTooSmall; }
    EXPECT(threw);
}

// XtcReader on the reference's benzene.xtc (copy under tests/golden): 5 frames of 12 atoms, times 4032..4040
// (tests/test_netcdf.rs:37-80); handler semantics of xtc_handler.rs:200-229.
static void xtc_tests(const char *path) {
    XtcReader r(path);
    EXPECT(r.nframes() == 5 && r.natoms() == 12);
    int n = 0;
    Float last = 0;
    while (auto st = r.read_state()) {
        EXPECT(st->coords.size() == 12 && st->pbox.has_value());
        EXPECT(st->coords[0].x > 1.6f && st->coords[0].x < 1.7f);      // 1.659 nm in frame 0
        last = st->time;
        ++n;
    }
    EXPECT(n == 5 && last == 4040.0f);
    EXPECT(!r.read_state().has_value());                               // Eof stays Eof
    r.seek_frame(3);
    EXPECT(r.read_state()->time == 4038.0f);
    r.seek_time(4035.0f);                                              // first frame with time >= t
    EXPECT(r.read_state()->time == 4036.0f);
    bool threw = false;
    try { r.seek_time(1.0e6f); } catch (const MolarError &e) { threw = e.code == MOLAR_HIP_ERR_IO; }
    EXPECT(threw);
    threw = false;
    try { XtcReader bad("/nonexistent.xtc"); } catch (const MolarError &e) { threw = e.code == MOLAR_HIP_ERR_IO; }
    EXPECT(threw);
    auto next = XtcReader::open_as_source(path, std::optional<size_t>(4), std::nullopt);
    EXPECT(next()->time == 4040.0f && !next().has_value());
    std::vector<float> all(5 * 12 * 3);
    r.read_frames(0, 5, all.data(), nullptr, 2);
    EXPECT(std::fabs(all[0] - 1.659f) < 1e-6f);
    // XtcWriter (write_state, xtc_handler.rs:117-168): what it writes, the reader gives back on the format's 0.001 nm grid
    {
        const std::string out = std::string(path) + ".rewritten.tmp";
        {
            XtcWriter w(out);
            XtcReader src(path);
            while (auto st = src.read_state()) w.write_state(*st);
            EXPECT(w.nframes() == 5);
        }
        XtcReader a(path), b(out);
        EXPECT(b.nframes() == 5 && b.natoms() == 12);
        for (int k = 0; k < 5; ++k) {
            auto sa = a.read_state(), sb = b.read_state();
            EXPECT(sa->time == sb->time);
            for (int i = 0; i < 12; ++i)
                EXPECT(sa->coords[i].x == sb->coords[i].x && sa->coords[i].y == sb->coords[i].y && sa->coords[i].z == sb->coords[i].z);
            for (int q = 0; q < 9; ++q) EXPECT(sa->pbox->colmajor9()[q] == sb->pbox->colmajor9()[q]);
        }
        std::remove(out.c_str());
    }
}

// run_sharded with one reader per worker (FrameSource::frame_count + concurrent_open): a synthetic XTC trajectory of 250k atoms
// is consumed by 8 host-only workers (devices = -1: no engine context) - every worker reads its own contiguous block through its
// own reader - and by the single-reader form (the same source with concurrent_open() = false); both must give what run() gives,
// for every -b / -e / --skip window, and the readers' frames/s are printed: 8 consumers must not wait for one decoder thread.
struct SumArgs {
    explicit SumArgs(const std::vector<std::string> &) {}
};
struct SumTask : AnalysisTask<SumTask, SumArgs> {
    static std::vector<std::pair<size_t, double>> result;       // (frame index, sum of coordinates), frame order
    static size_t result_frames;
    std::vector<std::pair<size_t, double>> rows;
    explicit SumTask(AnalysisContext<SumArgs> &) {}
    void process_frame(AnalysisContext<SumArgs> &ctx) {
        double s = 0;
        for (const Pos &p : ctx.sys.state.coords) s += (double)p.x + (double)p.y + (double)p.z;
        rows.emplace_back(ctx.frame_index, s + (double)ctx.sys.state.time);
    }
    void merge(SumTask &&o) { rows.insert(rows.end(), o.rows.begin(), o.rows.end()); }
    void post_process(AnalysisContext<SumArgs> &ctx) {
        std::sort(rows.begin(), rows.end());
        result = rows;
        result_frames = ctx.consumed_frames;
    }
    static std::string task_name() { return "sum"; }
};
std::vector<std::pair<size_t, double>> SumTask::result;
size_t SumTask::result_frames = 0;

struct OneReaderXtc : XtcFrameSource {       // the same files through the single-reader form
    using XtcFrameSource::XtcFrameSource;
    bool concurrent_open() const override { return false; }
};

static void sharded_reader_tests(const char *dir) {
    const size_t natoms = 250000, nframes = 40;
    const std::string path = std::string(dir) + "/sharded_readers_250k.xtc";
    Topology top;
    top.masses.assign(natoms, 1.0f);
    top.vdw.assign(natoms, 0.15f);
    State base;
    base.coords.resize(natoms);
    uint32_t seed = 7u;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return (float)((seed >> 8) & 0xFFFFFF) / 16777216.0f; };
    for (auto &p : base.coords) { p.x = 13.5f * rnd(); p.y = 13.5f * rnd(); p.z = 13.5f * rnd(); }
    Matrix3f m{};
    m.m[0] = m.m[4] = m.m[8] = 13.5f;
    base.pbox = PeriodicBox::from_matrix(m);
    {
        XtcWriter w(path);
        State st = base;
        for (size_t f = 0; f < nframes; ++f) {
            st.time = (Float)(10 * f);
            for (size_t i = f % 7; i < natoms; i += 7) st.coords[i].x += 0.011f;       // frames differ
            w.write_state(st);
        }
    }
    XtcFrameSource own(top, base);
    OneReaderXtc one(top, base);
    const std::vector<int> eight(8, -1);
    using V = std::vector<std::string>;
    double fps_own = 0, fps_wall_own = 0, fps_wall_one = 0;
    for (const V &argv : {V{"-f", "top", path}, V{"-f", "top", path, "-b", "3", "-e", "37"}, V{"-f", "top", path, "--skip", "3"},
                          V{"-f", "top", path, "-b", "5", "--skip", "4", "-e", "33"}, V{"-f", "top", path, path, "-b", "30", "-e", "55"},
                          V{"-f", "top", path, "--use_struct_file", "-e", "9"}}) {
        SumTask::run(argv, own);
        const auto want = SumTask::result;
        const size_t want_frames = SumTask::result_frames;
        EXPECT(!want.empty());
        auto t0 = std::chrono::steady_clock::now();
        SumTask::run_sharded(argv, own, eight);
        const double dt_own = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        EXPECT(SumTask::result == want && SumTask::result_frames == want_frames);
        if (argv.size() == 3) { fps_own = SumTask::last_reader_fps(); fps_wall_own = (double)want.size() / dt_own; }
        t0 = std::chrono::steady_clock::now();
        SumTask::run_sharded(argv, one, eight, 2);
        const double dt_one = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        EXPECT(SumTask::result == want && SumTask::result_frames == want_frames);
        if (argv.size() == 3) fps_wall_one = (double)want.size() / dt_one;
    }
    // a time-based window keeps the single reader (and the same answer)
    SumTask::run({"-f", "top", path, "-b", "95ps"}, own);
    const auto want_t = SumTask::result;
    SumTask::run_sharded({"-f", "top", path, "-b", "95ps"}, own, eight);
    EXPECT(SumTask::result == want_t && want_t.size() == 30);
    std::printf("run_sharded, 8 host-only workers, %zu frames of %zu atoms: own readers %.0f frames/s end to end (readers deliver %.0f frames/s "
                "summed over workers), single reader %.0f frames/s end to end, %u hardware threads\n",
                nframes, natoms, fps_wall_own, fps_own, fps_wall_one, std::thread::hardware_concurrency());
    // Not reader-bound: eight readers must beat the one producer thread clearly - on a host whose threads scale at all.  (Some
    // CI containers have eight oversubscribed vCPUs on which four decoder threads are no faster than one; calibrate with the
    // library's own multi-threaded window decode and skip the assertion there, numbers printed either way.)
    double scale = 0;
    {
        XtcReader r(path);
        std::vector<float> buf(16 * natoms * 3);
        auto t0 = std::chrono::steady_clock::now();
        r.read_frames(0, 16, buf.data(), nullptr, 1);
        const double t1 = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        t0 = std::chrono::steady_clock::now();
        r.read_frames(16, 16, buf.data(), nullptr, 4);
        const double t4 = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        scale = t1 / t4;
        std::printf("window decode of 16 frames: 1 thread %.0f frames/s, 4 threads %.0f frames/s (x %.2f)\n", 16 / t1, 16 / t4, scale);
    }
    if (scale >= 2.0) EXPECT(fps_wall_own > 1.5 * fps_wall_one);
    else std::printf("host threads do not scale here: the own-readers-beat-one-reader assertion is skipped\n");
    std::remove(path.c_str());
}

int main(int argc, char **argv) {
    suffix_tests();
    window_tests();
    pbcdims_tests();
    if (argc > 1) xtc_tests(argv[1]);
    if (argc > 2) sharded_reader_tests(argv[2]);
    if (failures) { std::printf("%d failure(s)\n", failures); return 1; }
    std::printf("all host-mirror CPU tests passed\n");
    return 0;
}
