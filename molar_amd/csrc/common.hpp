// common.hpp — context, device buffers and error plumbing of libmolar_hip.so.
// gfx950 only; host side is plain C++17 over the HIP runtime (no torch, no third-party libs).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/molar_hip.h"

namespace mh {

// ---------------------------------------------------------------- errors (tpr_last_error idiom, wrapper.cpp:32,156)

inline std::string &last_error() {
    static thread_local std::string e;
    return e;
}

inline int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    last_error() = buf;
    return code;
}

#define MH_HIP(expr)                                                                                   \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess)                                                                          \
            return ::mh::fail(MOLAR_HIP_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                              __FILE__, __LINE__);                                                     \
    } while (0)

#define MH_TRY(expr)          \
    do {                      \
        int _rc = (expr);     \
        if (_rc) return _rc;  \
    } while (0)

// ---------------------------------------------------------------- grow-only device buffer

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) {
            MH_HIP(hipFree(p));
            p = nullptr;
            cap = 0;
        }
        size_t want = bytes + bytes / 8 + 256;   // headroom: frames of one trajectory vary a little
        MH_HIP(hipMalloc(&p, want));
        cap = want;
        return 0;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T>
    T *as() const { return reinterpret_cast<T *>(p); }
};

// true if `p` is device-accessible HIP memory (device or managed); false for ordinary host memory
inline bool is_device_ptr(const void *p) {
    if (!p) return false;
    hipPointerAttribute_t a;
    hipError_t e = hipPointerGetAttributes(&a, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();   // unregistered host memory reports an error on some ROCm versions
        return false;
    }
    return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged || a.type == hipMemoryTypeArray;
}

// ---------------------------------------------------------------- per-set grid state of the cached search

struct GridSet {
    uint32_t n = 0;            // selected atoms handed in
    DevBuf xyz_stage;          // host xyz staged here
    DevBuf idx_stage;          // host idx staged here (uint64)
    DevBuf vdw_stage;
    const float *d_xyz = nullptr;
    const uint64_t *d_idx = nullptr;
    const float *d_vdw = nullptr;
    DevBuf key;                // u32 per atom: cell<<1 | wrapped, 0xFFFFFFFF = dropped
    DevBuf cell_count;         // u32 [ncells+1] -> scanned in place into cell_start
    DevBuf cnt_pad;            // cell counters at one per 128-byte line while binning (crowded grids)
    DevBuf cursor;             // u32 per atom: arrival order inside its cell
    DevBuf tmp_key;            // u32 per kept atom (unsorted inside the cell)
    DevBuf sort_buf;           // grids of large cells: keys and atom numbers, unsorted and sorted (4 x u32 per atom; devsort.hip)
    DevBuf sorted;             // float4 {x,y,z,id-bits} in reference cell order
    DevBuf sorted_vdw;         // float per sorted atom (vdw searches)
    DevBuf aabb;               // float4 lo/hi per cell
    DevBuf perm;               // float4 per sorted atom: the cell in Morton order, {x,y,z,position} (count pass of the fast path)
    DevBuf chunk_aabb;         // float4 lo/hi per 64-atom Morton chunk, slot (cell_start >> 6) + cell + k
    DevBuf h16;                // 8 x f16 per sorted atom (the reference's cell order, like `sorted`): hi/lo split of the position relative to the
                               // cell's origin and of its squared norm - the B operand of the matrix-core count pass
    DevBuf cell_org;           // float4 per cell: origin (centre of the bounding box), .w = bound on |position - origin|
};

}  // namespace mh

struct molar_hip_search64_state;        // the cached f64 search (search_f64.hip)

// frames per launch of molar_hip_search_histogram_frames (search.hip, hist_frames_group)
#ifndef MH_HIST_BATCH
#define MH_HIST_BATCH 16
#endif

struct molar_hip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int num_cus = 256;

    // pinned host scratch for small read-backs
    void *h_pinned = nullptr;
    size_t h_pinned_cap = 0;
    molar_hip_search64_state *s64 = nullptr;      // created by the first molar_hip_search_count_f64
    // ring of pinned chunks for large results that go to pageable host memory (hoststream.hpp), allocated on first use
    void *ring[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ring_ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};

    // ---- cached search (count -> fill)
    bool have_search = false;
    int kind = 0;
    bool use_box = false;
    uint8_t pbc = 0;
    float cutoff = 0.f;
    molar_hip_box box{};
    float lower[3]{}, upper[3]{};
    uint32_t dims[3]{1, 1, 1};
    uint64_t ntasks = 0;
    uint64_t total = 0;
    // Two generations of the grids: the pipelined resident search (molar_hip_search_resident_begin) builds the grid of
    // the next frame on `side_stream` while the pair kernels of the frame before it still read theirs.
    mh::GridSet set_store[2][2];
    mh::GridSet *set = set_store[0];     // the generation the cached search refers to (set[0], set[1]: first / second set)
    hipStream_t side_stream = nullptr;   // grid build of a pipelined search (created on first use, highest priority)
    hipEvent_t grid_done = nullptr;
    bool want_side = false;              // set by _begin / the asynchronous histogram call around their enqueue
    bool env_no_side = false;            // MOLAR_HIP_NO_SIDE_STREAM, read once in molar_hip_create
    bool env_no_mfma = false;            // MOLAR_HIP_NO_MFMA_COUNT: count pass of plain entries on the vector ALUs (A/B runs)
    bool env_no_mfma_wrapped = false;    // MOLAR_HIP_NO_MFMA_WRAPPED: count pass of wrapped entries on the vector ALUs (A/B runs)
    bool env_host_grid_wait = false;     // MOLAR_HIP_HOST_GRID_WAIT: the host, not the main stream, waits for the side stream's grid
    bool env_no_tile_sum = false;        // MOLAR_HIP_NO_TILE_SUM: slot offsets by the general three-kernel scan (A/B runs)
    uint32_t env_debug_skip = 0;         // MOLAR_HIP_DEBUG_SKIP (builds with -DMOLAR_HIP_DEBUG_KNOBS only), read once
    uint32_t env_debug_launch = 0, dbg_launches = 0;   // MOLAR_HIP_DEBUG_LAUNCH (same builds): which histogram launch records its wave times
    hipEvent_t gen_free[2] = {nullptr, nullptr};   // recorded on the main stream behind the last asynchronous reader of
                                                   // a grid generation (histogram calls that do not wait)
    hipEvent_t side_wait = nullptr;      // what the side stream has to wait for before it rebuilds the generation
    hipEvent_t side_wait2 = nullptr;     // ... and a second event (the count pass of the frame in flight, see count_done)
    hipEvent_t count_done = nullptr;     // recorded behind the count pass of every pipelined search (molar_hip_search_resident_begin)
    bool count_done_set = false;
    bool record_count_done = false;      // set around the enqueue of a pipelined search
    bool env_grid_late = false;          // MOLAR_HIP_GRID_LATE: the next frame's grid waits for the count pass of the frame in flight (A/B runs)
    bool env_no_bin_tile = false;        // MOLAR_HIP_NO_BIN_TILE: the grid's binning with one global atomic per atom (A/B runs)
    int hist_gen = 0;                    // generation of the last asynchronous histogram call
    bool on_side = false;                // launches currently go to side_stream (scans then use scan_tmp_side)
    mh::DevBuf scan_tmp_side;
    uint64_t nslots_bound = 0; // host-side upper bound of the slot count (sizes the launches)
    mh::DevBuf params;         // SearchParams block read by the pair kernels
    mh::DevBuf task_desc;      // TaskDesc per task (plan entry)
    mh::DevBuf task_nb;        // u32 per task (+1): 64-row blocks of the task, scanned in place -> first slot
    mh::DevBuf slot_desc;      // SlotDesc per slot (+1): what a wave of the pair kernels needs to start
    mh::DevBuf slot_cnt;       // u32 per slot (+1): results of the slot
    mh::DevBuf slot_base;      // u64 per slot (+1): output offset (last = grand total)
    mh::DevBuf tile_sum;       // u64 per tile of 256 slots: results of the tile (tile_sums_kernel -> slot_offsets_kernel)
    unsigned long long plan_out_cap = ~0ull;   // resident searches: the output capacity the plan kernel writes into the parameter block
    bool params_fresh = false;                 // the plan kernel of this search left a parameter block the count pass can use as it is
    unsigned long long params_fresh_cap = 0;   // ... written with this output capacity
    unsigned long long *sizes_dev = nullptr;   // resident searches: device-side address of the pinned 16 bytes the kernels write the two sizes to
    mh::DevBuf scan_tmp;       // block sums for the scans
    mh::DevBuf sort_tmp, sort_tmp_side;   // scratch of the device sort (devsort.hip), per stream
    mh::DevBuf scan_state;     // ticket + tile descriptors of the single-pass scans (zeroed by the plan kernel)
    mh::DevBuf fplan_tiles;    // plan_tiles_kernel -> plan_slots_kernel: slot / hit-history totals per tile of 256 plan entries
    mh::DevBuf out_pairs_set[2];   // ctx-owned result buffers (device-resident results / host staging); the second
    mh::DevBuf out_dist_set[2];    // set exists for the pipelined begin/end searches only
    mh::DevBuf &out_pairs = out_pairs_set[0];
    mh::DevBuf &out_dist = out_dist_set[0];
    // pipelined resident searches (molar_hip_search_resident_begin/_end): two result sets, two tickets
    struct Ticket {
        bool pending = false, degenerate = false;
        unsigned long long cap0 = 0, maskcap0 = 0, serial = 0, launched = 0, ntasks = 0;
        int kind = -1;
        unsigned long long occ_key = 0;
        hipEvent_t done = nullptr;
        molar_hip_search_desc desc{};
    } tickets[2];
    int next_ticket = 0;
    bool resident_no_dist = false;          // molar_hip_search_resident_planes: the resident searches fill the (i, j) plane only
    unsigned long long search_serial = 0;   // counts resident searches enqueued on this context
    // Resident searches launch their count and fill passes over the slots the plan of the SEARCH BEFORE came to (+ 3 % + 512)
    // instead of the host's bound (14 N / 64 + entries: 13 % above the real count on the headline frame - 3.7e4 workgroups per
    // pass that only find out they have nothing to do); the plan writes its real slot count beside the other sizes, and a
    // search that needed more slots than were launched is repeated with the bound where its sizes are read.
    unsigned long long trim_real = 0, trim_ntasks = 0;   // slots of the last resident search whose sizes were read, and its plan
    int trim_kind = -1;
    uint32_t slot_launch = 0;               // slots the passes of the resident search being enqueued launch (0: the bound)
    void *h_sizes = nullptr;                // pinned: 32 bytes of result sizes per ticket (total, history units, slots, occupied cells)
    // Occupied cells of the two sets' grids as the last finished search of this shape found them (the grid build counts them, the
    // offsets kernel hands them to the host with the sizes): small_cell_lanes() judges a frame by its atoms per OCCUPIED cell when it
    // knows - a slab or a solute in a mostly empty periodic box has a small mean and crowded cells.  occ_use: latched per search.
    unsigned long long occ_key = 0;
    uint32_t occ_cells[2] = {0, 0};
    bool occ_valid = false;
    uint32_t occ_use[2] = {0, 0};           // 0: not known for the search in hand (the grid's cell count stands in)
    mh::DevBuf out_ids;
    mh::DevBuf wide_i, wide_j; // usize widening
    mh::DevBuf hist;           // u64 bins
    mh::DevBuf dbg;            // builds with -DMOLAR_HIP_DEBUG_KNOBS: per-wave time accounting of hist_kernel
    mh::DevBuf hist_queue;     // slot queues and list counters of the fused histogram (hist_kernels.hpp)
    mh::DevBuf slot_desc_rest; // fused histogram: records of the slots the generic kernel takes (hist_plan_kernel)
    unsigned long long hist_frames = 0;   // fused-histogram launches of this context (mod 4: which pair of list counters a launch uses)
    // molar_hip_search_histogram_frames: groups of frames through one set of launches (search.hip, hist_frames_group)
    mh::GridSet hb_sets[2][MH_HIST_BATCH][2];           // grids of a group's frames (first / second set), two generations
    bool hb_zeroed[2][MH_HIST_BATCH][2] = {};           // a set's padded cell counters are zero (the frames' unpad kernel leaves them so)
    uint32_t hb_zeroed_cells[2][MH_HIST_BATCH][2] = {}; // ... for a grid of this many cells
    mh::DevBuf hb_lean[2], hb_rest[2];    // the group's two slot lists
    mh::DevBuf hb_blocks[2];              // [parameter blocks | grid records] of the group
    void *hb_pin = nullptr;               // pinned staging of those records: four slots
    hipEvent_t hb_pin_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    bool hb_pin_used[4] = {false, false, false, false};
    unsigned hb_pin_next = 0;
    bool hist_plan_now = false;           // set around prepare_search by the fused histogram of the fixed-cutoff kinds
    mh::DevBuf hist_edges;     // f32[nbins + 1]: smallest d2 that reaches each bin (hist_kernel), for the cached (min, max, nbins)
    float edges_min = 0.f, edges_max = 0.f;
    size_t edges_nbins = 0;    // 0: no table cached
    // SearchConnectivity on the device (molar_hip_search_connectivity / _fill)
    mh::DevBuf conn_deg, conn_off, conn_ent, conn_neigh;
    uint64_t conn_rows = 0, conn_entries = 0;
    bool have_conn = false;
    // molar_hip_within_hold: reuse of the first set's staged coordinates and grid across `within` requests
    bool within_hold = false, hold_valid = false;
    mh::GridSet *hold_set = nullptr;
    uint64_t hold_fp = 0;      // fingerprint of a host-memory first set as it was staged (0: device memory, read in place)
    const float *hold_xyz = nullptr;
    const uint64_t *hold_idx = nullptr;
    size_t hold_natoms = 0, hold_n = 0;
    int hold_ids_local = 0;
    bool hold_use_box = false;
    uint8_t hold_pbc = 0;
    molar_hip_box hold_box{};
    uint32_t hold_dims[3]{0, 0, 0};
    float hold_lower[3]{}, hold_upper[3]{};
    // `within` as a set (molar_hip_within_count / _fill)
    mh::DevBuf w_list;         // small second sets: ids found, in the order they were found (within_small_kernel)
    bool w_small = false;      // the cached within set was made by the small path (the list holds it)
    uint64_t w_list_dirty = 0; // flags set through the list by the previous small-path call, to be cleared before the next
    bool w_flags_all_dirty = false;   // the previous call went through the partner lists: any flag may be set
    mh::DevBuf w_flags, w_part_cnt, w_part, w_tile_cnt, w_tile_off;
    uint64_t within_nflags = 0, within_total = 0;
    bool have_within = false;
    bool skip_plan = false;    // set around prepare_search by callers that do not walk slots
    mh::DevBuf task_mu;        // u32 per task (+1): 64-word hit-history units of the task's slots
    mh::DevBuf task_moff;      // u64 per task (+1): exclusive scan of task_mu
    mh::DevBuf maskbuf;        // hit bits recorded by the count pass, replayed by the fill pass
    uint64_t mask_units = 0;

    // molar_hip_xtc_histogram: a second context for the decoder thread (its own stream and pinned staging), two windows of frames, bins + index
    molar_hip_ctx *aux = nullptr;
    mh::DevBuf xh_win[2], xh_bins;

    // ---- profiling (HIP events on `stream`)
    bool profiling = false;
    bool profile_frames = false;   // molar_hip_profile_enable(ctx, 2): ONE span (class 5) around count + offsets + fill of a resident search, none inside
    struct Span { int cls; hipEvent_t a, b; };
    std::vector<Span> spans;
    std::vector<hipEvent_t> event_pool;

    mh::DevBuf fit_redo;       // membrane fit: lipids k_membrane_fit_lanes hands to the one-lane kernel
    // ---- measure scratch
    mh::DevBuf m_xyz1, m_xyz2, m_idx1, m_idx2, m_mass1, m_mass2, m_partials, m_results, m_out;
};

namespace mh {

// RAII span: records an event pair around a group of launches when profiling is on
struct Prof {
    molar_hip_ctx *c;
    int cls;
    hipEvent_t a = nullptr, b = nullptr;
    static hipEvent_t get(molar_hip_ctx *c) {
        if (!c->event_pool.empty()) {
            hipEvent_t e = c->event_pool.back();
            c->event_pool.pop_back();
            return e;
        }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
    Prof(molar_hip_ctx *ctx, int k) : c(ctx), cls(k) {
        if (!c->profiling) return;
        if (c->profile_frames ? (k >= 1 && k <= 3) : k == 5) return;      // frame mode: the three passes share the frame's span
        a = get(c);
        b = get(c);
        (void)hipEventRecord(a, c->stream);
    }
    ~Prof() {
        if (!c->profiling || !a) return;
        (void)hipEventRecord(b, c->stream);
        c->spans.push_back({cls, a, b});
    }
};

inline int ensure_pinned(molar_hip_ctx *c, size_t bytes) {
    if (bytes <= c->h_pinned_cap) return 0;
    if (c->h_pinned) (void)hipHostFree(c->h_pinned);
    c->h_pinned = nullptr;
    c->h_pinned_cap = 0;
    MH_HIP(hipHostMalloc(&c->h_pinned, bytes + 4096, hipHostMallocDefault));
    c->h_pinned_cap = bytes + 4096;
    return 0;
}

// Make `src` (host or device, `bytes` long) readable by kernels: device pointers are used in
// place, host buffers are copied into `stage` on the context's stream.
template <class T>
inline int to_device(molar_hip_ctx *c, const T *src, size_t count, DevBuf &stage, const T **out) {
    if (!src || count == 0) {
        *out = nullptr;
        return 0;
    }
    if (is_device_ptr(src)) {
        *out = src;
        return 0;
    }
    MH_TRY(stage.reserve(count * sizeof(T)));
    MH_HIP(hipMemcpyAsync(stage.p, src, count * sizeof(T), hipMemcpyHostToDevice, c->stream));
    *out = stage.as<T>();
    return 0;
}

}  // namespace mh
