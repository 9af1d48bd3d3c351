// hoststream.hpp - results from HBM into the caller's (pageable) host memory at PCIe speed.
//
// MolAR's drivers return host Vecs (distance_search.rs:928-954): for a caller that keeps its buffers in ordinary memory the
// 4.3 GB pair list of the headline frame is the whole cost of a call.  hipMemcpy into pageable memory is staged by the
// runtime through one bounce buffer on one thread (16 GB/s measured, 0.26 s per frame).  Here the device-to-host
// transfer runs chunk by chunk into a ring of pinned buffers owned by the context while a few host threads empty the
// ring into the caller's arrays - and, for the usize-typed entry points, widen u32 -> u64 on the way, so that the
// link carries 12 bytes per pair instead of 20.
#pragma once

#include <algorithm>
#include <atomic>
#include <cstring>
#include <thread>
#include <vector>

#include "common.hpp"

namespace mh {

constexpr size_t RING_CHUNK = 16u << 20;      // bytes per pinned chunk
constexpr int RING_SLOTS = 8;                 // chunks in flight (128 MiB of pinned memory per context, allocated on first use)
constexpr int RING_WORKERS = 8;               // host threads emptying the ring

enum RingConv { RING_COPY = 0, RING_PAIRS_TO_U64 = 1, RING_U32_TO_U64 = 2 };

// one array to bring home: `bytes` of device memory at `src`; RING_COPY: to dst0; RING_PAIRS_TO_U64: (u32,u32) records to
// the two u64 arrays dst0 / dst1 (either may be null); RING_U32_TO_U64: u32 values to the u64 array dst0
struct RingJob {
    const void *src;
    size_t bytes;
    int conv;
    void *dst0, *dst1;
};

inline void ring_release(molar_hip_ctx *c) {
    for (int k = 0; k < RING_SLOTS; ++k) {
        if (c->ring[k]) (void)hipHostFree(c->ring[k]);
        if (c->ring_ev[k]) (void)hipEventDestroy(c->ring_ev[k]);
        c->ring[k] = nullptr;
        c->ring_ev[k] = nullptr;
    }
}

// all slots and events, or none: an allocation that fails half way (128 MiB of pinned memory per context) releases what it
// got, so that the next large fill tries again instead of finding ring[0] set and copying through null slots
inline int ring_ensure(molar_hip_ctx *c) {
    bool whole = true;
    for (int k = 0; k < RING_SLOTS; ++k) whole = whole && c->ring[k] && c->ring_ev[k];
    if (whole) return 0;
    ring_release(c);
    for (int k = 0; k < RING_SLOTS; ++k) {
        hipError_t e = hipHostMalloc(&c->ring[k], RING_CHUNK, hipHostMallocDefault);
        if (e != hipSuccess) c->ring[k] = nullptr;
        if (e == hipSuccess) {
            e = hipEventCreateWithFlags(&c->ring_ev[k], hipEventDisableTiming);
            if (e != hipSuccess) c->ring_ev[k] = nullptr;
        }
        if (e != hipSuccess) {
            ring_release(c);
            MH_HIP(e);
        }
    }
    return 0;
}

// Brings the jobs home, in order, on c->stream (so everything enqueued before - the fill kernel - is complete for them);
// returns when the caller's arrays are written.
inline int ring_to_host(molar_hip_ctx *c, const std::vector<RingJob> &jobs) {
    struct Chunk {
        const char *src;
        size_t bytes, off;      // off: byte offset of the chunk inside its job's source
        int job;
    };
    std::vector<Chunk> chunks;
    for (size_t j = 0; j < jobs.size(); ++j) {
        if (!jobs[j].src || jobs[j].bytes == 0) continue;
        for (size_t off = 0; off < jobs[j].bytes; off += RING_CHUNK)
            chunks.push_back(Chunk{static_cast<const char *>(jobs[j].src) + off, std::min(RING_CHUNK, jobs[j].bytes - off), off, (int)j});
    }
    if (chunks.empty()) return 0;
    MH_TRY(ring_ensure(c));
    const size_t n = chunks.size();
    std::vector<std::atomic<int>> enq(n), done(n);
    for (size_t i = 0; i < n; ++i) {
        enq[i].store(0, std::memory_order_relaxed);
        done[i].store(0, std::memory_order_relaxed);
    }
    std::atomic<size_t> next{0};
    std::atomic<int> failed{0};
    const int device = c->device;
    auto worker = [&]() {
        (void)hipSetDevice(device);
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= n) return;
            while (!enq[i].load(std::memory_order_acquire)) {
                if (failed.load(std::memory_order_relaxed)) return;
                std::this_thread::yield();
            }
            const int slot = (int)(i % RING_SLOTS);
            if (hipEventSynchronize(c->ring_ev[slot]) != hipSuccess) failed.store(1);
            const Chunk &ch = chunks[i];
            const RingJob &J = jobs[ch.job];
            const char *s = static_cast<const char *>(c->ring[slot]);
            if (J.conv == RING_COPY) {
                std::memcpy(static_cast<char *>(J.dst0) + ch.off, s, ch.bytes);
            } else if (J.conv == RING_PAIRS_TO_U64) {
                const uint32_t *p = reinterpret_cast<const uint32_t *>(s);
                const size_t first = ch.off / 8, cnt = ch.bytes / 8;
                uint64_t *oi = J.dst0 ? static_cast<uint64_t *>(J.dst0) + first : nullptr;
                uint64_t *oj = J.dst1 ? static_cast<uint64_t *>(J.dst1) + first : nullptr;
                if (oi) for (size_t k = 0; k < cnt; ++k) oi[k] = p[2 * k];
                if (oj) for (size_t k = 0; k < cnt; ++k) oj[k] = p[2 * k + 1];
            } else {
                const uint32_t *p = reinterpret_cast<const uint32_t *>(s);
                const size_t first = ch.off / 4, cnt = ch.bytes / 4;
                uint64_t *o = static_cast<uint64_t *>(J.dst0) + first;
                for (size_t k = 0; k < cnt; ++k) o[k] = p[k];
            }
            done[i].store(1, std::memory_order_release);
        }
    };
    const int nw = (int)std::min<size_t>(RING_WORKERS, n);
    std::vector<std::thread> pool;
    pool.reserve(nw);
    for (int w = 0; w < nw; ++w) pool.emplace_back(worker);
    hipError_t err = hipSuccess;
    for (size_t i = 0; i < n && err == hipSuccess; ++i) {
        if (i >= (size_t)RING_SLOTS)
            while (!done[i - RING_SLOTS].load(std::memory_order_acquire)) {
                if (failed.load(std::memory_order_relaxed)) break;
                std::this_thread::yield();
            }
        if (failed.load(std::memory_order_relaxed)) break;
        const int slot = (int)(i % RING_SLOTS);
        err = hipMemcpyAsync(c->ring[slot], chunks[i].src, chunks[i].bytes, hipMemcpyDeviceToHost, c->stream);
        if (err == hipSuccess) err = hipEventRecord(c->ring_ev[slot], c->stream);
        if (err == hipSuccess) enq[i].store(1, std::memory_order_release);
    }
    if (err != hipSuccess) failed.store(1);
    for (auto &t : pool) t.join();
    if (err != hipSuccess) return fail(MOLAR_HIP_ERR_HIP, "device-to-host stream: %s", hipGetErrorString(err));
    if (failed.load()) return fail(MOLAR_HIP_ERR_HIP, "device-to-host stream: event wait failed");
    return 0;
}

// true for host memory the runtime can DMA into directly (hipHostMalloc / hipHostRegister): one plain async copy is best
inline bool is_pinned_host(const void *p) {
    if (!p) return false;
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

}  // namespace mh
