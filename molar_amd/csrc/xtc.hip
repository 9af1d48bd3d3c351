// xtc.hip — XTC trajectory frames for the engine: index once, decode many frames in parallel on host
// threads, hand them to the GPU (pinned staging + async copies on the context's stream).
//
// Replaces, for the accelerated path, what MolAR gets from the `molly` crate through
// molar/src/io/xtc_handler.rs: read_state (:64-112), seek_frame (:200-218), seek_time (:220-229, 282-297).
// The format is GROMACS XTC (magic 1995, and 2023 with a 64-bit byte count).  A frame's bit stream is
// inherently serial (adaptive small-delta index, run lengths), so the parallel axis is the frame: a 1M-atom
// frame decodes in a few ms on one core, and T host threads keep a GPU that consumes ~400 frames/s fed.
//   * bit reader: 64-bit accumulator refilled 32 bits at a time;
//   * packed triples: the mixed-radix number is assembled little-endian into 64 (or 128) bits and split with two
//     divisions, instead of byte-wise long division;
//   * small deltas use a per-index reciprocal (exact for the 24-bit operands of the format).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <thread>
#include <vector>

#include "common.hpp"

namespace {

using namespace mh;

constexpr int FIRSTIDX = 9;
#define MOLAR_XTC_MAGIC_TABLE                                                                                             \
    0,       0,       0,       0,       0,       0,       0,       0,       0,       8,        10,       12,       16,       \
    20,      25,      32,      40,      50,      64,      80,      101,     128,     161,      203,      256,      322,      \
    406,     512,     645,     812,     1024,    1290,    1625,    2048,    2580,    3250,     4096,     5060,     6501,     \
    8192,    10321,   13003,   16384,   20642,   26007,   32768,   41285,   52015,   65536,    82570,    104031,   131072,   \
    165140,  208063,  262144,  330280,  416127,  524287,  660561,  832255,  1048576, 1321122,  1664510,  2097152,  2642245,  \
    3329021, 4194304, 5284491, 6658042, 8388607, 10568983, 13316085, 16777216
constexpr int MAGIC[] = {MOLAR_XTC_MAGIC_TABLE};
__device__ __attribute__((unused)) const int MAGIC_DEV[] = {MOLAR_XTC_MAGIC_TABLE};       // the same table for the device decoder (xtc_decode_kernel)
constexpr int LASTIDX = (int)(sizeof(MAGIC) / sizeof(*MAGIC)) - 1;
__host__ __device__ inline int magic_at(int k) {
#ifdef __HIP_DEVICE_COMPILE__
    return MAGIC_DEV[k];
#else
    return MAGIC[k];
#endif
}

__host__ __device__ inline uint32_t be32(const uint8_t *p) {
#ifdef __HIP_DEVICE_COMPILE__
    // (a window copied to the device starts at a frame header, so words are 4-byte aligned there too - but a misaligned
    // dword load faults on the device where the host merely slows down: byte loads)
    return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3];
#else
    uint32_t w;
    std::memcpy(&w, p, 4);
    return __builtin_bswap32(w);
#endif
}
__host__ __device__ inline float bef(const uint8_t *p) {
    const uint32_t u = be32(p);
    return __builtin_bit_cast(float, u);
}

struct FrameInfo {
    uint64_t offset, data_off, nbytes;
    int32_t natoms, step;
    float time, precision, box[9];
};

// MSB-first bit reader over [p, end); reads past the end deliver zeros and show up in consumed_bits()
struct Bits {
    const uint8_t *p, *end;
    uint64_t acc = 0;      // valid bits are the top `have` bits
    int have = 0;
    uint64_t loaded = 0;   // bits moved into acc so far (phantom tail bits included)
    __host__ __device__ inline void refill() {
        while (have <= 32) {
            uint32_t w;
            if (p + 4 <= end) {
                w = be32(p);
                p += 4;
            } else {
                w = 0;
                for (int k = 0; k < 4 && p < end; ++k) w |= (uint32_t)*p++ << (24 - 8 * k);
            }
            acc |= (uint64_t)w << (32 - have);
            have += 32;
            loaded += 32;
        }
    }
    __host__ __device__ inline uint32_t get(int n) {   // 0 <= n <= 32
        if (n == 0) return 0;
        if (have < n) refill();
        const uint32_t v = (uint32_t)(acc >> (64 - n));
        acc <<= n;
        have -= n;
        return v;
    }
    __host__ __device__ inline uint64_t consumed_bits() const { return loaded - (uint64_t)have; }
};

// number of bits needed for the product of three sizes (the format's `sizeofints`)
__host__ __device__ inline int bits_for_product(const uint32_t s[3]) {
    const unsigned __int128 prod = (unsigned __int128)s[0] * s[1] * s[2];
    // xdrfile counts bits of the little-endian byte array holding prod: 8*(nbytes-1) + bits of the top byte,
    // where the top byte's count is "while (top >= num) {nbits++; num*=2}" = bit length of top
    int nbytes = 0;
    unsigned __int128 t = prod;
    uint32_t top = 0;
    while (t) {
        top = (uint32_t)(t & 0xff);
        t >>= 8;
        ++nbytes;
    }
    if (nbytes == 0) return 0;
    int tb = 0;
    while (top >> tb) ++tb;
    return tb + 8 * (nbytes - 1);
}

__host__ __device__ inline int bits_for(uint32_t size) {
    int n = 0;
    uint64_t num = 1;
    while (size >= num && n < 32) {
        ++n;
        num <<= 1;
    }
    return n;
}

// Mixed-radix triple packed in `nbits` bits: the stream holds the number little-endian by bytes.
__host__ __device__ inline bool unpack3(Bits &b, int nbits, const uint32_t s[3], int out[3]) {
    if (nbits <= 64) {
        uint64_t v = 0;
        int shift = 0, left = nbits;
        while (left >= 32) {
            v |= (uint64_t)__builtin_bswap32(b.get(32)) << shift;
            shift += 32;
            left -= 32;
        }
        while (left > 8) {          // xdrfile: full bytes while MORE than 8 bits remain, then the rest in one piece
            v |= (uint64_t)b.get(8) << shift;
            shift += 8;
            left -= 8;
        }
        if (left > 0) v |= (uint64_t)b.get(left) << shift;
        const uint64_t q = v / s[2];
        out[2] = (int)(v - q * s[2]);
        const uint64_t q2 = q / s[1];
        out[1] = (int)(q - q2 * s[1]);
        out[0] = (int)q2;
        return true;
    }
#ifdef __HIP_DEVICE_COMPILE__
    return false;            // a triple wider than 64 bits (coordinates spanning > 2^21 grid units): the host decoder takes the frame
#else
    unsigned __int128 v = 0;
    int shift = 0, left = nbits;
    while (left > 8) {
        v |= (unsigned __int128)b.get(8) << shift;
        shift += 8;
        left -= 8;
    }
    if (left > 0) v |= (unsigned __int128)b.get(left) << shift;
    const unsigned __int128 q = v / s[2];
    out[2] = (int)(uint64_t)(v - q * s[2]);
    const unsigned __int128 q2 = q / s[1];
    out[1] = (int)(uint64_t)(q - q2 * s[1]);
    out[0] = (int)(uint64_t)q2;
    return true;
#endif
}

// Small-delta triple: all three radices equal `m` (< 2^24) and nbits <= 72; with m^3 < 2^64 for every index
// below 2^21, and the 128-bit path otherwise.
__host__ __device__ inline bool unpack3_small(Bits &b, int nbits, uint32_t m, int out[3]) {
    const uint32_t s[3] = {m, m, m};
    return unpack3(b, nbits, s, out);
}

// Integer coordinates of a corrupt stream can be anything: additions wrap (two's complement) instead of overflowing a
// signed int - the same values for every well-formed file, defined behaviour for the others (found by UBSan over
// tests/test_xtc_cpu.py::test_corrupt_streams_do_not_crash, tools/asan_host.sh).
__host__ __device__ inline int wrap_add(int a, int b) { return (int)((uint32_t)a + (uint32_t)b); }

// returns 0, or: 2-6 corrupt stream, 7 (device only) the frame needs the host decoder
__host__ __device__ int decode_frame(const uint8_t *file, const FrameInfo &fi, float *out) {
    const int natoms = fi.natoms;
    const uint8_t *h = file + fi.offset;
    if (natoms <= 9) {
        for (int k = 0; k < 3 * natoms; ++k) out[k] = bef(h + 56 + 4 * k);
        return 0;
    }
    int minint[3], maxint[3];
    for (int k = 0; k < 3; ++k) {
        minint[k] = (int32_t)be32(h + 60 + 4 * k);
        maxint[k] = (int32_t)be32(h + 72 + 4 * k);
    }
    int smallidx = (int32_t)be32(h + 84);
    if (smallidx < FIRSTIDX || smallidx > LASTIDX) return 2;
    uint32_t sizeint[3];
    int bitsizeint[3] = {0, 0, 0}, bitsize;
    for (int k = 0; k < 3; ++k) sizeint[k] = (uint32_t)maxint[k] - (uint32_t)minint[k] + 1u;
    if (sizeint[0] == 0u || sizeint[1] == 0u || sizeint[2] == 0u) return 2;      // maxint = minint - 1: no well-formed frame has that (and the radix would divide by zero)
    if ((sizeint[0] | sizeint[1] | sizeint[2]) > 0xffffffu) {
        for (int k = 0; k < 3; ++k) bitsizeint[k] = bits_for(sizeint[k]);
        bitsize = 0;
    } else {
        bitsize = bits_for_product(sizeint);
    }
    int smaller = magic_at(smallidx - 1 > FIRSTIDX ? smallidx - 1 : FIRSTIDX) / 2;
    int smallnum = magic_at(smallidx) / 2;
    uint32_t sizesmall = (uint32_t)magic_at(smallidx);
    const float inv_precision = 1.0f / fi.precision;
    const uint8_t *start = file + fi.data_off;
    Bits b{start, start + fi.nbytes};
    const uint64_t limit_bits = fi.nbytes * 8;
    int i = 0, run = 0;
    float *o = out;
    while (i < natoms) {
        int cur[3];
        if (bitsize == 0) {
            for (int k = 0; k < 3; ++k) cur[k] = (int)b.get(bitsizeint[k]);
        } else {
            if (!unpack3(b, bitsize, sizeint, cur)) return 7;
        }
        ++i;
        int px = wrap_add(cur[0], minint[0]), py = wrap_add(cur[1], minint[1]), pz = wrap_add(cur[2], minint[2]);
        int is_smaller = 0;
        if (b.get(1)) {
            run = (int)b.get(5);
            is_smaller = run % 3;
            run -= is_smaller;
            --is_smaller;
        }
        if (run > 0) {
            if (i + run / 3 > natoms) return 3;
            // first small atom is written BEFORE the atom it is coded against (water O/H ordering)
            int d[3];
            if (!unpack3_small(b, smallidx, sizesmall, d)) return 7;
            ++i;
            int qx = wrap_add(px, d[0] - smallnum), qy = wrap_add(py, d[1] - smallnum), qz = wrap_add(pz, d[2] - smallnum);
            o[0] = (float)qx * inv_precision; o[1] = (float)qy * inv_precision; o[2] = (float)qz * inv_precision;
            o[3] = (float)px * inv_precision; o[4] = (float)py * inv_precision; o[5] = (float)pz * inv_precision;
            o += 6;
            for (int k = 3; k < run; k += 3) {
                if (!unpack3_small(b, smallidx, sizesmall, d)) return 7;
                ++i;
                qx = wrap_add(qx, d[0] - smallnum); qy = wrap_add(qy, d[1] - smallnum); qz = wrap_add(qz, d[2] - smallnum);
                o[0] = (float)qx * inv_precision; o[1] = (float)qy * inv_precision; o[2] = (float)qz * inv_precision;
                o += 3;
            }
        } else {
            o[0] = (float)px * inv_precision; o[1] = (float)py * inv_precision; o[2] = (float)pz * inv_precision;
            o += 3;
        }
        if (b.consumed_bits() > limit_bits) return 5;
        if (is_smaller) {
            smallidx += is_smaller;
            if (smallidx < FIRSTIDX || smallidx > LASTIDX) return 4;
            if (is_smaller < 0) {
                smallnum = smaller;
                smaller = smallidx > FIRSTIDX ? magic_at(smallidx - 1) / 2 : 0;
            } else {
                smaller = smallnum;
                smallnum = magic_at(smallidx) / 2;
            }
            sizesmall = (uint32_t)magic_at(smallidx);
        }
    }
    return (b.consumed_bits() + 7) / 8 == fi.nbytes ? 0 : 6;      // the block must be consumed exactly
}

// The frame is the parallel axis of XTC (a frame's bit stream is serial): on the host one thread per frame, here ONE LANE
// per frame - 64 frames per wave in lockstep, each lane walking its own stream and writing its own 12 * natoms bytes.  A lane
// is ~25x slower than a host core on one stream (64-bit divisions in software, every load its own cache line), but a batch
// of a thousand frames keeps a thousand of them going: where 8 GPUs want 8 x 2000 frames/s of 250k atoms - 300 host threads
// by the per-thread rate, more than a 256-core node has - the consumers can decode for themselves (~1 % of their
// instruction budget).  Same code as the host path (decode_frame is compiled for both sides), so the same bits.
__global__ void __launch_bounds__(64) xtc_decode_kernel(const uint8_t *__restrict__ blob, const FrameInfo *__restrict__ fi, uint32_t count,
                                                        float *__restrict__ out, size_t natoms, int *__restrict__ status) {
    const uint32_t k = blockIdx.x * 64u + threadIdx.x;
    if (k >= count) return;
    status[k] = decode_frame(blob, fi[k], out + (size_t)k * natoms * 3);
}


// ------------------------------------------------------------------ writer (xtc_handler.rs:117-168: write_state)
// One frame in GROMACS' compressed coordinate format, the algorithm of xdrfile's xdr3dfcoord: coordinates rounded to
// integers at `precision`, every atom a mixed-radix triple over the frame's integer extent, atoms close to their predecessor
// (water molecules) coded as runs of small deltas with an adaptive delta size.  The MolAR side of this is molly::XTCWriter;
// here it writes the synthetic trajectories the XTC-fed benchmarks and tests read back (molar_hip_xtc_encode_frame).
struct BitWriter {
    uint8_t *p;
    size_t cap, n = 0;
    uint64_t acc = 0;
    int have = 0;
    bool overflow = false;
    void put(int nbits, uint32_t v) {          // 0 <= nbits <= 32, MSB first
        if (nbits == 0) return;
        acc = (acc << nbits) | (uint64_t)(nbits == 32 ? v : (v & ((1u << nbits) - 1u)));
        have += nbits;
        while (have >= 8) {
            if (n < cap) p[n++] = (uint8_t)(acc >> (have - 8));
            else overflow = true;
            have -= 8;
        }
    }
    size_t finish() {
        if (have > 0) {
            if (n < cap) p[n++] = (uint8_t)(acc << (8 - have));
            else overflow = true;
            have = 0;
        }
        return n;
    }
};

// the mixed-radix number ((u0 * s1) + u1) * s2 + u2, little-endian by bytes, in `nbits` bits (the inverse of unpack3)
inline void pack3(BitWriter &w, int nbits, const uint32_t s[3], const uint32_t u[3]) {
    unsigned __int128 v = ((unsigned __int128)u[0] * s[1] + u[1]) * s[2] + u[2];
    int left = nbits;
    while (left > 8) {
        w.put(8, (uint32_t)(v & 0xff));
        v >>= 8;
        left -= 8;
    }
    if (left > 0) w.put(left, (uint32_t)(v & 0xff));
}

inline void put_be32(uint8_t *p, uint32_t v) {
    p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v;
}
inline void put_bef(uint8_t *p, float f) { put_be32(p, __builtin_bit_cast(uint32_t, f)); }

// returns the frame's length, 0 when `cap` is too small, (size_t)-1 when a coordinate does not fit the format
size_t encode_frame(const float *xyz, size_t natoms, const float box9[9], int32_t step, float time, float precision, uint8_t *out, size_t cap) {
    if (cap < 56) return 0;
    put_be32(out, 1995u);
    put_be32(out + 4, (uint32_t)natoms);
    put_be32(out + 8, (uint32_t)step);
    put_bef(out + 12, time);
    for (int k = 0; k < 9; ++k) put_bef(out + 16 + 4 * k, box9[k]);
    put_be32(out + 52, (uint32_t)natoms);
    if (natoms <= 9) {                       // small systems are stored as plain floats
        if (cap < 56 + natoms * 12) return 0;
        for (size_t k = 0; k < 3 * natoms; ++k) put_bef(out + 56 + 4 * k, xyz[k]);
        return 56 + natoms * 12;
    }
    if (cap < 92) return 0;
    std::vector<int> ip(3 * natoms);
    int minint[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, maxint[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
    int mindiff = INT32_MAX, old[3] = {0, 0, 0};
    for (size_t i = 0; i < natoms; ++i) {
        int cur[3];
        for (int c = 0; c < 3; ++c) {
            const float lf = xyz[3 * i + c] * precision;
            if (!(std::fabs(lf) < 2147483000.0f)) return (size_t)-1;        // scaling would overflow the integer grid
            cur[c] = (int)(lf >= 0.0f ? lf + 0.5f : lf - 0.5f);
            ip[3 * i + c] = cur[c];
            if (cur[c] < minint[c]) minint[c] = cur[c];
            if (cur[c] > maxint[c]) maxint[c] = cur[c];
        }
        const long long diff = std::llabs((long long)old[0] - cur[0]) + std::llabs((long long)old[1] - cur[1]) + std::llabs((long long)old[2] - cur[2]);
        if (i > 0 && diff < mindiff) mindiff = (int)diff;
        old[0] = cur[0]; old[1] = cur[1]; old[2] = cur[2];
    }
    uint32_t sizeint[3];
    int bitsizeint[3] = {0, 0, 0}, bitsize;
    for (int c = 0; c < 3; ++c) {
        if ((long long)maxint[c] - (long long)minint[c] >= 0x7FFFFFFELL) return (size_t)-1;
        sizeint[c] = (uint32_t)(maxint[c] - minint[c]) + 1u;
    }
    if ((sizeint[0] | sizeint[1] | sizeint[2]) > 0xffffffu) {
        for (int c = 0; c < 3; ++c) bitsizeint[c] = bits_for(sizeint[c]);
        bitsize = 0;
    } else {
        bitsize = bits_for_product(sizeint);
    }
    int smallidx = FIRSTIDX;
    while (smallidx < LASTIDX && MAGIC[smallidx] < mindiff) ++smallidx;
    const int maxidx = LASTIDX < smallidx + 8 ? LASTIDX : smallidx + 8, minidx = maxidx - 8;
    int smaller = MAGIC[smallidx - 1 > FIRSTIDX ? smallidx - 1 : FIRSTIDX] / 2;
    int smallnum = MAGIC[smallidx] / 2;
    uint32_t sizesmall[3] = {(uint32_t)MAGIC[smallidx], (uint32_t)MAGIC[smallidx], (uint32_t)MAGIC[smallidx]};
    const int larger = MAGIC[maxidx] / 2;
    put_bef(out + 56, precision);
    for (int c = 0; c < 3; ++c) {
        put_be32(out + 60 + 4 * c, (uint32_t)minint[c]);
        put_be32(out + 72 + 4 * c, (uint32_t)maxint[c]);
    }
    put_be32(out + 84, (uint32_t)smallidx);
    const size_t hdr = 92;
    BitWriter w{out + hdr, cap - hdr};
    auto close = [](const int *a, const int *b, int lim) {
        return std::abs(a[0] - b[0]) < lim && std::abs(a[1] - b[1]) < lim && std::abs(a[2] - b[2]) < lim;
    };
    size_t i = 0;
    int prevrun = -1;
    const int *prev = nullptr;
    uint32_t tmp[30];
    while (i < natoms) {
        int *cur = &ip[3 * i];
        int is_smaller;
        if (smallidx < maxidx && i >= 1 && close(cur, prev, larger)) is_smaller = 1;
        else if (smallidx > minidx) is_smaller = -1;
        else is_smaller = 0;
        bool is_small = false;
        if (i + 1 < natoms && close(cur, cur + 3, smallnum)) {
            // the first two atoms change places: the second is written in full, the first as a delta in front of it
            for (int c = 0; c < 3; ++c) std::swap(cur[c], cur[3 + c]);
            is_small = true;
        }
        uint32_t u[3];
        for (int c = 0; c < 3; ++c) u[c] = (uint32_t)(cur[c] - minint[c]);
        if (bitsize == 0) for (int c = 0; c < 3; ++c) w.put(bitsizeint[c], u[c]);
        else pack3(w, bitsize, sizeint, u);
        prev = cur;
        cur += 3;
        ++i;
        int run = 0;
        if (!is_small && is_smaller == -1) is_smaller = 0;
        while (is_small && run < 8 * 3) {
            if (is_smaller == -1) {
                const long long dx = cur[0] - prev[0], dy = cur[1] - prev[1], dz = cur[2] - prev[2];
                if (dx * dx + dy * dy + dz * dz >= (long long)smaller * smaller) is_smaller = 0;
            }
            for (int c = 0; c < 3; ++c) tmp[run++] = (uint32_t)(cur[c] - prev[c] + smallnum);
            prev = cur;
            cur += 3;
            ++i;
            is_small = i < natoms && close(cur, prev, smallnum);
        }
        if (run != prevrun || is_smaller != 0) {
            prevrun = run;
            w.put(1, 1u);
            w.put(5, (uint32_t)(run + is_smaller + 1));
        } else {
            w.put(1, 0u);
        }
        for (int k = 0; k < run; k += 3) pack3(w, smallidx, sizesmall, &tmp[k]);
        if (is_smaller != 0) {
            smallidx += is_smaller;
            if (is_smaller < 0) {
                smallnum = smaller;
                smaller = smallidx > FIRSTIDX ? MAGIC[smallidx - 1] / 2 : 0;
            } else {
                smaller = smallnum;
                smallnum = MAGIC[smallidx] / 2;
            }
            sizesmall[0] = sizesmall[1] = sizesmall[2] = (uint32_t)MAGIC[smallidx];
        }
        if (w.overflow) return 0;
    }
    size_t nbytes = w.finish();
    if (w.overflow || nbytes > 0xFFFFFFFFull) return 0;
    put_be32(out + 88, (uint32_t)nbytes);
    while (nbytes & 3u) {
        if (hdr + nbytes >= cap) return 0;
        out[hdr + nbytes++] = 0;
    }
    return hdr + nbytes;
}

}  // namespace

struct molar_hip_xtc {
    const uint8_t *data = nullptr;
    size_t size = 0;
    bool mapped = false;
    std::vector<FrameInfo> frames;
};

namespace {

int build_index(molar_hip_xtc *x) {
    size_t off = 0;
    while (off + 56 <= x->size) {
        const uint8_t *p = x->data + off;
        const uint32_t magic = be32(p);
        if (magic != 1995 && magic != 2023) break;
        FrameInfo f{};
        f.offset = off;
        f.natoms = (int32_t)be32(p + 4);
        if (f.natoms < 0 || (int32_t)be32(p + 52) != f.natoms) break;
        f.step = (int32_t)be32(p + 8);
        f.time = bef(p + 12);
        for (int k = 0; k < 9; ++k) f.box[k] = bef(p + 16 + 4 * k);
        size_t len;
        if (f.natoms <= 9) {
            f.precision = 0.f;
            f.data_off = off + 56;
            f.nbytes = (uint64_t)f.natoms * 12;
            len = 56 + f.nbytes;
        } else {
            const size_t hdr = 56 + 32 + (magic == 2023 ? 8 : 4);
            if (off + hdr > x->size) break;
            f.precision = bef(p + 56);
            f.nbytes = magic == 2023 ? (((uint64_t)be32(p + 88) << 32) | be32(p + 92)) : be32(p + 88);
            f.data_off = off + hdr;
            len = hdr + ((f.nbytes + 3) & ~(uint64_t)3);
            if (f.nbytes * 8 < (uint64_t)f.natoms) break;   // every atom costs at least its flag bit: hostile header
        }
        if (off + len > x->size) break;       // truncated last frame: stop like an UnexpectedEof (xtc_handler.rs:325-332)
        x->frames.push_back(f);
        off += len;
    }
    return 0;
}

}  // namespace

extern "C" {

molar_hip_xtc *molar_hip_xtc_open_memory(const void *data, size_t bytes) {
    if (!data) {
        fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "xtc_open_memory: null data");
        return nullptr;
    }
    auto *x = new molar_hip_xtc;
    x->data = (const uint8_t *)data;
    x->size = bytes;
    build_index(x);
    return x;
}

molar_hip_xtc *molar_hip_xtc_open(const char *path) {
    if (!path) {
        fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "xtc_open: null path");
        return nullptr;
    }
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) {
        fail(MOLAR_HIP_ERR_IO, "xtc_open: cannot open '%s'", path);
        return nullptr;
    }
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size <= 0) {
        ::close(fd);
        fail(MOLAR_HIP_ERR_IO, "xtc_open: cannot stat '%s' or file is empty", path);
        return nullptr;
    }
    void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (m == MAP_FAILED) {
        fail(MOLAR_HIP_ERR_IO, "xtc_open: mmap of '%s' failed", path);
        return nullptr;
    }
    auto *x = new molar_hip_xtc;
    x->data = (const uint8_t *)m;
    x->size = (size_t)st.st_size;
    x->mapped = true;
    build_index(x);
    return x;
}

void molar_hip_xtc_close(molar_hip_xtc *x) {
    if (!x) return;
    if (x->mapped) munmap(const_cast<uint8_t *>(x->data), x->size);
    delete x;
}

size_t molar_hip_xtc_nframes(const molar_hip_xtc *x) { return x ? x->frames.size() : 0; }
size_t molar_hip_xtc_natoms(const molar_hip_xtc *x) { return x && !x->frames.empty() ? (size_t)x->frames[0].natoms : 0; }

int molar_hip_xtc_frame_info(const molar_hip_xtc *x, size_t frame, int32_t *natoms, int32_t *step, float *time,
                             float box9[9], float *precision) {
    if (!x || frame >= x->frames.size()) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "xtc_frame_info: frame %zu out of range", frame);
    const FrameInfo &f = x->frames[frame];
    if (natoms) *natoms = f.natoms;
    if (step) *step = f.step;
    if (time) *time = f.time;
    if (precision) *precision = f.precision;
    if (box9) std::memcpy(box9, f.box, sizeof f.box);
    return MOLAR_HIP_OK;
}

int molar_hip_xtc_seek_time(const molar_hip_xtc *x, float t, size_t *frame) {
    if (!x || !frame) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "xtc_seek_time: null argument");
    for (size_t k = 0; k < x->frames.size(); ++k)       // skip_to_time: first frame with time >= t (xtc_handler.rs:282-297)
        if (x->frames[k].time >= t) {
            *frame = k;
            return MOLAR_HIP_OK;
        }
    return fail(MOLAR_HIP_ERR_IO, "xtc_seek_time: no frame at or after t = %g", (double)t);
}

int molar_hip_xtc_encode_frame(const float *xyz, size_t natoms, const float *box9, int32_t step, float time, float precision,
                               uint8_t *out, size_t cap, size_t *out_len) {
    if ((!xyz && natoms) || !box9 || !out || !out_len) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "xtc_encode_frame: null argument");
    if (natoms > 0x7FFFFFFFull) return fail(MOLAR_HIP_ERR_TOO_LARGE, "xtc_encode_frame: too many atoms");
    if (!(precision > 0.0f)) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "xtc_encode_frame: precision must be positive (got %g)", (double)precision);
    const size_t len = encode_frame(xyz, natoms, box9, step, time, precision, out, cap);
    if (len == (size_t)-1) return fail(MOLAR_HIP_ERR_IO, "xtc_encode_frame: a coordinate times the precision does not fit the integer grid");
    if (len == 0) return fail(MOLAR_HIP_ERR_TOO_LARGE, "xtc_encode_frame: the buffer of %zu bytes is too small (96 + 16 * natoms always suffices)", cap);
    *out_len = len;
    return MOLAR_HIP_OK;
}

int molar_hip_xtc_read(molar_hip_ctx *c, const molar_hip_xtc *x, size_t first, size_t count, float *xyz, int nthreads) {
    if (!x || !xyz) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "xtc_read: null argument");
    if (count == 0) return MOLAR_HIP_OK;
    if (first + count > x->frames.size()) return fail(MOLAR_HIP_ERR_IO, "xtc_read: frames %zu..%zu past the end (%zu frames)", first, first + count, x->frames.size());
    const size_t natoms = (size_t)x->frames[first].natoms;
    for (size_t k = first; k < first + count; ++k)
        if ((size_t)x->frames[k].natoms != natoms) return fail(MOLAR_HIP_ERR_SIZES, "xtc_read: frame %zu has %d atoms, frame %zu has %zu", k, x->frames[k].natoms, first, natoms);
    const bool dev = is_device_ptr(xyz);
    if (dev && !c) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "xtc_read: device destination needs a context");
    float *host = xyz;
    const size_t fbytes = natoms * 12;
    if (dev) {
        MH_HIP(hipSetDevice(c->device));
        MH_TRY(ensure_pinned(c, fbytes * count));
        host = (float *)c->h_pinned;
    }
    int T = nthreads > 0 ? nthreads : (int)std::thread::hardware_concurrency();
    if (T < 1) T = 1;
    if ((size_t)T > count) T = (int)count;
    std::atomic<size_t> next{0};
    std::atomic<int> err{0};
    std::vector<std::atomic<uint8_t>> done(count);
    for (auto &d : done) d.store(0);
    auto work = [&]() {
        for (;;) {
            const size_t k = next.fetch_add(1);
            if (k >= count) return;
            const int rc = decode_frame(x->data, x->frames[first + k], host + k * natoms * 3);
            if (rc) err.store(rc);
            done[k].store(1, std::memory_order_release);
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < T; ++t) pool.emplace_back(work);
    if (dev) {
        // this thread ships frames to the GPU in order as they complete, the others decode
        if (T == 1) work();
        for (size_t k = 0; k < count; ++k) {
            while (!done[k].load(std::memory_order_acquire)) {
                if (T == 1) break;
                std::this_thread::yield();
            }
            (void)hipMemcpyAsync(xyz + k * natoms * 3, host + k * natoms * 3, fbytes, hipMemcpyHostToDevice, c->stream);
        }
    } else {
        work();
    }
    for (auto &t : pool) t.join();
    if (dev) MH_HIP(hipStreamSynchronize(c->stream));
    if (err.load()) return fail(MOLAR_HIP_ERR_IO, "xtc_read: corrupt compressed block (code %d)", err.load());
    return MOLAR_HIP_OK;
}

// BASELINE config 4 as ONE call for a compiled caller (the Rust AnalysisTask of an RDF has no device memory of its own): frames
// [first, first + count) of the trajectory are decoded on host threads into windows of 16 frames in HBM - by a helper thread on
// the context's auxiliary context, while the window before is in its histogram kernels - and every window goes through
// molar_hip_search_histogram_frames (one selection, every frame's own box from its header) into device-resident bins; at the end
// the bins are ADDED into the caller's host array (Histogram1D::add_one over the whole block, molar_membrane/src/stats.rs:29-35).
// (`two`: distance_search_double_pbc between the selections idx / idx2 of every frame; a NULL index = all atoms)
static int xtc_histogram_impl(molar_hip_ctx *c, const molar_hip_xtc *x, size_t first, size_t count, const uint64_t *idx, size_t n, bool two,
                              const uint64_t *idx2, size_t n2, float cutoff, uint8_t pbc, float hmin, float hmax, size_t nbins, uint64_t *bins,
                              int decode_threads) {
    if (!c || !x || !bins) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "xtc_histogram: null argument");
    if (nbins == 0 || nbins > 8192) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "search_histogram: nbins must be in 1..8192");
    if (count == 0) return MOLAR_HIP_OK;
    if (first + count > x->frames.size()) return fail(MOLAR_HIP_ERR_IO, "xtc_histogram: frames %zu..%zu past the end (%zu frames)", first, first + count, x->frames.size());
    if (is_device_ptr(bins) || is_device_ptr(idx) || is_device_ptr(idx2)) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "xtc_histogram: index and bins are host arrays");
    const size_t natoms = (size_t)x->frames[first].natoms;
    for (size_t k = first; k < first + count; ++k)
        if ((size_t)x->frames[k].natoms != natoms) return fail(MOLAR_HIP_ERR_SIZES, "xtc_histogram: frame %zu has %d atoms, frame %zu has %zu", k, x->frames[k].natoms, first, natoms);
    for (size_t k = 0; idx && k < n; ++k)
        if (idx[k] >= natoms) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "xtc_histogram: index %llu out of range", (unsigned long long)idx[k]);
    for (size_t k = 0; idx2 && k < n2; ++k)
        if (idx2[k] >= natoms) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "xtc_histogram: index %llu out of range", (unsigned long long)idx2[k]);
    MH_HIP(hipSetDevice(c->device));
    if (!c->aux) {
        c->aux = molar_hip_create(c->device);
        if (!c->aux) return MOLAR_HIP_ERR_HIP;
    }
    constexpr size_t WIN = 16;
    const size_t fbytes = natoms * 12;
    MH_TRY(c->xh_win[0].reserve(WIN * fbytes));
    MH_TRY(c->xh_win[1].reserve(WIN * fbytes));
    MH_TRY(c->xh_bins.reserve(nbins * 8 + (idx ? n * 8 : 0) + (idx2 ? n2 * 8 : 0)));
    unsigned long long *dbins = c->xh_bins.as<unsigned long long>();
    const uint64_t *didx = nullptr, *didx2 = nullptr;
    MH_HIP(hipMemsetAsync(dbins, 0, nbins * 8, c->stream));
    if (idx) {
        didx = reinterpret_cast<const uint64_t *>(dbins + nbins);
        MH_HIP(hipMemcpyAsync(const_cast<uint64_t *>(didx), idx, n * 8, hipMemcpyHostToDevice, c->stream));
    }
    if (idx2) {
        didx2 = reinterpret_cast<const uint64_t *>(dbins + nbins) + (idx ? n : 0);
        MH_HIP(hipMemcpyAsync(const_cast<uint64_t *>(didx2), idx2, n2 * 8, hipMemcpyHostToDevice, c->stream));
    }
    MH_HIP(hipStreamSynchronize(c->stream));
    const size_t nwin = (count + WIN - 1) / WIN;
    std::vector<float> boxes(WIN * 9);
    int dec_rc = 0;
    std::string dec_err;
    auto decode = [&](size_t w) {        // on the helper thread: the auxiliary context's stream and staging
        const size_t f0 = first + w * WIN, k = std::min(WIN, first + count - f0);
        dec_rc = molar_hip_xtc_read(c->aux, x, f0, k, c->xh_win[w & 1].as<float>(), decode_threads);
        if (dec_rc) dec_err = molar_hip_last_error();
    };
    std::thread helper(decode, (size_t)0);
    int rc = 0;
    for (size_t w = 0; w < nwin && !rc; ++w) {
        helper.join();
        if (dec_rc) { rc = fail(dec_rc, "%s", dec_err.c_str()); break; }
        // the window before this one has left its buffer free for the next decode only once its kernels are through
        if (w >= 1) rc = molar_hip_synchronize(c);
        if (!rc && w + 1 < nwin) helper = std::thread(decode, w + 1);
        if (rc) break;
        const size_t f0 = first + w * WIN, k = std::min(WIN, first + count - f0);
        for (size_t f = 0; f < k; ++f) std::memcpy(&boxes[9 * f], x->frames[f0 + f].box, 36);
        molar_hip_search_desc q{};
        q.kind = two ? MOLAR_HIP_SEARCH_DOUBLE : MOLAR_HIP_SEARCH_SINGLE;
        q.cutoff = cutoff;
        q.xyz1 = c->xh_win[w & 1].as<float>();
        q.natoms1 = natoms;
        q.idx1 = didx;
        q.n1 = idx ? n : 0;
        if (two) {
            q.xyz2 = q.xyz1;
            q.natoms2 = natoms;
            q.idx2 = didx2;
            q.n2 = idx2 ? n2 : 0;
        }
        q.box9 = boxes.data();
        q.pbc = pbc;
        rc = molar_hip_search_histogram_frames(c, &q, k, natoms * 3, natoms * 3, boxes.data(), hmin, hmax, nbins, reinterpret_cast<uint64_t *>(dbins));
    }
    if (helper.joinable()) helper.join();
    if (rc) {
        (void)molar_hip_synchronize(c);
        return rc;
    }
    std::vector<unsigned long long> h(nbins);
    MH_HIP(hipMemcpyAsync(h.data(), dbins, nbins * 8, hipMemcpyDeviceToHost, c->stream));
    MH_HIP(hipStreamSynchronize(c->stream));
    for (size_t b = 0; b < nbins; ++b) bins[b] += h[b];
    return MOLAR_HIP_OK;
}

int molar_hip_xtc_histogram(molar_hip_ctx *c, const molar_hip_xtc *x, size_t first, size_t count, const uint64_t *idx, size_t n, float cutoff,
                            uint8_t pbc, float hmin, float hmax, size_t nbins, uint64_t *bins, int decode_threads) {
    return xtc_histogram_impl(c, x, first, count, idx, n, false, nullptr, 0, cutoff, pbc, hmin, hmax, nbins, bins, decode_threads);
}

int molar_hip_xtc_histogram_double(molar_hip_ctx *c, const molar_hip_xtc *x, size_t first, size_t count, const uint64_t *idx1, size_t n1,
                                   const uint64_t *idx2, size_t n2, float cutoff, uint8_t pbc, float hmin, float hmax, size_t nbins, uint64_t *bins,
                                   int decode_threads) {
    return xtc_histogram_impl(c, x, first, count, idx1, n1, true, idx2, n2, cutoff, pbc, hmin, hmax, nbins, bins, decode_threads);
}

int molar_hip_xtc_read_device(molar_hip_ctx *c, const molar_hip_xtc *x, size_t first, size_t count, float *xyz_dev) {
    if (!c || !x || !xyz_dev) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "xtc_read_device: null argument");
    if (count == 0) return MOLAR_HIP_OK;
    if (!is_device_ptr(xyz_dev)) return fail(MOLAR_HIP_ERR_INVALID_ARGUMENT, "xtc_read_device: the destination must be device memory");
    if (first + count > x->frames.size()) return fail(MOLAR_HIP_ERR_IO, "xtc_read_device: frames %zu..%zu past the end (%zu frames)", first, first + count, x->frames.size());
    if (count > 0xFFFFFFFFull) return fail(MOLAR_HIP_ERR_TOO_LARGE, "xtc_read_device: too many frames in one call");
    const size_t natoms = (size_t)x->frames[first].natoms;
    for (size_t k = first; k < first + count; ++k)
        if ((size_t)x->frames[k].natoms != natoms) return fail(MOLAR_HIP_ERR_SIZES, "xtc_read_device: frame %zu has %d atoms, frame %zu has %zu", k, x->frames[k].natoms, first, natoms);
    MH_HIP(hipSetDevice(c->device));
    // the window's bytes [first frame's header, end of the last frame's block) and its frame table, offsets relative to the window
    const uint64_t off0 = x->frames[first].offset;
    const FrameInfo &last = x->frames[first + count - 1];
    const uint64_t end = last.natoms <= 9 ? last.data_off + last.nbytes : last.data_off + ((last.nbytes + 3) & ~(uint64_t)3);
    std::vector<FrameInfo> tab(x->frames.begin() + first, x->frames.begin() + first + count);
    for (auto &f : tab) { f.offset -= off0; f.data_off -= off0; }
    const size_t tab_bytes = count * sizeof(FrameInfo), st_bytes = count * sizeof(int);
    MH_TRY(c->m_xyz2.reserve((size_t)(end - off0) + 8));
    MH_TRY(c->m_idx2.reserve(tab_bytes + st_bytes));
    // From here on asynchronous copies read `tab` and write `status` (pageable, local): every way out waits for the stream first.
    std::vector<int> status(count);
    int *d_status = reinterpret_cast<int *>(c->m_idx2.as<char>() + tab_bytes);
    hipError_t e = hipMemcpyAsync(c->m_xyz2.p, x->data + off0, (size_t)(end - off0), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(c->m_idx2.p, tab.data(), tab_bytes, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(xtc_decode_kernel, dim3((unsigned)((count + 63) / 64)), dim3(64), 0, c->stream, c->m_xyz2.as<uint8_t>(),
                           c->m_idx2.as<FrameInfo>(), (uint32_t)count, xyz_dev, natoms, d_status);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(status.data(), d_status, st_bytes, hipMemcpyDeviceToHost, c->stream);
    if (e != hipSuccess) {
        (void)hipStreamSynchronize(c->stream);
        return fail(MOLAR_HIP_ERR_HIP, "xtc_read_device: %s", hipGetErrorString(e));
    }
    MH_HIP(hipStreamSynchronize(c->stream));
    std::vector<float> tmp;
    for (size_t k = 0; k < count; ++k) {
        if (status[k] == 0) continue;
        if (status[k] != 7) return fail(MOLAR_HIP_ERR_IO, "xtc_read_device: corrupt compressed block in frame %zu (code %d)", first + k, status[k]);
        tmp.resize(natoms * 3);                   // triples wider than 64 bits: the host decoder takes this frame
        const int rc = decode_frame(x->data, x->frames[first + k], tmp.data());
        if (rc) return fail(MOLAR_HIP_ERR_IO, "xtc_read_device: corrupt compressed block in frame %zu (code %d)", first + k, rc);
        MH_HIP(hipMemcpy(xyz_dev + k * natoms * 3, tmp.data(), natoms * 12, hipMemcpyHostToDevice));
    }
    return MOLAR_HIP_OK;
}

}  // extern "C"
