/*
 * xtc_oracle.c — CPU restatement of the XTC frame codec that feeds MolAR's per-frame path.
 *
 * TEST INFRASTRUCTURE ONLY (see molar_oracle.h).
 *
 * MolAR reads XTC through the third-party crate `molly` (molar/Cargo.toml:36, git, >= 0.6.1;
 * call sites molar/src/io/xtc_handler.rs:64-112, 200-300), which is NOT vendored under
 * /root/reference.  molly implements the GROMACS XTC format; this file restates the published
 * algorithm of that format (xdrfile 1.1.4, xdr3dfcoord / xdrfile_decompress_coord_float):
 * big-endian XDR header, mixed-radix packed integers, adaptive "small delta" runs with the
 * water-pair swap.  Magic 1995 (32-bit byte count) and 2023 (64-bit byte count, as molly's
 * read_nbytes(file, magic)) are accepted.
 *
 * PINNING: the reference's own asserting test tests/test_netcdf.rs:37-80 (benzene.xtc decodes to
 * benzene.nc within 1e-3 nm, time within 0.01) is replayed in tests/test_oracle_xtc.py from copies
 * of those two data files; structural checks (every frame header of the reference's new.xtc /
 * traj_comp.xtc lands on the magic, every compressed block is consumed exactly) run when
 * /root/reference is present.  Float conversion (coordinate * (1/precision)) follows xdrfile;
 * molly may differ by an ulp — bit-level parity of the floats is unpinned.
 *
 * The encoder at the bottom writes valid streams (fixed small-index, runs with the pair swap) so
 * that tests can make large inputs; it is not the GROMACS compressor.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define FIRSTIDX 9
static const int magicints[] = {
    0,       0,       0,       0,       0,       0,       0,       0,       0,       8,        10,       12,       16,
    20,      25,      32,      40,      50,      64,      80,      101,     128,     161,      203,      256,      322,
    406,     512,     645,     812,     1024,    1290,    1625,    2048,    2580,    3250,     4096,     5060,     6501,
    8192,    10321,   13003,   16384,   20642,   26007,   32768,   41285,   52015,   65536,    82570,    104031,   131072,
    165140,  208063,  262144,  330280,  416127,  524287,  660561,  832255,  1048576, 1321122,  1664510,  2097152,  2642245,
    3329021, 4194304, 5284491, 6658042, 8388607, 10568983, 13316085, 16777216};
#define LASTIDX ((int)(sizeof(magicints) / sizeof(*magicints)) - 1)

static uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
static float bef(const uint8_t *p) {
    uint32_t u = be32(p);
    float f;
    memcpy(&f, &u, 4);
    return f;
}

static int sizeofint(unsigned size) {
    unsigned num = 1;
    int nbits = 0;
    while (size >= num && nbits < 32) {
        nbits++;
        num <<= 1;
    }
    return nbits;
}

static int sizeofints(int n, const unsigned sizes[]) {
    unsigned bytes[32], nbytes = 1, bytecnt, tmp, num;
    int nbits = 0;
    bytes[0] = 1;
    for (int i = 0; i < n; i++) {
        tmp = 0;
        for (bytecnt = 0; bytecnt < nbytes; bytecnt++) {
            tmp = bytes[bytecnt] * sizes[i] + tmp;
            bytes[bytecnt] = tmp & 0xff;
            tmp >>= 8;
        }
        while (tmp != 0) {
            bytes[bytecnt++] = tmp & 0xff;
            tmp >>= 8;
        }
        nbytes = bytecnt;
    }
    num = 1;
    nbytes--;
    while (bytes[nbytes] >= num) {
        nbits++;
        num *= 2;
    }
    return nbits + (int)nbytes * 8;
}

typedef struct {
    const uint8_t *p;
    size_t n, cnt;
    unsigned lastbits, lastbyte;
    int overrun;
} bitrd;

static unsigned rd_byte(bitrd *b) {
    if (b->cnt >= b->n) {
        b->overrun = 1;
        return 0;
    }
    return b->p[b->cnt++];
}

static unsigned receivebits(bitrd *b, int nbits) {
    const unsigned mask = nbits >= 32 ? 0xffffffffu : ((1u << nbits) - 1u);
    unsigned num = 0;
    while (nbits >= 8) {
        b->lastbyte = (b->lastbyte << 8) | rd_byte(b);
        num |= (b->lastbyte >> b->lastbits) << (nbits - 8);
        nbits -= 8;
    }
    if (nbits > 0) {
        if (b->lastbits < (unsigned)nbits) {
            b->lastbits += 8;
            b->lastbyte = (b->lastbyte << 8) | rd_byte(b);
        }
        b->lastbits -= nbits;
        num |= (b->lastbyte >> b->lastbits) & ((1u << nbits) - 1u);
    }
    return num & mask;
}

static void receiveints(bitrd *b, int nints, int nbits, const unsigned sizes[], int nums[]) {
    int bytes[32], nbytes = 0;
    bytes[1] = bytes[2] = bytes[3] = 0;
    while (nbits > 8) {
        bytes[nbytes++] = (int)receivebits(b, 8);
        nbits -= 8;
    }
    if (nbits > 0) bytes[nbytes++] = (int)receivebits(b, nbits);
    for (int i = nints - 1; i > 0; i--) {
        unsigned num = 0;
        for (int j = nbytes - 1; j >= 0; j--) {
            num = (num << 8) | (unsigned)bytes[j];
            const unsigned p = num / sizes[i];
            bytes[j] = (int)p;
            num = num - p * sizes[i];
        }
        nums[i] = (int)num;
    }
    nums[0] = bytes[0] | (bytes[1] << 8) | (bytes[2] << 16) | (bytes[3] << 24);
}

/* Header of the frame at `p` (at least 56 readable bytes).  Returns total frame length in bytes, 0 on error. */
size_t orc_xtc_frame_header(const uint8_t *p, size_t avail, int32_t *natoms, int32_t *step, float *time, float box9[9],
                            float *precision) {
    if (avail < 56) return 0;
    const uint32_t magic = be32(p);
    if (magic != 1995 && magic != 2023) return 0;
    const int32_t n = (int32_t)be32(p + 4);
    if (n < 0 || (int32_t)be32(p + 52) != n) return 0;
    if (natoms) *natoms = n;
    if (step) *step = (int32_t)be32(p + 8);
    if (time) *time = bef(p + 12);
    if (box9) for (int k = 0; k < 9; ++k) box9[k] = bef(p + 16 + 4 * k);
    if (n <= 9) {
        if (precision) *precision = 0.f;
        return avail >= 56 + (size_t)n * 12 ? 56 + (size_t)n * 12 : 0;
    }
    if (avail < 56 + 32 + 4) return 0;
    if (precision) *precision = bef(p + 56);
    size_t hdr = 56 + 32, nbytes;
    if (magic == 2023) {
        if (avail < hdr + 8) return 0;
        nbytes = ((size_t)be32(p + hdr) << 32) | be32(p + hdr + 4);
        hdr += 8;
    } else {
        nbytes = be32(p + hdr);
        hdr += 4;
    }
    const size_t total = hdr + ((nbytes + 3) & ~(size_t)3);
    return total <= avail ? total : 0;
}

/* Decodes the frame at `p` into xyz[3*natoms] (nm).  Returns 0 on success. */
int orc_xtc_decode_frame(const uint8_t *p, size_t avail, float *xyz) {
    int32_t natoms;
    float precision;
    const size_t total = orc_xtc_frame_header(p, avail, &natoms, NULL, NULL, NULL, &precision);
    if (!total) return 1;
    if (natoms <= 9) {
        for (int k = 0; k < 3 * natoms; ++k) xyz[k] = bef(p + 56 + 4 * k);
        return 0;
    }
    const uint32_t magic = be32(p);
    int minint[3], maxint[3];
    for (int k = 0; k < 3; ++k) {
        minint[k] = (int32_t)be32(p + 60 + 4 * k);
        maxint[k] = (int32_t)be32(p + 72 + 4 * k);
    }
    int smallidx = (int32_t)be32(p + 84);
    if (smallidx < FIRSTIDX || smallidx > LASTIDX) return 2;
    const size_t hdr = 56 + 32 + (magic == 2023 ? 8 : 4);
    const size_t nbytes = magic == 2023 ? (((size_t)be32(p + 88) << 32) | be32(p + 92)) : be32(p + 88);
    unsigned sizeint[3], bitsizeint[3] = {0, 0, 0};
    int bitsize;
    for (int k = 0; k < 3; ++k) sizeint[k] = (unsigned)(maxint[k] - minint[k] + 1);
    if ((sizeint[0] | sizeint[1] | sizeint[2]) > 0xffffff) {
        for (int k = 0; k < 3; ++k) bitsizeint[k] = (unsigned)sizeofint(sizeint[k]);
        bitsize = 0;
    } else {
        bitsize = sizeofints(3, sizeint);
    }
    int tmpidx = smallidx - 1;
    if (tmpidx < FIRSTIDX) tmpidx = FIRSTIDX;
    int smaller = magicints[tmpidx] / 2;
    int smallnum = magicints[smallidx] / 2;
    unsigned sizesmall[3] = {(unsigned)magicints[smallidx], (unsigned)magicints[smallidx], (unsigned)magicints[smallidx]};
    const float inv_precision = 1.0f / precision;
    bitrd b = {p + hdr, nbytes, 0, 0, 0, 0};
    int i = 0, run = 0;
    float *out = xyz;
    while (i < natoms) {
        int thiscoord[3], prevcoord[3];
        if (bitsize == 0) {
            for (int k = 0; k < 3; ++k) thiscoord[k] = (int)receivebits(&b, (int)bitsizeint[k]);
        } else {
            receiveints(&b, 3, bitsize, sizeint, thiscoord);
        }
        i++;
        for (int k = 0; k < 3; ++k) {
            thiscoord[k] += minint[k];
            prevcoord[k] = thiscoord[k];
        }
        const int flag = (int)receivebits(&b, 1);
        int is_smaller = 0;
        if (flag == 1) {
            run = (int)receivebits(&b, 5);
            is_smaller = run % 3;
            run -= is_smaller;
            is_smaller--;
        }
        if (run > 0) {
            if (i + run / 3 > natoms) return 3;
            for (int k = 0; k < run; k += 3) {
                receiveints(&b, 3, smallidx, sizesmall, thiscoord);
                i++;
                for (int c = 0; c < 3; ++c) thiscoord[c] += prevcoord[c] - smallnum;
                if (k == 0) {
                    for (int c = 0; c < 3; ++c) {   /* the pair is stored swapped (water O/H trick) */
                        const int t = thiscoord[c];
                        thiscoord[c] = prevcoord[c];
                        prevcoord[c] = t;
                    }
                    for (int c = 0; c < 3; ++c) *out++ = (float)prevcoord[c] * inv_precision;
                } else {
                    for (int c = 0; c < 3; ++c) prevcoord[c] = thiscoord[c];
                }
                for (int c = 0; c < 3; ++c) *out++ = (float)thiscoord[c] * inv_precision;
            }
        } else {
            for (int c = 0; c < 3; ++c) *out++ = (float)thiscoord[c] * inv_precision;
        }
        smallidx += is_smaller;
        if (smallidx < FIRSTIDX || smallidx > LASTIDX) return 4;
        if (is_smaller < 0) {
            smallnum = smaller;
            smaller = smallidx > FIRSTIDX ? magicints[smallidx - 1] / 2 : 0;
        } else if (is_smaller > 0) {
            smaller = smallnum;
            smallnum = magicints[smallidx] / 2;
        }
        sizesmall[0] = sizesmall[1] = sizesmall[2] = (unsigned)magicints[smallidx];
        if (b.overrun) return 5;
    }
    /* the block must be consumed exactly (last partial byte allowed) */
    return b.cnt == nbytes ? 0 : 6;
}

/* Number of frames and (optionally) their byte offsets; stops at the first invalid header. */
size_t orc_xtc_index(const uint8_t *data, size_t size, uint64_t *offsets, size_t cap) {
    size_t off = 0, n = 0;
    while (off < size) {
        const size_t len = orc_xtc_frame_header(data + off, size - off, NULL, NULL, NULL, NULL, NULL);
        if (!len) break;
        if (offsets && n < cap) offsets[n] = off;
        n++;
        off += len;
    }
    return n;
}

/* ------------------------------------------------------------------ test-data encoder */

typedef struct {
    uint8_t *p;
    size_t cap, cnt;
    unsigned lastbits, lastbyte;
} bitwr;

static void sendbits(bitwr *w, int nbits, unsigned num) {
    unsigned lastbyte = w->lastbyte, lastbits = w->lastbits;
    while (nbits >= 8) {
        lastbyte = (lastbyte << 8) | ((num >> (nbits - 8)) /* & 0xff */);
        w->p[w->cnt++] = (uint8_t)(lastbyte >> lastbits);
        nbits -= 8;
    }
    if (nbits > 0) {
        lastbyte = (lastbyte << nbits) | num;
        lastbits += nbits;
        if (lastbits >= 8) {
            lastbits -= 8;
            w->p[w->cnt++] = (uint8_t)(lastbyte >> lastbits);
        }
    }
    w->lastbits = lastbits;
    w->lastbyte = lastbyte;
    if (lastbits > 0) w->p[w->cnt] = (uint8_t)(lastbyte << (8 - lastbits));
}

static void sendints(bitwr *w, int nints, int nbits, const unsigned sizes[], const unsigned nums[]) {
    unsigned bytes[32], nbytes = 0, bytecnt, tmp;
    tmp = nums[0];
    do {
        bytes[nbytes++] = tmp & 0xff;
        tmp >>= 8;
    } while (tmp != 0);
    for (int i = 1; i < nints; i++) {
        tmp = nums[i];
        for (bytecnt = 0; bytecnt < nbytes; bytecnt++) {
            tmp = bytes[bytecnt] * sizes[i] + tmp;
            bytes[bytecnt] = tmp & 0xff;
            tmp >>= 8;
        }
        while (tmp != 0) {
            bytes[bytecnt++] = tmp & 0xff;
            tmp >>= 8;
        }
        nbytes = bytecnt;
    }
    if ((unsigned)nbits >= nbytes * 8) {
        for (unsigned i = 0; i < nbytes; i++) sendbits(w, 8, bytes[i]);
        sendbits(w, nbits - (int)nbytes * 8, 0);
    } else {
        unsigned i;
        for (i = 0; i < nbytes - 1; i++) sendbits(w, 8, bytes[i]);
        sendbits(w, nbits - (int)(nbytes - 1) * 8, bytes[i]);
    }
}

static void put32(uint8_t *p, uint32_t v) {
    p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v;
}
static void putf(uint8_t *p, float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    put32(p, u);
}

/* Writes one frame; returns its length (0 if `cap` is too small).  A worst-case bound is 92 + 16*natoms. */
size_t orc_xtc_encode_frame(const float *xyz, int32_t natoms, const float box9[9], int32_t step, float time, float precision,
                            uint32_t magic, uint8_t *out, size_t cap) {
    if (cap < 56 + (natoms <= 9 ? (size_t)natoms * 12 : 40 + 16 * (size_t)natoms)) return 0;
    put32(out, magic); put32(out + 4, (uint32_t)natoms); put32(out + 8, (uint32_t)step); putf(out + 12, time);
    for (int k = 0; k < 9; ++k) putf(out + 16 + 4 * k, box9[k]);
    put32(out + 52, (uint32_t)natoms);
    if (natoms <= 9) {
        for (int k = 0; k < 3 * natoms; ++k) putf(out + 56 + 4 * k, xyz[k]);
        return 56 + (size_t)natoms * 12;
    }
    int *ip = (int *)malloc(sizeof(int) * 3 * (size_t)natoms);
    int minint[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, maxint[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
    for (int i = 0; i < natoms; ++i)
        for (int c = 0; c < 3; ++c) {
            const float f = xyz[3 * i + c] * precision;
            const int v = (int)(f >= 0 ? f + 0.5f : f - 0.5f);
            ip[3 * i + c] = v;
            if (v < minint[c]) minint[c] = v;
            if (v > maxint[c]) maxint[c] = v;
        }
    unsigned sizeint[3], bitsizeint[3] = {0, 0, 0};
    int bitsize;
    for (int c = 0; c < 3; ++c) sizeint[c] = (unsigned)(maxint[c] - minint[c] + 1);
    if ((sizeint[0] | sizeint[1] | sizeint[2]) > 0xffffff) {
        for (int c = 0; c < 3; ++c) bitsizeint[c] = (unsigned)sizeofint(sizeint[c]);
        bitsize = 0;
    } else {
        bitsize = sizeofints(3, sizeint);
    }
    /* fixed small index: deltas up to +-0.25 nm at precision 1000 */
    int smallidx = FIRSTIDX;
    while (smallidx < LASTIDX && magicints[smallidx] < 512) smallidx++;
    const int smallnum = magicints[smallidx] / 2;
    const unsigned sizesmall[3] = {(unsigned)magicints[smallidx], (unsigned)magicints[smallidx], (unsigned)magicints[smallidx]};
    putf(out + 56, precision);
    for (int c = 0; c < 3; ++c) { put32(out + 60 + 4 * c, (uint32_t)minint[c]); put32(out + 72 + 4 * c, (uint32_t)maxint[c]); }
    put32(out + 84, (uint32_t)smallidx);
    const size_t hdr = 56 + 32 + (magic == 2023 ? 8 : 4);
    bitwr w = {out + hdr, cap - hdr, 0, 0, 0};
    int i = 0, prevrun = -1;
    while (i < natoms) {
        /* Output order a0 a1 a2 ...: with a run the decoder emits [first small, large, rest], so the LARGE
         * record is a1, the first small record is a0 - a1, then a2 - a0, a3 - a2, ... (at most 8 small records) */
        int nsmall = 0;
        if (i + 1 < natoms) {
            const int *a0 = ip + 3 * i, *a1 = ip + 3 * (i + 1);
            int ok = 1;
            for (int c = 0; c < 3; ++c) {
                const int d = a0[c] - a1[c] + smallnum;
                if (d < 0 || d >= (int)sizesmall[c]) ok = 0;
            }
            if (ok) {
                nsmall = 1;
                const int *prev = a0;
                while (nsmall < 8 && i + 1 + nsmall < natoms) {
                    const int *cur = ip + 3 * (i + 1 + nsmall);
                    for (int c = 0; c < 3; ++c) {
                        const int d = cur[c] - prev[c] + smallnum;
                        if (d < 0 || d >= (int)sizesmall[c]) ok = 0;
                    }
                    if (!ok) break;
                    prev = cur;
                    nsmall++;
                }
            }
        }
        const int *large = nsmall > 0 ? ip + 3 * (i + 1) : ip + 3 * i;
        unsigned u[3];
        for (int c = 0; c < 3; ++c) u[c] = (unsigned)(large[c] - minint[c]);
        if (bitsize == 0) for (int c = 0; c < 3; ++c) sendbits(&w, (int)bitsizeint[c], u[c]);
        else sendints(&w, 3, bitsize, sizeint, u);
        const int run = 3 * nsmall;
        if (run != prevrun) {
            prevrun = run;
            sendbits(&w, 1, 1);
            sendbits(&w, 5, (unsigned)(run + 1));   /* is_smaller = 0 */
        } else {
            sendbits(&w, 1, 0);
        }
        if (nsmall > 0) {
            const int *a0 = ip + 3 * i, *a1 = ip + 3 * (i + 1);
            for (int c = 0; c < 3; ++c) u[c] = (unsigned)(a0[c] - a1[c] + smallnum);
            sendints(&w, 3, smallidx, sizesmall, u);
            const int *prev = a0;
            for (int k = 1; k < nsmall; ++k) {
                const int *cur = ip + 3 * (i + 1 + k);
                for (int c = 0; c < 3; ++c) u[c] = (unsigned)(cur[c] - prev[c] + smallnum);
                sendints(&w, 3, smallidx, sizesmall, u);
                prev = cur;
            }
            i += 1 + nsmall;
        } else {
            i += 1;
        }
    }
    free(ip);
    size_t nbytes = w.cnt + (w.lastbits ? 1 : 0);
    if (magic == 2023) { put32(out + 88, (uint32_t)((uint64_t)nbytes >> 32)); put32(out + 92, (uint32_t)nbytes); }
    else put32(out + 88, (uint32_t)nbytes);
    while (nbytes & 3) out[hdr + nbytes++] = 0;
    return hdr + nbytes;
}
