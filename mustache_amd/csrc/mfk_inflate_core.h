// mfk_inflate_core.h -- DEFLATE (RFC 1951) decoder for one BGZF block, written once for two callers: the gfx950 kernel in
// mfk_inflate.hip (one 64-lane wave per block) and a plain host build that the CPU tests check against zlib
// (tests/native/inflate_host.cpp).  The reference gets its inflate from htslib inside pysam (findflanks.py:22, 100-121);
// nothing of that code is used here: the decoder follows the RFC.
//
// How a wave decodes.  DEFLATE is serial inside a block: a symbol's position in the bit stream is known only after the
// symbol before it.  So all 64 lanes run the SAME decode (same instructions on the same values: no divergence, the table
// reads are LDS broadcasts), and the lane index is used only where there is something to share out: filling the lookup
// tables, holding the next 256 bytes of input (one dword per lane, fetched with a uniform readlane), and copying a match
// (lane k copies byte k).  The host build is the same code with a wave of one lane.  Two things the device build adds, both
// about instruction count (the kernel is bound by instruction issue, vector and scalar alike, not by memory): every branch
// on a value that all lanes share is taken through INFL_U below, and runs of literals are decoded in a loop of their own
// that holds nothing a literal does not need.
//
// Tables per decoder (4 KB, in LDS on the device): a 10-bit primary table for literal/length codes and a 9-bit one for
// distance codes (entry = symbol << 4 | code length, 0 = a longer code), and for the rare longer codes the canonical
// description itself (codes per length + symbols in canonical order), walked bit by bit.
#pragma once
#include <stdint.h>

#ifndef MFK_HD
#define MFK_HD
#endif
#ifndef MFK_TABLE
#define MFK_TABLE static const      // (the HIP build makes these __device__ arrays)
#endif
// lanes of one wave that wrote LDS for each other: no instruction may move across (free: the wave runs in lockstep)
#ifdef __HIP_DEVICE_COMPILE__
#define INFL_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#else
#define INFL_WAVE_SYNC() do { } while (0)
#endif

// A condition that is the same in every lane BY CONSTRUCTION (all lanes run the same decode) but not by anything the compiler
// can prove: taken as "any lane says so" (a ballot) it becomes a scalar branch; left as a vector compare, every such branch is compiled as masked
// divergent code (save the execution mask, branch if empty, restore), which was a good third of the kernel's instructions.
#ifdef __HIP_DEVICE_COMPILE__
#define INFL_U(c) (__builtin_amdgcn_ballot_w64((bool)(c)) != 0ull)   /* a vector compare into a scalar mask, and a scalar branch on it */
#else
#define INFL_U(c) (c)
#endif

#define INFL_LBITS 10
#define INFL_DBITS 9

enum InflStatus : int32_t {
    INFL_OK = 0,
    INFL_ERR_BTYPE = 1,        // reserved block type
    INFL_ERR_STORED = 2,       // LEN / NLEN of a stored block do not agree
    INFL_ERR_LENS = 3,         // bad code lengths (over-subscribed, bad repeat, no end-of-block code)
    INFL_ERR_CODE = 4,         // a bit pattern that is no code
    INFL_ERR_DIST = 5,         // distance beyond the start of the output, or a bad distance symbol
    INFL_ERR_OUT = 6,          // more output than the block's ISIZE
    INFL_ERR_SHORT = 7,        // the stream ended before the output was complete, or less output than ISIZE
    INFL_ERR_IN = 8,           // more bits consumed than the block holds
};

struct InflTables {
    uint16_t lit[1 << INFL_LBITS];
    uint16_t dst[1 << INFL_DBITS];
    uint16_t lcount[16], dcount[16];
    uint16_t lsym[288], dsym[32];
    uint8_t lens[320];
};

// ---- input: 32 bits at a byte position --------------------------------------------------------------------------------
// Host: straight from memory.  Device: the wave keeps 256 bytes of the stream, one dword per lane (one coalesced load per
// 256 bytes; a dependent global load per refill would cost a memory round trip every three or four symbols).
template <int WAVE>
struct InflInput;

template <>
struct InflInput<1> {
    const uint8_t *in; uint32_t n;
    MFK_HD void init(const uint8_t *p, uint32_t len, int) { in = p; n = len; }
    MFK_HD uint32_t fetch32(uint32_t pos, int) const {
        uint32_t w = 0;
        for (int k = 0; k < 4; k++) if (pos + k < n) w |= (uint32_t)in[pos + k] << (8 * k);
        return w;
    }
    MFK_HD uint8_t byte_at(uint32_t pos) const { return pos < n ? in[pos] : 0; }
};

#ifdef __HIPCC__
template <>
struct InflInput<64> {
    const uint8_t *in; uint32_t n;
    uint32_t cbase;      // byte position of lane 0's dword
    uint32_t chunk;      // this lane's dword
    __device__ void load(uint32_t base, int lane) {
        cbase = base;
        const uint32_t p = base + 4u * (uint32_t)lane;
        uint32_t w = 0;
        if (p + 4 <= n) {
            const uint8_t *a = in + p;   // (an unaligned dword load is fine on global memory; assembled from bytes to stay inside the buffer's alignment rules)
            w = (uint32_t)a[0] | ((uint32_t)a[1] << 8) | ((uint32_t)a[2] << 16) | ((uint32_t)a[3] << 24);
        } else {
            for (int k = 0; k < 4; k++) if (p + k < n) w |= (uint32_t)in[p + k] << (8 * k);
        }
        chunk = w;
    }
    __device__ void init(const uint8_t *p, uint32_t len, int lane) { in = p; n = len; load(0, lane); }
    __device__ uint32_t fetch32(uint32_t pos, int lane) {
        if (INFL_U(pos < cbase || pos + 8 > cbase + 256)) load(pos & ~3u, lane);   // wave-uniform
        const uint32_t rel = pos - cbase, idx = __builtin_amdgcn_readfirstlane((int)(rel >> 2));
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)chunk, (int)idx);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)chunk, (int)idx + 1);
        const uint32_t sh = (rel & 3u) * 8u;
        return sh ? (lo >> sh) | (hi << (32u - sh)) : lo;
    }
    __device__ uint8_t byte_at(uint32_t pos) const { return pos < n ? in[pos] : 0; }
};
#endif

// ---- canonical Huffman tables -----------------------------------------------------------------------------------------
MFK_HD inline uint32_t infl_rev(uint32_t c, int l) {
    uint32_t r = 0;
    for (int k = 0; k < l; k++) { r = (r << 1) | (c & 1u); c >>= 1; }
    return r;
}

// which lanes of the wave hold `pred` (bit k = lane k); the host's wave has one lane
template <int WAVE>
MFK_HD inline uint64_t infl_ballot(bool pred) {
#ifdef __HIP_DEVICE_COMPILE__
    if (WAVE == 64) return __ballot(pred);
#endif
    return pred ? 1ull : 0ull;
}
MFK_HD inline int infl_popc(uint64_t m) {
#ifdef __HIP_DEVICE_COMPILE__
    return __popcll(m);
#else
    return __builtin_popcountll(m);
#endif
}

// lens[0..n) -> primary table of `bits` bits (tab), codes per length (count), symbols in canonical order (sym).
// Returns 0, or 1 for an over-subscribed set of lengths.  (An incomplete set is allowed: a pattern that is no code fails
// when it is met, which is what RFC-conforming streams with a single distance code need.)
// Lane k takes symbols k, k + WAVE, ...: the rank of a symbol among the symbols of its length (= its canonical code minus the
// first code of that length) is a ballot and a popcount, so nothing here is a serial walk over the symbols and the per-length
// numbers stay in registers (statically indexed: unrolled over the 15 lengths).
template <int WAVE>
MFK_HD inline int infl_build(const uint8_t *lens, int n, uint16_t *tab, int bits, uint16_t *count, uint16_t *sym, int lane) {
    INFL_WAVE_SYNC();
    const uint64_t below = ((uint64_t)1 << lane) - 1u;
    uint32_t cnt[16];
#pragma unroll
    for (int l = 0; l < 16; l++) cnt[l] = 0;
    for (int s0 = 0; s0 < n; s0 += WAVE) {
        const int s = s0 + lane;
        const int my = s < n ? (int)lens[s] : 0;
#pragma unroll
        for (int l = 1; l < 16; l++) cnt[l] += (uint32_t)infl_popc(infl_ballot<WAVE>(my == l));
    }
    int left = 1;
#pragma unroll
    for (int l = 1; l < 16; l++) { left <<= 1; left -= (int)cnt[l]; if (left < 0) return 1; }
    uint32_t offs[16], first[16];   // position of a length's first symbol in canonical order; its first code
    offs[0] = 0; first[0] = 0; offs[1] = 0;
    uint32_t code = 0;
#pragma unroll
    for (int l = 1; l < 16; l++) { code = (code + (l > 1 ? cnt[l - 1] : 0u)) << 1; first[l] = code; if (l < 15) offs[l + 1] = offs[l] + cnt[l]; }
    for (int k = lane; k < (1 << bits); k += WAVE) tab[k] = 0;
#pragma unroll
    for (int l = 1; l < 16; l++) if (lane == 0) count[l] = (uint16_t)cnt[l];
    if (lane == 0) count[0] = 0;
    INFL_WAVE_SYNC();
    uint32_t seen[16];              // symbols of each length in the chunks before this one
#pragma unroll
    for (int l = 0; l < 16; l++) seen[l] = 0;
    for (int s0 = 0; s0 < n; s0 += WAVE) {
        const int s = s0 + lane;
        const int my = s < n ? (int)lens[s] : 0;
        uint32_t rank = 0, c0 = 0, o0 = 0;
#pragma unroll
        for (int l = 1; l < 16; l++) {
            const uint64_t m = infl_ballot<WAVE>(my == l);
            if (my == l) { rank = seen[l] + (uint32_t)infl_popc(m & below); c0 = first[l]; o0 = offs[l]; }
            seen[l] += (uint32_t)infl_popc(m);
        }
        if (my) {
            sym[o0 + rank] = (uint16_t)s;
            if (my <= bits) {
                const uint32_t r = infl_rev(c0 + rank, my);
                const uint16_t e = (uint16_t)((s << 4) | my);
                for (uint32_t k = r; k < (1u << bits); k += 1u << my) tab[k] = e;
            }
        }
    }
    INFL_WAVE_SYNC();
    return 0;
}

// one symbol: primary table, else the canonical walk.  Returns the symbol or -1; *used = bits of the code.
MFK_HD inline int infl_symbol(uint64_t buf, const uint16_t *tab, int bits, const uint16_t *count, const uint16_t *sym, int *used) {
    const uint16_t e = tab[buf & ((1u << bits) - 1u)];
    if (INFL_U(e & 15)) { *used = e & 15; return e >> 4; }
    int code = 0, first = 0, index = 0;
    for (int l = 1; l <= 15; l++) {
        code |= (int)((buf >> (l - 1)) & 1u);
        const int c = count[l];
        if (INFL_U(code - c < first)) { *used = l; return sym[index + (code - first)]; }
        index += c; first += c; first <<= 1; code <<= 1;
    }
    return -1;
}

MFK_TABLE uint8_t INFL_CLORD[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// Bytes written by one lane and read by another lane OF THE SAME WAVE go through memory on the device: the wave's stores so
// far must have landed before the loads are issued.  Workgroup scope is enough (one CU, one vector L1) and costs a wait for
// the outstanding stores; an agent-scope fence (__threadfence) also writes back and invalidates the L2 of this XCD, tens of
// microseconds per match: 0.6 GB/s instead of tens.
MFK_HD inline void infl_publish() {
#ifdef __HIP_DEVICE_COMPILE__
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
#endif
}

// ---- one block ----------------------------------------------------------------------------------------------------------
// in[0..clen): the raw deflate stream of a BGZF block; out[0..ulen): its ISIZE bytes.  Every lane of the wave calls this with
// the same arguments (and its lane index); every lane returns the same status.  Never reads outside in[0..clen) or
// out[0..ulen), never writes outside out[0..ulen), whatever the stream holds.
template <int WAVE>
MFK_HD inline int32_t infl_block(const uint8_t *in, uint32_t clen, uint8_t *out, uint32_t ulen, InflTables &T, int lane) {
    InflInput<WAVE> rd;
    rd.init(in, clen, lane);
    uint64_t buf = 0;
    int cnt = 0;
    uint32_t pos = 0;      // next input byte to put into the bit buffer
    uint32_t opos = 0;
    uint32_t seen = 0;     // out[0..seen) is known to have landed: a match that reads only from there need not wait for the stores behind it
#define INFL_FILL() do { if (INFL_U(cnt <= 32)) { buf |= (uint64_t)rd.fetch32(pos, lane) << cnt; cnt += 32; pos += 4; } } while (0)
#define INFL_TAKE(nb) do { buf >>= (nb); cnt -= (nb); } while (0)
    for (;;) {
        INFL_FILL();
        const int bfinal = (int)(buf & 1u), btype = (int)((buf >> 1) & 3u);
        INFL_TAKE(3);
        if (btype == 3) return INFL_ERR_BTYPE;
        if (btype == 0) {
            INFL_TAKE(cnt & 7);                    // to the next byte boundary
            INFL_FILL();
            const uint32_t len = (uint32_t)(buf & 0xFFFFu), nlen = (uint32_t)((buf >> 16) & 0xFFFFu);
            INFL_TAKE(32);
            if ((len ^ nlen) != 0xFFFFu) return INFL_ERR_STORED;
            // the bytes follow in the stream; what the bit buffer holds beyond this point is dropped and read again
            const uint32_t src = pos - (uint32_t)(cnt >> 3);
            if (src + len > clen) return INFL_ERR_IN;
            if (opos + len > ulen) return INFL_ERR_OUT;
            for (uint32_t k = (uint32_t)lane; k < len; k += WAVE) out[opos + k] = in[src + k];
            opos += len;
            pos = src + len; buf = 0; cnt = 0;
        } else {
            if (btype == 1) {
                for (int s = lane; s < 288; s += WAVE) T.lens[s] = (uint8_t)(s < 144 ? 8 : (s < 256 ? 9 : (s < 280 ? 7 : 8)));
                infl_build<WAVE>(T.lens, 288, T.lit, INFL_LBITS, T.lcount, T.lsym, lane);
                for (int s = lane; s < 30; s += WAVE) T.lens[s] = 5;
                infl_build<WAVE>(T.lens, 30, T.dst, INFL_DBITS, T.dcount, T.dsym, lane);
            } else {
                const int hlit = (int)(buf & 31u) + 257, hdist = (int)((buf >> 5) & 31u) + 1, hclen = (int)((buf >> 10) & 15u) + 4;
                INFL_TAKE(14);
                if (hlit > 286 || hdist > 30) return INFL_ERR_LENS;
                for (int s = lane; s < 19; s += WAVE) T.lens[s] = 0;
                for (int k = 0; k < hclen; k++) {
                    INFL_FILL();
                    if (lane == 0) T.lens[INFL_CLORD[k]] = (uint8_t)(buf & 7u);
                    INFL_TAKE(3);
                }
                // the code-length code uses the distance table's storage for a moment (7-bit codes: the primary table holds them all)
                if (infl_build<WAVE>(T.lens, 19, T.dst, 7, T.dcount, T.dsym, lane)) return INFL_ERR_LENS;
                int idx = 0, prev = 0;
                while (idx < hlit + hdist) {
                    INFL_FILL();
                    int used = 0;
                    const int s = infl_symbol(buf, T.dst, 7, T.dcount, T.dsym, &used);
                    if (s < 0) return INFL_ERR_CODE;
                    INFL_TAKE(used);
                    if (s < 16) { if (lane == 0) T.lens[idx] = (uint8_t)s; idx++; prev = s; }
                    else {
                        int rep, val = 0;
                        if (s == 16) { if (idx == 0) return INFL_ERR_LENS; val = prev; rep = 3 + (int)(buf & 3u); INFL_TAKE(2); }
                        else if (s == 17) { rep = 3 + (int)(buf & 7u); INFL_TAKE(3); }
                        else { rep = 11 + (int)(buf & 127u); INFL_TAKE(7); }
                        if (idx + rep > hlit + hdist) return INFL_ERR_LENS;
                        for (int k = lane; k < rep; k += WAVE) T.lens[idx + k] = (uint8_t)val;
                        idx += rep; prev = val;
                    }
                }
                INFL_WAVE_SYNC();
                if (T.lens[256] == 0) return INFL_ERR_LENS;
                // the distance lengths sit behind the literal/length ones: build them first into their own arrays (the builder reads
                // lens only), then the literal/length table
                if (infl_build<WAVE>(T.lens + hlit, hdist, T.dst, INFL_DBITS, T.dcount, T.dsym, lane)) return INFL_ERR_LENS;
                if (infl_build<WAVE>(T.lens, hlit, T.lit, INFL_LBITS, T.lcount, T.lsym, lane)) return INFL_ERR_LENS;
            }
            for (;;) {
#ifdef __HIP_DEVICE_COMPILE__
                // literals in a row, the common case: nothing in this loop that a literal does not need (every lane stores the
                // same byte to the same address: one request, and no change of the execution mask)
                // (one way out: a literal with no room left is found by the general code below as well.  After a refill the buffer
                // holds 33 bits or more and a code of the primary table is at most INFL_LBITS = 10: three literals per refill check.)
#define INFL_LITERAL()                                                                                                          \
                    {                                                                                                           \
                        const uint32_t e = T.lit[(uint32_t)buf & ((1u << INFL_LBITS) - 1u)];                                    \
                        if (!INFL_U((e & 15u) != 0u && e < (256u << 4) && opos < ulen)) break;   /* a match, the end, a long code */ \
                        out[opos] = (uint8_t)(e >> 4);                                                                          \
                        opos++;                                                                                                 \
                        INFL_TAKE((int)(e & 15u));                                                                              \
                    }
                for (;;) {
                    INFL_FILL();
                    INFL_LITERAL()
                    INFL_LITERAL()
                    INFL_LITERAL()
                }
#undef INFL_LITERAL
#endif
                INFL_FILL();
                int used = 0;
                int s = infl_symbol(buf, T.lit, INFL_LBITS, T.lcount, T.lsym, &used);
                if (INFL_U(s < 0)) return INFL_ERR_CODE;
                INFL_TAKE(used);
                if (INFL_U(s < 256)) {
                    if (INFL_U(opos >= ulen)) return INFL_ERR_OUT;
                    if (lane == 0) out[opos] = (uint8_t)s;
                    opos++;
                    continue;
                }
                if (INFL_U(s == 256)) break;
                s -= 257;
                if (INFL_U(s >= 29)) return INFL_ERR_CODE;
                // base and extra bits by arithmetic (RFC 1951 3.2.5 is regular): a table in memory costs a dependent load per match
                const int lext = s < 8 || s == 28 ? 0 : (s >> 2) - 1;
                const uint32_t lbase = s < 8 ? 3u + (uint32_t)s : (s == 28 ? 258u : ((4u + ((uint32_t)s & 3u)) << lext) + 3u);
                const uint32_t len = lbase + (uint32_t)(buf & ((1u << lext) - 1u));
                INFL_TAKE(lext);
                INFL_FILL();
                const int ds = infl_symbol(buf, T.dst, INFL_DBITS, T.dcount, T.dsym, &used);
                if (INFL_U(ds < 0)) return INFL_ERR_CODE;
                INFL_TAKE(used);
                if (INFL_U(ds >= 30)) return INFL_ERR_DIST;
                const int dext = ds < 4 ? 0 : (ds >> 1) - 1;
                const uint32_t dbase = ds < 4 ? (uint32_t)ds + 1u : ((2u + ((uint32_t)ds & 1u)) << dext) + 1u;
                const uint32_t dist = dbase + (uint32_t)(buf & ((1u << dext) - 1u));
                INFL_TAKE(dext);
                if (INFL_U(dist > opos)) return INFL_ERR_DIST;
                if (INFL_U(opos + len > ulen)) return INFL_ERR_OUT;
                // lane k copies byte k; a match that overlaps itself repeats its first `dist` bytes, all of them written before
                const uint32_t src0 = opos - dist, src1 = src0 + (dist >= len ? len : dist);   // the bytes it reads: out[src0..src1)
                if (INFL_U(src1 > seen)) { infl_publish(); seen = opos; }
                const uint8_t *from = out + src0;
                for (uint32_t k = (uint32_t)lane; k < len; k += WAVE) out[opos + k] = from[dist >= len ? k : k % dist];
                opos += len;
            }
        }
        if (bfinal) break;
    }
#undef INFL_FILL
#undef INFL_TAKE
    // bits taken from beyond the stream are zeros the reader made up
    if (pos - (uint32_t)(cnt >> 3) > clen) return INFL_ERR_IN;
    if (opos != ulen) return INFL_ERR_SHORT;
    return INFL_OK;
}
