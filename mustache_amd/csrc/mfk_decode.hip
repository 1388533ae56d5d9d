// mfk_decode.hip -- BAM decode on the device: compressed BGZF bytes in HBM -> the packed SoA reads of include/mfk.h in HBM.
// The device-side statement of what libmfkio.so (mfk_bamio.cpp) does on host threads and mustache_amd/pack.py::pack_reads
// states in Python; both restate what the reference gets from pysam while it iterates the BAM (findflanks.py:100-121) and
// evaluates sctools.py:5-31, 99-109 per read.  Nothing but the compressed file crosses PCIe.
//
//   k_bgzf_inflate    one wave per BGZF block (mfk_inflate.hip)
//   k_first_record    record boundaries: a record's length field chains from the record before it, so every block GUESSES its
//                     first record start (first offset where six well-formed records chain; 64 candidates per step) ...
//   k_walk            ... walks its own records from there, and the guesses are VERIFIED: block b's walk must land exactly
//                     on block b+1's guess.  By induction from the known first record the boundaries are then exact; a
//                     guess that fails fails the call (the caller falls back to the host reader), never the result.
//   k_parse_count     thread per record: what the packer decides per record (kept?, aligned blocks, pooled bytes) or its
//                     rejection; three scans place the rows
//   k_parse_fill      writes the nine columns, the block and pool side tables and the name hashes
//   k_finish          sortedness, first read of every contig, longest reference span
//   sort + k_link     `link` = previous read with the same query name: reads sorted by name hash (stable), equal hashes
//                     confirmed by comparing the names byte for byte
//   k_marks           the record index (one mark per 32 KiB of the inflated stream) that pairflanks' region pass reads
#include "mfk_host.h"

#include <vector>

extern "C" int mfk_bgzf_inflate(mfk_ctx *ctx, const void *d_comp, int64_t comp_bytes, const mfk_bgzf_block *blocks, int64_t n_blocks,
                                void *d_out, int64_t out_bytes, int32_t *h_status, int64_t *n_bad_out);

namespace {

enum { CM = 0, CI = 1, CD = 2, CN = 3, CS = 4, CH = 5, CP = 6, CEQ = 7, CX = 8 };
__device__ __forceinline__ bool d_ref_op(unsigned op) { return op == CM || op == CD || op == CN || op == CEQ || op == CX; }
__device__ __forceinline__ bool d_aligned_op(unsigned op) { return op == CM || op == CEQ || op == CX; }
__device__ __forceinline__ uint32_t d_le32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
__device__ __forceinline__ uint32_t d_le16(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }

// rejections of a record, worded by the caller like the host packer words them (mustache_amd/gpudecode.py)
enum { PE_NONE = 0, PE_NEG_LSEQ = 1, PE_TRUNC = 2, PE_CONTIG = 3, PE_END = 4, PE_UNSORTED = 5, PE_TAG = 6, PE_NO_MD_LEFT = 7, PE_NO_MD_RIGHT = 8,
       PE_CLIP_INSIDE = 9, PE_NO_ALIGNED = 10, PE_DELETION_END = 11, PE_LONG_QUERY = 12 };

struct DecodeState {
    uint8_t *u = nullptr; size_t u_cap = 0;          // inflated stream
    char *work = nullptr; size_t work_cap = 0;       // everything per block and per record, carved below
    uint64_t U = 0, first_off = 0;
    int64_t n_blocks = 0, n_rec = 0, n_reads = 0, n_cx = 0, n_blk = 0, n_pool = 0;
    uint64_t pool_bytes = 0;
    int32_t n_contigs = 0, pool_mode = 0;
    uint32_t exclude_flags = 0;
    // device arrays inside `work`
    uint64_t *blk_uoff = nullptr, *first = nullptr, *land = nullptr, *rec_off = nullptr, *poff = nullptr;
    uint32_t *cnt = nullptr, *base = nullptr, *keep = nullptr, *kidx = nullptr, *nblk = nullptr, *boff = nullptr, *plen = nullptr, *cxf = nullptr, *cxidx = nullptr,
             *plf = nullptr, *pidx = nullptr;
    unsigned long long *err = nullptr;               // min over records of (record index << 8 | code); ~0 = none
    uint32_t *flagw = nullptr;                       // [0] boundary check failed, [1] bad record size in a walk, [2] max_span
    bool ready = false;
};

struct DecParams { uint64_t U; int32_t n_contigs, pool_mode; uint32_t exclude_flags; };

// ---- record boundaries ----------------------------------------------------------------------------------------------------
__device__ bool d_plausible(const uint8_t *u, uint64_t U, int32_t n_contigs, uint64_t q, uint64_t *next) {
    if (q + 36 > U) return false;
    const uint32_t bs = d_le32(u + q);
    if (bs < 32 || bs > (1u << 24) || q + 4 + (uint64_t)bs > U) return false;
    const uint8_t *p = u + q + 4;
    const int32_t ref = (int32_t)d_le32(p), pos = (int32_t)d_le32(p + 4), mref = (int32_t)d_le32(p + 20), l_seq = (int32_t)d_le32(p + 16);
    if (ref < -1 || ref >= n_contigs || mref < -1 || mref >= n_contigs || pos < -1 || l_seq < 0) return false;
    const unsigned l_name = p[8], n_cig = d_le16(p + 12);
    if (!l_name) return false;
    const uint64_t fixed = 32 + (uint64_t)l_name + 4 * (uint64_t)n_cig + ((uint64_t)l_seq + 1) / 2 + (uint64_t)l_seq;
    if (fixed > bs || p[32 + l_name - 1] != 0) return false;
    for (unsigned i = 0; i + 1 < l_name; ++i) if (p[32 + i] < 33 || p[32 + i] > 126) return false;
    for (unsigned k = 0; k < n_cig; ++k) if ((d_le32(p + 32 + l_name + 4 * k) & 15u) > 8) return false;
    *next = q + 4 + (uint64_t)bs;
    return true;
}

__device__ bool d_chain_plausible(const uint8_t *u, uint64_t U, int32_t n_contigs, uint64_t q) {
    for (int d = 0; d < 6; ++d) {
        if (q == U) return d >= 1;    // the chain ends exactly where the stream ends
        uint64_t nx;
        if (!d_plausible(u, U, n_contigs, q, &nx)) return false;
        q = nx;
    }
    return true;
}

// one wave per block: first[b] = the guessed start of the first record at or behind the block's first byte (U = none)
__global__ __launch_bounds__(256) void k_first_record(const uint8_t *u, DecParams P, const uint64_t *blk_uoff, uint32_t n_blocks, uint64_t first_off, uint64_t *first) {
    const uint32_t b = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (b >= n_blocks) return;
    const uint64_t lo = blk_uoff[b];
    if (lo <= first_off) { if (lane == 0) first[b] = first_off; return; }   // the block that holds the first record, and those before it
    // a record is at most 2^24 + 4 bytes here (d_plausible), so a start exists within that distance or not at all
    const uint64_t hi = lo + (1u << 24) + 64 < P.U ? lo + (1u << 24) + 64 : P.U;
    uint64_t found = P.U;
    for (uint64_t q0 = lo; q0 < hi; q0 += 64) {
        const uint64_t q = q0 + (uint64_t)lane;
        const bool ok = q < hi && d_chain_plausible(u, P.U, P.n_contigs, q);
        const unsigned long long m = __ballot(ok);
        if (m) { found = q0 + (uint64_t)(__ffsll((long long)m) - 1); break; }
    }
    if (lane == 0) first[b] = found;
}

// thread per block: the records that START in [blk_uoff[b], blk_uoff[b+1]).  FILL = false: their number and where the walk lands;
// FILL = true: their offsets at base[b].., and the check that the walk landed on the next block's guess.
template <bool FILL>
__global__ void k_walk(const uint8_t *u, DecParams P, const uint64_t *blk_uoff, uint32_t n_blocks, const uint64_t *first, uint32_t *cnt, uint64_t *land,
                       const uint32_t *base, uint64_t *rec_off, uint32_t *flagw) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_blocks) return;
    const uint64_t end = b + 1 < n_blocks ? blk_uoff[b + 1] : P.U;
    uint64_t q = first[b];
    uint32_t n = 0;
    const uint32_t at = FILL ? base[b] : 0;
    while (q < end) {
        if (q + 4 > P.U) { atomicOr(&flagw[1], 2u); break; }                       // a length field cut off by the end of the stream
        const uint32_t bs = d_le32(u + q);
        if (bs < 32 || bs > (1u << 30)) { atomicOr(&flagw[1], 1u); break; }        // bad alignment record size
        if (q + 4 + (uint64_t)bs > P.U) { atomicOr(&flagw[1], 2u); break; }        // truncated record at the end of the file
        if (FILL) rec_off[at + n] = q;
        n++;
        q += 4 + (uint64_t)bs;
    }
    if (!FILL) { cnt[b] = n; land[b] = q; }
    else {
        const uint64_t want = b + 1 < n_blocks ? first[b + 1] : P.U;
        if (land[b] != want && !(flagw[1])) atomicOr(&flagw[0], 1u);
    }
}

// ---- one record -------------------------------------------------------------------------------------------------------------
struct RecFacts {
    int err;              // PE_*
    bool keep, simple, pool;
    int32_t ref_id, pos, rend, l_seq;
    uint32_t n_cig, l_name, mapq, lk, rk, nb, first_len, last_len;
    const uint8_t *cig;
};

// the MD tag of a record's tag area t[0..n): *md / *md_len (nullptr when there is none).  false = malformed tags.
__device__ bool d_find_md(const uint8_t *t, uint64_t n, const uint8_t **md, uint64_t *md_len) {
    *md = nullptr; *md_len = 0;
    uint64_t o = 0;
    while (o + 3 <= n) {
        const uint8_t t0 = t[o], t1 = t[o + 1], typ = t[o + 2];
        o += 3;
        if (typ == 'A' || typ == 'c' || typ == 'C') o += 1;
        else if (typ == 's' || typ == 'S') o += 2;
        else if (typ == 'i' || typ == 'I' || typ == 'f') o += 4;
        else if (typ == 'Z' || typ == 'H') {
            uint64_t e = o;
            while (e < n && t[e] != 0) e++;
            if (e >= n) return false;                                   // unterminated string tag
            if (t0 == 'M' && t1 == 'D') { *md = t + o; *md_len = e - o; return true; }
            o = e + 1;
        } else if (typ == 'B') {
            if (o + 5 > n) return false;                                // truncated array tag
            const uint8_t sub = t[o];
            const uint64_t w = (sub == 'c' || sub == 'C' || sub == 'A') ? 1 : (sub == 's' || sub == 'S') ? 2 : (sub == 'i' || sub == 'I' || sub == 'f') ? 4 : 0;
            if (!w) return false;                                       // bad array tag type
            o += 5 + (uint64_t)d_le32(t + o + 1) * w;
        } else return false;                                            // bad tag type
    }
    return true;
}

// mfk_bamio.cpp parse_record / pack.py::pack_reads, decision by decision (p = first byte behind block_size)
__device__ RecFacts d_record(const uint8_t *p, uint32_t bs, const DecParams &P) {
    RecFacts f;
    f.err = PE_NONE; f.keep = false; f.simple = true; f.pool = false; f.nb = 0; f.lk = f.rk = 0; f.rend = 0; f.first_len = f.last_len = 0;
    f.ref_id = (int32_t)d_le32(p); f.pos = (int32_t)d_le32(p + 4);
    f.l_name = p[8]; f.mapq = p[9]; f.n_cig = d_le16(p + 12);
    const unsigned flag = d_le16(p + 14);
    f.l_seq = (int32_t)d_le32(p + 16);
    f.cig = p + 32 + f.l_name;
    if (f.l_seq < 0) { f.err = PE_NEG_LSEQ; return f; }
    const uint64_t fixed = 32 + (uint64_t)f.l_name + 4 * (uint64_t)f.n_cig + ((uint64_t)f.l_seq + 1) / 2 + (uint64_t)f.l_seq;
    if (fixed > bs) { f.err = PE_TRUNC; return f; }
    if (f.n_cig == 0 || (flag & 4) || f.ref_id < 0) return f;            // unmapped: never reaches the CIGAR (quirk Q17)
    if (flag & P.exclude_flags) return f;                                // e.g. secondary / supplementary (samtools.py:9-18)
    if (f.ref_id >= P.n_contigs) { f.err = PE_CONTIG; return f; }
    const uint8_t *cig = f.cig;
    const unsigned n_cig = f.n_cig;
    int64_t rend64 = f.pos;
    for (unsigned k = 0; k < n_cig; ++k) { const uint32_t c = d_le32(cig + 4 * k); if (d_ref_op(c & 15u)) rend64 += c >> 4; }
    if (rend64 > 0x7FFFFFFFll) { f.err = PE_END; return f; }
    f.rend = (int32_t)rend64;
    const uint8_t *md; uint64_t md_len;
    if (!d_find_md(p + fixed, bs - fixed, &md, &md_len)) { f.err = PE_TAG; return f; }
    const uint32_t c_first = d_le32(cig), c_last = d_le32(cig + 4 * (n_cig - 1));
    const unsigned first = c_first & 15u, last = c_last & 15u;
    if (first == CS) f.lk = 1;
    else { if (!md || !md_len) { f.err = PE_NO_MD_LEFT; return f; } f.lk = md[0] == '0' ? 2 : 0; }
    if (last == CS) f.rk = 1;
    else {
        if (!md || !md_len) { f.err = PE_NO_MD_RIGHT; return f; }
        f.rk = (md[md_len - 1] == '0' && md_len >= 2 && !(md[md_len - 2] >= '0' && md[md_len - 2] <= '9')) ? 2 : 0;
    }
    unsigned lo = 0, hi = n_cig;
    while (lo < hi) { const unsigned op = d_le32(cig + 4 * lo) & 15u; if (op == CH || op == CS) ++lo; else break; }
    while (hi > lo) { const unsigned op = d_le32(cig + 4 * (hi - 1)) & 15u; if (op == CH || op == CS) --hi; else break; }
    unsigned n_core = 0, first_ref = 99, last_ref = 99;
    for (unsigned k = lo; k < hi; ++k) {
        const unsigned op = d_le32(cig + 4 * k) & 15u;
        if (op == CP) continue;
        ++n_core;
        if (op == CH || op == CS) { f.err = PE_CLIP_INSIDE; return f; }
        if (op != CI) { if (first_ref == 99) first_ref = op; last_ref = op; }
    }
    if (!n_core) { f.err = PE_NO_ALIGNED; return f; }
    if (first_ref == 99 || first_ref == CD || first_ref == CN || last_ref == CD || last_ref == CN) { f.err = PE_DELETION_END; return f; }
    f.simple = n_core == 1 && !(lo > 0 && first == CH && (d_le32(cig + 4 * (lo - 1)) & 15u) == CS);
    if (f.l_seq > 65535) { f.err = PE_LONG_QUERY; return f; }
    if (!f.simple) for (unsigned k = 0; k < n_cig; ++k) if (d_aligned_op(d_le32(cig + 4 * k) & 15u)) f.nb++;
    f.pool = (P.pool_mode == 2 || (P.pool_mode == 1 && (f.lk || f.rk))) && f.l_seq > 0;
    f.first_len = c_first >> 4; f.last_len = c_last >> 4;
    f.keep = true;
    return f;
}

__global__ void k_parse_count(const uint8_t *u, DecParams P, const uint64_t *rec_off, uint32_t n_rec, uint32_t *keep, uint32_t *nblk, uint32_t *plen,
                              uint32_t *cxf, uint32_t *plf, unsigned long long *err) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rec) return;
    const uint64_t q = rec_off[i];
    const RecFacts f = d_record(u + q + 4, d_le32(u + q), P);
    if (f.err) atomicMin(err, ((unsigned long long)i << 8) | (unsigned long long)f.err);
    keep[i] = f.keep ? 1u : 0u;
    nblk[i] = f.keep ? f.nb : 0u;
    cxf[i] = f.keep && !f.simple ? 1u : 0u;
    const uint32_t pl = f.keep && f.pool ? (uint32_t)f.l_seq + ((uint32_t)f.l_seq & 1u) : 0u;
    plen[i] = pl;
    plf[i] = pl ? 1u : 0u;
}

__device__ __forceinline__ uint64_t d_hash_name(const uint8_t *s, uint32_t n) {   // (any hash will do: equal hashes are confirmed on the names)
    uint64_t h = 0xcbf29ce484222325ull ^ ((uint64_t)n * 0x9e3779b97f4a7c15ull);
    for (uint32_t i = 0; i < n; i++) { h = (h ^ (uint64_t)s[i]) * 0x100000001b3ull; }
    h ^= h >> 32; h *= 0xd6e8feb86659fd93ull; h ^= h >> 29;
    return h;
}

struct DevCols {   // the caller's device buffers (mfk_reads of the fetch call)
    int32_t *ref_start, *ref_end, *link;
    uint16_t *contig_id, *qlen, *first_op_len, *last_op_len;
    uint8_t *mapq, *flags;
    uint32_t *cx_read, *cx_off; int32_t *blk_start, *blk_end, *blk_ins;
    uint32_t *pool_read; uint64_t *pool_off; uint8_t *pool_bases, *pool_quals;
};

__global__ void k_parse_fill(const uint8_t *u, DecParams P, const uint64_t *rec_off, uint32_t n_rec, const uint32_t *keep, const uint32_t *kidx,
                             const uint32_t *boff, const uint32_t *cxidx, const uint32_t *pidx, const uint64_t *poff, DevCols C,
                             uint64_t *hash, uint32_t *kept_rec, int64_t *keys) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rec || !keep[i]) return;
    const uint64_t q = rec_off[i];
    const uint8_t *p = u + q + 4;
    const RecFacts f = d_record(p, d_le32(u + q), P);
    const uint32_t k = kidx[i];
    unsigned flags = f.lk | (f.rk << 2);
    if (!f.simple) {
        flags |= 16;
        const uint32_t c = cxidx[i];
        C.cx_read[c] = k;
        C.cx_off[c] = boff[i];
        uint32_t at = boff[i], nb = 0;
        int32_t rp = f.pos, pending = 0;
        for (unsigned kk = 0; kk < f.n_cig; ++kk) {
            const uint32_t cg = d_le32(f.cig + 4 * kk);
            const unsigned op = cg & 15u;
            const int32_t l = (int32_t)(cg >> 4);
            if (d_aligned_op(op)) { C.blk_start[at] = rp; C.blk_end[at] = rp + l; C.blk_ins[at] = pending; at++; pending = 0; rp += l; ++nb; }
            else if (op == CI || (op == CS && kk > 0 && nb == 0)) pending += l;
            else if (op == CD || op == CN) rp += l;
        }
    }
    if (f.pool) {
        flags |= 32;
        const uint32_t pi = pidx[i];
        const uint64_t o = poff[i];
        C.pool_read[pi] = k;
        C.pool_off[pi] = o;
        const uint32_t l_seq = (uint32_t)f.l_seq, plen = l_seq + (l_seq & 1u), nbytes = plen >> 1;
        const uint8_t *seq = f.cig + 4 * (uint64_t)f.n_cig, *qual = seq + nbytes;
        uint8_t *pb = C.pool_bases + (o >> 1), *pq = C.pool_quals + o;
        for (uint32_t t = 0; t < nbytes; t++) pb[t] = seq[t];
        if (l_seq & 1u) pb[nbytes - 1] &= 0xF0;
        const bool noq = qual[0] == 0xFF;
        for (uint32_t t = 0; t < l_seq; t++) pq[t] = noq ? 0 : qual[t];
        if (l_seq & 1u) pq[l_seq] = 0;
    }
    C.ref_start[k] = f.pos; C.ref_end[k] = f.rend; C.contig_id[k] = (uint16_t)f.ref_id; C.qlen[k] = (uint16_t)f.l_seq;
    C.first_op_len[k] = (uint16_t)(f.first_len < 65535u ? f.first_len : 65535u);
    C.last_op_len[k] = (uint16_t)(f.last_len < 65535u ? f.last_len : 65535u);
    C.mapq[k] = (uint8_t)f.mapq; C.flags[k] = (uint8_t)flags;
    hash[k] = d_hash_name(p + 32, f.l_name ? f.l_name - 1 : 0);
    kept_rec[k] = i;
    keys[k] = ((int64_t)f.ref_id << 32) + (int64_t)(uint32_t)f.pos;
}

// sortedness (the reference needs a sorted, indexed BAM), first read of every contig, longest reference span
__global__ void k_finish(const int64_t *keys, DevCols C, uint32_t n, int32_t n_contigs, int64_t *crs, uint32_t *flagw, unsigned long long *err, const uint32_t *kept_rec) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t span = 0;
    if (k < n) {
        const int32_t c = C.contig_id[k];
        if (k == 0) for (int32_t x = 0; x <= c; x++) crs[x] = 0;
        else {
            if (keys[k] < keys[k - 1]) atomicMin(err, ((unsigned long long)kept_rec[k] << 8) | (unsigned long long)PE_UNSORTED);
            const int32_t pc = C.contig_id[k - 1];
            for (int32_t x = pc + 1; x <= c; x++) crs[x] = k;
        }
        if (k == n - 1) for (int32_t x = c + 1; x <= n_contigs; x++) crs[x] = n;
        span = (uint32_t)(C.ref_end[k] - C.ref_start[k]);
    }
    // the longest span: one atomic per wave (millions of atomics on one word were 0.7 ms of a 4 M-read file)
    for (int o = 32; o > 0; o >>= 1) { const uint32_t other = __shfl_xor(span, o, 64); span = other > span ? other : span; }
    if ((threadIdx.x & 63) == 0 && span) atomicMax(&flagw[2], span);
}

// reads sorted by (name hash, index): the previous read with the same query name
__global__ void k_link(const uint8_t *u, const uint64_t *rec_off, const uint32_t *kept_rec, const uint64_t *hs, const uint32_t *order, uint32_t n, int32_t *link) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint32_t i = order[j];
    const uint8_t *pi = u + rec_off[kept_rec[i]] + 4;
    const uint32_t li = pi[8] ? pi[8] - 1u : 0u;
    int32_t found = -1;
    for (int64_t t = (int64_t)j - 1; t >= 0 && hs[t] == hs[j]; t--) {
        const uint32_t c = order[t];
        const uint8_t *pc = u + rec_off[kept_rec[c]] + 4;
        const uint32_t lc = pc[8] ? pc[8] - 1u : 0u;
        if (lc != li) continue;
        bool same = true;
        for (uint32_t x = 0; x < li && same; x++) same = pi[32 + x] == pc[32 + x];
        if (same) { found = (int32_t)c; break; }
    }
    link[i] = found;
}

__global__ void k_iota(uint32_t *v, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = i;
}

// record index: the first record that starts in each 32 KiB granule of the inflated stream (mfk_bamio.cpp write_index)
__global__ void k_marks(const uint8_t *u, const uint64_t *rec_off, uint32_t n_rec, uint64_t granule, uint64_t *mark_u, int32_t *mark_rp) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rec) return;
    const uint64_t q = rec_off[i], g = q / granule;
    if (i && rec_off[i - 1] / granule == g) return;
    mark_u[g] = q;
    mark_rp[2 * g] = (int32_t)d_le32(u + q + 4);
    mark_rp[2 * g + 1] = (int32_t)d_le32(u + q + 8);
}

DecodeState *state_of(mfk_ctx *ctx) {
    if (!ctx->dec) ctx->dec = new (std::nothrow) DecodeState();
    return static_cast<DecodeState *>(ctx->dec);
}

bool ensure(uint8_t **p, size_t *cap, size_t need) {
    if (need <= *cap && *p) return true;
    if (*p) (void)hipFree(*p);
    *p = nullptr; *cap = 0;
    const size_t want = need + need / 8 + 4096;
    if (hipMalloc((void **)p, want) != hipSuccess) { (void)hipGetLastError(); return false; }
    *cap = want;
    return true;
}

template <typename T> T *carve(char **at, size_t n) {
    T *p = reinterpret_cast<T *>(*at);
    *at += ((n ? n : 1) * sizeof(T) + 255) & ~(size_t)255;
    return p;
}
inline size_t room(size_t n, size_t sz) { return ((n ? n : 1) * sz + 255) & ~(size_t)255; }

}   // namespace

void mfk_decode_release(mfk_ctx *ctx) {
    DecodeState *S = static_cast<DecodeState *>(ctx->dec);
    if (!S) return;
    if (S->u) (void)hipFree(S->u);
    if (S->work) (void)hipFree(S->work);
    delete S;
    ctx->dec = nullptr;
}

extern "C" {

// BGZF block table of a file in host memory (SAM specification 4.1).  blocks == NULL: only counts.  Empty blocks (the EOF
// marker) are left out.  file_off: NULL or the file offset of every kept block (for the record index).  Returns 0, or 1 with
// mfk_last_error for bytes that are not BGZF.
int mfk_bgzf_block_table(mfk_ctx *ctx, const void *file, int64_t size, mfk_bgzf_block *blocks, uint64_t *file_off, uint32_t *crc, int64_t cap, int64_t *n_blocks, int64_t *inflated_bytes) {
    if (!ctx || size < 0 || (size > 0 && !file) || !n_blocks || !inflated_bytes) return 1;
    const uint8_t *f = static_cast<const uint8_t *>(file);
    int64_t n = 0;
    uint64_t u = 0;
    for (uint64_t o = 0; o < (uint64_t)size;) {
        if (o + 18 > (uint64_t)size) { snprintf(ctx->err, sizeof ctx->err, "truncated BGZF block header"); return 1; }
        const uint8_t *h = f + o;
        if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) { snprintf(ctx->err, sizeof ctx->err, "not a BGZF (BAM) file"); return 1; }
        const uint64_t xlen = (uint64_t)h[10] | ((uint64_t)h[11] << 8);
        if (o + 12 + xlen > (uint64_t)size) { snprintf(ctx->err, sizeof ctx->err, "truncated BGZF block header"); return 1; }
        uint64_t x = 12, bsize = 0;
        while (x + 4 <= 12 + xlen) {
            const uint64_t slen = (uint64_t)h[x + 2] | ((uint64_t)h[x + 3] << 8);
            if (h[x] == 66 && h[x + 1] == 67 && slen == 2 && x + 6 <= 12 + xlen) bsize = ((uint64_t)h[x + 4] | ((uint64_t)h[x + 5] << 8)) + 1;
            x += 4 + slen;
        }
        if (!bsize || bsize < 12 + xlen + 8 || o + bsize > (uint64_t)size) { snprintf(ctx->err, sizeof ctx->err, "bad BGZF block size"); return 1; }
        const uint8_t *t = h + bsize - 4;
        const uint32_t isize = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
        if (isize > 65536) { snprintf(ctx->err, sizeof ctx->err, "BGZF block larger than 64 KiB"); return 1; }
        if (isize) {
            if (blocks && n < cap) {
                blocks[n].comp_off = o + 12 + xlen; blocks[n].out_off = u; blocks[n].comp_len = (uint32_t)(bsize - 12 - xlen - 8); blocks[n].out_len = isize;
                if (file_off) file_off[n] = o;
                if (crc) { const uint8_t *cp = h + bsize - 8; crc[n] = (uint32_t)cp[0] | ((uint32_t)cp[1] << 8) | ((uint32_t)cp[2] << 16) | ((uint32_t)cp[3] << 24); }
            }
            n++; u += isize;
        }
        o += bsize;
    }
    *n_blocks = n; *inflated_bytes = (int64_t)u;
    return 0;
}

// Stage 1 of the device decode.  d_comp: the whole BGZF file in device memory; blocks: its block table (host);
// first_record_off: offset in the inflated stream of the first alignment record (the caller parses the header: it needs the
// contig names anyway); pool_mode 0 none / 1 clipped / 2 all (MFKIO_POOL_*); exclude_flags as in mfkio_pack_bam_filtered.
// sizes[8] <- n_reads, n_cx, n_blk, n_pool, pool bytes (quality bytes; base bytes are half), max_span (filled by the fetch),
// n_records (all alignment records), inflated bytes.  Errors: 13 = the record boundaries could not be verified (use the host
// reader), 14 = a record the packer rejects (sizes[0] = its index among all records, sizes[1] = the reason, PE_* above),
// 15 = damaged stream (bad or truncated record).
int mfk_decode_bam(mfk_ctx *ctx, const void *d_comp, int64_t comp_bytes, const mfk_bgzf_block *blocks, const uint32_t *block_crc, int64_t n_blocks, int64_t first_record_off,
                   int32_t n_contigs, int32_t pool_mode, uint32_t exclude_flags, int64_t *sizes) {
    if (!ctx || !sizes || n_blocks < 0 || n_blocks > 0x7FFFFFF0 || first_record_off < 0 || n_contigs <= 0 || n_contigs > 65535 || pool_mode < 0 || pool_mode > 2) return 1;
    ctx->err[0] = 0;
    DecodeState *S = state_of(ctx);
    if (!S) return 1;
    S->ready = false;
    HIPCHK(hipSetDevice(ctx->device));
    uint64_t U = 0;
    for (int64_t k = 0; k < n_blocks; k++) {
        if (blocks[k].out_off != U) { snprintf(ctx->err, sizeof ctx->err, "mfk_decode_bam: block table is not the running sum of the block sizes"); return 1; }
        U += blocks[k].out_len;
    }
    if ((uint64_t)first_record_off > U) { snprintf(ctx->err, sizeof ctx->err, "mfk_decode_bam: first record behind the end of the stream"); return 1; }
    if (!ensure(&S->u, &S->u_cap, (size_t)U + 64)) { snprintf(ctx->err, sizeof ctx->err, "mfk_decode_bam: out of device memory for the inflated stream (%llu bytes)", (unsigned long long)U); return 1; }
    if (int rc = mfk_bgzf_inflate(ctx, d_comp, comp_bytes, blocks, n_blocks, S->u, (int64_t)U, nullptr, nullptr)) return rc;
    if (block_crc && n_blocks) {   // the gzip trailers' CRC-32 against the inflated bytes, as the host reader checks them
        arena_reset(ctx);
        hipStream_t cs = ctx->stream;
        mfk_bgzf_block *d_blocks = aalloc<mfk_bgzf_block>(ctx, (size_t)n_blocks);
        uint32_t *d_expect = aalloc<uint32_t>(ctx, (size_t)n_blocks), *d_bad = aalloc<uint32_t>(ctx, 64);
        if (!d_blocks || !d_expect || !d_bad) return 1;
        HIPCHK(hipMemcpyAsync(d_blocks, blocks, (size_t)n_blocks * sizeof(mfk_bgzf_block), hipMemcpyHostToDevice, cs));
        HIPCHK(hipMemcpyAsync(d_expect, block_crc, (size_t)n_blocks * 4, hipMemcpyHostToDevice, cs));
        HIPCHK(hipMemsetAsync(d_bad, 0, 4, cs));
        if (int rc = bgzf_crc32_launch(ctx, S->u, d_blocks, (uint32_t)n_blocks, d_expect, nullptr, d_bad, cs)) return rc;
        uint32_t bad = 0;
        HIPCHK(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, cs));
        HIPCHK(hipStreamSynchronize(cs));
        if (bad) { snprintf(ctx->err, sizeof ctx->err, "BGZF checksum mismatch (corrupt file): %u of %lld blocks", bad, (long long)n_blocks); return 12; }
    }
    S->U = U; S->first_off = (uint64_t)first_record_off; S->n_blocks = n_blocks; S->n_contigs = n_contigs; S->pool_mode = pool_mode; S->exclude_flags = exclude_flags;
    memset(sizes, 0, 8 * sizeof(int64_t));
    sizes[7] = (int64_t)U;
    hipStream_t st = ctx->stream;
    arena_reset(ctx);
    const DecParams P{U, n_contigs, pool_mode, exclude_flags};
    // ---- boundaries: per-block arrays live in the arena (small), the per-record ones in S->work (sized when n_rec is known)
    const uint32_t nb = (uint32_t)n_blocks;
    uint64_t *blk_uoff = aalloc<uint64_t>(ctx, nb), *first = aalloc<uint64_t>(ctx, nb), *land = aalloc<uint64_t>(ctx, nb);
    uint32_t *cnt = aalloc<uint32_t>(ctx, nb), *base = aalloc<uint32_t>(ctx, (size_t)nb + 1), *flagw = aalloc<uint32_t>(ctx, 64);
    unsigned long long *err = aalloc<unsigned long long>(ctx, 8);
    if (!blk_uoff || !first || !land || !cnt || !base || !flagw || !err) return 1;
    {
        std::vector<uint64_t> h(nb);
        for (uint32_t k = 0; k < nb; k++) h[k] = blocks[k].out_off;
        if (nb) HIPCHK(hipMemcpyAsync(blk_uoff, h.data(), (size_t)nb * 8, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemsetAsync(flagw, 0, 256, st));
        HIPCHK(hipMemsetAsync(err, 0xFF, 64, st));
        HIPCHK(hipStreamSynchronize(st));   // (h goes out of scope)
    }
    uint32_t n_rec = 0;
    if (nb) {
        PROF_LAUNCH(k_first_record, dim3((nb + 3) / 4), dim3(256), 0, st, (const uint8_t *)S->u, P, (const uint64_t *)blk_uoff, nb, S->first_off, first);
        KCHK();
        PROF_LAUNCH(k_walk<false>, dim3((nb + 63) / 64), dim3(64), 0, st, (const uint8_t *)S->u, P, (const uint64_t *)blk_uoff, nb, (const uint64_t *)first, cnt, land, (const uint32_t *)nullptr, (uint64_t *)nullptr, flagw);
        KCHK();
        if (int rc = dev_excl_scan32(ctx, cnt, base, nb)) return rc;
        HIPCHK(hipMemcpyAsync(&n_rec, base + nb, 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
    }
    S->n_rec = n_rec;
    // ---- per-record arrays
    const size_t R = n_rec;
    const size_t need = room(R, 8) /*rec_off*/ + 10 * room(R + 1, 4) + room(R + 1, 8) /*poff*/ + 4096;
    if (!ensure(reinterpret_cast<uint8_t **>(&S->work), &S->work_cap, need)) { snprintf(ctx->err, sizeof ctx->err, "mfk_decode_bam: out of device memory for %zu records", R); return 1; }
    char *at = S->work;
    S->rec_off = carve<uint64_t>(&at, R);
    S->keep = carve<uint32_t>(&at, R + 1); S->kidx = carve<uint32_t>(&at, R + 1); S->nblk = carve<uint32_t>(&at, R + 1); S->boff = carve<uint32_t>(&at, R + 1);
    S->plen = carve<uint32_t>(&at, R + 1); S->cxf = carve<uint32_t>(&at, R + 1); S->cxidx = carve<uint32_t>(&at, R + 1); S->plf = carve<uint32_t>(&at, R + 1);
    S->pidx = carve<uint32_t>(&at, R + 1); S->base = carve<uint32_t>(&at, R + 1) /* spare */;
    S->poff = carve<uint64_t>(&at, R + 1);
    uint32_t h_flag[4] = {0, 0, 0, 0};
    if (nb) {
        PROF_LAUNCH(k_walk<true>, dim3((nb + 63) / 64), dim3(64), 0, st, (const uint8_t *)S->u, P, (const uint64_t *)blk_uoff, nb, (const uint64_t *)first, cnt, land, (const uint32_t *)base, S->rec_off, flagw);
        KCHK();
        HIPCHK(hipMemcpyAsync(h_flag, flagw, 16, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
    }
    if (h_flag[1] & 1) { snprintf(ctx->err, sizeof ctx->err, "bad alignment record size (corrupt file)"); return 15; }
    if (h_flag[1] & 2) { snprintf(ctx->err, sizeof ctx->err, "truncated alignment record at the end of the file"); return 15; }
    if (h_flag[0]) { snprintf(ctx->err, sizeof ctx->err, "mfk_decode_bam: a guessed record boundary did not verify (use the host reader for this file)"); return 13; }
    sizes[6] = n_rec;
    if (n_rec) {
        PROF_LAUNCH(k_parse_count, dim3((n_rec + 255) / 256), dim3(256), 0, st, (const uint8_t *)S->u, P, (const uint64_t *)S->rec_off, n_rec, S->keep, S->nblk, S->plen, S->cxf, S->plf, err);
        KCHK();
        if (int rc = dev_excl_scan32(ctx, S->keep, S->kidx, n_rec)) return rc;
        if (int rc = dev_excl_scan32(ctx, S->nblk, S->boff, n_rec)) return rc;
        if (int rc = dev_excl_scan32(ctx, S->cxf, S->cxidx, n_rec)) return rc;
        if (int rc = dev_excl_scan32(ctx, S->plf, S->pidx, n_rec)) return rc;
        if (int rc = dev_excl_scan64(ctx, S->plen, S->poff, n_rec)) return rc;
        uint32_t t32[4] = {0, 0, 0, 0};
        uint64_t t64 = 0;
        unsigned long long h_err = ~0ull;
        HIPCHK(hipMemcpyAsync(&t32[0], S->kidx + n_rec, 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(&t32[1], S->boff + n_rec, 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(&t32[2], S->cxidx + n_rec, 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(&t32[3], S->pidx + n_rec, 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(&t64, S->poff + n_rec, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(&h_err, err, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (h_err != ~0ull) {
            sizes[0] = (int64_t)(h_err >> 8); sizes[1] = (int64_t)(h_err & 0xFF);
            snprintf(ctx->err, sizeof ctx->err, "mfk_decode_bam: record %lld is rejected by the packer (reason %d)", (long long)sizes[0], (int)sizes[1]);
            return 14;
        }
        S->n_reads = t32[0]; S->n_blk = t32[1]; S->n_cx = t32[2]; S->n_pool = t32[3]; S->pool_bytes = t64;
    } else { S->n_reads = S->n_blk = S->n_cx = S->n_pool = 0; S->pool_bytes = 0; }
    sizes[0] = S->n_reads; sizes[1] = S->n_cx; sizes[2] = S->n_blk; sizes[3] = S->n_pool; sizes[4] = (int64_t)S->pool_bytes;
    S->ready = true;
    ctx->call_ok = true;
    return 0;
}

// Stage 2: the caller has allocated the columns (device) from the sizes of stage 1.  dst: device pointers for the nine columns
// (n_reads), cx_read (n_cx), cx_off (n_cx + 1), blk_* (n_blk), pool_read (n_pool), pool_off (n_pool + 1), pool_bases
// (pool bytes / 2), pool_quals (pool bytes); its h_ fields are ignored.  h_contig_read_start: HOST, n_contigs + 1.
// marks: NULL, or HOST room for ceil(inflated bytes / granule) marks of the record index (u64 stream offset, -1 = no record
// starts in that granule) and mark_ref_pos (two i32 each).  *max_span <- the longest reference span.
int mfk_decode_bam_fetch(mfk_ctx *ctx, const mfk_reads *dst, int64_t *h_contig_read_start, int32_t *max_span, int64_t granule, uint64_t *marks, int32_t *mark_ref_pos) {
    if (!ctx || !dst || !h_contig_read_start || !max_span) return 1;
    DecodeState *S = static_cast<DecodeState *>(ctx->dec);
    if (!S || !S->ready) { snprintf(ctx->err, sizeof ctx->err, "mfk_decode_bam_fetch without a successful mfk_decode_bam"); return 1; }
    ctx->err[0] = 0;
    HIPCHK(hipSetDevice(ctx->device));
    arena_reset(ctx);
    hipStream_t st = ctx->stream;
    const uint32_t n_rec = (uint32_t)S->n_rec, n = (uint32_t)S->n_reads;
    const int32_t nc = S->n_contigs;
    const DecParams P{S->U, nc, S->pool_mode, S->exclude_flags};
    *max_span = 0;
    for (int32_t c = 0; c <= nc; c++) h_contig_read_start[c] = 0;
    if (n && (!dst->ref_start || !dst->ref_end || !dst->link || !dst->contig_id || !dst->qlen || !dst->first_op_len || !dst->last_op_len || !dst->mapq || !dst->flags)) { snprintf(ctx->err, sizeof ctx->err, "null column pointer"); return 1; }
    if ((S->n_cx && (!dst->cx_read || !dst->cx_off)) || (S->n_blk && (!dst->blk_start || !dst->blk_end || !dst->blk_ins)) || (S->n_pool && (!dst->pool_read || !dst->pool_off || !dst->pool_bases || !dst->pool_quals)) || !dst->cx_off || !dst->pool_off) { snprintf(ctx->err, sizeof ctx->err, "null side-table pointer"); return 1; }
    DevCols C;
    C.ref_start = const_cast<int32_t *>(dst->ref_start); C.ref_end = const_cast<int32_t *>(dst->ref_end); C.link = const_cast<int32_t *>(dst->link);
    C.contig_id = const_cast<uint16_t *>(dst->contig_id); C.qlen = const_cast<uint16_t *>(dst->qlen); C.first_op_len = const_cast<uint16_t *>(dst->first_op_len); C.last_op_len = const_cast<uint16_t *>(dst->last_op_len);
    C.mapq = const_cast<uint8_t *>(dst->mapq); C.flags = const_cast<uint8_t *>(dst->flags);
    C.cx_read = const_cast<uint32_t *>(dst->cx_read); C.cx_off = const_cast<uint32_t *>(dst->cx_off);
    C.blk_start = const_cast<int32_t *>(dst->blk_start); C.blk_end = const_cast<int32_t *>(dst->blk_end); C.blk_ins = const_cast<int32_t *>(dst->blk_ins);
    C.pool_read = const_cast<uint32_t *>(dst->pool_read); C.pool_off = const_cast<uint64_t *>(dst->pool_off);
    C.pool_bases = const_cast<uint8_t *>(dst->pool_bases); C.pool_quals = const_cast<uint8_t *>(dst->pool_quals);
    uint64_t *hash = aalloc<uint64_t>(ctx, n), *hash2 = aalloc<uint64_t>(ctx, n);
    uint32_t *kept_rec = aalloc<uint32_t>(ctx, n), *ord = aalloc<uint32_t>(ctx, n), *ord2 = aalloc<uint32_t>(ctx, n), *flagw = aalloc<uint32_t>(ctx, 64);
    int64_t *keys = aalloc<int64_t>(ctx, n), *crs = aalloc<int64_t>(ctx, (size_t)nc + 1);
    unsigned long long *err = aalloc<unsigned long long>(ctx, 8);
    if (!hash || !hash2 || !kept_rec || !ord || !ord2 || !flagw || !keys || !crs || !err) return 1;
    HIPCHK(hipMemsetAsync(flagw, 0, 256, st));
    HIPCHK(hipMemsetAsync(err, 0xFF, 64, st));
    HIPCHK(hipMemsetAsync(crs, 0, ((size_t)nc + 1) * 8, st));
    // the closing entries of the two offset tables
    {
        const uint32_t nblk_total = (uint32_t)S->n_blk;
        HIPCHK(hipMemcpyAsync(C.cx_off + S->n_cx, &nblk_total, 4, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(C.pool_off + S->n_pool, &S->pool_bytes, 8, hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st));
    }
    if (n_rec && n) {
        PROF_LAUNCH(k_parse_fill, dim3((n_rec + 255) / 256), dim3(256), 0, st, (const uint8_t *)S->u, P, (const uint64_t *)S->rec_off, n_rec, (const uint32_t *)S->keep, (const uint32_t *)S->kidx,
                    (const uint32_t *)S->boff, (const uint32_t *)S->cxidx, (const uint32_t *)S->pidx, (const uint64_t *)S->poff, C, hash, kept_rec, keys);
        KCHK();
        PROF_LAUNCH(k_finish, dim3((n + 255) / 256), dim3(256), 0, st, (const int64_t *)keys, C, n, nc, crs, flagw, err, (const uint32_t *)kept_rec);
        KCHK();
        // links: stable sort of the read indices by name hash
        PROF_LAUNCH(k_iota, dim3((n + 255) / 256), dim3(256), 0, st, ord, n);
        KCHK();
        if (int rc = dev_sort_pairs64(ctx, hash, hash2, ord, ord2, n, 0, 64)) return rc;
        PROF_LAUNCH(k_link, dim3((n + 255) / 256), dim3(256), 0, st, (const uint8_t *)S->u, (const uint64_t *)S->rec_off, (const uint32_t *)kept_rec, (const uint64_t *)hash2, (const uint32_t *)ord2, n, C.link);
        KCHK();
    }
    uint64_t *d_mark_u = nullptr; int32_t *d_mark_rp = nullptr;
    size_t n_marks = 0;
    if (marks && mark_ref_pos && granule > 0) {
        n_marks = (size_t)((S->U + (uint64_t)granule - 1) / (uint64_t)granule);
        d_mark_u = aalloc<uint64_t>(ctx, n_marks); d_mark_rp = aalloc<int32_t>(ctx, 2 * n_marks);
        if (!d_mark_u || !d_mark_rp) return 1;
        HIPCHK(hipMemsetAsync(d_mark_u, 0xFF, (n_marks ? n_marks : 1) * 8, st));
        if (n_rec) { PROF_LAUNCH(k_marks, dim3((n_rec + 255) / 256), dim3(256), 0, st, (const uint8_t *)S->u, (const uint64_t *)S->rec_off, n_rec, (uint64_t)granule, d_mark_u, d_mark_rp); KCHK(); }
        if (n_marks) {
            HIPCHK(hipMemcpyAsync(marks, d_mark_u, n_marks * 8, hipMemcpyDeviceToHost, st));
            HIPCHK(hipMemcpyAsync(mark_ref_pos, d_mark_rp, n_marks * 8, hipMemcpyDeviceToHost, st));
        }
    }
    uint32_t h_flag[4] = {0, 0, 0, 0};
    unsigned long long h_err = ~0ull;
    HIPCHK(hipMemcpyAsync(h_flag, flagw, 16, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&h_err, err, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(h_contig_read_start, crs, ((size_t)nc + 1) * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    S->ready = false;
    ctx->call_ok = true;
    if (h_err != ~0ull) { snprintf(ctx->err, sizeof ctx->err, "alignments are not coordinate-sorted (the reference needs a sorted, indexed BAM): record %lld", (long long)(h_err >> 8)); return 14; }
    *max_span = (int32_t)h_flag[2];
    return 0;
}

}   // extern "C"
