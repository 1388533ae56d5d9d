// libmfkio.so -- native BAM ingestion for the flank-discovery path (C-ABI in include/mfk_bamio.h).
//
// BGZF blocks are independent deflate streams, so a pass inflates a few thousand of them in parallel into one buffer,
// walks the record lengths once (sequential, one add per record), parses the records in parallel into per-thread
// column vectors and appends those to the pack in file order.  Records that straddle two passes are carried over.
// `link` (previous read with the same query name) is computed at the end: reads are partitioned by name hash, each
// partition is walked in file order by one thread with its own open-addressing table, names are compared byte for byte
// on a hash hit, so the result is exact.
//
// The per-read rules are those of mustache_amd/pack.py::pack_reads (which restates the reference's sctools.py:5-31,
// 99-109 and pysam's get_blocks); tests compare the two array for array.
#include "../../include/mfk_bamio.h"

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

inline uint32_t le32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
inline uint16_t le16(const uint8_t *p) { uint16_t v; memcpy(&v, p, 2); return v; }
inline double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

template <class F> void parallel(int T, F f) {
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back(f, t);
    f(0);
    for (auto &x : th) x.join();
}

// libdeflate is in the image as a runtime library only (no header): its four public entry points are bound by name when
// the library is present, zlib is used otherwise.
struct LibDeflate {
    void *handle = nullptr;
    void *(*alloc)() = nullptr;
    int (*decompress)(void *, const void *, size_t, void *, size_t, size_t *) = nullptr;
    void (*release)(void *) = nullptr;
    uint32_t (*crc)(uint32_t, const void *, size_t) = nullptr;
    bool ok() const { return handle != nullptr; }
};

const LibDeflate &libdeflate() {
    static LibDeflate ld = [] {
        LibDeflate d;
        const char *off = getenv("MFK_NO_LIBDEFLATE");
        if (off && *off && *off != '0') return d;
        void *h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) return d;
        d.alloc = (void *(*)())dlsym(h, "libdeflate_alloc_decompressor");
        d.decompress = (int (*)(void *, const void *, size_t, void *, size_t, size_t *))dlsym(h, "libdeflate_deflate_decompress");
        d.release = (void (*)(void *))dlsym(h, "libdeflate_free_decompressor");
        d.crc = (uint32_t(*)(uint32_t, const void *, size_t))dlsym(h, "libdeflate_crc32");
        if (d.alloc && d.decompress && d.release && d.crc) d.handle = h;
        else dlclose(h);
        return d;
    }();
    return ld;
}

struct Block {
    uint64_t payload;  // file offset of the deflate stream
    uint32_t clen;     // its length
    uint32_t ulen;     // ISIZE
    uint32_t crc;
};

// One BGZF block header (SAM spec 4.1: a gzip member with a 'BC' extra subfield holding the block size) at file offset o.
// Returns nullptr and fills *b / *bsize, or the reason the bytes are not a block.
const char *block_header(const uint8_t *file, size_t fsize, size_t o, const char *path, Block *b, size_t *bsize_out) {
    static thread_local std::string msg;
    if (o + 18 > fsize) return "truncated BGZF block header";
    const uint8_t *h = file + o;
    if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) { msg = std::string(path) + " is not a BGZF (BAM) file"; return msg.c_str(); }
    size_t xlen = le16(h + 10), x = 12, bsize = 0;
    if (o + 12 + xlen > fsize) return "truncated BGZF block header";
    while (x + 4 <= 12 + xlen) {
        size_t slen = le16(h + x + 2);
        if (h[x] == 66 && h[x + 1] == 67 && slen == 2 && x + 6 <= 12 + xlen) bsize = (size_t)le16(h + x + 4) + 1;
        x += 4 + slen;
    }
    if (!bsize || bsize < 12 + xlen + 8 || o + bsize > fsize) return "bad BGZF block size";
    b->payload = o + 12 + xlen;
    b->clen = (uint32_t)(bsize - 12 - xlen - 8);
    b->crc = le32(h + bsize - 8);
    b->ulen = le32(h + bsize - 4);
    if (b->ulen > 65536) return "BGZF block larger than 64 KiB";
    *bsize_out = bsize;
    return nullptr;
}

// disjoint sorted intervals + the (never seen in practice) inverted ones, kept so the test equals the Python predicate
struct IvSet {
    std::vector<int32_t> lo, hi;
    std::vector<std::pair<int32_t, int32_t>> odd;
    void build(std::vector<std::pair<int32_t, int32_t>> &v) {
        std::vector<std::pair<int32_t, int32_t>> good;
        for (auto &iv : v) (iv.first < iv.second ? good : odd).push_back(iv);
        std::sort(good.begin(), good.end());
        for (auto &iv : good) {
            if (!lo.empty() && iv.first <= hi.back()) hi.back() = std::max(hi.back(), iv.second);
            else { lo.push_back(iv.first); hi.push_back(iv.second); }
        }
    }
    // any interval with s < hi && e > lo
    bool hit(int32_t s, int32_t e) const {
        size_t k = std::upper_bound(hi.begin(), hi.end(), s) - hi.begin();
        if (k < hi.size() && lo[k] < e) return true;
        for (auto &iv : odd) if (s < iv.second && e > iv.first) return true;
        return false;
    }
};

struct Shared {
    int pool_mode = 0;
    uint32_t exclude_flags = 0;   // records with any of these FLAG bits are skipped like `samtools view -F` (formatbam: 0x900)
    int32_t n_contigs = 0;
    std::vector<IvSet> near_, win_;
};

// Record index: one mark per IDX_GRANULE bytes of the uncompressed stream = the offset of a record that starts there and
// that record's (reference id, position).  A coordinate-sorted file read from a mark onwards only holds records at or
// behind the mark's coordinate, so the reads over a region are found by inflating the few BGZF blocks between two marks
// (pairflanks: the reference does the same through the BAI and bam.fetch, pairflanks.py:393-409).
#define IDX_GRANULE ((uint64_t)32 << 10)
struct IdxEntry { uint64_t u; int32_t ref, pos; };

// what one thread produced from its share of one pass
struct Local {
    std::vector<int32_t> ref_start, ref_end;
    std::vector<uint16_t> contig_id, qlen, first_op, last_op;
    std::vector<uint8_t> mapq, flags, name_len;
    std::vector<uint64_t> hash;
    std::vector<uint32_t> name_off;
    std::vector<char> names;
    std::vector<uint32_t> cx_read, cx_nblk, pool_read, pool_plen;
    std::vector<int32_t> blk_start, blk_end, blk_ins;
    std::vector<uint8_t> pool_bases, pool_quals;
    int32_t max_span = 0;
    int64_t first_key = -1, last_key = -1;
    std::vector<IdxEntry> idx;   // block-index marks of this share (see "record index" below)
    uint64_t next_mark = 0;
    std::string err;
    bool format_err = false;  // err is about the file's bytes (MFKIO_ERR_FORMAT), not a record the packer rejects
    void clear() {
        ref_start.clear(); ref_end.clear(); contig_id.clear(); qlen.clear(); first_op.clear(); last_op.clear();
        mapq.clear(); flags.clear(); name_len.clear(); hash.clear(); name_off.clear(); names.clear();
        cx_read.clear(); cx_nblk.clear(); pool_read.clear(); pool_plen.clear();
        blk_start.clear(); blk_end.clear(); blk_ins.clear(); pool_bases.clear(); pool_quals.clear();
        max_span = 0; first_key = last_key = -1; err.clear(); format_err = false;
        idx.clear(); next_mark = 0;
    }
};

enum { CM = 0, CI = 1, CD = 2, CN = 3, CS = 4, CH = 5, CP = 6, CEQ = 7, CX = 8 };
inline bool aligned_op(unsigned op) { return op == CM || op == CEQ || op == CX; }
inline bool ref_op(unsigned op) { return op == CM || op == CD || op == CN || op == CEQ || op == CX; }

std::string describe(const uint8_t *p, int32_t pos) {
    unsigned l_name = p[8];
    std::string nm((const char *)p + 32, l_name ? l_name - 1 : 0);
    char b[64];
    snprintf(b, sizeof b, " (read %s at reference position %d)", nm.c_str(), pos);
    return b;
}

// MD tag of a record: pointer + length, or nullptr.  Follows the tag walk of mustache_amd/bamio.py::_find_md.
bool find_md(const uint8_t *t, size_t n, const uint8_t **md, size_t *md_len, std::string &err) {
    size_t o = 0;
    *md = nullptr;
    while (o + 3 <= n) {
        const uint8_t *tag = t + o;
        uint8_t typ = t[o + 2];
        o += 3;
        switch (typ) {
        case 'A': case 'c': case 'C': o += 1; break;
        case 's': case 'S': o += 2; break;
        case 'i': case 'I': case 'f': o += 4; break;
        case 'Z': case 'H': {
            const void *e = o < n ? memchr(t + o, 0, n - o) : nullptr;
            if (!e) { err = "unterminated string tag in an alignment record"; return false; }
            size_t len = (const uint8_t *)e - (t + o);
            if (tag[0] == 'M' && tag[1] == 'D') { *md = t + o; *md_len = len; return true; }
            o += len + 1;
            break;
        }
        case 'B': {
            if (o + 5 > n) { err = "truncated array tag in an alignment record"; return false; }
            uint8_t sub = t[o];
            size_t w = (sub == 'c' || sub == 'C' || sub == 'A') ? 1 : (sub == 's' || sub == 'S') ? 2
                       : (sub == 'i' || sub == 'I' || sub == 'f') ? 4 : 0;
            if (!w) { err = "bad BAM array tag type"; return false; }
            o += 5 + (size_t)le32(t + o + 1) * w;
            break;
        }
        default:
            err = "bad BAM tag type";
            return false;
        }
    }
    return true;
}

inline uint64_t hash_name(const uint8_t *s, size_t n) {
    uint64_t h = 0xcbf29ce484222325ull ^ (n * 0x9e3779b97f4a7c15ull);
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t w; memcpy(&w, s + i, 8);
        h = (h ^ w) * 0x9fb21c651e98df25ull;
        h ^= h >> 29;
    }
    uint64_t w = 0;
    memcpy(&w, s + i, n - i);
    h = (h ^ w) * 0x9fb21c651e98df25ull;
    h ^= h >> 32;
    h *= 0xd6e8feb86659fd93ull;
    h ^= h >> 29;
    return h;
}

// One record (p = first byte after block_size).  Returns false with L.err set when the packer rejects it.
bool parse_record(const Shared &sh, const uint8_t *p, uint32_t bs, Local &L) {
    int32_t ref_id = (int32_t)le32(p), pos = (int32_t)le32(p + 4);
    unsigned l_name = p[8], mapq = p[9], n_cig = le16(p + 12), flag = le16(p + 14);
    int32_t l_seq = (int32_t)le32(p + 16);
    if (l_seq < 0) { L.err = "negative sequence length in an alignment record"; L.format_err = true; return false; }
    size_t fixed = 32 + (size_t)l_name + 4 * (size_t)n_cig + ((size_t)l_seq + 1) / 2 + (size_t)l_seq;
    if (fixed > bs) { L.err = "truncated alignment record"; L.format_err = true; return false; }
    if (n_cig == 0 || (flag & 4) || ref_id < 0) return true;  // unmapped: never reaches the CIGAR (quirk Q17)
    if (flag & sh.exclude_flags) return true;                 // e.g. secondary / supplementary alignments (samtools.py:9-18)
    if (ref_id >= sh.n_contigs) { L.err = "alignment on a contig that is not in the header"; return false; }
    const uint8_t *cig = p + 32 + l_name;
    auto op_of = [&](unsigned k) { return le32(cig + 4 * k) & 15u; };
    auto len_of = [&](unsigned k) { return le32(cig + 4 * k) >> 4; };
    int64_t rend64 = pos;
    for (unsigned k = 0; k < n_cig; ++k) if (ref_op(op_of(k))) rend64 += len_of(k);
    if (rend64 > INT32_MAX) { L.err = "alignment end beyond 2^31" + describe(p, pos); return false; }
    int32_t rend = (int32_t)rend64;
    if (sh.pool_mode == MFKIO_POOL_REGIONS && !sh.near_[ref_id].hit(pos, rend)) return true;
    int64_t key = ((int64_t)ref_id << 32) + (int64_t)(uint32_t)pos;
    if (key < L.last_key) {
        L.err = "alignments are not coordinate-sorted (the reference needs a sorted, indexed BAM)";
        return false;
    }
    if (L.first_key < 0) L.first_key = key;
    L.last_key = key;
    // lenient clip kinds, sctools.py:17-31
    const uint8_t *md = nullptr;
    size_t md_len = 0;
    if (!find_md(p + fixed, bs - fixed, &md, &md_len, L.err)) { L.format_err = true; return false; }
    unsigned first = op_of(0), last = op_of(n_cig - 1), lk, rk;
    if (first == CS) lk = 1;
    else {
        if (!md || !md_len) {
            L.err = "read without MD tag: the reference's lenient soft-clip test needs it (sctools.py:28)" + describe(p, pos);
            return false;
        }
        lk = md[0] == '0' ? 2 : 0;
    }
    if (last == CS) rk = 1;
    else {
        if (!md || !md_len) {
            L.err = "read without MD tag: the reference's lenient soft-clip test needs it (sctools.py:20)" + describe(p, pos);
            return false;
        }
        rk = (md[md_len - 1] == '0' && md_len >= 2 && !(md[md_len - 2] >= '0' && md[md_len - 2] <= '9')) ? 2 : 0;
    }
    // the alignment core between the clips
    unsigned lo = 0, hi = n_cig;
    while (lo < hi && (op_of(lo) == CH || op_of(lo) == CS)) ++lo;
    while (hi > lo && (op_of(hi - 1) == CH || op_of(hi - 1) == CS)) --hi;
    unsigned n_core = 0, first_ref = 99, last_ref = 99;
    for (unsigned k = lo; k < hi; ++k) {
        unsigned op = op_of(k);
        if (op == CP) continue;
        ++n_core;
        if (op == CH || op == CS) { L.err = "clip operation inside an alignment:" + describe(p, pos); return false; }
        if (op != CI) { if (first_ref == 99) first_ref = op; last_ref = op; }
    }
    if (!n_core) { L.err = "alignment without aligned bases:" + describe(p, pos); return false; }
    if (first_ref == 99 || first_ref == CD || first_ref == CN || last_ref == CD || last_ref == CN) {
        L.err = "alignment starts or ends with a deletion/skip: unsupported" + describe(p, pos);
        return false;
    }
    bool simple = n_core == 1 && !(lo > 0 && first == CH && op_of(lo - 1) == CS);
    if (l_seq > 65535) { L.err = "query longer than 65535 bases" + describe(p, pos); return false; }
    if (rend - pos > L.max_span) L.max_span = rend - pos;
    unsigned flags = lk | (rk << 2);
    uint32_t idx = (uint32_t)L.ref_start.size();
    if (!simple) {
        flags |= 16;
        L.cx_read.push_back(idx);
        int32_t rp = pos, pending = 0;
        uint32_t nb = 0;
        for (unsigned k = 0; k < n_cig; ++k) {
            unsigned op = op_of(k);
            int32_t l = (int32_t)len_of(k);
            if (aligned_op(op)) {
                L.blk_start.push_back(rp); L.blk_end.push_back(rp + l); L.blk_ins.push_back(pending);
                pending = 0; rp += l; ++nb;
            } else if (op == CI || (op == CS && k > 0 && nb == 0)) pending += l;
            else if (op == CD || op == CN) rp += l;
        }
        L.cx_nblk.push_back(nb);
    }
    bool want = sh.pool_mode == MFKIO_POOL_ALL || (sh.pool_mode == MFKIO_POOL_CLIPPED && (lk || rk)) ||
                (sh.pool_mode == MFKIO_POOL_REGIONS && sh.win_[ref_id].hit(pos, rend));
    if (want && l_seq) {
        flags |= 32;
        L.pool_read.push_back(idx);
        uint32_t plen = (uint32_t)l_seq + ((uint32_t)l_seq & 1u), nb = plen >> 1;
        const uint8_t *seq = cig + 4 * (size_t)n_cig, *qual = seq + nb;
        size_t b0 = L.pool_bases.size(), q0 = L.pool_quals.size();
        L.pool_bases.resize(b0 + nb);
        memcpy(L.pool_bases.data() + b0, seq, nb);
        if (l_seq & 1) L.pool_bases[b0 + nb - 1] &= 0xF0;
        L.pool_quals.resize(q0 + plen, 0);
        if (qual[0] != 0xFF) memcpy(L.pool_quals.data() + q0, qual, (size_t)l_seq);
        L.pool_plen.push_back(plen);
    }
    L.ref_start.push_back(pos);
    L.ref_end.push_back(rend);
    L.contig_id.push_back((uint16_t)ref_id);
    L.qlen.push_back((uint16_t)l_seq);
    L.first_op.push_back((uint16_t)std::min<uint32_t>(len_of(0), 65535));
    L.last_op.push_back((uint16_t)std::min<uint32_t>(len_of(n_cig - 1), 65535));
    L.mapq.push_back((uint8_t)mapq);
    L.flags.push_back((uint8_t)flags);
    unsigned nl = l_name ? l_name - 1 : 0;
    L.name_off.push_back((uint32_t)L.names.size());
    L.name_len.push_back((uint8_t)nl);
    L.names.insert(L.names.end(), (const char *)p + 32, (const char *)p + 32 + nl);
    L.hash.push_back(hash_name(p + 32, nl));
    return true;
}

// growable POD array without value-initialisation: its pages are first touched by the threads that fill them
template <class T> struct Buf {
    T *p = nullptr;
    size_t n = 0, cap = 0;
    Buf() = default;
    Buf(const Buf &) = delete;
    Buf &operator=(const Buf &) = delete;
    ~Buf() { free(p); }
    void grow(size_t nn) {
        if (nn > cap) {
            size_t nc = std::max(nn, cap + cap / 2 + 4096);
            T *q = (T *)realloc(p, nc * sizeof(T));
            if (!q) throw std::bad_alloc();
            p = q;
            cap = nc;
        }
        n = nn;
    }
    void release() { free(p); p = nullptr; n = cap = 0; }
    T *data() { return p; }
    size_t size() const { return n; }
};

template <class T> void put(Buf<T> &dst, size_t at, const std::vector<T> &src) {
    if (!src.empty()) memcpy(dst.p + at, src.data(), src.size() * sizeof(T));
}

}  // namespace

struct mfkio_pack {
    std::string error;
    mfkio_arrays a;
    Buf<int32_t> ref_start, ref_end, link, blk_start, blk_end, blk_ins;
    Buf<uint16_t> contig_id, qlen, first_op, last_op;
    Buf<uint8_t> mapq, flags, pool_bases, pool_quals, name_len;
    Buf<uint32_t> cx_read, cx_off, pool_read;
    Buf<uint64_t> pool_off, hash, name_off;
    Buf<char> names;
    std::vector<int32_t> contig_len;
    std::vector<int64_t> contig_read_start;
    std::vector<char> contig_names;
    int32_t max_span = 0;
    int64_t last_key = -1;
    std::vector<IdxEntry> index;   // marks in file order (full passes with an index file to write)
};

namespace {

int fail(mfkio_pack *P, int code, const std::string &msg) {
    P->error = msg;
    return code;
}

// BAM header at the front of buf: >0 bytes consumed, 0 = need more data, <0 = not a BAM
int64_t parse_header(mfkio_pack *P, const uint8_t *buf, size_t len) {
    if (len < 12) return 0;
    if (memcmp(buf, "BAM\1", 4) != 0) return -1;
    size_t o = 8 + (size_t)le32(buf + 4);
    if (o + 4 > len) return 0;
    uint32_t n_ref = le32(buf + o);
    o += 4;
    std::vector<int32_t> lens;
    std::vector<char> names;
    for (uint32_t i = 0; i < n_ref; ++i) {
        if (o + 4 > len) return 0;
        uint32_t l_name = le32(buf + o);
        o += 4;
        if (o + l_name + 4 > len) return 0;
        size_t nl = l_name ? l_name - 1 : 0;
        names.insert(names.end(), (const char *)buf + o, (const char *)buf + o + nl);
        names.push_back('\0');
        o += l_name;
        lens.push_back((int32_t)le32(buf + o));
        o += 4;
    }
    P->contig_len = lens;
    P->contig_names = names;
    return (int64_t)o;
}

void compute_links(mfkio_pack *P, int T) {
    const int64_t n = (int64_t)P->ref_start.size();
    P->link.grow((size_t)n);
    if (n == 0) return;
    const uint64_t *hash = P->hash.data();
    auto part_of = [&](uint64_t h) { return (int)(((h >> 40) * (uint64_t)T) >> 24); };
    // stable partition of the read indices by name hash
    std::vector<std::vector<int64_t>> cnt(T, std::vector<int64_t>(T + 1, 0));
    auto range = [&](int t, int64_t &b, int64_t &e) { b = n * t / T; e = n * (t + 1) / T; };
    parallel(T, [&](int t) {
        int64_t b, e; range(t, b, e);
        for (int64_t i = b; i < e; ++i) cnt[t][part_of(hash[i])]++;
    });
    std::vector<int64_t> part_start(T + 1, 0);
    for (int p = 0; p < T; ++p) {
        int64_t s = 0;
        for (int t = 0; t < T; ++t) s += cnt[t][p];
        part_start[p + 1] = part_start[p] + s;
    }
    std::vector<std::vector<int64_t>> wr(T, std::vector<int64_t>(T, 0));
    for (int p = 0; p < T; ++p) {
        int64_t s = part_start[p];
        for (int t = 0; t < T; ++t) { wr[t][p] = s; s += cnt[t][p]; }
    }
    Buf<int32_t> order;
    order.grow((size_t)n);
    parallel(T, [&](int t) {
        int64_t b, e; range(t, b, e);
        std::vector<int64_t> w = wr[t];
        for (int64_t i = b; i < e; ++i) order.p[(size_t)w[part_of(hash[i])]++] = (int32_t)i;
    });
    const char *names = P->names.data();
    const uint64_t *noff = P->name_off.data();
    const uint8_t *nlen = P->name_len.data();
    int32_t *link = P->link.data();
    struct Slot { uint64_t h; int32_t last; int32_t pad; };
    parallel(T, [&](int p) {
        int64_t b = part_start[p], e = part_start[p + 1];
        if (b == e) return;
        size_t cap = 16;
        while (cap < (size_t)(e - b) * 2) cap <<= 1;
        std::vector<Slot> tab(cap, Slot{0, -1, 0});
        size_t mask = cap - 1;
        for (int64_t k = b; k < e; ++k) {
            int32_t i = order.p[(size_t)k];
            uint64_t h = hash[i];
            size_t s = (size_t)(h * 0x9e3779b97f4a7c15ull >> 20) & mask;
            for (;;) {
                Slot &sl = tab[s];
                if (sl.last < 0) { sl.h = h; sl.last = i; link[i] = -1; break; }
                if (sl.h == h && nlen[sl.last] == nlen[i] &&
                    memcmp(names + noff[sl.last], names + noff[i], nlen[i]) == 0) {
                    link[i] = sl.last;
                    sl.last = i;
                    break;
                }
                s = (s + 1) & mask;
            }
        }
    });
}

// Does an alignment record plausibly start at q?  Only used to GUESS where a thread's share of a pass starts: every
// guess is verified afterwards (the previous share's walk must land exactly on it), so a wrong guess costs time only.
bool plausible(const Shared &sh, const uint8_t *buf, size_t q, size_t len, size_t *next) {
    if (q + 36 > len) return false;
    uint32_t bs = le32(buf + q);
    if (bs < 32 || bs > (1u << 24) || q + 4 + (size_t)bs > len) return false;
    const uint8_t *p = buf + q + 4;
    int32_t ref = (int32_t)le32(p), pos = (int32_t)le32(p + 4), mref = (int32_t)le32(p + 20), l_seq = (int32_t)le32(p + 16);
    if (ref < -1 || ref >= sh.n_contigs || mref < -1 || mref >= sh.n_contigs || pos < -1 || l_seq < 0) return false;
    unsigned l_name = p[8], n_cig = le16(p + 12);
    if (!l_name) return false;
    size_t fixed = 32 + (size_t)l_name + 4 * (size_t)n_cig + ((size_t)l_seq + 1) / 2 + (size_t)l_seq;
    if (fixed > bs || p[32 + l_name - 1] != 0) return false;
    for (unsigned i = 0; i + 1 < l_name; ++i) if (p[32 + i] < 33 || p[32 + i] > 126) return false;
    for (unsigned k = 0; k < n_cig; ++k) if ((le32(p + 32 + l_name + 4 * k) & 15u) > 8) return false;
    *next = q + 4 + (size_t)bs;
    return true;
}

bool chain_plausible(const Shared &sh, const uint8_t *buf, size_t q, size_t len) {
    for (int d = 0; d < 6; ++d) {
        size_t nx;
        if (!plausible(sh, buf, q, len, &nx)) {
            // running into the end of the pass after a few good records is as good as it gets
            return d >= 3 && (q + 4 > len || q + 4 + (size_t)le32(buf + q) > len);
        }
        q = nx;
    }
    return true;
}

// links, first read of every contig, and the array pointers of the finished pack.  Returns the time the link pass began.
double finish_pack(mfkio_pack *P, int n_contigs, int T, double *t_links_done) {
    double t2 = now_s();
    compute_links(P, T);
    *t_links_done = now_s();
    // first read of each contig
    P->contig_read_start.assign((size_t)n_contigs + 1, 0);
    for (int c = 0; c <= n_contigs; ++c)
        P->contig_read_start[(size_t)c] = std::lower_bound(P->contig_id.p, P->contig_id.p + P->contig_id.n, (uint32_t)c,
                                                            [](uint16_t a, uint32_t b) { return (uint32_t)a < b; }) - P->contig_id.p;
    // release what only the link pass needed
    P->hash.release();
    P->name_off.release();
    P->name_len.release();
    P->names.release();

    mfkio_arrays &a = P->a;
    a.n_reads = (int64_t)P->ref_start.size();
    a.n_contigs = n_contigs;
    a.max_span = P->max_span;
    a.n_cx = (int64_t)P->cx_read.size();
    a.n_blk = (int64_t)P->blk_start.size();
    a.n_pool = (int64_t)P->pool_read.size();
    a.pool_bases_bytes = (int64_t)P->pool_bases.size();
    a.pool_quals_bytes = (int64_t)P->pool_quals.size();
    a.ref_start = P->ref_start.data(); a.ref_end = P->ref_end.data(); a.link = P->link.data();
    a.contig_id = P->contig_id.data(); a.qlen = P->qlen.data(); a.first_op_len = P->first_op.data();
    a.last_op_len = P->last_op.data(); a.mapq = P->mapq.data(); a.flags = P->flags.data();
    a.contig_len = P->contig_len.data(); a.contig_read_start = P->contig_read_start.data();
    a.cx_read = P->cx_read.data(); a.cx_off = P->cx_off.data();
    a.blk_start = P->blk_start.data(); a.blk_end = P->blk_end.data(); a.blk_ins = P->blk_ins.data();
    a.pool_read = P->pool_read.data(); a.pool_off = P->pool_off.data();
    a.pool_bases = P->pool_bases.data(); a.pool_quals = P->pool_quals.data();
    a.contig_names = P->contig_names.data();
    a.contig_names_bytes = (int64_t)P->contig_names.size();
    return t2;
}


// ---- record index file ---------------------------------------------------------------------------------------------
// "MFKI0002" | bam size u64 | bam mtime ns i64 | uncompressed bytes u64 | first record offset u64 | max_span i32 |
// exclude_flags u32 | n_marks u64 | n_blocks u64 | marks (u64 offset, i32 ref, i32 pos) | blocks (u64 file offset, u64 first
// uncompressed byte).  Written next to the findflanks table by a full pass, read by the region pass of pairflanks; a file
// that does not match the BAM it is used with (size, mtime, FLAG filter) is ignored.
static const char IDX_MAGIC[8] = {'M', 'F', 'K', 'I', '0', '0', '0', '2'};
inline int64_t mtime_ns(const struct stat &st) { return (int64_t)st.st_mtim.tv_sec * 1000000000ll + (int64_t)st.st_mtim.tv_nsec; }

void write_index(mfkio_pack *P, const char *path, const struct stat &st, uint64_t total_u, uint32_t exclude_flags,
                 const std::vector<Block> &blocks, const std::vector<uint64_t> &block_at, uint64_t hdr_u_end) {
    std::string tmp = std::string(path) + ".tmp";
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return;
    uint64_t h64[4] = {(uint64_t)st.st_size, (uint64_t)mtime_ns(st), total_u, hdr_u_end};
    int32_t h32[2] = {P->max_span, (int32_t)exclude_flags};
    uint64_t n[2] = {(uint64_t)P->index.size(), (uint64_t)blocks.size()};
    bool ok = fwrite(IDX_MAGIC, 8, 1, f) == 1 && fwrite(h64, sizeof h64, 1, f) == 1 && fwrite(h32, sizeof h32, 1, f) == 1 && fwrite(n, sizeof n, 1, f) == 1;
    if (ok && !P->index.empty()) ok = fwrite(P->index.data(), sizeof(IdxEntry), P->index.size(), f) == P->index.size();
    uint64_t u = 0;
    for (size_t k = 0; ok && k < blocks.size(); ++k) {
        uint64_t row[2] = {block_at[k], u};
        ok = fwrite(row, sizeof row, 1, f) == 1;
        u += blocks[k].ulen;
    }
    ok = fclose(f) == 0 && ok;
    if (!ok || rename(tmp.c_str(), path) != 0) unlink(tmp.c_str());
}

struct Index {
    uint64_t bam_size = 0, total_u = 0, hdr_u_end = 0;
    int64_t bam_mtime = 0;
    int32_t max_span = 0;
    uint32_t exclude_flags = 0;
    std::vector<IdxEntry> marks;
    std::vector<uint64_t> block_at, block_u;
};

bool load_index(const char *path, Index *ix) {
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    char magic[8];
    uint64_t h64[4], n[2];
    int32_t h32[2];
    bool ok = fread(magic, 8, 1, f) == 1 && memcmp(magic, IDX_MAGIC, 8) == 0 && fread(h64, sizeof h64, 1, f) == 1 &&
              fread(h32, sizeof h32, 1, f) == 1 && fread(n, sizeof n, 1, f) == 1 && n[0] < ((uint64_t)1 << 32) && n[1] < ((uint64_t)1 << 32);
    if (ok) {
        ix->bam_size = h64[0]; ix->bam_mtime = (int64_t)h64[1]; ix->total_u = h64[2]; ix->hdr_u_end = h64[3];
        ix->max_span = h32[0]; ix->exclude_flags = (uint32_t)h32[1];
        ix->marks.resize((size_t)n[0]);
        ok = n[0] == 0 || fread(ix->marks.data(), sizeof(IdxEntry), (size_t)n[0], f) == (size_t)n[0];
        std::vector<uint64_t> rows((size_t)n[1] * 2);
        ok = ok && (n[1] == 0 || fread(rows.data(), 16, (size_t)n[1], f) == (size_t)n[1]);
        if (ok) {
            ix->block_at.resize((size_t)n[1]); ix->block_u.resize((size_t)n[1]);
            for (size_t k = 0; k < (size_t)n[1]; ++k) { ix->block_at[k] = rows[2 * k]; ix->block_u[k] = rows[2 * k + 1]; }
        }
    }
    fclose(f);
    // marks and blocks ascend, or the file is not one of ours
    for (size_t k = 1; ok && k < ix->marks.size(); ++k) ok = ix->marks[k].u > ix->marks[k - 1].u;
    for (size_t k = 1; ok && k < ix->block_u.size(); ++k) ok = ix->block_u[k] > ix->block_u[k - 1] && ix->block_at[k] > ix->block_at[k - 1];
    // every offset inside the stream it describes (a damaged file must fail here, not index out of the tables later)
    if (ok && !ix->block_u.empty()) ok = ix->block_u.front() == 0 && ix->block_u.back() < ix->total_u && ix->block_at.back() < ix->bam_size;
    if (ok) ok = ix->hdr_u_end <= ix->total_u && ix->max_span >= 0 && ix->total_u < ((uint64_t)1 << 46);
    if (ok && !ix->marks.empty()) ok = ix->marks.back().u < ix->total_u && ix->marks.front().u >= ix->hdr_u_end;
    return ok;
}

// Shares (or index ranges) that were parsed side by side go into the pack in file order: offsets first (sequential, a few
// numbers per share), then parallel copies.
int place_shares(mfkio_pack *P, std::vector<Local> &locals, int S, const std::vector<int64_t> &seg_records, int64_t &n_records, bool keep_index) {
    // place the shares in file order: offsets first (sequential, a few numbers per share), then parallel copies
    struct Base { size_t reads, names, cx, blk, pool, pbytes, qbytes; };
    std::vector<Base> base((size_t)S);
    Base cur{P->ref_start.size(), P->names.size(), P->cx_read.size(), P->blk_start.size(), P->pool_read.size(),
             P->pool_bases.size(), P->pool_quals.size()};
    for (int s = 0; s < S; ++s) {
        Local &L = locals[(size_t)s];
        if (!L.err.empty()) return fail(P, L.format_err ? MFKIO_ERR_FORMAT : MFKIO_ERR_PACK, L.err);  // first in file order
        n_records += seg_records[(size_t)s];
        base[(size_t)s] = cur;
        if (L.ref_start.empty()) continue;
        if (L.first_key < P->last_key)
            return fail(P, MFKIO_ERR_PACK, "alignments are not coordinate-sorted (the reference needs a sorted, indexed BAM)");
        P->last_key = L.last_key;
        if (L.max_span > P->max_span) P->max_span = L.max_span;
        cur.reads += L.ref_start.size(); cur.names += L.names.size(); cur.cx += L.cx_read.size();
        cur.blk += L.blk_start.size(); cur.pool += L.pool_read.size(); cur.pbytes += L.pool_bases.size();
        cur.qbytes += L.pool_quals.size();
    }
    if (cur.reads > (size_t)INT32_MAX) return fail(P, MFKIO_ERR_PACK, "more than 2^31-1 reads");
    if (cur.blk > (size_t)UINT32_MAX) return fail(P, MFKIO_ERR_PACK, "more than 2^32-1 aligned blocks");
    P->ref_start.grow(cur.reads); P->ref_end.grow(cur.reads); P->contig_id.grow(cur.reads); P->qlen.grow(cur.reads);
    P->first_op.grow(cur.reads); P->last_op.grow(cur.reads); P->mapq.grow(cur.reads); P->flags.grow(cur.reads);
    P->name_len.grow(cur.reads); P->hash.grow(cur.reads); P->name_off.grow(cur.reads); P->names.grow(cur.names);
    P->cx_read.grow(cur.cx); P->cx_off.grow(cur.cx + 1); P->blk_start.grow(cur.blk); P->blk_end.grow(cur.blk);
    P->blk_ins.grow(cur.blk); P->pool_read.grow(cur.pool); P->pool_off.grow(cur.pool + 1);
    P->pool_bases.grow(cur.pbytes); P->pool_quals.grow(cur.qbytes);
    parallel(S, [&](int s) {
        const Local &L = locals[(size_t)s];
        const Base &b = base[(size_t)s];
        put(P->ref_start, b.reads, L.ref_start); put(P->ref_end, b.reads, L.ref_end);
        put(P->contig_id, b.reads, L.contig_id); put(P->qlen, b.reads, L.qlen); put(P->first_op, b.reads, L.first_op);
        put(P->last_op, b.reads, L.last_op); put(P->mapq, b.reads, L.mapq); put(P->flags, b.reads, L.flags);
        put(P->name_len, b.reads, L.name_len); put(P->hash, b.reads, L.hash); put(P->names, b.names, L.names);
        for (size_t k = 0; k < L.name_off.size(); ++k) P->name_off.p[b.reads + k] = b.names + L.name_off[k];
        uint32_t boff = (uint32_t)b.blk;
        for (size_t k = 0; k < L.cx_read.size(); ++k) {
            P->cx_read.p[b.cx + k] = (uint32_t)(b.reads + L.cx_read[k]);
            boff += L.cx_nblk[k];
            P->cx_off.p[b.cx + k + 1] = boff;
        }
        put(P->blk_start, b.blk, L.blk_start); put(P->blk_end, b.blk, L.blk_end); put(P->blk_ins, b.blk, L.blk_ins);
        uint64_t poff = b.qbytes;                             // base offsets count padded bases = quality bytes
        for (size_t k = 0; k < L.pool_read.size(); ++k) {
            P->pool_read.p[b.pool + k] = (uint32_t)(b.reads + L.pool_read[k]);
            poff += L.pool_plen[k];
            P->pool_off.p[b.pool + k + 1] = poff;
        }
        put(P->pool_bases, b.pbytes, L.pool_bases); put(P->pool_quals, b.qbytes, L.pool_quals);
    });
    if (keep_index)
        for (int s = 0; s < S; ++s) P->index.insert(P->index.end(), locals[(size_t)s].idx.begin(), locals[(size_t)s].idx.end());
    return MFKIO_OK;
}

int run(mfkio_pack *P, const char *path, int pool_mode, const mfkio_region *regions, int64_t n_regions, int T,
        int64_t chunk_bytes, uint32_t exclude_flags, const char *index_out) {
    double t_begin = now_s(), t_inflate = 0, t_parse = 0, t_par = 0, t_merge = 0;
    if (pool_mode < MFKIO_POOL_NONE || pool_mode > MFKIO_POOL_REGIONS) return fail(P, MFKIO_ERR_ARG, "bad pool_mode");
    if (pool_mode == MFKIO_POOL_REGIONS && n_regions > 0 && !regions) return fail(P, MFKIO_ERR_ARG, "regions is NULL");
    int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(P, MFKIO_ERR_IO, std::string("cannot open ") + path);
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); return fail(P, MFKIO_ERR_IO, std::string("cannot stat ") + path); }
    size_t fsize = (size_t)st.st_size;
    const uint8_t *file = nullptr;
    if (fsize) {
        void *m = mmap(nullptr, fsize, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) { close(fd); return fail(P, MFKIO_ERR_IO, std::string("cannot map ") + path); }
        file = (const uint8_t *)m;
        madvise(m, fsize, MADV_SEQUENTIAL);
    }
    close(fd);
    struct Unmap { const uint8_t *p; size_t n; ~Unmap() { if (p) munmap((void *)p, n); } } unmap{file, fsize};

    // BGZF block table (SAM spec 4.1): gzip members with a 'BC' extra subfield holding the block size
    std::vector<Block> blocks;
    uint64_t total_u = 0;
    std::vector<uint64_t> block_at;   // file offset of every block kept in `blocks` (for the index)
    for (size_t o = 0; o < fsize;) {
        Block b; size_t bsize = 0;
        if (const char *why = block_header(file, fsize, o, path, &b, &bsize)) return fail(P, MFKIO_ERR_FORMAT, why);
        if (b.ulen) { blocks.push_back(b); if (index_out) block_at.push_back(o); total_u += b.ulen; }
        o += bsize;
    }

    Shared sh;
    sh.pool_mode = pool_mode;
    sh.exclude_flags = exclude_flags;
    const LibDeflate &ld = libdeflate();
    std::vector<Local> locals((size_t)T);
    std::vector<uint64_t> uoff;
    std::vector<size_t> seg_start, seg_end;
    std::vector<int64_t> seg_records;
    size_t cap = 0, carry = 0;
    uint8_t *buf = nullptr;
    struct Free { uint8_t **p; ~Free() { free(*p); } } free_buf{&buf};
    bool header_done = false;
    uint64_t hdr_u_end = 0;   // offset of the first alignment record in the uncompressed stream
    int64_t n_records = 0, n_redone = 0;
    P->cx_off.grow(1);
    P->cx_off.p[0] = 0;
    P->pool_off.grow(1);
    P->pool_off.p[0] = 0;

    uint64_t u_done = 0;   // uncompressed bytes of the blocks before this pass
    for (size_t b0 = 0; b0 < blocks.size();) {
        size_t b1 = b0;
        uint64_t usum = 0;
        uoff.clear();
        while (b1 < blocks.size() && (b1 == b0 || usum < (uint64_t)chunk_bytes)) {
            uoff.push_back(usum);
            usum += blocks[b1].ulen;
            ++b1;
        }
        size_t len = carry + (size_t)usum;
        const uint64_t pass_base = u_done - carry;   // offset in the uncompressed stream of buf[0]
        u_done += usum;
        if (len > cap) {
            size_t ncap = len + len / 4 + 65536;
            uint8_t *nb = (uint8_t *)realloc(buf, ncap);
            if (!nb) return fail(P, MFKIO_ERR_IO, "out of memory for the inflate buffer");
            buf = nb;
            cap = ncap;
        }
        // inflate the pass
        double t0 = now_s();
        std::atomic<size_t> next{0};
        std::atomic<int> bad{0};
        const size_t nb = b1 - b0;
        parallel(T, [&](int) {
            void *dec = ld.ok() ? ld.alloc() : nullptr;
            z_stream zs;
            memset(&zs, 0, sizeof zs);
            bool z_ok = dec || inflateInit2(&zs, -15) == Z_OK;
            for (;;) {
                size_t k = next.fetch_add(8);
                if (k >= nb || !z_ok) break;
                for (size_t j = k; j < std::min(k + 8, nb); ++j) {
                    const Block &b = blocks[b0 + j];
                    uint8_t *dst = buf + carry + uoff[j];
                    uint32_t crc;
                    if (dec) {
                        size_t got = 0;
                        if (ld.decompress(dec, file + b.payload, b.clen, dst, b.ulen, &got) != 0 || got != b.ulen) { bad = 1; continue; }
                        crc = ld.crc(0, dst, b.ulen);
                    } else {
                        inflateReset(&zs);
                        zs.next_in = (Bytef *)(file + b.payload);
                        zs.avail_in = b.clen;
                        zs.next_out = dst;
                        zs.avail_out = b.ulen;
                        if (inflate(&zs, Z_FINISH) != Z_STREAM_END || zs.avail_out != 0) { bad = 1; continue; }
                        crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), dst, b.ulen);
                    }
                    if (crc != b.crc) bad = 2;
                }
            }
            if (!z_ok) bad = 1;
            if (dec) ld.release(dec);
            else if (z_ok) inflateEnd(&zs);
        });
        if (bad == 2) return fail(P, MFKIO_ERR_FORMAT, "BGZF checksum mismatch (corrupt file)");
        if (bad) return fail(P, MFKIO_ERR_FORMAT, "BGZF block does not inflate (corrupt file)");
        double t1 = now_s();
        t_inflate += t1 - t0;
        b0 = b1;

        size_t off = 0;
        if (!header_done) {
            int64_t h = parse_header(P, buf, len);
            if (h < 0) return fail(P, MFKIO_ERR_FORMAT, std::string(path) + " is not a BAM file");
            if (h == 0) { carry = len; continue; }
            header_done = true;
            off = (size_t)h;
            hdr_u_end = pass_base + off;
            if (P->contig_len.size() > 65535) return fail(P, MFKIO_ERR_PACK, "more than 65535 contigs");
            sh.n_contigs = (int32_t)P->contig_len.size();
            if (pool_mode == MFKIO_POOL_REGIONS) {
                std::vector<std::vector<std::pair<int32_t, int32_t>>> nr((size_t)sh.n_contigs), wn((size_t)sh.n_contigs);
                for (int64_t i = 0; i < n_regions; ++i) {
                    const mfkio_region &r = regions[i];
                    if (r.contig_id < 0 || r.contig_id >= sh.n_contigs) return fail(P, MFKIO_ERR_ARG, "region on an unknown contig");
                    nr[r.contig_id].push_back({r.lo, r.hi});
                    wn[r.contig_id].push_back({r.es, r.ee});
                }
                sh.near_.resize((size_t)sh.n_contigs);
                sh.win_.resize((size_t)sh.n_contigs);
                for (int c = 0; c < sh.n_contigs; ++c) { sh.near_[c].build(nr[c]); sh.win_[c].build(wn[c]); }
            }
        }
        // Shares of the pass.  Record lengths chain from the front, and following that chain alone is one cache miss per
        // record; so share t > 0 starts at a GUESSED record boundary near t/S of the pass, every share walks and parses
        // its own records, and the guesses are verified below.
        auto walk = [&](int s, int S) {
            Local &L = locals[(size_t)s];
            L.clear();
            size_t o = seg_start[(size_t)s], end = s + 1 < S ? seg_start[(size_t)s + 1] : SIZE_MAX;
            int64_t nrec = 0;
            while (o < end && o + 4 <= len) {
                uint32_t bs = le32(buf + o);
                if (bs < 32 || bs > (1u << 30)) { L.err = "bad alignment record size (corrupt file)"; L.format_err = true; break; }
                if (o + 4 + (size_t)bs > len) break;
                if (index_out) {
                    const uint64_t au = pass_base + o;
                    if (au >= L.next_mark) {
                        L.idx.push_back({au, (int32_t)le32(buf + o + 4), (int32_t)le32(buf + o + 8)});
                        L.next_mark = (au / IDX_GRANULE + 1) * IDX_GRANULE;
                    }
                }
                if (!parse_record(sh, buf + o + 4, bs, L)) break;
                o += 4 + (size_t)bs;
                ++nrec;
            }
            seg_end[(size_t)s] = o;
            seg_records[(size_t)s] = nrec;
        };
        int S = (int)std::min<size_t>((size_t)T, (len - off) / 65536);
        if (S < 1) S = 1;
        seg_start.assign((size_t)S, off);
        seg_end.assign((size_t)S, off);
        seg_records.assign((size_t)S, 0);
        if (S > 1) {
            std::vector<size_t> guess((size_t)S, SIZE_MAX);
            parallel(S, [&](int s) {
                if (!s) return;
                size_t lo = off + (len - off) / (size_t)S * (size_t)s, hi = off + (len - off) / (size_t)S * ((size_t)s + 1);
                for (size_t q = lo; q < hi; ++q)
                    if (chain_plausible(sh, buf, q, len)) { guess[(size_t)s] = q; break; }
            });
            seg_start.clear();
            seg_start.push_back(off);
            for (int s = 1; s < S; ++s) if (guess[(size_t)s] != SIZE_MAX) seg_start.push_back(guess[(size_t)s]);
            S = (int)seg_start.size();
        }
        parallel(S, [&](int s) { walk(s, S); });
        bool redo = false;
        for (int s = 0; s < S && !redo; ++s) {
            if (!locals[(size_t)s].err.empty()) break;            // a true error: every share before it was verified
            if (s + 1 < S && seg_end[(size_t)s] != seg_start[(size_t)s + 1]) redo = true;
        }
        if (redo) {                                               // a wrong guess: walk the pass from the front
            S = 1;
            ++n_redone;
            seg_start.assign(1, off);
            walk(0, 1);
        }
        double tp = now_s();
        t_par += tp - t1;
        if (int rc = place_shares(P, locals, S, seg_records, n_records, index_out != nullptr)) return rc;
        off = seg_end[(size_t)S - 1];
        carry = len - off;
        if (carry && off) memmove(buf, buf + off, carry);
        t_parse += now_s() - t1;
        t_merge += now_s() - tp;
    }
    if (getenv("MFK_IO_TRACE"))
        fprintf(stderr, "[mfkio] walk+parse %.3f s, merge %.3f s\n", t_par, t_merge);
    if (!header_done) return fail(P, MFKIO_ERR_FORMAT, std::string(path) + " is not a BAM file (no header)");
    if (carry) return fail(P, MFKIO_ERR_FORMAT, "truncated alignment record at the end of the file");

    if (index_out) write_index(P, index_out, st, total_u, exclude_flags, blocks, block_at, hdr_u_end);   // best effort: a directory that cannot be written costs pairflanks a full pass, nothing else
    double t3 = 0, t2 = finish_pack(P, sh.n_contigs, T, &t3);
    mfkio_arrays &a = P->a;
    a.n_records = n_records;
    a.compressed_bytes = (int64_t)fsize;
    a.uncompressed_bytes = (int64_t)total_u;
    a.n_threads = T;
    a.inflate_backend = ld.ok() ? 1 : 0;
    a.seconds_inflate = t_inflate;
    a.seconds_parse = t_parse;
    a.seconds_link = t3 - t2;
    a.seconds_total = now_s() - t_begin;
    a.n_redone_passes = n_redone;
    return MFKIO_OK;
}


// ---- the region pass through a record index ------------------------------------------------------------------------------
// pairflanks needs the reads over a few dozen to a few thousand short regions; the reference fetches them through the BAI
// (pairflanks.py:393-409).  With the record index findflanks left behind, only the BGZF blocks between the marks around each
// region are inflated and walked.  Every record the full region pass would pack is packed, in the same order, so the
// arrays are equal (tests/test_bamio_native.py).  Returns MFKIO_ERR_INDEX when the index cannot serve (missing, stale,
// another FLAG filter, inverted regions): the caller then runs the full pass.
inline int64_t mark_key(int32_t ref, int32_t pos) { return ref < 0 ? INT64_MAX : ((int64_t)ref << 32) + (int64_t)(uint32_t)(pos < 0 ? 0 : pos); }

int run_indexed(mfkio_pack *P, const char *path, const char *index_path, const mfkio_region *regions, int64_t n_regions, int T, uint32_t exclude_flags) {
    double t_begin = now_s(), t_inflate = 0, t_parse = 0;
    if (n_regions > 0 && !regions) return fail(P, MFKIO_ERR_ARG, "regions is NULL");
    Index ix;
    if (!load_index(index_path, &ix)) return fail(P, MFKIO_ERR_INDEX, std::string("no usable record index at ") + index_path);
    int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(P, MFKIO_ERR_IO, std::string("cannot open ") + path);
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); return fail(P, MFKIO_ERR_IO, std::string("cannot stat ") + path); }
    const size_t fsize = (size_t)st.st_size;
    if ((uint64_t)fsize != ix.bam_size || mtime_ns(st) != ix.bam_mtime || exclude_flags != ix.exclude_flags || ix.marks.empty() || ix.block_at.empty() || !fsize) {
        close(fd);
        return fail(P, MFKIO_ERR_INDEX, "the record index was made for another file or another FLAG filter");
    }
    void *m = mmap(nullptr, fsize, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return fail(P, MFKIO_ERR_IO, std::string("cannot map ") + path);
    const uint8_t *file = (const uint8_t *)m;
    struct Unmap { const uint8_t *p; size_t n; ~Unmap() { munmap((void *)p, n); } } unmap{file, fsize};
    const LibDeflate &ld = libdeflate();
    const size_t NB = ix.block_at.size();
    auto block_of = [&](uint64_t u) { return (size_t)(std::upper_bound(ix.block_u.begin(), ix.block_u.end(), u) - ix.block_u.begin()) - 1; };

    // blocks [bA, bB] -> buf (grown as needed); returns 0 or the failure, *base = uncompressed offset of buf[0]
    struct Inflated { std::vector<uint8_t> buf; uint64_t base = 0; };
    auto inflate_run = [&](size_t bA, size_t bB, Inflated &out, void *dec, z_stream *zs) -> const char * {
        out.base = ix.block_u[bA];
        const uint64_t end_u = bB + 1 < NB ? ix.block_u[bB + 1] : ix.total_u;
        out.buf.resize((size_t)(end_u - out.base));
        for (size_t k = bA; k <= bB; ++k) {
            Block b; size_t bsize = 0;
            if (ix.block_at[k] >= fsize) return "record index points beyond the file";
            if (const char *why = block_header(file, fsize, (size_t)ix.block_at[k], path, &b, &bsize)) return why;
            const uint64_t want = (k + 1 < NB ? ix.block_u[k + 1] : ix.total_u) - ix.block_u[k];
            if (b.ulen != want) return "record index does not match the BGZF blocks of the file";
            uint8_t *dst = out.buf.data() + (ix.block_u[k] - out.base);
            uint32_t crc;
            if (dec) {
                size_t got = 0;
                if (ld.decompress(dec, file + b.payload, b.clen, dst, b.ulen, &got) != 0 || got != b.ulen) return "BGZF block does not inflate (corrupt file)";
                crc = ld.crc(0, dst, b.ulen);
            } else {
                inflateReset(zs);
                zs->next_in = (Bytef *)(file + b.payload); zs->avail_in = b.clen; zs->next_out = dst; zs->avail_out = b.ulen;
                if (inflate(zs, Z_FINISH) != Z_STREAM_END || zs->avail_out != 0) return "BGZF block does not inflate (corrupt file)";
                crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), dst, b.ulen);
            }
            if (crc != b.crc) return "BGZF checksum mismatch (corrupt file)";
        }
        return nullptr;
    };
    struct Codec {
        const LibDeflate &ld; void *dec = nullptr; z_stream zs; bool ok = false;
        explicit Codec(const LibDeflate &l) : ld(l) { memset(&zs, 0, sizeof zs); dec = ld.ok() ? ld.alloc() : nullptr; ok = dec || inflateInit2(&zs, -15) == Z_OK; }
        ~Codec() { if (dec) ld.release(dec); else if (ok) inflateEnd(&zs); }
    };

    // header: the blocks up to the first record
    {
        Codec cd(ld);
        if (!cd.ok) return fail(P, MFKIO_ERR_IO, "cannot set up the inflater");
        Inflated hb;
        const size_t bH = block_of(ix.hdr_u_end ? ix.hdr_u_end - 1 : 0);
        if (const char *why = inflate_run(0, bH, hb, cd.dec, &cd.zs)) return fail(P, MFKIO_ERR_FORMAT, why);
        int64_t h = parse_header(P, hb.buf.data(), hb.buf.size());
        if (h <= 0 || (uint64_t)h != ix.hdr_u_end) return fail(P, MFKIO_ERR_INDEX, "the record index does not match the BAM header");
    }
    if (P->contig_len.size() > 65535) return fail(P, MFKIO_ERR_PACK, "more than 65535 contigs");
    Shared sh;
    sh.pool_mode = MFKIO_POOL_REGIONS;
    sh.exclude_flags = exclude_flags;
    sh.n_contigs = (int32_t)P->contig_len.size();
    {
        std::vector<std::vector<std::pair<int32_t, int32_t>>> nr((size_t)sh.n_contigs), wn((size_t)sh.n_contigs);
        for (int64_t i = 0; i < n_regions; ++i) {
            const mfkio_region &r = regions[i];
            if (r.contig_id < 0 || r.contig_id >= sh.n_contigs) return fail(P, MFKIO_ERR_ARG, "region on an unknown contig");
            nr[r.contig_id].push_back({r.lo, r.hi});
            wn[r.contig_id].push_back({r.es, r.ee});
        }
        sh.near_.resize((size_t)sh.n_contigs);
        sh.win_.resize((size_t)sh.n_contigs);
        for (int c = 0; c < sh.n_contigs; ++c) {
            sh.near_[c].build(nr[c]); sh.win_[c].build(wn[c]);
            if (!sh.near_[c].odd.empty()) return fail(P, MFKIO_ERR_INDEX, "inverted region: full pass");
        }
    }
    // stretches of the uncompressed stream that can hold a read over a region: from the last mark at or before
    // (contig, lo - max_span) to the first mark at or behind (contig, hi); neighbours merged
    std::vector<std::pair<uint64_t, uint64_t>> stretch;
    std::vector<int64_t> keys(ix.marks.size());
    for (size_t k = 0; k < ix.marks.size(); ++k) keys[k] = mark_key(ix.marks[k].ref, ix.marks[k].pos);
    for (size_t k = 1; k < keys.size(); ++k) if (keys[k] < keys[k - 1]) return fail(P, MFKIO_ERR_INDEX, "the record index is not in coordinate order");
    for (int c = 0; c < sh.n_contigs; ++c)
        for (size_t j = 0; j < sh.near_[c].lo.size(); ++j) {
            const int64_t lo = (int64_t)sh.near_[c].lo[j] - ix.max_span, hi = sh.near_[c].hi[j];
            if (hi <= 0) continue;
            const int64_t klo = ((int64_t)c << 32) + (lo < 0 ? 0 : lo), khi = ((int64_t)c << 32) + hi;
            size_t e0 = (size_t)(std::upper_bound(keys.begin(), keys.end(), klo) - keys.begin());   // first mark beyond klo
            e0 = e0 ? e0 - 1 : 0;
            const size_t e1 = (size_t)(std::lower_bound(keys.begin(), keys.end(), khi) - keys.begin());
            const uint64_t u0 = ix.marks[e0].u, u1 = e1 < keys.size() ? ix.marks[e1].u : ix.total_u;
            if (u1 <= u0) continue;
            if (!stretch.empty() && u0 <= stretch.back().second) stretch.back().second = std::max(stretch.back().second, u1);
            else stretch.push_back({u0, u1});
        }
    std::sort(stretch.begin(), stretch.end());
    {   // (contigs ascend and so do their stretches, but merged neighbours of different contigs may touch)
        std::vector<std::pair<uint64_t, uint64_t>> mg;
        for (auto &x : stretch) { if (!mg.empty() && x.first <= mg.back().second) mg.back().second = std::max(mg.back().second, x.second); else mg.push_back(x); }
        stretch.swap(mg);
    }
    P->cx_off.grow(1); P->cx_off.p[0] = 0;
    P->pool_off.grow(1); P->pool_off.p[0] = 0;
    int64_t n_records = 0, inflated_bytes = 0;
    // T groups of neighbouring stretches, balanced by bytes; a thread walks its group in file order into one Local (the
    // coordinates ascend from stretch to stretch, so the sorted-input check of parse_record holds across them)
    uint64_t all_bytes = 0;
    for (auto &x : stretch) all_bytes += x.second - x.first;
    int S = (int)std::min<size_t>((size_t)T, stretch.size());
    if (S < 1) S = 1;
    std::vector<size_t> g_begin((size_t)S + 1, stretch.size());
    {
        uint64_t acc = 0; int g = 0;
        g_begin[0] = 0;
        for (size_t k = 0; k < stretch.size(); ++k) {
            while (g + 1 < S && acc >= all_bytes * (uint64_t)(g + 1) / (uint64_t)S && k > g_begin[(size_t)g]) g_begin[(size_t)++g] = k;
            acc += stretch[k].second - stretch[k].first;
        }
        for (int q = g + 1; q <= S; ++q) g_begin[(size_t)q] = stretch.size();
    }
    std::vector<Local> locals((size_t)S);
    std::vector<int64_t> seg_records((size_t)S, 0);
    std::vector<std::string> why_bad((size_t)S);
    std::vector<double> t_inf((size_t)S, 0.0), t_all((size_t)S, 0.0);
    std::atomic<int64_t> inf_bytes{0};
    parallel(S, [&](int s) {
        const double t_in = now_s();
        Local &L = locals[(size_t)s];
        L.clear();
        Codec cd(ld);
        if (!cd.ok) { why_bad[(size_t)s] = "cannot set up the inflater"; return; }
        Inflated inf;
        int64_t nrec = 0;
        for (size_t k = g_begin[(size_t)s]; k < g_begin[(size_t)s + 1] && L.err.empty(); ++k) {
            const uint64_t u0 = stretch[k].first, u1 = stretch[k].second;
            const size_t bA = block_of(u0);
            size_t bB = block_of(u1 - 1);
            for (;;) {
                double t0 = now_s();
                if (const char *why = inflate_run(bA, bB, inf, cd.dec, &cd.zs)) { why_bad[(size_t)s] = why; return; }
                t_inf[(size_t)s] += now_s() - t0;
                const uint8_t *buf = inf.buf.data();
                const size_t len = inf.buf.size(), end = (size_t)(u1 - inf.base);
                size_t o = (size_t)(u0 - inf.base);
                // the last record of a stretch may run into the next block(s): look before anything is parsed
                size_t q = o;
                bool more = false;
                while (q < end) {
                    if (q + 4 > len) { more = true; break; }
                    uint32_t bs = le32(buf + q);
                    if (bs < 32 || bs > (1u << 30)) break;          // reported by the walk below
                    if (q + 4 + (size_t)bs > len) { more = true; break; }
                    q += 4 + (size_t)bs;
                }
                if (more && bB + 1 < NB) { bB = std::min(NB - 1, bB + 1 + (bB - bA)); continue; }
                if (more) { L.err = "truncated alignment record at the end of the file"; L.format_err = true; break; }
                while (o < end) {
                    uint32_t bs = le32(buf + o);
                    if (bs < 32 || bs > (1u << 30)) { L.err = "bad alignment record size (corrupt file or stale record index)"; L.format_err = true; break; }
                    if (!parse_record(sh, buf + o + 4, bs, L)) break;
                    o += 4 + (size_t)bs;
                    ++nrec;
                }
                if (L.err.empty() && o != end) { L.err = "the record index does not land on a record boundary (stale index)"; L.format_err = true; }
                inf_bytes += (int64_t)len;
                break;
            }
        }
        seg_records[(size_t)s] = nrec;
        t_all[(size_t)s] = now_s() - t_in;
    });
    for (int s = 0; s < S; ++s) if (!why_bad[(size_t)s].empty()) return fail(P, MFKIO_ERR_FORMAT, why_bad[(size_t)s]);
    {
        double ti = 0, ta = 0;
        for (int s = 0; s < S; ++s) { ti = std::max(ti, t_inf[(size_t)s]); ta = std::max(ta, t_all[(size_t)s]); }
        t_inflate = ti;
        double tp0 = now_s();
        if (int rc = place_shares(P, locals, S, seg_records, n_records, false)) return rc;
        t_parse = (ta - ti) + (now_s() - tp0);
    }
    inflated_bytes = inf_bytes.load();
    double t3 = 0, t2 = finish_pack(P, sh.n_contigs, T, &t3);
    mfkio_arrays &a = P->a;
    a.n_records = n_records;
    a.compressed_bytes = (int64_t)fsize;
    a.uncompressed_bytes = inflated_bytes;     // what this pass inflated, not the file's total
    a.n_threads = T;
    a.inflate_backend = ld.ok() ? 1 : 0;
    a.seconds_inflate = t_inflate;
    a.seconds_parse = t_parse;
    a.seconds_link = t3 - t2;
    a.seconds_total = now_s() - t_begin;
    a.n_redone_passes = 0;
    return MFKIO_OK;
}

}  // namespace

extern "C" {

int mfkio_pack_bam(const char *path, int pool_mode, const mfkio_region *regions, int64_t n_regions, int n_threads,
                   int64_t chunk_bytes, mfkio_pack **out) {
    return mfkio_pack_bam_filtered(path, pool_mode, regions, n_regions, n_threads, chunk_bytes, 0u, out);
}

int mfkio_pack_bam_filtered(const char *path, int pool_mode, const mfkio_region *regions, int64_t n_regions, int n_threads,
                            int64_t chunk_bytes, uint32_t exclude_flags, mfkio_pack **out) {
    if (!out) return MFKIO_ERR_ARG;
    mfkio_pack *P = new (std::nothrow) mfkio_pack();
    *out = P;
    if (!P) return MFKIO_ERR_IO;
    memset(&P->a, 0, sizeof P->a);
    if (!path) return fail(P, MFKIO_ERR_ARG, "path is NULL");
    int T = n_threads;
    if (T <= 0) {
        const char *e = getenv("MFK_THREADS");
        T = e ? atoi(e) : 0;
        if (T <= 0) T = (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    }
    T = std::min(T, 256);
    if (chunk_bytes <= 0) chunk_bytes = (int64_t)256 << 20;
    try {
        return run(P, path, pool_mode, regions, n_regions, T, chunk_bytes, exclude_flags, nullptr);
    } catch (const std::bad_alloc &) {
        return fail(P, MFKIO_ERR_IO, "out of memory while packing");
    } catch (const std::exception &e) {
        return fail(P, MFKIO_ERR_IO, std::string("unexpected failure: ") + e.what());
    }
}

static int thread_count(int n_threads) {
    int T = n_threads;
    if (T <= 0) {
        const char *e = getenv("MFK_THREADS");
        T = e ? atoi(e) : 0;
        if (T <= 0) T = (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    }
    return std::min(T, 256);
}

int mfkio_pack_bam_index(const char *path, int pool_mode, int n_threads, int64_t chunk_bytes, uint32_t exclude_flags, const char *index_out,
                         mfkio_pack **out) {
    if (!out) return MFKIO_ERR_ARG;
    mfkio_pack *P = new (std::nothrow) mfkio_pack();
    *out = P;
    if (!P) return MFKIO_ERR_IO;
    memset(&P->a, 0, sizeof P->a);
    if (!path) return fail(P, MFKIO_ERR_ARG, "path is NULL");
    if (pool_mode == MFKIO_POOL_REGIONS) return fail(P, MFKIO_ERR_ARG, "the record index is written by a full pass");
    if (chunk_bytes <= 0) chunk_bytes = (int64_t)256 << 20;
    try {
        return run(P, path, pool_mode, nullptr, 0, thread_count(n_threads), chunk_bytes, exclude_flags, index_out);
    } catch (const std::bad_alloc &) {
        return fail(P, MFKIO_ERR_IO, "out of memory while packing");
    } catch (const std::exception &e) {
        return fail(P, MFKIO_ERR_IO, std::string("unexpected failure: ") + e.what());
    }
}

int mfkio_pack_bam_regions_indexed(const char *path, const char *index_path, const mfkio_region *regions, int64_t n_regions, int n_threads,
                                   uint32_t exclude_flags, mfkio_pack **out) {
    if (!out) return MFKIO_ERR_ARG;
    mfkio_pack *P = new (std::nothrow) mfkio_pack();
    *out = P;
    if (!P) return MFKIO_ERR_IO;
    memset(&P->a, 0, sizeof P->a);
    if (!path || !index_path) return fail(P, MFKIO_ERR_ARG, "path is NULL");
    try {
        return run_indexed(P, path, index_path, regions, n_regions, thread_count(n_threads), exclude_flags);
    } catch (const std::bad_alloc &) {
        return fail(P, MFKIO_ERR_IO, "out of memory while packing");
    } catch (const std::exception &e) {
        return fail(P, MFKIO_ERR_IO, std::string("unexpected failure: ") + e.what());
    }
}

const mfkio_arrays *mfkio_pack_arrays(const mfkio_pack *p) { return p ? &p->a : nullptr; }
const char *mfkio_pack_error(const mfkio_pack *p) { return p ? p->error.c_str() : "no pack object"; }
void mfkio_pack_free(mfkio_pack *p) { delete p; }
int mfkio_version(void) { return 1; }

}  // extern "C"
