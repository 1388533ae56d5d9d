// mfk_device.h -- device-side view of the packed alignments and the per-read predicates of the reference's
// sctools.py / findflanks.py, shared by every kernel of libmfk.so.  gfx950 (MI355X) only.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mfk.h"

#define MFK_WAVE 64

// device copy of mfk_reads: every pointer is a device pointer (the two header arrays are uploaded by the library)
struct DevReads {
    int64_t n;
    const int32_t *ref_start, *ref_end, *link;
    const uint16_t *contig_id, *qlen, *first_op_len, *last_op_len;
    const uint8_t *mapq, *flags;
    int32_t n_contigs, max_span;
    uint32_t gpos_span;              // length of the flat site axis (all contigs + gaps)
    const int32_t *contig_len;       // n_contigs
    const int64_t *contig_rs;        // n_contigs + 1
    const uint32_t *contig_base;     // n_contigs: offset of the contig in the flat "gpos" coordinate (see mfk.hip)
    int64_t n_cx;
    const uint32_t *cx_read, *cx_off;
    const int32_t *blk_start, *blk_end, *blk_ins;
    int64_t n_pool;
    const uint32_t *pool_read;
    const uint64_t *pool_off;
    const uint8_t *pool_bases, *pool_quals;
    int64_t n_clip;                  // optional dense records of the clipped reads (mfk.h mfk_clip_rec), or clip_rec == nullptr
    const mfk_clip_rec *clip_rec;
    const uint32_t *clip_tile;
};

struct DevParams {
    int32_t minlen, mincount, minq, minial, mindist, lins, mcc, maxdr;
    double minratio, maxir, maxjsp;
};

// ---- sctools.py ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int d_lkind(uint8_t f) { return f & 3; }
__device__ __forceinline__ int d_rkind(uint8_t f) { return (f >> 2) & 3; }
// len(left_softclipped_sequence) / len(right_softclipped_sequence), sctools.py:111-142
__device__ __forceinline__ int d_llen(const DevReads &R, int64_t r, uint8_t f) {
    int k = d_lkind(f);
    return k == 1 ? (int)R.first_op_len[r] : (k == 2 ? 1 : 0);
}
__device__ __forceinline__ int d_rlen(const DevReads &R, int64_t r, uint8_t f) {
    int k = d_rkind(f);
    return k == 1 ? (int)R.last_op_len[r] : (k == 2 ? 1 : 0);
}
// left_softclipped_site_lenient / right_softclipped_site_lenient, sctools.py:99-109
__device__ __forceinline__ int32_t d_lsite(const DevReads &R, int64_t r, uint8_t f) {
    return d_lkind(f) == 1 ? R.ref_start[r] - 1 : R.ref_start[r];
}
__device__ __forceinline__ int32_t d_rsite(const DevReads &R, int64_t r, uint8_t f) {
    return d_rkind(f) == 1 ? R.ref_end[r] : R.ref_end[r] - 1;
}
// passes_read_filters findflanks.py:586-592 + read_meets_min_alignment_inner_length sctools.py:166-180
__device__ __forceinline__ bool d_passes(const DevReads &R, const DevParams &P, int64_t r, uint8_t f) {
    if ((int)R.mapq[r] < P.minq) return false;
    if (d_lkind(f) && d_rkind(f)) {
        int aligned = (int)R.qlen[r] - d_rlen(R, r, f) - d_llen(R, r, f);
        if (aligned < P.minial) return false;
    }
    return true;
}
// is_softclipped_lenient_at_site sctools.py:33-55
__device__ __forceinline__ bool d_clipped_at(const DevReads &R, int64_t r, uint8_t f, int32_t pos) {
    if (d_rkind(f) && d_rsite(R, r, f) == pos) return true;
    if (d_lkind(f) && d_lsite(R, r, f) == pos) return true;
    return false;
}

// soft-clip event of read r on one side (0 = 5p / right clip, 1 = 3p / left clip): true if the read passes the
// filters and is lenient-clipped there; *site = reference position of the site
__device__ __forceinline__ bool d_clip_event(const DevReads &R, const DevParams &P, int64_t r, int side, int32_t *site) {
    uint8_t f = R.flags[r];
    if (side == 0 ? !d_rkind(f) : !d_lkind(f)) return false;
    if (!d_passes(R, P, r, f)) return false;
    *site = side == 0 ? d_rsite(R, r, f) : d_lsite(R, r, f);
    return true;
}

// ---- aligned blocks (pysam get_blocks) ------------------------------------------------------------------------
struct DevBlocks {
    int n;
    const int32_t *s, *e, *ins;
    int32_t s1, e1;
    __device__ __forceinline__ int32_t bs(int i) const { return s ? s[i] : s1; }
    __device__ __forceinline__ int32_t be(int i) const { return s ? e[i] : e1; }
    __device__ __forceinline__ int32_t bi(int i) const { return s ? ins[i] : 0; }
};
__device__ __forceinline__ DevBlocks d_blocks(const DevReads &R, int64_t r, uint8_t f) {
    DevBlocks b;
    if (!(f & MFK_FLAG_COMPLEX)) {
        b.n = 1; b.s = b.e = b.ins = nullptr; b.s1 = R.ref_start[r]; b.e1 = R.ref_end[r];
        return b;
    }
    int64_t lo = 0, hi = R.n_cx;
    while (lo < hi) { int64_t m = (lo + hi) >> 1; if ((int64_t)R.cx_read[m] < r) lo = m + 1; else hi = m; }
    b.s1 = b.e1 = 0;
    if (lo >= R.n_cx || (int64_t)R.cx_read[lo] != r) { b.n = 0; b.s = b.e = b.ins = nullptr; return b; }  // corrupt table
    uint32_t o0 = R.cx_off[lo], o1 = R.cx_off[lo + 1];
    b.n = (int)(o1 - o0); b.s = R.blk_start + o0; b.e = R.blk_end + o0; b.ins = R.blk_ins + o0;
    return b;
}

enum { CAT_NONE = 0, CAT_RUNTHROUGH, CAT_SMALL5, CAT_SMALL3, CAT_LARGE5, CAT_LARGE3, CAT_DELETION };

// process_aligned_blocks_at_site findflanks.py:455-509 (priority order of the returns: quirk Q9)
__device__ inline int d_classify(const DevReads &R, const DevParams &P, int64_t r, uint8_t f, int32_t position, bool deletions_only) {
    if (!(f & MFK_FLAG_COMPLEX)) return (R.ref_start[r] <= position && R.ref_end[r] > position) ? CAT_RUNTHROUGH : CAT_NONE;
    DevBlocks b = d_blocks(R, r, f);
    bool runthrough = false, s5 = false, s3 = false, l5 = false, l3 = false, del = false;
    for (int i = 0; i < b.n; i++) if (b.bs(i) <= position && b.be(i) > position) runthrough = true;
    for (int i = 0; i + 1 < b.n; i++) {
        int32_t e1 = b.be(i), s2 = b.bs(i + 1);
        if (e1 == s2) {
            if (deletions_only) continue;
            int32_t ins = b.bi(i + 1);  // get_insertion_length pysamtools.py:186-209
            if (position == e1 - 1) { if (ins >= P.lins) l3 = true; else s3 = true; }
            else if (position == s2) { if (ins >= P.lins) l5 = true; else s5 = true; }
        } else if (e1 <= position && s2 > position) del = true;
    }
    if (s5) return CAT_SMALL5;
    if (s3) return CAT_SMALL3;
    if (l5) return CAT_LARGE5;
    if (l3) return CAT_LARGE3;
    if (del) return CAT_DELETION;
    if (runthrough) return CAT_RUNTHROUGH;
    return CAT_NONE;
}

// identify_large_insertion_at_site pairflanks.py:446-470 (quirk P8: returns at the first gap)
__device__ inline bool d_large_insertion_at(const DevReads &R, const DevParams &P, int64_t r, uint8_t f, int32_t position) {
    if (!(f & MFK_FLAG_COMPLEX)) return false;
    DevBlocks b = d_blocks(R, r, f);
    for (int i = 0; i + 1 < b.n; i++) {
        int32_t e1 = b.be(i), s2 = b.bs(i + 1);
        if (e1 == s2) { if (position == e1 - 1 || position == s2) return b.bi(i + 1) >= P.lins; }
        else return false;
    }
    return false;
}

// ---- read range of a region query (bam.fetch) -----------------------------------------------------------------
// reads of `contig` that can overlap [start, end): indices [lo, hi) with ref_start in [start - max_span, end)
__device__ inline void d_fetch_range(const DevReads &R, int32_t contig, int32_t start, int32_t end, int64_t *lo_out, int64_t *hi_out) {
    int64_t r0 = R.contig_rs[contig], r1 = R.contig_rs[contig + 1];
    int64_t lo = r0, hi = r1; int32_t lim = start - R.max_span;
    while (lo < hi) { int64_t m = (lo + hi) >> 1; if (R.ref_start[m] < lim) lo = m + 1; else hi = m; }
    int64_t a = lo;
    hi = r1;
    while (lo < hi) { int64_t m = (lo + hi) >> 1; if (R.ref_start[m] < end) lo = m + 1; else hi = m; }
    *lo_out = a; *hi_out = lo;
}
// Wave-cooperative lower bound: first index in [lo, hi) with a[idx] >= key (hi if none).  Every lane of one full wave
// calls it with the same arguments.  Each round the 64 lanes probe 64 evenly spaced elements at once, so a search over
// 2^24 reads costs 4-5 dependent HBM loads instead of the 24 of a one-lane binary search (the per-site searches of
// k_classify / k_spanning / k_direct_repeats are pure memory latency).
template <typename T, typename K>
__device__ inline int64_t d_wave_lower_bound(const T *a, int64_t lo, int64_t hi, K key) {
    const int lane = threadIdx.x & 63;
    while (hi - lo > 64) {
        const int64_t step = (hi - lo + 63) >> 6;
        const int64_t p = lo + (int64_t)(lane + 1) * step - 1;
        const bool less = p < hi && (K)a[p] < key;
        const int c = __popcll(__ballot(less));          // sorted: lanes 0 .. c-1 see "less"
        int64_t nlo = lo + (int64_t)c * step;
        if (nlo > hi) nlo = hi;
        int64_t nhi = nlo + step - 1;                    // a[nhi] >= key when nhi < hi
        if (nhi > hi) nhi = hi;
        lo = nlo; hi = nhi;
    }
    const int64_t p = lo + lane;
    const bool less = p < hi && (K)a[p] < key;
    return lo + __popcll(__ballot(less));
}
// d_fetch_range by one full wave (all 64 lanes call it; all get the result)
__device__ inline void d_fetch_range_wave(const DevReads &R, int32_t contig, int32_t start, int32_t end, int64_t *lo_out, int64_t *hi_out) {
    const int64_t r0 = R.contig_rs[contig], r1 = R.contig_rs[contig + 1];
    const int64_t a = d_wave_lower_bound<int32_t, int32_t>(R.ref_start, r0, r1, start - R.max_span);
    *lo_out = a;
    *hi_out = d_wave_lower_bound<int32_t, int32_t>(R.ref_start, a, r1, end);
}
// d_fetch_range by the first TWO waves of a workgroup, one bound each (the bounds do not depend on each other: the array
// is sorted, so the upper bound may search the whole contig): five dependent loads instead of ten.  Call with
// threadIdx.x < 128; range[0], range[1] (LDS) are valid after the next barrier.  An empty interval gives 0, 0.
__device__ inline void d_fetch_range_2waves(const DevReads &R, int32_t contig, int32_t start, int32_t end, int64_t *range) {
    const int w = threadIdx.x >> 6;
    int64_t v = 0;
    if (start < end) {
        const int64_t r0 = R.contig_rs[contig], r1 = R.contig_rs[contig + 1];
        v = d_wave_lower_bound<int32_t, int32_t>(R.ref_start, r0, r1, w == 0 ? start - R.max_span : end);
    }
    if ((threadIdx.x & 63) == 0) range[w] = v;
}
__device__ inline int64_t d_pool_index_wave(const DevReads &R, int64_t r) {
    if (R.n_pool == R.n) return r;   // every read is pooled (pool_read is ascending and distinct, so it is the identity)
    const int64_t lo = d_wave_lower_bound<uint32_t, int64_t>(R.pool_read, 0, R.n_pool, r);
    if (lo >= R.n_pool || (int64_t)R.pool_read[lo] != r) return -1;
    return lo;
}
// htslib overlap test for a read already known to have ref_start < end
__device__ __forceinline__ bool d_overlaps(const DevReads &R, int64_t r, int32_t start) {
    int32_t rs = R.ref_start[r], re = R.ref_end[r];
    if (re <= rs) re = rs + 1;
    return re > start;
}

// ---- sequence pool ----------------------------------------------------------------------------------------------
__device__ inline int64_t d_pool_index(const DevReads &R, int64_t r) {
    if (R.n_pool == R.n) return r;   // every read is pooled: no search
    int64_t lo = 0, hi = R.n_pool;
    while (lo < hi) { int64_t m = (lo + hi) >> 1; if ((int64_t)R.pool_read[m] < r) lo = m + 1; else hi = m; }
    if (lo >= R.n_pool || (int64_t)R.pool_read[lo] != r) return -1;
    return lo;
}
// character of a 4-bit BAM base code ("=ACMGRSVTWYHKDBN") from two 64-bit immediates: a `const char codes[17]` indexed by a
// register lives in constant MEMORY, i.e. every use is a load with the latency of one (3 us under load at the end of a job)
__device__ __forceinline__ char d_code_char(int c) {
    const uint64_t lo = 0x565352474d43413dull;                        // '=' 'A' 'C' 'M' 'G' 'R' 'S' 'V' (low byte first)
    const uint64_t hi = 0x4e42444b48595754ull;                        // 'T' 'W' 'Y' 'H' 'K' 'D' 'B' 'N'
    return (char)(((c & 8) ? hi : lo) >> (8 * (c & 7)));
}
__device__ __forceinline__ uint8_t d_pool_base(const DevReads &R, uint64_t k) { return (R.pool_bases[k >> 1] >> ((~k & 1) << 2)) & 15; }

// ---- block-wide helpers (256 threads = 4 waves) -----------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { uint32_t t = __shfl_up(v, d, 64); if (lane >= d) v += t; }
    return v;
}
// exclusive scan of one value per thread over the block; *total gets the block sum.  `sh` = 8 uint32 of LDS.
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *sh, uint32_t *total) {
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    uint32_t inc = wave_incl_scan(v);
    __syncthreads();
    if (lane == 63) sh[w] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
    for (int i = 0; i < nw; i++) { uint32_t s = sh[i]; if (i < w) base += s; tot += s; }
    *total = tot;
    return base + inc - v;
}

// 64-bit variant (sh = 16 x uint64 of LDS for up to 1024 threads)
__device__ __forceinline__ unsigned long long block_excl_scan64(unsigned long long v, unsigned long long *sh, unsigned long long *total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    unsigned long long inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const unsigned long long t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
    __syncthreads();
    if (lane == 63) sh[w] = inc;
    __syncthreads();
    unsigned long long base = 0, tot = 0;
    for (int i = 0; i < nw; i++) { const unsigned long long s = sh[i]; if (i < w) base += s; tot += s; }
    *total = tot;
    return base + inc - v;
}
