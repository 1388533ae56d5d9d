// mfk_findflanks.hip -- the findflanks pipeline of libmfk.so (replaces SoftclipParser driven as in the reference's
// _findflanks, findflanks.py:24-51).  Kernels, in call order:
//
//   k_scan_clips              parse_softclips (findflanks.py:100-121): streams the flags column (1 B/read, 16 reads per
//                             lane per load) and lists the lenient-clipped reads
//   k_clip_events             read filters for the listed reads and one 64-bit event key (gpos | side | read) per clip,
//                             staged in LDS, one global reservation per workgroup; the sort that follows makes the
//                             order deterministic (keys are unique)
//   rocPRIM radix sort        events -> (site, side, BAM order)
//   k_event_info              per event: site head, first-of-its-query-name (link-chain walk, quirk Q3), meets_minlength
//   k_site_accumulate         per-site distinct-name counts and flags (SoftclipSite, findflanks.py:727-745, 797-801)
//   k_filter_minlength/_mincount/k_mindistance   findflanks.py:124-200 (+ misc.takeClosest*, strict neighbours, Q5)
//   k_compact_sites           surviving sites -> dense "active" table
//   k_classify                parse_unclipped_read_info (findflanks.py:349-509): one workgroup per active site joins the
//                             reads overlapping [pos-1,pos+2) and counts distinct names per category in LDS
//   k_ratio_filter            filter_softclips_count_ratios (findflanks.py:203-239), fp64 like Python
//   run_consensus             mfk_consensus.hip
//   k_consensus_filters       filter_consensus_sequences_minlength/_mincount/filter_multiple (findflanks.py:241-342)
#include "mfk_host.h"
#include <stdlib.h>

#define SCAN_THREADS 256
#define SCAN_MAX_BLOCKS 2048

// ---- K1: clip events -------------------------------------------------------------------------------------------
// parse_softclips (findflanks.py:100-121) in two kernels.
//   k_scan_clips   streams the flags column (1 B/read; each lane loads 16 reads as one dwordx4, the next tile's load is
//                  issued before this tile is processed) and appends the lenient-clipped reads (~1.5 %) to a list,
//                  one reservation per workgroup and tile (block-wide prefix sum of the per-lane counts).
//   k_clip_events  one thread per listed read: gathers the columns its clip kinds need (all loads issued together),
//                  runs passes_read_filters and emits one 64-bit key (gpos | side | read) per clip, staged in LDS and
//                  flushed with one reservation per workgroup.
// Clipped reads pile up at the insertion sites (a 1000x site puts ~1000 of them into one 4096-read tile); with the list
// in between, those gathers are spread over the whole GPU instead of serialising inside one workgroup.  The order of
// list and keys is arbitrary: keys are unique and the sort that follows fixes the order.
#define EV_STAGE 2048
#define SCAN_TILE (SCAN_THREADS * 16)

__device__ __forceinline__ uint4 d_load_flags16(const DevReads &R, int64_t g, int64_t ngroups) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (g < ngroups) {
        const int64_t r0 = g << 4;
        if (r0 + 16 <= R.n) v = *reinterpret_cast<const uint4 *>(R.flags + r0);
        else {
            uint32_t w[4] = {0, 0, 0, 0};
            for (int k = 0; k < (int)(R.n - r0); k++) w[k >> 2] |= (uint32_t)R.flags[r0 + k] << ((k & 3) * 8);   // reads past the end stay 0 = unclipped
            v = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
    return v;
}

// list entry: read index | flags << 32.  With the packer's dense clip records (mfk_reads.clip_rec) the low word is the
// read's index in THAT table instead (clip_tile[tile] clipped reads lie before this 4096-read tile, and the in-tile prefix
// sum gives the read's rank among the tile's clipped reads), and the entry is written AT that index: one flat list in read
// order, no reservations.  Without the records the list has LIST_SEGS segments of seg_cap entries, each with its own counter
// (a cache line apart): a workgroup reserves room with one atomic per tile, and thousands of those on ONE address would
// serialise in the L2 and cost more than the whole stream.
#define LIST_SEGS 64
#define SEG_PAD 32     // counters 128 bytes apart
// A workgroup takes SCAN_SUB consecutive 4096-read tiles per round: one block-wide prefix sum (the four per-tile counts of a
// lane packed into one 64-bit word, 16 bits each: a tile holds at most 4096 clipped reads) and ONE list reservation for the
// 16 k reads.  The reservation is a returning atomic, a round trip that cannot be hidden behind the next round's loads; with
// one tile per round it was most of the kernel (C5: 17 k tiles, 8 rounds per workgroup, 54 us for 69 MB).
#define SCAN_SUB 4
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_clips(DevReads R, uint64_t *list, uint32_t *seg_count, uint32_t seg_cap) {
    __shared__ unsigned long long sh[8];
    __shared__ uint32_t s_base;
    const uint32_t seg = blockIdx.x % LIST_SEGS;
    if (!R.clip_rec) list += (size_t)seg * seg_cap;
    const int64_t ngroups = (R.n + 15) >> 4;
    const int64_t ntiles = (ngroups + SCAN_THREADS - 1) / SCAN_THREADS;          // 4096-read tiles
    const int64_t nrounds = (ntiles + SCAN_SUB - 1) / SCAN_SUB;
    uint4 nxt[SCAN_SUB];
#pragma unroll
    for (int s = 0; s < SCAN_SUB; s++) nxt[s] = d_load_flags16(R, ((int64_t)blockIdx.x * SCAN_SUB + s) * SCAN_THREADS + threadIdx.x, ngroups);
    for (int64_t round = blockIdx.x; round < nrounds; round += gridDim.x) {
        uint4 v[SCAN_SUB];
#pragma unroll
        for (int s = 0; s < SCAN_SUB; s++) v[s] = nxt[s];
        if (round + gridDim.x < nrounds) {
#pragma unroll
            for (int s = 0; s < SCAN_SUB; s++) nxt[s] = d_load_flags16(R, ((round + gridDim.x) * SCAN_SUB + s) * SCAN_THREADS + threadIdx.x, ngroups);
        }
        // 16-bit map of this lane's lenient-clipped reads in each of the tiles
        uint32_t cmap[SCAN_SUB];
        unsigned long long packed = 0;
#pragma unroll
        for (int s = 0; s < SCAN_SUB; s++) {
            const uint32_t w[4] = {v[s].x, v[s].y, v[s].z, v[s].w};
            uint32_t m = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                uint32_t t = w[j];
                t = (t | (t >> 1) | (t >> 2) | (t >> 3)) & 0x01010101u;
                m |= ((t & 1u) | ((t >> 7) & 2u) | ((t >> 14) & 4u) | ((t >> 21) & 8u)) << (4 * j);
            }
            cmap[s] = m;
            packed |= (unsigned long long)__popc(m) << (16 * s);
        }
        unsigned long long totals;
        const unsigned long long pre = block_excl_scan64(packed, sh, &totals);   // per tile: clipped reads of the lanes before this one
        if (totals && R.clip_rec) {
            // Dense clip records: a read's rank in the table (clipped reads before its tile + before it in the tile) IS its list
            // slot.  No reservation, no counter, nothing the round has to wait for; the list comes out in read order.
#pragma unroll
            for (int s = 0; s < SCAN_SUB; s++) {
                uint32_t m = cmap[s];
                if (!m) continue;
                uint32_t c = R.clip_tile[round * SCAN_SUB + s] + ((uint32_t)(pre >> (16 * s)) & 0xFFFFu);
                const uint64_t lo = (uint64_t)v[s].x | ((uint64_t)v[s].y << 32), hi = (uint64_t)v[s].z | ((uint64_t)v[s].w << 32);
                while (m) {
                    const int k = __builtin_ctz(m);
                    m &= m - 1;
                    const uint64_t f = ((k < 8 ? lo : hi) >> ((k & 7) * 8)) & 0xFF;
                    if ((int64_t)c < R.n_clip) list[c] = (uint64_t)c | (f << 32);
                    c++;
                }
            }
        } else if (totals) {   // block-uniform
            uint32_t tile_base[SCAN_SUB], total = 0;
#pragma unroll
            for (int s = 0; s < SCAN_SUB; s++) { tile_base[s] = total; total += (uint32_t)(totals >> (16 * s)) & 0xFFFFu; }
            if (threadIdx.x == 0) s_base = atomicAdd(&seg_count[seg * SEG_PAD], total);
            __syncthreads();
            const uint32_t base = s_base;
#pragma unroll
            for (int s = 0; s < SCAN_SUB; s++) {
                uint32_t slot = base + tile_base[s] + ((uint32_t)(pre >> (16 * s)) & 0xFFFFu), m = cmap[s];
                const int64_t r0 = ((round * SCAN_SUB + s) * SCAN_THREADS + threadIdx.x) << 4;
                const uint64_t lo = (uint64_t)v[s].x | ((uint64_t)v[s].y << 32), hi = (uint64_t)v[s].z | ((uint64_t)v[s].w << 32);
                while (m) {
                    const int k = __builtin_ctz(m);
                    m &= m - 1;
                    const uint64_t f = ((k < 8 ? lo : hi) >> ((k & 7) * 8)) & 0xFF;
                    if (slot < seg_cap) list[slot] = (uint64_t)(r0 + k) | (f << 32);
                    slot++;
                }
            }
            __syncthreads();   // s_base is reused by the next round (sh is protected by the barrier at the head of the scan)
        }
    }
}

// What k_clip_events / k_clip_events_bkt need of one listed read.  Two sources with the same content: the SoA columns (one
// sparse gather per column and read: every one costs a whole 64-byte sector of HBM traffic) or, when the caller passed
// them, the packer's dense 16-byte clip records (four to a sector, neighbours in the list are neighbours in the table).
struct ClipFields { uint32_t r, base; int32_t rs, re; int mq; bool ok, ial_bad; };
__device__ __forceinline__ ClipFields d_clip_fields(const DevReads &R, const DevParams &P, uint64_t e, int lk, int rk) {
    ClipFields c;
    const bool both = lk && rk;
    c.ok = true;
    if (R.clip_rec) {
        const uint32_t ci = (uint32_t)e;
        c.ok = (int64_t)ci < R.n_clip;
        const uint4 q = c.ok ? reinterpret_cast<const uint4 *>(R.clip_rec)[ci] : make_uint4(0, 0, 0, 0);
        c.r = q.x; c.rs = (int32_t)q.y; c.re = (int32_t)q.z; c.mq = (int)((q.w >> 16) & 0xFF);
        const uint32_t cid = q.w & 0xFFFFu;
        c.ok = c.ok && cid < (uint32_t)R.n_contigs && (int64_t)c.r < R.n;
        c.base = c.ok ? R.contig_base[cid] : 0;
        c.ial_bad = false;
        if (both && c.ok) {
            if (P.minial <= 255) c.ial_bad = (int)(q.w >> 24) < P.minial;   // the record holds min(inner length, 255)
            else {
                const int fo = R.first_op_len[c.r], lo = R.last_op_len[c.r];
                c.ial_bad = (int)R.qlen[c.r] - (rk == 1 ? lo : 1) - (lk == 1 ? fo : 1) < P.minial;
            }
        }
        return c;
    }
    const int64_t r = (int64_t)(e & 0x7fffffffu);
    c.r = (uint32_t)r;
    // Only the columns this read's clip kinds need, all issued together: none of these loads depends on another
    c.mq = R.mapq[r];
    const int ql = both ? R.qlen[r] : 0, fo = both ? R.first_op_len[r] : 0, lo = both ? R.last_op_len[r] : 0;
    c.rs = lk ? R.ref_start[r] : 0; c.re = rk ? R.ref_end[r] : 0;
    c.base = R.contig_base[R.contig_id[r]];
    const int llen = lk == 1 ? fo : (lk == 2 ? 1 : 0), rlen = rk == 1 ? lo : (rk == 2 ? 1 : 0);   // d_llen / d_rlen
    c.ial_bad = both && ql - rlen - llen < P.minial;                                               // d_passes
    return c;
}

// grid (x, LIST_SEGS): workgroup (x, seg) takes every gridDim.x-th 256-entry slice of list segment seg
__global__ __launch_bounds__(256) void k_clip_events(DevReads R, DevParams P, const uint64_t *list, const uint32_t *seg_count, uint32_t seg_cap,
                                                     uint64_t *ev, uint32_t *ev_count, uint32_t cap) {
    __shared__ uint64_t stage[EV_STAGE];
    __shared__ uint32_t s_n, s_base;
    uint32_t n, wg = blockIdx.x, nwg = gridDim.x;
    if (R.clip_rec) { n = (uint32_t)R.n_clip; wg = blockIdx.y * gridDim.x + blockIdx.x; nwg = gridDim.x * gridDim.y; }   // one flat list (k_scan_clips)
    else { const uint32_t cnt = seg_count[blockIdx.y * SEG_PAD]; n = cnt < seg_cap ? cnt : seg_cap; list += (size_t)blockIdx.y * seg_cap; }
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    for (uint32_t b0 = wg * 256; b0 < n; b0 += nwg * 256) {   // block-uniform trip count
        const uint32_t i = b0 + threadIdx.x;
        if (i < n) {
            const uint64_t e = list[i];
            const uint8_t f = (uint8_t)(e >> 32);
            const int lk = d_lkind(f), rk = d_rkind(f);
            const ClipFields c = d_clip_fields(R, P, e, lk, rk);
            if (c.ok && c.mq >= P.minq && !c.ial_bad) {                                                   // d_passes
                if (rk) {   // right clip -> 5p site (findflanks.py:117-118, 594-597); d_rsite
                    const int32_t site = rk == 1 ? c.re : c.re - 1;
                    stage[atomicAdd(&s_n, 1u)] = ((uint64_t)(c.base + (uint32_t)(site + 1)) << 32) | (uint64_t)c.r;
                }
                if (lk) {   // left clip -> 3p site (findflanks.py:120-121, 599-602); d_lsite
                    const int32_t site = lk == 1 ? c.rs - 1 : c.rs;
                    stage[atomicAdd(&s_n, 1u)] = ((uint64_t)(c.base + (uint32_t)(site + 1)) << 32) | (1ull << 31) | (uint64_t)c.r;
                }
            }
        }
        __syncthreads();
        if (s_n > EV_STAGE - 512) {   // the next round adds at most 512 keys
            const uint32_t m = s_n;
            if (threadIdx.x == 0) s_base = atomicAdd(ev_count, m);
            __syncthreads();
            for (uint32_t k = threadIdx.x; k < m; k += 256) if (s_base + k < cap) ev[s_base + k] = stage[k];
            __syncthreads();
            if (threadIdx.x == 0) s_n = 0;
        }
        __syncthreads();
    }
    const uint32_t m = s_n;
    if (threadIdx.x == 0 && m) s_base = atomicAdd(ev_count, m);
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < m; k += 256) if (s_base + k < cap) ev[s_base + k] = stage[k];
}

// ---- K1b: events straight into position buckets, sorted bucket by bucket in LDS ------------------------------------------
// The event keys have to end up sorted by (gpos, side, read).  A device-wide sort of a few hundred thousand 64-bit keys is
// ~17 short launches of rocPRIM's merge sort (100 us at C2, the largest single item of the step).  Instead the flat site
// axis is cut into B equal position ranges ("buckets", B a power of two chosen from the read count so that a bucket holds
// a few hundred keys); k_clip_events_bkt drops every key into its bucket (fixed capacity BKT_CAP, one returning atomic
// per run of equal buckets inside a wave: the clips of a deep site arrive together and would otherwise queue on one
// counter), and k_bucket_sort sorts each bucket in LDS (bitonic, one workgroup per bucket) and writes it at the prefix sum
// of the bucket counts.  Buckets are monotone in gpos, so the concatenation is the sorted list.  A bucket that overflows
// (a site deeper than BKT_CAP clips) makes the host fall back to the list + rocPRIM path: nothing is lost, only time.
#define BKT_CAP 4096

// Where a lane's key goes inside its run of equal buckets.  A run = the lanes that hold a key for the same bucket with no
// lane holding a key for ANOTHER bucket between them (lanes without a key of this kind do not break a run: 5p and 3p clips
// alternate along a wave, and a run that ended at every gap meant nearly one reservation per key).  The run's first lane
// reserves room for the whole run: one returning atomic per run.  With the flat list in read order the keys of a wave are
// sorted by position, so a wave makes one or two reservations per kind; the device's rate of returning atomics, not its
// bandwidth, was what bounded this kernel.
struct BktRun { int hl, len, off; bool head; };
__device__ __forceinline__ BktRun d_bucket_run(bool has, uint32_t b, int lane) {
    const unsigned long long hasmask = __ballot(has);
    const unsigned long long below = hasmask & ((1ull << lane) - 1ull);               // lanes with a key before this one
    const int prev = below ? 63 - __clzll((long long)below) : lane;
    const uint32_t prev_b = __shfl(b, prev, 64);
    BktRun r;
    r.head = has && (!below || prev_b != b);
    const unsigned long long headmask = __ballot(r.head);
    r.hl = 0; r.len = 0; r.off = 0;
    if (has) {
        const unsigned long long heads_here = headmask & ((2ull << lane) - 1ull);      // run heads at or below this lane (never empty)
        r.hl = 63 - __clzll((long long)heads_here);
        const unsigned long long later = r.hl == 63 ? 0ull : (headmask & ~((2ull << r.hl) - 1ull));
        const int end = later ? __ffsll((long long)later) - 1 : 64;                   // first lane of the next run
        const unsigned long long from_head = ~((1ull << r.hl) - 1ull);
        const unsigned long long run = hasmask & from_head & (end == 64 ? ~0ull : ((1ull << end) - 1ull));
        r.len = __popcll(run);
        r.off = __popcll(run & ((1ull << lane) - 1ull));
    }
    return r;
}

// One sweep of k_clip_events_bkt over the dense clip records, U entries per thread.  The kernel is a chain of dependent
// round trips (list entry -> record -> contig base -> bucket reservation -> store) at ~3 us each under load, and a sample
// with millions of clipped reads needs several sweeps per workgroup: with U entries per thread the trips of U entries are
// in flight together.  Every load is unconditional (indices clamped, validity carried in a flag), so nothing keeps the
// compiler from issuing a stage's loads back to back.
template <int U>
__device__ __forceinline__ void d_events_sweep_rec(const DevReads &R, const DevParams &P, const uint64_t *list, uint32_t n, uint32_t b0,
                                                   uint64_t *bkt, uint32_t *bkt_count, uint32_t B, uint64_t bmul, int lane) {
    uint64_t e[U]; bool in[U];
#pragma unroll
    for (int u = 0; u < U; u++) { const uint32_t i = b0 + (uint32_t)u * 256u + threadIdx.x; in[u] = i < n; e[u] = list[in[u] ? i : 0u]; }
    uint4 q[U];
#pragma unroll
    for (int u = 0; u < U; u++) { const uint32_t ci = (uint32_t)e[u]; in[u] = in[u] && (int64_t)ci < R.n_clip; q[u] = reinterpret_cast<const uint4 *>(R.clip_rec)[in[u] ? ci : 0u]; }
    uint32_t base[U];
#pragma unroll
    for (int u = 0; u < U; u++) { const uint32_t cid = q[u].w & 0xFFFFu; in[u] = in[u] && cid < (uint32_t)R.n_contigs && (int64_t)q[u].x < R.n; base[u] = R.contig_base[in[u] ? cid : 0u]; }
    uint64_t k5[U], k3[U]; bool h5[U], h3[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const uint8_t f = (uint8_t)(e[u] >> 32);
        const int lk = d_lkind(f), rk = d_rkind(f);
        const uint32_t r = q[u].x;
        const int32_t rs = (int32_t)q[u].y, re = (int32_t)q[u].z;
        bool ial_bad = false;
        if (lk && rk && in[u]) {
            if (P.minial <= 255) ial_bad = (int)(q[u].w >> 24) < P.minial;   // the record holds min(inner length, 255)
            else ial_bad = (int)R.qlen[r] - (rk == 1 ? (int)R.last_op_len[r] : 1) - (lk == 1 ? (int)R.first_op_len[r] : 1) < P.minial;
        }
        const bool ok = in[u] && (int)((q[u].w >> 16) & 0xFFu) >= P.minq && !ial_bad;
        h5[u] = ok && rk; h3[u] = ok && lk;
        const int32_t s5 = rk == 1 ? re : re - 1, s3 = lk == 1 ? rs - 1 : rs;
        k5[u] = h5[u] ? (((uint64_t)(base[u] + (uint32_t)(s5 + 1)) << 32) | (uint64_t)r) : 0ull;
        k3[u] = h3[u] ? (((uint64_t)(base[u] + (uint32_t)(s3 + 1)) << 32) | (1ull << 31) | (uint64_t)r) : 0ull;
    }
    uint32_t b5[U], b3[U], a5[U], a3[U]; int hl5[U], hl3[U], of5[U], of3[U];
#pragma unroll
    for (int u = 0; u < U; u++) {   // every reservation of the sweep is issued before any result is needed
        b5[u] = bmul ? (uint32_t)(((k5[u] >> 32) * bmul) >> 32) : (uint32_t)(k5[u] >> 32);
        b3[u] = bmul ? (uint32_t)(((k3[u] >> 32) * bmul) >> 32) : (uint32_t)(k3[u] >> 32);
        if (b5[u] >= B) b5[u] = B - 1;
        if (b3[u] >= B) b3[u] = B - 1;
        const BktRun r5 = d_bucket_run(h5[u], b5[u], lane), r3 = d_bucket_run(h3[u], b3[u], lane);
        hl5[u] = r5.hl; hl3[u] = r3.hl; of5[u] = r5.off; of3[u] = r3.off;
        a5[u] = 0; a3[u] = 0;
        if (r5.head) a5[u] = atomicAdd(&bkt_count[b5[u]], (uint32_t)r5.len);
        if (r3.head) a3[u] = atomicAdd(&bkt_count[b3[u]], (uint32_t)r3.len);
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
        const uint32_t base5 = __shfl(a5[u], hl5[u], 64), base3 = __shfl(a3[u], hl3[u], 64);
        if (h5[u]) { const uint32_t slot = base5 + (uint32_t)of5[u]; if (slot < BKT_CAP) bkt[(size_t)b5[u] * BKT_CAP + slot] = k5[u]; }
        if (h3[u]) { const uint32_t slot = base3 + (uint32_t)of3[u]; if (slot < BKT_CAP) bkt[(size_t)b3[u] * BKT_CAP + slot] = k3[u]; }
    }
}

// same grid and per-read logic as k_clip_events
__global__ __launch_bounds__(256) void k_clip_events_bkt(DevReads R, DevParams P, const uint64_t *list, const uint32_t *seg_count, uint32_t seg_cap,
                                                         uint64_t *bkt, uint32_t *bkt_count, uint32_t B, uint64_t bmul) {
    const int lane = threadIdx.x & 63;
    uint32_t n, wg = blockIdx.x, nwg = gridDim.x;
    if (R.clip_rec) {   // one flat list in read order (k_scan_clips); all branches here are block-uniform
        n = (uint32_t)R.n_clip; wg = blockIdx.y * gridDim.x + blockIdx.x; nwg = gridDim.x * gridDim.y;
        if (n > nwg * 256u) {   // several sweeps per workgroup: four entries per thread in flight
            for (uint32_t b0 = wg * 1024u; b0 < n; b0 += nwg * 1024u) d_events_sweep_rec<4>(R, P, list, n, b0, bkt, bkt_count, B, bmul, lane);
            return;
        }
    } else { const uint32_t cnt = seg_count[blockIdx.y * SEG_PAD]; n = cnt < seg_cap ? cnt : seg_cap; list += (size_t)blockIdx.y * seg_cap; }
    for (uint32_t b0 = wg * 256; b0 < n; b0 += nwg * 256) {   // block-uniform trip count
        const uint32_t i = b0 + threadIdx.x;
        uint64_t k5 = 0, k3 = 0;
        bool h5 = false, h3 = false;
        if (i < n) {
            const uint64_t e = list[i];
            const uint8_t f = (uint8_t)(e >> 32);
            const int lk = d_lkind(f), rk = d_rkind(f);
            const ClipFields c = d_clip_fields(R, P, e, lk, rk);
            if (c.ok && c.mq >= P.minq && !c.ial_bad) {
                if (rk) { const int32_t site = rk == 1 ? c.re : c.re - 1; k5 = ((uint64_t)(c.base + (uint32_t)(site + 1)) << 32) | (uint64_t)c.r; h5 = true; }
                if (lk) { const int32_t site = lk == 1 ? c.rs - 1 : c.rs; k3 = ((uint64_t)(c.base + (uint32_t)(site + 1)) << 32) | (1ull << 31) | (uint64_t)c.r; h3 = true; }
            }
        }
        // bucket = position range of the flat axis.  Any monotone map of gpos onto [0, B) will do (the buckets are sorted one by
        // one and concatenated), so instead of gpos * B / span (two 64-bit divisions per thread, a sixth of this kernel's time)
        // it is gpos * floor(2^32 B / span) >> 32 with the factor made by the host; an axis shorter than B is its own bucket index.
        uint32_t b5 = bmul ? (uint32_t)(((k5 >> 32) * bmul) >> 32) : (uint32_t)(k5 >> 32), b3 = bmul ? (uint32_t)(((k3 >> 32) * bmul) >> 32) : (uint32_t)(k3 >> 32);
        if (b5 >= B) b5 = B - 1;
        if (b3 >= B) b3 = B - 1;
        // both reservations are issued before either result is needed: one atomic round trip per sweep, not two
        const BktRun r5 = d_bucket_run(h5, b5, lane), r3 = d_bucket_run(h3, b3, lane);
        uint32_t base5 = 0, base3 = 0;
        if (r5.head) base5 = atomicAdd(&bkt_count[b5], (uint32_t)r5.len);
        if (r3.head) base3 = atomicAdd(&bkt_count[b3], (uint32_t)r3.len);
        base5 = __shfl(base5, r5.hl, 64);
        base3 = __shfl(base3, r3.hl, 64);
        if (h5) { const uint32_t slot = base5 + (uint32_t)r5.off; if (slot < BKT_CAP) bkt[(size_t)b5 * BKT_CAP + slot] = k5; }
        if (h3) { const uint32_t slot = base3 + (uint32_t)r3.off; if (slot < BKT_CAP) bkt[(size_t)b3 * BKT_CAP + slot] = k3; }
    }
}

// one workgroup per bucket: output offset = sum of the counts before it, bitonic sort in LDS, coalesced write
// Deep buckets (a 1000x site puts a few thousand keys into one) take several times as long as the others and set the
// kernel's duration when they are dispatched late: the host, which has read the counts anyway, lists the first
// DEEP_MAX of them (ascending) and the first workgroups of the grid take those; the workgroup that owns a listed bucket
// by position leaves it alone.
#define DEEP_MIN 512u
#define DEEP_MAX 60
struct DeepBuckets { uint16_t n, last; uint16_t id[DEEP_MAX]; };
#define BSORT_THREADS 512
// With the sorted keys of a whole position range in LDS the per-event information (k_event_info below, which the list +
// device-wide sort path still uses) costs almost no memory traffic here: an event is the first of its query name at its
// site-side unless an EARLIER event of the same site-side belongs to a read of its link chain, and "same site-side" is a
// run of neighbours in this bucket.  So an event with no such neighbour before it (every stray one-read clip, two thirds of
// all events of a deep isolate) fetches nothing but its CIGAR-end length, and the others look their chain reads up by binary
// search in LDS instead of gathering six per-read arrays of the mate.
__global__ __launch_bounds__(BSORT_THREADS) void k_bucket_sort(const uint64_t *bkt, const uint32_t *bkt_count, uint32_t B, uint64_t *out, DeepBuckets D,
                                                               DevReads R, DevParams P, uint8_t *info, uint32_t *head) {
    __shared__ uint64_t keys[BKT_CAP];
    __shared__ uint32_t red[BSORT_THREADS / 64];
    const bool listed = blockIdx.x < D.n;
    const uint32_t b = listed ? D.id[blockIdx.x] : blockIdx.x - D.n;
    const uint32_t n = bkt_count[b];
    if (n == 0 || n > BKT_CAP) return;   // block-uniform (an overflow never gets here: the host checks the counts first)
    if (!listed && D.n && n > DEEP_MIN && b <= D.last) return;   // taken by one of the first workgroups
    uint32_t part = 0;
    for (uint32_t j = threadIdx.x; j < b; j += BSORT_THREADS) part += bkt_count[j];
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    uint32_t npad = 64;
    while (npad < n) npad <<= 1;
    const uint64_t *src = bkt + (size_t)b * BKT_CAP;
    for (uint32_t i = threadIdx.x; i < npad; i += BSORT_THREADS) keys[i] = i < n ? src[i] : ~0ull;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < BSORT_THREADS / 64; w++) base += red[w];
    // Bitonic network.  A stage with distance j >= 128 pairs keys of different waves' pairs: block barrier per stage.  For
    // j <= 64 the 64 consecutive pairs of a wave stay inside that wave's own 128 keys, so all remaining stages of a merge
    // step run without block barriers (LDS operations of one wave execute in order): 21 barriers instead of 66 for 2048 keys.
    for (uint32_t k = 2; k <= npad; k <<= 1) {
        uint32_t j = k >> 1;
        for (; j >= 128; j >>= 1) {
            for (uint32_t t = threadIdx.x; t < (npad >> 1); t += BSORT_THREADS) {
                const uint32_t i = 2 * t - (t & (j - 1)), l = i + j;
                const uint64_t x = keys[i], y = keys[l];
                const bool up = (i & k) == 0;
                if ((x > y) == up) { keys[i] = y; keys[l] = x; }
            }
            __syncthreads();
        }
        for (uint32_t t = threadIdx.x; t < (npad >> 1); t += BSORT_THREADS)
            for (uint32_t jj = j; jj > 0; jj >>= 1) {
                const uint32_t i = 2 * t - (t & (jj - 1)), l = i + jj;
                const uint64_t x = keys[i], y = keys[l];
                const bool up = (i & k) == 0;
                if ((x > y) == up) { keys[i] = y; keys[l] = x; }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // this wave's exchanges are complete before its next stage reads
            }
        __syncthreads();
    }
    for (uint32_t i = threadIdx.x; i < n; i += BSORT_THREADS) {
        const uint64_t key = keys[i];
        out[base + i] = key;
        // info bits as in k_event_info.  All events of a position are in this bucket, so the bucket's first key starts a site.
        const uint32_t gpos = (uint32_t)(key >> 32);
        const int side = (int)((key >> 31) & 1);
        const int64_t r = (int64_t)(key & 0x7fffffffu);
        const bool h = i == 0 || (uint32_t)(keys[i - 1] >> 32) != gpos;
        bool first = true;
        if (i > 0 && (keys[i - 1] >> 31) == (key >> 31)) {   // there are earlier events of this site-side
            const uint64_t run_key = key & ~(uint64_t)0x7fffffffu;
            uint32_t lb = 0;
            { uint32_t hi = i; while (lb < hi) { const uint32_t m = (lb + hi) >> 1; if (keys[m] < run_key) lb = m + 1; else hi = m; } }
            int64_t prev = r;
            for (int64_t q = R.link[r]; q >= 0 && q < prev; prev = q, q = R.link[q]) {   // strictly decreasing chain
                // read q has a clip event here (d_clip_event true at this gpos) iff its key is in the run
                const uint64_t want = run_key | (uint64_t)q;
                uint32_t a = lb, b = i;
                while (a < b) { const uint32_t m = (a + b) >> 1; if (keys[m] < want) a = m + 1; else b = m; }
                if (a < i && keys[a] == want) { first = false; break; }
            }
        }
        const bool meets = side == 0 ? ((int)R.last_op_len[r] >= P.minlen) : ((int)R.first_op_len[r] >= P.minlen);
        info[base + i] = (uint8_t)((h ? 1 : 0) | (first ? 2 : 0) | (meets ? 4 : 0) | (side ? 8 : 0));
        head[base + i] = h ? 1u : 0u;
    }
}

// ---- K2: per-event information ---------------------------------------------------------------------------------
// info bits: 1 site head, 2 first read of its query name at this site-side, 4 meets_minlength, 8 side (0 = 5p, 1 = 3p)
__global__ void k_event_info(DevReads R, DevParams P, const uint64_t *keys, uint32_t E, uint8_t *info, uint32_t *head) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= E) return;
    uint64_t key = keys[i];
    uint32_t gpos = (uint32_t)(key >> 32);
    int side = (int)((key >> 31) & 1);
    int64_t r = (int64_t)(key & 0x7fffffffu);
    bool h = i == 0 || (uint32_t)(keys[i - 1] >> 32) != gpos;
    uint16_t contig = R.contig_id[r];
    uint32_t base = R.contig_base[contig];
    bool first = true;
    int64_t prev = r;
    for (int64_t q = R.link[r]; q >= 0 && q < prev; prev = q, q = R.link[q]) {   // strictly decreasing chain
        int32_t site;
        if (R.contig_id[q] == contig && d_clip_event(R, P, q, side, &site) && base + (uint32_t)(site + 1) == gpos) { first = false; break; }
    }
    // meets_minlength_right / _left: raw length of the last / first CIGAR op (findflanks.py:604-614, quirk Q1)
    bool meets = side == 0 ? ((int)R.last_op_len[r] >= P.minlen) : ((int)R.first_op_len[r] >= P.minlen);
    info[i] = (uint8_t)((h ? 1 : 0) | (first ? 2 : 0) | (meets ? 4 : 0) | (side ? 8 : 0));
    head[i] = h ? 1u : 0u;
}

// ---- K3: site table ----------------------------------------------------------------------------------------------
struct SiteTab {
    uint32_t *gpos; int32_t *contig; uint32_t *cnt2;   // cnt2[2*s + side] distinct query names
    uint32_t *sflags;                                  // bit0 meets_minlength_5p, bit1 meets_minlength_3p
    uint32_t *ev_begin, *ev_mid, *ev_end;              // event ranges: [begin, mid) 5p events, [mid, end) 3p events
    uint32_t *st;                                      // bit0 keep_5p, bit1 keep_3p, bit2 alive
};

// Events are sorted, so the events of one (site, side) are neighbours: a segmented scan inside each wave adds them up
// and only the last lane of a run touches the site's counters (a 1000x site would otherwise put ~500 atomics on one word).
__global__ __launch_bounds__(256) void k_site_accumulate(DevReads R, const uint64_t *keys, const uint8_t *info, const uint32_t *head_scan, uint32_t E, SiteTab T) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool valid = i < E;
    uint32_t seg = 0xFFFFFFFFu, add = 0, orb = 0, s = 0;
    int side = 0;
    if (valid) {
        const uint8_t in = info[i];
        s = head_scan[i] + (in & 1) - 1;   // exclusive scan of heads + own head - 1
        const uint64_t key = keys[i];
        side = (in >> 3) & 1;
        if (in & 1) {
            T.gpos[s] = (uint32_t)(key >> 32);
            T.contig[s] = (int32_t)R.contig_id[key & 0x7fffffffu];
            T.ev_begin[s] = i;
            if (s > 0) T.ev_end[s - 1] = i;
        }
        if (i == E - 1) T.ev_end[s] = E;
        if (side == 1 && ((in & 1) || !((info[i - 1] >> 3) & 1))) T.ev_mid[s] = i;
        seg = 2 * s + (uint32_t)side;
        add = (in & 2) ? 1u : 0u;
        orb = (in & 4) ? 1u : 0u;
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t k2 = __shfl_up(seg, d, 64), a2 = __shfl_up(add, d, 64), o2 = __shfl_up(orb, d, 64);
        if (lane >= d && k2 == seg) { add += a2; orb |= o2; }
    }
    const uint32_t nxt = __shfl_down(seg, 1, 64);
    if (valid && (lane == 63 || nxt != seg)) {
        if (add) atomicAdd(&T.cnt2[seg], add);
        if (orb) atomicOr(&T.sflags[s], 1u << side);
    }
}

// site state + filter_softclips_minlength (findflanks.py:124-145) + filter_softclips_mincount (148-169) in one pass;
// `upto`: 0 = state only, 1 = + minlength, 2 = + mincount
__global__ __launch_bounds__(256) void k_site_filters(SiteTab T, const uint32_t *n_sites_p, int mincount, int upto, uint32_t *stats) {
    const uint32_t n = *n_sites_p;
    uint32_t c1 = 0, c2 = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (T.ev_mid[i] == 0xFFFFFFFFu) T.ev_mid[i] = T.ev_end[i];
        uint32_t s = 7;
        if (upto >= 1) {
            uint32_t m = T.sflags[i];
            if (!(m & 1)) s &= ~1u;
            if (!(m & 2)) s &= ~2u;
            if (!(s & 3)) s &= ~4u;
            c1 += (s >> 2) & 1;
        }
        if (upto >= 2 && (s & 4)) {
            if ((int64_t)T.cnt2[2 * i] < (int64_t)mincount) s &= ~1u;
            if ((int64_t)T.cnt2[2 * i + 1] < (int64_t)mincount) s &= ~2u;
            if (!(s & 3)) s &= ~4u;
            c2 += (s >> 2) & 1;
        }
        T.st[i] = s;
    }
    // one pair of atomics per workgroup (the grid is capped, see the launch)
    __shared__ uint32_t sh[2];
    if (threadIdx.x < 2) sh[threadIdx.x] = 0;
    __syncthreads();
    for (int o = 32; o > 0; o >>= 1) { c1 += __shfl_xor(c1, o, 64); c2 += __shfl_xor(c2, o, 64); }
    if ((threadIdx.x & 63) == 0) { if (c1) atomicAdd(&sh[0], c1); if (c2) atomicAdd(&sh[1], c2); }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (blockIdx.x == 0) stats[0] = n;
        if (sh[0]) atomicAdd(&stats[1], sh[0]);
        if (sh[1]) atomicAdd(&stats[2], sh[1]);
    }
}

// filter_softclips_mindistance findflanks.py:171-200.  A site with 5p clips needs a surviving site with 3p clips at a
// strictly smaller position within mindist (misc.takeClosestSmaller, has_nearby_3p_mate 551-558); a site with 3p clips
// needs a surviving 5p site strictly to the right (has_nearby_5p_mate 542-549).  Counts decide, not keep flags (Q5).
// The table is sorted by gpos with distinct positions; dead sites are skipped (they left the dict in an earlier round).
__global__ void k_mindistance(const uint32_t *gpos, const uint32_t *cnt2, const uint32_t *st_in, uint32_t *st_out, const uint32_t *n_p, uint32_t n_val, int mindist, uint32_t *stat, uint32_t *alive_flag) {
    uint32_t n = n_p ? *n_p : n_val;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    int alive = 0;
    if (i < n) {
        uint32_t s = st_in[i];
        if (s & 4) {
            uint32_t g = gpos[i];
            uint32_t md = mindist < 0 ? 0u : (uint32_t)mindist;
            if (cnt2[2 * i] > 0) {
                bool found = false;
                if (mindist >= 1)
                    for (int64_t j = (int64_t)i - 1; j >= 0 && g - gpos[j] <= md; j--)
                        if ((st_in[j] & 4) && cnt2[2 * j + 1] > 0) { found = true; break; }
                if (!found) s &= ~1u;
            }
            if (cnt2[2 * i + 1] > 0) {
                bool found = false;
                if (mindist >= 1)
                    for (uint32_t j = i + 1; j < n && gpos[j] - g <= md; j++)
                        if ((st_in[j] & 4) && cnt2[2 * j] > 0) { found = true; break; }
                if (!found) s &= ~2u;
            }
            if (!(s & 3)) s &= ~4u;
            alive = (s >> 2) & 1;
        }
        st_out[i] = s;
        if (alive_flag) alive_flag[i] = (s >> 2) & 1;
    }
    int c = __syncthreads_count(alive);
    if (threadIdx.x == 0 && c) atomicAdd(stat, (uint32_t)c);
}

__global__ void k_alive_flags(const uint32_t *st, const uint32_t *n_p, uint32_t cap, uint32_t *flag) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cap) return;
    flag[i] = (i < *n_p && (st[i] & 4)) ? 1u : 0u;
}

// active-site table (dense): everything later stages need, in site (gpos) order
struct ActTab {
    uint32_t *site, *gpos; int32_t *contig; uint32_t *cnt2, *st;
    uint32_t *counts;   // counts[8*k + c]: runthrough si5 si3 li5 li3 del updel downdel
};

__global__ void k_compact_sites(SiteTab T, const uint32_t *n_p, const uint32_t *scan, ActTab A) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= *n_p || !(T.st[i] & 4)) return;
    uint32_t k = scan[i];
    A.site[k] = i; A.gpos[k] = T.gpos[i]; A.contig[k] = T.contig[i];
    A.cnt2[2 * k] = T.cnt2[2 * i]; A.cnt2[2 * k + 1] = T.cnt2[2 * i + 1];
    A.st[k] = T.st[i];
}

// ---- K6: parse_unclipped_read_info findflanks.py:349-509 ----------------------------------------------------------
// One workgroup per active site.  Reads overlapping [pos-1, pos+2) (clamped, get_reads_at_site 399-415) that pass the
// read filters are classified at pos (six categories) and, deletions only, at pos-1 and pos+1 (380-396).  A read counts
// for a set only if no earlier read of the same query name (link chain) falls in the same set: that is len(set(names)).
// sets: 0 runthrough, 1-2 small insertion 5p/3p, 3-4 large insertion 5p/3p, 5 deletion (qi 0, at pos); 6 / 7 deletion at
// pos-1 / pos+1 (qi 1 / 2); recall only: 8 / 9 lenient right / left clip at pos (qi 3 / 4, recall.py:96-112)
__device__ __forceinline__ int d_set_of(const DevReads &R, const DevParams &P, int64_t r, uint8_t f, int32_t q, int qi) {
    if (qi == 3) return (d_rkind(f) && d_rsite(R, r, f) == q) ? 8 : -1;   // is_right_softclipped_lenient_at_site -> 5p
    if (qi == 4) return (d_lkind(f) && d_lsite(R, r, f) == q) ? 9 : -1;   // is_left_softclipped_lenient_at_site -> 3p
    if (d_clipped_at(R, r, f, q)) return -1;                      // findflanks.py:429-430
    int cat = d_classify(R, P, r, f, q, qi != 0);
    if (qi == 0) return cat - 1;                                  // CAT_NONE -> -1
    return cat == CAT_DELETION ? 5 + qi : -1;                     // 6 upstream, 7 downstream
}

// The columns of a read that nearly every test of the tally looks at, loaded TOGETHER (one round trip): the per-read predicates
// of mfk_device.h load what they need where they need it, behind the branch that decides whether they run, which makes
// a chain of five dependent round trips per read and five more for its mate.  MFK_KEEP pins the loads before the first branch
// (the compiler otherwise sinks each load to its first use).
struct RdLite { int32_t rs, re; uint32_t f, mq; };
#define MFK_KEEP(v) asm volatile("" : "+v"(v))
__device__ __forceinline__ RdLite d_lite(const DevReads &R, int64_t r) {
    RdLite x; x.rs = R.ref_start[r]; x.re = R.ref_end[r]; x.f = R.flags[r]; x.mq = R.mapq[r];
    return x;
}
__device__ __forceinline__ bool l_overlaps(const RdLite &x, int32_t start) { return (x.re <= x.rs ? x.rs + 1 : x.re) > start; }   // d_overlaps
__device__ __forceinline__ bool l_passes(const DevReads &R, const DevParams &P, int64_t r, const RdLite &x) {                     // d_passes
    if ((int)x.mq < P.minq) return false;
    if (d_lkind((uint8_t)x.f) && d_rkind((uint8_t)x.f)) {   // clipped at both ends (rare): the lengths come from memory
        const int aligned = (int)R.qlen[r] - d_rlen(R, r, (uint8_t)x.f) - d_llen(R, r, (uint8_t)x.f);
        if (aligned < P.minial) return false;
    }
    return true;
}
__device__ __forceinline__ int l_set_of(const DevReads &R, const DevParams &P, int64_t r, const RdLite &x, int32_t q, int qi) {   // d_set_of
    const uint8_t f = (uint8_t)x.f;
    const int lk = d_lkind(f), rk = d_rkind(f);
    const bool rclip = rk && (rk == 1 ? x.re : x.re - 1) == q, lclip = lk && (lk == 1 ? x.rs - 1 : x.rs) == q;
    if (qi == 3) return rclip ? 8 : -1;
    if (qi == 4) return lclip ? 9 : -1;
    if (rclip || lclip) return -1;                                 // findflanks.py:429-430
    const int cat = (f & MFK_FLAG_COMPLEX) ? d_classify(R, P, r, f, q, qi != 0) : ((x.rs <= q && x.re > q) ? CAT_RUNTHROUGH : CAT_NONE);
    if (qi == 0) return cat - 1;                                   // CAT_NONE -> -1
    return cat == CAT_DELETION ? 5 + qi : -1;                      // 6 upstream, 7 downstream
}

// the whole workgroup tallies one site into cnt[NSETS] (LDS); range = 2 x int64 of LDS
template <int NSETS>
__device__ __forceinline__ void d_site_tally(const DevReads &R, const DevParams &P, int32_t contig, int32_t pos, uint32_t *cnt, int64_t *range) {
    const int32_t clen = R.contig_len[contig];
    int32_t start = pos - 1, end = pos + 2;
    if (start < 0) start = 0;
    if (end > clen) end = clen;
    if (threadIdx.x < NSETS) cnt[threadIdx.x] = 0;
    if (threadIdx.x < 128) d_fetch_range_2waves(R, contig, start, end, range);   // first two waves: one bound of the read range each
    __syncthreads();
    const int64_t lo = range[0], hi = range[1];
    uint32_t mine[NSETS];   // per-thread tallies: one LDS atomic per set and wave at the end
#pragma unroll
    for (int c = 0; c < NSETS; c++) mine[c] = 0;
    constexpr int NQ = NSETS > 8 ? 5 : 3;
    for (int64_t r = lo + threadIdx.x; r < hi; r += blockDim.x) {
        RdLite x = d_lite(R, r);
        int32_t lk1 = R.link[r];
        MFK_KEEP(x.rs); MFK_KEEP(x.re); MFK_KEEP(x.f); MFK_KEEP(x.mq); MFK_KEEP(lk1);
        if (!l_overlaps(x, start)) continue;
        if (!l_passes(R, P, r, x)) continue;
        // the previous read of the name (the mate, as a rule), loaded as a whole too
        const bool has_p = lk1 >= 0 && (int64_t)lk1 < r;
        RdLite y; y.rs = 0; y.re = 0; y.f = 0; y.mq = 0;
        int32_t lk2 = -1; uint32_t pc = 0xFFFFFFFFu;
        if (has_p) {
            y = d_lite(R, lk1); lk2 = R.link[lk1]; pc = R.contig_id[lk1];
            MFK_KEEP(y.rs); MFK_KEEP(y.re); MFK_KEEP(y.f); MFK_KEEP(y.mq); MFK_KEEP(lk2); MFK_KEEP(pc);
        }
        const bool p_in = has_p && (int32_t)pc == contig && y.rs < end && l_overlaps(y, start) && l_passes(R, P, lk1, y);
        for (int qi = 0; qi < NQ; qi++) {
            int32_t q = qi == 1 ? pos - 1 : (qi == 2 ? pos + 1 : pos);
            if (qi == 1 && pos - 1 < 0) continue;                 // findflanks.py:382
            if (qi == 2 && pos + 1 >= clen) continue;             // findflanks.py:387
            int set = l_set_of(R, P, r, x, q, qi);
            if (set < 0) continue;
            // a read counts for a set only if no earlier read of its name falls in the same set: len(set(names))
            bool dup = p_in && l_set_of(R, P, lk1, y, q, qi) == set;
            if (!dup && has_p) {
                int64_t prev = lk1;
                for (int64_t p = lk2; p >= 0 && p < prev; prev = p, p = R.link[p]) {   // third and further reads of the name (rare)
                    if ((int32_t)R.contig_id[p] != contig) continue;
                    if (!(R.ref_start[p] < end && d_overlaps(R, p, start))) continue;
                    const uint8_t fp = R.flags[p];
                    if (!d_passes(R, P, p, fp)) continue;
                    if (d_set_of(R, P, p, fp, q, qi) == set) { dup = true; break; }
                }
            }
            if (!dup) {
#pragma unroll
                for (int c = 0; c < NSETS; c++) if (c == set) mine[c]++;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < NSETS; c++) {
        uint32_t v = mine[c];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(&cnt[c], v);
    }
    __syncthreads();
}

__global__ __launch_bounds__(1024) void k_classify(DevReads R, DevParams P, ActTab A, uint32_t n_act) {
    __shared__ uint32_t cnt[8];
    __shared__ int64_t range[2];
    const uint32_t k = blockIdx.x;
    if (k >= n_act) return;
    const int32_t contig = A.contig[k];
    const int32_t pos = (int32_t)(A.gpos[k] - R.contig_base[contig]) - 1;
    d_site_tally<8>(R, P, contig, pos, cnt, range);
    if (threadIdx.x < 8) A.counts[8 * k + threadIdx.x] = cnt[threadIdx.x];
}

// recall (Recaller.parse_clipped_and_unclipped_read_info, recall.py:50-112): the same tallies at GIVEN sites, plus the
// distinct names of the reads lenient-clipped at the site.  sites: contig, pos pairs; counts: 10 per site in the order
// softclip 5p, softclip 3p, runthrough, small ins 5p/3p, large ins 5p/3p, deletion, upstream / downstream deletion
__global__ __launch_bounds__(1024) void k_recall(DevReads R, DevParams P, const int32_t *sites, uint32_t n, int32_t *counts) {
    __shared__ uint32_t cnt[10];
    __shared__ int64_t range[2];
    const uint32_t k = blockIdx.x;
    if (k >= n) return;
    d_site_tally<10>(R, P, sites[2 * k], sites[2 * k + 1], cnt, range);
    if (threadIdx.x < 10) counts[10 * k + threadIdx.x] = (int32_t)cnt[threadIdx.x < 2 ? 8 + threadIdx.x : threadIdx.x - 2];
}

// ---- K7: filter_softclips_count_ratios findflanks.py:203-239; ratios 827-845 (float64, quirk Q8) -------------------
__global__ void k_ratio_filter(ActTab A, uint32_t n_act, double minratio, double maxir, uint32_t *stat) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    int alive = 0;
    if (k < n_act) {
        uint32_t s = A.st[k];
        if (s & 4) {
            const uint32_t *c = A.counts + 8 * k;
            uint32_t sc5 = A.cnt2[2 * k], sc3 = A.cnt2[2 * k + 1];
            uint32_t run = c[0], si5 = c[1], si3 = c[2], li5 = c[3], li3 = c[4], del = c[5], up = c[6], down = c[7];
            double tot = (double)(sc5 + sc3 + run + si5 + si3 + li5 + li3 + del);     // get_total_count 789-795 (Q7)
            if (s & 1) { if ((double)(sc5 + li5 + li3) / tot < minratio || (double)(si5 + del) / tot > maxir) s &= ~1u; }
            if (s & 2) { if ((double)(sc3 + li5 + li3) / tot < minratio || (double)(si3 + del) / tot > maxir) s &= ~2u; }
            if (s & 3) { if ((double)down / tot > maxir || (double)up / tot > maxir) s &= ~3u; }
            if (!(s & 3)) s &= ~4u;
            A.st[k] = s;
            alive = (s >> 2) & 1;
        }
    }
    int c = __syncthreads_count(alive);
    if (threadIdx.x == 0 && c) atomicAdd(stat, (uint32_t)c);
}

// the min-distance stencil of k_mindistance on a status array in LDS (one workgroup, n <= blockDim.x): thread i's new status
__device__ __forceinline__ uint32_t d_mindistance_lds(const uint32_t *gpos, const uint32_t *cnt2, const uint32_t *st_lds, uint32_t i, uint32_t n, int mindist) {
    uint32_t s = st_lds[i];
    if (!(s & 4)) return s;
    const uint32_t g = gpos[i], md = mindist < 0 ? 0u : (uint32_t)mindist;
    if (cnt2[2 * i] > 0) {
        bool found = false;
        if (mindist >= 1)
            for (int64_t j = (int64_t)i - 1; j >= 0 && g - gpos[j] <= md; j--)
                if ((st_lds[j] & 4) && cnt2[2 * j + 1] > 0) { found = true; break; }
        if (!found) s &= ~1u;
    }
    if (cnt2[2 * i + 1] > 0) {
        bool found = false;
        if (mindist >= 1)
            for (uint32_t j = i + 1; j < n && gpos[j] - g <= md; j++)
                if ((st_lds[j] & 4) && cnt2[2 * j] > 0) { found = true; break; }
        if (!found) s &= ~2u;
    }
    if (!(s & 3)) s &= ~4u;
    return s;
}

// k_ratio_filter followed by k_mindistance for up to 1024 active sites: one workgroup, one launch
__global__ __launch_bounds__(1024) void k_ratio_mindist(ActTab A, uint32_t n_act, double minratio, double maxir, int mindist, uint32_t *st_out, uint32_t *stat_ratio, uint32_t *stat_dist) {
    __shared__ uint32_t st_lds[1024];
    const uint32_t k = threadIdx.x;
    uint32_t s = 0;
    if (k < n_act) {
        s = A.st[k];
        if (s & 4) {
            const uint32_t *c = A.counts + 8 * k;
            uint32_t sc5 = A.cnt2[2 * k], sc3 = A.cnt2[2 * k + 1];
            uint32_t run = c[0], si5 = c[1], si3 = c[2], li5 = c[3], li3 = c[4], del = c[5], up = c[6], down = c[7];
            double tot = (double)(sc5 + sc3 + run + si5 + si3 + li5 + li3 + del);     // get_total_count 789-795 (Q7)
            if (s & 1) { if ((double)(sc5 + li5 + li3) / tot < minratio || (double)(si5 + del) / tot > maxir) s &= ~1u; }
            if (s & 2) { if ((double)(sc3 + li5 + li3) / tot < minratio || (double)(si3 + del) / tot > maxir) s &= ~2u; }
            if (s & 3) { if ((double)down / tot > maxir || (double)up / tot > maxir) s &= ~3u; }
            if (!(s & 3)) s &= ~4u;
        }
    }
    st_lds[k] = s;
    const int c1 = __syncthreads_count((int)((s >> 2) & 1));
    uint32_t s2 = 0;
    if (k < n_act) { s2 = d_mindistance_lds(A.gpos, A.cnt2, st_lds, k, n_act, mindist); st_out[k] = s2; }
    const int c2 = __syncthreads_count((int)((s2 >> 2) & 1));
    if (threadIdx.x == 0) { if (c1) atomicAdd(stat_ratio, (uint32_t)c1); if (c2) atomicAdd(stat_dist, (uint32_t)c2); }
}

// ---- consensus filters findflanks.py:241-342 -----------------------------------------------------------------------
__global__ void k_consensus_filters(ActTab A, uint32_t n_act, const uint32_t *n_pass_len, const uint32_t *n_pass_both, uint32_t *stats) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_act) return;
    uint32_t s = A.st[k];
    if (!(s & 4)) return;
    // filter_consensus_sequences_minlength
    uint32_t n6 = 0, n7 = 0, n8 = 0;
    for (int side = 0; side < 2; side++) if (s & (1u << side)) { uint32_t c = n_pass_len[2 * k + side]; if (!c) s &= ~(1u << side); else n6 += c; }
    if (!(s & 3)) { A.st[k] = s & ~4u; return; }
    atomicAdd(&stats[6], n6);
    // filter_consensus_sequences_mincount
    for (int side = 0; side < 2; side++) if (s & (1u << side)) { uint32_t c = n_pass_both[2 * k + side]; if (!c) s &= ~(1u << side); else { n7 += c; n8 += 1; } }
    if (!(s & 3)) { A.st[k] = s & ~4u; return; }
    atomicAdd(&stats[7], n7);
    atomicAdd(&stats[8], n8);   // filter_multiple_consensus_sequences keeps one per remaining side
    A.st[k] = s;
}

// k_consensus_filters followed by the last k_mindistance for up to 1024 active sites: one workgroup, one launch
__global__ __launch_bounds__(1024) void k_consfilters_mindist(ActTab A, uint32_t n_act, const uint32_t *n_pass_len, const uint32_t *n_pass_both, int mindist, uint32_t *st_out, uint32_t *stats) {
    __shared__ uint32_t st_lds[1024];
    const uint32_t k = threadIdx.x;
    uint32_t s = 0;
    if (k < n_act) {
        s = A.st[k];
        if (s & 4) {
            uint32_t n6 = 0, n7 = 0, n8 = 0;
            for (int side = 0; side < 2; side++) if (s & (1u << side)) { uint32_t c = n_pass_len[2 * k + side]; if (!c) s &= ~(1u << side); else n6 += c; }
            if (!(s & 3)) s &= ~4u;
            else {
                atomicAdd(&stats[6], n6);
                for (int side = 0; side < 2; side++) if (s & (1u << side)) { uint32_t c = n_pass_both[2 * k + side]; if (!c) s &= ~(1u << side); else { n7 += c; n8 += 1; } }
                if (!(s & 3)) s &= ~4u;
                else { atomicAdd(&stats[7], n7); atomicAdd(&stats[8], n8); }
            }
        }
    }
    st_lds[k] = s;
    __syncthreads();
    uint32_t s2 = 0;
    if (k < n_act) { s2 = d_mindistance_lds(A.gpos, A.cnt2, st_lds, k, n_act, mindist); st_out[k] = s2; }
    const int c2 = __syncthreads_count((int)((s2 >> 2) & 1));
    if (threadIdx.x == 0 && c2) atomicAdd(&stats[9], (uint32_t)c2);
}

// ---- result rows on the device (thousands of sites): make_dataframe findflanks.py:631-670 ---------------------------------------
// With tens of thousands of sites the host loop that builds the site and flank rows from nine read-back arrays (and the
// read-back of a fixed-stride block of consensus characters) costs more than a millisecond.  Then the rows are laid out
// here: k_rows_count gives every active site its number of site rows (0/1), flank rows (0..2) and consensus bytes, two
// scans place them, k_rows_fill writes mfk_site / mfk_flank records and the consensus characters back to back, and the host
// copies three finished arrays.
__global__ void k_rows_count(ActTab A, uint32_t n_act, const int32_t *cons_len, uint32_t *c_rows, uint32_t *c_bytes) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_act) return;
    const uint32_t st = A.st[k];
    uint32_t rows = 0, bytes = 0;
    if (st & 4) {
        rows = 1u;                                                  // site rows in the low 24 bits, flank rows above
        if (st & 1) { rows += 1u << 24; bytes += (uint32_t)cons_len[2 * k]; }
        if (st & 2) { rows += 1u << 24; bytes += (uint32_t)cons_len[2 * k + 1]; }
    }
    c_rows[k] = rows; c_bytes[k] = bytes;
}

// one wave per active site
__global__ __launch_bounds__(256) void k_rows_fill(DevReads R, ActTab A, uint32_t n_act, const int32_t *cons_count, const int32_t *cons_len, const char *cons_chars, uint32_t cons_stride,
                                                   const uint64_t *o_rows, const uint64_t *o_bytes, mfk_site *sites, mfk_flank *flanks, char *seqs) {
    const int lane = threadIdx.x & 63;
    const uint32_t k = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= n_act) return;
    const uint32_t st = A.st[k];
    if (!(st & 4)) return;
    const uint64_t orow = o_rows[k];
    const uint32_t so = (uint32_t)(orow & 0xFFFFFFull);
    uint32_t fo = (uint32_t)(orow >> 24);
    uint64_t bo = o_bytes[k];
    const int32_t contig = A.contig[k];
    const int32_t pos = (int32_t)((int64_t)A.gpos[k] - (int64_t)R.contig_base[contig] - 1);
    const uint32_t *c = A.counts + 8 * (size_t)k;
    mfk_site s;
    s.contig_id = contig; s.pos = pos;
    s.softclip_count_5p = (int32_t)A.cnt2[2 * k]; s.softclip_count_3p = (int32_t)A.cnt2[2 * k + 1];
    s.runthrough_count = (int32_t)c[0]; s.small_insertion_count_5p = (int32_t)c[1]; s.small_insertion_count_3p = (int32_t)c[2];
    s.large_insertion_count_5p = (int32_t)c[3]; s.large_insertion_count_3p = (int32_t)c[4]; s.deletion_count = (int32_t)c[5];
    s.upstream_deletion_count = (int32_t)c[6]; s.downstream_deletion_count = (int32_t)c[7];
    s.keep_5p = (int32_t)(st & 1); s.keep_3p = (int32_t)((st >> 1) & 1);
    if (lane == 0) sites[so] = s;
    for (int side = 0; side < 2; side++) {
        if (!(st & (1u << side))) continue;   // wave-uniform
        const size_t job = 2 * (size_t)k + side;
        const int32_t len = cons_len[job];
        if (lane == 0) {
            mfk_flank f;
            f.contig_id = contig; f.pos = pos; f.orient = side == 0 ? 5 : 3;
            f.softclip_count_5p = s.softclip_count_5p; f.softclip_count_3p = s.softclip_count_3p; f.runthrough_count = s.runthrough_count;
            f.small_insertion_count_5p = s.small_insertion_count_5p; f.small_insertion_count_3p = s.small_insertion_count_3p;
            f.large_insertion_count_5p = s.large_insertion_count_5p; f.large_insertion_count_3p = s.large_insertion_count_3p;
            f.deletion_count = s.deletion_count; f.upstream_deletion_count = s.upstream_deletion_count; f.downstream_deletion_count = s.downstream_deletion_count;
            f.total_count = s.softclip_count_5p + s.softclip_count_3p + s.runthrough_count + s.small_insertion_count_5p + s.small_insertion_count_3p +
                            s.large_insertion_count_5p + s.large_insertion_count_3p + s.deletion_count;   // get_total_count findflanks.py:789-795
            f.consensus_softclip_count = cons_count[job];
            f.seq_len = len; f.seq_off = (int64_t)bo;
            flanks[fo] = f;
        }
        const char *src = cons_chars + job * cons_stride;
        for (int d = lane; d < len; d += 64) seqs[bo + d] = src[d];
        fo++; bo += (uint64_t)len;
    }
}

// =====================================================================================================================
static const char BASE_ASCII[] = "=ACMGRSVTWYHKDBN";

extern "C" int mfk_findflanks(mfk_ctx *ctx, const mfk_reads *reads, const mfk_params *params, int stop_after, int64_t *n_sites_out, int64_t *n_flanks_out, int64_t *seq_bytes_out) {
    return findflanks_impl(ctx, reads, params, stop_after, n_sites_out, n_flanks_out, seq_bytes_out, nullptr);
}

int findflanks_impl(mfk_ctx *ctx, const mfk_reads *reads, const mfk_params *params, int stop_after, int64_t *n_sites_out, int64_t *n_flanks_out, int64_t *seq_bytes_out, FlankHandoff *fh) {
    if (!ctx || !reads || !params) return 1;
    if (fh) fh->active = false;
    HIPCHK(hipSetDevice(ctx->device));
    ctx->err[0] = 0;
    ctx->sites.clear(); ctx->flanks.clear(); ctx->seqs.clear();
    memset(ctx->stats, 0, sizeof ctx->stats);
    if (n_sites_out) *n_sites_out = 0;
    if (n_flanks_out) *n_flanks_out = 0;
    if (seq_bytes_out) *seq_bytes_out = 0;
    if (stop_after < MFK_STAGE_PARSE) stop_after = MFK_STAGE_PARSE;
    HostTimer ht("findflanks");
    arena_reset(ctx);
    DevReads R; DevParams P;
    if (int rc = upload_reads(ctx, reads, params, &R, &P)) return rc;
    hipStream_t st = ctx->stream;
    SPAN_EVENT(ev0);
    // (stage 1 below) -------------------------------------------------------------------------------
    const int64_t ngroups = (R.n + 15) >> 4;
    unsigned nblk = (unsigned)((ngroups + SCAN_THREADS * SCAN_SUB - 1) / (SCAN_THREADS * SCAN_SUB));   // one round of SCAN_SUB tiles per workgroup and turn
    if (nblk > SCAN_MAX_BLOCKS) nblk = SCAN_MAX_BLOCKS;
    if (nblk < 1) nblk = 1;
    uint32_t *d_stats = nullptr, *ev_count = nullptr;   // ev_count[0] events, [SEG_PAD * (1 + seg)] listed reads of a segment
    uint32_t E = 0;
    uint64_t *ev = nullptr, *bkt = nullptr;
    uint32_t *bkt_count = nullptr;
    uint64_t cap64 = (uint64_t)R.n / 8 + 65536, seg_cap = cap64 / LIST_SEGS + 4096;   // clips are a few percent of the reads; grown on overflow
    std::vector<uint32_t> h_cnt((size_t)SEG_PAD * (LIST_SEGS + 1)), h_bkt;
    // position buckets (K1b): about a thousand reads' worth of events per bucket, 64 .. 4096 buckets
    uint32_t B = 64;
    while (B < 4096 && (uint64_t)B * 16 * 1024 < (uint64_t)R.n) B <<= 1;
    const char *nb_env = getenv("MFK_NO_BUCKET_SORT");
    bool bucketed = !(nb_env && *nb_env && *nb_env != '0') && R.gpos_span > 0;
    for (int attempt = 0; attempt < 3; attempt++) {
        if (cap64 > 2 * (uint64_t)R.n + 16) cap64 = 2 * (uint64_t)R.n + 16;
        if (cap64 > 0xFFFFFFF0ull || seg_cap > 0x7FFFFFF0ull) { snprintf(ctx->err, sizeof ctx->err, "more than 2^32 soft-clip events in one call"); return 1; }
        if (bucketed) bkt = aalloc<uint64_t>(ctx, (size_t)B * BKT_CAP);
        else ev = aalloc<uint64_t>(ctx, (size_t)cap64);
        // the list of clipped reads: 64 segments with a counter each, or (dense clip records) one flat list with a slot per record,
        // zeroed first so that a table that does not restate the columns loses clips instead of leaving slots unwritten
        const bool dense = R.clip_rec != nullptr;
        uint64_t *list = nullptr;
        if (!dense) { list = aalloc<uint64_t>(ctx, (size_t)seg_cap * LIST_SEGS); if (!list) return 1; }
        if (bucketed ? !bkt : !ev) return 1;
        FillGroup z(ctx, 0);
        if (dense) z.want(&list, (size_t)R.n_clip + 16);
        z.want(&d_stats, MFK_N_STATS); z.want(&ev_count, h_cnt.size());
        if (bucketed) z.want(&bkt_count, B);
        if (z.commit()) return 1;
        PROF_LAUNCH(k_scan_clips, dim3(nblk), dim3(SCAN_THREADS), 0, st, R, list, ev_count + SEG_PAD, (uint32_t)seg_cap);
        KCHK();
        if (bucketed) {
            PROF_LAUNCH(k_clip_events_bkt, dim3(32, LIST_SEGS), dim3(256), 0, st, R, P, (const uint64_t *)list, (const uint32_t *)(ev_count + SEG_PAD), (uint32_t)seg_cap, bkt, bkt_count, B, R.gpos_span > B ? (((uint64_t)B << 32) / R.gpos_span) : (uint64_t)0);
        } else {
            PROF_LAUNCH(k_clip_events, dim3(8, LIST_SEGS), dim3(256), 0, st, R, P, (const uint64_t *)list, (const uint32_t *)(ev_count + SEG_PAD), (uint32_t)seg_cap, ev, ev_count, (uint32_t)cap64);
        }
        KCHK();
        {
            ReadBatch rb(ctx);
            rb.add(ev_count, h_cnt.data(), h_cnt.size() * 4);
            if (bucketed) { h_bkt.resize(B); rb.add(bkt_count, h_bkt.data(), (size_t)B * 4); }
            if (int rc = rb.flush()) return rc;
        }
        ht.lap("scan + events + read-back");
        uint64_t seg_max = 0, listed = 0, bmax = 0, bsum = 0;
        if (R.clip_rec) listed = (uint64_t)R.n_clip;   // (flat list: nothing to overflow)
        else for (int sgi = 0; sgi < LIST_SEGS; sgi++) { uint64_t c = h_cnt[(size_t)SEG_PAD * (1 + sgi)]; listed += c; if (c > seg_max) seg_max = c; }
        if (bucketed) { for (uint32_t c : h_bkt) { bsum += c; if (c > bmax) bmax = c; } }
        if (bsum > 0xFFFFFFF0ull) { snprintf(ctx->err, sizeof ctx->err, "more than 2^32 soft-clip events in one call"); return 1; }
        E = bucketed ? (uint32_t)bsum : h_cnt[0];
        if (seg_max <= seg_cap && (bucketed ? bmax <= BKT_CAP : (uint64_t)E <= cap64)) break;
        if (attempt == 2) { snprintf(ctx->err, sizeof ctx->err, "soft-clip event buffers still too small after two retries"); return 1; }
        if (seg_max > seg_cap) { seg_cap = seg_max + 16; if (cap64 < 2 * listed + 16) cap64 = 2 * listed + 16; }   // every read was counted: exact sizes
        else if (bucketed) { bucketed = false; if (cap64 < 2 * listed + 16) cap64 = 2 * listed + 16; }               // a site deeper than a bucket: list + device-wide sort
        else cap64 = (uint64_t)E + 16;                                                                            // every event was counted
    }
    ctx->stats[10] = E;

    uint32_t n_act = 0;
    ActTab A; memset(&A, 0, sizeof A);
    SiteTab T; memset(&T, 0, sizeof T);
    uint64_t *ev_sorted = nullptr; uint8_t *ev_info = nullptr;
    uint32_t *ascan = nullptr; const uint32_t *n_sites_p = nullptr;
    if (E > 0) {
        ev_sorted = aalloc<uint64_t>(ctx, E);
        ev_info = aalloc<uint8_t>(ctx, E);
        uint32_t *head = aalloc<uint32_t>(ctx, E), *head_scan = aalloc<uint32_t>(ctx, (size_t)E + 1);
        T.gpos = aalloc<uint32_t>(ctx, E); T.contig = aalloc<int32_t>(ctx, E);
        T.ev_begin = aalloc<uint32_t>(ctx, E); T.ev_end = aalloc<uint32_t>(ctx, E); T.st = aalloc<uint32_t>(ctx, E);
        uint32_t *st2 = aalloc<uint32_t>(ctx, E), *alive = nullptr;
        ascan = aalloc<uint32_t>(ctx, (size_t)E + 1);
        // the site arrays are first touched by k_site_accumulate: their fill runs on the third stream, next to the sort, the
        // event pass and the scan
        FillGroup z(ctx, 0), f(ctx, 0xFF);
        z.want(&T.cnt2, 2 * (size_t)E); z.want(&T.sflags, E); z.want(&alive, E);
        f.want(&T.ev_mid, E);
        if (!ev_info || !head || !head_scan || !T.gpos || !T.contig || !T.ev_begin || !T.ev_end || !T.st || !st2 || !ascan || !ev_sorted) return 1;
        ctx->fill_stream = ctx->stream3;
        const int fill_rc = fill_commit2(z, f);
        ctx->fill_stream = nullptr;
        if (fill_rc) return 1;
        HIPCHK(hipEventRecord(ctx->ev_fill, ctx->stream3));
        if (bucketed) {
            DeepBuckets D; D.n = 0; D.last = 0;
            for (uint32_t b = 0; b < B && D.n < DEEP_MAX; b++) if (h_bkt[b] > DEEP_MIN) { D.id[D.n++] = (uint16_t)b; D.last = (uint16_t)b; }
            PROF_LAUNCH(k_bucket_sort, dim3(B + D.n), dim3(BSORT_THREADS), 0, st, (const uint64_t *)bkt, (const uint32_t *)bkt_count, B, ev_sorted, D, R, P, ev_info, head);
            KCHK();
        } else if (int rc = dev_sort_keys64(ctx, ev, ev_sorted, E, 0, 64)) return rc;
        const unsigned eb = nblocks(E, 256);
        if (!bucketed) {   // (the bucket sort has written the per-event information itself)
            PROF_LAUNCH(k_event_info, dim3(eb), dim3(256), 0, st, R, P, (const uint64_t *)ev_sorted, E, ev_info, head);
            KCHK();
        }
        if (int rc = dev_excl_scan32(ctx, head, head_scan, E)) return rc;
        n_sites_p = head_scan + E;
        HIPCHK(hipStreamWaitEvent(st, ctx->ev_fill, 0));
        PROF_LAUNCH(k_site_accumulate, dim3(eb), dim3(256), 0, st, R, (const uint64_t *)ev_sorted, (const uint8_t *)ev_info, (const uint32_t *)head_scan, E, T);
        KCHK();
        const int upto = stop_after >= MFK_STAGE_MINCOUNT ? 2 : (stop_after >= MFK_STAGE_MINLENGTH ? 1 : 0);
        PROF_LAUNCH(k_site_filters, dim3(eb < 256 ? eb : 256), dim3(256), 0, st, T, n_sites_p, P.mincount, upto, d_stats);
        KCHK();
        if (stop_after >= MFK_STAGE_MINDIST1) {
            PROF_LAUNCH(k_mindistance, dim3(eb), dim3(256), 0, st, (const uint32_t *)T.gpos, (const uint32_t *)T.cnt2, (const uint32_t *)T.st, st2, n_sites_p, 0u, P.mindist, d_stats + 3, alive);
            KCHK();
            T.st = st2;
        } else {
            PROF_LAUNCH(k_alive_flags, dim3(eb), dim3(256), 0, st, (const uint32_t *)T.st, n_sites_p, E, alive);
            KCHK();
        }
        // surviving sites -> dense active table
        if (int rc = dev_excl_scan32(ctx, alive, ascan, E)) return rc;
        if (int rc = read_back(ctx, ascan + E, &n_act, 4)) return rc;
        ht.lap("sites + read-back");
    }
    ConsensusIO io; memset(&io, 0, sizeof io);
    if (n_act > 0) {
        A.site = aalloc<uint32_t>(ctx, n_act); A.gpos = aalloc<uint32_t>(ctx, n_act); A.contig = aalloc<int32_t>(ctx, n_act);
        A.cnt2 = aalloc<uint32_t>(ctx, 2 * (size_t)n_act); A.st = aalloc<uint32_t>(ctx, n_act);
        uint32_t *st2 = aalloc<uint32_t>(ctx, n_act);
        FillGroup z(ctx, 0);
        z.want(&A.counts, 8 * (size_t)n_act);
        if (stop_after >= MFK_STAGE_CONSENSUS) {
            z.want(&io.cons_count, 2 * (size_t)n_act); z.want(&io.cons_len, 2 * (size_t)n_act);
            z.want(&io.n_pass_len, 2 * (size_t)n_act); z.want(&io.n_pass_both, 2 * (size_t)n_act);
        }
        if (!A.site || !A.gpos || !A.contig || !A.cnt2 || !A.st || !st2 || z.commit()) return 1;
        const unsigned eb = nblocks(E, 256), ab = nblocks(n_act, 256);
        PROF_LAUNCH(k_compact_sites, dim3(eb), dim3(256), 0, st, T, n_sites_p, (const uint32_t *)ascan, A);
        KCHK();
        if (stop_after >= MFK_STAGE_UNCLIPPED) {
            // few sites (a deep isolate): wide workgroups keep each site's gather chain short; thousands of sites: narrow
            // ones, so that four times as many sites are resident per CU
            PROF_LAUNCH(k_classify, dim3(n_act), dim3(n_act >= 2048 ? 256 : 1024), 0, st, R, P, A, n_act);
            KCHK();
        }
        const bool one_wg = n_act <= 1024;   // a few hundred sites: filter pairs in one launch of one workgroup
        if (one_wg && stop_after >= MFK_STAGE_MINDIST2) {
            PROF_LAUNCH(k_ratio_mindist, dim3(1), dim3(1024), 0, st, A, n_act, P.minratio, P.maxir, P.mindist, st2, d_stats + 4, d_stats + 5);
            KCHK();
            uint32_t *t = A.st; A.st = st2; st2 = t;
        } else {
        if (stop_after >= MFK_STAGE_RATIOS) {
            PROF_LAUNCH(k_ratio_filter, dim3(ab), dim3(256), 0, st, A, n_act, P.minratio, P.maxir, d_stats + 4);
            KCHK();
        }
        if (stop_after >= MFK_STAGE_MINDIST2) {
            PROF_LAUNCH(k_mindistance, dim3(ab), dim3(256), 0, st, (const uint32_t *)A.gpos, (const uint32_t *)A.cnt2, (const uint32_t *)A.st, st2, (const uint32_t *)nullptr, n_act, P.mindist, d_stats + 5, (uint32_t *)nullptr);
            KCHK();
            uint32_t *t = A.st; A.st = st2; st2 = t;
        }
        }
        if (stop_after >= MFK_STAGE_CONSENSUS) {
            io.ev_keys = ev_sorted; io.ev_info = ev_info; io.n_events = E;
            io.n_act = n_act; io.act_site = A.site; io.act_st = A.st; io.act_cnt2 = A.cnt2;
            io.site_ev_begin = T.ev_begin; io.site_ev_mid = T.ev_mid; io.site_ev_end = T.ev_end;
            io.work = reinterpret_cast<unsigned long long *>(d_stats + 12);   // 8-byte aligned: d_stats is an arena block
            ht.lap("classify .. enqueue");
            if (int rc = run_consensus(ctx, R, P, io)) return rc;
            ht.lap("consensus");
            if (one_wg && stop_after >= MFK_STAGE_ALL) {
                PROF_LAUNCH(k_consfilters_mindist, dim3(1), dim3(1024), 0, st, A, n_act, (const uint32_t *)io.n_pass_len, (const uint32_t *)io.n_pass_both, P.mindist, st2, d_stats);
                KCHK();
                uint32_t *t = A.st; A.st = st2; st2 = t;
            } else {
                PROF_LAUNCH(k_consensus_filters, dim3(ab), dim3(256), 0, st, A, n_act, (const uint32_t *)io.n_pass_len, (const uint32_t *)io.n_pass_both, d_stats);
                KCHK();
            }
        }
        if (stop_after >= MFK_STAGE_ALL && !one_wg) {
            PROF_LAUNCH(k_mindistance, dim3(ab), dim3(256), 0, st, (const uint32_t *)A.gpos, (const uint32_t *)A.cnt2, (const uint32_t *)A.st, st2, (const uint32_t *)nullptr, n_act, P.mindist, d_stats + 9, (uint32_t *)nullptr);
            KCHK();
            uint32_t *t = A.st; A.st = st2; st2 = t;
        }
    }
    // thousands of sites: the rows are assembled on the device
    const bool dev_rows = stop_after >= MFK_STAGE_ALL && n_act > 2048 && n_act < (1u << 24) && io.cons_chars != nullptr;   // 24 bits of site rows in the packed scan
    uint64_t *o_rows = nullptr, *o_bytes = nullptr; mfk_site *d_sites = nullptr; mfk_flank *d_flanks = nullptr; char *d_seqs = nullptr;
    if (dev_rows) {
        uint32_t *c_rows = aalloc<uint32_t>(ctx, n_act), *c_bytes = aalloc<uint32_t>(ctx, n_act);
        o_rows = aalloc<uint64_t>(ctx, (size_t)n_act + 1); o_bytes = aalloc<uint64_t>(ctx, (size_t)n_act + 1);
        d_sites = aalloc<mfk_site>(ctx, n_act); d_flanks = aalloc<mfk_flank>(ctx, 2 * (size_t)n_act);
        d_seqs = aalloc<char>(ctx, 2 * (size_t)n_act * io.cons_stride + 16);
        if (!c_rows || !c_bytes || !o_rows || !o_bytes || !d_sites || !d_flanks || !d_seqs) return 1;
        PROF_LAUNCH(k_rows_count, dim3(nblocks(n_act, 256)), dim3(256), 0, st, A, n_act, (const int32_t *)io.cons_len, c_rows, c_bytes);
        KCHK();
        if (int rc = dev_excl_scan64(ctx, c_rows, o_rows, n_act)) return rc;
        if (int rc = dev_excl_scan64(ctx, c_bytes, o_bytes, n_act)) return rc;
        PROF_LAUNCH(k_rows_fill, dim3(nblocks(n_act, 4)), dim3(256), 0, st, R, A, n_act, (const int32_t *)io.cons_count, (const int32_t *)io.cons_len, (const char *)io.cons_chars, io.cons_stride,
                    (const uint64_t *)o_rows, (const uint64_t *)o_bytes, d_sites, d_flanks, d_seqs);
        KCHK();
    }
    if (dev_rows && fh) {
        // mfk_flankpairs: the pairing stage takes the rows where they are and reads everything back with its own read-backs
        fh->active = true; fh->n_act = n_act;
        fh->d_sites = d_sites; fh->d_flanks = d_flanks; fh->d_seqs = d_seqs;
        fh->d_rows_total = o_rows + n_act; fh->d_bytes_total = o_bytes + n_act;
        fh->d_stats = d_stats; fh->d_cons_err = io.err_dev; fh->sure_small = io.sure_small;
        ht.lap("rows handed over on the device");
        return 0;
    }
    SPAN_EVENT(ev1);
    // ---- results to the host -------------------------------------------------------------------------------------
    uint32_t h_stats[MFK_N_STATS], h_scan_err = 0;
    if (dev_rows) {
        uint64_t t_rows = 0, t_bytes = 0;
        {
            ReadBatch rb(ctx);
            rb.add(d_stats, h_stats, sizeof h_stats);
            rb.add(ctx->scan_err, &h_scan_err, 4);
            if (io.err_dev) rb.add(io.err_dev, io.err_words, 16);
            rb.add(o_rows + n_act, &t_rows, 8); rb.add(o_bytes + n_act, &t_bytes, 8);
            if (int rc = rb.flush()) return rc;
        }
        io.err_flags = io.err_words[0];
        if (io.sure_small && io.err_words[3]) { snprintf(ctx->err, sizeof ctx->err, "internal: a consensus job did not finish in k_job_small although its size bound said it would"); return 11; }
        if (int rc = check_scan_error(ctx, h_scan_err)) return rc;
        if (io.err_flags & 2u) { snprintf(ctx->err, sizeof ctx->err, "a clipped read at a surviving site has no sequence in the pool (pack with pool='clipped')"); return 2; }
        const size_t ns = (size_t)(t_rows & 0xFFFFFFull), nf = (size_t)(t_rows >> 24);
        ctx->sites.resize(ns); ctx->flanks.resize(nf); ctx->seqs.resize((size_t)t_bytes);
        {
            ReadBatch rb(ctx);
            rb.add(d_sites, ctx->sites.data(), ns * sizeof(mfk_site)); rb.add(d_flanks, ctx->flanks.data(), nf * sizeof(mfk_flank));
            rb.add(d_seqs, &ctx->seqs[0], (size_t)t_bytes);
            if (int rc = rb.flush()) return rc;
        }
        ht.lap("results read-back (rows made on the device)");
        for (int i = 0; i < 10; i++) ctx->stats[i] = h_stats[i];
        ctx->stats[12] = (int64_t)((uint64_t)h_stats[12] | ((uint64_t)h_stats[13] << 32));
        prof_collect(ctx);
        float ms = 0.f;
        if (ctx->span_on && hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1) == hipSuccess) ctx->last_ms = ms;
    } else {
    std::vector<uint32_t> gpos(n_act), cnt2(2 * (size_t)n_act), stv(n_act), counts(8 * (size_t)n_act);
    std::vector<int32_t> contig(n_act);
    std::vector<int32_t> ccount, clen; std::vector<char> cchars;
    const bool have_cons = stop_after >= MFK_STAGE_CONSENSUS && n_act > 0;
    {
        ReadBatch rb(ctx);
        rb.add(d_stats, h_stats, sizeof h_stats);
        rb.add(ctx->scan_err, &h_scan_err, 4);
        if (io.err_dev) rb.add(io.err_dev, io.err_words, 16);
        if (n_act > 0) {
            rb.add(A.gpos, gpos.data(), n_act * 4); rb.add(A.contig, contig.data(), n_act * 4);
            rb.add(A.cnt2, cnt2.data(), 2 * (size_t)n_act * 4); rb.add(A.st, stv.data(), n_act * 4);
            rb.add(A.counts, counts.data(), 8 * (size_t)n_act * 4);
        }
        if (have_cons) {
            ccount.resize(2 * (size_t)n_act); clen.resize(2 * (size_t)n_act);
            cchars.resize((size_t)2 * n_act * io.cons_stride + 1);
            rb.add(io.cons_count, ccount.data(), 2 * (size_t)n_act * 4); rb.add(io.cons_len, clen.data(), 2 * (size_t)n_act * 4);
            if (io.cons_stride && io.cons_chars) rb.add(io.cons_chars, cchars.data(), (size_t)2 * n_act * io.cons_stride);
        }
        if (int rc = rb.flush()) return rc;
    }
    ht.lap("results read-back");
    io.err_flags = io.err_words[0];
    if (io.sure_small && io.err_words[3]) { snprintf(ctx->err, sizeof ctx->err, "internal: a consensus job did not finish in k_job_small although its size bound said it would"); return 11; }
    if (int rc = check_scan_error(ctx, h_scan_err)) return rc;
    for (int i = 0; i < 10; i++) ctx->stats[i] = h_stats[i];
    ctx->stats[12] = (int64_t)((uint64_t)h_stats[12] | ((uint64_t)h_stats[13] << 32));
    prof_collect(ctx);
    float ms = 0.f;
    if (ctx->span_on && hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1) == hipSuccess) ctx->last_ms = ms;
    if (io.err_flags & 2u) { snprintf(ctx->err, sizeof ctx->err, "a clipped read at a surviving site has no sequence in the pool (pack with pool='clipped')"); return 2; }
    if (n_act > 0) {
        // contig_base on the host (same formula as upload_reads)
        std::vector<int64_t> base(reads->n_contigs);
        { int64_t pad = (int64_t)params->min_distance_to_mate + params->max_direct_repeat_length + 64; if (pad < 64) pad = 64; int64_t b = 0; for (int c = 0; c < reads->n_contigs; c++) { base[c] = b; b += (int64_t)reads->h_contig_len[c] + 2 + pad; } }
        ctx->sites.reserve(n_act);
        if (stop_after >= MFK_STAGE_ALL) { ctx->flanks.reserve(2 * (size_t)n_act); ctx->seqs.reserve((size_t)n_act * 128); }
        for (uint32_t k = 0; k < n_act; k++) {
            if (!(stv[k] & 4)) continue;
            mfk_site s; memset(&s, 0, sizeof s);
            s.contig_id = contig[k]; s.pos = (int32_t)((int64_t)gpos[k] - base[contig[k]] - 1);
            s.softclip_count_5p = (int32_t)cnt2[2 * k]; s.softclip_count_3p = (int32_t)cnt2[2 * k + 1];
            const uint32_t *c = &counts[8 * (size_t)k];
            s.runthrough_count = c[0]; s.small_insertion_count_5p = c[1]; s.small_insertion_count_3p = c[2];
            s.large_insertion_count_5p = c[3]; s.large_insertion_count_3p = c[4]; s.deletion_count = c[5];
            s.upstream_deletion_count = c[6]; s.downstream_deletion_count = c[7];
            s.keep_5p = stv[k] & 1; s.keep_3p = (stv[k] >> 1) & 1;
            ctx->sites.push_back(s);
            if (stop_after < MFK_STAGE_ALL) continue;
            // make_dataframe findflanks.py:631-670: 5p row then 3p row, one consensus per kept side
            for (int side = 0; side < 2; side++) {
                if (!(stv[k] & (1u << side))) continue;
                mfk_flank f; memset(&f, 0, sizeof f);
                f.contig_id = s.contig_id; f.pos = s.pos; f.orient = side == 0 ? 5 : 3;
                f.softclip_count_5p = s.softclip_count_5p; f.softclip_count_3p = s.softclip_count_3p; f.runthrough_count = s.runthrough_count;
                f.small_insertion_count_5p = s.small_insertion_count_5p; f.small_insertion_count_3p = s.small_insertion_count_3p;
                f.large_insertion_count_5p = s.large_insertion_count_5p; f.large_insertion_count_3p = s.large_insertion_count_3p;
                f.deletion_count = s.deletion_count; f.upstream_deletion_count = s.upstream_deletion_count; f.downstream_deletion_count = s.downstream_deletion_count;
                f.total_count = s.softclip_count_5p + s.softclip_count_3p + s.runthrough_count + s.small_insertion_count_5p + s.small_insertion_count_3p + s.large_insertion_count_5p + s.large_insertion_count_3p + s.deletion_count;
                size_t job = 2 * (size_t)k + side;
                f.consensus_softclip_count = ccount[job];
                f.seq_len = clen[job]; f.seq_off = (int64_t)ctx->seqs.size();
                ctx->seqs.append(cchars.data() + job * io.cons_stride, (size_t)clen[job]);
                ctx->flanks.push_back(f);
            }
        }
    }
    }
    ctx->stats[11] = (int64_t)ctx->flanks.size();
    ht.lap("host table");
    ctx->call_ok = true;
    if (n_sites_out) *n_sites_out = (int64_t)ctx->sites.size();
    if (n_flanks_out) *n_flanks_out = (int64_t)ctx->flanks.size();
    if (seq_bytes_out) *seq_bytes_out = (int64_t)ctx->seqs.size();
    (void)BASE_ASCII;
    return 0;
}

extern "C" int mfk_findflanks_fetch(mfk_ctx *ctx, mfk_site *sites, mfk_flank *flanks, char *seqs, int64_t *stats) {
    if (!ctx) return 1;
    if (sites && !ctx->sites.empty()) memcpy(sites, ctx->sites.data(), ctx->sites.size() * sizeof(mfk_site));
    if (flanks && !ctx->flanks.empty()) memcpy(flanks, ctx->flanks.data(), ctx->flanks.size() * sizeof(mfk_flank));
    if (seqs && !ctx->seqs.empty()) memcpy(seqs, ctx->seqs.data(), ctx->seqs.size());
    if (stats) memcpy(stats, ctx->stats, sizeof ctx->stats);
    return 0;
}

// ---- recall ---------------------------------------------------------------------------------------------------------
extern "C" int mfk_recall(mfk_ctx *ctx, const mfk_reads *reads, const mfk_params *params, const int32_t *h_contig_id, const int32_t *h_pos,
                          int64_t n_sites, int32_t *h_counts) {
    if (!ctx || !reads || !params || n_sites < 0 || (n_sites > 0 && (!h_contig_id || !h_pos || !h_counts))) return 1;
    HIPCHK(hipSetDevice(ctx->device));
    ctx->err[0] = 0;
    if (n_sites == 0) return 0;
    if (n_sites > 0x7FFFFFFF / 10) { snprintf(ctx->err, sizeof ctx->err, "too many sites in one call"); return 1; }
    for (int64_t i = 0; i < n_sites; i++) {
        if (h_contig_id[i] < 0 || h_contig_id[i] >= reads->n_contigs) { snprintf(ctx->err, sizeof ctx->err, "site %lld: contig id %d out of range", (long long)i, h_contig_id[i]); return 1; }
        if (h_pos[i] < -1 || h_pos[i] > reads->h_contig_len[h_contig_id[i]]) { snprintf(ctx->err, sizeof ctx->err, "site %lld: position %d outside contig", (long long)i, h_pos[i]); return 1; }
    }
    arena_reset(ctx);
    DevReads R; DevParams P;
    if (int rc = upload_reads(ctx, reads, params, &R, &P)) return rc;
    hipStream_t st = ctx->stream;
    const size_t pin_off = upload_reads_pinned_bytes(reads->n_contigs), bytes = (size_t)n_sites * 8;
    char *hp = (char *)pinned_buf(ctx, pin_off + bytes);
    int32_t *d_sites = aalloc<int32_t>(ctx, 2 * (size_t)n_sites), *d_counts = aalloc<int32_t>(ctx, 10 * (size_t)n_sites);
    if (!hp || !d_sites || !d_counts) return 1;
    int32_t *hs = (int32_t *)(hp + pin_off);
    for (int64_t i = 0; i < n_sites; i++) { hs[2 * i] = h_contig_id[i]; hs[2 * i + 1] = h_pos[i]; }
    HIPCHK(hipMemcpyAsync(d_sites, hs, bytes, hipMemcpyHostToDevice, st));
    SPAN_EVENT(ev0);
    PROF_LAUNCH(k_recall, dim3((unsigned)n_sites), dim3(n_sites >= 2048 ? 256 : 1024), 0, st, R, P, (const int32_t *)d_sites, (uint32_t)n_sites, d_counts);
    KCHK();
    SPAN_EVENT(ev1);
    if (int rc = read_back(ctx, d_counts, h_counts, 10 * (size_t)n_sites * 4)) return rc;
    prof_collect(ctx);
    float ms = 0.f;
    if (ctx->span_on && hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1) == hipSuccess) ctx->last_ms = ms;
    ctx->call_ok = true;
    return 0;
}

// ---- recall without any allocation ------------------------------------------------------------------------------------
// mfk_recall_noalloc: mfk_recall for callers that cannot allocate (HIP graph capture around them, serving loops, harnesses
// that forbid hipMalloc).  What mfk_recall takes from the arena and the growable pinned buffer -- the contig header, the
// staged site list and the device-side counts -- lives in ONE file-scope device array of fixed size instead.  Every GPU
// holds its own copy of that array at its own address: the address is looked up once per device (noalloc_setup, called by
// mfk_create with the device current) and only that pointer is ever handed to kernels and copies.  The pinned staging
// block on the host side is allocated by mfk_create as well.  All calls on one GPU share the array (and keep the last
// call's bytes in it): calls that overlap on one GPU race, whatever ctx they use.  Nothing in it has to start at zero:
// k_recall writes all ten counts of every site it is given and reads nothing it has not been sent in this call.
#define NA_MAX_SITES 16384
#define NA_MAX_CONTIGS 4096
#define NA_HDR_BYTES ((((size_t)(NA_MAX_CONTIGS + 1) * 8 + (size_t)NA_MAX_CONTIGS * 8) + 255) & ~(size_t)255)   // = upload_reads_pinned_bytes(NA_MAX_CONTIGS)
#define NA_SITES_BYTES ((size_t)NA_MAX_SITES * 8)
#define NA_COUNTS_BYTES ((size_t)NA_MAX_SITES * 40)
#define NA_BYTES (NA_HDR_BYTES + NA_SITES_BYTES + NA_COUNTS_BYTES)   // 852,224 bytes per GPU
static __device__ __align__(256) unsigned char scratch[NA_BYTES];
#define NA_MAX_DEVICES 64
static unsigned char *g_na_ptr[NA_MAX_DEVICES];   // device address of `scratch` on each GPU (the array's name is no device address in host code)

int noalloc_setup(mfk_ctx *ctx) {
    if (ctx->device < 0 || ctx->device >= NA_MAX_DEVICES) return 0;   // such a device simply has no no-allocation entry point
    HIPCHK(hipSetDevice(ctx->device));
    if (!g_na_ptr[ctx->device]) {
        unsigned char *p = nullptr;
        HIPCHK(hipGetSymbolAddress((void **)&p, HIP_SYMBOL(scratch)));
        if (!p || ((uintptr_t)p & 255) != 0) { snprintf(ctx->err, sizeof ctx->err, "no-allocation scratch array has no usable device address"); return 1; }
        g_na_ptr[ctx->device] = p;
    }
    HIPCHK(hipHostMalloc((void **)&ctx->na_pinned, NA_BYTES, hipHostMallocDefault));
    return 0;
}

void noalloc_teardown(mfk_ctx *ctx) {
    if (ctx->na_pinned) (void)hipHostFree(ctx->na_pinned);
    ctx->na_pinned = nullptr;
}

extern "C" int mfk_recall_noalloc(mfk_ctx *ctx, const mfk_reads *reads, const mfk_params *params, const int32_t *h_contig_id, const int32_t *h_pos,
                                  int64_t n_sites, int32_t *h_counts) {
    if (!ctx || !reads || !params || n_sites < 0 || (n_sites > 0 && (!h_contig_id || !h_pos || !h_counts))) return 1;
    HIPCHK(hipSetDevice(ctx->device));
    ctx->err[0] = 0;
    if (n_sites == 0) return 0;
    if (n_sites > NA_MAX_SITES) { snprintf(ctx->err, sizeof ctx->err, "mfk_recall_noalloc takes at most %d sites per call (got %lld): split the list or use mfk_recall", NA_MAX_SITES, (long long)n_sites); return 9; }
    if (reads->n_contigs <= 0 || reads->n_contigs > NA_MAX_CONTIGS) { snprintf(ctx->err, sizeof ctx->err, "mfk_recall_noalloc takes at most %d contigs (got %d): use mfk_recall", NA_MAX_CONTIGS, reads->n_contigs); return 9; }
    unsigned char *dev = (ctx->device >= 0 && ctx->device < NA_MAX_DEVICES) ? g_na_ptr[ctx->device] : nullptr;
    if (!dev || !ctx->na_pinned) { snprintf(ctx->err, sizeof ctx->err, "the no-allocation scratch was not set up for device %d", ctx->device); return 1; }
    for (int64_t i = 0; i < n_sites; i++) {
        if (h_contig_id[i] < 0 || h_contig_id[i] >= reads->n_contigs) { snprintf(ctx->err, sizeof ctx->err, "site %lld: contig id %d out of range", (long long)i, h_contig_id[i]); return 1; }
        if (h_pos[i] < -1 || h_pos[i] > reads->h_contig_len[h_contig_id[i]]) { snprintf(ctx->err, sizeof ctx->err, "site %lld: position %d outside contig", (long long)i, h_pos[i]); return 1; }
    }
    hipStream_t st = ctx->stream;
    // the pinned block may still be the source / target of the previous call's copies only if that call failed half-way
    HIPCHK(hipStreamSynchronize(st));
    DevReads R; DevParams P;
    char *h_hdr = ctx->na_pinned;
    int32_t *h_sites = (int32_t *)(ctx->na_pinned + NA_HDR_BYTES), *h_cnt = (int32_t *)(ctx->na_pinned + NA_HDR_BYTES + NA_SITES_BYTES);
    size_t used = 0;
    if (int rc = prepare_reads(ctx, reads, params, &R, &P, h_hdr, &used)) return rc;
    unsigned char *d_hdr = dev, *d_sites = dev + NA_HDR_BYTES, *d_cnt = dev + NA_HDR_BYTES + NA_SITES_BYTES;
    set_header_ptrs(&R, (char *)d_hdr);
    for (int64_t i = 0; i < n_sites; i++) { h_sites[2 * i] = h_contig_id[i]; h_sites[2 * i + 1] = h_pos[i]; }
    HIPCHK(hipMemcpyAsync(d_hdr, h_hdr, (used + 15) & ~(size_t)15, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_sites, h_sites, (size_t)n_sites * 8, hipMemcpyHostToDevice, st));
    SPAN_EVENT(ev0);
    hipLaunchKernelGGL(k_recall, dim3((unsigned)n_sites), dim3(n_sites >= 2048 ? 256 : 1024), 0, st, R, P, (const int32_t *)d_sites, (uint32_t)n_sites, (int32_t *)d_cnt);
    KCHK();
    SPAN_EVENT(ev1);
    HIPCHK(hipMemcpyAsync(h_cnt, d_cnt, (size_t)n_sites * 40, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    memcpy(h_counts, h_cnt, (size_t)n_sites * 40);
    float ms = 0.f;
    if (ctx->span_on && hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1) == hipSuccess) ctx->last_ms = ms;
    return 0;
}
