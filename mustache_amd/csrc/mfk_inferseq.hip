// mfk_inferseq.hip -- pairing of flank alignments for inferseq-reference / -assembly (replaces what the reference's
// InferSequence does after the aligner has run, inferseq.py:56-70: __prefilter_reads 103-126, AlignedPairs.add_read /
// match_forward_reverse 214-290, filter_smallest_overlapping 321-362, filter_pairs_max_internal_softclip_prop 364-380,
// filter_best_alignments 305-318, filter_pairs_size 382-390).
//
// One workgroup per pair id ("group").  The reference keeps, per group, four read lists and four position -> read
// dictionaries and walks them with sorted lists and bisection; here every alignment is a thread and every question the
// reference asks a dictionary becomes a scan over the group's alignments (a group is the handful of places where the two
// flanks of one insertion align: tens of records, thousands in a repeat-rich genome):
//   dictionary entry        = the LAST kept read of its (strand, mate, contig, position)         -> rep[]
//   takeClosestLarger       = smallest reference_end > reference_start among the other mate's reverse entries
//   takeClosestSmaller      = largest reference_start < reference_end among the other mate's forward entries
//   `added` (name tuples)   = a candidate is dropped if one created earlier has the same names and coordinates
//   set iteration order     = creation order (forward reads of mate 1, of mate 2, reverse reads of mate 1, of mate 2, each
//                             in BAM order): the reference's own order here depends on object addresses.
#include "mfk_host.h"

#include <algorithm>
#include <numeric>

struct IsqWork {
    const mfk_aln *a;            // sorted by group, BAM order inside a group
    const uint32_t *goff;        // n_groups + 1
    uint8_t *alive, *rep, *valid, *valid2, *keep, *fin;
    int32_t *pf, *pr;            // candidate c (made by read c): forward / reverse read, sorted positions
};

__device__ __forceinline__ uint32_t d_isq_ord(const mfk_aln &x, uint32_t local, uint32_t ng) {
    return (uint32_t)((x.is_reverse ? 2 : 0) + (x.mate == 1 ? 0 : 1)) * ng + local;
}
__device__ __forceinline__ bool d_isq_same_read(const mfk_aln &x, const mfk_aln &y) {   // the fields of the `added` tuple
    return x.name == y.name && x.contig_id == y.contig_id && x.ref_start == y.ref_start && x.ref_end == y.ref_end;
}

__global__ __launch_bounds__(256) void k_inferseq_group(IsqWork W, double min_ident, double max_clip, long long max_size, long long min_size) {
    __shared__ long long s_best;
    const uint32_t b = W.goff[blockIdx.x], e = W.goff[blockIdx.x + 1], ng = e - b;
    const mfk_aln *A = W.a;
    if (threadIdx.x == 0) s_best = (long long)0x8000000000000000ull;
    // __prefilter_reads
    for (uint32_t i = b + threadIdx.x; i < e; i += blockDim.x) {
        const mfk_aln x = A[i];
        bool ok = !((double)(x.aln_len - x.nm) / (double)x.aln_len < min_ident);
        if (!x.is_reverse) { if (x.left_clip > 1 && x.ref_start - 1 >= 0) ok = false; }
        else { if (x.right_clip > 1 && x.ref_end < x.contig_len) ok = false; }
        W.alive[i] = ok ? 1 : 0;
        W.keep[i] = 0;
    }
    __syncthreads();
    // dictionary entries: the last kept read of a (strand, mate, contig, position) key
    for (uint32_t i = b + threadIdx.x; i < e; i += blockDim.x) {
        bool rep = W.alive[i] != 0;
        if (rep) {
            const mfk_aln x = A[i];
            const int32_t key = x.is_reverse ? x.ref_end : x.ref_start;
            for (uint32_t j = i + 1; j < e && rep; j++) {
                const mfk_aln y = A[j];
                if (W.alive[j] && y.is_reverse == x.is_reverse && y.mate == x.mate && y.contig_id == x.contig_id && (y.is_reverse ? y.ref_end : y.ref_start) == key) rep = false;
            }
        }
        W.rep[i] = rep ? 1 : 0;
    }
    __syncthreads();
    // match_forward_reverse: one candidate per kept read
    for (uint32_t i = b + threadIdx.x; i < e; i += blockDim.x) {
        int32_t f = -1, r = -1;
        if (W.alive[i]) {
            const mfk_aln x = A[i];
            int32_t best = -1, bestpos = 0;
            for (uint32_t j = b; j < e; j++) {
                if (!W.rep[j]) continue;
                const mfk_aln y = A[j];
                if (y.is_reverse == x.is_reverse || y.mate == x.mate || y.contig_id != x.contig_id) continue;
                if (!x.is_reverse) { if (y.ref_end > x.ref_start && (best < 0 || y.ref_end < bestpos)) { best = (int32_t)j; bestpos = y.ref_end; } }
                else { if (y.ref_start < x.ref_end && (best < 0 || y.ref_start > bestpos)) { best = (int32_t)j; bestpos = y.ref_start; } }
            }
            if (best >= 0) { if (!x.is_reverse) { f = (int32_t)i; r = best; } else { f = best; r = (int32_t)i; } }
        }
        W.pf[i] = f; W.pr[i] = r;
        W.valid[i] = f >= 0 ? 1 : 0;
    }
    __syncthreads();
    // the `added` set: drop a candidate when an earlier one names the same two alignments
    for (uint32_t c = b + threadIdx.x; c < e; c += blockDim.x) {
        bool v = W.valid[c] != 0;
        if (v) {
            const uint32_t oc = d_isq_ord(A[c], c - b, ng);
            const mfk_aln fx = A[W.pf[c]], rx = A[W.pr[c]];
            for (uint32_t q = b; q < e && v; q++) {
                if (q == c || !W.valid[q] || d_isq_ord(A[q], q - b, ng) > oc) continue;
                if (d_isq_same_read(A[W.pf[q]], fx) && d_isq_same_read(A[W.pr[q]], rx)) v = false;
            }
        }
        W.valid2[c] = v ? 1 : 0;
    }
    __syncthreads();
    // filter_smallest_overlapping
    for (uint32_t p = b + threadIdx.x; p < e; p += blockDim.x) {
        if (!W.valid2[p]) continue;
        const mfk_aln fp = A[W.pf[p]], rp = A[W.pr[p]];
        uint32_t nf = 0, nr = 0;
        for (uint32_t q = b; q < e; q++) {
            if (!W.valid2[q]) continue;
            const mfk_aln fq = A[W.pf[q]], rq = A[W.pr[q]];
            if (fq.name == fp.name && fq.ref_start == fp.ref_start) nf++;
            if (rq.name == rp.name && rq.ref_end == rp.ref_end) nr++;
        }
        if (nf == 1 && nr == 1) { W.keep[p] = 1; continue; }
        int32_t best = -1; long long bl = 0; uint32_t bo = 0;
        for (uint32_t q = b; q < e; q++) {
            if (!W.valid2[q]) continue;
            const mfk_aln fq = A[W.pf[q]], rq = A[W.pr[q]];
            const bool inf = fq.name == fp.name && fq.ref_start == fp.ref_start, inr = rq.name == rp.name && rq.ref_end == rp.ref_end;
            const bool in = (nf > 1 && nr == 1) ? inf : ((nf == 1 && nr > 1) ? inr : (inf || inr));
            if (!in) continue;
            const long long len = (long long)rq.ref_end - fq.ref_start;
            const uint32_t oq = d_isq_ord(A[q], q - b, ng);
            if (best < 0 || len < bl || (len == bl && oq < bo)) { best = (int32_t)q; bl = len; bo = oq; }
        }
        if (best >= 0) W.keep[best] = 1;   // several threads may store the same 1
    }
    __syncthreads();
    // filter_pairs_max_internal_softclip_prop, then the best alignment score of the group
    for (uint32_t p = b + threadIdx.x; p < e; p += blockDim.x) {
        bool k = W.keep[p] != 0;
        if (k) {
            const mfk_aln f = A[W.pf[p]], r = A[W.pr[p]];
            if (f.right_clip > 0 && f.ref_end < r.ref_end && (double)f.right_clip / (double)f.qlen > max_clip) k = false;
            else if (r.left_clip > 0 && r.ref_start > f.ref_start && (double)r.left_clip / (double)r.qlen > max_clip) k = false;
            if (k) atomicMax(&s_best, (long long)f.as + (long long)r.as);
        }
        W.fin[p] = k ? 1 : 0;
    }
    __syncthreads();
    const long long best_as = s_best;
    // filter_best_alignments + filter_pairs_size
    for (uint32_t p = b + threadIdx.x; p < e; p += blockDim.x) {
        if (!W.fin[p]) continue;
        const mfk_aln f = A[W.pf[p]], r = A[W.pr[p]];
        const long long len = (long long)r.ref_end - f.ref_start;
        if ((long long)f.as + (long long)r.as != best_as || len < min_size || len > max_size) W.fin[p] = 0;
    }
}

extern "C" int mfk_inferseq_pairs(mfk_ctx *ctx, const mfk_aln *alns, int64_t n, double min_perc_identity, double max_internal_softclip_prop,
                                  int64_t max_inferseq_size, int64_t min_inferseq_size, int64_t *n_pairs) {
    if (!ctx || n < 0 || !n_pairs || (n > 0 && !alns)) return 1;
    HIPCHK(hipSetDevice(ctx->device));
    ctx->err[0] = 0;
    ctx->isq_f.clear(); ctx->isq_r.clear();
    *n_pairs = 0;
    if (n == 0) return 0;
    if (n > 0x7FFFFFF0ll) { snprintf(ctx->err, sizeof ctx->err, "too many alignments in one call"); return 1; }
    int32_t n_groups = 0;
    for (int64_t i = 0; i < n; i++) {
        if (alns[i].aln_len == 0) { snprintf(ctx->err, sizeof ctx->err, "ZeroDivisionError: alignment %lld has query_alignment_length 0 (pysamtools.py:176)", (long long)i); return 7; }
        if (alns[i].group < 0) { snprintf(ctx->err, sizeof ctx->err, "alignment %lld: negative group id", (long long)i); return 1; }
        if (alns[i].group + 1 > n_groups) n_groups = alns[i].group + 1;
    }
    // group the records (stable: BAM order inside a group)
    std::vector<int32_t> order((size_t)n);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return alns[x].group < alns[y].group; });
    arena_reset(ctx);
    hipStream_t st = ctx->stream;
    const size_t bytes_a = (size_t)n * sizeof(mfk_aln), bytes_g = ((size_t)n_groups + 1) * 4;
    char *hp = (char *)pinned_buf(ctx, bytes_a + bytes_g + 256);
    if (!hp) return 1;
    mfk_aln *ha = (mfk_aln *)hp;
    uint32_t *hg = (uint32_t *)(hp + ((bytes_a + 255) & ~(size_t)255));
    for (int64_t i = 0; i < n; i++) ha[i] = alns[order[(size_t)i]];
    { int64_t i = 0; for (int32_t g = 0; g <= n_groups; g++) { while (i < n && ha[i].group < g) i++; hg[g] = (uint32_t)i; } }
    IsqWork W;
    mfk_aln *da = aalloc<mfk_aln>(ctx, (size_t)n);
    uint32_t *dg = aalloc<uint32_t>(ctx, (size_t)n_groups + 1);
    W.alive = aalloc<uint8_t>(ctx, (size_t)n); W.rep = aalloc<uint8_t>(ctx, (size_t)n); W.valid = aalloc<uint8_t>(ctx, (size_t)n);
    W.valid2 = aalloc<uint8_t>(ctx, (size_t)n); W.keep = aalloc<uint8_t>(ctx, (size_t)n); W.fin = aalloc<uint8_t>(ctx, (size_t)n);
    W.pf = aalloc<int32_t>(ctx, (size_t)n); W.pr = aalloc<int32_t>(ctx, (size_t)n);
    if (!da || !dg || !W.alive || !W.rep || !W.valid || !W.valid2 || !W.keep || !W.fin || !W.pf || !W.pr) return 1;
    W.a = da; W.goff = dg;
    HIPCHK(hipMemcpyAsync(da, ha, bytes_a, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(dg, hg, bytes_g, hipMemcpyHostToDevice, st));
    HIPCHK(hipEventRecord(ctx->ev0, st));
    PROF_LAUNCH(k_inferseq_group, dim3((unsigned)n_groups), dim3(256), 0, st, W, min_perc_identity, max_internal_softclip_prop,
                (long long)max_inferseq_size, (long long)min_inferseq_size);
    KCHK();
    HIPCHK(hipEventRecord(ctx->ev1, st));
    std::vector<uint8_t> fin((size_t)n);
    std::vector<int32_t> pf((size_t)n), pr((size_t)n);
    {
        ReadBatch rb(ctx);
        rb.add(W.fin, fin.data(), (size_t)n); rb.add(W.pf, pf.data(), (size_t)n * 4); rb.add(W.pr, pr.data(), (size_t)n * 4);
        if (int rc = rb.flush()) return rc;
    }
    prof_collect(ctx);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1) == hipSuccess) ctx->last_ms = ms;
    // surviving pairs: groups ascending, creation order inside a group; indices back to the caller's numbering
    // (ha lives in the pinned buffer that the read-back reused: take the records from the caller's array)
    std::vector<std::pair<uint32_t, int64_t>> tmp;
    for (int32_t g = 0; g < n_groups; g++) {
        int64_t b = 0, e = 0;
        { auto lo = std::lower_bound(order.begin(), order.end(), g, [&](int32_t x, int32_t gg) { return alns[x].group < gg; });
          auto hi = std::upper_bound(order.begin(), order.end(), g, [&](int32_t gg, int32_t x) { return gg < alns[x].group; });
          b = lo - order.begin(); e = hi - order.begin(); }
        tmp.clear();
        for (int64_t c = b; c < e; c++)
            if (fin[(size_t)c]) {
                const mfk_aln &x = alns[order[(size_t)c]];
                tmp.push_back({(uint32_t)((x.is_reverse ? 2 : 0) + (x.mate == 1 ? 0 : 1)) * (uint32_t)(e - b) + (uint32_t)(c - b), c});
            }
        std::sort(tmp.begin(), tmp.end());
        for (auto &t : tmp) { ctx->isq_f.push_back(order[(size_t)pf[(size_t)t.second]]); ctx->isq_r.push_back(order[(size_t)pr[(size_t)t.second]]); }
    }
    *n_pairs = (int64_t)ctx->isq_f.size();
    ctx->call_ok = true;   // only ctx->stream was used and it has been read back: the next arena_reset need not drain anything
    return 0;
}

// ---- inferseq-database: its own pairing code (inferseqdatabase.py:116-322) ------------------------------------------------------
// One workgroup per pair id.  The reference keeps reads[pair][contig][flank] lists in BAM order and, per contig that holds
// alignments of both flanks (get_pairs 183-208): match_pairs (269-300) sorts all (flank-1 read, flank-2 read) couples by the
// distance of their positions, farthest first (a stable sort: ties in row-major order), and takes couples greedily; a couple is
// kept when its reads lie at the two ends of the database element on opposite strands (pair_mapped_both_ends 234-266); per pair
// id only the couples with the best summed alignment score stay (keep_best_alignment_score 303-318).
// The greedy choice is resolved in parallel rounds like assign_pairs: a couple is taken when it is the first, in the sort
// order, of all undecided couples of its flank-1 read AND of its flank-2 read.
//   out: for a flank-1 read i that got a partner: part[i] = partner, dist[i], ford[i] = first kept read of that contig in
//   the group (the order in which the reference meets the contigs), fin[i] = 1 if the couple survives all filters.
struct IsqDbWork {
    const mfk_aln *a; const uint32_t *goff;
    uint8_t *alive, *state, *fin;      // state of a read in the matching of its contig: 0 undecided, 1 taken
    int32_t *part, *cand, *ford; long long *dist;
};

__device__ __forceinline__ long long d_isqdb_dist(const mfk_aln &r1, const mfk_aln &r2) {
    const long long p1 = r1.is_reverse ? r1.ref_end : r1.ref_start;
    const long long p2 = r2.is_reverse ? r2.ref_end : r1.ref_start;   // reference inferseqdatabase.py:284: read1's start, as written there
    return p1 > p2 ? p1 - p2 : p2 - p1;
}

__global__ __launch_bounds__(256) void k_inferseq_db_group(IsqDbWork W, double min_ident, double max_clip, long long max_edge) {
    __shared__ long long s_best;
    __shared__ int s_undecided;
    const uint32_t b = W.goff[blockIdx.x], e = W.goff[blockIdx.x + 1];
    const mfk_aln *A = W.a;
    if (threadIdx.x == 0) s_best = 0;   // best_score starts at 0 (inferseqdatabase.py:306)
    // prefilter_reads 142-180
    for (uint32_t i = b + threadIdx.x; i < e; i += blockDim.x) {
        const mfk_aln x = A[i];
        bool ok = !((double)(x.aln_len - x.nm) / (double)x.aln_len < min_ident);
        if (ok && !x.is_reverse) {
            if (x.right_clip > 0 && x.ref_end < x.contig_len && (double)x.right_clip / (double)x.qlen > max_clip) ok = false;
            else if ((long long)x.ref_start > max_edge) ok = false;
            else if (x.left_clip > 0) { long long d = (long long)x.ref_start - x.left_clip; if (d < 0) d = -d; if (d > max_edge) ok = false; }
        } else if (ok) {
            if (x.left_clip > 0 && x.ref_start - 1 >= 0 && (double)x.left_clip / (double)x.qlen > max_clip) ok = false;
            else if ((long long)x.contig_len - x.ref_end > max_edge) ok = false;
            else if (x.right_clip > 0) { long long d = (long long)x.contig_len - ((long long)x.ref_end + x.right_clip); if (d < 0) d = -d; if (d > max_edge) ok = false; }
        }
        W.alive[i] = ok ? 1 : 0;
        W.state[i] = 0; W.fin[i] = 0; W.part[i] = -1; W.cand[i] = -1; W.dist[i] = 0; W.ford[i] = -1;
    }
    __syncthreads();
    // per read: first kept read of its contig; does the contig pass reads_mapped_both_ends (211-232) with both flanks present
    for (uint32_t i = b + threadIdx.x; i < e; i += blockDim.x) {
        if (!W.alive[i]) continue;
        const mfk_aln x = A[i];
        int32_t first = -1;
        bool has1 = false, has2 = false, f1 = false, f2 = false, t1 = false, t2 = false;
        for (uint32_t j = b; j < e; j++) {
            if (!W.alive[j]) continue;
            const mfk_aln y = A[j];
            if (y.contig_id != x.contig_id) continue;
            if (first < 0) first = (int32_t)j;
            if (y.mate == 1) has1 = true; else has2 = true;
            if (y.is_reverse && (long long)y.contig_len - y.ref_end <= max_edge) { if (y.mate == 1) t1 = true; else t2 = true; }
            else if (!y.is_reverse && (long long)y.ref_start <= max_edge) { if (y.mate == 1) f1 = true; else f2 = true; }
        }
        const bool go = has1 && has2 && ((f1 && t2) || (t1 && f2));
        W.ford[i] = first;
        if (!go) W.state[i] = 2;   // the contig takes no part in the matching
    }
    __syncthreads();
    // match_pairs 269-300 in parallel rounds.  Order of the couples: distance descending, then flank-1 read, then flank-2 read
    // (both in BAM order).
    for (uint32_t round = 0; round <= e - b; round++) {
        if (threadIdx.x == 0) s_undecided = 0;
        __syncthreads();
        for (uint32_t i = b + threadIdx.x; i < e; i += blockDim.x) {   // every undecided read names its first couple
            W.cand[i] = -1;
            if (!W.alive[i] || W.state[i]) continue;
            const mfk_aln x = A[i];
            int32_t best = -1; long long bd = -1;
            for (uint32_t j = b; j < e; j++) {
                if (!W.alive[j] || W.state[j]) continue;
                const mfk_aln y = A[j];
                if (y.contig_id != x.contig_id || y.mate == x.mate) continue;
                const long long d = x.mate == 1 ? d_isqdb_dist(x, y) : d_isqdb_dist(y, x);
                if (d > bd) { bd = d; best = (int32_t)j; }   // ties: the earlier partner (j ascends)
            }
            W.cand[i] = best;
            if (best >= 0) s_undecided = 1;
        }
        __syncthreads();
        if (!s_undecided) break;
        for (uint32_t i = b + threadIdx.x; i < e; i += blockDim.x) {
            if (!W.alive[i] || W.state[i] || A[i].mate != 1) continue;
            const int32_t j = W.cand[i];
            if (j >= 0 && W.cand[j] == (int32_t)i) { W.part[i] = j; W.dist[i] = d_isqdb_dist(A[i], A[j]); }
        }
        __syncthreads();
        for (uint32_t i = b + threadIdx.x; i < e; i += blockDim.x) {
            if (!W.alive[i] || W.state[i]) continue;
            if (A[i].mate == 1) { if (W.part[i] >= 0) W.state[i] = 1; }
            else { const int32_t c = W.cand[i]; if (c >= 0 && W.part[c] == (int32_t)i) W.state[i] = 1; }
            if (W.cand[i] < 0) W.state[i] = 2;   // nobody left to pair with
        }
        __syncthreads();
    }
    __syncthreads();
    // pair_mapped_both_ends 234-266 + the strand test of get_pairs 200-206, then keep_best_alignment_score 303-318
    for (uint32_t i = b + threadIdx.x; i < e; i += blockDim.x) {
        const int32_t j = W.part[i];
        if (j < 0) continue;
        const mfk_aln r1 = A[i], r2 = A[j];
        bool f1 = false, t1 = false, f2 = false, t2 = false;
        if (r1.is_reverse && (long long)r1.contig_len - r1.ref_end <= max_edge) t1 = true; else if (!r1.is_reverse && (long long)r1.ref_start <= max_edge) f1 = true;
        if (r2.is_reverse && (long long)r2.contig_len - r2.ref_end <= max_edge) t2 = true; else if (!r2.is_reverse && (long long)r2.ref_start <= max_edge) f2 = true;
        const bool ok = ((f1 && t2) || (t1 && f2)) && (r1.is_reverse != r2.is_reverse);
        if (ok) { W.fin[i] = 1; atomicMax(&s_best, (long long)r1.as + (long long)r2.as); }
    }
    __syncthreads();
    const long long best_as = s_best;
    for (uint32_t i = b + threadIdx.x; i < e; i += blockDim.x)
        if (W.fin[i] && (long long)A[i].as + (long long)A[W.part[i]].as != best_as) W.fin[i] = 0;
}

extern "C" int mfk_inferseq_db_pairs(mfk_ctx *ctx, const mfk_aln *alns, int64_t n, double min_perc_identity, double max_internal_softclip_prop,
                                     int64_t max_edge_distance, int64_t *n_pairs) {
    if (!ctx || n < 0 || !n_pairs || (n > 0 && !alns)) return 1;
    HIPCHK(hipSetDevice(ctx->device));
    ctx->err[0] = 0;
    ctx->isq_f.clear(); ctx->isq_r.clear();
    *n_pairs = 0;
    if (n == 0) return 0;
    if (n > 0x7FFFFFF0ll) { snprintf(ctx->err, sizeof ctx->err, "too many alignments in one call"); return 1; }
    int32_t n_groups = 0;
    for (int64_t i = 0; i < n; i++) {
        if (alns[i].aln_len == 0) { snprintf(ctx->err, sizeof ctx->err, "ZeroDivisionError: alignment %lld has query_alignment_length 0 (pysamtools.py:176)", (long long)i); return 7; }
        if (alns[i].qlen == 0) { snprintf(ctx->err, sizeof ctx->err, "ZeroDivisionError: alignment %lld has an empty query sequence (sctools.py:182-186)", (long long)i); return 7; }
        if (alns[i].group < 0) { snprintf(ctx->err, sizeof ctx->err, "alignment %lld: negative group id", (long long)i); return 1; }
        if (alns[i].group + 1 > n_groups) n_groups = alns[i].group + 1;
    }
    std::vector<int32_t> order((size_t)n);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return alns[x].group < alns[y].group; });
    arena_reset(ctx);
    hipStream_t st = ctx->stream;
    const size_t bytes_a = (size_t)n * sizeof(mfk_aln), bytes_g = ((size_t)n_groups + 1) * 4;
    char *hp = (char *)pinned_buf(ctx, bytes_a + bytes_g + 256);
    if (!hp) return 1;
    mfk_aln *ha = (mfk_aln *)hp;
    uint32_t *hg = (uint32_t *)(hp + ((bytes_a + 255) & ~(size_t)255));
    for (int64_t i = 0; i < n; i++) ha[i] = alns[order[(size_t)i]];
    { int64_t i = 0; for (int32_t g = 0; g <= n_groups; g++) { while (i < n && ha[i].group < g) i++; hg[g] = (uint32_t)i; } }
    std::vector<uint32_t> goff(hg, hg + n_groups + 1);
    IsqDbWork W;
    mfk_aln *da = aalloc<mfk_aln>(ctx, (size_t)n);
    uint32_t *dg = aalloc<uint32_t>(ctx, (size_t)n_groups + 1);
    W.alive = aalloc<uint8_t>(ctx, (size_t)n); W.state = aalloc<uint8_t>(ctx, (size_t)n); W.fin = aalloc<uint8_t>(ctx, (size_t)n);
    W.part = aalloc<int32_t>(ctx, (size_t)n); W.cand = aalloc<int32_t>(ctx, (size_t)n); W.ford = aalloc<int32_t>(ctx, (size_t)n);
    W.dist = aalloc<long long>(ctx, (size_t)n);
    if (!da || !dg || !W.alive || !W.state || !W.fin || !W.part || !W.cand || !W.ford || !W.dist) return 1;
    W.a = da; W.goff = dg;
    HIPCHK(hipMemcpyAsync(da, ha, bytes_a, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(dg, hg, bytes_g, hipMemcpyHostToDevice, st));
    HIPCHK(hipEventRecord(ctx->ev0, st));
    PROF_LAUNCH(k_inferseq_db_group, dim3((unsigned)n_groups), dim3(256), 0, st, W, min_perc_identity, max_internal_softclip_prop, (long long)max_edge_distance);
    KCHK();
    HIPCHK(hipEventRecord(ctx->ev1, st));
    std::vector<uint8_t> fin((size_t)n);
    std::vector<int32_t> part((size_t)n), ford((size_t)n);
    std::vector<long long> dist((size_t)n);
    {
        ReadBatch rb(ctx);
        rb.add(W.fin, fin.data(), (size_t)n); rb.add(W.part, part.data(), (size_t)n * 4); rb.add(W.ford, ford.data(), (size_t)n * 4); rb.add(W.dist, dist.data(), (size_t)n * 8);
        if (int rc = rb.flush()) return rc;
    }
    prof_collect(ctx);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1) == hipSuccess) ctx->last_ms = ms;
    // the reference's list order: pair ids (groups ascending here), contigs as first met among the kept reads, couples in the
    // order match_pairs took them (distance descending, flank-1 read, flank-2 read); (forward, reverse) = the read on the + strand first
    struct Row { int32_t ford; long long dist; int32_t i, j; };
    std::vector<Row> rows;
    for (int32_t g = 0; g < n_groups; g++) {
        rows.clear();
        for (uint32_t c = goff[(size_t)g]; c < goff[(size_t)g + 1]; c++)
            if (fin[c]) rows.push_back({ford[c], dist[c], (int32_t)c, part[c]});
        std::sort(rows.begin(), rows.end(), [](const Row &x, const Row &y) {
            if (x.ford != y.ford) return x.ford < y.ford;
            if (x.dist != y.dist) return x.dist > y.dist;
            if (x.i != y.i) return x.i < y.i;
            return x.j < y.j;
        });
        for (auto &r : rows) {
            const int32_t a = order[(size_t)r.i], b2 = order[(size_t)r.j];
            if (alns[a].is_reverse) { ctx->isq_f.push_back(b2); ctx->isq_r.push_back(a); }
            else { ctx->isq_f.push_back(a); ctx->isq_r.push_back(b2); }
        }
    }
    *n_pairs = (int64_t)ctx->isq_f.size();
    ctx->call_ok = true;   // only ctx->stream was used and it has been read back: the next arena_reset need not drain anything
    return 0;
}

extern "C" int mfk_inferseq_pairs_fetch(mfk_ctx *ctx, int32_t *forward, int32_t *reverse) {
    if (!ctx) return 1;
    if (!ctx->isq_f.empty()) {
        if (!forward || !reverse) return 1;
        memcpy(forward, ctx->isq_f.data(), ctx->isq_f.size() * sizeof(int32_t));
        memcpy(reverse, ctx->isq_r.data(), ctx->isq_r.size() * sizeof(int32_t));
    }
    return 0;
}
