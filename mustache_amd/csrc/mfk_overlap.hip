// mfk_overlap.hip -- inferseq-overlap: the best ungapped sliding overlap of the 5p flank (query) and the
// 3p flank (reference), replaces get_best_sliding_alignment, inferseqoverlap.py:94-164.
//
// The reference walks a window state machine; the windows it visits are exactly the Lq + Lr - 1 diagonals
// d = q_start - r_start = Lq-1, Lq-2, ..., -(Lr-1), each with its full overlap, in that order.  Visit t is diagonal
// Lq-1-t: q_start = max(d, 0), r_start = max(-d, 0), length = min(Lq - q_start, Lr - r_start).  Score = matches -
// mismatches; the first visit with the strictly largest score wins (inferseqoverlap.py:133).
// One workgroup per pair, one thread per diagonal; both sequences sit in LDS when they fit.
#include "mfk_host.h"

#define OV_LDS_BYTES 32768

__global__ __launch_bounds__(256) void k_overlap(const char *s5, const int64_t *off5, const char *s3, const int64_t *off3, uint32_t n, int32_t *out) {
    __shared__ char lds[OV_LDS_BYTES];
    __shared__ unsigned long long best;
    const uint32_t k = blockIdx.x;
    if (k >= n) return;
    const char *q = s5 + off5[k], *r = s3 + off3[k];
    const int64_t Lq = off5[k + 1] - off5[k], Lr = off3[k + 1] - off3[k];
    if (threadIdx.x == 0) best = 0;
    if (Lq + Lr <= OV_LDS_BYTES) {
        for (int64_t i = threadIdx.x; i < Lq; i += blockDim.x) lds[i] = q[i];
        for (int64_t i = threadIdx.x; i < Lr; i += blockDim.x) lds[Lq + i] = r[i];
        q = lds; r = lds + Lq;
    }
    __syncthreads();
    const int64_t T = Lq + Lr - 1;
    for (int64_t t = threadIdx.x; t < T; t += blockDim.x) {
        const int64_t d = Lq - 1 - t;
        const int64_t qs = d > 0 ? d : 0, rs = d < 0 ? -d : 0;
        const int64_t len = (Lq - qs) < (Lr - rs) ? (Lq - qs) : (Lr - rs);
        int64_t score = 0;
        for (int64_t i = 0; i < len; i++) score += q[qs + i] == r[rs + i] ? 1 : -1;
        // larger score first, then the earlier visit: (score + 2^31) << 32 | (2^32 - 1 - t)
        const unsigned long long key = ((unsigned long long)(score + 0x80000000ll) << 32) | (unsigned long long)(0xFFFFFFFFull - (unsigned long long)t);
        atomicMax(&best, key);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int64_t score = (int64_t)(best >> 32) - 0x80000000ll;
        const int64_t t = (int64_t)(0xFFFFFFFFull - (best & 0xFFFFFFFFull));
        const int64_t d = Lq - 1 - t;
        const int64_t qs = d > 0 ? d : 0, rs = d < 0 ? -d : 0;
        const int64_t len = (Lq - qs) < (Lr - rs) ? (Lq - qs) : (Lr - rs);
        int32_t *o = out + 6 * (size_t)k;
        o[0] = (int32_t)score; o[1] = (int32_t)((len - score) / 2);
        o[2] = (int32_t)rs; o[3] = (int32_t)(rs + len); o[4] = (int32_t)qs; o[5] = (int32_t)(qs + len);
    }
}

extern "C" int mfk_overlap(mfk_ctx *ctx, const char *seqs5, const int64_t *off5, const char *seqs3, const int64_t *off3, int64_t n, int32_t *out) {
    if (!ctx || n < 0 || (n > 0 && (!seqs5 || !off5 || !seqs3 || !off3 || !out))) return 1;
    HIPCHK(hipSetDevice(ctx->device));
    ctx->err[0] = 0;
    if (n == 0) return 0;
    if (n > 0x7FFFFFF0ll / 6) { snprintf(ctx->err, sizeof ctx->err, "too many pairs in one call"); return 1; }
    for (int64_t k = 0; k < n; k++) {
        const int64_t lq = off5[k + 1] - off5[k], lr = off3[k + 1] - off3[k];
        // the reference indexes the first window of both strings: an empty string is an IndexError there
        if (lq <= 0 || lr <= 0) { snprintf(ctx->err, sizeof ctx->err, "IndexError: pair %lld has an empty flank sequence (inferseqoverlap.py:126)", (long long)k); return 7; }
        if (lq + lr >= 0x7FFFFFFFll) { snprintf(ctx->err, sizeof ctx->err, "pair %lld: sequences too long", (long long)k); return 1; }
    }
    arena_reset(ctx);
    hipStream_t st = ctx->stream;
    const size_t b5 = (size_t)off5[n], b3 = (size_t)off3[n], bo = ((size_t)n + 1) * 8;
    const size_t o_off5 = 0, o_off3 = o_off5 + bo, o_s5 = o_off3 + bo, o_s3 = o_s5 + ((b5 + 7) & ~(size_t)7), total = o_s3 + ((b3 + 7) & ~(size_t)7);
    char *hp = (char *)pinned_buf(ctx, total);
    char *d = aalloc<char>(ctx, total);
    int32_t *d_out = aalloc<int32_t>(ctx, 6 * (size_t)n);
    if (!hp || !d || !d_out) return 1;
    // offsets rebased to the first sequence
    int64_t *h5 = (int64_t *)(hp + o_off5), *h3 = (int64_t *)(hp + o_off3);
    for (int64_t k = 0; k <= n; k++) { h5[k] = off5[k] - off5[0]; h3[k] = off3[k] - off3[0]; }
    memcpy(hp + o_s5, seqs5 + off5[0], (size_t)(off5[n] - off5[0]));
    memcpy(hp + o_s3, seqs3 + off3[0], (size_t)(off3[n] - off3[0]));
    HIPCHK(hipMemcpyAsync(d, hp, total, hipMemcpyHostToDevice, st));
    HIPCHK(hipEventRecord(ctx->ev0, st));
    PROF_LAUNCH(k_overlap, dim3((unsigned)n), dim3(256), 0, st, (const char *)(d + o_s5), (const int64_t *)(d + o_off5), (const char *)(d + o_s3), (const int64_t *)(d + o_off3), (uint32_t)n, d_out);
    KCHK();
    HIPCHK(hipEventRecord(ctx->ev1, st));
    if (int rc = read_back(ctx, d_out, out, 6 * (size_t)n * 4)) return rc;
    prof_collect(ctx);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1) == hipSuccess) ctx->last_ms = ms;
    ctx->call_ok = true;
    return 0;
}
