// mfk_inflate.hip -- BGZF blocks inflated on the GPU: one 64-lane wave per block (mfk_inflate_core.h says how a wave
// decodes).  SURVEY.md 8(f) N3 lists "optionally GPU inflate" for the BAM ingestion row; the reference gets its inflate from
// htslib inside pysam (findflanks.py:22, 100-121).  This is the first stage of a decode that stays on the device (DESIGN.md
// section 10): compressed bytes in HBM -> inflated stream in HBM.  The shipped stage drivers still inflate on the host
// (libmfkio.so), because with a host parser behind it the inflated stream would have to cross PCIe again.
#include "mfk_host.h"

#define MFK_HD __device__
#define MFK_TABLE __device__ const
#include "mfk_inflate_core.h"

#define INFL_WAVES 4   // decoders per workgroup: 4 x 4 KB of tables, eight workgroups per CU

__global__ __launch_bounds__(64 * INFL_WAVES) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_bgzf_inflate(const uint8_t *comp, const mfk_bgzf_block *blocks, uint32_t n_blocks, uint8_t *out, int32_t *status, uint32_t *n_bad) {
    __shared__ InflTables T[INFL_WAVES];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t b = blockIdx.x * INFL_WAVES + (uint32_t)wave;
    if (b >= n_blocks) return;   // (wave-uniform; nothing below needs the other waves)
    const mfk_bgzf_block blk = blocks[b];
    const int32_t rc = infl_block<64>(comp + blk.comp_off, blk.comp_len, out + blk.out_off, blk.out_len, T[wave], lane);
    if (lane == 0) {
        status[b] = rc;
        if (rc) atomicAdd(n_bad, 1u);
    }
}

// ---- CRC-32 of the inflated blocks (the gzip trailer of a BGZF block holds it; the host reader checks it, so does this path) ----
// A wave per block: lane k takes a 16-byte-aligned slice (one table step per byte, 16 bytes per load), and the 64 slice CRCs are
// folded in lane order with crc(A || B) = crc(A) * x^(8 |B|) + crc(B) in GF(2)[x] / P (the combine of zlib's crc32_combine:
// multiplication mod the CRC polynomial, x^(2^n) from a 32-entry table built by squaring).
#define CRC_POLY 0xedb88320u
__device__ __forceinline__ uint32_t d_multmodp(uint32_t a, uint32_t b) {
    uint32_t p = 0;
#pragma unroll 1
    for (int k = 31; k >= 0; k--) {
        if (a & (1u << k)) p ^= b;
        b = (b & 1u) ? (b >> 1) ^ CRC_POLY : b >> 1;
    }
    return p;
}
__device__ __forceinline__ uint32_t d_x2nmodp(uint32_t n_bytes, const uint32_t *x2n) {   // x^(8 n) mod P
    uint32_t p = 1u << 31, n = n_bytes;
    unsigned k = 3;
    while (n) { if (n & 1u) p = d_multmodp(x2n[k & 31u], p); n >>= 1; k++; }
    return p;
}

__global__ __launch_bounds__(256) void k_bgzf_crc32(const uint8_t *data, const mfk_bgzf_block *blocks, uint32_t n_blocks, const uint32_t *expect, uint32_t *crc_out, uint32_t *n_bad) {
    __shared__ uint32_t tab[256], x2n[32];
    {
        uint32_t c = threadIdx.x;
        for (int k = 0; k < 8; k++) c = (c & 1u) ? (c >> 1) ^ CRC_POLY : c >> 1;
        tab[threadIdx.x] = c;
        if (threadIdx.x == 0) { uint32_t p = 1u << 30; x2n[0] = p; for (int n = 1; n < 32; n++) { p = d_multmodp(p, p); x2n[n] = p; } }
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t b = blockIdx.x * 4 + (uint32_t)wave;
    if (b >= n_blocks) return;
    const mfk_bgzf_block blk = blocks[b];
    const uint8_t *a0 = data + blk.out_off;
    const uint32_t n = blk.out_len;
    uint32_t head = (uint32_t)((16u - (uint32_t)((uintptr_t)a0 & 15u)) & 15u);
    if (head > n) head = n;
    const uint32_t m = n - head;
    const uint32_t S = ((m + 63u) / 64u + 15u) & ~15u;                        // slice length, a multiple of 16
    uint32_t lo = head + (uint32_t)lane * S, hi = lo + S;
    if (lo > n) lo = n;
    if (hi > n) hi = n;
    uint32_t c = 0xFFFFFFFFu, len = hi - lo;
    if (lane == 0) { for (uint32_t i = 0; i < head; i++) c = tab[(c ^ a0[i]) & 0xFFu] ^ (c >> 8); len += head; }
    uint32_t i = lo;
    for (; i + 16 <= hi; i += 16) {
        const uint4 v = *reinterpret_cast<const uint4 *>(a0 + i);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint32_t x = w[q];
#pragma unroll
            for (int t = 0; t < 4; t++) { c = tab[(c ^ x) & 0xFFu] ^ (c >> 8); x >>= 8; }
        }
    }
    for (; i < hi; i++) c = tab[(c ^ a0[i]) & 0xFFu] ^ (c >> 8);
    c = len ? ~c : 0u;                                                        // the CRC of this lane's bytes (of nothing: 0)
    // fold in lane order: after step s lane i (a multiple of 2^(s+1)) holds the bytes of lanes i .. i + 2^(s+1) - 1
    for (int s = 1; s < 64; s <<= 1) {
        const uint32_t pc = __shfl_down(c, s, 64), pl = __shfl_down(len, s, 64);
        if ((lane & (2 * s - 1)) == 0) {
            if (pl) c = len ? (d_multmodp(d_x2nmodp(pl, x2n), c) ^ pc) : pc;
            len += pl;
        }
    }
    if (lane == 0) {
        if (crc_out) crc_out[b] = c;
        if (expect && expect[b] != c) atomicAdd(n_bad, 1u);
    }
}

// CRC-32 of the bytes [out_off, out_off + out_len) of every block; d_expect: NULL or the values they must have (mismatches are
// counted in *d_bad), d_crc_out: NULL or where the values go
int bgzf_crc32_launch(mfk_ctx *ctx, const uint8_t *d_data, const mfk_bgzf_block *d_blocks, uint32_t n, const uint32_t *d_expect, uint32_t *d_crc_out, uint32_t *d_bad, hipStream_t st) {
    if (!n) return 0;
    PROF_LAUNCH(k_bgzf_crc32, dim3((n + 3) / 4), dim3(256), 0, st, d_data, d_blocks, n, d_expect, d_crc_out, d_bad);
    KCHK();
    return 0;
}

// CRC-32 (the one of gzip / zlib) of the byte ranges out_off .. out_off + out_len of `d_data` (DEVICE), one per block (HOST arrays)
extern "C" int mfk_bgzf_crc32(mfk_ctx *ctx, const void *d_data, int64_t data_bytes, const mfk_bgzf_block *blocks, int64_t n_blocks, uint32_t *h_crc) {
    if (!ctx || !h_crc || data_bytes < 0 || n_blocks < 0 || n_blocks > 0x7FFFFFFF || (n_blocks > 0 && (!blocks || !d_data))) return 1;
    ctx->err[0] = 0;
    if (!n_blocks) return 0;
    for (int64_t k = 0; k < n_blocks; k++)
        if (blocks[k].out_off > (uint64_t)data_bytes || (uint64_t)blocks[k].out_len > (uint64_t)data_bytes - blocks[k].out_off) { snprintf(ctx->err, sizeof ctx->err, "mfk_bgzf_crc32: block %lld lies outside the buffer", (long long)k); return 1; }
    HIPCHK(hipSetDevice(ctx->device));
    arena_reset(ctx);
    hipStream_t st = ctx->stream;
    mfk_bgzf_block *d_blocks = aalloc<mfk_bgzf_block>(ctx, (size_t)n_blocks);
    uint32_t *d_crc = aalloc<uint32_t>(ctx, (size_t)n_blocks), *d_bad = aalloc<uint32_t>(ctx, 64);
    if (!d_blocks || !d_crc || !d_bad) return 1;
    HIPCHK(hipMemcpyAsync(d_blocks, blocks, (size_t)n_blocks * sizeof(mfk_bgzf_block), hipMemcpyHostToDevice, st));
    if (int rc = bgzf_crc32_launch(ctx, (const uint8_t *)d_data, d_blocks, (uint32_t)n_blocks, nullptr, d_crc, d_bad, st)) return rc;
    HIPCHK(hipMemcpyAsync(h_crc, d_crc, (size_t)n_blocks * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    ctx->call_ok = true;
    return 0;
}

// blocks: HOST array; d_comp / d_out: DEVICE buffers.  Every block's byte ranges are checked against the buffer sizes here,
// and the decoder never leaves a block's ranges whatever the stream holds.  h_status: NULL or room for n_blocks codes (0 = ok,
// else InflStatus).  Returns 0 when every block inflated to exactly its out_len bytes, 12 when some did not (*n_bad says how
// many), 1 for bad arguments.  The CRC32 of a BGZF block is NOT checked here (the host reader checks it; a caller that needs
// it on this path checks the inflated bytes itself).
extern "C" int mfk_bgzf_inflate(mfk_ctx *ctx, const void *d_comp, int64_t comp_bytes, const mfk_bgzf_block *blocks, int64_t n_blocks,
                                void *d_out, int64_t out_bytes, int32_t *h_status, int64_t *n_bad_out) {
    if (!ctx) return 1;
    ctx->err[0] = 0;
    if (n_bad_out) *n_bad_out = 0;
    if (comp_bytes < 0 || out_bytes < 0 || n_blocks < 0 || n_blocks > 0x7FFFFFFF || (n_blocks > 0 && (!blocks || !d_comp || !d_out))) { snprintf(ctx->err, sizeof ctx->err, "mfk_bgzf_inflate: bad arguments"); return 1; }
    if (n_blocks == 0) return 0;
    for (int64_t k = 0; k < n_blocks; k++) {
        const mfk_bgzf_block &b = blocks[k];
        if (b.comp_off > (uint64_t)comp_bytes || (uint64_t)b.comp_len > (uint64_t)comp_bytes - b.comp_off || b.out_off > (uint64_t)out_bytes || (uint64_t)b.out_len > (uint64_t)out_bytes - b.out_off) {
            snprintf(ctx->err, sizeof ctx->err, "mfk_bgzf_inflate: block %lld lies outside the buffers", (long long)k); return 1;
        }
    }
    HIPCHK(hipSetDevice(ctx->device));
    arena_reset(ctx);
    hipStream_t st = ctx->stream;
    mfk_bgzf_block *d_blocks = aalloc<mfk_bgzf_block>(ctx, (size_t)n_blocks);
    int32_t *d_status = aalloc<int32_t>(ctx, (size_t)n_blocks);
    uint32_t *d_bad = aalloc<uint32_t>(ctx, 64);
    if (!d_blocks || !d_status || !d_bad) return 1;
    HIPCHK(hipMemcpyAsync(d_blocks, blocks, (size_t)n_blocks * sizeof(mfk_bgzf_block), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync(d_bad, 0, 4, st));
    PROF_LAUNCH(k_bgzf_inflate, dim3((unsigned)((n_blocks + INFL_WAVES - 1) / INFL_WAVES)), dim3(64 * INFL_WAVES), 0, st, (const uint8_t *)d_comp, (const mfk_bgzf_block *)d_blocks, (uint32_t)n_blocks, (uint8_t *)d_out, d_status, d_bad);
    KCHK();
    uint32_t bad = 0;
    HIPCHK(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, st));
    if (h_status) HIPCHK(hipMemcpyAsync(h_status, d_status, (size_t)n_blocks * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    ctx->call_ok = true;
    if (n_bad_out) *n_bad_out = bad;
    if (bad) { snprintf(ctx->err, sizeof ctx->err, "mfk_bgzf_inflate: %u of %lld blocks do not inflate to their stated size (corrupt data)", bad, (long long)n_blocks); return 12; }
    return 0;
}
