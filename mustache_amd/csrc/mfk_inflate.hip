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
