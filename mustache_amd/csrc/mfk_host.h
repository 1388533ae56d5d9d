// mfk_host.h -- host-side context of libmfk.so: one HIP stream, a growable device arena that is reset at the start
// of every compute call, a pinned staging buffer for the few counters the host reads back, and result storage.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <string>
#include <vector>

#include "mfk_device.h"

struct Chunk { char *p; size_t cap, used; };

void *mfk_pin_alloc(size_t bytes);   // mfk_api.hip: pinned + device-mapped, or heap memory (remembered as such) when pinning fails
void mfk_pin_free(void *p);

// Result tables live in pinned, device-mapped host memory: the gather kernel of a read-back writes them in place (no staging
// copy of megabytes by the host; the tables of C5 are 4 MB per call).  Capacity survives clear(), so a loop over calls
// allocates once.
template <typename T> struct PinnedAlloc {
    using value_type = T;
    PinnedAlloc() = default;
    template <typename U> PinnedAlloc(const PinnedAlloc<U> &) {}
    // pinned memory is scarcer than the heap: when hipHostMalloc refuses, the table lives in pageable memory (ReadBatch::flush
    // stages read-backs into pageable destinations anyway), so no exception leaves an extern "C" entry point for that reason
    T *allocate(size_t n) { return static_cast<T *>(mfk_pin_alloc(n * sizeof(T))); }
    void deallocate(T *p, size_t) { mfk_pin_free(p); }
    template <typename U> bool operator==(const PinnedAlloc<U> &) const { return true; }
    template <typename U> bool operator!=(const PinnedAlloc<U> &) const { return false; }
};
template <typename T> using PinVec = std::vector<T, PinnedAlloc<T>>;
using PinString = std::basic_string<char, std::char_traits<char>, PinnedAlloc<char>>;

struct mfk_ctx {
    int device;
    hipStream_t stream;
    hipStream_t stream2;            // side stream for kernel chains that do not depend on each other (consensus stage)
    hipEvent_t ev0, ev1;
    hipStream_t stream3;            // second side stream (pairflanks: spanning reads next to the direct repeats)
    hipEvent_t ev_fork, ev_join, ev_join3, ev_fork3, ev_fill;
    hipStream_t fill_stream;   // while set, FillGroup launches go to this stream instead of `stream` (fills that overlap the kernels before their first consumer)
    unsigned long long *scan_state; uint32_t scan_epoch;   // tile aggregates of k_scan_chain, tagged with the call's epoch (never cleared between calls)
    uint32_t *scan_err;             // device word behind the aggregates: set when a workgroup of k_scan_chain gave up waiting for a predecessor
    bool span_on;                   // the two events that bracket a call's device work are recorded (mfk_last_device_ms); mfk_span_timing turns them off
    bool call_ok;                   // the previous compute call ran to its end: its side streams are joined (else arena_reset drains them)
    double last_ms;
    std::vector<Chunk> chunks;      // device arena
    void *pinned; size_t pinned_cap;
    char *hdr_dev; size_t hdr_cap;   // contig header on the device (own allocation, survives arena resets) ...
    std::vector<char> hdr_last;      // ... and the bytes it holds: an unchanged header is not uploaded again
    uint32_t *gather_count;         // device word: blocks of the current gather that have finished (the last one resets it)
    uint32_t *gather_flag, *gather_flag_dev;   // pinned word of its own (host / device address) the sequence number lands in
    uint32_t gather_seq;            // sequence number the gather kernel publishes behind its data; the host polls for it
    char err[1024];
    // optional per-kernel timing (mfk_profile_enable): HIP events around every launch on the ctx stream
    bool prof_on;
    std::vector<std::string> prof_filter;      // empty = every kernel
    bool prof_skip;                            // the launch between prof_begin / prof_end is not bracketed
    std::vector<hipEvent_t> prof_events;       // pool, two per launch
    std::vector<const char *> prof_names;      // launches of the current call
    size_t prof_used;
    std::vector<std::string> prof_kernel;      // accumulated totals since the last reset
    std::vector<double> prof_ms;
    std::vector<int64_t> prof_calls;
    // findflanks results (host)
    PinVec<mfk_site> sites;
    PinVec<mfk_flank> flanks;
    PinString seqs;
    int64_t stats[MFK_N_STATS];
    // pairflanks results (host)
    PinVec<mfk_pair> pairs;
    PinString strs;
    int64_t pstats[MFK_N_STATS];
    std::vector<int32_t> isq_f, isq_r;   // inferseq pairing result (mfk_inferseq_pairs)
    bool sj_attr;                   // k_job_small may use more than 64 KB of dynamic LDS on this device (attribute set)
    uint32_t sj_static;             // its static LDS in bytes (the first pass sizes its dynamic part for four workgroups per CU)
    char *na_pinned;                // pinned staging of mfk_recall_noalloc (header | sites | counts), allocated by mfk_create
    std::vector<int32_t> ir_ov;     // external inverted-repeat table for the next pairflanks call: rows i5 i3 has a1s a1e a2s a2e, sorted
    // multi-GPU merge (mfk_comm.hip): RCCL communicator of this ctx and the result of the last mfk_allgather_sites
    void *dec = nullptr;            // device decode (mfk_decode.hip): the state between mfk_decode_bam and mfk_decode_bam_fetch
    void *comm; int comm_rank, comm_world;
    std::vector<uint8_t> gat; size_t gat_msg, gat_hint; std::vector<int64_t> gat_counts; int32_t gat_row_bytes;   // gat_hint: largest table of the last gather
};

#define HIPCHK(call)                                                                                         \
    do {                                                                                                     \
        hipError_t e_ = (call);                                                                              \
        if (e_ != hipSuccess) {                                                                              \
            snprintf(ctx->err, sizeof ctx->err, "%s failed at %s:%d: %s", #call, __FILE__, __LINE__,        \
                     hipGetErrorString(e_));                                                                 \
            return 100;                                                                                      \
        }                                                                                                    \
    } while (0)

#define KCHK() HIPCHK(hipGetLastError())
// first / last event of a call's device work: two API calls (4 us each here) that a caller in a tight loop can do without
#define SPAN_EVENT(ev) do { if (ctx->span_on) HIPCHK(hipEventRecord(ctx->ev, st)); else ctx->last_ms = 0.0; } while (0)

void prof_begin(mfk_ctx *ctx, const char *name, hipStream_t stream = nullptr);   // events go to `stream` (default: the ctx stream)
void prof_end(mfk_ctx *ctx, hipStream_t stream = nullptr);
void prof_collect(mfk_ctx *ctx);               // call after the stream has been synchronised
#define PROF_LAUNCH(kern, grid, block, lds, stream, ...)                      \
    do {                                                                      \
        prof_begin(ctx, #kern, stream);                                       \
        hipLaunchKernelGGL(kern, grid, block, lds, stream, __VA_ARGS__);      \
        prof_end(ctx, stream);                                                \
    } while (0)

// host-side wall clock of the sections of a compute call, printed to stderr when MFK_HOST_TIMING is set (diagnostic)
#include <chrono>
struct HostTimer {
    const char *what; bool on; std::chrono::steady_clock::time_point t0, last;
    explicit HostTimer(const char *w) : what(w) { static const bool e = getenv("MFK_HOST_TIMING") != nullptr; on = e; if (on) t0 = last = std::chrono::steady_clock::now(); }
    void lap(const char *name) {
        if (!on) return;
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[mfk host] %s: %-28s %8.1f us (at %8.1f)\n", what, name, std::chrono::duration<double, std::micro>(t - last).count(), std::chrono::duration<double, std::micro>(t - t0).count());
        last = t;
    }
};

// arena ----------------------------------------------------------------------------------------------------------
void arena_reset(mfk_ctx *ctx);                       // frees nothing on the fast path; merges chunks after growth
void *arena_alloc(mfk_ctx *ctx, size_t bytes);        // 256-byte aligned device memory valid until the next reset
template <typename T> static inline T *aalloc(mfk_ctx *ctx, size_t n) { return (T *)arena_alloc(ctx, (n ? n : 1) * sizeof(T)); }
int upload_pinned(mfk_ctx *ctx, void *dev, const void *pinned_host, size_t bytes);   // small host -> device copy of pinned_buf bytes by a kernel
void *pinned_buf(mfk_ctx *ctx, size_t bytes);      // grows (after draining the stream) and returns the staging buffer

// device -> host read-backs: one stream synchronisation per batch
int read_back(mfk_ctx *ctx, const void *dev, void *host, size_t bytes);
struct ReadBatch {
    struct Item { const void *dev; void *host; size_t bytes, off; };
    mfk_ctx *ctx; std::vector<Item> items; size_t total;
    explicit ReadBatch(mfk_ctx *c) : ctx(c), total(0) {}
    void add(const void *dev, void *host, size_t bytes) { if (bytes) { items.push_back({dev, host, bytes, total}); total += (bytes + 15) & ~(size_t)15; } }
    int flush();
};

// several arrays that start out filled with one byte value (0x00 or 0xFF): one arena block, one fill
struct FillGroup {
    struct Req { void **out; size_t bytes, off; };
    mfk_ctx *ctx; int value; std::vector<Req> reqs; size_t total;
    FillGroup(mfk_ctx *c, int v) : ctx(c), value(v), total(0) {}
    template <typename T> void want(T **out, size_t n) { size_t b = ((n ? n : 1) * sizeof(T) + 255) & ~(size_t)255; reqs.push_back({(void **)out, b, total}); total += b; }
    int commit();    // allocates, assigns the pointers, queues the fill; non-zero on failure
};
int fill_commit2(FillGroup &zeros, FillGroup &ones);   // both groups (values 0 and 0xFF) with one allocation and one launch

size_t upload_reads_pinned_bytes(int n_contigs);
int upload_reads(mfk_ctx *ctx, const mfk_reads *reads, const mfk_params *params, DevReads *R, DevParams *P);
int prepare_reads(mfk_ctx *ctx, const mfk_reads *reads, const mfk_params *params, DevReads *R, DevParams *P, char *h, size_t *used);   // validation, R / P, header bytes into h
void set_header_ptrs(DevReads *R, char *d_hdr);
int check_scan_error(mfk_ctx *ctx, uint32_t h_scan_err);   // after the read-back that carried scan_err: non-zero fails the call
int noalloc_setup(mfk_ctx *ctx);     // mfk_findflanks.hip: device address of the no-allocation scratch array + pinned staging, once per ctx (mfk_create)
void noalloc_teardown(mfk_ctx *ctx);

static inline unsigned nblocks(int64_t n, int threads) { int64_t b = (n + threads - 1) / threads; return (unsigned)(b < 1 ? 1 : b); }

// device-wide primitives (rocPRIM) on the ctx stream, temp storage from the arena
void mfk_decode_release(mfk_ctx *ctx);   // mfk_decode.hip: frees ctx->dec
int bgzf_crc32_launch(mfk_ctx *ctx, const uint8_t *d_data, const mfk_bgzf_block *d_blocks, uint32_t n, const uint32_t *d_expect, uint32_t *d_crc_out, uint32_t *d_bad, hipStream_t st);   // mfk_inflate.hip
int dev_sort_keys64(mfk_ctx *ctx, uint64_t *keys_in, uint64_t *keys_out, size_t n, int begin_bit, int end_bit);
int dev_sort_pairs64(mfk_ctx *ctx, uint64_t *keys_in, uint64_t *keys_out, uint32_t *vals_in, uint32_t *vals_out, size_t n, int begin_bit, int end_bit);
int dev_excl_scan32(mfk_ctx *ctx, const uint32_t *in, uint32_t *out, size_t n);   // out has n+1 entries: out[n] = total
int dev_excl_scan64(mfk_ctx *ctx, const uint32_t *in, uint64_t *out, size_t n);   // 64-bit sums, out[n] = total

// mfk_flankpairs with thousands of sites: findflanks leaves its rows on the device (k_rows_fill) and the pairing stage reads
// them there; what findflanks would have read back (row totals, counters, error words, the three tables) travels with the
// pairing stage's own read-backs instead of draining the queue twice in between
struct FlankHandoff {
    bool active;                       // set by findflanks_impl when it handed over instead of reading back
    uint32_t n_act;                    // at most n_act site rows and 2 n_act flank rows
    const mfk_site *d_sites; const mfk_flank *d_flanks; const char *d_seqs;
    const uint64_t *d_rows_total;      // site rows | flank rows << 24
    const uint64_t *d_bytes_total;     // consensus characters
    const uint32_t *d_stats;           // MFK_N_STATS counters of the findflanks kernels
    const uint32_t *d_cons_err;        // error word of the consensus stage (may be null); 4 words: err, max_len, max_R, n_big
    bool sure_small;                   // n_big must come back 0 (ConsensusIO::sure_small)
};
int findflanks_impl(mfk_ctx *ctx, const mfk_reads *reads, const mfk_params *params, int stop_after, int64_t *n_sites_out, int64_t *n_flanks_out, int64_t *seq_bytes_out, FlankHandoff *fh);
int pairflanks_impl(mfk_ctx *ctx, const mfk_reads *reads, const mfk_params *params, const mfk_flank *flanks, const char *flank_seqs, int64_t n_flanks,
                    const int32_t *h_contig_rank, const uint8_t *genome, const int64_t *h_genome_off, int64_t *n_pairs_out, int64_t *str_bytes_out, FlankHandoff *fh);

// consensus stage (mfk_consensus.hip): fills, per active site k and side s, cons_count / cons_len / n_pass_len /
// n_pass_both and the consensus characters (ASCII) at cons_chars + job * cons_stride
struct ConsensusIO {
    // events, sorted by (gpos, side, read)
    const uint64_t *ev_keys; const uint8_t *ev_info; uint32_t n_events;
    // active sites
    uint32_t n_act; const uint32_t *act_site; const uint32_t *act_st;     // st bits: 1 keep5, 2 keep3, 4 alive
    const uint32_t *act_cnt2;                                              // distinct query names per active site and side
    const uint32_t *site_ev_begin, *site_ev_mid, *site_ev_end;             // indexed by site
    // outputs, indexed by job = 2*k + side (side 0 = 5p, 1 = 3p)
    int32_t *cons_count, *cons_len; uint32_t *n_pass_len, *n_pass_both;
    char *cons_chars; uint32_t cons_stride;
    const uint32_t *err_dev; uint32_t err_flags;   // device error word of the consensus stage, read back with the results (4 words: err, max_len, max_R, n_big)
    uint32_t err_words[4]; bool sure_small;        // sure_small: k_job_small was queued without reading its outcome back; n_big must come back 0
    unsigned long long *work;                      // device counter: columns x limbs of the edit-distance recurrences run (stats[12])
};
int run_consensus(mfk_ctx *ctx, const DevReads &R, const DevParams &P, ConsensusIO &io);
