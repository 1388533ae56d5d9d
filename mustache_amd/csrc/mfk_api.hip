// mfk_api.hip -- lifecycle, device arena, input upload and rocPRIM wrappers of libmfk.so
#include "mfk_host.h"

#include <rocprim/rocprim.hpp>

extern "C" {

#define SCAN_CHAIN_TILES 256   // k_scan_chain: at most this many 8192-element tiles (2 M elements), rocPRIM beyond
mfk_ctx *mfk_create(int device_id) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device_id < 0 || device_id >= ndev) return nullptr;
    if (hipSetDevice(device_id) != hipSuccess) return nullptr;
    mfk_ctx *ctx = new mfk_ctx();
    ctx->device = device_id;
    ctx->last_ms = 0.0;
    ctx->pinned = nullptr;
    ctx->pinned_cap = 0;
    ctx->gather_count = nullptr;
    ctx->hdr_dev = nullptr; ctx->hdr_cap = 0;
    ctx->gather_flag = nullptr; ctx->gather_flag_dev = nullptr;
    ctx->gather_seq = 0;
    ctx->scan_state = nullptr; ctx->scan_err = nullptr; ctx->call_ok = true; ctx->span_on = true;
    { const char *e0 = getenv("MFK_SCAN_EPOCH0"); ctx->scan_epoch = e0 ? (uint32_t)atoi(e0) & 0xFFFFu : 0u; }   // test hook: start near the wrap
    ctx->fill_stream = nullptr;
    ctx->na_pinned = nullptr;
    ctx->sj_attr = false; ctx->sj_static = 0;
    ctx->comm = nullptr; ctx->comm_rank = 0; ctx->comm_world = 0; ctx->gat_msg = 0; ctx->gat_hint = 0; ctx->gat_row_bytes = 0;
    ctx->err[0] = 0;
    ctx->prof_on = false;
    ctx->prof_skip = false;
    ctx->prof_used = 0;
    memset(ctx->stats, 0, sizeof ctx->stats);
    memset(ctx->pstats, 0, sizeof ctx->pstats);
    if (hipMalloc((void **)&ctx->gather_count, 256) != hipSuccess || hipMemset(ctx->gather_count, 0, 256) != hipSuccess) { delete ctx; return nullptr; }
    if (hipMalloc((void **)&ctx->scan_state, SCAN_CHAIN_TILES * 8 + 256) != hipSuccess || hipMemset(ctx->scan_state, 0, SCAN_CHAIN_TILES * 8 + 256) != hipSuccess) { delete ctx; return nullptr; }
    ctx->scan_err = reinterpret_cast<uint32_t *>(ctx->scan_state + SCAN_CHAIN_TILES);
    if (hipHostMalloc((void **)&ctx->gather_flag, 256, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer((void **)&ctx->gather_flag_dev, ctx->gather_flag, 0) != hipSuccess) { delete ctx; return nullptr; }
    ctx->gather_flag[0] = 0;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return nullptr; }
    if (hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking) != hipSuccess) { (void)hipStreamDestroy(ctx->stream); delete ctx; return nullptr; }
    if (hipStreamCreateWithFlags(&ctx->stream3, hipStreamNonBlocking) != hipSuccess) { (void)hipStreamDestroy(ctx->stream2); (void)hipStreamDestroy(ctx->stream); delete ctx; return nullptr; }
    if (hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_join3, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_fork3, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_fill, hipEventDisableTiming) != hipSuccess) { delete ctx; return nullptr; }
    if (noalloc_setup(ctx) != 0) { mfk_destroy(ctx); return nullptr; }
    return ctx;
}

void mfk_destroy(mfk_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipStreamSynchronize(ctx->stream2);
    (void)hipStreamSynchronize(ctx->stream3);
    (void)mfk_comm_free(ctx);
    mfk_decode_release(ctx);
    for (auto &c : ctx->chunks) (void)hipFree(c.p);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->gather_count) (void)hipFree(ctx->gather_count);
    if (ctx->hdr_dev) (void)hipFree(ctx->hdr_dev);
    if (ctx->scan_state) (void)hipFree(ctx->scan_state);
    if (ctx->gather_flag) (void)hipHostFree(ctx->gather_flag);
    noalloc_teardown(ctx);
    for (auto e : ctx->prof_events) (void)hipEventDestroy(e);
    (void)hipEventDestroy(ctx->ev0);
    (void)hipEventDestroy(ctx->ev1);
    (void)hipEventDestroy(ctx->ev_fork);
    (void)hipEventDestroy(ctx->ev_join);
    (void)hipEventDestroy(ctx->ev_join3);
    (void)hipEventDestroy(ctx->ev_fork3);
    (void)hipEventDestroy(ctx->ev_fill);
    (void)hipStreamDestroy(ctx->stream3);
    (void)hipStreamDestroy(ctx->stream2);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char *mfk_last_error(mfk_ctx *ctx) { return ctx ? ctx->err : "null ctx"; }
int mfk_version(void) { return 1; }
double mfk_last_device_ms(mfk_ctx *ctx) { return ctx ? ctx->last_ms : 0.0; }
int mfk_span_timing(mfk_ctx *ctx, int on) { if (!ctx) return 1; ctx->span_on = on != 0; if (!on) ctx->last_ms = 0.0; return 0; }

int mfk_profile_enable(mfk_ctx *ctx, int on) {
    if (!ctx) return 1;
    ctx->prof_on = on != 0;
    ctx->prof_kernel.clear(); ctx->prof_ms.clear(); ctx->prof_calls.clear();
    return 0;
}

int mfk_profile_select(mfk_ctx *ctx, const char *names) {
    if (!ctx) return 1;
    ctx->prof_filter.clear();
    ctx->prof_kernel.clear(); ctx->prof_ms.clear(); ctx->prof_calls.clear();
    if (!names) return 0;
    const char *p = names;
    while (*p) {
        const char *e = strchr(p, '\n');
        size_t len = e ? (size_t)(e - p) : strlen(p);
        if (len) ctx->prof_filter.emplace_back(p, len);
        p += len + (e ? 1 : 0);
    }
    return 0;
}

int mfk_profile_fetch(mfk_ctx *ctx, char *names, int64_t names_cap, double *ms, int64_t *calls, int64_t cap, int64_t *n) {
    if (!ctx || !n) return 1;
    int64_t k = 0, o = 0;
    for (size_t i = 0; i < ctx->prof_kernel.size() && k < cap; i++, k++) {
        const std::string &nm = ctx->prof_kernel[i];
        if (names && o + (int64_t)nm.size() + 1 < names_cap) { memcpy(names + o, nm.c_str(), nm.size()); o += (int64_t)nm.size(); names[o++] = '\n'; }
        if (ms) ms[k] = ctx->prof_ms[i];
        if (calls) calls[k] = ctx->prof_calls[i];
    }
    if (names && names_cap > 0) names[o < names_cap ? o : names_cap - 1] = 0;
    *n = k;
    return 0;
}

}  // extern "C"

// ---- result-table memory (PinnedAlloc) -------------------------------------------------------------------------------
#include <mutex>
#include <unordered_set>
static std::mutex g_pin_mu;
static std::unordered_set<void *> g_pin_heap;   // blocks that came from malloc because hipHostMalloc refused
void *mfk_pin_alloc(size_t bytes) {
    void *p = nullptr;
    static const bool no_pin = getenv("MFK_NO_PINNED_TABLES") != nullptr;   // test hook: every table in pageable memory
    if (!no_pin && hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocMapped) == hipSuccess && p) return p;
    (void)hipGetLastError();
    p = malloc(bytes ? bytes : 1);
    if (!p) throw std::bad_alloc();   // the heap itself is exhausted: as for any std::vector in the process
    std::lock_guard<std::mutex> g(g_pin_mu);
    g_pin_heap.insert(p);
    return p;
}
void mfk_pin_free(void *p) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> g(g_pin_mu);
        auto it = g_pin_heap.find(p);
        if (it != g_pin_heap.end()) { g_pin_heap.erase(it); free(p); return; }
    }
    (void)hipHostFree(p);
}

void prof_begin(mfk_ctx *ctx, const char *name, hipStream_t stream) {
    if (!ctx->prof_on) return;
    ctx->prof_skip = false;
    if (!ctx->prof_filter.empty()) {
        bool hit = false;
        for (auto &f : ctx->prof_filter) if (f == name) { hit = true; break; }
        if (!hit) { ctx->prof_skip = true; return; }
    }
    if (ctx->prof_used + 2 > ctx->prof_events.size()) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { ctx->prof_on = false; return; }
        ctx->prof_events.push_back(a); ctx->prof_events.push_back(b);
    }
    ctx->prof_names.push_back(name);
    (void)hipEventRecord(ctx->prof_events[ctx->prof_used], stream ? stream : ctx->stream);
}

void prof_end(mfk_ctx *ctx, hipStream_t stream) {
    if (ctx->prof_skip) { ctx->prof_skip = false; return; }
    if (!ctx->prof_on || ctx->prof_names.size() * 2 != ctx->prof_used + 2) return;
    (void)hipEventRecord(ctx->prof_events[ctx->prof_used + 1], stream ? stream : ctx->stream);
    ctx->prof_used += 2;
}

void prof_collect(mfk_ctx *ctx) {
    if (ctx->prof_on) {
        (void)hipStreamSynchronize(ctx->stream);
        for (size_t i = 0; i * 2 + 1 < ctx->prof_used && i < ctx->prof_names.size(); i++) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, ctx->prof_events[2 * i], ctx->prof_events[2 * i + 1]) != hipSuccess) continue;
            size_t k = 0;
            for (; k < ctx->prof_kernel.size(); k++) if (ctx->prof_kernel[k] == ctx->prof_names[i]) break;
            if (k == ctx->prof_kernel.size()) { ctx->prof_kernel.push_back(ctx->prof_names[i]); ctx->prof_ms.push_back(0.0); ctx->prof_calls.push_back(0); }
            ctx->prof_ms[k] += ms; ctx->prof_calls[k] += 1;
        }
    }
    ctx->prof_names.clear();
    ctx->prof_used = 0;
}

// ---- arena ---------------------------------------------------------------------------------------------------
void arena_reset(mfk_ctx *ctx) {
    ctx->prof_names.clear();   // a call that failed half-way leaves its launches behind
    ctx->prof_used = 0;
    if (!ctx->call_ok) {
        // the previous call returned early (an error, or an entry point that does not mark its end): kernels or fills it had
        // queued on the side streams may still be running and would write into the arena blocks handed out again below
        (void)hipStreamSynchronize(ctx->stream2);
        (void)hipStreamSynchronize(ctx->stream3);
        (void)hipStreamSynchronize(ctx->stream);
    }
    ctx->call_ok = false;
    if (ctx->chunks.size() > 1) {
        // the previous call had to grow: replace the pieces by one chunk of the total size
        (void)hipStreamSynchronize(ctx->stream);
        size_t total = 0;
        for (auto &c : ctx->chunks) { total += c.cap; (void)hipFree(c.p); }
        ctx->chunks.clear();
        char *p = nullptr;
        if (hipMalloc((void **)&p, total) == hipSuccess) ctx->chunks.push_back({p, total, 0});
    }
    for (auto &c : ctx->chunks) c.used = 0;
}

void *arena_alloc(mfk_ctx *ctx, size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    if (!ctx->chunks.empty()) {
        Chunk &c = ctx->chunks.back();
        if (c.used + bytes <= c.cap) { void *p = c.p + c.used; c.used += bytes; return p; }
    }
    size_t cap = (size_t)64 << 20;
    if (!ctx->chunks.empty() && ctx->chunks.back().cap * 2 > cap) cap = ctx->chunks.back().cap * 2;
    if (bytes > cap) cap = bytes;
    char *p = nullptr;
    if (hipMalloc((void **)&p, cap) != hipSuccess) {
        snprintf(ctx->err, sizeof ctx->err, "hipMalloc of %zu bytes failed", cap);
        return nullptr;
    }
    ctx->chunks.push_back({p, cap, bytes});
    return p;
}

int read_back(mfk_ctx *ctx, const void *dev, void *host, size_t bytes) {
    ReadBatch b(ctx);
    b.add(dev, host, bytes);
    return b.flush();
}

// up to GATHER_MAX pieces of device memory -> one staging buffer in pinned host memory, written by the kernel itself
// (zero-copy): a read-back of many small result arrays is one launch instead of one copy per array.  The kernel then
// publishes a sequence number behind its data (every thread fences its stores system-wide, the last block to finish
// stores the number with release semantics) and the host polls that word instead of sleeping in a stream
// synchronisation: 6.8 us instead of 11 us per read-back on this box (scripts/micro/sync_cost.hip).
#define GATHER_MAX 16
#define DIRECT_MIN 65536   // read-back items from this size on are tried for an in-place write (one address lookup each)
struct GatherArgs { const uint8_t *src[GATHER_MAX]; uint8_t *dst[GATHER_MAX]; uint32_t bytes[GATHER_MAX]; int n; };   // dst: device address of pinned host memory
__global__ __launch_bounds__(256) void k_gather(GatherArgs g, uint32_t *count, uint32_t *flag, uint32_t seq) {
    for (int i = blockIdx.y; i < g.n; i += gridDim.y) {
        const uint8_t *src = g.src[i];
        uint8_t *out = g.dst[i];
        const uint32_t nb = g.bytes[i];
        if ((((uintptr_t)src | (uintptr_t)out) & 15) == 0) {   // 16 bytes per lane: stores to host memory are few and wide
            const uint32_t nq = nb >> 4;
            for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < nq; k += gridDim.x * blockDim.x)
                reinterpret_cast<uint4 *>(out)[k] = reinterpret_cast<const uint4 *>(src)[k];
            for (uint32_t k = (nq << 4) + blockIdx.x * blockDim.x + threadIdx.x; k < nb; k += gridDim.x * blockDim.x) out[k] = src[k];
        } else if ((((uintptr_t)src | (uintptr_t)out) & 3) == 0) {
            const uint32_t nw = nb >> 2;
            for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < nw; k += gridDim.x * blockDim.x)
                reinterpret_cast<uint32_t *>(out)[k] = reinterpret_cast<const uint32_t *>(src)[k];
            for (uint32_t k = (nw << 2) + blockIdx.x * blockDim.x + threadIdx.x; k < nb; k += gridDim.x * blockDim.x) out[k] = src[k];
        } else
            for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < nb; k += gridDim.x * blockDim.x) out[k] = src[k];
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t n_blocks = gridDim.x * gridDim.y;
        if (atomicAdd(count, 1u) == n_blocks - 1) {
            atomicExch(count, 0u);   // ready for the next gather on this stream
            __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// host -> device for the small staging blocks (contig header, flank table): a kernel that reads the pinned bytes through
// their device address; launching it costs the host less than a hipMemcpyAsync and the stream no copy-engine hand-over
__global__ __launch_bounds__(256) void k_upload(const uint4 *src, uint4 *dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
int upload_pinned(mfk_ctx *ctx, void *dev, const void *pinned_host, size_t bytes) {
    void *dsrc = nullptr;
    if ((bytes & 15) == 0 && (((uintptr_t)dev | (uintptr_t)pinned_host) & 15) == 0 && bytes <= ((size_t)1 << 22) &&
        hipHostGetDevicePointer(&dsrc, const_cast<void *>(pinned_host), 0) == hipSuccess) {
        const size_t n16 = bytes / 16;
        size_t blocks = (n16 + 255) / 256; if (blocks < 1) blocks = 1; if (blocks > 256) blocks = 256;
        hipLaunchKernelGGL(k_upload, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, (const uint4 *)dsrc, (uint4 *)dev, n16);
        KCHK();
        return 0;
    }
    HIPCHK(hipMemcpyAsync(dev, pinned_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    return 0;
}

int check_scan_error(mfk_ctx *ctx, uint32_t h_scan_err) {
    if (!h_scan_err) return 0;
    (void)hipMemsetAsync(ctx->scan_err, 0, 4, ctx->stream);
    snprintf(ctx->err, sizeof ctx->err, "a prefix scan gave up waiting for an earlier tile (k_scan_chain): results of this call are not valid; set MFK_NO_SCAN_CHAIN=1 to use the two-launch scan");
    return 10;
}

int ReadBatch::flush() {
    if (items.empty()) return 0;
    // A large item whose destination is pinned, device-mapped memory (the result tables, PinVec / PinString) is written in place;
    // everything else goes through the staging buffer and is copied out by the host.
    std::vector<void *> direct(items.size(), nullptr);
    size_t staged = 0;
    for (size_t i = 0; i < items.size(); i++) {
        if (items[i].bytes >= DIRECT_MIN) {
            void *d = nullptr;
            if (hipHostGetDevicePointer(&d, items[i].host, 0) == hipSuccess && d) direct[i] = d;
            else (void)hipGetLastError();   // pageable destination: not an error of the call
        }
        if (!direct[i]) { items[i].off = staged; staged += (items[i].bytes + 15) & ~(size_t)15; }
    }
    char *pin = (char *)pinned_buf(ctx, staged ? staged : 16);
    if (!pin) { snprintf(ctx->err, sizeof ctx->err, "pinned allocation failed"); return 1; }
    bool small = items.size() <= GATHER_MAX && total < ((size_t)1 << 24);
    void *dpin = nullptr;
    if (small && hipHostGetDevicePointer(&dpin, pin, 0) != hipSuccess) small = false;
    if (small) {
        GatherArgs g; g.n = (int)items.size();
        size_t biggest = 0;
        for (int i = 0; i < g.n; i++) {
            g.src[i] = (const uint8_t *)items[i].dev; g.bytes[i] = (uint32_t)items[i].bytes;
            g.dst[i] = direct[i] ? (uint8_t *)direct[i] : (uint8_t *)dpin + items[i].off;
            if (items[i].bytes > biggest) biggest = items[i].bytes;
        }
        unsigned bx = (unsigned)((biggest / 16 + 255) / 256); if (bx < 1) bx = 1; if (bx > 64) bx = 64;
        uint32_t seq = ++ctx->gather_seq;
        if (seq == 0) seq = ++ctx->gather_seq;
        volatile uint32_t *hflag = ctx->gather_flag;   // sequence numbers only grow: no reset needed between gathers
        hipLaunchKernelGGL(k_gather, dim3(bx, g.n), dim3(256), 0, ctx->stream, g, ctx->gather_count, ctx->gather_flag_dev, seq);
        KCHK();
        // poll; if the number does not arrive (a fault in an earlier kernel), the stream synchronisation reports why
        bool seen = false;
        for (uint64_t spin = 0; spin < ((uint64_t)1 << 26); spin++) {
            if (__atomic_load_n((const uint32_t *)hflag, __ATOMIC_ACQUIRE) == seq) { seen = true; break; }
            if ((spin & 0xFFFFF) == 0xFFFFF && hipStreamQuery(ctx->stream) != hipErrorNotReady) break;
        }
        if (!seen) HIPCHK(hipStreamSynchronize(ctx->stream));
    } else {
        for (size_t i = 0; i < items.size(); i++)
            HIPCHK(hipMemcpyAsync(direct[i] ? items[i].host : (void *)(pin + items[i].off), items[i].dev, items[i].bytes, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    for (size_t i = 0; i < items.size(); i++) if (!direct[i]) memcpy(items[i].host, pin + items[i].off, items[i].bytes);
    items.clear(); total = 0;
    return 0;
}

// fills of a phase: [0, n_zero) 16-byte pieces of zeros, then ones up to n_total -- one launch for both groups (a
// hipMemsetAsync each costs a launch of its own plus more host time than a kernel launch)
__global__ __launch_bounds__(256) void k_fill2(uint4 *p, size_t n_zero, size_t n_total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_total; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t v = i < n_zero ? 0u : ~0u;   // (a select between two uint4 constants made the compiler keep both in scratch memory and load from there)
        p[i] = make_uint4(v, v, v, v);
    }
}

static int fill_launch(mfk_ctx *ctx, char *p, size_t zero_bytes, size_t one_bytes) {
    const size_t n_zero = zero_bytes / 16, n_total = (zero_bytes + one_bytes) / 16;   // group sizes are multiples of 256
    if (!n_total) return 0;
    size_t blocks = (n_total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_fill2, dim3((unsigned)blocks), dim3(256), 0, ctx->fill_stream ? ctx->fill_stream : ctx->stream, (uint4 *)p, n_zero, n_total);
    KCHK();
    return 0;
}

int FillGroup::commit() {
    if (reqs.empty()) return 0;
    char *p = (char *)arena_alloc(ctx, total);
    if (!p) return 1;
    for (auto &r : reqs) *r.out = p + r.off;
    if (value == 0) return fill_launch(ctx, p, total, 0);
    if (value == 0xFF) return fill_launch(ctx, p, 0, total);
    HIPCHK(hipMemsetAsync(p, value, total, ctx->fill_stream ? ctx->fill_stream : ctx->stream));
    return 0;
}

// a zero group and a 0xFF group in one arena block and one launch
int fill_commit2(FillGroup &z, FillGroup &f) {
    mfk_ctx *ctx = z.ctx;
    if (z.reqs.empty()) return f.commit();
    if (f.reqs.empty()) return z.commit();
    char *p = (char *)arena_alloc(ctx, z.total + f.total);
    if (!p) return 1;
    for (auto &r : z.reqs) *r.out = p + r.off;
    for (auto &r : f.reqs) *r.out = p + z.total + r.off;
    return fill_launch(ctx, p, z.total, f.total);
}

void *pinned_buf(mfk_ctx *ctx, size_t bytes) {
    if (bytes > ctx->pinned_cap) {
        (void)hipStreamSynchronize(ctx->stream);   // copies from / into the old buffer may still be queued
        if (ctx->pinned) (void)hipHostFree(ctx->pinned);
        ctx->pinned = nullptr;
        size_t cap = bytes < 4096 ? 4096 : bytes * 2;
        if (hipHostMalloc(&ctx->pinned, cap, hipHostMallocMapped) != hipSuccess) { ctx->pinned_cap = 0; return nullptr; }
        ctx->pinned_cap = cap;
    }
    return ctx->pinned;
}

// ---- inputs ----------------------------------------------------------------------------------------------------
// The flat coordinate "gpos": contig c occupies [base[c], base[c] + len[c] + 2) (site positions run from -1 to len),
// contigs are separated by a gap wider than any neighbourhood the filters look at, so one sorted u32 axis serves all
// contigs and no neighbour search can cross a contig boundary.
size_t upload_reads_pinned_bytes(int nc) { return (((size_t)(nc + 1) * 8 + (size_t)nc * 8) + 255) & ~(size_t)255; }

// validation + everything of R / P that is not a device address of the header; the header bytes [read starts i64 x (nc+1) |
// lengths i32 x nc | bases u32 x nc] go to `h` (upload_reads_pinned_bytes(nc) bytes); *used = bytes of it that carry data
int prepare_reads(mfk_ctx *ctx, const mfk_reads *in, const mfk_params *pr, DevReads *R, DevParams *P, char *h, size_t *used) {
    if (in->n_reads < 0 || in->n_reads >= ((int64_t)1 << 31)) { snprintf(ctx->err, sizeof ctx->err, "n_reads %lld out of range (max 2^31-1 per call)", (long long)in->n_reads); return 1; }
    if (in->n_contigs <= 0 || in->n_contigs > 65535) { snprintf(ctx->err, sizeof ctx->err, "n_contigs %d out of range", in->n_contigs); return 1; }
    if (in->n_reads > 0 && (!in->ref_start || !in->ref_end || !in->link || !in->contig_id || !in->qlen || !in->first_op_len || !in->last_op_len || !in->mapq || !in->flags)) {
        snprintf(ctx->err, sizeof ctx->err, "null column pointer in mfk_reads"); return 1;
    }
    if (((uintptr_t)in->flags & 15) != 0) { snprintf(ctx->err, sizeof ctx->err, "flags column must be 16-byte aligned"); return 1; }
    if (in->h_contig_read_start[0] != 0 || in->h_contig_read_start[in->n_contigs] != in->n_reads) { snprintf(ctx->err, sizeof ctx->err, "contig_read_start does not cover the reads"); return 1; }
    R->n = in->n_reads;
    R->ref_start = in->ref_start; R->ref_end = in->ref_end; R->link = in->link;
    R->contig_id = in->contig_id; R->qlen = in->qlen; R->first_op_len = in->first_op_len; R->last_op_len = in->last_op_len;
    R->mapq = in->mapq; R->flags = in->flags;
    R->n_contigs = in->n_contigs; R->max_span = in->max_span < 1 ? 1 : in->max_span;
    R->n_cx = in->n_cx; R->cx_read = in->cx_read; R->cx_off = in->cx_off;
    R->blk_start = in->blk_start; R->blk_end = in->blk_end; R->blk_ins = in->blk_ins;
    R->n_pool = in->n_pool; R->pool_read = in->pool_read; R->pool_off = in->pool_off;
    R->pool_bases = in->pool_bases; R->pool_quals = in->pool_quals;
    R->n_clip = 0; R->clip_rec = nullptr; R->clip_tile = nullptr;
    if (in->clip_rec) {
        if (!in->clip_tile || in->n_clip < 0 || in->n_clip > in->n_reads) { snprintf(ctx->err, sizeof ctx->err, "clip_rec needs clip_tile and 0 <= n_clip <= n_reads (got n_clip %lld)", (long long)in->n_clip); return 1; }
        if (((uintptr_t)in->clip_rec & 15) != 0) { snprintf(ctx->err, sizeof ctx->err, "clip_rec must be 16-byte aligned"); return 1; }
        R->n_clip = in->n_clip; R->clip_rec = in->clip_rec; R->clip_tile = in->clip_tile;
    }
    const int nc = in->n_contigs;
    int64_t pad = (int64_t)pr->min_distance_to_mate + pr->max_direct_repeat_length + 64;
    if (pad < 64) pad = 64;
    int64_t *h_rs = (int64_t *)h; int32_t *h_len = (int32_t *)(h_rs + nc + 1); uint32_t *h_base = (uint32_t *)(h_len + nc);
    int64_t base = 0;
    for (int c = 0; c < nc; c++) {
        h_len[c] = in->h_contig_len[c];
        h_base[c] = (uint32_t)base;
        base += (int64_t)in->h_contig_len[c] + 2 + pad;
        if (in->h_contig_read_start[c + 1] < in->h_contig_read_start[c]) { snprintf(ctx->err, sizeof ctx->err, "contig_read_start is not monotone"); return 1; }
    }
    if (base >= ((int64_t)1 << 32)) { snprintf(ctx->err, sizeof ctx->err, "total reference length %lld does not fit the 32-bit site axis", (long long)base); return 1; }
    for (int c = 0; c <= nc; c++) h_rs[c] = in->h_contig_read_start[c];
    R->gpos_span = (uint32_t)base;
    *used = (size_t)((char *)(h_base + nc) - h);
    P->minlen = pr->min_softclip_length; P->mincount = pr->min_softclip_count; P->minq = pr->min_alignment_quality;
    P->minial = pr->min_alignment_inner_length; P->mindist = pr->min_distance_to_mate; P->lins = pr->large_insertion_cutoff;
    P->mcc = pr->min_count_consensus; P->maxdr = pr->max_direct_repeat_length;
    P->minratio = pr->min_softclip_ratio; P->maxir = pr->max_indel_ratio; P->maxjsp = pr->max_junction_spanning_prop;
    return 0;
}

// the header arrays live at `d_hdr` on the device, laid out as prepare_reads wrote them
void set_header_ptrs(DevReads *R, char *d_hdr) {
    const int nc = R->n_contigs;
    int64_t *d_rs = (int64_t *)d_hdr; int32_t *d_len = (int32_t *)(d_rs + nc + 1); uint32_t *d_base = (uint32_t *)(d_len + nc);
    R->contig_len = d_len; R->contig_rs = d_rs; R->contig_base = d_base;
}

int upload_reads(mfk_ctx *ctx, const mfk_reads *in, const mfk_params *pr, DevReads *R, DevParams *P) {
    if (in->n_contigs <= 0 || in->n_contigs > 65535) { snprintf(ctx->err, sizeof ctx->err, "n_contigs %d out of range", in->n_contigs); return 1; }
    // header arrays staged in pinned memory, one copy
    const size_t bytes = upload_reads_pinned_bytes(in->n_contigs);
    char *h = (char *)pinned_buf(ctx, bytes);
    if (!h) { snprintf(ctx->err, sizeof ctx->err, "pinned allocation failed"); return 1; }
    size_t used = 0;
    if (int rc = prepare_reads(ctx, in, pr, R, P, h, &used)) return rc;
    // The header lives in an allocation of its own and is uploaded only when its bytes change: findflanks, pairflanks
    // and recall on the same reads (and every step of a loop over resident reads) reuse it, which saves a copy and the
    // gap behind it at the head of every call.
    if (bytes > ctx->hdr_cap) {
        HIPCHK(hipStreamSynchronize(ctx->stream));
        if (ctx->hdr_dev) (void)hipFree(ctx->hdr_dev);
        ctx->hdr_dev = nullptr; ctx->hdr_cap = 0; ctx->hdr_last.clear();
        HIPCHK(hipMalloc((void **)&ctx->hdr_dev, bytes * 2));
        ctx->hdr_cap = bytes * 2;
    }
    char *d_hdr = ctx->hdr_dev;
    if (ctx->hdr_last.size() != used || memcmp(ctx->hdr_last.data(), h, used) != 0) {
        ctx->hdr_last.clear();   // stays empty if the copy fails
        if (int rc = upload_pinned(ctx, d_hdr, h, bytes)) return rc;
        // no synchronisation: later read-backs into the same pinned bytes are ordered behind this copy on the stream
        ctx->hdr_last.assign(h, h + used);
    }
    set_header_ptrs(R, d_hdr);
    return 0;
}

// ---- rocPRIM wrappers ------------------------------------------------------------------------------------------
int dev_sort_keys64(mfk_ctx *ctx, uint64_t *in, uint64_t *out, size_t n, int begin_bit, int end_bit) {
    if (n == 0) return 0;
    size_t tb = 0;
    HIPCHK(rocprim::radix_sort_keys(nullptr, tb, in, out, n, (unsigned)begin_bit, (unsigned)end_bit, ctx->stream));
    void *tmp = arena_alloc(ctx, tb ? tb : 1);
    if (!tmp) return 1;
    prof_begin(ctx, "rocprim::radix_sort_keys");
    HIPCHK(rocprim::radix_sort_keys(tmp, tb, in, out, n, (unsigned)begin_bit, (unsigned)end_bit, ctx->stream));
    prof_end(ctx);
    return 0;
}

int dev_sort_pairs64(mfk_ctx *ctx, uint64_t *kin, uint64_t *kout, uint32_t *vin, uint32_t *vout, size_t n, int begin_bit, int end_bit) {
    if (n == 0) return 0;
    size_t tb = 0;
    HIPCHK(rocprim::radix_sort_pairs(nullptr, tb, kin, kout, vin, vout, n, (unsigned)begin_bit, (unsigned)end_bit, ctx->stream));
    void *tmp = arena_alloc(ctx, tb ? tb : 1);
    if (!tmp) return 1;
    prof_begin(ctx, "rocprim::radix_sort_pairs");
    HIPCHK(rocprim::radix_sort_pairs(tmp, tb, kin, kout, vin, vout, n, (unsigned)begin_bit, (unsigned)end_bit, ctx->stream));
    prof_end(ctx);
    return 0;
}

// ---- exclusive scans with the total appended: out has n+1 entries ----------------------------------------------------
// n <= SCAN_SMALL: one workgroup walks the array (one launch; 6 us for 8 k elements, but 28 us for 50 k, measured at C1,
// where rocPRIM's two launches take 14).  Larger: rocPRIM's look-back scan over n+1 elements whose
// last input is a virtual zero, so the total lands in out[n] without an extra kernel.
#define SCAN_SMALL (16 * 1024)
#define SCAN_ITEMS 8
template <typename TOut>
__global__ __launch_bounds__(1024) void k_scan_small(const uint32_t *in, TOut *out, uint32_t n) {
    __shared__ TOut wsum[16];
    __shared__ TOut carry_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    TOut carry = 0;
    for (uint32_t base = 0; base < n; base += 1024 * SCAN_ITEMS) {
        const uint32_t i0 = base + threadIdx.x * SCAN_ITEMS;
        uint32_t v[SCAN_ITEMS];
        TOut tsum = 0;
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; k++) { v[k] = i0 + k < n ? in[i0 + k] : 0u; tsum += v[k]; }
        TOut inc = tsum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { TOut t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        TOut wbase = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) { TOut s = wsum[w]; if (w < wave) wbase += s; tot += s; }
        TOut run = carry + wbase + inc - tsum;
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; k++) { if (i0 + k < n) out[i0 + k] = run; run += v[k]; }
        carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) out[n] = carry;
    (void)carry_s;
}

// SCAN_SMALL < n <= SCAN_CHAIN_TILES tiles: one launch, one 8192-element tile per workgroup.  A workgroup publishes its
// tile's sum (epoch << 48 | sum, one 64-bit store) as soon as it has it, then its first wave reads the entries of all
// tiles before it, 64 at a time, waiting for those not yet tagged with this call's epoch.  A workgroup only waits for
// lower-numbered ones, which are dispatched before it.  The epoch changes with every call, so the state array is never
// cleared (the host clears it when the 16-bit epoch wraps).  Sums stay below 2^48 (inputs are 32-bit, n <= 2^21).
template <typename TOut>
__global__ __launch_bounds__(1024) void k_scan_chain(const uint32_t *in, TOut *out, uint32_t n, unsigned long long *state, uint32_t epoch, uint32_t *err) {
    __shared__ unsigned long long wsum[16];
    __shared__ unsigned long long s_prefix;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t tile = blockIdx.x;
    const uint32_t i0 = tile * (1024 * SCAN_ITEMS) + threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    if (i0 + SCAN_ITEMS <= n && ((uintptr_t)in & 15) == 0) {
        const uint4 a = *reinterpret_cast<const uint4 *>(in + i0), b = *reinterpret_cast<const uint4 *>(in + i0 + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; k++) v[k] = i0 + k < n ? in[i0 + k] : 0u;
    }
    unsigned long long tsum = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) tsum += v[k];
    unsigned long long inc = tsum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { unsigned long long t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    unsigned long long wbase = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) { unsigned long long s = wsum[w]; if (w < wave) wbase += s; tot += s; }
    if (wave == 0) {
        if (lane == 0) __hip_atomic_store(&state[tile], ((unsigned long long)epoch << 48) | tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long acc = 0;
        for (uint32_t p0 = 0; p0 < tile; p0 += 64) {
            const uint32_t p = p0 + lane;
            if (p < tile) {
                unsigned long long e;
                uint32_t spins = 0;
                do {
                    e = __hip_atomic_load(&state[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } while ((uint32_t)(e >> 48) != epoch && ++spins < (1u << 24));
                // a predecessor never ran: never hang, never kill the process: its sum counts as 0 (every offset stays inside
                // the arrays sized by the true total) and the error word fails the call at its next read-back
                if ((uint32_t)(e >> 48) != epoch) atomicOr(err, 1u);
                else acc += e & 0xFFFFFFFFFFFFull;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (lane == 0) s_prefix = acc;
    }
    __syncthreads();
    const unsigned long long before = s_prefix;
    unsigned long long run = before + wbase + inc - tsum;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) { if (i0 + k < n) out[i0 + k] = (TOut)run; run += v[k]; }
    if (tile == gridDim.x - 1 && threadIdx.x == 0) out[n] = (TOut)(before + tot);
}

template <typename TOut>
struct PaddedIn {
    const uint32_t *in; size_t n;
    __device__ TOut operator()(size_t i) const { return i < n ? (TOut)in[i] : (TOut)0; }
};

template <typename TOut>
static int dev_excl_scan(mfk_ctx *ctx, const uint32_t *in, TOut *out, size_t n) {
    if (n == 0) { HIPCHK(hipMemsetAsync(out, 0, sizeof(TOut), ctx->stream)); return 0; }
    if (n <= SCAN_SMALL) {
        prof_begin(ctx, "k_scan_small");
        hipLaunchKernelGGL(k_scan_small<TOut>, dim3(1), dim3(1024), 0, ctx->stream, in, out, (uint32_t)n);
        prof_end(ctx);
        KCHK();
        return 0;
    }
    static const bool no_chain = getenv("MFK_NO_SCAN_CHAIN") && atoi(getenv("MFK_NO_SCAN_CHAIN"));
    if (n <= (size_t)SCAN_CHAIN_TILES * 1024 * SCAN_ITEMS && !no_chain) {
        if (++ctx->scan_epoch >= 0xFFFFu) {   // the tag wraps: entries of 65534 calls ago must not pass for this call's
            HIPCHK(hipMemsetAsync(ctx->scan_state, 0, SCAN_CHAIN_TILES * 8, ctx->stream));
            ctx->scan_epoch = 1;
        }
        const unsigned tiles = (unsigned)((n + 1024 * SCAN_ITEMS - 1) / (1024 * SCAN_ITEMS));
        prof_begin(ctx, "k_scan_chain");
        hipLaunchKernelGGL(k_scan_chain<TOut>, dim3(tiles), dim3(1024), 0, ctx->stream, in, out, (uint32_t)n, ctx->scan_state, ctx->scan_epoch, ctx->scan_err);
        prof_end(ctx);
        KCHK();
        return 0;
    }
    size_t tb = 0;
    auto it = rocprim::make_transform_iterator(rocprim::make_counting_iterator<size_t>(0), PaddedIn<TOut>{in, n});
    HIPCHK(rocprim::exclusive_scan(nullptr, tb, it, out, (TOut)0, n + 1, rocprim::plus<TOut>(), ctx->stream));
    void *tmp = arena_alloc(ctx, tb ? tb : 1);
    if (!tmp) return 1;
    prof_begin(ctx, "rocprim::exclusive_scan");
    HIPCHK(rocprim::exclusive_scan(tmp, tb, it, out, (TOut)0, n + 1, rocprim::plus<TOut>(), ctx->stream));
    prof_end(ctx);
    return 0;
}

int dev_excl_scan32(mfk_ctx *ctx, const uint32_t *in, uint32_t *out, size_t n) { return dev_excl_scan<uint32_t>(ctx, in, out, n); }
int dev_excl_scan64(mfk_ctx *ctx, const uint32_t *in, uint64_t *out, size_t n) { return dev_excl_scan<uint64_t>(ctx, in, out, n); }
