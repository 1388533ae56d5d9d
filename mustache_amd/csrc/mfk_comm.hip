// mfk_comm.hip -- the one exchange step of the multi-GPU path: an all-gather of per-rank tables (site / flank / pair rows of
// fixed width plus their string bytes) over RCCL, inside the library.  The reference has no counterpart (its processes never
// communicate; the Snakemake workflow only schedules independent jobs): this is what a cohort-level
// consumer or the coordinate-sharded findflanks -> pairflanks hand-over needs once the samples / coordinate ranges of one
// job live on several GPUs.
//
// RCCL is bound at run time (dlopen), not at link time: a process that already holds an RCCL (PyTorch ships its own copy and
// loads it for torch.distributed) keeps exactly one, and a single-GPU user of libmfk.so needs none.
//
// The tables are kilobytes to a few megabytes, so the collective is latency-bound (tens of microseconds over xGMI) and the
// design goal is ONE collective per merge: every rank sends a 16-byte header (rows, string bytes) followed by its rows and
// strings, padded to a common size.  Only when some rank's table is larger than that size does a second all-gather with the
// true maximum follow; all ranks see the same headers, so they agree on that without talking.
#include "mfk_host.h"

#include <dlfcn.h>
#include <rccl/rccl.h>   // types and prototypes only: every call goes through the table below

namespace {
struct Rccl {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    char why[256] = {0};
};
Rccl g_rccl;

bool rccl_load() {
    if (g_rccl.handle) return true;
    // a copy that is already in the process first (any of the names it may have been loaded under), then the ROCm one
    const char *names[] = {"librccl.so", "librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names) if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
    const char *paths[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *p : paths) if (!h) h = dlopen(p, RTLD_NOW | RTLD_GLOBAL);
    if (!h) { snprintf(g_rccl.why, sizeof g_rccl.why, "librccl.so not found: %s", dlerror()); return false; }
#define MFK_SYM(field, name)                                                                                       \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name));                                      \
    if (!g_rccl.field) { snprintf(g_rccl.why, sizeof g_rccl.why, "%s missing from librccl.so", name); return false; }
    MFK_SYM(GetUniqueId, "ncclGetUniqueId")
    MFK_SYM(CommInitRank, "ncclCommInitRank")
    MFK_SYM(AllGather, "ncclAllGather")
    MFK_SYM(CommDestroy, "ncclCommDestroy")
    MFK_SYM(GetErrorString, "ncclGetErrorString")
#undef MFK_SYM
    g_rccl.handle = h;
    return true;
}
}   // namespace

#define NCCLCHK(call)                                                                                               \
    do {                                                                                                            \
        ncclResult_t r_ = (call);                                                                                   \
        if (r_ != ncclSuccess) {                                                                                    \
            snprintf(ctx->err, sizeof ctx->err, "%s failed: %s", #call, g_rccl.GetErrorString(r_));                 \
            return 101;                                                                                             \
        }                                                                                                           \
    } while (0)

extern "C" int mfk_comm_unique_id(mfk_ctx *ctx, void *id128) {
    if (!ctx || !id128) return 1;
    if (!rccl_load()) { snprintf(ctx->err, sizeof ctx->err, "%s", g_rccl.why); return 102; }
    static_assert(sizeof(ncclUniqueId) == MFK_COMM_ID_BYTES, "MFK_COMM_ID_BYTES");
    ncclUniqueId id;
    NCCLCHK(g_rccl.GetUniqueId(&id));
    memcpy(id128, &id, sizeof id);
    return 0;
}

extern "C" int mfk_comm_free(mfk_ctx *ctx) {
    if (!ctx) return 1;
    if (ctx->comm) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
        (void)g_rccl.CommDestroy(reinterpret_cast<ncclComm_t>(ctx->comm));
        ctx->comm = nullptr;
    }
    ctx->comm_world = 0; ctx->comm_rank = 0;
    // what the last gather left belongs to the communicator that made it: the size hint is only the same on every rank
    // because every rank of THAT communicator saw the same headers; a ctx that joins another one starts from 16 KB again
    ctx->gat_hint = 0; ctx->gat_msg = 0; ctx->gat_row_bytes = 0;
    ctx->gat.clear(); ctx->gat_counts.clear();
    return 0;
}

// size of the next exchange when the caller passes pad_bytes <= 0 (diagnostic; tests check that a new communicator starts over)
extern "C" int64_t mfk_allgather_next_bytes(mfk_ctx *ctx) {
    if (!ctx) return -1;
    size_t size = ctx->gat_hint ? ctx->gat_hint + ctx->gat_hint / 4 : ((size_t)16 << 10);
    return (int64_t)((size + 255) & ~(size_t)255);
}

extern "C" int mfk_comm_init(mfk_ctx *ctx, const void *id128, int rank, int world) {
    if (!ctx || !id128 || world < 1 || rank < 0 || rank >= world) return 1;
    if (!rccl_load()) { snprintf(ctx->err, sizeof ctx->err, "%s", g_rccl.why); return 102; }
    mfk_comm_free(ctx);
    HIPCHK(hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    ncclComm_t comm = nullptr;
    NCCLCHK(g_rccl.CommInitRank(&comm, world, id, rank));
    ctx->comm = comm; ctx->comm_rank = rank; ctx->comm_world = world;
    return 0;
}

extern "C" int mfk_allgather_sites(mfk_ctx *ctx, const void *rows, int64_t n_rows, int32_t row_bytes, const void *blob, int64_t blob_bytes,
                                   int64_t pad_bytes, int64_t *counts) {
    if (!ctx || n_rows < 0 || row_bytes <= 0 || blob_bytes < 0 || (n_rows > 0 && !rows) || (blob_bytes > 0 && !blob)) return 1;
    if (!ctx->comm) { snprintf(ctx->err, sizeof ctx->err, "mfk_allgather_sites: no communicator (call mfk_comm_init first)"); return 1; }
    HIPCHK(hipSetDevice(ctx->device));
    ctx->err[0] = 0;
    const int world = ctx->comm_world;
    const size_t body = (size_t)n_rows * (size_t)row_bytes + (size_t)blob_bytes;
    // pad_bytes <= 0: the common size follows the tables: 16 KB at first, then the largest table of the previous gather plus a
    // quarter (every rank has seen the same headers, so every rank arrives at the same size): a loop over steps with tables of
    // a few KB moves a few KB per rank, not a fixed maximum
    size_t size = pad_bytes > 0 ? (size_t)pad_bytes : (ctx->gat_hint ? ctx->gat_hint + ctx->gat_hint / 4 : ((size_t)16 << 10));
    size = (size + 255) & ~(size_t)255;
    ctx->gat_counts.assign(2 * (size_t)world, 0);
    ctx->gat_row_bytes = row_bytes;
    hipStream_t st = ctx->stream;
    for (int attempt = 0; attempt < 2; attempt++) {
        const size_t msg = 16 + size;
        arena_reset(ctx);
        char *pin = (char *)pinned_buf(ctx, msg * ((size_t)world + 1));
        char *d_send = aalloc<char>(ctx, msg), *d_recv = aalloc<char>(ctx, msg * (size_t)world);
        if (!pin || !d_send || !d_recv) { snprintf(ctx->err, sizeof ctx->err, "mfk_allgather_sites: out of memory"); return 1; }
        const int64_t head[2] = {n_rows, blob_bytes};
        memcpy(pin, head, 16);
        const size_t nb = body < size ? body : size;           // a table beyond the common size goes in the second exchange
        const size_t rb = (size_t)n_rows * (size_t)row_bytes;
        if (nb) {
            const size_t r1 = rb < nb ? rb : nb;
            if (r1) memcpy(pin + 16, rows, r1);
            if (nb > r1) memcpy(pin + 16 + r1, blob, nb - r1);
        }
        HIPCHK(hipMemcpyAsync(d_send, pin, 16 + nb, hipMemcpyHostToDevice, st));
        NCCLCHK(g_rccl.AllGather(d_send, d_recv, msg, ncclUint8, reinterpret_cast<ncclComm_t>(ctx->comm), st));
        char *got = pin + msg;
        // headers first would cost a second round trip: the tables are small, copy everything
        HIPCHK(hipMemcpyAsync(got, d_recv, msg * (size_t)world, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        size_t need = 0;
        for (int r = 0; r < world; r++) {
            int64_t h2[2];
            memcpy(h2, got + (size_t)r * msg, 16);
            if (h2[0] < 0 || h2[1] < 0) { snprintf(ctx->err, sizeof ctx->err, "mfk_allgather_sites: rank %d sent a negative count", r); return 1; }
            ctx->gat_counts[2 * r] = h2[0]; ctx->gat_counts[2 * r + 1] = h2[1];
            const size_t b = (size_t)h2[0] * (size_t)row_bytes + (size_t)h2[1];
            if (b > need) need = b;
        }
        ctx->gat_hint = need < 4096 ? 4096 : need;
        if (need <= size) {
            // keep what the ranks sent, not the padding: rank r's rows and strings at r * (16 + need)
            const size_t keep = 16 + ((need + 15) & ~(size_t)15);
            ctx->gat.resize(keep * (size_t)world);
            for (int r = 0; r < world; r++)
                memcpy(ctx->gat.data() + (size_t)r * keep, got + (size_t)r * msg, 16 + (size_t)ctx->gat_counts[2 * r] * (size_t)row_bytes + (size_t)ctx->gat_counts[2 * r + 1]);
            ctx->gat_msg = keep;
            if (counts) memcpy(counts, ctx->gat_counts.data(), 2 * (size_t)world * sizeof(int64_t));
            ctx->call_ok = true;
            return 0;
        }
        size = (need + 255) & ~(size_t)255;   // every rank computes the same value from the same headers
    }
    snprintf(ctx->err, sizeof ctx->err, "mfk_allgather_sites: sizes changed between the two exchanges");
    return 1;
}

extern "C" int mfk_allgather_fetch(mfk_ctx *ctx, int rank, void *rows, void *blob) {
    if (!ctx || rank < 0 || rank >= ctx->comm_world || ctx->gat.empty()) return 1;
    const char *p = reinterpret_cast<const char *>(ctx->gat.data()) + (size_t)rank * ctx->gat_msg + 16;
    const size_t rb = (size_t)ctx->gat_counts[2 * rank] * (size_t)ctx->gat_row_bytes, bb = (size_t)ctx->gat_counts[2 * rank + 1];
    if (rows && rb) memcpy(rows, p, rb);
    if (blob && bb) memcpy(blob, p + rb, bb);
    return 0;
}

// every rank's rows back to back (rank order) and every rank's strings back to back: one call instead of one per rank
extern "C" int mfk_allgather_fetch_all(mfk_ctx *ctx, void *rows, void *blob) {
    if (!ctx || ctx->comm_world < 1 || ctx->gat.empty()) return 1;
    char *ro = static_cast<char *>(rows), *bo = static_cast<char *>(blob);
    for (int r = 0; r < ctx->comm_world; r++) {
        const char *p = reinterpret_cast<const char *>(ctx->gat.data()) + (size_t)r * ctx->gat_msg + 16;
        const size_t rb = (size_t)ctx->gat_counts[2 * r] * (size_t)ctx->gat_row_bytes, bb = (size_t)ctx->gat_counts[2 * r + 1];
        if (ro && rb) { memcpy(ro, p, rb); ro += rb; }
        if (bo && bb) { memcpy(bo, p + rb, bb); bo += bb; }
    }
    return 0;
}
