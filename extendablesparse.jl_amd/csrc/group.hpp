// group.hpp -- column-range shards across the GPUs of a node, one process per GPU: the
// esp_group_* entry points.  Included at the end of shard.hip (it drives the esp_shard_* calls of one handle); the exchange
// POLICY is group_policy.hpp (host-only), this file binds it to the device and holds the RCCL transport.
//
// What the reference does with threads -- GenericMTExtendableSparseMatrixCSC: one buffer per `tid`, flush! =
// Base.sum(xmatrices, csc) (src/matrix/genericmtextendablesparsematrixcsc.jl:45-51,87-99) -- happens here across
// processes: every rank appends whatever its part of the assembly loop produces, esp_group_flush routes every
// pending entry to the rank that owns its column (owner(col) = floor((col-1)*P/n)) with ONE all-to-all-v and runs the
// local flush.  The received entries are ordered by source rank and keep the source's append order: the result equals
// ONE buffer fed the ranks' streams in turn.
//
// Transport: RCCL (xGMI), loaded at run time with dlopen -- the library that a host process already holds (PyTorch
// bundles its own librccl.so.1; two copies in one process would clash) or the system one; ESP_RCCL_LIB overrides.
// Grouped ncclSend/ncclRecv ON THE HANDLE'S STREAM: the exchange is stream-ordered behind the partition kernel and in
// front of the bucket kernel, no host synchronisation in between.  A host that brings its own transport (MPI, the
// in-process harness of the tests) passes a callback table instead (esp_group_create_comm).
#pragma once
#include <mutex>
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "group_policy.hpp"

struct RcclApi {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
static RcclApi g_rccl;
static std::string g_rccl_err;
static std::mutex g_rccl_mutex;  // (handles of different host threads may create their groups at the same time)

static bool rccl_load() {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.lib) return true;
    void *lib = nullptr;
    if (const char *e = getenv("ESP_RCCL_LIB")) lib = dlopen(e, RTLD_NOW | RTLD_LOCAL);
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);  // the copy the process already holds
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib) {
        const char *why = dlerror();  // (one call: dlerror() clears the pending message)
        g_rccl_err = std::string("cannot load librccl.so.1 (") + (why ? why : "?") + "); set ESP_RCCL_LIB";
        return false;
    }
    RcclApi a;
    a.lib = lib;
#define ESP_SYM(field, name)                                            \
    a.field = reinterpret_cast<decltype(a.field)>(dlsym(lib, name));    \
    if (!a.field) {                                                     \
        g_rccl_err = std::string("librccl has no symbol ") + name;      \
        return false;                                                   \
    }
    ESP_SYM(GetUniqueId, "ncclGetUniqueId")
    ESP_SYM(CommInitRank, "ncclCommInitRank")
    ESP_SYM(CommDestroy, "ncclCommDestroy")
    ESP_SYM(GroupStart, "ncclGroupStart")
    ESP_SYM(GroupEnd, "ncclGroupEnd")
    ESP_SYM(Send, "ncclSend")
    ESP_SYM(Recv, "ncclRecv")
    ESP_SYM(AllGather, "ncclAllGather")
    ESP_SYM(GetErrorString, "ncclGetErrorString")
#undef ESP_SYM
    g_rccl = a;
    return true;
}

struct esp_group {
    esp_handle *h = nullptr;
    int P = 1, me = 0;
    espgroup::Policy pol;    // the exchange policy (group_policy.hpp: host-only, also run under sanitizers by a CPU test)
    bool own_rccl = false;
    ncclComm_t nccl = nullptr;
    DevBuf ag;               // staging of the small all-gathers (RCCL transport)
    DevBuf rkeys, rvals, rcnts;    // receive buffers: alive until the local flush has read them
    // test hook (esp_debug_group_loopback; a single-rank group over the library's RCCL transport): nothing is skipped
    // because there is only one rank -- the small agreements run as a real ncclAllGather on the second stream, and every
    // flush sends the rank's own range to ITSELF through rccl_alltoallv (grouped ncclSend / ncclRecv to self, 1 GiB
    // rounds), wipes the range and restores it from what arrived.  The transport then really runs on a one-GPU box.
    bool loopback = false;
    DevBuf loop;
    i64 loop_bytes = 0;            // bytes that travelled through RCCL in the last flush
    std::string err;
};

#define GFAIL(g, code, ...)                                  \
    do {                                                     \
        char _b[512];                                        \
        snprintf(_b, sizeof _b, __VA_ARGS__);                \
        (g)->err = _b;                                       \
        if ((g)->h) (g)->h->err = _b;                        \
        g_err = _b;                                          \
        return (code);                                       \
    } while (0)

// ---- RCCL transport ---------------------------------------------------------------------------------
static int32_t rccl_allgather_i64(void *ctx, const int64_t *send, int32_t count, int64_t *recv) {
    esp_group *g = static_cast<esp_group *>(ctx);
    esp_handle *h = g->h;
    if (g->P == 1 && !g->loopback) {
        memcpy(recv, send, sizeof(int64_t) * (size_t)count);
        return ESP_OK;
    }
    CK(aux_ready(h));
    // on the handle's second stream (highest priority): a tiny collective queued on the main stream would wait for
    // whatever big kernel runs there
    CK(ensure(h, g->ag, sizeof(i64) * (size_t)count * (size_t)(g->P + 1)));
    i64 *d_send = (i64 *)g->ag.p, *d_recv = d_send + count;
    HIPCK(h, hipMemcpyAsync(d_send, send, sizeof(i64) * (size_t)count, hipMemcpyHostToDevice, h->aux));
    const ncclResult_t r = g_rccl.AllGather(d_send, d_recv, (size_t)count, ncclInt64, g->nccl, h->aux);
    if (r != ncclSuccess) GFAIL(g, ESP_ERR_HIP, "ncclAllGather failed: %s", g_rccl.GetErrorString(r));
    HIPCK(h, hipMemcpyAsync(recv, d_recv, sizeof(i64) * (size_t)count * (size_t)g->P, hipMemcpyDeviceToHost, h->aux));
    HIPCK(h, hipStreamSynchronize(h->aux));
    return ESP_OK;
}

// one grouped launch per round; a round moves at most 1 GiB per pair
static int32_t rccl_alltoallv(void *ctx, const void *const *send, const int64_t *send_bytes, void *const *recv,
                              const int64_t *recv_bytes, void *hip_stream) {
    esp_group *g = static_cast<esp_group *>(ctx);
    const i64 ROUND = (i64)1 << 30;
    i64 big = 0;
    const bool self_too = g->loopback;  // (test hook: the rank is its own peer)
    for (int q = 0; q < g->P; q++)
        if (q != g->me || self_too) big = std::max(big, std::max<i64>(send_bytes[q], recv_bytes[q]));
    // (the number of rounds must be the same on both ends of a pair: every send of `big` bytes is matched by a
    // receive of the same size, so max over my own pairs is enough for each pair taken alone; rounds beyond a pair's
    // size move nothing)
    for (i64 off = 0; off < big; off += ROUND) {
        ncclResult_t r = g_rccl.GroupStart();
        for (int q = 0; q < g->P && r == ncclSuccess; q++) {
            if (q == g->me && !self_too) continue;
            const i64 s = std::min(ROUND, send_bytes[q] - off), t = std::min(ROUND, recv_bytes[q] - off);
            if (s > 0) r = g_rccl.Send((const char *)send[q] + off, (size_t)s, ncclChar, q, g->nccl, (hipStream_t)hip_stream);
            if (t > 0 && r == ncclSuccess) r = g_rccl.Recv((char *)recv[q] + off, (size_t)t, ncclChar, q, g->nccl, (hipStream_t)hip_stream);
        }
        const ncclResult_t e = g_rccl.GroupEnd();
        if (r == ncclSuccess) r = e;
        if (r != ncclSuccess) GFAIL(g, ESP_ERR_HIP, "RCCL all-to-all-v failed: %s", g_rccl.GetErrorString(r));
    }
    return ESP_OK;
}

// ---- lifetime ---------------------------------------------------------------------------------------
extern "C" int32_t esp_group_unique_id(uint8_t *id128) {
    if (!id128) return ESP_ERR_INVALID;
    if (!rccl_load()) FAIL((esp_handle *)nullptr, ESP_ERR_UNSUPPORTED, "esp_group_unique_id: %s", g_rccl_err.c_str());
    static_assert(sizeof(ncclUniqueId) == 128, "the id travels as 128 bytes");
    ncclUniqueId id;
    const ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) FAIL((esp_handle *)nullptr, ESP_ERR_HIP, "ncclGetUniqueId failed: %s", g_rccl.GetErrorString(r));
    memcpy(id128, &id, 128);
    return ESP_OK;
}

// what esp_group_create* changes on the caller's handle, put back when the group cannot be made after all
struct HandleShardState {
    bool shard_user, win_excl;
    u64 win_base, win_span;
    i64 wc0, wc1;
    explicit HandleShardState(const esp_handle *h) : shard_user(h->shard_user), win_excl(h->win_excl), win_base(h->win_base), win_span(h->win_span), wc0(h->wc0), wc1(h->wc1) {}
    void restore(esp_handle *h) const {
        h->shard_user = shard_user, h->win_excl = win_excl;
        h->win_base = win_base, h->win_span = win_span;
        h->wc0 = wc0, h->wc1 = wc1;
    }
};

static void bind_shard_ops(esp_group *g);
static int32_t group_common(esp_handle *h, int32_t nranks, int32_t rank, esp_group **out, esp_group **made) {
    if (!h || !out) return ESP_ERR_INVALID;
    *out = nullptr;
    if (nranks < 1 || nranks > 256 || rank < 0 || rank >= nranks) FAIL(h, ESP_ERR_INVALID, "esp_group_create: rank %d of %d", rank, nranks);
    if (h->count != 0 || h->nnz != 0) FAIL(h, ESP_ERR_STATE, "esp_group_create: the handle must be empty (its column window is declared now)");
    esp_group *g = new esp_group();
    g->h = h;
    g->P = nranks;
    g->me = rank;
    g->pol.init(nranks, rank);
    bind_shard_ops(g);
    const HandleShardState before(h);
    h->shard_user = true;
    const i64 c0 = shard_col0(h->n, nranks, rank), c1 = shard_col0(h->n, nranks, rank + 1);
    if (c1 > c0) {  // after the exchange every pending column is owned: flushes and reset! work on the own range only
        const int32_t st = esp_set_column_window(h, c0 + 1, c1);
        if (st != ESP_OK) {
            before.restore(h);
            delete g;
            return st;
        }
    }
    *made = g;
    return ESP_OK;
}

extern "C" int32_t esp_group_create(esp_handle *h, int32_t nranks, int32_t rank, const uint8_t *id128, esp_group **out) {
    if (!id128) return ESP_ERR_INVALID;
    if (!rccl_load()) FAIL(h, ESP_ERR_UNSUPPORTED, "esp_group_create: %s", g_rccl_err.c_str());
    esp_group *g = nullptr;
    const HandleShardState before(h);
    CK(group_common(h, nranks, rank, out, &g));
    (void)hipSetDevice(h->device);
    ncclUniqueId id;
    memcpy(&id, id128, 128);
    ncclResult_t r = ncclSuccess;
    if (getenv("ESP_DEBUG_FAIL_COMM_INIT"))  // (test hook: the failure path below without a broken fabric)
        r = ncclInvalidArgument;
    else
        r = g_rccl.CommInitRank(&g->nccl, nranks, id, rank);
    if (r != ncclSuccess) {
        before.restore(h);  // (the handle is the caller's plain handle again: no shard, no column window)
        delete g;
        FAIL(h, ESP_ERR_HIP, "ncclCommInitRank failed: %s", g_rccl.GetErrorString(r));
    }
    g->own_rccl = true;
    g->pol.comm.ctx = g;
    g->pol.comm.allgather_i64 = rccl_allgather_i64;
    g->pol.comm.alltoallv_dev = rccl_alltoallv;
    *out = g;
    return ESP_OK;
}

extern "C" int32_t esp_group_create_comm(esp_handle *h, int32_t nranks, int32_t rank, const esp_comm_t *comm, esp_group **out) {
    if (!comm || !comm->allgather_i64 || !comm->alltoallv_dev) return ESP_ERR_INVALID;
    esp_group *g = nullptr;
    CK(group_common(h, nranks, rank, out, &g));
    g->pol.comm = *comm;
    *out = g;
    return ESP_OK;
}

extern "C" int32_t esp_group_destroy(esp_group *g) {
    if (!g) return ESP_OK;
    if (g->h) {
        (void)hipSetDevice(g->h->device);
        (void)hipStreamSynchronize(g->h->stream);
    }
    release(g->ag);
    release(g->rkeys);
    release(g->rvals);
    release(g->rcnts);
    release(g->loop);
    if (g->own_rccl && g->nccl) (void)g_rccl.CommDestroy(g->nccl);
    delete g;
    return ESP_OK;
}

extern "C" int32_t esp_group_handle(esp_group *g, esp_handle **out) {
    if (!g || !out) return ESP_ERR_INVALID;
    *out = g->h;
    return ESP_OK;
}
extern "C" const char *esp_group_last_error(const esp_group *g) { return g ? g->err.c_str() : g_err.c_str(); }

extern "C" int32_t esp_group_column_range(const esp_group *g, int64_t *col_lo, int64_t *col_hi) {
    if (!g) return ESP_ERR_INVALID;
    if (col_lo) *col_lo = shard_col0(g->h->n, g->P, g->me) + 1;
    if (col_hi) *col_hi = shard_col0(g->h->n, g->P, g->me + 1);
    return ESP_OK;
}

// ---- the exchange -----------------------------------------------------------------------------------
// The policy (consensus, back-off, offsets, who sends what to whom) is espgroup::Policy (group_policy.hpp); here: what it
// asks of the shard, bound to the esp_shard_* calls on the device, and the loop-back test hook.

// loop-back test hook: `bytes` at `ptr` (device) go to this very rank through the library's RCCL all-to-all-v, the source
// is wiped and then restored from what arrived -- all on the handle's stream, like a real exchange
static int32_t loopback_roundtrip(esp_group *g, void *ptr, i64 bytes) {
    if (!g->loopback || bytes <= 0) return ESP_OK;
    if (!g->own_rccl || g->P != 1) GFAIL(g, ESP_ERR_STATE, "esp_debug_group_loopback: a single-rank group over the RCCL transport only");
    esp_handle *h = g->h;
    CK(ensure(h, g->loop, (size_t)bytes));
    const void *sp[1] = {ptr};
    void *rp[1] = {g->loop.p};
    const i64 sb[1] = {bytes}, rb[1] = {bytes};
    HIPCK(h, hipMemsetAsync(g->loop.p, 0xA5, (size_t)bytes, h->stream));
    CK(rccl_alltoallv(g, sp, sb, rp, rb, (void *)h->stream));
    HIPCK(h, hipMemsetAsync(ptr, 0, (size_t)bytes, h->stream));
    HIPCK(h, hipMemcpyAsync(ptr, g->loop.p, (size_t)bytes, hipMemcpyDeviceToDevice, h->stream));
    g->loop_bytes += bytes;
    return ESP_OK;
}
extern "C" int32_t esp_debug_group_loopback(esp_group *g, int32_t on, int64_t *bytes_last_flush) {
    if (!g) return ESP_ERR_INVALID;
    if (on >= 0) {
        if (on && (!g->own_rccl || g->P != 1)) GFAIL(g, ESP_ERR_STATE, "esp_debug_group_loopback: a single-rank group over the RCCL transport only");
        g->loopback = on != 0;
    }
    if (bytes_last_flush) *bytes_last_flush = g->loop_bytes;
    return ESP_OK;
}

static void bind_shard_ops(esp_group *g) {
    espgroup::ShardOps &o = g->pol.ops;
    o.ctx = g;
    o.pending = [](void *c, i64 *count) -> int32_t {
        *count = static_cast<esp_group *>(c)->h->count;
        return ESP_OK;
    };
    o.partition = [](void *c, int P, int me, i64 eps, int32_t *ok, void **k, void **v, void **cnt, i64 *eoff, i64 *nb) -> int32_t {
        return esp_shard_partition(static_cast<esp_group *>(c)->h, P, me, eps, ok, reinterpret_cast<uint64_t **>(k), reinterpret_cast<double **>(v),
                                   reinterpret_cast<int64_t **>(cnt), eoff, nb);
    };
    o.plan = [](void *c, int P, int me, i64 eps) -> int32_t { return esp_shard_plan(static_cast<esp_group *>(c)->h, P, me, eps); };
    o.assemble = [](void *c, const void *const *rk, const void *const *rv, const void *const *rc, const i64 *n, int32_t *ok) -> int32_t {
        return esp_shard_assemble(static_cast<esp_group *>(c)->h, reinterpret_cast<const uint64_t *const *>(rk), reinterpret_cast<const double *const *>(rv),
                                  reinterpret_cast<const int64_t *const *>(rc), n, ok);
    };
    o.counts = [](void *c, int P, i64 *counts) -> int32_t { return esp_shard_counts(static_cast<esp_group *>(c)->h, P, counts); };
    o.exchange_begin = [](void *c, int P, int me, i64 lower, i64 higher, void **sk, void **sv, i64 *soff) -> int32_t {
        return esp_shard_exchange_begin(static_cast<esp_group *>(c)->h, P, me, lower, higher, reinterpret_cast<uint64_t **>(sk),
                                        reinterpret_cast<double **>(sv), soff);
    };
    o.exchange_place = [](void *c, i64 pos, const void *k, const void *v, i64 n) -> int32_t {
        return esp_shard_exchange_place(static_cast<esp_group *>(c)->h, pos, static_cast<const uint64_t *>(k), static_cast<const double *>(v), n);
    };
    o.recv_buffers = [](void *c, i64 nrecv, i64 ncounts, void **rk, void **rv, void **rc) -> int32_t {
        esp_group *g = static_cast<esp_group *>(c);
        esp_handle *h = g->h;
        CK(ensure(h, g->rkeys, sizeof(u64) * (size_t)std::max<i64>(nrecv, 1)));
        CK(ensure(h, g->rvals, sizeof(double) * (size_t)std::max<i64>(nrecv, 1)));
        CK(ensure(h, g->rcnts, sizeof(i64) * (size_t)std::max<i64>(ncounts, 1)));
        *rk = g->rkeys.p, *rv = g->rvals.p, *rc = g->rcnts.p;
        return ESP_OK;
    };
    o.flush = [](void *c, int32_t mode, i64 *local_nnz, int32_t *changed) -> int32_t {
        esp_handle *h = static_cast<esp_group *>(c)->h;
        int64_t z = 0;
        CK(esp_flush(h, mode, &z, changed));  // (returns after the bucket kernel has read the receive buffers)
        *local_nnz = h->nnz;
        return ESP_OK;
    };
    o.stream = [](void *c) -> void * { return (void *)static_cast<esp_group *>(c)->h->stream; };
    o.after_split = [](void *c, int partitioned, void *keys, void *vals, i64 entries) -> int32_t {
        esp_group *g = static_cast<esp_group *>(c);
        if (!g->loopback) return ESP_OK;
        esp_handle *h = g->h;
        if (partitioned) {  // (the partitioned ranges -- all the rank's own -- travel through RCCL; 4-byte keys lie inside them)
            CK(loopback_roundtrip(g, keys, 8 * entries));
            return loopback_roundtrip(g, vals, 8 * entries);
        }
        CK(loopback_roundtrip(g, h->keys.p, 8 * h->count));  // (one rank: the pending buffer is the own chunk)
        return loopback_roundtrip(g, h->vals.p, 8 * h->count);
    };
}

// COLLECTIVE: every rank of the group calls it (like flush! of the MT wrapper it is the synchronisation point)
extern "C" int32_t esp_group_flush(esp_group *g, int32_t mode, int64_t *local_nnz, int32_t *pattern_changed) {
    if (!g) return ESP_ERR_INVALID;
    esp_handle *h = g->h;
    (void)hipSetDevice(h->device);
    g->err.clear();
    g->loop_bytes = 0;
    const int32_t st = g->pol.flush(mode, local_nnz, pattern_changed);
    if (st != ESP_OK) {
        const std::string why = g->pol.err + (h->err.empty() ? "" : ": " + h->err) + (g->err.empty() ? "" : ": " + g->err);
        GFAIL(g, st, "%s", why.c_str());
    }
    return ESP_OK;
}

static int32_t group_offsets(esp_group *g) {
    const int32_t st = g->pol.offsets();
    if (st != ESP_OK) GFAIL(g, st, "%s", g->pol.err.c_str());
    return ESP_OK;
}

// COLLECTIVE on first use after a flush (one all-gather of the local nnz)
extern "C" int32_t esp_group_nnz(esp_group *g, int64_t *global_nnz, int64_t *nnz_before_me) {
    if (!g) return ESP_ERR_INVALID;
    CK(group_offsets(g));
    if (global_nnz) *global_nnz = g->pol.nnz_offsets[(size_t)g->P];
    if (nnz_before_me) *nnz_before_me = g->pol.nnz_offsets[(size_t)g->me];
    return ESP_OK;
}

// This shard's part of the global CSC, stitched: colptr_own has (col_hi - col_lo + 2) entries = the GLOBAL 1-based
// colptr[col_lo .. col_hi + 1]; rowval / nzval the local_nnz entries of the own columns.  COLLECTIVE like esp_group_nnz.
extern "C" int32_t esp_group_get_csc(esp_group *g, int64_t *colptr_own, int64_t *rowval, double *nzval) {
    if (!g || !colptr_own) return ESP_ERR_INVALID;
    esp_handle *h = g->h;
    (void)hipSetDevice(h->device);
    CK(group_offsets(g));
    const i64 c0 = shard_col0(h->n, g->P, g->me), c1 = shard_col0(h->n, g->P, g->me + 1);
    CK(fix_tail(h));
    HIPCK(h, hipMemcpyAsync(colptr_own, (const i64 *)h->colptr.p + c0, sizeof(i64) * (size_t)(c1 - c0 + 1), hipMemcpyDeviceToHost, h->stream));
    HIPCK(h, hipStreamSynchronize(h->stream));
    if (colptr_own[0] != 1 || colptr_own[c1 - c0] != h->nnz + 1)
        GFAIL(g, ESP_ERR_STATE, "esp_group_get_csc: entries outside the owned column range");
    const i64 off = g->pol.nnz_offsets[(size_t)g->me];
    for (i64 c = 0; c <= c1 - c0; c++) colptr_own[c] += off;
    if (h->nnz > 0) {
        if (!rowval || !nzval) return ESP_ERR_INVALID;
        CK(d2h_pipelined(h, rowval, h->rowval.p, sizeof(i64) * (size_t)h->nnz));
        CK(d2h_pipelined(h, nzval, h->nzval.p, sizeof(double) * (size_t)h->nnz));
    }
    return ESP_OK;
}

// 1 = partitioned exchange, 2 = in-place exchange; entries this rank sent to other ranks in the last flush
extern "C" int32_t esp_group_last_exchange(const esp_group *g, int32_t *kind, int64_t *sent_off_rank) {
    if (!g) return ESP_ERR_INVALID;
    if (kind) *kind = g->pol.last_exchange;
    if (sent_off_rank) *sent_off_rank = g->pol.sent_off_rank;
    return ESP_OK;
}
