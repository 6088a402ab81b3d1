// group.hpp -- column-range shards across the GPUs of a node, one process per GPU: the
// esp_group_* entry points.  Included at the end of esparse_hip.hip (it drives the esp_shard_* calls of one handle).
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
#include <dlfcn.h>
#include <rccl/rccl.h>

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

static bool rccl_load() {
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
    esp_comm_t comm{};       // the transport in use (RCCL-backed table or the host's)
    bool own_rccl = false;
    ncclComm_t nccl = nullptr;
    DevBuf ag;               // staging of the small all-gathers (RCCL transport)
    // exchange policy (every decision that changes the communication pattern is taken from all-gathered data)
    i64 eps = -1;            // entries per shard of the previous flush: fixes the digit width of the partitioned exchange
    int part_skip = 0, part_penalty = 0;
    int last_exchange = 0;   // 1 partitioned, 2 in place
    i64 sent_off_rank = 0;
    i64 local_nnz = 0;
    std::vector<i64> nnz_offsets;  // P + 1, valid when offsets_valid
    bool offsets_valid = false;
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

static int32_t group_common(esp_handle *h, int32_t nranks, int32_t rank, esp_group **out, esp_group **made) {
    if (!h || !out) return ESP_ERR_INVALID;
    *out = nullptr;
    if (nranks < 1 || nranks > 256 || rank < 0 || rank >= nranks) FAIL(h, ESP_ERR_INVALID, "esp_group_create: rank %d of %d", rank, nranks);
    if (h->count != 0 || h->nnz != 0) FAIL(h, ESP_ERR_STATE, "esp_group_create: the handle must be empty (its column window is declared now)");
    esp_group *g = new esp_group();
    g->h = h;
    g->P = nranks;
    g->me = rank;
    g->nnz_offsets.assign((size_t)nranks + 1, 0);
    h->shard_user = true;
    const i64 c0 = shard_col0(h->n, nranks, rank), c1 = shard_col0(h->n, nranks, rank + 1);
    if (c1 > c0) {  // after the exchange every pending column is owned: flushes and reset! work on the own range only
        const int32_t st = esp_set_column_window(h, c0 + 1, c1);
        if (st != ESP_OK) {
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
    CK(group_common(h, nranks, rank, out, &g));
    (void)hipSetDevice(h->device);
    ncclUniqueId id;
    memcpy(&id, id128, 128);
    const ncclResult_t r = g_rccl.CommInitRank(&g->nccl, nranks, id, rank);
    if (r != ncclSuccess) {
        delete g;
        FAIL(h, ESP_ERR_HIP, "ncclCommInitRank failed: %s", g_rccl.GetErrorString(r));
    }
    g->own_rccl = true;
    g->comm.ctx = g;
    g->comm.allgather_i64 = rccl_allgather_i64;
    g->comm.alltoallv_dev = rccl_alltoallv;
    *out = g;
    return ESP_OK;
}

extern "C" int32_t esp_group_create_comm(esp_handle *h, int32_t nranks, int32_t rank, const esp_comm_t *comm, esp_group **out) {
    if (!comm || !comm->allgather_i64 || !comm->alltoallv_dev) return ESP_ERR_INVALID;
    esp_group *g = nullptr;
    CK(group_common(h, nranks, rank, out, &g));
    g->comm = *comm;
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
// all_gather of a small int64 vector -> P x len (same on every rank)
static int32_t gather_ints(esp_group *g, const std::vector<i64> &mine, std::vector<i64> *all) {
    all->assign(mine.size() * (size_t)g->P, 0);
    const int32_t st = g->comm.allgather_i64(g->comm.ctx, mine.data(), (int32_t)mine.size(), all->data());
    if (st != ESP_OK) GFAIL(g, st, "esp_group: all-gather failed (%d)%s%s", st, g->err.empty() ? "" : ": ", g->err.c_str());
    return ESP_OK;
}
static int32_t exchange(esp_group *g, const std::vector<const void *> &sp, const std::vector<i64> &sb, const std::vector<void *> &rp,
                        const std::vector<i64> &rb) {
    if (g->P == 1) return ESP_OK;
    const int32_t st = g->comm.alltoallv_dev(g->comm.ctx, sp.data(), sb.data(), rp.data(), rb.data(), (void *)g->h->stream);
    if (st != ESP_OK) GFAIL(g, st, "esp_group: all-to-all-v failed (%d)%s%s", st, g->err.empty() ? "" : ": ", g->err.c_str());
    return ESP_OK;
}

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

// One partition pass per rank (owner split + first pass of the local flush: esp_shard_partition), ranges and per-digit
// counts to the owners, pieces assembled without a copy (esp_shard_assemble).  *done = false when the ranks agreed to
// use the in-place exchange for this flush (some rank's stream is not pre-sorted, or the plan does not apply).
static int32_t group_exchange_partitioned(esp_group *g, bool *done) {
    *done = false;
    esp_handle *h = g->h;
    const int P = g->P, me = g->me;
    if (g->part_skip > 0) {
        g->part_skip--;
        return ESP_OK;
    }
    std::vector<i64> all;
    if (g->eps < 0) {  // first flush: the digit width comes from the global number of entries
        CK(gather_ints(g, {h->count}, &all));
        i64 sum = 0;
        for (i64 x : all) sum += x;
        g->eps = (sum + P - 1) / P;
    }
    int32_t ok = 0;
    uint64_t *dk = nullptr;
    double *dv = nullptr;
    int64_t *dc = nullptr;
    std::vector<i64> eoff((size_t)P + 1, 0);
    int64_t nb = 0;
    CK(esp_shard_partition(h, P, me, g->eps, &ok, &dk, &dv, &dc, eoff.data(), &nb));
    if (ok) {  // (loop-back hook: the partitioned ranges -- all the rank's own -- travel through RCCL; 4-byte keys lie inside them)
        CK(loopback_roundtrip(g, dk, 8 * eoff[(size_t)P]));
        CK(loopback_roundtrip(g, dv, 8 * eoff[(size_t)P]));
    }
    std::vector<i64> mine((size_t)P + 1, 0);
    mine[0] = ok ? 1 : 0;
    for (int r = 0; r < P; r++) mine[(size_t)r + 1] = ok ? eoff[(size_t)r + 1] - eoff[(size_t)r] : 0;
    CK(gather_ints(g, mine, &all));
    const size_t W = (size_t)P + 1;
    bool all_ok = true;
    i64 total = 0;
    for (int q = 0; q < P; q++) {
        all_ok = all_ok && all[(size_t)q * W] != 0;
        for (int r = 0; r < P; r++) total += all[(size_t)q * W + 1 + (size_t)r];
    }
    if (!all_ok) {  // plain exchange now and for the next few flushes (the pending entries are intact)
        g->part_penalty = std::min(16, 2 * g->part_penalty + 1);
        g->part_skip = g->part_penalty;
        g->eps = -1;
        (void)esp_shard_plan(h, P, me, -1);
        return ESP_OK;
    }
    g->part_penalty = 0;
    g->eps = total ? (total + P - 1) / P : -1;
    // the producers of the NEXT assembly partition for the next flush's esp_shard_partition themselves (the append is
    // the partition, as on one GPU): every rank knows the same entries-per-shard from this flush's all-gather
    (void)esp_shard_plan(h, P, me, g->eps);
    std::vector<i64> in_x((size_t)P, 0), out_x((size_t)P, 0), ro((size_t)P + 1, 0);
    i64 sent = 0;
    for (int r = 0; r < P; r++) {
        in_x[(size_t)r] = r == me ? 0 : mine[(size_t)r + 1];
        out_x[(size_t)r] = r == me ? 0 : all[(size_t)r * W + 1 + (size_t)me];
        sent += in_x[(size_t)r];
        ro[(size_t)r + 1] = ro[(size_t)r] + out_x[(size_t)r];
    }
    const i64 nrecv = ro[(size_t)P];
    CK(ensure(h, g->rkeys, sizeof(u64) * (size_t)std::max<i64>(nrecv, 1)));
    CK(ensure(h, g->rvals, sizeof(double) * (size_t)std::max<i64>(nrecv, 1)));
    CK(ensure(h, g->rcnts, sizeof(i64) * (size_t)P * (size_t)std::max<i64>(nb, 1)));
    // three grouped launches -- keys, values, counts -- each with every pair's send and receive, stream-ordered behind
    // the partition's scatter kernel: no host synchronisation
    std::vector<const void *> sp((size_t)P, nullptr);
    std::vector<void *> rp((size_t)P, nullptr);
    std::vector<i64> sb((size_t)P, 0), rb((size_t)P, 0);
    for (int pass = 0; pass < 3; pass++) {
        for (int q = 0; q < P; q++) {
            if (q == me) continue;
            if (pass == 0) {
                sp[(size_t)q] = dk + eoff[(size_t)q], sb[(size_t)q] = 8 * in_x[(size_t)q];
                rp[(size_t)q] = (u64 *)g->rkeys.p + ro[(size_t)q], rb[(size_t)q] = 8 * out_x[(size_t)q];
            } else if (pass == 1) {
                sp[(size_t)q] = dv + eoff[(size_t)q], sb[(size_t)q] = 8 * in_x[(size_t)q];
                rp[(size_t)q] = (double *)g->rvals.p + ro[(size_t)q], rb[(size_t)q] = 8 * out_x[(size_t)q];
            } else {
                sp[(size_t)q] = dc + (size_t)q * (size_t)nb, sb[(size_t)q] = 8 * nb;
                rp[(size_t)q] = (i64 *)g->rcnts.p + (size_t)q * (size_t)nb, rb[(size_t)q] = 8 * nb;
            }
        }
        CK(exchange(g, sp, sb, rp, rb));
    }
    std::vector<const uint64_t *> rk((size_t)P, nullptr);
    std::vector<const double *> rv((size_t)P, nullptr);
    std::vector<const int64_t *> rc((size_t)P, nullptr);
    for (int q = 0; q < P; q++) {
        rk[(size_t)q] = (const u64 *)g->rkeys.p + ro[(size_t)q];
        rv[(size_t)q] = (const double *)g->rvals.p + ro[(size_t)q];
        rc[(size_t)q] = (const i64 *)g->rcnts.p + (size_t)q * (size_t)nb;
    }
    int32_t ok2 = 0;
    CK(esp_shard_assemble(h, rk.data(), rv.data(), rc.data(), out_x.data(), &ok2));  // (ok2 = 0: plain pending buffer instead)
    g->sent_off_rank = sent;
    g->last_exchange = 1;
    *done = true;
    return ESP_OK;
}

// any stream: stable partition by owner, the own chunk stays where it is (esp_shard_exchange_begin / _place)
static int32_t group_exchange_inplace(esp_group *g) {
    esp_handle *h = g->h;
    const int P = g->P, me = g->me;
    std::vector<i64> counts((size_t)P, 0), all;
    CK(esp_shard_counts(h, P, counts.data()));
    CK(gather_ints(g, counts, &all));
    std::vector<i64> in_x((size_t)P, 0), out_x((size_t)P, 0), ro((size_t)P + 1, 0);
    i64 lower = 0, higher = 0, sent = 0;
    for (int q = 0; q < P; q++) {
        in_x[(size_t)q] = q == me ? 0 : counts[(size_t)q];
        out_x[(size_t)q] = q == me ? 0 : all[(size_t)q * (size_t)P + (size_t)me];
        (q < me ? lower : higher) += out_x[(size_t)q];
        sent += in_x[(size_t)q];
        ro[(size_t)q + 1] = ro[(size_t)q] + out_x[(size_t)q];
    }
    uint64_t *sk = nullptr;
    double *sv = nullptr;
    std::vector<i64> soff((size_t)P + 1, 0);
    CK(esp_shard_exchange_begin(h, P, me, lower, higher, &sk, &sv, soff.data()));
    if (g->loopback) {  // (loop-back hook, one rank: the pending buffer is the own chunk)
        CK(loopback_roundtrip(g, h->keys.p, 8 * h->count));
        CK(loopback_roundtrip(g, h->vals.p, 8 * h->count));
    }
    const i64 nrecv = lower + higher;
    CK(ensure(h, g->rkeys, sizeof(u64) * (size_t)std::max<i64>(nrecv, 1)));
    CK(ensure(h, g->rvals, sizeof(double) * (size_t)std::max<i64>(nrecv, 1)));
    std::vector<const void *> sp((size_t)P, nullptr);
    std::vector<void *> rp((size_t)P, nullptr);
    std::vector<i64> sb((size_t)P, 0), rb((size_t)P, 0);
    for (int pass = 0; pass < 2; pass++) {
        for (int q = 0; q < P; q++) {
            if (q == me) continue;
            sb[(size_t)q] = 8 * in_x[(size_t)q];
            rb[(size_t)q] = 8 * out_x[(size_t)q];
            if (pass == 0) {
                sp[(size_t)q] = sk + soff[(size_t)q];
                rp[(size_t)q] = (u64 *)g->rkeys.p + ro[(size_t)q];
            } else {
                sp[(size_t)q] = sv + soff[(size_t)q];
                rp[(size_t)q] = (double *)g->rvals.p + ro[(size_t)q];
            }
        }
        CK(exchange(g, sp, sb, rp, rb));
    }
    CK(esp_shard_exchange_place(h, 0, (const u64 *)g->rkeys.p, (const double *)g->rvals.p, lower));
    CK(esp_shard_exchange_place(h, lower + counts[(size_t)me], (const u64 *)g->rkeys.p + lower, (const double *)g->rvals.p + lower, higher));
    g->sent_off_rank = sent;
    g->last_exchange = 2;
    return ESP_OK;
}

// COLLECTIVE: every rank of the group calls it (like flush! of the MT wrapper it is the synchronisation point)
extern "C" int32_t esp_group_flush(esp_group *g, int32_t mode, int64_t *local_nnz, int32_t *pattern_changed) {
    if (!g) return ESP_ERR_INVALID;
    esp_handle *h = g->h;
    (void)hipSetDevice(h->device);
    g->err.clear();
    g->loop_bytes = 0;
    bool done = false;
    CK(group_exchange_partitioned(g, &done));
    if (!done) CK(group_exchange_inplace(g));
    int64_t z = 0;
    CK(esp_flush(h, mode, &z, pattern_changed));  // (returns after the bucket kernel has read the receive buffers)
    g->local_nnz = h->nnz;
    g->offsets_valid = false;
    if (local_nnz) *local_nnz = h->nnz;
    return ESP_OK;
}

static int32_t group_offsets(esp_group *g) {
    if (g->offsets_valid) return ESP_OK;
    std::vector<i64> all;
    CK(gather_ints(g, {g->local_nnz}, &all));
    g->nnz_offsets[0] = 0;
    for (int q = 0; q < g->P; q++) g->nnz_offsets[(size_t)q + 1] = g->nnz_offsets[(size_t)q] + all[(size_t)q];
    g->offsets_valid = true;
    return ESP_OK;
}

// COLLECTIVE on first use after a flush (one all-gather of the local nnz)
extern "C" int32_t esp_group_nnz(esp_group *g, int64_t *global_nnz, int64_t *nnz_before_me) {
    if (!g) return ESP_ERR_INVALID;
    CK(group_offsets(g));
    if (global_nnz) *global_nnz = g->nnz_offsets[(size_t)g->P];
    if (nnz_before_me) *nnz_before_me = g->nnz_offsets[(size_t)g->me];
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
    const i64 off = g->nnz_offsets[(size_t)g->me];
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
    if (kind) *kind = g->last_exchange;
    if (sent_off_rank) *sent_off_rank = g->sent_off_rank;
    return ESP_OK;
}
