// handle.hip -- libesparse_hip: lifetime, buffers, append, CSC in and out, timing, debug getters (see internal.hpp for the map of the translation units)
#include "internal.hpp"

#include <pthread.h>
#include <sched.h>
#include <sys/mman.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>

thread_local std::string g_err;

int32_t ensure(esp_handle *h, DevBuf &b, size_t need, bool keep) {
    if (b.bytes >= need && b.p) return ESP_OK;
    size_t want = need;
    if (keep && b.bytes) want = std::max(need, b.bytes + b.bytes / 2);
    want = (want + 255) & ~(size_t)255;
    void *np = nullptr;
    if (!(keep && b.p && b.bytes)) {
        // nothing to carry over: the old allocation goes first (a 256^3 buffer that is replaced never needs old + new at once)
        release(b);
        HIPCK(h, hipMalloc(&np, want));
        b.p = np;
        b.bytes = want;
        return ESP_OK;
    }
    HIPCK(h, hipMalloc(&np, want));
    hipError_t e = hipMemcpyAsync(np, b.p, b.bytes, hipMemcpyDeviceToDevice, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) {  // (the buffer stays as it was; the new allocation does not leak)
        (void)hipFree(np);
        FAIL(h, ESP_ERR_HIP, "growing a device buffer failed: %s", hipGetErrorString(e));
    }
    {
        DevBuf old = b;  // (through release: the experiments build may keep it poisoned)
        release(old);
    }
    b.p = np;
    b.bytes = want;
    return ESP_OK;
}
#ifdef ESP_EXPERIMENTS
// ESP_POISON_FREE (experiments build): a released buffer is not freed but filled with 0xA5 and kept -- whoever still READS it
// meets garbage at once (the parity fuzz then fails deterministically, single-threaded), and esp_exp_check_graveyard (called by
// esp_destroy) finds whoever WROTE to it afterwards.  The hunt for use-after-release, GPU AddressSanitizer not being available.
namespace {
struct Grave { void *p; size_t bytes; };
std::vector<Grave> g_graves;
size_t g_grave_bytes = 0;  // (above 64 GB the oldest are checked one last time and really freed)
std::mutex g_graves_m;
__global__ void grave_check_k(const unsigned long long *p, size_t words, unsigned long long *bad) {
    size_t n = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x)
        n += p[i] != 0xA5A5A5A5A5A5A5A5ull;
    if (n) atomicAdd(bad, (unsigned long long)n);
}
bool poison_release(DevBuf &b) {
    if (!esp_exp_env("ESP_POISON_FREE")) return false;
    (void)hipDeviceSynchronize();
    (void)hipMemset(b.p, 0xA5, b.bytes);
    (void)hipDeviceSynchronize();
    std::lock_guard<std::mutex> lk(g_graves_m);
    g_graves.push_back({b.p, b.bytes});
    g_grave_bytes += b.bytes;
    return true;
}
}  // namespace
void esp_exp_check_graveyard() {
    if (!esp_exp_env("ESP_POISON_FREE")) return;
    std::lock_guard<std::mutex> lk(g_graves_m);
    (void)hipDeviceSynchronize();
    unsigned long long *d_bad = nullptr;
    if (hipMalloc((void **)&d_bad, 8) != hipSuccess) return;
    size_t total = 0;
    for (const Grave &g : g_graves) {
        (void)hipMemset(d_bad, 0, 8);
        hipLaunchKernelGGL(grave_check_k, dim3(1024), dim3(256), 0, 0, (const unsigned long long *)g.p, g.bytes / 8, d_bad);
        unsigned long long bad = 0;
        (void)hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost);
        if (bad) {
            fprintf(stderr, "POISON: %llu words of a released buffer (%p, %zu bytes) were written after its release\n", bad, g.p, g.bytes);
            (void)hipMemset(g.p, 0xA5, g.bytes);
        }
        total += g.bytes;
    }
    (void)hipFree(d_bad);
    size_t drop = 0;
    while (g_grave_bytes > ((size_t)64 << 30) && drop < g_graves.size()) {
        (void)hipFree(g_graves[drop].p);
        g_grave_bytes -= g_graves[drop].bytes;
        drop++;
    }
    g_graves.erase(g_graves.begin(), g_graves.begin() + (long)drop);
    if (esp_exp_env("ESP_POISON_VERBOSE")) fprintf(stderr, "POISON: %zu released buffers, %.1f MB checked\n", g_graves.size(), (double)total / 1e6);
}
#endif
void release(DevBuf &b) {
#ifdef ESP_EXPERIMENTS
    if (b.p && poison_release(b)) {
        b.p = nullptr;
        b.bytes = 0;
        return;
    }
#endif
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.bytes = 0;
}

void release_all(esp_handle *h) {
    for (DevBuf *b : {&h->keys, &h->vals, &h->keys2, &h->vals2, &h->hist, &h->segs, &h->colend, &h->newkey,
                      &h->newval, &h->heads, &h->misc, &h->colptr, &h->colptr2, &h->rowval, &h->nzval, &h->rowval2,
                      &h->nzval2, &h->seg[0], &h->seg[1], &h->tilef[0], &h->tilef[1], &h->segcnt, &h->segout, &h->tseg, &h->ttile, &h->runbuf, &h->chunkbuf, &h->parttab, &h->piecetab, &h->asmwork, &h->sumrange, &h->csr_rowptr, &h->csr_perm, &h->csr_col, &h->csr_tmp, &h->csr_val, &h->mul_x, &h->mul_r, &h->lazy_hold, &h->elemplan.sorted, &h->elemplan.cellrec, &h->elemplan.segtab, &h->stage.d_rows, &h->stage.d_cols, &h->stage.d_vals, &h->stage.d_kinds, &h->bulk.d_rows, &h->bulk.d_cols, &h->bulk.d_vals, &h->bulk.d_kinds})
        release(*b);
    for (esp_handle::StageArea *sa : {&h->stage, &h->bulk}) {
        if (sa->rows) (void)hipHostFree(sa->rows);
        if (sa->cols) (void)hipHostFree(sa->cols);
        if (sa->vals) (void)hipHostFree(sa->vals);
        if (sa->kinds) (void)hipHostFree(sa->kinds);
        sa->rows = sa->cols = nullptr;
        sa->vals = nullptr;
        sa->kinds = nullptr;
        sa->cap = 0;
    }
    for (int i = 0; i < 2; i++) {
        if (h->bounce.pin[i]) (void)hipHostFree(h->bounce.pin[i]);
        if (h->bounce.ev[i]) (void)hipEventDestroy(h->bounce.ev[i]);
        h->bounce.pin[i] = nullptr, h->bounce.ev[i] = nullptr;
    }
    h->bounce.bytes = 0;
    for (int i = 0; i < 2; i++) {
        if (h->cpack.keys[i]) (void)hipHostFree(h->cpack.keys[i]);
        if (h->cpack.vals[i]) (void)hipHostFree(h->cpack.vals[i]);
        if (h->cpack.done[i]) (void)hipEventDestroy(h->cpack.done[i]);
        h->cpack.keys[i] = nullptr, h->cpack.vals[i] = nullptr, h->cpack.done[i] = nullptr, h->cpack.busy[i] = false;
    }
    h->cpack.cap = 0;
}

// ------------------------------------------------------------------------ timing
hipEvent_t ev_get(esp_handle *h) {
    if (!h->ev_pool.empty()) {
        hipEvent_t e = h->ev_pool.back();
        h->ev_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
void timing_collect(esp_handle *h) {
    if (h->spans.empty()) return;
    (void)hipStreamSynchronize(h->stream);
    for (auto &s : h->spans) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
            if (s.stage >= 0) {
                h->acc.ms[s.stage] += ms;
                h->acc.launches[s.stage] += s.launches;
            } else {
                h->acc.flush_ms += ms;
                h->acc.flushes += 1;
            }
        }
        h->ev_pool.push_back(s.a);
        h->ev_pool.push_back(s.b);
    }
    h->spans.clear();
}

// ------------------------------------------------------------------------ lifetime
extern "C" const char *esp_version(void) { return "esparse-hip 0.1 (gfx950)"; }

extern "C" const char *esp_last_error(const esp_handle *h) { return h ? h->err.c_str() : g_err.c_str(); }

// colptr behind the window := nnz+1, if it was left stale by windowed flushes
int32_t fix_tail(esp_handle *h) {
    if (h->ones_pending) {
        h->ones_pending = false;
        h->tail_stale = false;
        if (h->colptr.p)
            hipLaunchKernelGGL(fill_i64_k, dim3(grid_for(h->n + 1, 256)), dim3(256), 0, h->stream, (i64 *)h->colptr.p, h->n + 1, (i64)1);
        HIPCK(h, hipGetLastError());
        return ESP_OK;
    }
    if (!h->tail_stale) return ESP_OK;
    h->tail_stale = false;
    const i64 from = h->wc1 + 1, cnt = h->n + 1 - from;
    if (cnt > 0 && h->colptr.p)
        hipLaunchKernelGGL(fill_i64_k, dim3(grid_for(cnt, 256)), dim3(256), 0, h->stream, (i64 *)h->colptr.p + from, cnt, h->nnz + 1);
    HIPCK(h, hipGetLastError());
    return ESP_OK;
}

int32_t init_empty_csc(esp_handle *h) {
    CK(ensure(h, h->colptr, sizeof(i64) * (size_t)(h->n + 1)));
    if (h->csc_valid && windowed(h)) {
        // every entry was inside the window: colptr is 1 up to it already, the part behind it is refreshed lazily
        i64 c0, cnt;
        col_range(h, &c0, &cnt);
        hipLaunchKernelGGL(fill_i64_k, dim3(grid_for(cnt, 256)), dim3(256), 0, h->stream, (i64 *)h->colptr.p + c0, cnt, (i64)1);
        h->tail_stale = h->wc1 < h->n;
    } else if (h->csc_valid && !windowed(h) && h->n > 4096) {
        h->ones_pending = true;  // (filled on first use unless a fresh flush writes all of colptr before)
        h->tail_stale = false;
    } else {
        hipLaunchKernelGGL(fill_i64_k, dim3(grid_for(h->n + 1, 256)), dim3(256), 0, h->stream, (i64 *)h->colptr.p,
                           h->n + 1, (i64)1);
        h->tail_stale = false;
        h->ones_pending = false;
    }
    h->nnz = 0;
    h->pattern_version++, h->values_version++;
    h->csc_valid = true;
    return ESP_OK;
}

extern "C" int32_t esp_create(int64_t m, int64_t n, int32_t device, int64_t capacity_hint, esp_handle **out) {
    if (!out) FAIL((esp_handle *)nullptr, ESP_ERR_INVALID, "esp_create: out is NULL");
    *out = nullptr;
    if (m < 0 || n < 0) FAIL((esp_handle *)nullptr, ESP_ERR_INVALID, "esp_create: negative dimension");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        FAIL((esp_handle *)nullptr, ESP_ERR_NODEVICE, "esp_create: no HIP device available (libesparse_hip has no CPU path)");
    if (device < 0 || device >= ndev)
        FAIL((esp_handle *)nullptr, ESP_ERR_INVALID, "esp_create: device %d out of range (0..%d)", device, ndev - 1);
    const int rb = bits_for(m), cb = bits_for(n);
    if (rb + cb + ESP_TAG_BITS > 64)
        FAIL((esp_handle *)nullptr, ESP_ERR_UNSUPPORTED, "esp_create: %lld x %lld needs %d key bits (max 62)",
             (long long)m, (long long)n, rb + cb);
    esp_handle *h = new esp_handle();
    h->m = m;
    h->n = n;
    h->device = device;
    h->L = KeyLayout{rb, cb};
    h->win_base = 0;
    h->win_span = (u64)std::max<i64>(n, 1) << rb;
    h->wc0 = 0;
    h->wc1 = n;
    h->hint = capacity_hint > 0 ? capacity_hint : 0;
    memset(&h->acc, 0, sizeof h->acc);
    if (const char *e = getenv("ESP_DEBUG_FORCE_PATH")) h->force_path = atoi(e);  // (same-box A/B of the test hooks)
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
        delete h;
        FAIL((esp_handle *)nullptr, ESP_ERR_HIP, "esp_create: cannot create a stream on device %d", device);
    }
    h->own_stream = true;
    if (hipHostMalloc((void **)&h->pin_scalar, 64, hipHostMallocDefault) != hipSuccess) {
        (void)hipStreamDestroy(h->stream);
        delete h;
        FAIL((esp_handle *)nullptr, ESP_ERR_NOMEM, "esp_create: pinned scalar allocation failed");
    }
    int32_t st = init_empty_csc(h);
    if (st == ESP_OK && capacity_hint > 0) {
        st = ensure(h, h->keys, sizeof(u64) * (size_t)capacity_hint);
        if (st == ESP_OK) st = ensure(h, h->vals, sizeof(double) * (size_t)capacity_hint);
        if (st == ESP_OK) h->cap = capacity_hint;
    }
    if (st != ESP_OK) {
        g_err = h->err;
        esp_destroy(h);
        return st;
    }
    *out = h;
    return ESP_OK;
}

extern "C" int32_t esp_destroy(esp_handle *h) {
    if (!h) return ESP_OK;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    if (h->sumtmp) {
        (void)esp_destroy(h->sumtmp);
        h->sumtmp = nullptr;
    }
    release_all(h);
#ifdef ESP_EXPERIMENTS
    esp_exp_check_graveyard();
#endif
    if (h->pin_scalar) (void)hipHostFree(h->pin_scalar);
    if (h->pin_mw) (void)hipHostFree(h->pin_mw);
    if (h->pin_asm) (void)hipHostFree(h->pin_asm);
    if (h->pin_mw_done) (void)hipEventDestroy(h->pin_mw_done);
    for (auto &s : h->spans) {
        (void)hipEventDestroy(s.a);
        (void)hipEventDestroy(s.b);
    }
    for (auto e : h->ev_pool) (void)hipEventDestroy(e);
    if (h->aux_ev) (void)hipEventDestroy(h->aux_ev);
    if (h->aux) (void)hipStreamDestroy(h->aux);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return ESP_OK;
}

// The Generic wrappers of the reference replace their buffer by a fresh T_ext(m,n) after every flush!
// (genericextendablesparsematrixcsc.jl:34, genericmt...:47-49) and leave the old one to the garbage collector, which
// does not see device or pinned memory: the shim calls this on the old buffer right after `buffer + csc` returned.
// The handle stays valid (an empty matrix with an empty buffer; the staging chunk pointers of esp_stage_begin are
// gone); everything is allocated again on next use.
extern "C" int32_t esp_release_buffers(esp_handle *h) {
    if (!h) return ESP_ERR_INVALID;
    (void)hipSetDevice(h->device);
    HIPCK(h, hipStreamSynchronize(h->stream));
    timing_collect(h);
    if (h->sumtmp) {  // (the scratch matrix of esp_flush_sum's batched folds)
        (void)esp_destroy(h->sumtmp);
        h->sumtmp = nullptr;
    }
    release_all(h);
    h->cap = 0;
    h->count = 0;
    h->chunk_cap = 0;
    h->chunk_pb = 0;
    h->rawplan.valid = false;
    h->genplan.valid = false;
    h->elemplan.valid = false;
    h->shard_offsets.plan_id = 0;  // (the counts it vouches for lived in parttab)
    pending_changed(h);
    h->csc_valid = false;
    h->ones_pending = false;
    h->tail_stale = false;
    h->csr_version = 0;
    h->csr_val_version = 0;
    return init_empty_csc(h);
}

// Base.copy(ext) (extendable.jl:279-285): a second handle with the same CSC, the same pending entries
// (the copy of lnkmatrix) and the same column window; device-to-device copies only.
extern "C" int32_t esp_clone(esp_handle *h, esp_handle **out) {
    if (!h || !out) return ESP_ERR_INVALID;
    *out = nullptr;
    (void)hipSetDevice(h->device);
    esp_handle *c = nullptr;
    CK(esp_create(h->m, h->n, h->device, h->hint, &c));
    auto fail = [&](int32_t st) {
        h->err = c->err;
        esp_destroy(c);
        return st;
    };
    int32_t st = ESP_OK;
    if ((st = pending_materialize(h)) != ESP_OK) return fail(st);
    HIPCK(h, hipStreamSynchronize(h->stream));  // everything the copy reads is complete
    if (h->count > 0) {
        if ((st = reserve_append(c, h->count)) != ESP_OK) return fail(st);
        if (hipMemcpyAsync(c->keys.p, h->keys.p, sizeof(u64) * (size_t)h->count, hipMemcpyDeviceToDevice, c->stream) != hipSuccess ||
            hipMemcpyAsync(c->vals.p, h->vals.p, sizeof(double) * (size_t)h->count, hipMemcpyDeviceToDevice, c->stream) != hipSuccess)
            return fail(ESP_ERR_HIP);
        c->count = h->count;
    }
    if ((st = fix_tail(h)) != ESP_OK) return fail(st);
    HIPCK(h, hipStreamSynchronize(h->stream));
    if (hipMemcpyAsync(c->colptr.p, h->colptr.p, sizeof(i64) * (size_t)(h->n + 1), hipMemcpyDeviceToDevice, c->stream) != hipSuccess) return fail(ESP_ERR_HIP);
    if (h->nnz > 0) {
        if ((st = ensure(c, c->rowval, sizeof(i64) * (size_t)h->nnz)) != ESP_OK) return fail(st);
        if ((st = ensure(c, c->nzval, sizeof(double) * (size_t)h->nnz)) != ESP_OK) return fail(st);
        if (hipMemcpyAsync(c->rowval.p, h->rowval.p, sizeof(i64) * (size_t)h->nnz, hipMemcpyDeviceToDevice, c->stream) != hipSuccess ||
            hipMemcpyAsync(c->nzval.p, h->nzval.p, sizeof(double) * (size_t)h->nnz, hipMemcpyDeviceToDevice, c->stream) != hipSuccess)
            return fail(ESP_ERR_HIP);
    }
    c->nnz = h->nnz;
    c->pattern_version++, c->values_version++;
    c->win_base = h->win_base;
    c->win_span = h->win_span;
    c->wc0 = h->wc0;
    c->wc1 = h->wc1;
    c->win_excl = h->win_excl;
    c->seen_maxrun = h->seen_maxrun;
    if (hipStreamSynchronize(c->stream) != hipSuccess) return fail(ESP_ERR_HIP);
    *out = c;
    return ESP_OK;
}

extern "C" int32_t esp_set_stream(esp_handle *h, void *hip_stream) {
    if (!h) return ESP_ERR_INVALID;
    (void)hipStreamSynchronize(h->stream);
    timing_collect(h);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    h->stream = (hipStream_t)hip_stream;
    h->own_stream = false;
    return ESP_OK;
}
extern "C" int32_t esp_synchronize(esp_handle *h) {
    if (!h) return ESP_ERR_INVALID;
    HIPCK(h, hipStreamSynchronize(h->stream));
    return ESP_OK;
}
extern "C" int32_t esp_size(const esp_handle *h, int64_t *m, int64_t *n) {
    if (!h) return ESP_ERR_INVALID;
    if (m) *m = h->m;
    if (n) *n = h->n;
    return ESP_OK;
}
extern "C" int32_t esp_key_layout(const esp_handle *h, int32_t *row_bits, int32_t *col_bits) {
    if (!h) return ESP_ERR_INVALID;
    if (row_bits) *row_bits = h->L.rb;
    if (col_bits) *col_bits = h->L.cb;
    return ESP_OK;
}
extern "C" int32_t esp_pending(const esp_handle *h, int64_t *count) {
    if (!h || !count) return ESP_ERR_INVALID;
    *count = h->count;
    return ESP_OK;
}
extern "C" int32_t esp_nnz(const esp_handle *h, int64_t *nnz) {
    if (!h || !nnz) return ESP_ERR_INVALID;
    *nnz = h->nnz;
    return ESP_OK;
}

// ------------------------------------------------------------------------ append
int32_t reserve_append(esp_handle *h, i64 add) {
    // between esp_shard_assemble and esp_flush the pending entries are spread over the caller's receive buffers
    // (the handle's count is their logical total): nothing can be appended behind them
    if (h->part_assembled)
        FAIL(h, ESP_ERR_STATE, "append: the pending entries are assembled shard pieces (esp_shard_assemble): esp_flush first");
    // An append behind a bucket-ordered batch: the batch stays as it is (its 4-byte keys fill the front half of their
    // slots), the new entries follow it as packed keys, and the flush partitions only them (flush_pre_tail).  A shard's
    // batch, or force_path 19: back to packed keys first.
    CK(lazy_expand(h));  // (a batch still held as sorted items: whatever comes behind it needs its updates where they belong)
    h->pre_keep = h->pre.valid && h->pre.mw_P == 0 && h->force_path != ESP_PATH_NO_BATCH_TAIL && h->count == h->pre.E + h->pre.tail;
    if (!h->pre_keep) CK(pending_materialize(h));
    const i64 need = h->count + add;
    if (need <= h->cap) return ESP_OK;
    i64 ncap = std::max<i64>(need, h->cap + h->cap / 2);
    ncap = std::max<i64>(ncap, 1024);
    // keep contents: only the first count entries matter
    DevBuf nk, nv;
    CK(ensure(h, nk, sizeof(u64) * (size_t)ncap));
    CK(ensure(h, nv, sizeof(double) * (size_t)ncap));
    if (h->count > 0) {
        HIPCK(h, hipMemcpyAsync(nk.p, h->keys.p, sizeof(u64) * (size_t)h->count, hipMemcpyDeviceToDevice, h->stream));
        HIPCK(h, hipMemcpyAsync(nv.p, h->vals.p, sizeof(double) * (size_t)h->count, hipMemcpyDeviceToDevice, h->stream));
        HIPCK(h, hipStreamSynchronize(h->stream));
    }
    release(h->keys);
    release(h->vals);
    h->keys = nk;
    h->vals = nv;
    h->cap = ncap;
    return ESP_OK;
}

// pack count triples that already sit in device memory; checks bounds before committing
int32_t pack_device(esp_handle *h, const i64 *d_rows, const i64 *d_cols, const double *d_vals,
                           const uint8_t *d_kinds, int kind_all, int op, i64 count) {
    if (count == 0) return ESP_OK;
    if (!d_kinds && (kind_all < 0 || kind_all > 3)) FAIL(h, ESP_ERR_INVALID, "append: kind %d invalid", kind_all);
    if (op != ESP_OP_ADD && op != ESP_OP_SUB) FAIL(h, ESP_ERR_UNSUPPORTED, "append: op %d not supported on the device path", op);
    if (!d_kinds) {  // an empty buffer and one kind: the append is the partition (no packed stream is written)
        bool took = false;
        CK(append_partitioned(h, d_rows, d_cols, d_vals, kind_all, op, count, &took));
        if (took) return ESP_OK;
        CK(append_tail_partitioned(h, d_rows, d_cols, d_vals, kind_all, op, count, &took));  // (behind a batch over a stored pattern)
        if (took) return ESP_OK;
        CK(append_first_pass(h, d_rows, d_cols, d_vals, kind_all, op, count, &took));  // (a shuffled stream: the first radix pass of its flush, from the caller's arrays)
        if (took) return ESP_OK;
    }
    CK(reserve_append(h, count));
    h->pin_scalar[0] = ~0ull;
    CK(ensure(h, h->misc, 256));
    unsigned long long *d_err = (unsigned long long *)h->misc.p;
    HIPCK(h, hipMemcpyAsync(d_err, h->pin_scalar, 8, hipMemcpyHostToDevice, h->stream));
    {
        Span sp(h, ESP_ST_APPEND);
        hipLaunchKernelGGL(espgen::pack_k, dim3(grid_for(count, espgen::THREADS)), dim3(espgen::THREADS), 0, h->stream,
                           d_rows, d_cols, d_vals, d_kinds, kind_all, op == ESP_OP_SUB ? 1 : 0, count, h->m, h->n, h->L,
                           (u64 *)h->keys.p + h->count, (double *)h->vals.p + h->count, d_err, (i64)0);
        sp.add(1);
    }
    HIPCK(h, hipMemcpyAsync(h->pin_scalar, d_err, 8, hipMemcpyDeviceToHost, h->stream));
    HIPCK(h, hipStreamSynchronize(h->stream));
    if (h->pin_scalar[0] != ~0ull) {
        h->pre_keep = false;  // (nothing was appended behind the batch after all)
        FAIL(h, ESP_ERR_BOUNDS, "BoundsError: entry %llu of the batch has an index outside %lld x %lld (or a bad kind)",
             (unsigned long long)h->pin_scalar[0], (long long)h->m, (long long)h->n);
    }
    if (!d_kinds) note_kind(h, kind_all, count);
    h->count += count;
    pending_changed(h);
    return ESP_OK;
}

int32_t ensure_stage(esp_handle *h, esp_handle::StageArea &sa, i64 want) {
    if (want <= sa.cap) return ESP_OK;
    i64 cap = std::max<i64>(want, 1 << 16);
    HIPCK(h, hipStreamSynchronize(h->stream));  // (a transfer out of the old area may be in flight)
    if (sa.rows) (void)hipHostFree(sa.rows);
    if (sa.cols) (void)hipHostFree(sa.cols);
    if (sa.vals) (void)hipHostFree(sa.vals);
    if (sa.kinds) (void)hipHostFree(sa.kinds);
    sa.rows = sa.cols = nullptr;
    sa.vals = nullptr;
    sa.kinds = nullptr;
    sa.cap = 0;
    HIPCK(h, hipHostMalloc((void **)&sa.rows, sizeof(i64) * (size_t)cap, hipHostMallocDefault));
    HIPCK(h, hipHostMalloc((void **)&sa.cols, sizeof(i64) * (size_t)cap, hipHostMallocDefault));
    HIPCK(h, hipHostMalloc((void **)&sa.vals, sizeof(double) * (size_t)cap, hipHostMallocDefault));
    HIPCK(h, hipHostMalloc((void **)&sa.kinds, (size_t)cap, hipHostMallocDefault));
    CK(ensure(h, sa.d_rows, sizeof(i64) * (size_t)cap));
    CK(ensure(h, sa.d_cols, sizeof(i64) * (size_t)cap));
    CK(ensure(h, sa.d_vals, sizeof(double) * (size_t)cap));
    CK(ensure(h, sa.d_kinds, (size_t)cap));
    sa.cap = cap;
    return ESP_OK;
}

extern "C" int32_t esp_stage_begin(esp_handle *h, int64_t want, int64_t **rows, int64_t **cols, double **vals,
                                   uint8_t **kinds, int64_t *got) {
    if (!h || !rows || !cols || !vals || !got) return ESP_ERR_INVALID;
    (void)hipSetDevice(h->device);
    if (want <= 0) want = 1 << 20;
    want = std::min<i64>(want, (i64)1 << 26);
    CK(ensure_stage(h, h->stage, want));
    *rows = h->stage.rows;
    *cols = h->stage.cols;
    *vals = h->stage.vals;
    if (kinds) *kinds = h->stage.kinds;
    *got = h->stage.cap;
    return ESP_OK;
}

// the staged chunk as device triplets (large chunks of one kind: the append may be the partition, pack_device)
static int32_t commit_as_triplets(esp_handle *h, int64_t count, int32_t kind_all, int32_t op) {
    {
        Span sp(h, ESP_ST_COPY);
        HIPCK(h, hipMemcpyAsync(h->stage.d_rows.p, h->stage.rows, sizeof(i64) * (size_t)count, hipMemcpyHostToDevice, h->stream));
        HIPCK(h, hipMemcpyAsync(h->stage.d_cols.p, h->stage.cols, sizeof(i64) * (size_t)count, hipMemcpyHostToDevice, h->stream));
        HIPCK(h, hipMemcpyAsync(h->stage.d_vals.p, h->stage.vals, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, h->stream));
        sp.add(3);
        if (kind_all < 0) {
            HIPCK(h, hipMemcpyAsync(h->stage.d_kinds.p, h->stage.kinds, (size_t)count, hipMemcpyHostToDevice, h->stream));
            sp.add(1);
        }
    }
    return pack_device(h, (const i64 *)h->stage.d_rows.p, (const i64 *)h->stage.d_cols.p, (const double *)h->stage.d_vals.p,
                       kind_all < 0 ? (const uint8_t *)h->stage.d_kinds.p : nullptr, kind_all, op, count);
}

// A small pool of host threads for the copies and conversions of the host boundary (one core moves 10-20 GB/s, PCIe takes
// ~55 GB/s and the conversions touch 2-3 bytes per byte moved): created on first use, shared by all handles; a call that
// finds the pool busy (another handle's host thread) runs its parts itself.
namespace {
// the CPUs of the calling thread's NUMA node (from /sys/devices/system/node/node*/cpulist), cut down to what the process may
// use; false: unknown (no sysfs, one node only, ...): no binding
bool numa_cpus_of_current_thread(cpu_set_t *out) {
    const int cpu = sched_getcpu();
    if (cpu < 0) return false;
    cpu_set_t allowed;
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return false;
    int nodes = 0;
    bool found = false;
    for (int node = 0; node < 64; node++) {
        char path[96];
        snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
        FILE *f = fopen(path, "r");
        if (!f) continue;
        nodes++;
        char buf[4096];
        const size_t len = fread(buf, 1, sizeof buf - 1, f);
        fclose(f);
        buf[len] = 0;
        cpu_set_t set;
        CPU_ZERO(&set);
        bool mine = false;
        for (char *p = buf; *p;) {  // "0-63,128-191"
            char *end = nullptr;
            const long a = strtol(p, &end, 10);
            if (end == p) break;
            long b = a;
            p = end;
            if (*p == '-') {
                b = strtol(p + 1, &end, 10);
                p = end;
            }
            for (long c = a; c <= b && c < CPU_SETSIZE; c++) {
                if (CPU_ISSET((int)c, &allowed)) CPU_SET((int)c, &set);
                mine = mine || c == cpu;
            }
            while (*p == ',' || *p == '\n' || *p == ' ') p++;
        }
        if (mine && CPU_COUNT(&set) > 0) {
            *out = set;
            found = true;
        }
    }
    return found && nodes > 1;
}
// ESP_HOST_THREADS = the host threads esp_append_host / esp_get_csc may use beside the caller's (documented in the header: the
// one environment setting of the product path besides ESP_RCCL_LIB; read once)
static int host_threads_env() {
    static const int v = [] {
        const char *e = getenv("ESP_HOST_THREADS");
        return e ? std::max(1, atoi(e)) : 0;
    }();
    return v;
}
struct HostPool {
    std::mutex m, busy;
    std::condition_variable cv_go, cv_done;
    std::vector<std::thread> th;
    const std::function<void(int)> *job = nullptr;
    int nparts = 0, next = 0, pending = 0;
    unsigned long long epoch = 0;
    bool stop = false;
    int size() const { return (int)th.size(); }
    void start() {
        const unsigned hw = std::thread::hardware_concurrency();
        unsigned want = 8u;  // (measured on the 2-socket EPYC of the GPU box: 4 and 8 threads pack at the rate PCIe takes, 16 are slower)
        if (host_threads_env() > 0) want = (unsigned)host_threads_env();
        const int n = (int)std::max(1u, std::min(want, hw > 2 ? hw - 1 : 1u)) - 1 > 0 ? (int)std::max(1u, std::min(want, hw > 2 ? hw - 1 : 1u)) - 1 : 1;  // (the caller works too)
        // the pool stays on the NUMA node of the thread that first needs it: the caller's arrays were most likely touched
        // there (first touch), and a two-socket host moves remote pages at a fraction of the local rate (append of the 256^3
        // stream: 61 ms with the threads beside the data, up to 91 ms with them anywhere)
        cpu_set_t node_set;
        const bool bind = numa_cpus_of_current_thread(&node_set);
        for (int i = 0; i < n; i++)
            th.emplace_back([this, bind, node_set] {
                if (bind) (void)pthread_setaffinity_np(pthread_self(), sizeof(cpu_set_t), &node_set);
                unsigned long long seen = 0;
                std::unique_lock<std::mutex> lk(m);
                for (;;) {
                    cv_go.wait(lk, [&] { return stop || (epoch != seen && next < nparts); });
                    if (stop) return;
                    while (next < nparts) {
                        const int part = next++;
                        const std::function<void(int)> *f = job;
                        lk.unlock();
                        (*f)(part);
                        lk.lock();
                        if (--pending == 0) cv_done.notify_all();
                    }
                    seen = epoch;
                }
            });
    }
    ~HostPool() {
        {
            std::lock_guard<std::mutex> lk(m);
            stop = true;
        }
        cv_go.notify_all();
        for (auto &t : th) t.join();
    }
    // fn(part) for part = 0 .. parts-1, the caller takes parts too; returns when all are done
    void run(int parts, const std::function<void(int)> &fn) {
        if (parts <= 1 || !busy.try_lock()) {
            for (int p = 0; p < parts; p++) fn(p);
            return;
        }
        {
            std::lock_guard<std::mutex> lk(m);
            if (th.empty()) start();
            job = &fn, nparts = parts, next = 0, pending = parts;
            epoch++;
        }
        cv_go.notify_all();
        {
            std::unique_lock<std::mutex> lk(m);
            while (next < nparts) {
                const int part = next++;
                lk.unlock();
                fn(part);
                lk.lock();
                --pending;
            }
            cv_done.wait(lk, [&] { return pending == 0; });
            job = nullptr;
        }
        busy.unlock();
    }
};
HostPool &host_pool() {
    static HostPool p;
    return p;
}
// [0, count) cut into parts of at least min_part elements, at most `cap` parts
template <typename F>
void host_parallel(size_t count, size_t min_part, int cap, F body) {
    static const int env_cap = host_threads_env() > 0 ? host_threads_env() + 1 : 1 << 20;
    cap = std::min(cap, env_cap);
    const int parts = (int)std::max<size_t>(1, std::min<size_t>((size_t)cap, count / std::max<size_t>(min_part, 1)));
    const size_t part = (count + (size_t)parts - 1) / (size_t)parts;
    const std::function<void(int)> fn = [&](int p) { body(std::min(count, (size_t)p * part), std::min(count, ((size_t)p + 1) * part)); };
    host_pool().run(parts, fn);
}
}  // namespace

// fn(part) for part = 0 .. parts-1 on the host pool's threads and the caller's; returns when all are done (one job at a time: a
// second caller runs its parts itself)
void host_run_parts(int parts, const std::function<void(int)> &fn) { host_pool().run(parts, fn); }

void par_memcpy(void *dst, const void *src, size_t bytes) {
    host_parallel(bytes, (size_t)2 << 20, 8, [=](size_t a, size_t b) {
        const size_t a64 = a & ~(size_t)63, b64 = b == bytes ? b : (b & ~(size_t)63);  // (parts start on 64-byte boundaries)
        if (b64 > a64) memcpy((char *)dst + a64, (const char *)src + a64, b64 - a64);
    });
}

// Packing on the host: (row, col[, kind]) -> the packed key of the handle's layout, values copied (negated for op = SUB
// unless the kind is SET), with a few threads; *bad = smallest index (relative to index_base) of an entry outside m x n
// or of a bad kind (BoundsError), else left alone.  The boundary then moves 16 bytes per entry over PCIe instead of 24
// (25 with kinds), straight into the append buffer: no staging arrays on the device, no pack kernel.
// (one block of host_pack: branch-free, compiled for AVX-512 / AVX2 as well and picked at load time -- the library itself is
// built for the baseline x86-64)
#if defined(__HIP_DEVICE_COMPILE__)
#define ESP_HOST_SIMD  // (the device pass of hipcc parses host functions too and knows no x86 function multiversioning)
#else
#define ESP_HOST_SIMD __attribute__((target_clones("avx512f", "avx2", "default")))
#endif
// (function multiversioning does not take templates: one pair of overloads per index type)
#define ESP_PACK_BLOCKS(TI)                                                                                                                          \
    ESP_HOST_SIMD static u64 pack_block_one(const TI *rows, const TI *cols, const double *vals, u64 kind, bool neg, i64 cnt, u64 um, u64 un, int rb,  \
                                            u64 *keys_out, double *vals_out) {                                                                      \
        u64 anybad = 0;                                                                                                                              \
        for (i64 i = 0; i < cnt; i++) {                                                                                                              \
            const u64 r0 = (u64)(i64)rows[i] - 1ull, c0 = (u64)(i64)cols[i] - 1ull;                                                                  \
            anybad |= (u64)(r0 >= um) | (u64)(c0 >= un);                                                                                             \
            keys_out[i] = (((c0 << rb) | r0) << ESP_TAG_BITS) | kind;                                                                                \
        }                                                                                                                                            \
        if (neg)                                                                                                                                     \
            for (i64 i = 0; i < cnt; i++) vals_out[i] = -vals[i];                                                                                    \
        else                                                                                                                                         \
            memcpy(vals_out, vals, sizeof(double) * (size_t)cnt);                                                                                    \
        return anybad;                                                                                                                               \
    }                                                                                                                                                \
    ESP_HOST_SIMD static u64 pack_block_six(const TI *rows, const TI *cols, const double *vals, bool neg, i64 cnt, u64 um, u64 un, int rb,           \
                                            uint32_t *lo_out, uint16_t *hi_out, double *vals_out) {                                                 \
        u64 anybad = 0;                                                                                                                              \
        for (i64 i = 0; i < cnt; i++) {                                                                                                              \
            const u64 r0 = (u64)(i64)rows[i] - 1ull, c0 = (u64)(i64)cols[i] - 1ull;                                                                  \
            anybad |= (u64)(r0 >= um) | (u64)(c0 >= un);                                                                                             \
            const u64 key = (c0 << rb) | r0;                                                                                                         \
            lo_out[i] = (uint32_t)key;                                                                                                               \
            hi_out[i] = (uint16_t)(key >> 32);                                                                                                       \
        }                                                                                                                                            \
        if (neg)                                                                                                                                     \
            for (i64 i = 0; i < cnt; i++) vals_out[i] = -vals[i];                                                                                    \
        else                                                                                                                                         \
            memcpy(vals_out, vals, sizeof(double) * (size_t)cnt);                                                                                    \
        return anybad;                                                                                                                               \
    }                                                                                                                                                \
    ESP_HOST_SIMD static u64 pack_block_kinds(const TI *rows, const TI *cols, const double *vals, const uint8_t *kinds, bool negate, i64 cnt, u64 um, \
                                              u64 un, int rb, u64 *keys_out, double *vals_out) {                                                    \
        u64 anybad = 0;                                                                                                                              \
        for (i64 i = 0; i < cnt; i++) {                                                                                                              \
            const u64 r0 = (u64)(i64)rows[i] - 1ull, c0 = (u64)(i64)cols[i] - 1ull, kind = (u64)kinds[i];                                            \
            anybad |= (u64)(r0 >= um) | (u64)(c0 >= un) | (u64)(kind > 3ull);                                                                        \
            keys_out[i] = (((c0 << rb) | r0) << ESP_TAG_BITS) | kind;                                                                                \
            const double v = vals[i];                                                                                                                \
            vals_out[i] = (negate && kind != (u64)ESP_SET) ? -v : v;                                                                                 \
        }                                                                                                                                            \
        return anybad;                                                                                                                               \
    }
ESP_PACK_BLOCKS(int64_t)
ESP_PACK_BLOCKS(int32_t)
#undef ESP_PACK_BLOCKS

// lo6 / hi6 != nullptr (one kind for the batch, row + column bits <= 48): the key goes out in SIX bytes -- its low 32 bits and its
// next 16 in two arrays, no kind bits -- instead of keys_out: 14 instead of 16 bytes per entry over PCIe (unpack6_k puts the
// packed key together on the device)
template <typename TI>
static void host_pack(const TI *rows, const TI *cols, const double *vals, const uint8_t *kinds, int kind_all, bool negate, i64 count, i64 m, i64 n,
                      KeyLayout L, u64 *keys_out, double *vals_out, i64 index_base, std::atomic<i64> *bad, uint32_t *lo6 = nullptr,
                      uint16_t *hi6 = nullptr) {
    host_parallel((size_t)count, (size_t)1 << 17, 8, [=](size_t sa_, size_t sb_) {
        const i64 a = (i64)sa_, b = (i64)sb_;
        i64 first_bad = -1;
        // blocks of 2048 entries without a branch (the compiler vectorises them): range checks as unsigned compares,
        // OR-ed into one flag; only a block that raised it is walked again for the first offender
        const u64 um = (u64)m, un = (u64)n;
        const int rb = L.rb;
        for (i64 i0 = a; i0 < b; i0 += 2048) {
            const i64 i1 = std::min(b, i0 + 2048);
            const u64 anybad = lo6 ? pack_block_six(rows + i0, cols + i0, vals + i0, negate && kind_all != ESP_SET, i1 - i0, um, un, rb, lo6 + i0, hi6 + i0,
                                                    vals_out + i0)
                               : kinds ? pack_block_kinds(rows + i0, cols + i0, vals + i0, kinds + i0, negate, i1 - i0, um, un, rb, keys_out + i0, vals_out + i0)
                                     : pack_block_one(rows + i0, cols + i0, vals + i0, (u64)kind_all, negate && kind_all != ESP_SET, i1 - i0, um, un, rb,
                                                          keys_out + i0, vals_out + i0);
            if (anybad && first_bad < 0) {
                for (i64 i = i0; i < i1; i++) {
                    const i64 r = (i64)rows[i], c = (i64)cols[i];
                    const int kind = kinds ? (int)kinds[i] : kind_all;
                    if (!(1 <= r && r <= m && 1 <= c && c <= n) || kind < 0 || kind > 3) {
                        first_bad = i;
                        break;
                    }
                }
            }
        }
        if (first_bad >= 0) {
            i64 cur = bad->load();
            const i64 mine = index_base + first_bad;
            while ((cur < 0 || mine < cur) && !bad->compare_exchange_weak(cur, mine)) {
            }
        }
    });
}

// six-byte keys of one kind (host_pack) -> packed keys of the append buffer
static __global__ void unpack6_k(const uint32_t *__restrict__ lo, const uint16_t *__restrict__ hi, u32 kind, i64 n, u64 *__restrict__ out) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = ((((u64)hi[i] << 32) | (u64)lo[i]) << ESP_TAG_BITS) | (u64)kind;
}

// Bulk append from host arrays: the batch is packed on the host (host_pack) into the pinned staging area in chunks, two
// halves in flight -- the packing of chunk i+1 overlaps the PCIe transfer of chunk i, which lands in the append buffer
// itself; the call is one batch: nothing is committed when an index lies outside the matrix.
template <typename TI>
static int32_t append_host_t(esp_handle *h, const TI *rows, const TI *cols, const double *vals, const uint8_t *kinds, int32_t kind_all, int32_t op,
                             int64_t count) {
    if (!h || count < 0 || (count > 0 && (!rows || !cols || !vals))) return ESP_ERR_INVALID;
    if (count == 0) return ESP_OK;
    (void)hipSetDevice(h->device);
    if (!kinds && (kind_all < 0 || kind_all > 3)) FAIL(h, ESP_ERR_INVALID, "append: kind %d invalid", kind_all);
    if (op != ESP_OP_ADD && op != ESP_OP_SUB) FAIL(h, ESP_ERR_UNSUPPORTED, "append: op %d not supported on the device path", op);
    const i64 chunk = std::min<i64>((i64)1 << 22, std::max<i64>(count, 1));
    esp_handle::StageArea &sa = h->bulk;
    CK(ensure_stage(h, sa, 2 * chunk));  // (its `rows` half holds packed keys here, `vals` the values)
    // one kind for the batch and at most 48 key bits: six-byte keys over PCIe (the low 32 bits in the `rows` area, the next 16 in the
    // `cols` area; two halves of device staging in the area's device mirrors); ESP_HOST_KEYS8: never
    const bool six = !kinds && h->L.rb + h->L.cb <= 48 && h->force_path != ESP_PATH_HOST_KEYS8;  // (38: test hook, packed eight-byte keys)
    if (six) {
        CK(ensure(h, sa.d_rows, sizeof(uint32_t) * (size_t)(2 * chunk)));
        CK(ensure(h, sa.d_cols, sizeof(uint16_t) * (size_t)(2 * chunk)));
    }
    CK(reserve_append(h, count));
    hipEvent_t done[2] = {nullptr, nullptr};
    for (int i = 0; i < 2; i++) HIPCK(h, hipEventCreateWithFlags(&done[i], hipEventDisableTiming));
    std::atomic<i64> bad(-1);
    int32_t rc = ESP_OK;
    i64 it = 0;
    const bool trace = esp_exp_env("ESP_HOST_TRACE") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_wait = 0.0, t_pack = 0.0;
    const double t_begin = now();
    for (i64 off = 0; off < count && rc == ESP_OK && bad.load() < 0; off += chunk, it++) {
        const i64 c = std::min<i64>(chunk, count - off);
        const int half = (int)(it & 1);
        const i64 so = half ? chunk : 0;  // this half of the pinned arrays
        const double t0 = now();
        if (it >= 2 && hipEventSynchronize(done[half]) != hipSuccess) rc = ESP_ERR_HIP;
        const double t1 = now();
        t_wait += t1 - t0;
        uint32_t *lo6 = six ? (uint32_t *)sa.rows + so : nullptr;
        uint16_t *hi6 = six ? (uint16_t *)sa.cols + so : nullptr;
        host_pack<TI>(rows + off, cols + off, vals + off, kinds ? kinds + off : nullptr, kind_all, op == ESP_OP_SUB, c, h->m, h->n, h->L,
                      (u64 *)sa.rows + so, sa.vals + so, off, &bad, lo6, hi6);
        t_pack += now() - t1;
        Span sp(h, ESP_ST_COPY);
        if (six) {
            uint32_t *d_lo = (uint32_t *)sa.d_rows.p + so;
            uint16_t *d_hi = (uint16_t *)sa.d_cols.p + so;
            if (hipMemcpyAsync(d_lo, lo6, sizeof(uint32_t) * (size_t)c, hipMemcpyHostToDevice, h->stream) != hipSuccess ||
                hipMemcpyAsync(d_hi, hi6, sizeof(uint16_t) * (size_t)c, hipMemcpyHostToDevice, h->stream) != hipSuccess ||
                hipMemcpyAsync((double *)h->vals.p + h->count + off, sa.vals + so, sizeof(double) * (size_t)c, hipMemcpyHostToDevice, h->stream) != hipSuccess)
                rc = ESP_ERR_HIP;
            hipLaunchKernelGGL(unpack6_k, dim3(grid_for(c, 256)), dim3(256), 0, h->stream, (const uint32_t *)d_lo, (const uint16_t *)d_hi, (u32)kind_all, c,
                               (u64 *)h->keys.p + h->count + off);
            sp.add(4);
        } else {
        if (hipMemcpyAsync((u64 *)h->keys.p + h->count + off, (u64 *)sa.rows + so, sizeof(u64) * (size_t)c, hipMemcpyHostToDevice, h->stream) != hipSuccess ||
            hipMemcpyAsync((double *)h->vals.p + h->count + off, sa.vals + so, sizeof(double) * (size_t)c, hipMemcpyHostToDevice, h->stream) != hipSuccess)
            rc = ESP_ERR_HIP;
        sp.add(2);
        }
        (void)hipEventRecord(done[half], h->stream);
    }
    const hipError_t e2 = hipStreamSynchronize(h->stream);
    if (trace)
        fprintf(stderr, "esp_append_host: %lld entries in %lld chunks: %.2f ms (packing %.2f, waiting for transfers %.2f), pool of %d\n", (long long)count,
                (long long)it, now() - t_begin, t_pack, t_wait, host_pool().size());
    for (int i = 0; i < 2; i++) (void)hipEventDestroy(done[i]);
    if (rc != ESP_OK || e2 != hipSuccess) {
        h->pre_keep = false;
        FAIL(h, ESP_ERR_HIP, "esp_append_host: transfer failed");
    }
    if (bad.load() >= 0) {
        h->pre_keep = false;
        FAIL(h, ESP_ERR_BOUNDS, "BoundsError: entry %lld of the batch has an index outside %lld x %lld (or a bad kind)", (long long)(bad.load() + 1),
             (long long)h->m, (long long)h->n);
    }
    if (!kinds) note_kind(h, kind_all, count);
    h->count += count;
    pending_changed(h);
    return ESP_OK;
}

extern "C" int32_t esp_append_host(esp_handle *h, const int64_t *rows, const int64_t *cols, const double *vals, const uint8_t *kinds,
                                   int32_t kind_all, int32_t op, int64_t count) {
    return append_host_t<int64_t>(h, rows, cols, vals, kinds, kind_all, op, count);
}
// the same for Int32 index arrays (ExtendableSparseMatrix{Float64, Int32}: extendable.jl:10-25 is generic in Ti)
extern "C" int32_t esp_append_host_i32(esp_handle *h, const int32_t *rows, const int32_t *cols, const double *vals, const uint8_t *kinds,
                                       int32_t kind_all, int32_t op, int64_t count) {
    return append_host_t<int32_t>(h, rows, cols, vals, kinds, kind_all, op, count);
}

// esp_commit: the staged chunk is packed on the host (keys + values, bounds checked there: the answer is immediate) into one
// of two pinned halves, which leaves for the append buffer ASYNCHRONOUSLY -- the call returns as soon as the chunk may be
// refilled, i.e. after the packing; the transfer of chunk i overlaps the loop that fills chunk i+1, and no kernel and no
// stream synchronisation stands between two chunks (round 3: three or four copies, a pack kernel and a round trip for the
// bounds flag per chunk).  Chunks of more than 2^20 entries of one kind go as device triplets instead (commit_as_triplets:
// such an append on an empty buffer may be the partition).
extern "C" int32_t esp_commit(esp_handle *h, int64_t count, int32_t kind_all, int32_t op) {
    if (!h) return ESP_ERR_INVALID;
    if (count < 0 || count > h->stage.cap) FAIL(h, ESP_ERR_INVALID, "esp_commit: count %lld exceeds the staged chunk", (long long)count);
    if (count == 0) return ESP_OK;
    (void)hipSetDevice(h->device);
    if (kind_all > 3) FAIL(h, ESP_ERR_INVALID, "append: kind %d invalid", kind_all);
    if (op != ESP_OP_ADD && op != ESP_OP_SUB) FAIL(h, ESP_ERR_UNSUPPORTED, "append: op %d not supported on the device path", op);
    if (kind_all >= 0 && count > ((i64)1 << 20)) return commit_as_triplets(h, count, kind_all, op);
    esp_handle::CommitPack &cp = h->cpack;
    if (cp.cap < h->stage.cap) {  // (the chunk grew: new halves; transfers out of the old ones have to finish first)
        HIPCK(h, hipStreamSynchronize(h->stream));
        for (int i = 0; i < 2; i++) {
            if (cp.keys[i]) (void)hipHostFree(cp.keys[i]);
            if (cp.vals[i]) (void)hipHostFree(cp.vals[i]);
            cp.keys[i] = nullptr, cp.vals[i] = nullptr, cp.busy[i] = false;
            HIPCK(h, hipHostMalloc((void **)&cp.keys[i], sizeof(u64) * (size_t)h->stage.cap, hipHostMallocDefault));
            HIPCK(h, hipHostMalloc((void **)&cp.vals[i], sizeof(double) * (size_t)h->stage.cap, hipHostMallocDefault));
            if (!cp.done[i]) HIPCK(h, hipEventCreateWithFlags(&cp.done[i], hipEventDisableTiming));
        }
        cp.cap = h->stage.cap;
    }
    const int half = cp.next;
    if (cp.busy[half]) HIPCK(h, hipEventSynchronize(cp.done[half]));  // (the transfer of two chunks ago)
    cp.busy[half] = false;
    std::atomic<i64> bad(-1);
    host_pack<int64_t>(h->stage.rows, h->stage.cols, h->stage.vals, kind_all < 0 ? h->stage.kinds : nullptr, kind_all, op == ESP_OP_SUB, count, h->m, h->n,
                       h->L, cp.keys[half], cp.vals[half], 0, &bad);
    if (bad.load() >= 0)
        FAIL(h, ESP_ERR_BOUNDS, "BoundsError: entry %lld of the batch has an index outside %lld x %lld (or a bad kind)", (long long)(bad.load() + 1),
             (long long)h->m, (long long)h->n);
    CK(reserve_append(h, count));
    {
        Span sp(h, ESP_ST_COPY);
        HIPCK(h, hipMemcpyAsync((u64 *)h->keys.p + h->count, cp.keys[half], sizeof(u64) * (size_t)count, hipMemcpyHostToDevice, h->stream));
        HIPCK(h, hipMemcpyAsync((double *)h->vals.p + h->count, cp.vals[half], sizeof(double) * (size_t)count, hipMemcpyHostToDevice, h->stream));
        sp.add(2);
    }
    HIPCK(h, hipEventRecord(cp.done[half], h->stream));
    cp.busy[half] = true;
    cp.next = 1 - half;
    if (kind_all >= 0) note_kind(h, kind_all, count);
    h->count += count;
    pending_changed(h);
    return ESP_OK;
}

extern "C" int32_t esp_append_device(esp_handle *h, const int64_t *d_rows, const int64_t *d_cols, const double *d_vals,
                                     const uint8_t *d_kinds, int32_t kind_all, int32_t op, int64_t count) {
    if (!h || count < 0 || (count > 0 && (!d_rows || !d_cols || !d_vals))) return ESP_ERR_INVALID;
    (void)hipSetDevice(h->device);
    return pack_device(h, d_rows, d_cols, d_vals, d_kinds, kind_all, op, count);
}

extern "C" int32_t esp_append_packed(esp_handle *h, const uint64_t *d_keys, const double *d_vals, int64_t count) {
    if (!h || count < 0 || (count > 0 && (!d_keys || !d_vals))) return ESP_ERR_INVALID;
    if (count == 0) return ESP_OK;
    (void)hipSetDevice(h->device);
    CK(reserve_append(h, count));
    Span sp(h, ESP_ST_COPY);
    HIPCK(h, hipMemcpyAsync((u64 *)h->keys.p + h->count, d_keys, sizeof(u64) * (size_t)count, hipMemcpyDeviceToDevice, h->stream));
    HIPCK(h, hipMemcpyAsync((double *)h->vals.p + h->count, d_vals, sizeof(double) * (size_t)count, hipMemcpyDeviceToDevice, h->stream));
    sp.add(2);
    h->count += count;
    pending_changed(h);
    return ESP_OK;
}


// All pending entries of the following flushes have their column in [col_lo, col_hi] (1-based).
// The radix partition then spends its bits on that window only (a column shard after the
// exchange).  Entries outside the window make esp_flush return ESP_ERR_STATE.
extern "C" int32_t esp_set_column_window(esp_handle *h, int64_t col_lo, int64_t col_hi) {
    if (!h) return ESP_ERR_INVALID;
    if (col_lo < 1 || col_hi > h->n || col_lo > col_hi) FAIL(h, ESP_ERR_INVALID, "column window [%lld,%lld] outside 1..%lld", (long long)col_lo, (long long)col_hi, (long long)h->n);
    (void)hipSetDevice(h->device);
    CK(fix_tail(h));  // (with the old window)
    h->win_base = (u64)(col_lo - 1) << h->L.rb;
    h->win_span = (u64)(col_hi - col_lo + 1) << h->L.rb;
    h->wc0 = col_lo - 1;
    h->wc1 = col_hi;
    h->win_excl = h->nnz == 0 && h->count == 0;  // nothing stored or pending outside it, and flushes enforce it from now on
    return ESP_OK;
}


// ------------------------------------------------------------------------ CSC side
extern "C" int32_t esp_set_csc(esp_handle *h, const int64_t *colptr, const int64_t *rowval, const double *nzval, int64_t nnz) {
    if (!h || !colptr || nnz < 0 || (nnz > 0 && (!rowval || !nzval))) return ESP_ERR_INVALID;
    if (colptr[0] != 1 || colptr[h->n] != nnz + 1) FAIL(h, ESP_ERR_INVALID, "esp_set_csc: colptr[1]=%lld colptr[n+1]=%lld nnz=%lld violate the CSC invariants", (long long)colptr[0], (long long)colptr[h->n], (long long)nnz);
    (void)hipSetDevice(h->device);
    CK(ensure(h, h->colptr, sizeof(i64) * (size_t)(h->n + 1)));
    CK(ensure(h, h->rowval, sizeof(i64) * (size_t)std::max<i64>(nnz, 1)));
    CK(ensure(h, h->nzval, sizeof(double) * (size_t)std::max<i64>(nnz, 1)));
    Span sp(h, ESP_ST_COPY);
    CK(h2d_pipelined(h, h->colptr.p, colptr, sizeof(i64) * (size_t)(h->n + 1)));
    if (nnz > 0) {
        CK(h2d_pipelined(h, h->rowval.p, rowval, sizeof(i64) * (size_t)nnz));
        CK(h2d_pipelined(h, h->nzval.p, nzval, sizeof(double) * (size_t)nnz));
    }
    sp.add(3);
    h->nnz = nnz;
    h->pattern_version++, h->values_version++;
    h->csc_valid = true;
    h->win_excl = false;  // (the uploaded CSC may hold entries outside a declared window)
    h->tail_stale = false;
    h->ones_pending = false;
    return ESP_OK;
}

// Device -> pageable host memory (a Julia Vector, a NumPy array) through two pinned bounce buffers: the
// PCIe transfer of chunk i+1 overlaps the (multi-threaded) host copy of chunk i.  A plain hipMemcpy into
// pageable memory runs at ~10 GB/s here, this at ~45 GB/s.
int32_t ensure_bounce(esp_handle *h) {
    esp_handle::Bounce &b = h->bounce;
    const size_t want = (size_t)32 << 20;
    for (int i = 0; i < 2; i++) {
        if (!b.pin[i]) {
            const hipError_t e = hipHostMalloc((void **)&b.pin[i], want, hipHostMallocDefault);
            if (e != hipSuccess) {
                b.pin[i] = nullptr;
                FAIL(h, ESP_ERR_NOMEM, "pinned bounce buffer: %s", hipGetErrorString(e));
            }
        }
        if (!b.ev[i]) HIPCK(h, hipEventCreateWithFlags(&b.ev[i], hipEventDisableTiming));
    }
    b.bytes = want;
    return ESP_OK;
}

// A large destination the caller has just allocated (the vectors of a fresh SparseMatrixCSC: Base.sum and `lnk + csc` return new
// arrays) is untouched memory: the host copy below pays one page fault per 4 KiB -- 560 MB: 48 ms with eight threads on the hosts
// of this pool, against 4.5 ms for the copy itself (tools/r6_prefault.c).  With transparent huge pages in `madvise` mode a hint on
// the 2 MiB-aligned interior makes those faults 2 MiB ones: 7 ms.  A hint only (no contents, no semantics change; ignored where
// the range is not anonymous memory or the kernel has no THP).
static void hint_huge_pages(void *dst, size_t bytes) {
    const uintptr_t two_mb = (uintptr_t)2 << 20;
    const uintptr_t a = ((uintptr_t)dst + two_mb - 1) & ~(two_mb - 1), b = ((uintptr_t)dst + bytes) & ~(two_mb - 1);
    if (b > a) (void)madvise((void *)a, (size_t)(b - a), MADV_HUGEPAGE);
}

int32_t d2h_pipelined(esp_handle *h, void *dst, const void *d_src, size_t bytes) {
    if (bytes == 0) return ESP_OK;
    const size_t small = (size_t)8 << 20;
    if (bytes <= small) {
        HIPCK(h, hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, h->stream));
        HIPCK(h, hipStreamSynchronize(h->stream));
        return ESP_OK;
    }
    hint_huge_pages(dst, bytes);
    CK(ensure_bounce(h));
    HIPCK(h, hipStreamSynchronize(h->stream));  // (an earlier transfer through the halves is over)
    char *const *pin = h->bounce.pin;
    const size_t chunk = h->bounce.bytes;
    hipEvent_t *ev = h->bounce.ev;
    const size_t nchunks = (bytes + chunk - 1) / chunk;
    int32_t rc = ESP_OK;
    hipError_t he = hipSuccess;
    for (size_t c = 0; c <= nchunks && rc == ESP_OK; c++) {
        if (c < nchunks) {  // issue the transfer of chunk c
            const size_t o = c * chunk, len = std::min(chunk, bytes - o);
            if ((he = hipMemcpyAsync(pin[c & 1], (const char *)d_src + o, len, hipMemcpyDeviceToHost, h->stream)) != hipSuccess ||
                (he = hipEventRecord(ev[c & 1], h->stream)) != hipSuccess)
                rc = ESP_ERR_HIP;
        }
        if (c > 0 && rc == ESP_OK) {  // ... while chunk c-1 moves from its bounce buffer to the caller
            const size_t o = (c - 1) * chunk, len = std::min(chunk, bytes - o);
            if ((he = hipEventSynchronize(ev[(c - 1) & 1])) != hipSuccess) rc = ESP_ERR_HIP;
            else par_memcpy((char *)dst + o, pin[(c - 1) & 1], len);
        }
    }
    (void)hipStreamSynchronize(h->stream);
    if (rc != ESP_OK) FAIL(h, ESP_ERR_HIP, "device-to-host transfer failed: %s", hipGetErrorString(he));
    return ESP_OK;
}

// Pageable host memory -> device: the (multi-threaded) host copy of chunk i+1 into one pinned half overlaps the PCIe transfer
// of chunk i out of the other.  A plain hipMemcpy from pageable memory stages through the runtime's own buffer on one thread
// (the values of a 7 10^7-entry matrix: 560 MB in ~30 ms; this: ~13).  Returns when the device holds the data.
int32_t h2d_pipelined(esp_handle *h, void *d_dst, const void *src, size_t bytes) {
    if (bytes == 0) return ESP_OK;
    const size_t small = (size_t)8 << 20;
    if (bytes <= small) {
        HIPCK(h, hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, h->stream));
        HIPCK(h, hipStreamSynchronize(h->stream));
        return ESP_OK;
    }
    CK(ensure_bounce(h));
    HIPCK(h, hipStreamSynchronize(h->stream));  // (an earlier transfer out of / into the halves is over)
    char *const *pin = h->bounce.pin;
    const size_t chunk = h->bounce.bytes;
    hipEvent_t *ev = h->bounce.ev;
    const size_t nchunks = (bytes + chunk - 1) / chunk;
    int32_t rc = ESP_OK;
    hipError_t he = hipSuccess;
    for (size_t c = 0; c < nchunks && rc == ESP_OK; c++) {
        const size_t o = c * chunk, len = std::min(chunk, bytes - o);
        const int half = (int)(c & 1);
        if (c >= 2 && (he = hipEventSynchronize(ev[half])) != hipSuccess) rc = ESP_ERR_HIP;  // (chunk c-2 has left this half)
        if (rc == ESP_OK) {
            par_memcpy(pin[half], (const char *)src + o, len);
            if ((he = hipMemcpyAsync((char *)d_dst + o, pin[half], len, hipMemcpyHostToDevice, h->stream)) != hipSuccess ||
                (he = hipEventRecord(ev[half], h->stream)) != hipSuccess)
                rc = ESP_ERR_HIP;
        }
    }
    (void)hipStreamSynchronize(h->stream);
    if (rc != ESP_OK) FAIL(h, ESP_ERR_HIP, "host-to-device transfer failed: %s", hipGetErrorString(he));
    return ESP_OK;
}

// Int64 -> u32 on the device, u32 -> Int64 on the way from the bounce buffers into the caller's array: row indices (and
// colptr values) of a matrix with fewer than 2^32 rows (entries) cross PCIe as 4 bytes
static __global__ void narrow_i64_k(const i64 *__restrict__ in, i64 n, u32 *__restrict__ out) {
    const i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (g < n) out[g] = (u32)in[g];
}
static void par_widen(i64 *dst, const u32 *src, size_t count) {
    host_parallel(count, (size_t)1 << 18, 8, [=](size_t a, size_t b) {
        for (size_t i = a; i < b; i++) dst[i] = (i64)src[i];
    });
}
// `count` Int64 values of device array d_src (all below 2^32) into dst: narrowed by a kernel into scratch, moved as u32
// through the two pinned bounce buffers, widened by the host copy that fills the caller's array anyway
static int32_t d2h_narrow(esp_handle *h, i64 *dst, const i64 *d_src, i64 count, DevBuf &scratch) {
    if (count == 0) return ESP_OK;
    CK(ensure(h, scratch, sizeof(u32) * (size_t)count));
    hipLaunchKernelGGL(narrow_i64_k, dim3(grid_for(count, 256)), dim3(256), 0, h->stream, d_src, count, (u32 *)scratch.p);
    HIPCK(h, hipGetLastError());
    CK(ensure_bounce(h));
    if ((size_t)count * sizeof(i64) > ((size_t)8 << 20)) hint_huge_pages(dst, (size_t)count * sizeof(i64));
    char *const *pin = h->bounce.pin;
    const size_t chunk = h->bounce.bytes / sizeof(u32);  // elements per bounce buffer
    hipEvent_t *ev = h->bounce.ev;
    const size_t total = (size_t)count, nchunks = (total + chunk - 1) / chunk;
    int32_t rc = ESP_OK;
    for (size_t c = 0; c <= nchunks && rc == ESP_OK; c++) {
        if (c < nchunks) {
            const size_t o = c * chunk, len = std::min(chunk, total - o);
            if (hipMemcpyAsync(pin[c & 1], (const u32 *)scratch.p + o, sizeof(u32) * len, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
                hipEventRecord(ev[c & 1], h->stream) != hipSuccess)
                rc = ESP_ERR_HIP;
        }
        if (c > 0 && rc == ESP_OK) {
            const size_t o = (c - 1) * chunk, len = std::min(chunk, total - o);
            if (hipEventSynchronize(ev[(c - 1) & 1]) != hipSuccess) rc = ESP_ERR_HIP;
            else par_widen(dst + o, (const u32 *)pin[(c - 1) & 1], len);
        }
    }
    (void)hipStreamSynchronize(h->stream);
    if (rc != ESP_OK) FAIL(h, ESP_ERR_HIP, "device-to-host transfer failed");
    return ESP_OK;
}

extern "C" int32_t esp_get_csc(esp_handle *h, int64_t *colptr, int64_t *rowval, double *nzval) {
    if (!h || !colptr) return ESP_ERR_INVALID;
    (void)hipSetDevice(h->device);
    CK(fix_tail(h));
    if (h->nnz > 0 && (!rowval || !nzval)) FAIL(h, ESP_ERR_INVALID, "esp_get_csc: rowval/nzval NULL with nnz>0");
    Span sp(h, ESP_ST_COPY);
    // (large matrices whose indices fit 32 bits: colptr and rowval cross PCIe as u32 -- 12 instead of 16 bytes per entry)
    const bool narrow = h->nnz >= ((i64)1 << 20) && h->m < ((i64)1 << 32) && h->nnz + 1 < ((i64)1 << 32);
    if (narrow) {
        CK(d2h_narrow(h, colptr, (const i64 *)h->colptr.p, h->n + 1, h->heads));
        CK(d2h_narrow(h, rowval, (const i64 *)h->rowval.p, h->nnz, h->heads));
    } else {
        CK(d2h_pipelined(h, colptr, h->colptr.p, sizeof(i64) * (size_t)(h->n + 1)));
        if (h->nnz > 0) CK(d2h_pipelined(h, rowval, h->rowval.p, sizeof(i64) * (size_t)h->nnz));
    }
    if (h->nnz > 0) CK(d2h_pipelined(h, nzval, h->nzval.p, sizeof(double) * (size_t)h->nnz));
    sp.add(3);
    return ESP_OK;
}

// The Int32 forms of the CSC transfers (ExtendableSparseMatrix{Float64,Int32}: extendable.jl:10-25 is generic in Ti).  The device
// CSC stays Int64; the index arrays are narrowed / widened by a small kernel beside the transfer, so they cross PCIe as 4 bytes.
static __global__ void widen_i32_k(const int32_t *__restrict__ in, i64 n, i64 *__restrict__ out) {
    const i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (g < n) out[g] = (i64)in[g];
}
extern "C" int32_t esp_get_csc_i32(esp_handle *h, int32_t *colptr, int32_t *rowval, double *nzval) {
    if (!h || !colptr) return ESP_ERR_INVALID;
    (void)hipSetDevice(h->device);
    CK(fix_tail(h));
    if (h->nnz > 0 && (!rowval || !nzval)) FAIL(h, ESP_ERR_INVALID, "esp_get_csc_i32: rowval/nzval NULL with nnz>0");
    if (h->m > (i64)INT32_MAX || h->nnz + 1 > (i64)INT32_MAX)
        FAIL(h, ESP_ERR_UNSUPPORTED, "esp_get_csc_i32: %lld rows / %lld entries do not fit Int32 indices", (long long)h->m, (long long)h->nnz);
    Span sp(h, ESP_ST_COPY);
    const i64 counts[2] = {h->n + 1, h->nnz};
    const i64 *src[2] = {(const i64 *)h->colptr.p, (const i64 *)h->rowval.p};
    int32_t *dst[2] = {colptr, rowval};
    for (int a = 0; a < 2; a++) {
        if (counts[a] == 0) continue;
        CK(ensure(h, h->heads, sizeof(u32) * (size_t)counts[a]));
        hipLaunchKernelGGL(narrow_i64_k, dim3(grid_for(counts[a], 256)), dim3(256), 0, h->stream, src[a], counts[a], (u32 *)h->heads.p);
        HIPCK(h, hipGetLastError());
        CK(d2h_pipelined(h, dst[a], h->heads.p, sizeof(u32) * (size_t)counts[a]));
    }
    if (h->nnz > 0) CK(d2h_pipelined(h, nzval, h->nzval.p, sizeof(double) * (size_t)h->nnz));
    sp.add(3);
    return ESP_OK;
}
extern "C" int32_t esp_set_csc_i32(esp_handle *h, const int32_t *colptr, const int32_t *rowval, const double *nzval, int64_t nnz) {
    if (!h || !colptr || nnz < 0 || (nnz > 0 && (!rowval || !nzval))) return ESP_ERR_INVALID;
    if (colptr[0] != 1 || (i64)colptr[h->n] != nnz + 1) FAIL(h, ESP_ERR_INVALID, "esp_set_csc_i32: colptr[1]=%d colptr[n+1]=%d nnz=%lld violate the CSC invariants", colptr[0], colptr[h->n], (long long)nnz);
    (void)hipSetDevice(h->device);
    CK(ensure(h, h->colptr, sizeof(i64) * (size_t)(h->n + 1)));
    CK(ensure(h, h->rowval, sizeof(i64) * (size_t)std::max<i64>(nnz, 1)));
    CK(ensure(h, h->nzval, sizeof(double) * (size_t)std::max<i64>(nnz, 1)));
    CK(ensure(h, h->heads, sizeof(int32_t) * (size_t)std::max<i64>(std::max<i64>(nnz, h->n + 1), 1)));
    Span sp(h, ESP_ST_COPY);
    const i64 counts[2] = {h->n + 1, nnz};
    const int32_t *src[2] = {colptr, rowval};
    i64 *dst[2] = {(i64 *)h->colptr.p, (i64 *)h->rowval.p};
    for (int a = 0; a < 2; a++) {
        if (counts[a] == 0) continue;
        CK(h2d_pipelined(h, h->heads.p, src[a], sizeof(int32_t) * (size_t)counts[a]));
        hipLaunchKernelGGL(widen_i32_k, dim3(grid_for(counts[a], 256)), dim3(256), 0, h->stream, (const int32_t *)h->heads.p, counts[a], dst[a]);
        HIPCK(h, hipGetLastError());
    }
    if (nnz > 0) CK(h2d_pipelined(h, h->nzval.p, nzval, sizeof(double) * (size_t)nnz));
    sp.add(3);
    HIPCK(h, hipStreamSynchronize(h->stream));
    h->nnz = nnz;
    h->pattern_version++, h->values_version++;
    h->csc_valid = true;
    h->win_excl = false;
    h->tail_stale = false;
    h->ones_pending = false;
    return ESP_OK;
}

extern "C" int32_t esp_get_nzval(esp_handle *h, double *nzval) {
    if (!h) return ESP_ERR_INVALID;
    if (h->nnz == 0) return ESP_OK;
    if (!nzval) return ESP_ERR_INVALID;
    (void)hipSetDevice(h->device);
    return d2h_pipelined(h, nzval, h->nzval.p, sizeof(double) * (size_t)h->nnz);
}

extern "C" int32_t esp_csc_device(esp_handle *h, const int64_t **d_colptr, const int64_t **d_rowval, const double **d_nzval) {
    if (!h) return ESP_ERR_INVALID;
    (void)hipSetDevice(h->device);
    CK(fix_tail(h));
    if (d_colptr) *d_colptr = (const i64 *)h->colptr.p;
    if (d_rowval) *d_rowval = (const i64 *)h->rowval.p;
    if (d_nzval) *d_nzval = (const double *)h->nzval.p;
    return ESP_OK;
}

extern "C" int32_t esp_clear_pending(esp_handle *h) {
    if (!h) return ESP_ERR_INVALID;
    h->count = 0;
    pending_changed(h);
    return ESP_OK;
}

extern "C" int32_t esp_reset(esp_handle *h) {
    if (!h) return ESP_ERR_INVALID;
    (void)hipSetDevice(h->device);
    h->hits_off = false;  // (a new matrix: its re-assemblies may be what the re-assembly kernel takes)
    h->count = 0;
    pending_changed(h);
    if (h->lazy_hold.p) release(h->lazy_hold);  // (hipFree waits for whatever still reads it)
    return init_empty_csc(h);
}

extern "C" int32_t esp_zero_values(esp_handle *h) {
    if (!h) return ESP_ERR_INVALID;
    if (h->nnz == 0) return ESP_OK;
    (void)hipSetDevice(h->device);
    HIPCK(h, hipMemsetAsync(h->nzval.p, 0, sizeof(double) * (size_t)h->nnz, h->stream));
    h->values_version++;
    return ESP_OK;
}


// test/bench hook: 0 = automatic, 1 = (same as 0), 2 = force the general global path
extern "C" int32_t esp_debug_force_path(esp_handle *h, int32_t path) {
    if (!h) return ESP_ERR_INVALID;
    h->force_path = path;
    return ESP_OK;
}
extern "C" int32_t esp_debug_plan_cap(esp_handle *h, double cap) {
    if (!h) return ESP_ERR_INVALID;
    h->debug_plan_cap = cap > 0.0 ? cap : 0.0;
    return ESP_OK;
}
extern "C" int32_t esp_debug_fail_next_bucket_stage(esp_handle *h) {
    if (!h) return ESP_ERR_INVALID;
    h->debug_fail_bucket = true;
    return ESP_OK;
}
extern "C" int32_t esp_debug_last_run_order(const esp_handle *h, int32_t *kind) {
    if (!h || !kind) return ESP_ERR_INVALID;
    *kind = h->last_run_order;
    return ESP_OK;
}
extern "C" int32_t esp_debug_last_fold_update(const esp_handle *h, int32_t *on) {
    if (!h || !on) return ESP_ERR_INVALID;
    *on = h->last_fold_update;
    return ESP_OK;
}
extern "C" int32_t esp_debug_last_key_bytes(const esp_handle *h, int32_t *bytes) {
    if (!h || !bytes) return ESP_ERR_INVALID;
    *bytes = h->last_key_bytes;
    return ESP_OK;
}
extern "C" int32_t esp_debug_last_colptr_direct(const esp_handle *h, int32_t *direct) {
    if (!h || !direct) return ESP_ERR_INVALID;
    *direct = h->last_colptr_direct;
    return ESP_OK;
}
extern "C" int32_t esp_debug_last_plan_reused(const esp_handle *h, int32_t *reused) {
    if (!h || !reused) return ESP_ERR_INVALID;
    *reused = h->last_plan_reused;
    return ESP_OK;
}
extern "C" int32_t esp_debug_last_partition(const esp_handle *h, int32_t *kind) {
    if (!h || !kind) return ESP_ERR_INVALID;
    *kind = h->last_partition;
    return ESP_OK;
}
extern "C" int32_t esp_debug_last_path(const esp_handle *h, int32_t *path) {
    if (!h || !path) return ESP_ERR_INVALID;
    *path = h->last_path;
    return ESP_OK;
}


// ------------------------------------------------------------------------ measurement
extern "C" int32_t esp_timing_enable(esp_handle *h, int32_t on) {
    if (!h) return ESP_ERR_INVALID;
    timing_collect(h);
    h->timing = on != 0;
    h->timing_level = (on == 1 || on == 3) ? on : 2;  // 1: big kernels; 3: bucket / fold kernel only; else every stage
    return ESP_OK;
}
extern "C" int32_t esp_timing(esp_handle *h, esp_timing_t *out, int32_t clear) {
    if (!h || !out) return ESP_ERR_INVALID;
    timing_collect(h);
    *out = h->acc;
    if (clear) memset(&h->acc, 0, sizeof h->acc);
    return ESP_OK;
}

