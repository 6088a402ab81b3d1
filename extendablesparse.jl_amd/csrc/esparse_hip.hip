// esparse_hip.hip -- C ABI of libesparse_hip.so (see include/esparse_hip.h) and the host
// orchestration of the flush pipeline.  gfx950 only; no CPU fallback of any kind: without
// a GPU every entry point that needs one returns ESP_ERR_NODEVICE.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <thread>

#include "common.hpp"
#include "fold.hpp"
#include "generators.hpp"
#include "femitems.hpp"
#include "local_args.hpp"
#include "merge.hpp"
#include "radix.hpp"
#include "runpart.hpp"
#include "scan.hpp"

bool esplocal::launch(const Variant &v, unsigned grid, hipStream_t stream, const Args &a) {
    if (v.pieces) return v.small_variant ? launch_pieces_small(v, grid, stream, a) : v.fresh ? launch_pieces_fresh(v, grid, stream, a) : launch_pieces_stored(v, grid, stream, a);
    return v.small_variant ? launch_small(v, grid, stream, a) : launch_regular(v, grid, stream, a);
}

// ------------------------------------------------------------------------ handle
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
};

struct TimedSpan {
    int stage;
    hipEvent_t a, b;
    int launches;
};

struct esp_handle {
    i64 m = 0, n = 0;
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    KeyLayout L{1, 1};
    std::string err;

    // COO append buffer
    DevBuf keys, vals;
    i64 cap = 0, count = 0;
    // ping-pong / scratch
    DevBuf keys2, vals2, hist, segs, colend, newkey, newval, heads, misc, seg[2], tilef[2], segcnt, segout;
    int force_path = 0, last_path = 0;
    DevBuf runbuf, chunkbuf;
    i64 chunk_cap = 0, hint = 0;
    int chunk_pb = 0;
    int runs_skip = 0, runs_penalty = 0;  // back-off after a stream turned out not to be pre-sorted
    int seen_maxrun = 0;                  // longest column run the bucket kernel met in the last flush
    int last_partition = 0;               // 1 = run-based single pass, 2 = 8-bit passes only, 4 = the producer's, 7 = shard pieces
    // Producer-side partition: a device-side producer appended to the empty buffer with its PART kernel (runpart.hpp,
    // "the append IS the partition"): the pending entries lie bucket by bucket -- a stable permutation of the stream --
    // and the flush starts at the bucket kernel.  Bucket starts: seg[1] (S + 1 entries).
    struct PrePart {
        bool valid = false;
        int K = 0, pb = 0;       // bits of the key window / of the prefix: S = 1 << pb buckets
        int key_bytes = 8;       // 4: `keys` holds u32 keys (the bits below the prefix); every entry has the kind `kind`
        int kind = 0;
        i64 E = 0, maxlen = 0;   // entries, longest bucket
        i64 tail = 0;            // packed entries appended BEHIND the E bucket-ordered ones (count = E + tail)
        u64 base = 0, span = 0;  // the key window it was made for
        double Ee = 0.0;         // (plan_entries of the batch: spread bookkeeping)
        // column shards: the batch was partitioned by (owner, digit inside the owner's column range) -- what
        // esp_shard_partition produces; mw_P windows of mw_nb digits, digit width 2^mw_shift, plan made for mw_eps
        int mw_P = 0, mw_me = 0, mw_shift = 0;
        u32 mw_nb = 0;
        i64 mw_eps = 0;
        bool own32 = false;      // ... and the shard's OWN range holds 4-byte keys of kind `kind` (the sent ranges: packed)
    } pre;
    bool pre_keep = false;       // reserve_append: the append that follows goes behind the bucket-ordered batch
    // sort_msd over ITEM records (femitems.hpp): a segment may hold plan_cap records (the bucket kernel's capacity in
    // updates / updates per item), and a shuffled stream need not be tried as a pre-sorted one
    i64 plan_cap = 0;
    bool item_mode = false;
    bool shard_user = false;     // the handle is driven through esp_shard_*: its flushes partition by owner first
    int last_shard_source = 0;   // esp_shard_partition: 1 = its own pass moved the entries, 2 = the producer had
    int last_local_small = 0;    // the last flush's bucket kernel was the small variant (3 workgroups per CU)
    // esp_shard_plan: the producers that find the buffer empty partition for the next esp_shard_partition(P, me, eps)
    struct ShardPlan {
        bool valid = false;
        int P = 0, me = 0;
        i64 eps = 0;
    } shard_plan;
    // column window of the pending entries (whole matrix by default)
    u64 win_base = 0, win_span = 0;
    // A shard works on its column range only.  When the window [wc0, wc1) (0-based columns) was
    // declared on an empty matrix and kept since, no entry can lie outside it: the per-column work of a
    // flush (colend clear, colptr scan) then runs over the window only -- with P shards the global colptr
    // has P times the columns a shard owns.  colptr[c] = 1 for c <= wc0 always; the part behind the window
    // (= nnz+1) is rewritten only when somebody needs the whole array (tail_stale).
    i64 wc0 = 0, wc1 = 0;
    bool win_excl = false, tail_stale = false;
    // reset! of an unwindowed matrix leaves colptr := 1 to whoever reads it next (fix_tail): the fresh flush that
    // normally follows rewrites every entry
    bool ones_pending = false;
    // device CSC (Julia layout) + spare set for rebuilds
    DevBuf colptr, rowval, nzval, rowval2, nzval2;
    i64 nnz = 0;
    bool csc_valid = false;  // colptr initialised
    // host staging (pinned) + device staging
    // pinned staging areas (+ their device mirrors): `stage` is the one esp_stage_begin hands to the caller
    // (its pointers stay valid until the caller asks for a larger one); `bulk` is private to esp_append_host
    struct StageArea {
        i64 cap = 0;
        i64 *rows = nullptr, *cols = nullptr;
        double *vals = nullptr;
        uint8_t *kinds = nullptr;
        DevBuf d_rows, d_cols, d_vals, d_kinds;
    } stage, bulk;
    unsigned long long *pin_scalar = nullptr;  // pinned, 8 slots
    u64 *pin_mw = nullptr;  // pinned source of prepart_begin's asynchronous upload of the window bases (<= MW_MAX)
    hipEvent_t pin_mw_done = nullptr;
    // kind bookkeeping of the pending batch: when every pending entry was appended with ONE known kind the run-based
    // partition hands the bucket kernel 4-byte keys (the key bits below the partition prefix) instead of packed keys
    i64 kind_noted = 0;     // pending entries appended with a single known kind
    int kind_uniform = -1;  // that kind; -1 none yet, -2 mixed / an append of unknown kinds (until the buffer is empty again)
    int last_key_bytes = 8;      // esp_debug_last_key_bytes
    // longest segment / average segment of the last bucket-path flush (0: not known): an assembly that repeats on
    // a handle with regular data (spread ~1.0x) is planned one partition bit tighter when the predicted longest
    // segment still fits the bucket kernel -- half-full segments cost that kernel up to 1.8x
    double seen_spread = 0.0;
    int last_fold_update = 0;    // the register tiers of the last flush ran their UPDATE-only fold
    bool part_own32 = false;       // the own range of the partitioned buffer holds 4-byte keys (kind part_kind32)
    int part_kind32 = 0;
    bool part_own_update = false;  // esp_shard_partition: every pending entry was appended as an UPDATE
    bool part_all_update = false;  // esp_shard_assemble: ... and so is every received entry (checked on the device)
    int last_run_order = 0;      // esp_debug_last_run_order
    int last_colptr_direct = 0;  // the bucket kernel of the last flush wrote colptr itself
    hipStream_t aux = nullptr;   // second stream + event: small device-to-host reads beside a running kernel
    hipEvent_t aux_ev = nullptr;  // (created on first use, aux_ready)
    // shard cache
    bool shard_valid = false;
    int shard_P = 0;
    // partitioned exchange (esp_shard_partition / esp_shard_assemble)
    bool part_valid = false;      // the pending buffer is partitioned by (owner, digit); tables in parttab
    bool part_assembled = false;  // piece tables are built: the next flush runs the bucket kernel on them
    int part_P = 0, part_me = 0, part_shift = 0;
    u32 part_nb = 0;
    u64 part_base = 0, part_span = 0;
    i64 part_total = 0, part_maxlen = 0, part_own_lo = 0;
    DevBuf parttab, piecetab;
    // row-wise view of the device CSC for mul! (built on first use after a pattern change)
    unsigned long long pattern_version = 1, csr_version = 0;
    unsigned long long values_version = 1, csr_val_version = 0;  // nzval changed / row-wise copy of the values
    DevBuf csr_rowptr, csr_perm, csr_col, csr_tmp, csr_val, mul_x, mul_r;
    // timing
    bool timing = false;
    int timing_level = 2;
    std::vector<hipEvent_t> ev_pool;
    std::vector<TimedSpan> spans;
    esp_timing_t acc;
    hipEvent_t flush_a = nullptr, flush_b = nullptr;
};

static thread_local std::string g_err;
static void par_memcpy(void *dst, const void *src, size_t bytes);

// the pending entries changed: whatever was derived from them is stale
static inline void pending_changed(esp_handle *h) {
    if (h->count == 0) {
        h->kind_noted = 0;
        h->kind_uniform = -1;
    } else if (h->kind_noted != h->count) {
        h->kind_uniform = -2;  // (some entries came or went without note_kind: sticky until the buffer is empty)
    }
    h->shard_valid = false;
    h->part_valid = false;
    h->part_assembled = false;
    // (whoever changes a bucket-ordered buffer called pending_materialize first -- or appends behind it)
    if (h->pre.valid && h->pre_keep && h->count >= h->pre.E)
        h->pre.tail = h->count - h->pre.E;
    else
        h->pre.valid = false;
    h->pre_keep = false;
}
static int32_t pending_materialize(esp_handle *h);

// set-up of a producer-side partition (prepart_* below, next to run_partition)
struct PartSetup {
    bool on = false;
    esprun::PartOut out;   // for the producer's PART kernel
    esprun::RunSink sink;  // for its COUNT kernel
    u32 *err = nullptr;    // window flag of the COUNT kernel
    int K = 0, pb = 0, kind = -1;
    i64 E = 0, chunks = 0;
    double Ee = 0.0;
    i64 NB = 0;            // buckets (1 << pb, or shards * digits per shard)
    int mw_P = 0, mw_me = 0, mw_shift = 0;
    u32 mw_nb = 0;
    i64 mw_eps = 0;
    i64 *seg_out = nullptr, *runs_off = nullptr;
    const unsigned long long *bucket_count = nullptr;
    u64 *coarse = nullptr;
    const u32 *dcount = nullptr;
    const u64 *dlist = nullptr;
};
static int32_t prepart_begin(esp_handle *h, i64 E, i64 chunks, int kind, PartSetup *ps);
static int32_t prepart_rank(esp_handle *h, PartSetup *ps);
static int32_t prepart_finish(esp_handle *h, PartSetup *ps, bool *took);

// call right before h->count grows by cnt entries that all carry `kind`
static inline void note_kind(esp_handle *h, int kind, i64 cnt) {
    if (h->count == 0 && h->kind_noted == 0 && h->kind_uniform == -1) h->kind_uniform = kind;
    else if (h->kind_uniform != kind) h->kind_uniform = -2;
    h->kind_noted += cnt;
}

#define FAIL(h, code, ...)                                   \
    do {                                                     \
        char _b[512];                                        \
        snprintf(_b, sizeof _b, __VA_ARGS__);                \
        if (h) (h)->err = _b;                                \
        g_err = _b;                                          \
        return (code);                                       \
    } while (0)

#define HIPCK(h, call)                                                                          \
    do {                                                                                        \
        hipError_t _e = (call);                                                                 \
        if (_e != hipSuccess)                                                                   \
            FAIL(h, _e == hipErrorOutOfMemory ? ESP_ERR_NOMEM : ESP_ERR_HIP, "%s failed: %s",   \
                 #call, hipGetErrorString(_e));                                                 \
    } while (0)

#define CK(...)                       \
    do {                              \
        int32_t _s = (__VA_ARGS__);   \
        if (_s != ESP_OK) return _s;  \
    } while (0)

static int32_t ensure(esp_handle *h, DevBuf &b, size_t need, bool keep = false) {
    if (b.bytes >= need && b.p) return ESP_OK;
    size_t want = need;
    if (keep && b.bytes) want = std::max(need, b.bytes + b.bytes / 2);
    want = (want + 255) & ~(size_t)255;
    void *np = nullptr;
    HIPCK(h, hipMalloc(&np, want));
    if (keep && b.p && b.bytes) {
        HIPCK(h, hipMemcpyAsync(np, b.p, b.bytes, hipMemcpyDeviceToDevice, h->stream));
        HIPCK(h, hipStreamSynchronize(h->stream));
    }
    if (b.p) (void)hipFree(b.p);
    b.p = np;
    b.bytes = want;
    return ESP_OK;
}
static void release(DevBuf &b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.bytes = 0;
}

static void release_all(esp_handle *h) {
    for (DevBuf *b : {&h->keys, &h->vals, &h->keys2, &h->vals2, &h->hist, &h->segs, &h->colend, &h->newkey,
                      &h->newval, &h->heads, &h->misc, &h->colptr, &h->rowval, &h->nzval, &h->rowval2,
                      &h->nzval2, &h->seg[0], &h->seg[1], &h->tilef[0], &h->tilef[1], &h->segcnt, &h->segout, &h->runbuf, &h->chunkbuf, &h->parttab, &h->piecetab, &h->csr_rowptr, &h->csr_perm, &h->csr_col, &h->csr_tmp, &h->csr_val, &h->mul_x, &h->mul_r, &h->stage.d_rows, &h->stage.d_cols, &h->stage.d_vals, &h->stage.d_kinds, &h->bulk.d_rows, &h->bulk.d_cols, &h->bulk.d_vals, &h->bulk.d_kinds})
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
}

// ------------------------------------------------------------------------ timing
static hipEvent_t ev_get(esp_handle *h) {
    if (!h->ev_pool.empty()) {
        hipEvent_t e = h->ev_pool.back();
        h->ev_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
static void timing_collect(esp_handle *h) {
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
struct Span {
    esp_handle *h;
    int stage;
    hipEvent_t a = nullptr;
    int launches = 0;
    Span(esp_handle *hh, int st) : h(hh), stage(st) {
        // timing level 1 brackets the big kernels only: the ~20 tiny launches of the "scan" stage would cost
        // more in event records (two per span) than they run
        if (h->timing && (h->timing_level == 2 || (h->timing_level == 1 && st != ESP_ST_SCAN) ||
                          (h->timing_level == 3 && (st == ESP_ST_LOCAL || st == ESP_ST_FOLD)))) {
            a = ev_get(h);
            (void)hipEventRecord(a, h->stream);
        }
    }
    void add(int l) { launches += l; }
    ~Span() {
        if (h->timing && a) {
            hipEvent_t b = ev_get(h);
            (void)hipEventRecord(b, h->stream);
            h->spans.push_back({stage, a, b, launches});
            if (h->spans.size() > 2048) timing_collect(h);
        }
    }
};

// ------------------------------------------------------------------------ small kernels
__global__ void set_i64_k(i64 *p, i64 a, i64 b, i64 c, i64 d) {
    p[0] = a;
    p[1] = b;
    p[2] = c;
    p[3] = d;
}
__global__ void fill_i64_k(i64 *p, i64 n, i64 v) {
    const i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (g < n) p[g] = v;
}

// The 64-byte block of partition results (longest bucket + four flag words) goes to pinned HOST memory with plain
// stores: the host then needs neither a copy engine nor a blit kernel -- which may queue behind the kernel that fills
// the chip -- to read it, only the event recorded behind this launch.
__global__ void publish_block_k(const unsigned long long *__restrict__ src, unsigned long long *__restrict__ host_dst) {
    if (threadIdx.x < 8) __hip_atomic_store(&host_dst[threadIdx.x], src[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

static inline unsigned grid_for(i64 n, int threads) { return (unsigned)std::max<i64>(1, ceil_div<i64>(n, threads)); }

// ------------------------------------------------------------------------ lifetime
extern "C" const char *esp_version(void) { return "esparse-hip 0.1 (gfx950)"; }

extern "C" const char *esp_last_error(const esp_handle *h) { return h ? h->err.c_str() : g_err.c_str(); }

static inline bool windowed(const esp_handle *h) { return h->win_excl && (h->wc0 > 0 || h->wc1 < h->n); }
// first entry and number of entries of the per-column arrays (colptr, colend: n+1 entries) a flush touches
static inline void col_range(const esp_handle *h, i64 *c0, i64 *cnt) {
    *c0 = windowed(h) ? h->wc0 : 0;
    *cnt = windowed(h) ? h->wc1 - h->wc0 + 1 : h->n + 1;
}
// colptr behind the window := nnz+1, if it was left stale by windowed flushes
static int32_t fix_tail(esp_handle *h) {
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

static int32_t init_empty_csc(esp_handle *h) {
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
    release_all(h);
    if (h->pin_scalar) (void)hipHostFree(h->pin_scalar);
    if (h->pin_mw) (void)hipHostFree(h->pin_mw);
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

static int32_t init_empty_csc(esp_handle *h);
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
    release_all(h);
    h->cap = 0;
    h->count = 0;
    h->chunk_cap = 0;
    h->chunk_pb = 0;
    pending_changed(h);
    h->csc_valid = false;
    h->ones_pending = false;
    h->tail_stale = false;
    h->csr_version = 0;
    h->csr_val_version = 0;
    return init_empty_csc(h);
}

static int32_t reserve_append(esp_handle *h, i64 add);
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
static int32_t reserve_append(esp_handle *h, i64 add) {
    // between esp_shard_assemble and esp_flush the pending entries are spread over the caller's receive buffers
    // (the handle's count is their logical total): nothing can be appended behind them
    if (h->part_assembled)
        FAIL(h, ESP_ERR_STATE, "append: the pending entries are assembled shard pieces (esp_shard_assemble): esp_flush first");
    // An append behind a bucket-ordered batch: the batch stays as it is (its 4-byte keys fill the front half of their
    // slots), the new entries follow it as packed keys, and the flush partitions only them (flush_pre_tail).  A shard's
    // batch, or force_path 19: back to packed keys first.
    h->pre_keep = h->pre.valid && h->pre.mw_P == 0 && h->force_path != 19 && h->count == h->pre.E + h->pre.tail;
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
static int32_t append_partitioned(esp_handle *h, const i64 *d_rows, const i64 *d_cols, const double *d_vals, int kind, int op, i64 count,
                                  bool *took);
static int32_t pack_device(esp_handle *h, const i64 *d_rows, const i64 *d_cols, const double *d_vals,
                           const uint8_t *d_kinds, int kind_all, int op, i64 count) {
    if (count == 0) return ESP_OK;
    if (!d_kinds && (kind_all < 0 || kind_all > 3)) FAIL(h, ESP_ERR_INVALID, "append: kind %d invalid", kind_all);
    if (op != ESP_OP_ADD && op != ESP_OP_SUB) FAIL(h, ESP_ERR_UNSUPPORTED, "append: op %d not supported on the device path", op);
    if (!d_kinds) {  // an empty buffer and one kind: the append is the partition (no packed stream is written)
        bool took = false;
        CK(append_partitioned(h, d_rows, d_cols, d_vals, kind_all, op, count, &took));
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
    if (h->pin_scalar[0] != ~0ull)
        FAIL(h, ESP_ERR_BOUNDS, "BoundsError: entry %llu of the batch has an index outside %lld x %lld (or a bad kind)",
             (unsigned long long)h->pin_scalar[0], (long long)h->m, (long long)h->n);
    if (!d_kinds) note_kind(h, kind_all, count);
    h->count += count;
    pending_changed(h);
    return ESP_OK;
}

static int32_t ensure_stage(esp_handle *h, esp_handle::StageArea &sa, i64 want) {
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

extern "C" int32_t esp_commit(esp_handle *h, int64_t count, int32_t kind_all, int32_t op) {
    if (!h) return ESP_ERR_INVALID;
    if (count < 0 || count > h->stage.cap) FAIL(h, ESP_ERR_INVALID, "esp_commit: count %lld exceeds the staged chunk", (long long)count);
    if (count == 0) return ESP_OK;
    (void)hipSetDevice(h->device);
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

// host memcpy with a few threads: one core moves ~10 GB/s, PCIe takes ~55 GB/s
static void par_memcpy(void *dst, const void *src, size_t bytes) {
    const size_t min_part = (size_t)4 << 20;
    int nt = (int)std::min<size_t>(4, bytes / min_part);
    if (nt <= 1) {
        memcpy(dst, src, bytes);
        return;
    }
    std::thread th[4];
    const size_t part = ((bytes / (size_t)nt) + 63) & ~(size_t)63;
    for (int i = 0; i < nt; i++) {
        const size_t o = (size_t)i * part;
        const size_t c = i == nt - 1 ? bytes - o : part;
        th[i] = std::thread([=] { memcpy((char *)dst + o, (const char *)src + o, c); });
    }
    for (int i = 0; i < nt; i++) th[i].join();
}

// Bulk append from host arrays: the batch goes through the pinned staging area in chunks, two halves in
// flight (the host copy of chunk i+1 overlaps the PCIe transfer and the pack kernel of chunk i); bounds
// are checked on the device and read back ONCE: the call is one batch, nothing is committed on error.
extern "C" int32_t esp_append_host(esp_handle *h, const int64_t *rows, const int64_t *cols, const double *vals,
                                   const uint8_t *kinds, int32_t kind_all, int32_t op, int64_t count) {
    if (!h || count < 0 || (count > 0 && (!rows || !cols || !vals))) return ESP_ERR_INVALID;
    if (count == 0) return ESP_OK;
    (void)hipSetDevice(h->device);
    if (!kinds && (kind_all < 0 || kind_all > 3)) FAIL(h, ESP_ERR_INVALID, "append: kind %d invalid", kind_all);
    if (op != ESP_OP_ADD && op != ESP_OP_SUB) FAIL(h, ESP_ERR_UNSUPPORTED, "append: op %d not supported on the device path", op);
    const i64 chunk = std::min<i64>((i64)1 << 22, std::max<i64>(count, 1));
    esp_handle::StageArea &sa = h->bulk;
    CK(ensure_stage(h, sa, 2 * chunk));
    CK(reserve_append(h, count));
    CK(ensure(h, h->misc, 256));
    unsigned long long *d_err = (unsigned long long *)h->misc.p;
    h->pin_scalar[0] = ~0ull;
    HIPCK(h, hipMemcpyAsync(d_err, h->pin_scalar, 8, hipMemcpyHostToDevice, h->stream));
    hipEvent_t done[2] = {nullptr, nullptr};
    for (int i = 0; i < 2; i++) HIPCK(h, hipEventCreateWithFlags(&done[i], hipEventDisableTiming));
    int32_t rc = ESP_OK;
    i64 it = 0;
    for (i64 off = 0; off < count && rc == ESP_OK; off += chunk, it++) {
        const i64 c = std::min<i64>(chunk, count - off);
        const int half = (int)(it & 1);
        const i64 so = half ? chunk : 0;  // this half of the staging arrays (host and device)
        if (it >= 2 && hipEventSynchronize(done[half]) != hipSuccess) rc = ESP_ERR_HIP;
        par_memcpy(sa.rows + so, rows + off, sizeof(i64) * (size_t)c);
        par_memcpy(sa.cols + so, cols + off, sizeof(i64) * (size_t)c);
        par_memcpy(sa.vals + so, vals + off, sizeof(double) * (size_t)c);
        if (kinds) memcpy(sa.kinds + so, kinds + off, (size_t)c);
        i64 *dr = (i64 *)sa.d_rows.p + so, *dc = (i64 *)sa.d_cols.p + so;
        double *dv = (double *)sa.d_vals.p + so;
        uint8_t *dk = (uint8_t *)sa.d_kinds.p + so;
        Span sp(h, ESP_ST_COPY);
        if (hipMemcpyAsync(dr, sa.rows + so, sizeof(i64) * (size_t)c, hipMemcpyHostToDevice, h->stream) != hipSuccess ||
            hipMemcpyAsync(dc, sa.cols + so, sizeof(i64) * (size_t)c, hipMemcpyHostToDevice, h->stream) != hipSuccess ||
            hipMemcpyAsync(dv, sa.vals + so, sizeof(double) * (size_t)c, hipMemcpyHostToDevice, h->stream) != hipSuccess ||
            (kinds && hipMemcpyAsync(dk, sa.kinds + so, (size_t)c, hipMemcpyHostToDevice, h->stream) != hipSuccess))
            rc = ESP_ERR_HIP;
        sp.add(kinds ? 4 : 3);
        hipLaunchKernelGGL(espgen::pack_k, dim3(grid_for(c, espgen::THREADS)), dim3(espgen::THREADS), 0, h->stream, (const i64 *)dr,
                           (const i64 *)dc, (const double *)dv, kinds ? (const uint8_t *)dk : nullptr, kind_all, op == ESP_OP_SUB ? 1 : 0, c,
                           h->m, h->n, h->L, (u64 *)h->keys.p + h->count + off, (double *)h->vals.p + h->count + off, d_err, off);
        (void)hipEventRecord(done[half], h->stream);
    }
    hipError_t e1 = hipMemcpyAsync(h->pin_scalar, d_err, 8, hipMemcpyDeviceToHost, h->stream);
    hipError_t e2 = hipStreamSynchronize(h->stream);
    for (int i = 0; i < 2; i++) (void)hipEventDestroy(done[i]);
    if (rc != ESP_OK || e1 != hipSuccess || e2 != hipSuccess || hipGetLastError() != hipSuccess)
        FAIL(h, ESP_ERR_HIP, "esp_append_host: transfer failed");
    if (h->pin_scalar[0] != ~0ull)
        FAIL(h, ESP_ERR_BOUNDS, "BoundsError: entry %llu of the batch has an index outside %lld x %lld (or a bad kind)",
             (unsigned long long)h->pin_scalar[0], (long long)h->m, (long long)h->n);
    if (!kinds) note_kind(h, kind_all, count);
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

// stream position of node g (0-based) in the k,j,i loop nest: host copy of espgen::fd_offset
static i64 fd_offset_host(i64 nx, i64 ny, i64 nz, i64 g) {
    const i64 N = nx * ny * nz;
    if (g >= N) {
        i64 E = 4 * (nx - 1) * ny * nz + (nx == 1 ? 1 : 2) * ny * nz;
        E += 4 * nx * (ny - 1) * nz + (ny > 2 ? 2 * nx * nz : 0);
        E += 4 * nx * ny * (nz - 1) + (nz > 2 ? 2 * nx * ny : 0);
        return E;
    }
    const i64 i = g % nx + 1, j = (g / nx) % ny + 1, k = g / (nx * ny) + 1;
    const i64 CX = 4 * (nx - 1) + (nx == 1 ? 1 : 2);
    const i64 CY = 4 * (ny - 1) + (ny > 2 ? 2 : 0);
    const i64 PX = 4 * (i - 1) + (i > 1 ? 1 : 0);
    const i64 PY = 4 * (j - 1) + ((ny > 2 && j > 1) ? 1 : 0);
    const i64 PZ = 4 * (k - 1) + ((nz > 2 && k > 1) ? 1 : 0);
    const i64 cy = (j < ny ? 4 : 0) + ((ny > 2 && (j == 1 || j == ny)) ? 1 : 0);
    const i64 cz = (k < nz ? 4 : 0) + ((nz > 2 && (k == 1 || k == nz)) ? 1 : 0);
    return (k - 1) * (ny * CX + nx * CY) + nx * ny * PZ + (j - 1) * CX + nx * PY + (j - 1) * nx * cz + PX + (i - 1) * (cy + cz);
}

extern "C" int32_t esp_generate_fdrand_range(esp_handle *h, int64_t nx, int64_t ny, int64_t nz, uint64_t seed, int32_t rand_mode,
                                             int32_t kind, int64_t node_begin, int64_t node_end) {
    if (!h) return ESP_ERR_INVALID;
    if (nx < 1 || ny < 1 || nz < 1) FAIL(h, ESP_ERR_INVALID, "fdrand: bad grid");
    const i64 N = nx * ny * nz;
    if (h->m != N || h->n != N) FAIL(h, ESP_ERR_INVALID, "Matrix size mismatch");  // sprand.jl:66-68
    if (kind != ESP_UPDATE && kind != ESP_RAWUPDATE && kind != ESP_COO) FAIL(h, ESP_ERR_INVALID, "fdrand: kind must be UPDATE, RAWUPDATE or COO");
    if (rand_mode < 0 || rand_mode > 2) FAIL(h, ESP_ERR_INVALID, "fdrand: rand_mode");
    if (node_begin < 0 || node_end > N || node_begin > node_end) FAIL(h, ESP_ERR_INVALID, "fdrand: node range");
    if (node_begin == node_end) return ESP_OK;
    (void)hipSetDevice(h->device);
    const i64 off_b = fd_offset_host(nx, ny, nz, node_begin);
    const i64 E = fd_offset_host(nx, ny, nz, node_end) - off_b;
    CK(reserve_append(h, E));
    espgen::FdArgs a;
    a.nx = nx;
    a.ny = ny;
    a.nz = nz;
    a.hx = 1.0 / (double)nx;
    a.hy = 1.0 / (double)ny;
    a.hz = 1.0 / (double)nz;
    a.seed = seed;
    a.rand_mode = rand_mode;
    a.kind = kind;
    a.total = E;
    a.g_begin = node_begin;
    a.g_end = node_end;
    a.off_begin = off_b;
    // (magic = 2^64 / d + 1: n / d = high half of magic * n for n, d < 2^32, d >= 2)
    a.fast = (N < ((i64)1 << 32) && nx >= 2) ? 1 : 0;
    a.magic_nx = a.fast ? ~0ull / (u64)nx + 1ull : 0;
    a.magic_nxny = a.fast ? ~0ull / (u64)(nx * ny) + 1ull : 0;
    a.L = h->L;
    a.keys = (u64 *)h->keys.p + h->count;
    a.vals = (double *)h->vals.p + h->count;
    CK(ensure(h, h->misc, 256));
    const dim3 grid(grid_for(node_end - node_begin, espgen::THREADS)), block(espgen::THREADS);
    // the append is the partition when the buffer is empty and the stream is one an assembly loop emits: COUNT launch
    // (ALU only), two tiny ranking launches, then every update goes straight to its bucket
    PartSetup ps;
    CK(prepart_begin(h, E, (i64)grid.x, kind, &ps));
    a.part = ps.out;
    bool took = false;
    if (ps.on) {
        {
            Span sp(h, ESP_ST_HIST);
            hipLaunchKernelGGL(espgen::fd_count_k, grid, block, 0, h->stream, a, ps.sink, ps.err);
            sp.add(1);
        }
        CK(prepart_rank(h, &ps));
        {
            Span sp(h, ESP_ST_APPEND);
            if (ps.out.k32)
                hipLaunchKernelGGL((espgen::fdrand_part_k<true, true>), grid, block, 0, h->stream, a);
            else if (ps.out.s32)
                hipLaunchKernelGGL((espgen::fdrand_part_k<true, false>), grid, block, 0, h->stream, a);
            else
                hipLaunchKernelGGL((espgen::fdrand_part_k<false, false>), grid, block, 0, h->stream, a);
            sp.add(1);
        }
        CK(prepart_finish(h, &ps, &took));
    }
    if (!took) {  // stream order (the PART launch left without a store when the stream turned out not to be pre-sorted)
        Span sp(h, ESP_ST_APPEND);
        hipLaunchKernelGGL(espgen::fdrand_k, grid, block, 0, h->stream, a);
        sp.add(1);
    }
    HIPCK(h, hipGetLastError());
    note_kind(h, kind, E);
    h->count += E;
    pending_changed(h);
    if (took) h->pre.valid = true;  // (else: whatever pending_changed left -- an earlier batch with this call as its tail)
    return ESP_OK;
}

extern "C" int32_t esp_generate_fdrand(esp_handle *h, int64_t nx, int64_t ny, int64_t nz, uint64_t seed,
                                       int32_t rand_mode, int32_t kind) {
    if (!h) return ESP_ERR_INVALID;
    return esp_generate_fdrand_range(h, nx, ny, nz, seed, rand_mode, kind, 0, nx * ny * nz);
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

static int32_t item_produce_fem(esp_handle *h, const espgen::FemArgs &fa, i64 E, bool *took);

extern "C" int32_t esp_generate_fem(esp_handle *h, int32_t dim, int64_t npd, uint64_t seed, int32_t order_mode) {
    if (!h) return ESP_ERR_INVALID;
    if ((dim != 2 && dim != 3) || npd < 2) FAIL(h, ESP_ERR_INVALID, "fem: dim must be 2 or 3 and npd >= 2");
    const i64 nn = dim == 2 ? npd * npd : npd * npd * npd;
    if (h->m != nn || h->n != nn) FAIL(h, ESP_ERR_INVALID, "Matrix size mismatch");
    (void)hipSetDevice(h->device);
    const i64 q = npd - 1;
    const i64 nc = dim == 2 ? 2 * q * q : 6 * q * q * q;
    const i64 E = nc * (dim + 1) * (dim + 2);
    CK(reserve_append(h, E));
    espgen::FemArgs a;
    a.dim = dim;
    a.npd = npd;
    a.ncells = nc;
    a.seed = seed;
    a.order_mode = order_mode;
    int bits = 2;
    while (((u64)1 << bits) < (u64)nc) bits += 2;
    a.bits = bits;
    espgen::fem_fill_magic(a);
    a.h = 1.0 / (double)(npd - 1);
    a.L = h->L;
    a.keys = (u64 *)h->keys.p + h->count;
    a.vals = (double *)h->vals.p + h->count;
    CK(ensure(h, h->misc, 256));
    const dim3 grid(grid_for(nc, espgen::FEM_CELLS)), block(espgen::FEM_CELLS);
    PartSetup ps;
    CK(prepart_begin(h, E, (i64)grid.x, ESP_RAWUPDATE, &ps));  // (a shuffled cell order fails the COUNT launch's digit limit)
    a.part = ps.out;
    bool took = false;
    if (ps.on) {
        {
            Span sp(h, ESP_ST_HIST);
            hipLaunchKernelGGL(espgen::fem_count_k, grid, block, 0, h->stream, a, ps.sink, ps.err);
            sp.add(1);
        }
        CK(prepart_rank(h, &ps));
        {
            Span sp(h, ESP_ST_APPEND);
            if (ps.out.k32)
                hipLaunchKernelGGL((espgen::fem_part_k<true, true>), grid, block, 0, h->stream, a);
            else if (ps.out.s32)
                hipLaunchKernelGGL((espgen::fem_part_k<true, false>), grid, block, 0, h->stream, a);
            else
                hipLaunchKernelGGL((espgen::fem_part_k<false, false>), grid, block, 0, h->stream, a);
            sp.add(1);
        }
        CK(prepart_finish(h, &ps, &took));
    }
    // a shuffled stream: the producer partitions its ITEMS and stores every update at its bucket position (femitems.hpp)
    if (!took) CK(item_produce_fem(h, a, E, &took));
    if (!took) {
        Span sp(h, ESP_ST_APPEND);
        hipLaunchKernelGGL(espgen::fem_k, grid, block, 0, h->stream, a);
        sp.add(1);
    }
    HIPCK(h, hipGetLastError());
    note_kind(h, ESP_RAWUPDATE, E);
    h->count += E;
    pending_changed(h);
    if (took) h->pre.valid = true;  // (else: whatever pending_changed left -- an earlier batch with this call as its tail)
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
    HIPCK(h, hipMemcpyAsync(h->colptr.p, colptr, sizeof(i64) * (size_t)(h->n + 1), hipMemcpyHostToDevice, h->stream));
    if (nnz > 0) {
        HIPCK(h, hipMemcpyAsync(h->rowval.p, rowval, sizeof(i64) * (size_t)nnz, hipMemcpyHostToDevice, h->stream));
        HIPCK(h, hipMemcpyAsync(h->nzval.p, nzval, sizeof(double) * (size_t)nnz, hipMemcpyHostToDevice, h->stream));
    }
    sp.add(3);
    HIPCK(h, hipStreamSynchronize(h->stream));
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
static int32_t d2h_pipelined(esp_handle *h, void *dst, const void *d_src, size_t bytes) {
    if (bytes == 0) return ESP_OK;
    const size_t small = (size_t)8 << 20;
    if (bytes <= small) {
        HIPCK(h, hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, h->stream));
        HIPCK(h, hipStreamSynchronize(h->stream));
        return ESP_OK;
    }
    CK(ensure_stage(h, h->bulk, (i64)1 << 22));  // rows / cols of the bulk area: 2 x 32 MiB pinned
    char *pin[2] = {(char *)h->bulk.rows, (char *)h->bulk.cols};
    const size_t chunk = (size_t)h->bulk.cap * 8;
    hipEvent_t ev[2] = {nullptr, nullptr};
    for (int i = 0; i < 2; i++) HIPCK(h, hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
    const size_t nchunks = (bytes + chunk - 1) / chunk;
    int32_t rc = ESP_OK;
    for (size_t c = 0; c <= nchunks && rc == ESP_OK; c++) {
        if (c < nchunks) {  // issue the transfer of chunk c
            const size_t o = c * chunk, len = std::min(chunk, bytes - o);
            if (hipMemcpyAsync(pin[c & 1], (const char *)d_src + o, len, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
                hipEventRecord(ev[c & 1], h->stream) != hipSuccess)
                rc = ESP_ERR_HIP;
        }
        if (c > 0 && rc == ESP_OK) {  // ... while chunk c-1 moves from its bounce buffer to the caller
            const size_t o = (c - 1) * chunk, len = std::min(chunk, bytes - o);
            if (hipEventSynchronize(ev[(c - 1) & 1]) != hipSuccess) rc = ESP_ERR_HIP;
            else par_memcpy((char *)dst + o, pin[(c - 1) & 1], len);
        }
    }
    (void)hipStreamSynchronize(h->stream);
    for (int i = 0; i < 2; i++) (void)hipEventDestroy(ev[i]);
    if (rc != ESP_OK) FAIL(h, ESP_ERR_HIP, "device-to-host transfer failed");
    return ESP_OK;
}

extern "C" int32_t esp_get_csc(esp_handle *h, int64_t *colptr, int64_t *rowval, double *nzval) {
    if (!h || !colptr) return ESP_ERR_INVALID;
    (void)hipSetDevice(h->device);
    CK(fix_tail(h));
    if (h->nnz > 0 && (!rowval || !nzval)) FAIL(h, ESP_ERR_INVALID, "esp_get_csc: rowval/nzval NULL with nnz>0");
    Span sp(h, ESP_ST_COPY);
    CK(d2h_pipelined(h, colptr, h->colptr.p, sizeof(i64) * (size_t)(h->n + 1)));
    if (h->nnz > 0) {
        CK(d2h_pipelined(h, rowval, h->rowval.p, sizeof(i64) * (size_t)h->nnz));
        CK(d2h_pipelined(h, nzval, h->nzval.p, sizeof(double) * (size_t)h->nnz));
    }
    sp.add(3);
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
    h->count = 0;
    pending_changed(h);
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

// exclusive scan helpers with scratch carved from h->misc
template <typename T, bool MAX>
static int32_t scan_inplace(esp_handle *h, T *data, i64 n, DevBuf &ws, int *launches) {
    CK(ensure(h, ws, sizeof(T) * (size_t)espscan::workspace_elems(n)));
    *launches += espscan::exclusive<T, MAX>(h->stream, data, data, n, (T *)ws.p);
    return ESP_OK;
}

extern "C" int32_t esp_dropzeros(esp_handle *h, int64_t *new_nnz) {
    if (!h) return ESP_ERR_INVALID;
    (void)hipSetDevice(h->device);
    CK(fix_tail(h));
    const i64 Z = h->nnz;
    if (Z == 0) {
        if (new_nnz) *new_nnz = 0;
        return ESP_OK;
    }
    if (Z >= 0xFFFFFFF0ll) FAIL(h, ESP_ERR_UNSUPPORTED, "dropzeros: nnz too large");
    CK(ensure(h, h->vals2, sizeof(u32) * (size_t)(Z + 1)));
    u32 *flag = (u32 *)h->vals2.p;
    hipLaunchKernelGGL(espfold::nonzero_flags_k, dim3(grid_for(Z + 1, 256)), dim3(256), 0, h->stream, (const double *)h->nzval.p, Z, flag);
    int l = 0;
    CK(scan_inplace<u32, false>(h, flag, Z + 1, h->hist, &l));
    HIPCK(h, hipMemcpyAsync(h->pin_scalar, flag + Z, 4, hipMemcpyDeviceToHost, h->stream));
    HIPCK(h, hipStreamSynchronize(h->stream));
    const i64 Zk = (i64) * (u32 *)h->pin_scalar;
    if (Zk != Z) {
        CK(ensure(h, h->rowval2, sizeof(i64) * (size_t)std::max<i64>(Zk, 1)));
        CK(ensure(h, h->nzval2, sizeof(double) * (size_t)std::max<i64>(Zk, 1)));
        hipLaunchKernelGGL(espfold::dropzeros_compact_k, dim3(grid_for(Z, 256)), dim3(256), 0, h->stream, (const i64 *)h->rowval.p,
                           (const double *)h->nzval.p, Z, flag, (i64 *)h->rowval2.p, (double *)h->nzval2.p);
        CK(ensure(h, h->colend, sizeof(i64) * (size_t)(h->n + 1)));
        hipLaunchKernelGGL(espfold::dropzeros_colptr_k, dim3(grid_for(h->n + 1, 256)), dim3(256), 0, h->stream, (const i64 *)h->colptr.p,
                           h->n + 1, flag, (i64 *)h->colend.p);
        std::swap(h->colptr, h->colend);
        std::swap(h->rowval, h->rowval2);
        std::swap(h->nzval, h->nzval2);
        h->nnz = Zk;
        h->pattern_version++, h->values_version++;
    }
    if (new_nnz) *new_nnz = h->nnz;
    return ESP_OK;
}

extern "C" int32_t esp_getindex(esp_handle *h, int64_t i, int64_t j, double *value, int32_t *found) {
    if (!h || !value) return ESP_ERR_INVALID;
    if (!(1 <= i && i <= h->m && 1 <= j && j <= h->n)) FAIL(h, ESP_ERR_BOUNDS, "BoundsError: (%lld,%lld) outside %lld x %lld", (long long)i, (long long)j, (long long)h->m, (long long)h->n);
    (void)hipSetDevice(h->device);
    CK(fix_tail(h));
    CK(ensure(h, h->misc, 256));
    espfold::Csc c{(const i64 *)h->colptr.p, (const i64 *)h->rowval.p, (double *)h->nzval.p, h->nnz};
    double *d_out = (double *)h->misc.p + 8;
    hipLaunchKernelGGL(espfold::getindex_k, dim3(1), dim3(1), 0, h->stream, c, i - 1, j - 1, d_out);
    HIPCK(h, hipMemcpyAsync(h->pin_scalar, d_out, 16, hipMemcpyDeviceToHost, h->stream));
    HIPCK(h, hipStreamSynchronize(h->stream));
    const double *r = (const double *)h->pin_scalar;
    *value = r[0];
    if (found) *found = r[1] != 0.0;
    return ESP_OK;
}

// ---- getindex(buffer, i, j): the value the pending entries alone give position (i,j) ---------------------------
// SparseMatrixLNK's getindex (sparsematrixlnk.jl:151-171) returns what the inserts so far left at (i,j), zero if
// there is no entry.  The device buffer holds the calls themselves: the matching ones are collected (buffer position,
// kind, value), ordered by position -- the call order, also in a bucket-ordered batch -- and folded by the state
// machine of fold.hpp.  A slow path by design (one pass over the pending keys per call): GenericExtendableSparseMatrixCSC
// reaches it for reads of positions that are not in the CSC yet (genericextendablesparsematrixcsc.jl:60-69).
constexpr int PENDING_MATCH_CAP = 2048;
__global__ void pending_matches_k(const u64 *__restrict__ keys, const double *__restrict__ vals, i64 E, u64 target,
                                  unsigned long long *__restrict__ count, u64 *__restrict__ mpos, double *__restrict__ mval) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 p = (i64)blockIdx.x * blockDim.x + threadIdx.x; p < E; p += stride) {
        const u64 k = keys[p];
        if ((k >> ESP_TAG_BITS) == target) {
            const unsigned long long at = atomicAdd(count, 1ull);
            if (at < (unsigned long long)PENDING_MATCH_CAP) {
                mpos[at] = ((u64)p << ESP_TAG_BITS) | (k & ESP_TAG_MASK);
                mval[at] = vals[p];
            }
        }
    }
}
__global__ __launch_bounds__(256) void pending_fold_k(const unsigned long long *__restrict__ count, const u64 *__restrict__ mpos,
                                                      const double *__restrict__ mval, double *__restrict__ out) {
    __shared__ u64 spos[PENDING_MATCH_CAP];
    __shared__ double sval[PENDING_MATCH_CAP];
    const int n = (int)min(*count, (unsigned long long)PENDING_MATCH_CAP);
    for (int q = threadIdx.x; q < n; q += 256) {  // rank sort by buffer position (positions are distinct)
        const u64 me = mpos[q];
        int r = 0;
        for (int o = 0; o < n; o++) r += mpos[o] < me ? 1 : 0;
        spos[r] = me;
        sval[r] = mval[q];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        bool present = false;
        double acc = 0.0;
        for (int q = 0; q < n; q++) espfold::fold_step(present, acc, (u32)(spos[q] & ESP_TAG_MASK), sval[q]);
        out[0] = present ? acc : 0.0;
        out[1] = present ? 1.0 : 0.0;
    }
}
extern "C" int32_t esp_pending_getindex(esp_handle *h, int64_t i, int64_t j, double *value, int32_t *found) {
    if (!h || !value) return ESP_ERR_INVALID;
    if (!(1 <= i && i <= h->m && 1 <= j && j <= h->n)) FAIL(h, ESP_ERR_BOUNDS, "BoundsError: (%lld,%lld) outside %lld x %lld", (long long)i, (long long)j, (long long)h->m, (long long)h->n);
    *value = 0.0;
    if (found) *found = 0;
    if (h->count == 0) return ESP_OK;
    (void)hipSetDevice(h->device);
    if (h->part_assembled) FAIL(h, ESP_ERR_STATE, "esp_pending_getindex: the pending entries are spread over shard pieces (flush first)");
    CK(pending_materialize(h));  // (packed keys)
    const size_t bytes = 64 + (sizeof(u64) + sizeof(double)) * (size_t)PENDING_MATCH_CAP;
    CK(ensure(h, h->heads, bytes));
    unsigned long long *cnt = (unsigned long long *)h->heads.p;
    double *d_out = (double *)h->heads.p + 2;
    u64 *mpos = (u64 *)((char *)h->heads.p + 64);
    double *mval = (double *)(mpos + PENDING_MATCH_CAP);
    HIPCK(h, hipMemsetAsync(cnt, 0, 64, h->stream));
    const u64 target = ((u64)(j - 1) << h->L.rb) | (u64)(i - 1);
    const unsigned grid = (unsigned)std::min<i64>(4096, std::max<i64>(1, ceil_div<i64>(h->count, 256)));
    hipLaunchKernelGGL(pending_matches_k, dim3(grid), dim3(256), 0, h->stream, (const u64 *)h->keys.p, (const double *)h->vals.p, h->count,
                       target, cnt, mpos, mval);
    hipLaunchKernelGGL(pending_fold_k, dim3(1), dim3(256), 0, h->stream, (const unsigned long long *)cnt, (const u64 *)mpos,
                       (const double *)mval, d_out);
    HIPCK(h, hipGetLastError());
    HIPCK(h, hipMemcpyAsync(h->pin_scalar, cnt, 32, hipMemcpyDeviceToHost, h->stream));
    HIPCK(h, hipStreamSynchronize(h->stream));
    if (h->pin_scalar[0] > (unsigned long long)PENDING_MATCH_CAP)
        FAIL(h, ESP_ERR_UNSUPPORTED, "esp_pending_getindex: more than %d pending updates of (%lld,%lld); flush first", PENDING_MATCH_CAP, (long long)i, (long long)j);
    const double *r = (const double *)(h->pin_scalar + 2);
    *value = r[0];
    if (found) *found = r[1] != 0.0;
    return ESP_OK;
}

extern "C" int32_t esp_pattern_hash(esp_handle *h, uint64_t *hash) {
    if (!h || !hash) return ESP_ERR_INVALID;
    (void)hipSetDevice(h->device);
    CK(fix_tail(h));
    CK(ensure(h, h->misc, 256));
    unsigned long long *acc = (unsigned long long *)h->misc.p + 16;
    HIPCK(h, hipMemsetAsync(acc, 0, 16, h->stream));
    const i64 work = std::max<i64>(h->n + 1, h->nnz);
    const unsigned grid = (unsigned)std::min<i64>(2048, std::max<i64>(1, ceil_div<i64>(work, espfold::THREADS)));
    hipLaunchKernelGGL(espfold::pattern_hash_k, dim3(grid), dim3(espfold::THREADS), 0, h->stream, (const i64 *)h->colptr.p, h->n + 1,
                       (const i64 *)h->rowval.p, h->nnz, acc);
    HIPCK(h, hipMemcpyAsync(h->pin_scalar, acc, 16, hipMemcpyDeviceToHost, h->stream));
    HIPCK(h, hipStreamSynchronize(h->stream));
    const u64 h1 = h->pin_scalar[0], h2 = h->pin_scalar[1];
    *hash = esp_mix64(h1 ^ esp_mix64(h2 + 0xD1B54A32D192ED03ull));
    return ESP_OK;
}

// ------------------------------------------------------------------------ sort
// One stable partition pass over S segments (device arrays seg_start/tile_first).
// max_tiles bounds the grid; the scanned histogram stays in h->hist.
static int32_t partition_pass(esp_handle *h, espradix::Pass &p, i64 max_tiles) {
    const int R = 1 << p.bits;
    const i64 hn = max_tiles * R;
    const size_t hist_bytes = sizeof(u64) * (size_t)(hn + espscan::workspace_elems(hn));
    CK(ensure(h, h->hist, hist_bytes));
    p.hist = (u64 *)h->hist.p;
    HIPCK(h, hipMemsetAsync(p.hist, 0, sizeof(u64) * (size_t)hn, h->stream));
    {
        Span sp(h, ESP_ST_HIST);
        hipLaunchKernelGGL(espradix::tile_hist_k, dim3(espradix::scatter_grid(max_tiles)), dim3(espradix::THREADS), 0, h->stream, p);
        sp.add(1);
    }
    {
        Span sp(h, ESP_ST_SCAN);
        sp.add(espscan::exclusive<u64, false>(h->stream, p.hist, p.hist, hn, p.hist + hn));
    }
    {
        Span sp(h, ESP_ST_SCATTER);
        if (p.bits > 8)
            hipLaunchKernelGGL((espradix::scatter_k<true>), dim3(espradix::scatter_grid(max_tiles)), dim3(espradix::THREADS), 0, h->stream, p);
        else
            hipLaunchKernelGGL((espradix::scatter_k<false>), dim3(espradix::scatter_grid(max_tiles)), dim3(espradix::THREADS), 0, h->stream, p);
        sp.add(1);
    }
    return ESP_OK;
}

// full stable LSD sort of the pending entries on their (col,row) bits.  Result in *sk/*sv.
static int32_t sort_pending_lsd(esp_handle *h, const u64 **sk, const double **sv) {
    const i64 E = h->count;
    CK(ensure(h, h->keys2, sizeof(u64) * (size_t)E));
    CK(ensure(h, h->vals2, sizeof(double) * (size_t)E));
    CK(ensure(h, h->segs, sizeof(i64) * 8));
    CK(ensure(h, h->misc, 256));
    const i64 T = ceil_div<i64>(E, espradix::TILE);
    i64 *segs = (i64 *)h->segs.p;
    hipLaunchKernelGGL(set_i64_k, dim3(1), dim3(1), 0, h->stream, segs, (i64)0, E, (i64)0, T);
    u64 *kin = (u64 *)h->keys.p, *kout = (u64 *)h->keys2.p;
    double *vin = (double *)h->vals.p, *vout = (double *)h->vals2.p;
    const int K = h->L.sort_bits();
    for (int done = 0; done < K; done += 8) {
        espradix::Pass p;
        p.keys_in = kin;
        p.vals_in = vin;
        p.keys_out = kout;
        p.vals_out = vout;
        p.seg_start = segs;
        p.tile_first = segs + 2;
        p.S = 1;
        p.owner_P = 0;
        p.owner_n = 1;
        p.colshift = 0;
        p.base = 0;
        p.span = ~0ull;
        p.err = (u32 *)h->misc.p + 60;
        p.shift = done;
        p.bits = std::min(8, K - done);
        CK(partition_pass(h, p, T));
        std::swap(kin, kout);
        std::swap(vin, vout);
    }
    HIPCK(h, hipGetLastError());
    *sk = kin;
    *sv = vin;
    // make keys/vals the scratch pair for the caller: after an odd number of passes the sorted
    // data lives in keys2/vals2
    if (kin != (u64 *)h->keys.p) {
        std::swap(h->keys, h->keys2);
        std::swap(h->vals, h->vals2);
        // capacities may differ: keep cap consistent with the smaller of the two
        h->cap = (i64)std::min(h->keys.bytes / sizeof(u64), h->vals.bytes / sizeof(double));
    }
    return ESP_OK;
}

// ------------------------------------------------------------------------ flush
// MSD plan: partition on the top key bits until every segment fits the LDS bucket kernel.
// Returns local_ok=false when the general (global LSD + fold_k) path must be used instead.
struct Sorted {
    const u64 *sk;
    const double *sv;
    bool in_primary;  // data in h->keys/vals (true) or h->keys2/vals2 (false)
    int S;
    const i64 *seg_start;
    int rem_bits;
    bool local_ok;
    bool all_update = false;  // PIECES: every entry of every piece is an UPDATE (esp_shard_assemble checked)
    int key_bytes = 8;  // 4: sk holds 32-bit keys (the bits below the prefix); every entry has the kind `kind`
    int kind = 0;
    i64 maxlen = esplocal::CAP;  // longest segment
    i64 total = -1;              // entries of all segments, if the caller knows (lets flush_local drop the segments behind the last column)
    int p32_piece = -1;          // PIECES: the piece that holds 4-byte keys of kind `kind` from position p32_lo on
    bool all32 = false;          // PIECES: EVERY piece holds 4-byte keys of kind `kind`
    i64 p32_lo = 0;
    // PIECES (partitioned shard exchange): segments are concatenations of per-source pieces
    int npieces = 0;
    const i64 *pstart = nullptr;
    const void *const *ptab = nullptr;
    bool has_base = false;  // the segments' key base, if it is not the handle's window / shard range
    u64 base = 0;
};

// ---- run lists of the pending entries (runpart.hpp) -------------------------------------------
// Persistent per-handle arrays: every chunk's runs, the digits' own run lists, the bucket totals.  They are
// filled by run_hist_k at flush time or by the COUNT launch of a producer whose append is the partition.
struct ChunkArrays {
    u32 *runs_d, *runs_c;
    u64 *nruns;
    unsigned long long *bucket_count;
    u32 *overflow;
    u32 *dcount;  // (directly in front of bucket_count: one memset clears both)
    u64 *dlist;
    u64 *coarse;  // totals of 256 digits each
    size_t clear_bytes;  // dcount .. bucket_count[NB]
};

static int32_t chunk_arrays(esp_handle *h, i64 Ccap, int pb, ChunkArrays *out) {
    const i64 NB = (i64)1 << pb;
    const i64 RM = Ccap * esprun::RMAX;
    size_t off = 0;
    auto carve = [&](size_t bytes) {
        size_t o = off;
        off += (bytes + 255) & ~(size_t)255;
        return o;
    };
    const size_t o_rd = carve(sizeof(u32) * (size_t)RM), o_rc = carve(sizeof(u32) * (size_t)RM);
    const size_t o_nr = carve(sizeof(u64) * (size_t)(Ccap + 1 + espscan::workspace_elems(Ccap + 1)));
    const size_t o_dc = carve(sizeof(u32) * (size_t)NB);
    const size_t o_bc = carve(sizeof(u64) * (size_t)(NB + 1 + espscan::workspace_elems(NB + 1)));
    const size_t o_ov = carve(64);
    const size_t o_dl = carve(sizeof(u64) * (size_t)NB * esprun::DCAP);
    const size_t o_co = carve(sizeof(u64) * (size_t)(NB / 256 + 2));
    if (h->chunkbuf.bytes < off || h->chunk_cap != Ccap || h->chunk_pb != pb) {
        CK(ensure(h, h->chunkbuf, off));
        h->chunk_cap = Ccap;
        h->chunk_pb = pb;
    }
    char *B = (char *)h->chunkbuf.p;
    out->runs_d = (u32 *)(B + o_rd);
    out->runs_c = (u32 *)(B + o_rc);
    out->nruns = (u64 *)(B + o_nr);
    out->bucket_count = (unsigned long long *)(B + o_bc);
    out->overflow = (u32 *)(B + o_ov);
    out->dcount = (u32 *)(B + o_dc);
    out->dlist = (u64 *)(B + o_dl);
    out->coarse = (u64 *)(B + o_co);
    // (a multiple of 256 bytes -- the runtime splits an odd-sized memset into two launches; what follows the totals
    // inside their carve is scan workspace)
    out->clear_bytes = ((o_bc + sizeof(u64) * (size_t)(NB + 1) + 255) & ~(size_t)255) - o_dc;
    return ESP_OK;
}

// The digits of a partition cut the 2^K keys of the window's bit range, of which only `span` exist (a matrix
// with 2^k + 1 columns fills half of it): the plan counts the entries as if the empty part were filled as well,
// so that the occupied buckets come out at the planned fill.
// planned average fill of a segment (fraction of the bucket kernel's capacity); ESP_PLAN_FILL overrides (experiments)
static double plan_fill() {
    static const double f = [] {
        const char *e = getenv("ESP_PLAN_FILL");
        const double v = e ? atof(e) : 0.9;
        return v > 0.1 && v <= 1.0 ? v : 0.9;
    }();
    return f;
}
// records a segment of the partition may hold: the bucket kernel's capacity, or what an item partition says (plan_cap)
static inline i64 seg_cap(const esp_handle *h) { return h->plan_cap > 0 ? h->plan_cap : (i64)esplocal::CAP; }
static double plan_entries(i64 E, int K, u64 span) {
    const double full = std::ldexp(1.0, K);
    return span > 0 && (double)span < full ? (double)E * full / (double)span : (double)E;
}
// bits the run-based pass would resolve for E pending entries in a K-bit key window holding `span` keys (0: not used)
static int plan_run_bits(i64 E, int K, u64 span) {
    int planned = 0;
    if (E > esplocal::CAP) {
        const double target = plan_fill() * esplocal::CAP, Ee = plan_entries(E, K, span);
        while (planned < K && Ee / (double)((i64)1 << planned) > target) planned++;
    }
    return planned > 8 ? std::min(planned, 20) : 0;
}
// prefix bits that bring the segments of E pending entries under the bucket kernel's capacity at the planned fill,
// corrected by what the handle's last flush saw (seen_spread)
static int plan_prefix_bits(const esp_handle *h, i64 E, int K, double *Ee_out) {
    int planned = 0;
    const double Ee = plan_entries(E, K, h->win_span);  // (see plan_entries: the window fills only part of its 2^K keys)
    if (E > seg_cap(h)) {
        double target = plan_fill() * (double)seg_cap(h);
        // (test hook: plan as if the bucket kernel took segments of this many entries -- many prefix bits, i.e. the 9-bit
        // passes, at sizes a CPU oracle can follow)
        if (const char *e = getenv("ESP_DEBUG_PLAN_CAP")) target = std::min(target, std::max(8.0, atof(e)));
        while (planned < K && Ee / (double)((i64)1 << planned) > target) planned++;
    }
    if (planned > 0 && h->seen_spread > 0.0 && h->seen_spread < 2.0 &&
        Ee / (double)((i64)1 << (planned - 1)) * h->seen_spread <= 0.98 * (double)seg_cap(h))
        planned--;  // (see seen_spread; a wrong guess costs one further pass and corrects itself)
    // ... and irregular data (the longest segment well above the average) gets the bits up front that the last flush
    // had to add in a further pass
    for (int extra = 0; extra < 3 && planned > 0 && planned < K && h->seen_spread >= 1.0 && h->seen_spread < 8.0 &&
                        Ee / (double)((i64)1 << planned) * h->seen_spread > (double)seg_cap(h);
         extra++)
        planned++;
    *Ee_out = Ee;
    return planned;
}
static int window_bits(const esp_handle *h) {
    int K = 1;
    while (K < 62 && ((u64)1 << K) < h->win_span) K++;
    return K;
}

// Single-pass partition on the top `pb` (9..20) bits of the key window, for pre-sorted streams
// (runpart.hpp).  *ok=false when some chunk holds too many distinct digits: nothing was moved and the
// caller uses the 8-bit passes.  On success kout/vout hold the partitioned entries, seg_out (NB+1
// entries, device) the bucket starts and tile_first_out the tile index of every bucket.
// several key windows side by side (column shards): see esprun::Args
struct MultiWin {
    int P;
    u32 nb;
    const u64 *d_base;
};

static int32_t aux_ready(esp_handle *h) {
    if (!h->aux) {
        // highest priority: its few tiny launches must get workgroup slots WHILE a kernel that fills the chip runs
        // on the main stream (at equal priority they were seen to start only after that kernel had drained)
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        HIPCK(h, hipStreamCreateWithPriority(&h->aux, hipStreamNonBlocking, greatest));
    }
    if (!h->aux_ev) HIPCK(h, hipEventCreateWithFlags(&h->aux_ev, hipEventDisableTiming));
    return ESP_OK;
}

// The plan of the partition by (owner, digit inside the owner's column range): every rank derives the same one from
// (n, P, entries_per_shard).  ok = false: small or odd problem (the plain exchange serves it).
static inline i64 shard_col0(i64 n, int P, int r) { return (i64)(((__int128)r * (__int128)n + P - 1) / P); }  // ceil(r*n/P)
struct MwPlan {
    bool ok = false;
    int K = 0, shift = 0, pb = 0;
    u64 nb64 = 0;
    i64 NB = 0;
    std::vector<u64> base;
};
static MwPlan shard_mw_plan(const esp_handle *h, int P, i64 entries_per_shard) {
    MwPlan m;
    m.base.resize((size_t)P);
    u64 maxspan = 1;
    for (int r = 0; r < P; r++) {
        const i64 c0 = shard_col0(h->n, P, r), c1 = shard_col0(h->n, P, r + 1);
        m.base[(size_t)r] = (u64)c0 << h->L.rb;
        maxspan = std::max(maxspan, (u64)(c1 - c0) << h->L.rb);
    }
    int K = 1;
    while (K < 62 && ((u64)1 << K) < maxspan) K++;
    const int pbw = plan_run_bits(std::max<i64>(entries_per_shard, 1), K, maxspan);
    if (pbw == 0 || K - pbw > esplocal::MAX_REM_BITS) return m;
    m.K = K;
    m.shift = K - pbw;
    m.nb64 = ((maxspan - 1) >> m.shift) + 1;
    m.NB = (i64)m.nb64 * P;
    if (m.NB > ((i64)1 << 24)) return m;
    m.pb = 1;
    while (((i64)1 << m.pb) < m.NB) m.pb++;
    m.ok = true;
    return m;
}

// ---- producer-side partition: host side (the kernels: runpart.hpp "the append IS the partition") ----------
// prepart_begin: a device-side producer is about to append E entries in `chunks` chunks (= its workgroups, at most
// esprun::TILE entries each) -- kind >= 0: all of that kind.  ps->on = true when the append can be the partition:
// empty buffer, a prefix of 9..20 bits that brings the buckets under the bucket kernel's capacity without reaching
// into the row bits, handle not driven through esp_shard_* (its flush partitions by owner first).  Clears the tables.
// force_path 16: never.
static int32_t prepart_begin(esp_handle *h, i64 E, i64 chunks, int kind, PartSetup *ps) {
    ps->on = false;
    memset(&ps->out, 0, sizeof ps->out);
    if (h->count != 0 || E <= esplocal::CAP || chunks >= ((i64)1 << 38)) return ESP_OK;
    if (h->force_path == 2 || h->force_path == 5 || h->force_path == 12 || h->force_path == 16) return ESP_OK;
    if (h->runs_skip > 0) return ESP_OK;  // (the handle's last streams were not pre-sorted: back-off, see sort_msd)
    int K, pb, shift;
    i64 NB;
    double Ee = 0.0;
    MwPlan mw;
    if (h->shard_user) {
        // a shard: partition by (owner, digit inside the owner's column range) -- what the next esp_shard_partition
        // would do in a pass of its own -- if the caller announced that call (esp_shard_plan)
        if (!h->shard_plan.valid || h->force_path == 11) return ESP_OK;
        const int P = h->shard_plan.P;
        if (P > esprun::MW_MAX || P > esplocal::MAX_PIECES || (double)h->n * (double)P >= 9.0e18) return ESP_OK;
        mw = shard_mw_plan(h, P, h->shard_plan.eps);
        if (!mw.ok || mw.shift < h->L.rb) return ESP_OK;
        K = mw.K, pb = mw.pb, shift = mw.shift, NB = mw.NB;
        const size_t o_cnt = 256 * 8;  // (the table layout of esp_shard_partition: window bases | owner offsets | counts)
        CK(ensure(h, h->parttab, o_cnt + sizeof(i64) * (size_t)(NB + 1)));
        // (the copy is asynchronous: its source lives in the handle, not in this function's frame -- and a flush that
        // still reads an earlier table from the same vector has been synchronised by then: every flush ends in a wait)
        if (!h->pin_mw) {
            HIPCK(h, hipHostMalloc((void **)&h->pin_mw, sizeof(u64) * esprun::MW_MAX, hipHostMallocDefault));
            HIPCK(h, hipEventCreateWithFlags(&h->pin_mw_done, hipEventDisableTiming));
        } else {
            HIPCK(h, hipEventSynchronize(h->pin_mw_done));  // (the previous upload has read the buffer)
        }
        memcpy(h->pin_mw, mw.base.data(), sizeof(u64) * (size_t)P);
        HIPCK(h, hipMemcpyAsync(h->parttab.p, h->pin_mw, sizeof(u64) * (size_t)P, hipMemcpyHostToDevice, h->stream));
        HIPCK(h, hipEventRecord(h->pin_mw_done, h->stream));
    } else {
        K = window_bits(h);
        pb = plan_prefix_bits(h, E, K, &Ee);
        shift = K - pb;
        if (pb <= 8 || pb > 20 || shift < h->L.rb || shift > esplocal::MAX_REM_BITS) return ESP_OK;
        NB = (i64)1 << pb;
    }
    ChunkArrays ca;
    CK(chunk_arrays(h, chunks + 64, pb, &ca));
    CK(ensure(h, h->runbuf, sizeof(i64) * (size_t)chunks * esprun::RMAX));
    CK(ensure(h, h->seg[1], sizeof(i64) * (size_t)(NB + 1)));
    CK(ensure(h, h->misc, 256));
    CK(aux_ready(h));
    unsigned long long *d_maxlen = (unsigned long long *)h->misc.p + 24;
    u32 *flags = (u32 *)h->misc.p + 60;
    HIPCK(h, hipMemsetAsync(d_maxlen, 0, 64, h->stream));  // maxlen .. the four flag words
    HIPCK(h, hipMemsetAsync(ca.dcount, 0, ca.clear_bytes, h->stream));
    ps->sink = esprun::RunSink{ca.runs_d, ca.runs_c, ca.nruns, ca.bucket_count, flags + 1, ca.dcount, ca.dlist};
    ps->err = flags;
    ps->K = K;
    ps->pb = pb;
    ps->kind = kind;
    ps->E = E;
    ps->chunks = chunks;
    ps->Ee = Ee;
    ps->NB = NB;
    if (mw.ok) {
        ps->mw_P = h->shard_plan.P;
        ps->mw_me = h->shard_plan.me;
        ps->mw_shift = mw.shift;
        ps->mw_nb = (u32)mw.nb64;
        ps->mw_eps = h->shard_plan.eps;
    }
    ps->seg_out = (i64 *)h->seg[1].p;
    ps->runs_off = (i64 *)h->runbuf.p;
    ps->bucket_count = ca.bucket_count;
    ps->coarse = ca.coarse;
    ps->dcount = ca.dcount;
    ps->dlist = ca.dlist;
    esprun::PartOut &o = ps->out;
    o.runs_d = ca.runs_d;
    o.runs_off = ps->runs_off;
    o.nruns = ca.nruns;
    o.flags = flags;
    o.maxlen = d_maxlen;
    o.cap = esplocal::CAP;
    // (a shard's ranges travel to other ranks as packed keys: 4-byte keys only without windows)
    o.k32 = (!mw.ok && kind >= 0 && h->force_path != 14 && shift <= 32) ? 1 : 0;
    o.s32 = shift <= 32 ? 1 : 0;
    // (a shard's own range never leaves the GPU: 4-byte keys there -- force_path 14: packed keys everywhere)
    o.own32 = (mw.ok && kind >= 0 && h->force_path != 14 && shift <= 32) ? 1 : 0;
    o.mw_me = ps->mw_me;
    o.own_lo = (const i64 *)h->seg[1].p + (size_t)ps->mw_me * (size_t)ps->mw_nb;
    o.mw_P = mw.ok ? ps->mw_P : 0;
    o.mw_nb = ps->mw_nb;
    o.mw_base = (const u64 *)h->parttab.p;
    if (mw.ok) {
        o.mw_own_base = mw.base[(size_t)ps->mw_me];
        o.mw_own_width = ps->mw_me + 1 < ps->mw_P ? mw.base[(size_t)ps->mw_me + 1] - o.mw_own_base : ~0ull - o.mw_own_base;
    }
    o.shift = shift;
    o.base = h->win_base;
    o.span = h->win_span;
    o.keys_out = (u64 *)h->keys.p;
    o.vals_out = (double *)h->vals.p;
    o.chunk_base = 0;
    ps->on = true;
    return ESP_OK;
}
// between the COUNT and the PART launch: bucket starts and run offsets (the two ranking launches of run_partition)
static int32_t prepart_rank(esp_handle *h, PartSetup *ps) {
    const i64 NB = ps->NB;
    unsigned long long *d_maxlen = (unsigned long long *)h->misc.p + 24;
    u32 *flags = (u32 *)h->misc.p + 60;
    {
        Span sp(h, ESP_ST_SCAN);
        const unsigned g = (unsigned)grid_for(NB + 1, esprun::THREADS);
        hipLaunchKernelGGL(esprun::run_coarse_k, dim3(g), dim3(esprun::THREADS), 0, h->stream, ps->bucket_count, NB, ps->coarse);
        hipLaunchKernelGGL(esprun::run_rank_k, dim3(g), dim3(esprun::THREADS), 0, h->stream, ps->bucket_count, (const u64 *)ps->coarse,
                           ps->dcount, ps->dlist, NB, ps->seg_out, ps->runs_off, d_maxlen, flags + 3);
        sp.add(2);
    }
    // (the host reads the flags while the PART launch runs)
    hipLaunchKernelGGL(publish_block_k, dim3(1), dim3(64), 0, h->stream, (const unsigned long long *)d_maxlen, h->pin_scalar);
    HIPCK(h, hipEventRecord(h->aux_ev, h->stream));
    return ESP_OK;
}
// after the PART launch was issued: *took = false when it left without a store (window error, a chunk with too many
// digits, a digit with too many runs: the caller issues the plain producer); else the handle's buffer is bucket-ordered
static int32_t prepart_finish(esp_handle *h, PartSetup *ps, bool *took) {
    *took = false;
    HIPCK(h, hipEventSynchronize(h->aux_ev));  // (publish_block_k has written the block to pin_scalar)
    const u32 f_err = (u32)h->pin_scalar[6], f_over = (u32)(h->pin_scalar[6] >> 32), f_many = (u32)(h->pin_scalar[7] >> 32);
    if (f_err | f_over | f_many) {
        // (flags[0] stays set for nobody: the plain producer follows and the flush's own partition checks the window)
        HIPCK(h, hipMemsetAsync((u32 *)h->misc.p + 60, 0, 16, h->stream));
        if (f_over | f_many) {  // not a pre-sorted stream: neither this handle's producers nor its next flushes try again soon
            h->runs_penalty = std::min(16, 2 * h->runs_penalty + 1);
            h->runs_skip = h->runs_penalty + 1;  // (+1: the flush of this very batch)
        }
        return ESP_OK;
    }
    if (ps->out.k32 && (i64)h->pin_scalar[0] > (i64)esplocal::CAP) return ESP_OK;  // (the K32 launch left without a store)
    esp_handle::PrePart &pp = h->pre;
    pp.K = ps->K;
    pp.pb = ps->pb;
    pp.maxlen = (i64)h->pin_scalar[0];
    pp.key_bytes = ps->out.k32 ? 4 : 8;
    pp.kind = ps->kind;
    pp.E = ps->E;
    pp.tail = 0;
    pp.base = h->win_base;
    pp.span = h->win_span;
    pp.Ee = ps->Ee;
    pp.mw_P = ps->mw_P;
    pp.mw_me = ps->mw_me;
    pp.mw_shift = ps->mw_shift;
    pp.mw_nb = ps->mw_nb;
    pp.mw_eps = ps->mw_eps;
    pp.own32 = ps->out.own32 != 0;
    *took = true;  // (the caller sets pre.valid once the entries are counted in)
    return ESP_OK;
}
// A bucket-ordered pending buffer is a valid pending buffer -- a stable permutation of the stream -- once its keys are
// packed keys again: every call that reads or extends the pending entries other than the flush they were written for
static int32_t pending_materialize(esp_handle *h) {
    if (!h->pre.valid) return ESP_OK;
    h->pre.valid = false;
    h->part_own32 = false;
    if (h->pre.mw_P > 0 && h->pre.own32 && h->count > 0) {
        // a shard's batch: the own range holds 4-byte keys -> packed keys; the other ranges are copied as they are
        const esp_handle::PrePart &pp = h->pre;
        const i64 nb = (i64)pp.mw_nb, d0 = (i64)pp.mw_me * nb;
        CK(ensure(h, h->keys2, std::max(h->keys.bytes, sizeof(u64) * (size_t)h->count)));
        std::vector<i64> lohi(2);
        HIPCK(h, hipMemcpyAsync(&lohi[0], (const i64 *)h->seg[1].p + d0, sizeof(i64), hipMemcpyDeviceToHost, h->stream));
        HIPCK(h, hipMemcpyAsync(&lohi[1], (const i64 *)h->seg[1].p + d0 + nb, sizeof(i64), hipMemcpyDeviceToHost, h->stream));
        HIPCK(h, hipStreamSynchronize(h->stream));
        const i64 lo = lohi[0], hi_ = lohi[1], E = pp.E;
        Span sp(h, ESP_ST_COPY);
        if (lo > 0) HIPCK(h, hipMemcpyAsync(h->keys2.p, h->keys.p, sizeof(u64) * (size_t)lo, hipMemcpyDeviceToDevice, h->stream));
        if (E > hi_)
            HIPCK(h, hipMemcpyAsync((u64 *)h->keys2.p + hi_, (const u64 *)h->keys.p + hi_, sizeof(u64) * (size_t)(E - hi_), hipMemcpyDeviceToDevice, h->stream));
        const u64 base = (u64)shard_col0(h->n, pp.mw_P, pp.mw_me) << h->L.rb;
        hipLaunchKernelGGL(esprun::expand_own_keys_k, dim3((unsigned)nb), dim3(esprun::THREADS), 0, h->stream, (const u64 *)h->keys.p,
                           (const i64 *)h->seg[1].p, d0, pp.mw_shift, base, (u32)pp.kind, (u64 *)h->keys2.p);
        sp.add(3);
        HIPCK(h, hipGetLastError());
        std::swap(h->keys, h->keys2);
        h->cap = (i64)std::min(h->keys.bytes / sizeof(u64), h->vals.bytes / sizeof(double));
        return ESP_OK;
    }
    if (h->pre.key_bytes != 4 || h->count == 0) return ESP_OK;
    const esp_handle::PrePart &pp = h->pre;
    CK(ensure(h, h->keys2, std::max(h->keys.bytes, sizeof(u64) * (size_t)h->count)));
    Span sp(h, ESP_ST_COPY);
    hipLaunchKernelGGL(esprun::expand_keys_k, dim3((unsigned)((i64)1 << pp.pb)), dim3(esprun::THREADS), 0, h->stream, (const u32 *)h->keys.p,
                       (const i64 *)h->seg[1].p, pp.K - pp.pb, pp.base, (u32)pp.kind, (u64 *)h->keys2.p);
    sp.add(1);
    if (h->count > pp.E) {  // (the packed entries behind the batch)
        HIPCK(h, hipMemcpyAsync((u64 *)h->keys2.p + pp.E, (const u64 *)h->keys.p + pp.E, sizeof(u64) * (size_t)(h->count - pp.E),
                                hipMemcpyDeviceToDevice, h->stream));
        sp.add(1);
    }
    HIPCK(h, hipGetLastError());
    std::swap(h->keys, h->keys2);
    h->cap = (i64)std::min(h->keys.bytes / sizeof(u64), h->vals.bytes / sizeof(double));
    return ESP_OK;
}

// mw != nullptr: buckets = mw->P * mw->nb (window r = digits [r*nb, (r+1)*nb)), pb = bits covering them
// triplets of one kind as the source of a partition (esprun::Args::raw_*)
struct RawSource {
    const i64 *rows, *cols;
    int kind, negate;
    unsigned long long *d_err;
};
static int32_t run_partition(esp_handle *h, const u64 *kin, const double *vin, u64 *kout, double *vout, int K, int pb,
                             i64 *seg_out, u64 *tile_first_out, bool *tiles_ready, bool *ok, i64 *maxlen_out,
                             const MultiWin *mw = nullptr, int mw_shift = 0, bool allow_k32 = false, int *key_bytes_out = nullptr,
                             i64 E_in = -1, const RawSource *raw = nullptr) {
    const i64 E = E_in >= 0 ? E_in : h->count;  // (E_in: the entries behind a producer's batch, flush_pre_tail; a raw batch)
    const i64 NB = mw ? (i64)mw->P * (i64)mw->nb : (i64)1 << pb;
    CK(ensure(h, h->misc, 256));
    const i64 C = ceil_div<i64>(E, esprun::TILE);
    ChunkArrays ca;
    CK(chunk_arrays(h, C + 64, pb, &ca));
    const i64 RM = C * esprun::RMAX;
    // scratch of this call
    size_t off = 0;
    auto carve = [&](size_t bytes) {
        size_t o = off;
        off += (bytes + 255) & ~(size_t)255;
        return o;
    };
    const size_t o_ro = carve(sizeof(i64) * (size_t)RM);
    const size_t o_hd = carve(sizeof(u64) * (size_t)NB);
    const size_t o_lk = carve(sizeof(u64) * (size_t)RM), o_lk2 = carve(sizeof(u64) * (size_t)RM);
    const size_t o_lv = carve(sizeof(double) * (size_t)RM), o_lv2 = carve(sizeof(double) * (size_t)RM);
    const size_t o_sc = carve(sizeof(u64) * (size_t)(RM + 1 + espscan::workspace_elems(RM + 1)));
    CK(ensure(h, h->runbuf, off));
    char *B = (char *)h->runbuf.p;
    esprun::Args a;
    a.keys_in = kin;
    a.vals_in = vin;
    a.keys_out = kout;
    a.vals_out = vout;
    a.E = E;
    a.shift = mw ? mw_shift : K - pb;
    a.base = h->win_base;
    a.span = h->win_span;
    a.mw_P = mw ? mw->P : 0;
    a.mw_nb = mw ? mw->nb : 0;
    a.mw_base = mw ? mw->d_base : nullptr;
    if (raw) {
        a.raw_rows = raw->rows, a.raw_cols = raw->cols;
        a.raw_m = h->m, a.raw_n = h->n;
        a.raw_rb = h->L.rb, a.raw_kind = raw->kind, a.raw_negate = raw->negate;
        a.raw_err = raw->d_err;
    }
    a.err = (u32 *)h->misc.p + 60;
    a.overflow = ca.overflow;
    a.runs_d = ca.runs_d;
    a.runs_c = ca.runs_c;
    a.runs_off = (i64 *)(B + o_ro);
    a.nruns = ca.nruns;
    a.bucket_count = ca.bucket_count;
    u64 *nruns = a.nruns, *bstart = (u64 *)ca.bucket_count, *head = (u64 *)(B + o_hd);
    u64 *lk = (u64 *)(B + o_lk), *lk2 = (u64 *)(B + o_lk2), *sc = (u64 *)(B + o_sc);
    double *lv = (double *)(B + o_lv), *lv2 = (double *)(B + o_lv2);
    // flags[0] window error, [1] a chunk with too many digits, [2] (run-list sort passes), [3] a digit with too
    // many runs; maxlen in front of them: one 64-byte block the CALLER zeroed
    unsigned long long *d_maxlen = (unsigned long long *)h->misc.p + 24;
    u32 *flags = (u32 *)h->misc.p + 60;
    a.dcount = nullptr;
    a.dlist = nullptr;
    a.nruns_raw = 0;
    a.flags = nullptr;
    a.maxlen = nullptr;
    a.cap = 0;
    if (key_bytes_out) *key_bytes_out = 8;
    *tiles_ready = false;
    // ranked: every digit collects its own runs, ONE kernel turns them into run offsets (run_rank_k), the
    // scatter kernel follows without a host round trip (force_path 12: the radix-ordered run list instead)
    const bool ranked = h->force_path != 12;
    h->last_run_order = 2;
    if (ranked) CK(aux_ready(h));
    {
        a.overflow = flags + 1;
        if (ranked) {
            a.dcount = ca.dcount;
            a.dlist = ca.dlist;
            HIPCK(h, hipMemsetAsync(ca.dcount, 0, ca.clear_bytes, h->stream));
        } else {
            HIPCK(h, hipMemsetAsync(bstart, 0, sizeof(u64) * (size_t)(NB + 1), h->stream));
        }
        Span sp(h, ESP_ST_HIST);
        if (raw)
            hipLaunchKernelGGL((esprun::run_hist_k<false, true>), dim3((unsigned)C), dim3(esprun::THREADS), 0, h->stream, a, (i64)0);
        else if (mw)
            hipLaunchKernelGGL((esprun::run_hist_k<true>), dim3((unsigned)C), dim3(esprun::THREADS), 0, h->stream, a, (i64)0);
        else
            hipLaunchKernelGGL((esprun::run_hist_k<false>), dim3((unsigned)C), dim3(esprun::THREADS), 0, h->stream, a, (i64)0);
        sp.add(1);
    }
    if (ranked) {
        {
            Span sp(h, ESP_ST_SCAN);
            const unsigned g = (unsigned)grid_for(NB + 1, esprun::THREADS);
            u64 *coarse = ca.coarse;
            hipLaunchKernelGGL(esprun::run_coarse_k, dim3(g), dim3(esprun::THREADS), 0, h->stream, (const unsigned long long *)ca.bucket_count,
                               NB, coarse);
            hipLaunchKernelGGL(esprun::run_rank_k, dim3(g), dim3(esprun::THREADS), 0, h->stream, (const unsigned long long *)ca.bucket_count,
                               (const u64 *)coarse, (const u32 *)ca.dcount, (const u64 *)ca.dlist, NB, seg_out, a.runs_off, d_maxlen,
                               flags + 3);
            sp.add(2);
        }
        // the host reads the flags on the second stream while the scatter kernel (which leaves at once when one
        // of them is set) already runs
        hipLaunchKernelGGL(publish_block_k, dim3(1), dim3(64), 0, h->stream, (const unsigned long long *)d_maxlen, h->pin_scalar);
        HIPCK(h, hipEventRecord(h->aux_ev, h->stream));
        a.nruns_raw = 1;
        a.flags = flags;
        // 4-byte keys for the bucket kernel: one kind for all pending entries, <= 32 key bits below the prefix, no
        // further pass (force_path 14: packed keys always)
        const bool k32 = allow_k32 && key_bytes_out && !mw && h->force_path != 14 &&
                         (raw || (h->kind_uniform >= 0 && h->kind_noted == h->count)) && a.shift <= 32;
        a.maxlen = d_maxlen;
        a.cap = esplocal::CAP;
        {
            Span sp(h, ESP_ST_SCATTER);
            if (raw && k32)
                hipLaunchKernelGGL((esprun::run_scatter_k<false, true, true>), dim3((unsigned)C), dim3(esprun::THREADS), 0, h->stream, a);
            else if (raw)
                hipLaunchKernelGGL((esprun::run_scatter_k<false, false, true>), dim3((unsigned)C), dim3(esprun::THREADS), 0, h->stream, a);
            else if (mw)
                hipLaunchKernelGGL((esprun::run_scatter_k<true, false>), dim3((unsigned)C), dim3(esprun::THREADS), 0, h->stream, a);
            else if (k32)
                hipLaunchKernelGGL((esprun::run_scatter_k<false, true>), dim3((unsigned)C), dim3(esprun::THREADS), 0, h->stream, a);
            else
                hipLaunchKernelGGL((esprun::run_scatter_k<false, false>), dim3((unsigned)C), dim3(esprun::THREADS), 0, h->stream, a);
            sp.add(1);
        }
        HIPCK(h, hipEventSynchronize(h->aux_ev));  // (publish_block_k has written the block to pin_scalar)
        const u32 f_err = (u32)h->pin_scalar[6], f_over = (u32)(h->pin_scalar[6] >> 32), f_many = (u32)(h->pin_scalar[7] >> 32);
        if (f_over) {
            *ok = false;
            return ESP_OK;
        }
        if (f_err) FAIL(h, ESP_ERR_STATE, "esp_flush: a pending entry lies outside the declared column window (partition)");
        h->last_run_order = f_many ? 3 : 1;
        if (!f_many) {
            if (key_bytes_out) *key_bytes_out = (k32 && (i64)h->pin_scalar[0] <= (i64)esplocal::CAP) ? 4 : 8;
            *maxlen_out = (i64)h->pin_scalar[0];
            HIPCK(h, hipGetLastError());
            *ok = true;
            return ESP_OK;
        }
        // some digit has more runs than its list holds (nothing was moved): order the run list with the radix passes
        a.nruns_raw = 0;
        a.flags = nullptr;
    }
    HIPCK(h, hipMemsetAsync(nruns + C, 0, sizeof(u64), h->stream));
    {
        Span sp(h, ESP_ST_SCAN);
        sp.add(espscan::exclusive<u64, false>(h->stream, nruns, nruns, C + 1, nruns + C + 1));
        sp.add(espscan::exclusive<u64, false>(h->stream, bstart, bstart, NB + 1, bstart + NB + 1));
    }
    // bucket starts are final here: tiles per bucket and the longest bucket come with the same sync
    {
        Span sp(h, ESP_ST_SCAN);
        HIPCK(h, hipMemcpyAsync(seg_out, bstart, sizeof(i64) * (size_t)(NB + 1), hipMemcpyDeviceToDevice, h->stream));
        HIPCK(h, hipMemsetAsync(d_maxlen, 0, 8, h->stream));
        hipLaunchKernelGGL(espradix::seg_tiles_k, dim3(grid_for(NB + 1, 256)), dim3(256), 0, h->stream, (const i64 *)seg_out, NB,
                           (i64)espradix::TILE, tile_first_out, d_maxlen);
        sp.add(1 + espscan::exclusive<u64, false>(h->stream, tile_first_out, tile_first_out, NB + 1, tile_first_out + NB + 1));
    }
    *tiles_ready = true;
    HIPCK(h, hipMemcpyAsync(h->pin_scalar, a.overflow, 4, hipMemcpyDeviceToHost, h->stream));
    HIPCK(h, hipMemcpyAsync(h->pin_scalar + 1, nruns + C, 8, hipMemcpyDeviceToHost, h->stream));
    HIPCK(h, hipMemcpyAsync(h->pin_scalar + 2, d_maxlen, 8, hipMemcpyDeviceToHost, h->stream));
    HIPCK(h, hipMemcpyAsync(h->pin_scalar + 3, a.err, 4, hipMemcpyDeviceToHost, h->stream));
    HIPCK(h, hipStreamSynchronize(h->stream));
    if ((u32)h->pin_scalar[0]) {
        *ok = false;
        return ESP_OK;
    }
    if ((u32)h->pin_scalar[3]) FAIL(h, ESP_ERR_STATE, "esp_flush: a pending entry lies outside the declared column window (partition)");
    const i64 R = (i64)h->pin_scalar[1];
    *maxlen_out = (i64)h->pin_scalar[2];
    {
        Span sp(h, ESP_ST_SCAN);
        hipLaunchKernelGGL(esprun::run_pack_k, dim3(grid_for(RM, 256)), dim3(256), 0, h->stream, (const u32 *)a.runs_d, (const u32 *)a.runs_c,
                           (const u64 *)nruns, C, lk, lv);
        sp.add(1);
    }
    // stable sort of the run list by digit with the ordinary 8-bit passes (chunk order is kept)
    {
        CK(ensure(h, h->segs, sizeof(i64) * 8));
        i64 *segs = (i64 *)h->segs.p;
        const i64 TR = ceil_div<i64>(R, espradix::TILE);
        hipLaunchKernelGGL(set_i64_k, dim3(1), dim3(1), 0, h->stream, segs, (i64)0, R, (i64)0, TR);
        u64 *ki = lk, *ko = lk2;
        double *vi = lv, *vo = lv2;
        for (int done = 0; done < pb; done += 8) {
            espradix::Pass p;
            p.keys_in = ki;
            p.vals_in = vi;
            p.keys_out = ko;
            p.vals_out = vo;
            p.seg_start = segs;
            p.tile_first = segs + 2;
            p.S = 1;
            p.owner_P = 0;
            p.owner_n = 1;
            p.colshift = 0;
            p.base = 0;
            p.span = ~0ull;
            p.err = (u32 *)h->misc.p + 62;
            p.shift = done;
            p.bits = std::min(8, pb - done);
            CK(partition_pass(h, p, TR));
            std::swap(ki, ko);
            std::swap(vi, vo);
        }
        lk = ki;
        lv = vi;
    }
    {
        Span sp(h, ESP_ST_SCAN);
        hipLaunchKernelGGL(esprun::run_counts_k, dim3(grid_for(R + 1, 256)), dim3(256), 0, h->stream, (const double *)lv, R, sc);
        sp.add(1 + espscan::exclusive<u64, false>(h->stream, sc, sc, R + 1, sc + R + 1));
        hipLaunchKernelGGL(esprun::run_heads_k, dim3(grid_for(R, 256)), dim3(256), 0, h->stream, (const u64 *)lk, (const u64 *)sc, R, head);
        hipLaunchKernelGGL(esprun::run_offsets_k, dim3(grid_for(R, 256)), dim3(256), 0, h->stream, (const u64 *)lk, (const double *)lv,
                           (const u64 *)sc, (const u64 *)head, (const u64 *)bstart, R, a.runs_off);
        sp.add(2);
    }
    {
        Span sp(h, ESP_ST_SCATTER);
        if (raw)
            hipLaunchKernelGGL((esprun::run_scatter_k<false, false, true>), dim3((unsigned)C), dim3(esprun::THREADS), 0, h->stream, a);
        else if (mw)
            hipLaunchKernelGGL((esprun::run_scatter_k<true, false>), dim3((unsigned)C), dim3(esprun::THREADS), 0, h->stream, a);
        else
            hipLaunchKernelGGL((esprun::run_scatter_k<false, false>), dim3((unsigned)C), dim3(esprun::THREADS), 0, h->stream, a);
        sp.add(1);
    }
    HIPCK(h, hipGetLastError());
    *ok = true;
    return ESP_OK;
}

// esp_append_device / esp_append_host / esp_commit on an EMPTY buffer, all entries of one kind: the run-based single pass
// (runpart.hpp) reads the caller's triplets directly -- a count pass over the columns, then one kernel that reads rows,
// columns and values and stores key and value at their bucket position (4-byte keys when they fit).  What used to be
// pack (24 B read, 16 B written) + histogram (8 B) + scatter (16 B + 12 B) per entry is 8 B + 24 B read, 12 B written,
// and the flush starts at the bucket kernel (h->pre, as after a device-side producer).  *took = false: the stream is
// no pre-sorted one (or the plan does not apply): nothing was appended, the caller packs in stream order.
static int32_t append_partitioned(esp_handle *h, const i64 *d_rows, const i64 *d_cols, const double *d_vals, int kind, int op, i64 count,
                                  bool *took) {
    *took = false;
    if (h->count != 0 || count <= esplocal::CAP || h->shard_user || h->runs_skip > 0) return ESP_OK;
    // (test hooks that pin another path; 27: this one off)
    if (h->force_path == 2 || h->force_path == 5 || h->force_path == 12 || h->force_path == 16 || h->force_path == 19 || h->force_path == 27)
        return ESP_OK;
    const int K = window_bits(h);
    double Ee = 0.0;
    const int pb = plan_prefix_bits(h, count, K, &Ee);
    const int shift = K - pb;
    if (pb <= 8 || pb > 20 || shift < h->L.rb || shift > esplocal::MAX_REM_BITS) return ESP_OK;
    CK(reserve_append(h, count));
    const i64 NB = (i64)1 << pb;
    CK(ensure(h, h->seg[1], sizeof(i64) * (size_t)(NB + 1)));
    CK(ensure(h, h->tilef[1], sizeof(u64) * (size_t)(NB + 1 + espscan::workspace_elems(NB + 1))));
    CK(ensure(h, h->misc, 256));
    unsigned long long *d_err = (unsigned long long *)h->misc.p;
    h->pin_scalar[0] = ~0ull;
    HIPCK(h, hipMemcpyAsync(d_err, h->pin_scalar, 8, hipMemcpyHostToDevice, h->stream));
    HIPCK(h, hipMemsetAsync((unsigned long long *)h->misc.p + 24, 0, 64, h->stream));  // maxlen .. the four flag words (run_partition)
    const RawSource raw{d_rows, d_cols, kind, (op == ESP_OP_SUB && kind != ESP_SET) ? 1 : 0, d_err};
    bool tr = false, ok = false;
    i64 ml = count;
    int kb = 8;
    CK(run_partition(h, nullptr, d_vals, (u64 *)h->keys.p, (double *)h->vals.p, K, pb, (i64 *)h->seg[1].p, (u64 *)h->tilef[1].p, &tr, &ok, &ml,
                     nullptr, 0, /*allow_k32=*/true, &kb, count, &raw));
    HIPCK(h, hipMemcpyAsync(h->pin_scalar, d_err, 8, hipMemcpyDeviceToHost, h->stream));
    HIPCK(h, hipStreamSynchronize(h->stream));
    if (h->pin_scalar[0] != ~0ull)
        FAIL(h, ESP_ERR_BOUNDS, "BoundsError: entry %llu of the batch has an index outside %lld x %lld (or a bad kind)",
             (unsigned long long)h->pin_scalar[0], (long long)h->m, (long long)h->n);
    if (!ok) {  // not a pre-sorted stream: neither this handle's appends nor its next flushes try again soon
        h->runs_penalty = std::min(16, 2 * h->runs_penalty + 1);
        h->runs_skip = h->runs_penalty + 1;
        return ESP_OK;
    }
    h->runs_penalty = 0;
    esp_handle::PrePart &pp = h->pre;
    pp.K = K;
    pp.pb = pb;
    pp.maxlen = ml;
    pp.key_bytes = kb;
    pp.kind = kind;
    pp.E = count;
    pp.tail = 0;
    pp.base = h->win_base;
    pp.span = h->win_span;
    pp.Ee = Ee;
    pp.mw_P = 0, pp.mw_me = 0, pp.mw_shift = 0, pp.mw_nb = 0, pp.mw_eps = 0;
    pp.own32 = false;
    note_kind(h, kind, count);
    h->count += count;
    pending_changed(h);
    h->pre.valid = true;
    *took = true;
    return ESP_OK;
}

static int32_t sort_msd(esp_handle *h, Sorted *out) {
    const i64 E = h->count;
    // sort bits of the key window: (key>>2) - win_base lies in [0, win_span)
    int K = 1;
    while (K < 62 && ((u64)1 << K) < h->win_span) K++;
    CK(ensure(h, h->keys2, sizeof(u64) * (size_t)E));
    CK(ensure(h, h->vals2, sizeof(double) * (size_t)E));
    CK(ensure(h, h->misc, 256));
    unsigned long long *d_maxlen = (unsigned long long *)h->misc.p + 24;
    u32 *d_werr = (u32 *)h->misc.p + 60;
    HIPCK(h, hipMemsetAsync(d_maxlen, 0, 64, h->stream));  // maxlen .. the four flag words (run_partition)

    double Ee = 0.0;
    int planned = plan_prefix_bits(h, E, K, &Ee);
    const int planned_run = planned;  // the run-based pass takes up to 20 bits at once: no need to be tight
    // one bit short of a whole number of 8-bit passes: an average fill of up to 95 % is worth
    // trying with one pass less (the longest segment is checked after the planned passes and a
    // further pass is added only if a segment really overflows)
    if (planned > 8 && planned % 8 == 1 && Ee / (double)((i64)1 << (planned - 1)) <= 0.95 * (double)seg_cap(h)) planned--;
    // (digits of 9 bits only where they save a whole pass -- 17 or 18 bits in two passes: a tile then holds 8 entries per
    // digit instead of 16; force_path 23: never)
    const int npass8 = (planned + 7) / 8, npass9 = (planned + espradix::MAX_BITS - 1) / espradix::MAX_BITS;
    const int npass = (npass9 < npass8 && h->force_path != 23) ? npass9 : npass8;

    int cur = 0, S = 1, done = 0;
    CK(ensure(h, h->seg[0], sizeof(i64) * 4));
    CK(ensure(h, h->tilef[0], sizeof(u64) * (size_t)(4 + espscan::workspace_elems(4))));
    const i64 T = ceil_div<i64>(E, espradix::TILE);
    bool tiles_ready = false;  // seg[cur] has its tile table in tilef[cur] (only the 8-bit passes need one)
    u64 *kin = (u64 *)h->keys.p, *kout = (u64 *)h->keys2.p;
    double *vin = (double *)h->vals.p, *vout = (double *)h->vals2.p;
    i64 maxlen = E;
    bool ok = true;
    int pass_idx = 0;
    int npass_eff = npass;
    bool window_checked = false;
    h->last_partition = 2;
    // pre-sorted streams: the first (up to) 16 bits in ONE pass (runpart.hpp)
    if (planned_run > 8 && h->force_path != 5 && !h->item_mode) {
        if (h->runs_skip > 0) {
            h->runs_skip--;
        } else {
            const int pb = std::min(planned_run, 20);
            const int S2 = 1 << pb;
            CK(ensure(h, h->seg[1], sizeof(i64) * (size_t)(S2 + 1)));
            CK(ensure(h, h->tilef[1], sizeof(u64) * (size_t)(S2 + 1 + espscan::workspace_elems(S2 + 1))));
            bool took = false;
            i64 ml = E;
            bool tr = false;
            int kb = 8;
            CK(run_partition(h, kin, vin, kout, vout, K, pb, (i64 *)h->seg[1].p, (u64 *)h->tilef[1].p, &tr, &took, &ml, nullptr, 0,
                             /*allow_k32=*/pb >= planned_run, &kb));
            if (took) {
                tiles_ready = tr;
                out->key_bytes = kb;
                out->kind = h->kind_uniform;
            }
            if (took) {
                std::swap(kin, kout);
                std::swap(vin, vout);
                cur = 1;
                S = S2;
                done = pb;
                planned = std::max(planned_run, pb);
                npass_eff = (planned - pb + 7) / 8;
                h->last_partition = 1;
                h->runs_penalty = 0;
                maxlen = ml;
                window_checked = true;
            } else {
                h->runs_penalty = std::min(16, 2 * h->runs_penalty + 1);
                h->runs_skip = h->runs_penalty;
            }
        }
    }
    for (;;) {
        int bits;
        if (pass_idx < npass_eff) {
            // spread the planned bits evenly over the planned passes
            bits = (planned - done + (npass_eff - pass_idx) - 1) / (npass_eff - pass_idx);
        } else {
            if (maxlen <= seg_cap(h)) break;
            if (done >= K || done >= 24) {
                ok = false;
                break;
            }
            // just enough further bits to bring the longest segment under the capacity
            bits = 1;
            while (bits < 8 && (double)maxlen / (double)(1 << bits) > 0.8 * (double)seg_cap(h)) bits++;
            bits = std::min(bits, K - done);
        }
        if (!tiles_ready) {
            Span sp(h, ESP_ST_SCAN);
            if (S == 1) {  // the whole buffer as one segment
                hipLaunchKernelGGL(set_i64_k, dim3(1), dim3(1), 0, h->stream, (i64 *)h->seg[0].p, (i64)0, E, (i64)0, (i64)0);
                hipLaunchKernelGGL(set_i64_k, dim3(1), dim3(1), 0, h->stream, (i64 *)h->tilef[0].p, (i64)0, T, (i64)0, (i64)0);
                sp.add(2);
            } else {  // (segments of the run-based pass, whose ranked flavour leaves the tile table to its user)
                u64 *tf = (u64 *)h->tilef[cur].p;
                HIPCK(h, hipMemsetAsync(d_maxlen, 0, 8, h->stream));
                hipLaunchKernelGGL(espradix::seg_tiles_k, dim3(grid_for((i64)S + 1, 256)), dim3(256), 0, h->stream, (const i64 *)h->seg[cur].p,
                                   (i64)S, (i64)espradix::TILE, tf, d_maxlen);
                sp.add(1 + espscan::exclusive<u64, false>(h->stream, tf, tf, (i64)S + 1, tf + S + 1));
            }
            tiles_ready = true;
        }
        espradix::Pass p;
        p.keys_in = kin;
        p.vals_in = vin;
        p.keys_out = kout;
        p.vals_out = vout;
        p.seg_start = (const i64 *)h->seg[cur].p;
        p.tile_first = (const i64 *)h->tilef[cur].p;
        p.S = S;
        p.owner_P = 0;
        p.owner_n = 1;
        p.colshift = 0;
        p.base = h->win_base;
        p.span = h->win_span;
        p.err = d_werr;
        p.bits = bits;
        p.shift = K - done - bits;
        const i64 max_tiles = S == 1 ? T : T + S;
        CK(partition_pass(h, p, max_tiles));
        const int S2 = S << bits;
        CK(ensure(h, h->seg[1 - cur], sizeof(i64) * (size_t)(S2 + 1)));
        CK(ensure(h, h->tilef[1 - cur], sizeof(u64) * (size_t)(S2 + 1 + espscan::workspace_elems(S2 + 1))));
        {
            Span sp(h, ESP_ST_SCAN);
            hipLaunchKernelGGL(espradix::new_segments_k, dim3(grid_for(S2 + 1, 256)), dim3(256), 0, h->stream, (const u64 *)h->hist.p,
                               (const i64 *)h->seg[cur].p, (const i64 *)h->tilef[cur].p, S, bits, (i64 *)h->seg[1 - cur].p, E);
            HIPCK(h, hipMemsetAsync(d_maxlen, 0, 8, h->stream));
            u64 *tf = (u64 *)h->tilef[1 - cur].p;
            hipLaunchKernelGGL(espradix::seg_tiles_k, dim3(grid_for(S2 + 1, 256)), dim3(256), 0, h->stream, (const i64 *)h->seg[1 - cur].p,
                               (i64)S2, (i64)espradix::TILE, tf, d_maxlen);
            sp.add(2 + espscan::exclusive<u64, false>(h->stream, tf, tf, S2 + 1, tf + S2 + 1));
        }
        std::swap(kin, kout);
        std::swap(vin, vout);
        cur = 1 - cur;
        S = S2;
        done += bits;
        pass_idx++;
        if (pass_idx >= npass_eff) {
            HIPCK(h, hipMemcpyAsync(h->pin_scalar, d_maxlen, 8, hipMemcpyDeviceToHost, h->stream));
            HIPCK(h, hipStreamSynchronize(h->stream));
            maxlen = (i64)h->pin_scalar[0];
        }
    }
    if (S == 1 && !tiles_ready)  // no pass at all: the buffer is the one segment
        hipLaunchKernelGGL(set_i64_k, dim3(1), dim3(1), 0, h->stream, (i64 *)h->seg[0].p, (i64)0, E, (i64)0, (i64)0);
    HIPCK(h, hipGetLastError());
    if (pass_idx > 0 && !(window_checked && pass_idx == 0)) {  // the partition passes clamp and report keys outside the window
        HIPCK(h, hipMemcpyAsync(h->pin_scalar + 2, d_werr, 4, hipMemcpyDeviceToHost, h->stream));
        HIPCK(h, hipStreamSynchronize(h->stream));
        if ((u32)h->pin_scalar[2]) FAIL(h, ESP_ERR_STATE, "esp_flush: a pending entry lies outside the declared column window (partition)");
    }
    if (out->key_bytes == 4 && (pass_idx > 0 || cur != 1))
        FAIL(h, ESP_ERR_STATE, "esp_flush: internal error (4-byte keys met a further partition pass)");
    out->sk = kin;
    out->sv = vin;
    out->in_primary = (kin == (u64 *)h->keys.p);
    out->S = S;
    out->total = E;
    out->seg_start = (const i64 *)h->seg[cur].p;
    out->rem_bits = K - done;
    out->local_ok = ok && (K - done) <= esplocal::MAX_REM_BITS && maxlen <= seg_cap(h);
    out->maxlen = maxlen;
    h->seen_spread = (done > 0 && Ee > 0.0) ? (double)maxlen * std::ldexp(1.0, done) / Ee : 0.0;
    return ESP_OK;
}

// Shuffled FEM stream on an empty buffer (femitems.hpp): item records -> the flush's own partition passes over them ->
// every update stored once at its bucket position; the handle is left as after a producer-side partition (h->pre).
// *took = false: not applicable, the caller appends in stream order.
static int32_t item_produce_fem(esp_handle *h, const espgen::FemArgs &fa, i64 E, bool *took) {
    *took = false;
    if (h->count != 0 || E <= esplocal::CAP || windowed(h) || h->shard_user) return ESP_OK;
    // (test hooks that pin another path: 2 general, 5 / 12 / 16 partition flavours, 19 plain pending buffer, 25 this one off)
    if (h->force_path == 2 || h->force_path == 5 || h->force_path == 12 || h->force_path == 16 || h->force_path == 19 || h->force_path == 25)
        return ESP_OK;
    const int W = fa.dim + 2, ni = fa.dim + 1;
    const i64 NI = fa.ncells * ni;
    if (NI >= 0xFFFFFFF0ll || (fa.ncells >> 40) != 0 || 2 * NI > E) return ESP_OK;
    // two ping-pong pairs of item records inside the flush's scratch pair (sized for E updates: E / W items each)
    CK(ensure(h, h->keys2, sizeof(u64) * (size_t)E));
    CK(ensure(h, h->vals2, sizeof(double) * (size_t)E));
    espitem::Args a;
    a.fem = fa;
    a.nitems = NI;
    a.ikeys = (u64 *)h->keys2.p;
    a.ivals = (double *)h->vals2.p;
    {
        Span sp(h, ESP_ST_APPEND);
        hipLaunchKernelGGL(espitem::fem_items_k, dim3(grid_for(fa.ncells, espitem::THREADS)), dim3(espitem::THREADS), 0, h->stream, a);
        sp.add(1);
    }
    // the flush's partition over the items: a temporary view of the handle (sort_msd reads count, keys/vals, keys2/vals2)
    const DevBuf k0 = h->keys, v0 = h->vals, k2 = h->keys2, v2 = h->vals2;
    const i64 count0 = h->count;
    const double spread0 = h->seen_spread;
    h->keys.p = a.ikeys, h->keys.bytes = sizeof(u64) * (size_t)NI;
    h->vals.p = a.ivals, h->vals.bytes = sizeof(double) * (size_t)NI;
    h->keys2.p = a.ikeys + NI, h->keys2.bytes = sizeof(u64) * (size_t)NI;
    h->vals2.p = a.ivals + NI, h->vals2.bytes = sizeof(double) * (size_t)NI;
    h->count = NI;
    h->plan_cap = (i64)esplocal::CAP / W;
    h->item_mode = true;
    Sorted st;
    const int32_t rc = sort_msd(h, &st);
    h->keys = k0, h->vals = v0, h->keys2 = k2, h->vals2 = v2;
    h->count = count0;
    h->plan_cap = 0;
    h->item_mode = false;
    if (rc != ESP_OK) return rc;
    const int K = window_bits(h);
    if (!st.local_ok || st.S < 2 || st.rem_bits < h->L.rb || st.maxlen * W > (i64)esplocal::CAP) {
        h->seen_spread = spread0;
        return ESP_OK;  // (no segment table the bucket kernel takes: the plain producer and the flush's own passes)
    }
    const bool k32 = st.rem_bits <= 32 && h->force_path != 14;
    a.sorted = st.sv;
    a.rem_bits = st.rem_bits;
    a.base = h->win_base;
    a.keys_out = (u64 *)h->keys.p;
    a.vals_out = (double *)h->vals.p;
    CK(ensure(h, h->seg[1], sizeof(i64) * (size_t)(st.S + 1)));
    {
        Span sp(h, ESP_ST_APPEND);
        const dim3 grid(grid_for(NI, espitem::THREADS)), block(espitem::THREADS);
        if (k32)
            hipLaunchKernelGGL(espitem::fem_expand_k<true>, grid, block, 0, h->stream, a);
        else
            hipLaunchKernelGGL(espitem::fem_expand_k<false>, grid, block, 0, h->stream, a);
        hipLaunchKernelGGL(espitem::scale_segments_k, dim3(grid_for((i64)st.S + 1, 256)), dim3(256), 0, h->stream, st.seg_start, (i64)st.S + 1,
                           (i64)W, (i64 *)h->seg[1].p);
        sp.add(2);
    }
    HIPCK(h, hipGetLastError());
    esp_handle::PrePart &pp = h->pre;
    pp.K = K;
    pp.pb = K - st.rem_bits;
    pp.maxlen = st.maxlen * W;
    pp.key_bytes = k32 ? 4 : 8;
    pp.kind = ESP_RAWUPDATE;
    pp.E = E;
    pp.tail = 0;
    pp.base = h->win_base;
    pp.span = h->win_span;
    pp.Ee = plan_entries(E, K, h->win_span);
    pp.mw_P = 0, pp.mw_me = 0, pp.mw_shift = 0, pp.mw_nb = 0, pp.mw_eps = 0;
    pp.own32 = false;
    *took = true;  // (the caller sets pre.valid once the entries are counted in)
    return ESP_OK;
}

// colend (u64, n+1, zero-initialised, filled with column ends) -> colptr; merges with the old
// CSC when there is one.  New entries are in h->newkey/h->newval (Z0>0) or already in
// h->rowval/h->nzval (Z0==0).
static int32_t finish_csc(esp_handle *h, i64 Z0, i64 Zn, const u64 *new_key, const double *new_val) {
    const i64 N1 = h->n + 1;
    u64 *colend = (u64 *)h->colend.p;
    i64 c0, ccnt;  // the columns this flush can have touched (a shard's window, else all)
    col_range(h, &c0, &ccnt);
    if (windowed(h)) h->tail_stale = h->wc1 < h->n;
    if (Z0 == 0) {
        // colptr = 1 + exclusive max-scan of the column ends, written by the scan's last pass
        Span sp(h, ESP_ST_COLPTR);
        sp.add(espscan::exclusive<u64, true>(h->stream, colend + c0, (u64 *)h->colptr.p + c0, ccnt, colend + N1, (u64)1));
        if (!windowed(h)) h->ones_pending = false;  // (every entry of colptr was written)
        h->nnz = Zn;
        h->pattern_version++, h->values_version++;
        return ESP_OK;
    }
    const i64 Zt = Z0 + Zn;
    {
        Span sp(h, ESP_ST_COLPTR);
        sp.add(espscan::exclusive<u64, true>(h->stream, colend + c0, colend + c0, ccnt, colend + N1));
    }
    CK(ensure(h, h->rowval2, sizeof(i64) * (size_t)Zt));
    CK(ensure(h, h->nzval2, sizeof(double) * (size_t)Zt));
    if (h->force_path == 17) {
        // (test hook: the merge-path join over a per-entry column array, kept as a second implementation of the same join)
        if (Z0 >= 0xFFFFFFF0ll) FAIL(h, ESP_ERR_UNSUPPORTED, "esp_flush: CSC too large for the 32-bit column index");
        {
            Span sp(h, ESP_ST_COLPTR);
            const i64 hn = Z0 + 1;  // column index of every stored entry
            CK(ensure(h, h->heads, sizeof(u32) * (size_t)(hn + espscan::workspace_elems(hn))));
            u32 *heads = (u32 *)h->heads.p;
            HIPCK(h, hipMemsetAsync(heads, 0, sizeof(u32) * (size_t)hn, h->stream));
            hipLaunchKernelGGL(espfold::col_heads_k, dim3(grid_for(ccnt - 1, 256)), dim3(256), 0, h->stream, (const i64 *)h->colptr.p, c0, ccnt - 1,
                               heads);
            sp.add(1 + espscan::exclusive<u32, true>(h->stream, heads, heads, hn, heads + hn));
        }
        Span sp(h, ESP_ST_MERGE);
        espmerge::Args a;
        a.old_col = (const u32 *)h->heads.p + 1;
        a.old_row = (const i64 *)h->rowval.p;
        a.old_val = (const double *)h->nzval.p;
        a.Z0 = Z0;
        a.new_key = new_key;
        a.new_val = new_val;
        a.Zn = Zn;
        a.rb = h->L.rb;
        a.out_row = (i64 *)h->rowval2.p;
        a.out_val = (double *)h->nzval2.p;
        hipLaunchKernelGGL(espmerge::merge_k, dim3(grid_for(Zt, espmerge::TILE)), dim3(espmerge::THREADS), 0, h->stream, a);
        sp.add(1);
    } else {
        // column-tiled join: every stored and every new entry finds its own place in its merged column.  The stored
        // entries outside the flush's column range (a shard's window) keep their order: in front of the range they
        // stay where they are, behind it they move up by Zn.
        Span sp(h, ESP_ST_MERGE);
        const i64 ncols = ccnt - 1;
        if (windowed(h)) {
            // (win_excl: nothing is stored outside the window)
        } else if (c0 != 0 || ncols != h->n) {
            FAIL(h, ESP_ERR_STATE, "esp_flush: internal error (column range of the join)");
        }
        espmerge::ColArgs a;
        a.old_colptr = (const i64 *)h->colptr.p;
        a.old_row = (const i64 *)h->rowval.p;
        a.old_val = (const double *)h->nzval.p;
        a.newstart = (const u64 *)colend;
        a.new_key = new_key;
        a.new_val = new_val;
        a.rb = h->L.rb;
        a.c_begin = c0;
        a.ncols = ncols;
        a.out_row = (i64 *)h->rowval2.p;
        a.out_val = (double *)h->nzval2.p;
        hipLaunchKernelGGL(espmerge::colmerge_k, dim3(grid_for(ncols, espmerge::CT)), dim3(espmerge::THREADS), 0, h->stream, a);
        sp.add(1);
    }
    {
        Span sp(h, ESP_ST_COLPTR);
        hipLaunchKernelGGL(espfold::colptr_finish_k, dim3(grid_for(ccnt, 256)), dim3(256), 0, h->stream, (const u64 *)colend + c0,
                           (const i64 *)h->colptr.p + c0, ccnt, (i64 *)h->colptr.p + c0);
        sp.add(1);
    }
    std::swap(h->rowval, h->rowval2);
    std::swap(h->nzval, h->nzval2);
    h->nnz = Zt;
    h->pattern_version++, h->values_version++;
    return ESP_OK;
}

static int32_t prepare_outputs(esp_handle *h, i64 Z0, i64 Zn) {
    const i64 N1 = h->n + 1;
    CK(ensure(h, h->colend, sizeof(u64) * (size_t)(N1 + espscan::workspace_elems(N1))));
    HIPCK(h, hipMemsetAsync(h->colend.p, 0, sizeof(u64) * (size_t)N1, h->stream));
    if (Z0 == 0) {
        CK(ensure(h, h->rowval, sizeof(i64) * (size_t)Zn));
        CK(ensure(h, h->nzval, sizeof(double) * (size_t)Zn));
    } else {
        CK(ensure(h, h->newkey, sizeof(u64) * (size_t)Zn));
        CK(ensure(h, h->newval, sizeof(double) * (size_t)Zn));
    }
    return ESP_OK;
}

// fast path: LDS bucket kernel over the MSD segments; writes the final arrays itself
static int32_t flush_local(esp_handle *h, const Sorted &st, int mode, i64 *Zn_out) {
    const i64 Z0 = h->nnz;
    const i64 N1 = h->n + 1;
    // Segments behind the last column hold nothing (a matrix of 10^7 columns fills 60 % of the 2^24 its column bits span:
    // 40 % of the segments, each of which would still draw a ticket, resolve its offset and leave): the launch ends at
    // the segment of the last column; that segment checks that every entry lies in front of its end (Args::total).
    int S = st.S;
    i64 total_check = -1;
    if (st.npieces == 0 && st.total >= 0 && st.seg_start && st.rem_bits >= h->L.rb && st.rem_bits - h->L.rb < 40 && S > 1) {
        const int clb0 = st.rem_bits - h->L.rb;
        // (the segments cut the key window: its first column, and the column behind its last)
        const i64 c_first = (i64)(h->win_base >> h->L.rb), c_last = (i64)((h->win_base + h->win_span) >> h->L.rb);
        const i64 need = ceil_div<i64>(c_last - c_first, (i64)1 << clb0);
        if (need >= 1 && need < (i64)S) {
            S = (int)need;
            total_check = st.total;
        }
    }
    // esp_flush normalised the buffers: data in keys/vals, scratch pair = keys2/vals2
    u64 *tk = (u64 *)h->keys2.p;
    double *tv = (double *)h->vals2.p;
    // look-back granules: one per segment | ticket, error flag | longest run | one per group of 64 segments
    // (cleared as a multiple of 256 bytes: the runtime splits an odd-sized memset into two launches)
    const i64 G = ((((i64)S + 63) / 64 + 1 + S + 2 + 31) & ~(i64)31) - (S + 2);
    CK(ensure(h, h->segout, sizeof(u64) * (size_t)(S + 4 + G)));
    u64 *status = (u64 *)h->segout.p;
    HIPCK(h, hipMemsetAsync(status, 0, sizeof(u64) * (size_t)(S + 2 + G), h->stream));
    CK(ensure(h, h->colend, sizeof(u64) * (size_t)(N1 + espscan::workspace_elems(N1))));
    esplocal::Args a;
    const char *stop_env = getenv("ESP_LOCAL_STOP");
    // A fresh matrix whose segments are whole blocks of <= CL_MAX columns that start at the first column of the
    // range this flush can touch (all columns, or the column window of a shard) and cover it: every segment writes
    // the colptr of its own columns (no column-end marks, no memset and no scan over the columns).
    // force_path 13: marks + scan.
    bool direct = false;
    const i64 col_begin = windowed(h) ? h->wc0 : 0, col_end = windowed(h) ? h->wc1 : h->n;
    {
        const int clb = st.rem_bits - h->L.rb;
        const u64 seg_base = st.has_base ? st.base : st.npieces > 0 ? h->part_base : h->win_base;
        direct = Z0 == 0 && clb >= 0 && clb <= esplocal::CL_MAX_BITS && h->force_path != 3 && h->force_path != 13 && !stop_env &&
                 seg_base == ((u64)col_begin << h->L.rb) && col_begin + ((i64)S << clb) >= col_end;
    }
    h->last_colptr_direct = direct ? 1 : 0;
    if (!direct) {
        i64 c0, cnt;
        col_range(h, &c0, &cnt);
        HIPCK(h, hipMemsetAsync((u64 *)h->colend.p + c0, 0, sizeof(u64) * (size_t)cnt, h->stream));
    }
    // (a failed flush must not leave a half-written colptr behind)
    auto restore_colptr = [&]() {
        if (direct) {
            hipLaunchKernelGGL(fill_i64_k, dim3(grid_for(N1, 256)), dim3(256), 0, h->stream, (i64 *)h->colptr.p, N1, (i64)1);
            h->tail_stale = false;
            h->ones_pending = false;
        }
    };
    // The small variant of the bucket kernel (3 workgroups per CU) serves segments of at most 3072 entries over at most
    // 256 columns whose column runs the register tiers take; a segment with longer runs (it cannot know before it counts)
    // goes through the variant's slow tier, and what it reports sends the next flushes to the regular kernel.
    // force_path 18: never.
    bool small_variant = false;
    {
        const int clb = st.rem_bits - h->L.rb;
        // (what the handle's last flush saw decides; a handle without history tries it when the columns hold few entries
        // on average -- a stencil's 12, not a 3-D FEM mesh's 120)
        const double per_col = (double)h->count / (double)std::max<i64>(col_end - col_begin, 1);
        // (runs of 17..24 want the 24-input network, which does not fit the variant's 80 registers: measured 4.7 against
        // 4.0 ms on 2-D FEM)
        const bool runs_fit = h->seen_maxrun > 0 ? h->seen_maxrun <= 16 : per_col <= 16.0;
        small_variant = st.maxlen <= 6 * esplocal::THREADS && clb >= 0 && clb <= 8 &&
                        st.rem_bits <= esplocal::REG_MAX_REM && runs_fit && h->force_path != 3 &&
                        h->force_path != 18 && !stop_env;
    }
    h->last_local_small = small_variant ? 1 : 0;
    {
        Span sp(h, ESP_ST_LOCAL);
        a.kind32 = (u32)((st.key_bytes == 4 || st.p32_piece >= 0 || st.all32) ? st.kind : 0);
        a.k32_piece = st.p32_piece;
        a.k32_lo = st.p32_lo;
        h->last_key_bytes = (st.p32_piece >= 0 || st.all32) ? 4 : st.key_bytes;
        a.colptr_out = direct ? (i64 *)h->colptr.p : nullptr;
        a.no_group = h->force_path == 24 ? 1 : 0;  // 24: test hook, long column runs through the radix tier
        // (every pending entry was noted with one kind; pieces of other ranks carry kinds this handle has not seen)
        a.kind_all = (st.npieces == 0 && h->kind_uniform >= 0 && h->kind_noted == h->count && h->force_path != 15) ? h->kind_uniform : -1;
        a.col_end = col_end;
        a.n_cols = h->n;
        a.keys_in = st.sk;
        a.vals_in = st.sv;
        a.seg_start = st.seg_start;
        a.S = S;
        a.rem_bits = st.rem_bits;
        a.base = st.has_base ? st.base : st.npieces > 0 ? h->part_base : h->win_base;
        a.rb = h->L.rb;
        {
            const int clb = st.rem_bits - h->L.rb;
            a.cl_bits = (clb >= 0 && clb <= esplocal::CL_MAX_BITS && h->force_path != 3) ? clb : -1;
        }
        // a segment is a whole number of columns when the prefix does not reach into the row bits
        a.col_aligned = st.rem_bits >= h->L.rb ? 1 : 0;
        a.csc = espfold::Csc{(const i64 *)h->colptr.p, (const i64 *)h->rowval.p, (double *)h->nzval.p, Z0};
        a.mode = mode;
        a.out_row = (i64 *)tk;
        a.out_key = tk;
        a.out_val = tv;
        a.colend = (u64 *)h->colend.p;
        a.status = status;
        a.gstatus = status + S + 2;
        a.npieces = st.npieces;
        a.total = total_check;
        a.pstart = st.pstart;
        a.ptab = st.ptab;
        a.ticket = (u32 *)(status + S);
        a.err = (u32 *)(status + S) + 1;
        a.maxrun_seen = (u32 *)(status + S) + 2;  // (zeroed with the granules)
        {
            a.stop_after = stop_env ? atoi(stop_env) : 0;
            a.stamps = nullptr;
            if (getenv("ESP_LOCAL_STAMPS")) {  // diagnostics: per-segment phase stamps, dumped to a file
                CK(ensure(h, h->heads, sizeof(u64) * (size_t)S * 16));
                HIPCK(h, hipMemsetAsync(h->heads.p, 0, sizeof(u64) * (size_t)S * 16, h->stream));
                a.stamps = (unsigned long long *)h->heads.p;
            }
        }
        const i64 max_grid = h->force_path == 4 ? 64 : esplocal::MAX_GRID;  // 4: test hook, many launches
        for (i64 first = 0; first < S; first += max_grid) {
            const unsigned grid = (unsigned)std::min<i64>(max_grid, S - first);
            a.first = first;
            // (the variant with the 24-input register tier for a matrix whose last flush met such runs)
            const bool big = h->seen_maxrun > 16 && h->seen_maxrun <= esplocal::REG_RUN && h->force_path != 26;  // (26: test hook, runs of 17..24 through the group tier)
            // (key format: 0 packed, 1 four-byte keys of one kind, 2 four-byte keys that are all UPDATEs)
            // (3: packed keys whose kinds are all UPDATE -- the pieces of a shard)
            // (4 / 5: pieces of which one -- a shard's own range -- holds 4-byte keys; 5: everything is an UPDATE)
            // (6 / 7: pieces that all hold 4-byte keys of one kind -- a producer's batch and its tail; 7: UPDATE)
            const int keys = st.npieces > 0 ? (st.all32 ? (st.kind == ESP_UPDATE && h->force_path != 15 ? 7 : 6)
                                               : st.p32_piece >= 0 ? (st.all_update && h->force_path != 15 ? 5 : 4)
                                                                   : (st.all_update && h->force_path != 15 ? 3 : 0))
                                            : st.key_bytes != 4 ? 0 : (st.kind == ESP_UPDATE && h->force_path != 15 ? 2 : 1);
            h->last_fold_update = keys >= 2 ? 1 : 0;
            // (the instantiations live in local_*.hip)
            const esplocal::Variant var{Z0 == 0, st.npieces > 0, big && !small_variant, small_variant, keys};
            if (!esplocal::launch(var, grid, h->stream, a)) FAIL(h, ESP_ERR_STATE, "esp_flush: internal error (no bucket kernel for this flush)");
        }
        sp.add(1);
    }
    HIPCK(h, hipMemcpyAsync(h->pin_scalar, status + (S - 1), 24, hipMemcpyDeviceToHost, h->stream));  // last granule | ticket, err | maxrun
    HIPCK(h, hipMemcpyAsync(h->pin_scalar + 3, (u32 *)h->misc.p + 60, 4, hipMemcpyDeviceToHost, h->stream));
    HIPCK(h, hipStreamSynchronize(h->stream));

    if ((u32)h->pin_scalar[3]) {
        restore_colptr();
        FAIL(h, ESP_ERR_STATE, "esp_flush: a pending entry lies outside the declared column window");
    }
    if (a.stamps) {
        std::vector<u64> st((size_t)S * 16);
        HIPCK(h, hipMemcpy(st.data(), a.stamps, sizeof(u64) * st.size(), hipMemcpyDeviceToHost));
        if (FILE *f = fopen(getenv("ESP_LOCAL_STAMPS"), "wb")) {
            fwrite(st.data(), sizeof(u64), st.size(), f);
            fclose(f);
        }
    }
    if (direct && !windowed(h)) h->ones_pending = false;  // (the bucket kernel wrote every entry of colptr)
    const u32 lookback_err = (u32)(h->pin_scalar[1] >> 32);
    h->seen_maxrun = (int)(u32)(h->pin_scalar[2] >> 0 & 0xFFFFFFFFull);
    if (lookback_err & 7u) restore_colptr();
    if (lookback_err & 2u) FAIL(h, ESP_ERR_STATE, "esp_flush: a pending entry lies outside the declared column window (bucket)");
    if (lookback_err & 1u) FAIL(h, ESP_ERR_HIP, "esp_flush: look-back chain timed out inside the bucket kernel");
    if (lookback_err & 4u) FAIL(h, ESP_ERR_HIP, "esp_flush: internal error (early segment total differs from the folded total)");
    const i64 Zn = (i64)(h->pin_scalar[0] & esplocal::ST_VAL);
    *Zn_out = Zn;
    if (a.stop_after || Zn == 0) return ESP_OK;
    if (Z0 == 0) {
        // the scratch pair now holds rowval/nzval: rotate the buffers instead of copying.  The old
        // (empty) CSC arrays become the next flush's scratch pair: bring them to the same capacity
        // once, so that the rotation never shrinks the scratch pair (a 10 GB hipMalloc per flush
        // costs more than the flush itself)
        CK(ensure(h, h->rowval, h->keys2.bytes));
        CK(ensure(h, h->nzval, h->vals2.bytes));
        std::swap(h->rowval, h->keys2);
        std::swap(h->nzval, h->vals2);
        if (direct) {  // colptr is complete (behind a column window it is refreshed lazily, as after the scan)
            if (windowed(h)) h->tail_stale = h->wc1 < h->n;
            h->nnz = Zn;
            h->pattern_version++, h->values_version++;
            return ESP_OK;
        }
        return finish_csc(h, 0, Zn, nullptr, nullptr);
    }
    return finish_csc(h, Z0, Zn, (const u64 *)tk, (const double *)tv);
}

// general path: finish with a full stable LSD sort and the global fold (any run length)
static int32_t flush_global(esp_handle *h, int mode, i64 *Zn_out) {
    const i64 E = h->count;
    const i64 Z0 = h->nnz;
    const u64 *sk;
    const double *sv;
    CK(sort_pending_lsd(h, &sk, &sv));  // sorted data now in h->keys/h->vals; keys2/vals2 are scratch
    u32 *flag = (u32 *)h->vals2.p;        // E+1 u32 fits in E doubles (E>=1)
    double *fval = (double *)h->keys2.p;  // E doubles
    espfold::Csc csc{(const i64 *)h->colptr.p, (const i64 *)h->rowval.p, (double *)h->nzval.p, Z0};
    {
        Span sp(h, ESP_ST_FOLD);
        hipLaunchKernelGGL(espfold::fold_k, dim3(grid_for(E + 1, espfold::THREADS)), dim3(espfold::THREADS), 0, h->stream, sk, sv, E,
                           csc, h->L.rb, mode, flag, fval);
        sp.add(1);
    }
    {
        Span sp(h, ESP_ST_SCAN);
        int l = 0;
        CK(scan_inplace<u32, false>(h, flag, E + 1, h->hist, &l));
        sp.add(l);
    }
    HIPCK(h, hipMemcpyAsync(h->pin_scalar, flag + E, 4, hipMemcpyDeviceToHost, h->stream));
    HIPCK(h, hipStreamSynchronize(h->stream));
    const i64 Zn = (i64) * (u32 *)h->pin_scalar;
    *Zn_out = Zn;
    if (Zn == 0) return ESP_OK;
    CK(prepare_outputs(h, Z0, Zn));
    {
        Span sp(h, ESP_ST_FOLD);
        if (Z0 == 0)
            hipLaunchKernelGGL((espfold::compact_k<true>), dim3(grid_for(E, espfold::THREADS)), dim3(espfold::THREADS), 0, h->stream, sk,
                               (const double *)fval, E, (const u32 *)flag, h->L.rb, (i64 *)h->rowval.p, (u64 *)nullptr,
                               (double *)h->nzval.p, (u64 *)h->colend.p);
        else
            hipLaunchKernelGGL((espfold::compact_k<false>), dim3(grid_for(E, espfold::THREADS)), dim3(espfold::THREADS), 0, h->stream, sk,
                               (const double *)fval, E, (const u32 *)flag, h->L.rb, (i64 *)nullptr, (u64 *)h->newkey.p,
                               (double *)h->newval.p, (u64 *)h->colend.p);
        sp.add(1);
    }
    return finish_csc(h, Z0, Zn, (const u64 *)h->newkey.p, (const double *)h->newval.p);
}

__global__ void piece_totals_k(const i64 *__restrict__ pstart, int P, i64 nb, unsigned long long *__restrict__ maxlen,
                               unsigned long long *__restrict__ negative);

// bucket starts of a tail sorted by its prefix digit: first position whose digit is >= d, for d = 0 .. NB
__global__ void tail_bucket_starts_k(const u64 *__restrict__ keys, i64 T, u64 base, u64 span, int shift, i64 NB, i64 *__restrict__ out) {
    const i64 d = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (d > NB) return;
    i64 lo = 0, hi = T;
    while (lo < hi) {
        const i64 mid = (lo + hi) >> 1;
        u64 kn = (keys[mid] >> ESP_TAG_BITS) - base;
        kn = kn < span ? kn : span - 1;
        if ((i64)(kn >> shift) < d)
            lo = mid + 1;
        else
            hi = mid;
    }
    out[d] = lo;
}

// A producer's bucket-ordered batch with packed entries appended behind it (a re-assembly whose mesh gained a few
// couplings: the generator's batch, then the new positions): the tail alone goes through the run-based partition with
// the batch's prefix bits, and the bucket kernel reads every segment as two pieces -- the batch's bucket (4-byte keys
// when the producer wrote them), then the tail's.  Stream order is kept: the tail's entries come after the batch's.
static int32_t flush_pre_tail(esp_handle *h, int mode, i64 *Zn, bool *served) {
    *served = false;
    const esp_handle::PrePart pp = h->pre;
    const i64 E0 = pp.E, T = pp.tail, NB = (i64)1 << pp.pb;
    CK(ensure(h, h->newkey, sizeof(u64) * (size_t)T));
    CK(ensure(h, h->newval, sizeof(double) * (size_t)T));
    CK(ensure(h, h->seg[0], sizeof(i64) * (size_t)(NB + 1)));
    CK(ensure(h, h->tilef[0], sizeof(u64) * (size_t)(NB + 1 + espscan::workspace_elems(NB + 1))));
    CK(ensure(h, h->misc, 256));
    unsigned long long *d_maxlen = (unsigned long long *)h->misc.p + 24;
    HIPCK(h, hipMemsetAsync(d_maxlen, 0, 64, h->stream));  // maxlen .. the four flag words (run_partition)
    bool took = false, tiles = false;
    i64 ml = 0;
    // (a tail of the batch's kind goes out as 4-byte keys too: the bucket kernel then reads nothing but such keys)
    const bool tail32 = pp.key_bytes == 4 && h->kind_uniform == pp.kind && h->kind_noted == h->count;
    int tail_bytes = 8;
    CK(run_partition(h, (const u64 *)h->keys.p + E0, (const double *)h->vals.p + E0, (u64 *)h->newkey.p, (double *)h->newval.p, pp.K,
                     pp.pb, (i64 *)h->seg[0].p, (u64 *)h->tilef[0].p, &tiles, &took, &ml, nullptr, 0, tail32, &tail_bytes, T));
    if (!took) {
        // no pre-sorted stream (a few entries spread over many buckets, say): stable 8-bit passes over the prefix bits of the
        // tail alone, least significant first; the last one lands in newkey/newval (keys2/vals2, the bucket kernel's
        // output, serve as the other half of the ping-pong until then)
        CK(ensure(h, h->keys2, sizeof(u64) * (size_t)(E0 + T)));
        CK(ensure(h, h->vals2, sizeof(double) * (size_t)(E0 + T)));
        CK(ensure(h, h->segs, sizeof(i64) * 8));
        i64 *segs = (i64 *)h->segs.p;
        const i64 TR = ceil_div<i64>(T, espradix::TILE);
        hipLaunchKernelGGL(set_i64_k, dim3(1), dim3(1), 0, h->stream, segs, (i64)0, T, (i64)0, TR);
        HIPCK(h, hipMemsetAsync(d_maxlen, 0, 64, h->stream));
        const int npass = (pp.pb + 7) / 8;
        const u64 *ki = (const u64 *)h->keys.p + E0;
        const double *vi = (const double *)h->vals.p + E0;
        bool to_new = (npass & 1) != 0;
        for (int done = 0; done < pp.pb; done += 8) {
            espradix::Pass p;
            p.keys_in = ki;
            p.vals_in = vi;
            p.keys_out = to_new ? (u64 *)h->newkey.p : (u64 *)h->keys2.p;
            p.vals_out = to_new ? (double *)h->newval.p : (double *)h->vals2.p;
            p.seg_start = segs;
            p.tile_first = segs + 2;
            p.S = 1;
            p.owner_P = 0;
            p.owner_n = 1;
            p.colshift = 0;
            p.base = h->win_base;
            p.span = h->win_span;
            p.err = (u32 *)h->misc.p + 60;
            p.shift = pp.K - pp.pb + done;
            p.bits = std::min(8, pp.pb - done);
            CK(partition_pass(h, p, TR));
            ki = p.keys_out;
            vi = p.vals_out;
            to_new = !to_new;
        }
        {
            Span sp(h, ESP_ST_SCAN);
            hipLaunchKernelGGL(tail_bucket_starts_k, dim3(grid_for(NB + 1, 256)), dim3(256), 0, h->stream, (const u64 *)h->newkey.p, T,
                               h->win_base, h->win_span, pp.K - pp.pb, NB, (i64 *)h->seg[0].p);
            sp.add(1);
        }
        HIPCK(h, hipMemcpyAsync(h->pin_scalar, (u32 *)h->misc.p + 60, 4, hipMemcpyDeviceToHost, h->stream));
        HIPCK(h, hipStreamSynchronize(h->stream));
        if ((u32)h->pin_scalar[0]) FAIL(h, ESP_ERR_STATE, "esp_flush: a pending entry lies outside the declared column window (partition)");
        h->last_run_order = 0;
    }
    // pointer table (keys | values) | piece starts: batch, tail
    const size_t o_ps = 256 * 8;
    CK(ensure(h, h->piecetab, o_ps + sizeof(i64) * 2 * (size_t)(NB + 1)));
    char *TB = (char *)h->piecetab.p;
    const void *tab[4] = {h->keys.p, h->newkey.p, h->vals.p, h->newval.p};
    i64 *pstart = (i64 *)(TB + o_ps);
    HIPCK(h, hipMemcpyAsync(TB, tab, sizeof(tab), hipMemcpyHostToDevice, h->stream));
    HIPCK(h, hipMemcpyAsync(pstart, h->seg[1].p, sizeof(i64) * (size_t)(NB + 1), hipMemcpyDeviceToDevice, h->stream));
    HIPCK(h, hipMemcpyAsync(pstart + (NB + 1), h->seg[0].p, sizeof(i64) * (size_t)(NB + 1), hipMemcpyDeviceToDevice, h->stream));
    HIPCK(h, hipMemsetAsync(d_maxlen, 0, 16, h->stream));
    {
        Span sp(h, ESP_ST_SCAN);
        hipLaunchKernelGGL(piece_totals_k, dim3(grid_for(NB, 256)), dim3(256), 0, h->stream, (const i64 *)pstart, 2, NB, d_maxlen, d_maxlen + 1);
        sp.add(1);
    }
    HIPCK(h, hipMemcpyAsync(h->pin_scalar, d_maxlen, 8, hipMemcpyDeviceToHost, h->stream));
    HIPCK(h, hipStreamSynchronize(h->stream));  // (tab is read by the copy above)
    const i64 merged = (i64)h->pin_scalar[0];
    if (merged > (i64)esplocal::CAP) return ESP_OK;
    HIPCK(h, hipMemsetAsync((u32 *)h->misc.p + 60, 0, 4, h->stream));
    Sorted st;
    st.sk = (const u64 *)h->keys.p;
    st.sv = (const double *)h->vals.p;
    st.in_primary = true;
    st.S = (int)NB;
    st.seg_start = nullptr;
    st.rem_bits = pp.K - pp.pb;
    st.local_ok = true;
    st.npieces = 2;
    st.all_update = h->kind_uniform == ESP_UPDATE && h->kind_noted == h->count;
    st.ptab = (const void *const *)TB;
    st.pstart = pstart;
    st.maxlen = merged;
    st.has_base = true;
    st.base = h->win_base;
    if (pp.key_bytes == 4 && tail_bytes == 4) {
        st.all32 = true;
        st.kind = pp.kind;
    } else if (pp.key_bytes == 4) {
        st.p32_piece = 0;
        st.p32_lo = 0;
        st.kind = pp.kind;
    }
    CK(ensure(h, h->keys2, sizeof(u64) * (size_t)(E0 + T)));
    CK(ensure(h, h->vals2, sizeof(double) * (size_t)(E0 + T)));
    CK(flush_local(h, st, mode, Zn));
    *served = true;
    return ESP_OK;
}

extern "C" int32_t esp_flush(esp_handle *h, int32_t mode, int64_t *new_nnz, int32_t *pattern_changed) {
    if (!h) return ESP_ERR_INVALID;
    if (mode != ESP_FLUSH_ROUTED && mode != ESP_FLUSH_PLUS) FAIL(h, ESP_ERR_INVALID, "esp_flush: mode");
    (void)hipSetDevice(h->device);
    if (pattern_changed) *pattern_changed = 0;
    i64 E = h->count;
    if (E == 0) {
        if (new_nnz) *new_nnz = h->nnz;
        return ESP_OK;
    }
    if (E >= 0xFFFFFFF0ll) FAIL(h, ESP_ERR_UNSUPPORTED, "esp_flush: %lld pending entries exceed the 2^32 limit of one flush", (long long)E);
    hipEvent_t fa = nullptr;
    if (h->timing) {
        fa = ev_get(h);
        (void)hipEventRecord(fa, h->stream);
    }
    i64 Zn = 0;
    bool use_local = h->force_path != 2;
    if (h->ones_pending && windowed(h)) CK(fix_tail(h));  // (cannot happen: a window is declared through fix_tail)
    if (h->pre.valid) {  // the producer's partition serves this flush if nothing changed since (appends BEHIND it may have)
        const esp_handle::PrePart &pp = h->pre;
        const bool usable = use_local && !h->part_assembled && pp.mw_P == 0 && pp.E + pp.tail == E && pp.base == h->win_base &&
                            pp.span == h->win_span && pp.maxlen <= (i64)esplocal::CAP && pp.K - pp.pb <= esplocal::MAX_REM_BITS;
        if (!usable) CK(pending_materialize(h));
    }
    bool served = false, split = false;
    i64 Zsplit = 0;  // new entries of the batch's own flush
    if (h->pre.valid && h->pre.tail > 0 && h->nnz > 0 && mode == ESP_FLUSH_ROUTED && h->force_path != 22) {
        // Batch + tail over a stored pattern (a re-assembly whose mesh gained couplings): the batch by itself -- its buckets
        // fit the small variant of the bucket kernel, and a batch of hits emits nothing and needs no join -- then the tail
        // as a flush of its own.  A ROUTED flush may be cut at any stream position: flush! between two calls of an
        // ExtendableSparseMatrix never changes a result (extendable.jl:159-255: a call either hits the CSC or goes to the
        // buffer, which the flush adds as it is).  Not so csc + buffer (ESP_FLUSH_PLUS): the buffer is folded by itself
        // first.  force_path 22: one flush over two pieces, as on a fresh matrix.
        const esp_handle::PrePart pp = h->pre;
        Sorted st;
        st.sk = (const u64 *)h->keys.p;
        st.sv = (const double *)h->vals.p;
        st.in_primary = true;
        st.S = 1 << pp.pb;
        st.total = pp.E;
        st.seg_start = (const i64 *)h->seg[1].p;
        st.rem_bits = pp.K - pp.pb;
        st.local_ok = true;
        st.key_bytes = pp.key_bytes;
        st.kind = pp.kind;
        st.maxlen = pp.maxlen;
        CK(ensure(h, h->keys2, sizeof(u64) * (size_t)E));
        CK(ensure(h, h->vals2, sizeof(double) * (size_t)E));
        CK(flush_local(h, st, mode, &Zsplit));  // (on failure everything is still pending)
        // the tail is the pending buffer now: packed keys, moved to the front (chunks of at most E0 entries: source and
        // destination of one copy never overlap)
        const i64 E0 = pp.E, T = pp.tail;
        {
            Span sp(h, ESP_ST_COPY);
            for (i64 at = 0; at < T; at += E0) {
                const i64 c = std::min(E0, T - at);
                hipError_t e1 = hipMemcpyAsync((u64 *)h->keys.p + at, (const u64 *)h->keys.p + E0 + at, sizeof(u64) * (size_t)c, hipMemcpyDeviceToDevice, h->stream);
                if (e1 == hipSuccess)
                    e1 = hipMemcpyAsync((double *)h->vals.p + at, (const double *)h->vals.p + E0 + at, sizeof(double) * (size_t)c, hipMemcpyDeviceToDevice, h->stream);
                if (e1 != hipSuccess) {  // (the batch is in the matrix already: it must not stay pending and be applied again)
                    h->count = 0;
                    pending_changed(h);
                    FAIL(h, ESP_ERR_HIP, "esp_flush: %s while moving the entries behind a flushed batch; they were dropped", hipGetErrorString(e1));
                }
                sp.add(2);
            }
        }
        const bool one_kind = h->kind_uniform >= 0 && h->kind_noted == h->count;
        h->count = T;
        h->pre.valid = false;
        h->kind_noted = one_kind ? T : 0;
        if (!one_kind) h->kind_uniform = -2;
        h->shard_valid = h->part_valid = false;
        h->values_version++;
        E = T;
        split = true;
        // (the tail gets a plan of its own: fewer, fuller segments -- with the batch's 2^16 buckets, small variant and 4-byte
        // keys included, its bucket kernel took 1.4 instead of 0.9 ms at config 3: time follows the number of segments)
    }
    if (h->pre.valid && h->pre.tail > 0) {
        // batch + tail: only the tail is partitioned, the bucket kernel reads every segment as two pieces
        CK(flush_pre_tail(h, mode, &Zn, &served));
        if (!served) CK(pending_materialize(h));  // (a merged segment is too long, or the tail is no pre-sorted stream)
    }
    if (served) {
        h->last_partition = 5;
    } else if (h->pre.valid) {
        const esp_handle::PrePart &pp = h->pre;
        Sorted st;
        st.sk = (const u64 *)h->keys.p;
        st.sv = (const double *)h->vals.p;
        st.in_primary = true;
        st.S = 1 << pp.pb;
        st.total = pp.E;
        st.seg_start = (const i64 *)h->seg[1].p;
        st.rem_bits = pp.K - pp.pb;
        st.local_ok = true;
        st.key_bytes = pp.key_bytes;
        st.kind = pp.kind;
        st.maxlen = pp.maxlen;
        CK(ensure(h, h->keys2, sizeof(u64) * (size_t)E));
        CK(ensure(h, h->vals2, sizeof(double) * (size_t)E));
        CK(flush_local(h, st, mode, &Zn));
        h->last_partition = 4;
        h->runs_penalty = 0;
        h->seen_spread = pp.Ee > 0.0 ? (double)pp.maxlen * std::ldexp(1.0, pp.pb) / pp.Ee : 0.0;
    } else if (h->part_assembled) {
        // partitioned shard exchange: the segments are already formed (esp_shard_assemble)
        CK(ensure(h, h->misc, 256));
        HIPCK(h, hipMemsetAsync((u32 *)h->misc.p + 60, 0, 4, h->stream));
        Sorted st;
        char *T = (char *)h->piecetab.p;
        st.sk = (const u64 *)h->keys.p;
        st.sv = (const double *)h->vals.p;
        st.in_primary = true;
        st.S = (int)h->part_nb;
        st.seg_start = nullptr;
        st.rem_bits = h->part_shift;
        st.local_ok = true;
        st.npieces = h->part_P;
        st.all_update = h->part_all_update;
        st.ptab = (const void *const *)T;
        st.pstart = (const i64 *)(T + 256 * 8);
        st.maxlen = h->part_maxlen;
        if (h->part_own32) {
            st.p32_piece = h->part_me;
            st.p32_lo = h->part_own_lo;
            st.kind = h->part_kind32;
        }
        CK(ensure(h, h->keys2, sizeof(u64) * (size_t)std::max<i64>(h->part_total, 1)));
        CK(ensure(h, h->vals2, sizeof(double) * (size_t)std::max<i64>(h->part_total, 1)));
        CK(flush_local(h, st, mode, &Zn));
        h->last_partition = 7;
    } else if (use_local) {
        Sorted st;
        CK(sort_msd(h, &st));
        if (!st.in_primary) {  // keep "pending data lives in keys/vals" true for the general path
            std::swap(h->keys, h->keys2);
            std::swap(h->vals, h->vals2);
            h->cap = (i64)std::min(h->keys.bytes / sizeof(u64), h->vals.bytes / sizeof(double));
            st.in_primary = true;
        }
        if (st.local_ok) {
            const int32_t rc = flush_local(h, st, mode, &Zn);
            if (rc != ESP_OK && st.key_bytes == 4) {
                // the batch stays pending: its packed keys are intact in the scratch pair (the partition wrote the 4-byte
                // keys into the other one)
                std::swap(h->keys, h->keys2);
                std::swap(h->vals, h->vals2);
                h->cap = (i64)std::min(h->keys.bytes / sizeof(u64), h->vals.bytes / sizeof(double));
            }
            CK(rc);
        } else {
            use_local = false;
        }
    }
    if (!use_local && !h->part_assembled) CK(flush_global(h, mode, &Zn));
    if (split) h->last_partition = 6;
    h->last_path = (use_local || h->part_assembled) ? 1 : 2;
    if ((Zn > 0 || Zsplit > 0) && pattern_changed) *pattern_changed = 1;
    h->values_version++;  // (hits were applied in place)
    HIPCK(h, hipGetLastError());
    h->count = 0;
    pending_changed(h);
    if (h->timing && fa) {
        hipEvent_t fb = ev_get(h);
        (void)hipEventRecord(fb, h->stream);
        h->spans.push_back({-1, fa, fb, 0});
    }
    if (new_nnz) *new_nnz = h->nnz;
    return ESP_OK;
}

// test/bench hook: 0 = automatic, 1 = (same as 0), 2 = force the general global path
extern "C" int32_t esp_debug_force_path(esp_handle *h, int32_t path) {
    if (!h) return ESP_ERR_INVALID;
    h->force_path = path;
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

// ------------------------------------------------------------------------ shards
// Column-range shards: owner(col) = floor((col-1)*P/n).  Both calls share one
// histogram + scan of the pending entries by owner; the export is one stable partition pass, so
// every destination receives its entries in this shard's append order.
static int32_t shard_prepare(esp_handle *h, int P, espradix::Pass *out) {
    if (P < 1 || P > 256) FAIL(h, ESP_ERR_INVALID, "shards: nshards must be in 1..256");
    h->shard_user = true;
    CK(pending_materialize(h));
    h->part_valid = h->part_assembled = false;  // (its tables share scratch arrays with this path)
    if ((double)h->n * (double)P >= 9.0e18) FAIL(h, ESP_ERR_UNSUPPORTED, "shards: n*nshards overflows");
    const i64 E = h->count;
    int bits = 1;
    while ((1 << bits) < P) bits++;
    const i64 T = std::max<i64>(1, ceil_div<i64>(E, espradix::TILE));
    CK(ensure(h, h->seg[0], sizeof(i64) * 4));
    CK(ensure(h, h->tilef[0], sizeof(u64) * (size_t)(4 + espscan::workspace_elems(4))));
    espradix::Pass p;
    p.keys_in = (const u64 *)h->keys.p;
    p.vals_in = (const double *)h->vals.p;
    p.keys_out = nullptr;
    p.vals_out = nullptr;
    p.seg_start = (const i64 *)h->seg[0].p;
    p.tile_first = (const i64 *)h->tilef[0].p;
    p.S = 1;
    p.shift = 0;
    p.bits = bits;
    p.base = 0;
    p.span = ~0ull;
    p.err = nullptr;
    p.owner_P = P;
    p.owner_n = h->n;
    p.colshift = ESP_TAG_BITS + h->L.rb;
    const int R = 1 << bits;
    const i64 hn = T * R;
    CK(ensure(h, h->hist, sizeof(u64) * (size_t)(hn + espscan::workspace_elems(hn))));
    p.hist = (u64 *)h->hist.p;
    if (!(h->shard_valid && h->shard_P == P)) {
        hipLaunchKernelGGL(set_i64_k, dim3(1), dim3(1), 0, h->stream, (i64 *)h->seg[0].p, (i64)0, E, (i64)0, (i64)0);
        hipLaunchKernelGGL(set_i64_k, dim3(1), dim3(1), 0, h->stream, (i64 *)h->tilef[0].p, (i64)0, ceil_div<i64>(E, espradix::TILE), (i64)0, (i64)0);
        HIPCK(h, hipMemsetAsync(p.hist, 0, sizeof(u64) * (size_t)hn, h->stream));
        if (E > 0) {
            Span sp(h, ESP_ST_HIST);
            hipLaunchKernelGGL(espradix::tile_hist_k, dim3(espradix::scatter_grid(T)), dim3(espradix::THREADS), 0, h->stream, p);
            sp.add(1);
        }
        {
            Span sp(h, ESP_ST_SCAN);
            sp.add(espscan::exclusive<u64, false>(h->stream, p.hist, p.hist, hn, p.hist + hn));
        }
        // owner offsets = scanned value of (digit, tile 0); R+1 entries into seg[1]
        CK(ensure(h, h->seg[1], sizeof(i64) * (size_t)(R + 1)));
        hipLaunchKernelGGL(espradix::new_segments_k, dim3(grid_for(R + 1, 256)), dim3(256), 0, h->stream, (const u64 *)p.hist,
                           (const i64 *)h->seg[0].p, (const i64 *)h->tilef[0].p, 1, bits, (i64 *)h->seg[1].p, E);
        HIPCK(h, hipGetLastError());
        h->shard_valid = true;
        h->shard_P = P;
    }
    *out = p;
    return ESP_OK;
}

static int32_t shard_offsets(esp_handle *h, int P, int64_t *offsets /* P+1 */) {
    std::vector<i64> tmp((size_t)P + 1);
    HIPCK(h, hipMemcpyAsync(tmp.data(), h->seg[1].p, sizeof(i64) * (size_t)(P + 1), hipMemcpyDeviceToHost, h->stream));
    HIPCK(h, hipStreamSynchronize(h->stream));
    for (int d = 0; d <= P; d++) offsets[d] = tmp[(size_t)d];
    offsets[P] = h->count;  // digits >= P never occur
    return ESP_OK;
}

extern "C" int32_t esp_shard_counts(esp_handle *h, int32_t nshards, int64_t *counts) {
    if (!h || !counts) return ESP_ERR_INVALID;
    (void)hipSetDevice(h->device);
    espradix::Pass p;
    CK(shard_prepare(h, nshards, &p));
    std::vector<int64_t> off((size_t)nshards + 1);
    CK(shard_offsets(h, nshards, off.data()));
    for (int d = 0; d < nshards; d++) counts[d] = off[(size_t)d + 1] - off[(size_t)d];
    return ESP_OK;
}

extern "C" int32_t esp_shard_export(esp_handle *h, int32_t nshards, uint64_t *d_keys, double *d_vals, int64_t *offsets) {
    if (!h || !offsets) return ESP_ERR_INVALID;
    if (h->count > 0 && (!d_keys || !d_vals)) return ESP_ERR_INVALID;
    (void)hipSetDevice(h->device);
    espradix::Pass p;
    CK(shard_prepare(h, nshards, &p));
    CK(shard_offsets(h, nshards, offsets));
    if (h->count > 0) {
        p.keys_out = (u64 *)d_keys;
        p.vals_out = d_vals;
        Span sp(h, ESP_ST_SCATTER);
        hipLaunchKernelGGL((espradix::scatter_k<false>), dim3(espradix::scatter_grid(ceil_div<i64>(h->count, espradix::TILE))), dim3(espradix::THREADS), 0,
                           h->stream, p);
        sp.add(1);
        HIPCK(h, hipGetLastError());
        HIPCK(h, hipStreamSynchronize(h->stream));
    }
    return ESP_OK;
}

// In-place exchange (avoids copying what a rank owns itself).  The pending entries are partitioned by
// owner: this rank's own chunk goes straight to its final place behind the `recv_lower` entries it
// will receive from lower ranks, the other chunks go, compacted in owner order, to a send region
// behind the new pending area of the same buffers.  The caller exchanges the send region (RCCL) and
// drops the received chunks in with esp_shard_exchange_place.
__global__ void add_digit_delta_k(u64 *hist, i64 T, int R, const i64 *__restrict__ delta) {
    const i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= T * R) return;
    hist[g] += (u64)delta[g / T];
}

extern "C" int32_t esp_shard_exchange_begin(esp_handle *h, int32_t nshards, int32_t self, int64_t recv_lower,
                                            int64_t recv_higher, uint64_t **d_send_keys, double **d_send_vals,
                                            int64_t *send_offsets) {
    if (!h || !d_send_keys || !d_send_vals || !send_offsets) return ESP_ERR_INVALID;
    if (self < 0 || self >= nshards || recv_lower < 0 || recv_higher < 0) FAIL(h, ESP_ERR_INVALID, "esp_shard_exchange_begin: arguments");
    (void)hipSetDevice(h->device);
    espradix::Pass p;
    CK(shard_prepare(h, nshards, &p));
    std::vector<int64_t> off((size_t)nshards + 1);
    CK(shard_offsets(h, nshards, off.data()));
    const i64 E = h->count;
    const i64 own = off[(size_t)self + 1] - off[(size_t)self];
    const i64 others = E - own;
    const i64 newcount = recv_lower + own + recv_higher;
    const i64 SR = newcount;  // send region starts behind the new pending area
    CK(ensure(h, h->keys2, sizeof(u64) * (size_t)(SR + others + 1)));
    CK(ensure(h, h->vals2, sizeof(double) * (size_t)(SR + others + 1)));
    const int R = 1 << p.bits;
    std::vector<i64> delta((size_t)R, 0);
    i64 sent = 0;
    for (int d = 0; d < nshards; d++) {
        const i64 start = off[(size_t)d], cnt = off[(size_t)d + 1] - start;
        send_offsets[d] = sent;
        if (d == self) {
            delta[(size_t)d] = recv_lower - start;
        } else {
            delta[(size_t)d] = SR + sent - start;
            sent += cnt;
        }
    }
    send_offsets[nshards] = sent;
    if (E > 0) {
        const i64 T = ceil_div<i64>(E, espradix::TILE);
        CK(ensure(h, h->seg[1], sizeof(i64) * (size_t)(R + 1)));
        HIPCK(h, hipMemcpyAsync(h->seg[1].p, delta.data(), sizeof(i64) * (size_t)R, hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(add_digit_delta_k, dim3(grid_for(T * R, 256)), dim3(256), 0, h->stream, p.hist, T, R, (const i64 *)h->seg[1].p);
        p.keys_out = (u64 *)h->keys2.p;
        p.vals_out = (double *)h->vals2.p;
        Span sp(h, ESP_ST_SCATTER);
        hipLaunchKernelGGL((espradix::scatter_k<false>), dim3(espradix::scatter_grid(T)), dim3(espradix::THREADS), 0, h->stream, p);
        sp.add(1);
        HIPCK(h, hipGetLastError());
    }
    HIPCK(h, hipStreamSynchronize(h->stream));
    // the partitioned buffers become the pending buffers
    std::swap(h->keys, h->keys2);
    std::swap(h->vals, h->vals2);
    h->cap = (i64)std::min(h->keys.bytes / sizeof(u64), h->vals.bytes / sizeof(double));
    h->count = newcount;
    h->kind_noted = 0;  // entries of other ranks, with kinds of their own, are about to be placed among these
    pending_changed(h);
    if (h->count > 0) h->kind_uniform = -2;
    *d_send_keys = (uint64_t *)h->keys.p + SR;
    *d_send_vals = (double *)h->vals.p + SR;
    return ESP_OK;
}

extern "C" int32_t esp_shard_exchange_place(esp_handle *h, int64_t position, const uint64_t *d_keys, const double *d_vals,
                                            int64_t count) {
    if (!h || position < 0 || count < 0 || position + count > h->count) return ESP_ERR_INVALID;
    if (count == 0) return ESP_OK;
    (void)hipSetDevice(h->device);
    h->kind_uniform = -2;  // (foreign entries: kinds unknown to this handle's bookkeeping)
    h->kind_noted = 0;
    Span sp(h, ESP_ST_COPY);
    HIPCK(h, hipMemcpyAsync((u64 *)h->keys.p + position, d_keys, sizeof(u64) * (size_t)count, hipMemcpyDeviceToDevice, h->stream));
    HIPCK(h, hipMemcpyAsync((double *)h->vals.p + position, d_vals, sizeof(double) * (size_t)count, hipMemcpyDeviceToDevice, h->stream));
    sp.add(2);
    HIPCK(h, hipStreamSynchronize(h->stream));
    return ESP_OK;
}

// ---- mul!(r, A, x) on the device CSC -------------------------------------------------------------
// LinearAlgebra.mul!(r, ext, x) (abstractextendablesparsematrixcsc.jl:179-181 -> SparseArrays; the
// coloured loop of genericmtextendablesparsematrixcsc.jl:124-143 visits the columns in the same
// order): r .= 0, then column by column r[rows[i]] += vals[i]*x[col].  Every r[i] is therefore the
// left-to-right sum over its row's entries in increasing column order, products and sums rounded
// separately.  The device reproduces exactly that with a row-wise view of the CSC: a stable sort of
// the entry indices by row (built once per pattern, values are gathered through it, so numeric
// re-assembly does not invalidate it) and one thread per row adding in column order.  No atomics:
// bit-identical to the reference loop.
__global__ void csr_keys_k(const i64 *__restrict__ colptr, const i64 *__restrict__ rowval, i64 n, u64 *__restrict__ key,
                           double *__restrict__ payload, u64 *__restrict__ colidx) {
    const i64 c = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    for (i64 p = colptr[c] - 1; p < colptr[c + 1] - 1; p++) {
        key[p] = (u64)(rowval[p] - 1) << ESP_TAG_BITS;
        payload[p] = __longlong_as_double((long long)p);
        colidx[p] = (u64)c;
    }
}
__global__ void csr_finish_k(const u64 *__restrict__ skey, const double *__restrict__ spayload, const u64 *__restrict__ colidx, i64 Z,
                             u32 *__restrict__ perm, u32 *__restrict__ tcol, u64 *__restrict__ rowend) {
    const i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= Z) return;
    const u64 p = (u64)__double_as_longlong(spayload[k]);
    perm[k] = (u32)p;
    tcol[k] = (u32)colidx[p];
    const u64 row = skey[k] >> ESP_TAG_BITS;
    if (k == Z - 1 || (skey[k + 1] >> ESP_TAG_BITS) != row) rowend[row + 1] = (u64)(k + 1);
}
// rowptr0 = exclusive-max-scanned row ends shifted by one: entries of row i = [rowptr0[i], rowptr0[i+1])
// row-wise copy of the values (refreshed when nzval changed: one gather per assembly, then every product
// of a solver loop streams it)
__global__ void csr_values_k(const u32 *__restrict__ perm, const double *__restrict__ nzval, i64 Z, double *__restrict__ rval) {
    const i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < Z) rval[k] = nzval[perm[k]];
}
__global__ __launch_bounds__(256) void spmv_rows_k(const u64 *__restrict__ rowptr0, const double *__restrict__ rval,
                                                   const u32 *__restrict__ tcol, const double *__restrict__ x, i64 m,
                                                   double *__restrict__ r) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    double acc = 0.0;  // r .= zero(eltype)
    const u64 b = rowptr0[i + 1], e = rowptr0[i + 2];
    for (u64 k = b; k < e; k++) acc = acc + rval[k] * x[tcol[k]];
    r[i] = acc;
}

static int32_t build_csr(esp_handle *h) {
    const i64 Z = h->nnz, m = h->m;
    const i64 M2 = m + 2;
    CK(ensure(h, h->csr_rowptr, sizeof(u64) * (size_t)(M2 + espscan::workspace_elems(M2))));
    u64 *rowptr = (u64 *)h->csr_rowptr.p;
    HIPCK(h, hipMemsetAsync(rowptr, 0, sizeof(u64) * (size_t)M2, h->stream));
    if (Z > 0) {
        if (Z >= 0xFFFFFFF0ll || h->n >= 0xFFFFFFF0ll) FAIL(h, ESP_ERR_UNSUPPORTED, "esp_mul: the row-wise index holds 32-bit positions and columns");
        CK(ensure(h, h->csr_perm, sizeof(u32) * (size_t)Z));
        CK(ensure(h, h->csr_col, sizeof(u32) * (size_t)Z));
        // scratch: keys A/B, payload A/B, colidx
        CK(ensure(h, h->csr_tmp, sizeof(u64) * (size_t)Z * 5));
        u64 *kA = (u64 *)h->csr_tmp.p, *kB = kA + Z;
        double *vA = (double *)(kB + Z), *vB = vA + Z;
        u64 *colidx = (u64 *)(vB + Z);
        hipLaunchKernelGGL(csr_keys_k, dim3(grid_for(h->n, 256)), dim3(256), 0, h->stream, (const i64 *)h->colptr.p, (const i64 *)h->rowval.p,
                           h->n, kA, vA, colidx);
        CK(ensure(h, h->segs, sizeof(i64) * 8));
        CK(ensure(h, h->misc, 256));
        i64 *segs = (i64 *)h->segs.p;
        const i64 T = ceil_div<i64>(Z, espradix::TILE);
        hipLaunchKernelGGL(set_i64_k, dim3(1), dim3(1), 0, h->stream, segs, (i64)0, Z, (i64)0, T);
        u64 *ki = kA, *ko = kB;
        double *vi = vA, *vo = vB;
        for (int done = 0; done < h->L.rb; done += 8) {  // stable LSD sort by row: columns stay ascending
            espradix::Pass p;
            p.keys_in = ki;
            p.vals_in = vi;
            p.keys_out = ko;
            p.vals_out = vo;
            p.seg_start = segs;
            p.tile_first = segs + 2;
            p.S = 1;
            p.owner_P = 0;
            p.owner_n = 1;
            p.colshift = 0;
            p.base = 0;
            p.span = ~0ull;
            p.err = (u32 *)h->misc.p + 62;
            p.shift = done;
            p.bits = std::min(8, h->L.rb - done);
            CK(partition_pass(h, p, T));
            std::swap(ki, ko);
            std::swap(vi, vo);
        }
        hipLaunchKernelGGL(csr_finish_k, dim3(grid_for(Z, 256)), dim3(256), 0, h->stream, (const u64 *)ki, (const double *)vi,
                           (const u64 *)colidx, Z, (u32 *)h->csr_perm.p, (u32 *)h->csr_col.p, rowptr);
    }
    // rowptr[i+1] holds the end of row i (0 for empty rows): running maximum = start of the next row
    espscan::exclusive<u64, true>(h->stream, rowptr, rowptr, M2, rowptr + M2);
    HIPCK(h, hipGetLastError());
    if (h->csr_tmp.p) {  // the scratch is 5 arrays of nnz: not worth keeping
        HIPCK(h, hipStreamSynchronize(h->stream));
        (void)hipFree(h->csr_tmp.p);
        h->csr_tmp = DevBuf{};
    }
    h->csr_version = h->pattern_version;
    return ESP_OK;
}

extern "C" int32_t esp_mul(esp_handle *h, const double *x, double *r, int32_t on_device) {
    if (!h || !x || !r) return ESP_ERR_INVALID;
    if (h->count != 0) FAIL(h, ESP_ERR_STATE, "esp_mul: pending entries (flush first, like mul!(r, ext, x) does)");
    (void)hipSetDevice(h->device);
    if (!h->csc_valid) CK(init_empty_csc(h));
    CK(fix_tail(h));
    if (h->csr_version != h->pattern_version) {
        CK(build_csr(h));
        h->csr_val_version = 0;
    }
    if (h->csr_val_version != h->values_version && h->nnz > 0) {
        CK(ensure(h, h->csr_val, sizeof(double) * (size_t)h->nnz));
        hipLaunchKernelGGL(csr_values_k, dim3(grid_for(h->nnz, 256)), dim3(256), 0, h->stream, (const u32 *)h->csr_perm.p,
                           (const double *)h->nzval.p, h->nnz, (double *)h->csr_val.p);
        h->csr_val_version = h->values_version;
    }
    const double *dx = x;
    double *dr = r;
    if (!on_device) {
        CK(ensure(h, h->mul_x, sizeof(double) * (size_t)std::max<i64>(h->n, 1)));
        CK(ensure(h, h->mul_r, sizeof(double) * (size_t)std::max<i64>(h->m, 1)));
        HIPCK(h, hipMemcpyAsync(h->mul_x.p, x, sizeof(double) * (size_t)h->n, hipMemcpyHostToDevice, h->stream));
        dx = (const double *)h->mul_x.p;
        dr = (double *)h->mul_r.p;
    }
    if (h->m > 0)
        hipLaunchKernelGGL(spmv_rows_k, dim3(grid_for(h->m, 256)), dim3(256), 0, h->stream, (const u64 *)h->csr_rowptr.p,
                           (const double *)h->csr_val.p, (const u32 *)h->csr_col.p, dx, h->m, dr);
    HIPCK(h, hipGetLastError());
    if (!on_device) HIPCK(h, hipMemcpyAsync(r, dr, sizeof(double) * (size_t)h->m, hipMemcpyDeviceToHost, h->stream));
    HIPCK(h, hipStreamSynchronize(h->stream));
    return ESP_OK;
}

// ---- Dirichlet edits of the assembled CSC (sparsematrixcsc.jl:97-140) --------------------------------
// one thread per column, the same statements as the reference loops (order inside a column kept)
__global__ void mark_dirichlet_k(const i64 *__restrict__ colptr, const i64 *__restrict__ rowval, const double *__restrict__ nzval,
                                 i64 n, double penalty, uint8_t *__restrict__ marker) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint8_t d = 0;
    for (i64 j = colptr[i] - 1; j < colptr[i + 1] - 1; j++)
        if (rowval[j] == i + 1 && nzval[j] >= penalty) d = 1;
    marker[i] = d;
}
__global__ void eliminate_dirichlet_k(const i64 *__restrict__ colptr, const i64 *__restrict__ rowval, double *__restrict__ nzval, i64 n,
                                      const uint8_t *__restrict__ marker) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool mine = marker[i] != 0;
    for (i64 j = colptr[i] - 1; j < colptr[i + 1] - 1; j++) {
        const i64 r = rowval[j] - 1;
        double v = nzval[j];
        if (mine) v = r == i ? 1.0 : 0.0;                 // A[:,i] = 0, A[i,i] = 1
        if (r != i && marker[r] != 0) v = 0.0;            // A[r,:] = 0 for a marked row r
        nzval[j] = v;
    }
}
static int32_t dirichlet_call(esp_handle *h, uint8_t *marker, int32_t on_device, bool mark, double penalty) {
    if (!h || !marker) return ESP_ERR_INVALID;
    if (h->m != h->n) FAIL(h, ESP_ERR_INVALID, "dirichlet: the matrix must be square");
    if (h->count != 0) FAIL(h, ESP_ERR_STATE, "dirichlet: pending entries (flush first)");
    (void)hipSetDevice(h->device);
    if (!h->csc_valid) CK(init_empty_csc(h));
    CK(fix_tail(h));
    const i64 n = h->n;
    uint8_t *dm = marker;
    if (!on_device) {
        CK(ensure(h, h->mul_x, (size_t)std::max<i64>(n, 1)));
        dm = (uint8_t *)h->mul_x.p;
        if (!mark) HIPCK(h, hipMemcpyAsync(dm, marker, (size_t)n, hipMemcpyHostToDevice, h->stream));
    }
    if (n > 0) {
        if (mark)
            hipLaunchKernelGGL(mark_dirichlet_k, dim3(grid_for(n, 256)), dim3(256), 0, h->stream, (const i64 *)h->colptr.p,
                               (const i64 *)h->rowval.p, (const double *)h->nzval.p, n, penalty, dm);
        else
            hipLaunchKernelGGL(eliminate_dirichlet_k, dim3(grid_for(n, 256)), dim3(256), 0, h->stream, (const i64 *)h->colptr.p,
                               (const i64 *)h->rowval.p, (double *)h->nzval.p, n, (const uint8_t *)dm);
        if (!mark) h->values_version++;
    }
    HIPCK(h, hipGetLastError());
    if (!on_device && mark) HIPCK(h, hipMemcpyAsync(marker, dm, (size_t)n, hipMemcpyDeviceToHost, h->stream));
    HIPCK(h, hipStreamSynchronize(h->stream));
    return ESP_OK;
}
extern "C" int32_t esp_mark_dirichlet(esp_handle *h, double penalty, uint8_t *marker, int32_t on_device) {
    return dirichlet_call(h, marker, on_device, true, penalty);
}
extern "C" int32_t esp_eliminate_dirichlet(esp_handle *h, const uint8_t *marker, int32_t on_device) {
    return dirichlet_call(h, const_cast<uint8_t *>(marker), on_device, false, 0.0);
}

// ---- set-up of the point preconditioners on the device CSC -------------------------------------
// jacobi(A) (factorizations/jacobi.jl:5-12): invdiag[i] = one(Tv) / A[i,i]; getindex of a position that is not stored
// gives zero, i.e. Inf.  ilu0(A) (factorizations/ilu0.jl:8-41): idiag[j] = index of the diagonal entry of column j in
// rowval/nzval; xdiag: iteration j of the reference's loop first sets xdiag[j] = 1/nzval[idiag[j]] and then updates
// xdiag[i] for rows i > j -- every such update is overwritten when iteration i sets xdiag[i] itself, so the loop leaves
// xdiag[j] = 1/nzval[idiag[j]] (restated literally in oracle/esparse_oracle.c: orc_ilu0).  One thread per column.
__global__ void diag_setup_k(espfold::Csc c, i64 n, double *__restrict__ inv, i64 *__restrict__ idiag, unsigned long long *__restrict__ missing) {
    const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const i64 pos = c.nnz > 0 ? espfold::csc_find(c, j, j) : -1;
    if (idiag) {
        idiag[j] = pos + 1;
        if (pos < 0) atomicMin(missing, (unsigned long long)(j + 1));
    }
    inv[j] = 1.0 / (pos >= 0 ? c.nzval[pos] : 0.0);
}
static int32_t diag_setup(esp_handle *h, double *inv, int64_t *idiag, int32_t on_device, const char *what) {
    if (!h || !inv) return ESP_ERR_INVALID;
    if (h->m != h->n) FAIL(h, ESP_ERR_INVALID, "%s: the matrix must be square", what);
    if (h->count != 0) FAIL(h, ESP_ERR_STATE, "%s: pending entries (flush first)", what);
    (void)hipSetDevice(h->device);
    if (!h->csc_valid) CK(init_empty_csc(h));
    CK(fix_tail(h));
    const i64 n = h->n;
    if (n == 0) return ESP_OK;
    double *d_inv = inv;
    i64 *d_idiag = idiag;
    if (!on_device) {
        CK(ensure(h, h->mul_x, sizeof(double) * (size_t)n));
        d_inv = (double *)h->mul_x.p;
        if (idiag) {
            CK(ensure(h, h->mul_r, sizeof(i64) * (size_t)n));
            d_idiag = (i64 *)h->mul_r.p;
        }
    }
    CK(ensure(h, h->misc, 256));
    unsigned long long *d_missing = (unsigned long long *)h->misc.p + 20;
    h->pin_scalar[0] = ~0ull;
    HIPCK(h, hipMemcpyAsync(d_missing, h->pin_scalar, 8, hipMemcpyHostToDevice, h->stream));
    espfold::Csc c{(const i64 *)h->colptr.p, (const i64 *)h->rowval.p, (double *)h->nzval.p, h->nnz};
    hipLaunchKernelGGL(diag_setup_k, dim3(grid_for(n, 256)), dim3(256), 0, h->stream, c, n, d_inv, d_idiag, d_missing);
    HIPCK(h, hipGetLastError());
    HIPCK(h, hipMemcpyAsync(h->pin_scalar, d_missing, 8, hipMemcpyDeviceToHost, h->stream));
    if (!on_device) {
        HIPCK(h, hipMemcpyAsync(inv, d_inv, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, h->stream));
        if (idiag) HIPCK(h, hipMemcpyAsync(idiag, d_idiag, sizeof(i64) * (size_t)n, hipMemcpyDeviceToHost, h->stream));
    }
    HIPCK(h, hipStreamSynchronize(h->stream));
    if (idiag && h->pin_scalar[0] != ~0ull)
        FAIL(h, ESP_ERR_INVALID, "%s: column %llu has no stored diagonal entry (the reference reads an undefined idiag there)", what,
             (unsigned long long)h->pin_scalar[0]);
    return ESP_OK;
}
extern "C" int32_t esp_jacobi_setup(esp_handle *h, double *invdiag, int32_t on_device) {
    return diag_setup(h, invdiag, nullptr, on_device, "esp_jacobi_setup");
}
extern "C" int32_t esp_ilu0_setup(esp_handle *h, double *xdiag, int64_t *idiag, int32_t on_device) {
    if (!idiag) return ESP_ERR_INVALID;
    return diag_setup(h, xdiag, idiag, on_device, "esp_ilu0_setup");
}

// ---- partitioned exchange -------------------------------------------------------------------
// The owner partition of esp_shard_exchange_begin and the first partition pass of the local flush are
// ONE pass here: every rank partitions its pending entries by (owner, digit inside the owner's key
// window) with the run-based single pass, sends every other owner its range together with the
// per-digit counts, and the bucket kernel of the receiving rank reads a segment as the concatenation
// of one piece per source rank (rank order, source order inside: the same deterministic order as one
// buffer fed the ranks' streams in turn).  Nothing is moved a second time and the own range is
// never copied.
__global__ void gather_stride_k(const i64 *__restrict__ src, i64 stride, int count, i64 *__restrict__ dst) {
    const int i = threadIdx.x;
    if (i < count) dst[i] = src[(size_t)i * (size_t)stride];
}

__global__ void diff_counts_k(const i64 *__restrict__ bstart, i64 NB, i64 *__restrict__ cnt) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < NB) cnt[i] = bstart[i + 1] - bstart[i];
}

// one workgroup per source q: pstart[q][0..nb] = exclusive scan of that source's per-digit counts
// (q == me: the bucket starts of the own range -- absolute positions in the partitioned buffer -- which the host
// copies into the row with a device-to-device copy; the workgroup only writes the summary)
// summary[q] = entries of source q (q == me: of the own range), summary[P] = start of the own range
__global__ __launch_bounds__(1024) void piece_scan_k(const i64 *const *__restrict__ counts, const i64 *__restrict__ own_bstart, int me,
                                                     i64 nb, i64 *__restrict__ pstart, i64 *__restrict__ summary) {
    __shared__ i64 lw[2][16];
    const int q = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
    i64 *out = pstart + (size_t)q * (size_t)(nb + 1);
    if (q == me) {
        if (t == 0) {
            summary[q] = own_bstart[nb] - own_bstart[0];
            summary[gridDim.x] = own_bstart[0];
        }
        return;
    }
    const i64 *c = counts[q];
    i64 carry = 0;  // (the same in every thread: all of them add up the 16 wave totals of a round)
    int buf = 0;
    for (i64 b0 = 0; b0 < nb; b0 += 1024, buf ^= 1) {
        const i64 d = b0 + t;
        const i64 x = d < nb ? c[d] : 0;
        i64 inc = x;
#pragma unroll
        for (int dlt = 1; dlt < 64; dlt <<= 1) {
            const i64 o = __shfl_up(inc, dlt, 64);
            if (lane >= dlt) inc += o;
        }
        if (lane == 63) lw[buf][w] = inc;
        __syncthreads();  // (one barrier per round: the wave totals alternate between two buffers)
        i64 pre = carry, tot = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const i64 v = lw[buf][i];
            pre += i < w ? v : 0;
            tot += v;
        }
        if (d < nb) out[d] = pre + inc - x;
        carry += tot;
    }
    if (t == 0) {
        out[nb] = carry;
        summary[q] = carry;
    }
}

// *other = 1 when a received entry is not an UPDATE (the UPDATE-only fold of the bucket kernel is then not used)
__global__ void kinds_check_k(const u64 *__restrict__ keys, i64 count, unsigned long long *__restrict__ other) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const bool bad = i < count && (u32)(keys[i] & ESP_TAG_MASK) != (u32)ESP_UPDATE;
    if (__ballot(bad) != 0ull && (threadIdx.x & 63) == 0) atomicOr(other, 1ull);
}

// merged length of every segment; longest one
__global__ void piece_totals_k(const i64 *__restrict__ pstart, int P, i64 nb, unsigned long long *__restrict__ maxlen,
                               unsigned long long *__restrict__ negative) {
    const i64 d = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    i64 tot = 0;
    bool neg = false;
    if (d < nb)
        for (int q = 0; q < P; q++) {
            const i64 len = pstart[(size_t)q * (size_t)(nb + 1) + d + 1] - pstart[(size_t)q * (size_t)(nb + 1) + d];
            neg |= len < 0;
            tot += len;
        }
    if (neg) atomicAdd(negative, 1ull);
#pragma unroll
    for (int o = 32; o; o >>= 1) {  // one atomic per wave
        const i64 x = __shfl_xor(tot, o, 64);
        tot = x > tot ? x : tot;
    }
    if ((threadIdx.x & 63) == 0 && tot > 0) atomicMax(maxlen, (unsigned long long)tot);
}

extern "C" int32_t esp_shard_plan(esp_handle *h, int32_t nshards, int32_t self, int64_t entries_per_shard) {
    if (!h) return ESP_ERR_INVALID;
    h->shard_user = true;
    h->shard_plan.valid = nshards >= 1 && self >= 0 && self < nshards && entries_per_shard >= 0;
    h->shard_plan.P = nshards;
    h->shard_plan.me = self;
    h->shard_plan.eps = entries_per_shard;
    return ESP_OK;
}
extern "C" int32_t esp_debug_last_local_small(const esp_handle *h, int32_t *small) {
    if (!h || !small) return ESP_ERR_INVALID;
    *small = h->last_local_small;
    return ESP_OK;
}
extern "C" int32_t esp_debug_last_shard_source(const esp_handle *h, int32_t *kind) {
    if (!h || !kind) return ESP_ERR_INVALID;
    *kind = h->last_shard_source;
    return ESP_OK;
}

extern "C" int32_t esp_shard_partition(esp_handle *h, int32_t nshards, int32_t self, int64_t entries_per_shard, int32_t *ok,
                                       uint64_t **d_keys, double **d_vals, int64_t **d_counts, int64_t *entry_offsets,
                                       int64_t *digits_per_shard) {
    if (!h || !ok || !d_keys || !d_vals || !d_counts || !entry_offsets || !digits_per_shard) return ESP_ERR_INVALID;
    *ok = 0;
    const int P = nshards;
    if (P < 1 || self < 0 || self >= P || entries_per_shard < 0) FAIL(h, ESP_ERR_INVALID, "esp_shard_partition: arguments");
    (void)hipSetDevice(h->device);
    h->shard_user = true;
    const i64 E = h->count;
    // the producer already partitioned this very batch for this very call (esp_shard_plan): nothing to move
    const bool from_producer = h->pre.valid && h->pre.mw_P == nshards && h->pre.mw_me == self && h->pre.mw_eps == entries_per_shard &&
                               h->pre.E == E && h->pre.key_bytes == 8 && h->force_path != 11;
    if (!from_producer) CK(pending_materialize(h));
    // (a producer's batch whose own range holds 4-byte keys stays described by `pre` until esp_shard_assemble hands it
    // to the bucket kernel: every other reader of the pending keys goes through pending_materialize)
    h->part_own32 = from_producer && h->pre.own32;
    h->part_kind32 = h->pre.kind;
    if (!h->part_own32) h->pre.valid = false;
    h->last_shard_source = 0;
    h->part_valid = h->part_assembled = false;
    h->part_own_update = h->kind_uniform == ESP_UPDATE && h->kind_noted == h->count;
    if (P > esprun::MW_MAX || P > esplocal::MAX_PIECES || h->force_path == 11) return ESP_OK;  // caller uses the plain exchange
    if ((double)h->n * (double)P >= 9.0e18) FAIL(h, ESP_ERR_UNSUPPORTED, "shards: n*nshards overflows");
    if (E >= 0xFFFFFFF0ll) FAIL(h, ESP_ERR_UNSUPPORTED, "esp_shard_partition: too many pending entries");
    // every rank derives the same plan from (n, P, entries_per_shard)
    const MwPlan plan = shard_mw_plan(h, P, entries_per_shard);
    if (!plan.ok) return ESP_OK;  // small or odd problem: plain exchange
    const std::vector<u64> &base = plan.base;
    const int K = plan.K, shift = plan.shift, pb = plan.pb;
    const u64 nb64 = plan.nb64;
    const i64 NB = plan.NB;
    if (from_producer && (h->pre.mw_shift != shift || h->pre.mw_nb != (u32)nb64))
        FAIL(h, ESP_ERR_STATE, "esp_shard_partition: internal error (the producer's plan differs)");
    // tables: bases (<= 64 u64) | owner offsets (<= 65 i64) | counts (NB i64)
    const size_t o_cnt = 256 * 8;
    CK(ensure(h, h->parttab, o_cnt + sizeof(i64) * (size_t)(NB + 1)));
    char *T = (char *)h->parttab.p;
    // (a producer's batch: prepart_begin wrote the window bases; the copy would queue behind the PART launch)
    if (!from_producer) HIPCK(h, hipMemcpyAsync(T, base.data(), sizeof(u64) * (size_t)P, hipMemcpyHostToDevice, h->stream));
    i64 *cnt = (i64 *)(T + o_cnt);
    CK(ensure(h, h->seg[1], sizeof(i64) * (size_t)(NB + 1)));
    CK(ensure(h, h->tilef[1], sizeof(u64) * (size_t)(NB + 1 + espscan::workspace_elems(NB + 1))));
    i64 *bstart = (i64 *)h->seg[1].p;
    std::vector<i64> off((size_t)P + 1, 0);
    if (from_producer) {
        h->last_shard_source = 2;  // (bucket starts in seg[1], entries in place: the PART launch wrote them)
        h->last_run_order = 0;
        h->shard_valid = false;
    } else if (E > 0) {
        h->last_shard_source = 1;
        CK(ensure(h, h->keys2, sizeof(u64) * (size_t)E));
        CK(ensure(h, h->vals2, sizeof(double) * (size_t)E));
        CK(ensure(h, h->misc, 256));
        HIPCK(h, hipMemsetAsync((unsigned long long *)h->misc.p + 24, 0, 64, h->stream));  // maxlen .. flag words
        MultiWin mw{P, (u32)nb64, (const u64 *)T};
        bool took = false, tiles = false;
        i64 ml = 0;
        CK(run_partition(h, (const u64 *)h->keys.p, (const double *)h->vals.p, (u64 *)h->keys2.p, (double *)h->vals2.p, K, pb, bstart,
                         (u64 *)h->tilef[1].p, &tiles, &took, &ml, &mw, shift));
        if (!took) return ESP_OK;  // not a pre-sorted stream: plain exchange
        std::swap(h->keys, h->keys2);
        std::swap(h->vals, h->vals2);
        h->cap = (i64)std::min(h->keys.bytes / sizeof(u64), h->vals.bytes / sizeof(double));
        h->shard_valid = false;
    } else {
        HIPCK(h, hipMemsetAsync(bstart, 0, sizeof(i64) * (size_t)(NB + 1), h->stream));
    }
    // The counts and owner ranges follow from the bucket starts, which the ranking kernel wrote BEFORE the scatter
    // kernel started: they are produced on the second stream (it already waits for the ranking kernel) and this call
    // returns while the entries are still being moved -- the caller's consensus round runs beside the scatter
    // kernel; esp_synchronize() before the key/value arrays are read.
    // (a producer's batch: the bucket starts were final behind ITS ranking kernel, the PART launch may still run)
    hipStream_t qs = (E > 0 && (h->last_run_order == 1 || from_producer) && h->aux && h->aux_ev) ? h->aux : h->stream;
    if (qs == h->aux) HIPCK(h, hipStreamWaitEvent(h->aux, h->aux_ev, 0));  // (recorded right behind the ranking kernel)
    hipLaunchKernelGGL(diff_counts_k, dim3(grid_for(NB, 256)), dim3(256), 0, qs, (const i64 *)bstart, NB, cnt);
    // owner ranges = bucket starts at every multiple of nb
    i64 *d_off = (i64 *)(T + 64 * 8);
    hipLaunchKernelGGL(gather_stride_k, dim3(1), dim3(128), 0, qs, (const i64 *)bstart, (i64)nb64, P + 1, d_off);
    HIPCK(h, hipMemcpyAsync(off.data(), d_off, sizeof(i64) * (size_t)(P + 1), hipMemcpyDeviceToHost, qs));
    HIPCK(h, hipStreamSynchronize(qs));
    for (int r = 0; r <= P; r++) entry_offsets[r] = off[(size_t)r];
    *digits_per_shard = (int64_t)nb64;
    *d_keys = (uint64_t *)h->keys.p;
    *d_vals = (double *)h->vals.p;
    *d_counts = cnt;
    h->part_valid = true;
    h->part_P = P;
    h->part_me = self;
    h->part_shift = shift;
    h->part_nb = (u32)nb64;
    h->part_base = base[(size_t)self];
    h->part_span = (u64)(shard_col0(h->n, P, self + 1) - shard_col0(h->n, P, self)) << h->L.rb;
    *ok = 1;
    return ESP_OK;
}

extern "C" int32_t esp_shard_assemble(esp_handle *h, const uint64_t *const *d_recv_keys, const double *const *d_recv_vals,
                                      const int64_t *const *d_recv_counts, const int64_t *recv_entries, int32_t *ok) {
    if (!h || !d_recv_keys || !d_recv_vals || !d_recv_counts || !recv_entries || !ok) return ESP_ERR_INVALID;
    *ok = 0;
    if (!h->part_valid) FAIL(h, ESP_ERR_STATE, "esp_shard_assemble: no partitioned pending buffer (esp_shard_partition first; no appends in between)");
    (void)hipSetDevice(h->device);
    const int P = h->part_P, me = h->part_me;
    const i64 nb = (i64)h->part_nb;
    const i64 *bstart = (const i64 *)h->seg[1].p + (size_t)me * (size_t)nb;  // own range of the bucket starts
    // pointer table (keys | values | counts of every source) | summary | piece starts
    const size_t o_sum = 192 * 8, o_ps = 256 * 8;
    CK(ensure(h, h->piecetab, o_ps + sizeof(i64) * (size_t)P * (size_t)(nb + 1)));
    char *T = (char *)h->piecetab.p;
    std::vector<const void *> tab(192, nullptr);
    i64 total_recv = 0;
    for (int q = 0; q < P; q++) {
        if (q == me) {
            tab[(size_t)q] = h->keys.p;
            tab[(size_t)P + q] = h->vals.p;
        } else {
            if (recv_entries[q] < 0 || (recv_entries[q] > 0 && (!d_recv_keys[q] || !d_recv_vals[q])) || !d_recv_counts[q])
                FAIL(h, ESP_ERR_INVALID, "esp_shard_assemble: received block %d", q);
            tab[(size_t)q] = d_recv_keys[q];
            tab[(size_t)P + q] = d_recv_vals[q];
            tab[128 + (size_t)q] = d_recv_counts[q];
            total_recv += recv_entries[q];
        }
    }
    HIPCK(h, hipMemcpyAsync(T, tab.data(), 192 * 8, hipMemcpyHostToDevice, h->stream));
    i64 *pstart = (i64 *)(T + o_ps);
    i64 *d_sum = (i64 *)(T + o_sum);
    CK(ensure(h, h->misc, 256));
    unsigned long long *d_maxlen = (unsigned long long *)h->misc.p + 24;
    HIPCK(h, hipMemsetAsync(d_maxlen, 0, 24, h->stream));
    {
        Span sp(h, ESP_ST_SCAN);
        if (h->part_own_update)  // (the received blocks are the cross-shard pairs only: a few launches over little data)
            for (int q = 0; q < P; q++)
                if (q != me && recv_entries[q] > 0) {
                    hipLaunchKernelGGL(kinds_check_k, dim3(grid_for(recv_entries[q], 256)), dim3(256), 0, h->stream, (const u64 *)d_recv_keys[q],
                                       (i64)recv_entries[q], d_maxlen + 2);
                    sp.add(1);
                }
        HIPCK(h, hipMemcpyAsync(pstart + (size_t)me * (size_t)(nb + 1), bstart, sizeof(i64) * (size_t)(nb + 1), hipMemcpyDeviceToDevice, h->stream));
        hipLaunchKernelGGL(piece_scan_k, dim3((unsigned)P), dim3(1024), 0, h->stream, (const i64 *const *)(T + 128 * 8), bstart, me, nb, pstart, d_sum);
        hipLaunchKernelGGL(piece_totals_k, dim3(grid_for(nb, 256)), dim3(256), 0, h->stream, (const i64 *)pstart, P, nb, d_maxlen, d_maxlen + 1);
        sp.add(2);
    }
    std::vector<i64> last((size_t)P + 1);
    HIPCK(h, hipMemcpyAsync(last.data(), d_sum, sizeof(i64) * (size_t)(P + 1), hipMemcpyDeviceToHost, h->stream));
    unsigned long long mx[3];
    HIPCK(h, hipMemcpyAsync(mx, d_maxlen, 24, hipMemcpyDeviceToHost, h->stream));
    HIPCK(h, hipStreamSynchronize(h->stream));
    HIPCK(h, hipGetLastError());
    for (int q = 0; q < P; q++)
        if (q != me && last[(size_t)q] != recv_entries[q])
            FAIL(h, ESP_ERR_STATE, "esp_shard_assemble: block from shard %d holds %lld entries, its digit counts sum to %lld", q,
                 (long long)recv_entries[q], (long long)last[(size_t)q]);
    if (mx[1]) FAIL(h, ESP_ERR_STATE, "esp_shard_assemble: negative digit count in a received block");
    h->part_all_update = h->part_own_update && mx[2] == 0;
    const i64 own_n = last[(size_t)me];
    const i64 own[2] = {last[(size_t)P], last[(size_t)P] + own_n};
    const i64 total = own_n + total_recv;
    if (total >= 0xFFFFFFF0ll) FAIL(h, ESP_ERR_UNSUPPORTED, "esp_shard_assemble: too many entries for one flush");
    if ((i64)mx[0] > esplocal::CAP) {
        // a merged segment does not fit the bucket kernel: hand the entries over as a plain pending
        // buffer (lower ranks, own range, higher ranks) -- the next flush partitions it as usual
        if (h->part_own32) {  // (packed keys for the own range first)
            h->count = h->pre.E;
            CK(pending_materialize(h));
        }
        CK(ensure(h, h->keys2, sizeof(u64) * (size_t)std::max<i64>(total, 1)));
        CK(ensure(h, h->vals2, sizeof(double) * (size_t)std::max<i64>(total, 1)));
        i64 at = 0;
        Span sp(h, ESP_ST_COPY);
        for (int q = 0; q < P; q++) {
            const u64 *sk = q == me ? (const u64 *)h->keys.p + own[0] : d_recv_keys[q];
            const double *sv = q == me ? (const double *)h->vals.p + own[0] : d_recv_vals[q];
            const i64 c = q == me ? own_n : recv_entries[q];
            if (c > 0) {
                HIPCK(h, hipMemcpyAsync((u64 *)h->keys2.p + at, sk, sizeof(u64) * (size_t)c, hipMemcpyDeviceToDevice, h->stream));
                HIPCK(h, hipMemcpyAsync((double *)h->vals2.p + at, sv, sizeof(double) * (size_t)c, hipMemcpyDeviceToDevice, h->stream));
                sp.add(2);
            }
            at += c;
        }
        HIPCK(h, hipStreamSynchronize(h->stream));
        std::swap(h->keys, h->keys2);
        std::swap(h->vals, h->vals2);
        h->cap = (i64)std::min(h->keys.bytes / sizeof(u64), h->vals.bytes / sizeof(double));
        h->count = total;
        h->kind_noted = 0;  // (the received entries carry kinds of their own)
        pending_changed(h);
        if (h->count > 0) h->kind_uniform = -2;
        return ESP_OK;
    }
    h->count = total;
    h->part_total = total;
    h->part_maxlen = (i64)mx[0];
    h->part_own_lo = own[0];
    h->pre.valid = false;  // (the batch is the bucket kernel's now; part_own32 says how its own range is stored)
    h->part_assembled = true;
    *ok = 1;
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

// ------------------------------------------------------------------------ groups (one process per GPU)
#include "group.hpp"
