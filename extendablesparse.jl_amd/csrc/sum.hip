// sum.hip -- libesparse_hip: Base.sum(buffers, csc) as ONE call (esp_flush_sum), and the values-only upload (esp_set_nzval)
// of a plug-in whose CSC stays attached to its handle between flushes
#include "internal.hpp"
#include <thread>

#include <chrono>

namespace {

// the entries of a device CSC as COO-kind pending entries (column-major = the CSC's own order: a pre-sorted stream)
__global__ __launch_bounds__(256) void csc_as_coo_k(const i64 *__restrict__ colptr, const i64 *__restrict__ rowval, const double *__restrict__ nzval,
                                                    i64 n, KeyLayout L, u64 *__restrict__ keys, double *__restrict__ vals) {
    const i64 c = (i64)blockIdx.x * 256 + threadIdx.x;
    if (c >= n) return;
    const i64 a = colptr[c] - 1, b = colptr[c + 1] - 1;
    for (i64 e = a; e < b; e++) {
        keys[e] = esp_pack(L, rowval[e], c + 1, ESP_COO);
        vals[e] = nzval[e];
    }
}
// the buffers of a batched fold: where their packed keys / values lie, how many, where they go in the scratch matrix; a kernel
// over all of them takes 4096 entries per workgroup (blk[j] = first workgroup of buffer j)
struct BufTab {
    const u64 *keys[64];
    const double *vals[64];
    i64 count[64], at[64];
    u64 shift[64];
    u32 blk[65];
    int q;
};
__device__ __forceinline__ int buf_of_block(const BufTab &b, u32 blk) {
    int j = 0;
#pragma unroll
    for (int step = 32; step; step >>= 1)
        if (j + step < b.q && b.blk[j + step] <= blk) j += step;
    return j;
}
// smallest / largest column (0-based) among every buffer's packed keys: out[2 j] = min, out[2 j + 1] = max (initialised to ~0 / 0)
__global__ __launch_bounds__(256) void col_range_k(BufTab b, int colshift, unsigned long long *__restrict__ out_all) {
    __shared__ u64 smin[4], smax[4];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int jb = buf_of_block(b, blockIdx.x);
    const u64 *keys = b.keys[jb];
    const i64 count = b.count[jb], first = (i64)(blockIdx.x - b.blk[jb]) * 4096;
    unsigned long long *out = out_all + 2 * jb;
    u64 lo = ~0ull, hi = 0ull;
    for (i64 i = first + t; i < min(count, first + 4096); i += 256) {
        const u64 c = keys[i] >> colshift;
        lo = c < lo ? c : lo;
        hi = c > hi ? c : hi;
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) {
        const u64 a = (u64)__shfl_xor((long long)lo, o, 64), b = (u64)__shfl_xor((long long)hi, o, 64);
        lo = a < lo ? a : lo;
        hi = b > hi ? b : hi;
    }
    if (lane == 0) smin[w] = lo, smax[w] = hi;
    __syncthreads();
    if (t == 0) {
        for (int i = 1; i < 4; i++) {
            lo = smin[i] < lo ? smin[i] : lo;
            hi = smax[i] > hi ? smax[i] : hi;
        }
        atomicMin(&out[0], (unsigned long long)lo);
        atomicMax(&out[1], (unsigned long long)hi);
    }
}
// the scratch matrix of the batched folds back into COO entries of the buffers' matrix: its column c lies in block j =
// the last one with off[j] <= c, and is column c - off[j] + cmin[j] there
struct BlockTab {
    i64 off[65];
    i64 cmin[64];
    int q;
};
__global__ __launch_bounds__(256) void blocks_as_coo_k(const i64 *__restrict__ colptr, const i64 *__restrict__ rowval, const double *__restrict__ nzval,
                                                       i64 nv, KeyLayout L, BlockTab tab, u64 *__restrict__ keys, double *__restrict__ vals) {
    // entry-parallel (coalesced reads and stores): a workgroup takes 256 columns, their colptr slice in LDS, every entry finds its
    // column by a binary search there (csc_keys_k's scheme)
    __shared__ i64 cp[257];
    const int t = threadIdx.x;
    const i64 c0 = (i64)blockIdx.x * 256;
    for (int q = t; q <= 256; q += 256) cp[q] = colptr[min(c0 + q, nv)] - 1;
    __syncthreads();
    const i64 e0 = cp[0], e1 = cp[256];
    for (i64 e = e0 + t; e < e1; e += 256) {
        int lo = 0, hi = 256;  // invariant cp[lo] <= e < cp[hi]
#pragma unroll
        for (int step = 0; step < 8; step++) {
            const int mid = (lo + hi) >> 1;
            if (cp[mid] <= e)
                lo = mid;
            else
                hi = mid;
        }
        const i64 c = c0 + lo;
        int j = 0;
#pragma unroll
        for (int step = 32; step; step >>= 1)
            if (j + step < tab.q && tab.off[j + step] <= c) j += step;
        keys[e] = esp_pack(L, rowval[e], c - tab.off[j] + tab.cmin[j] + 1, ESP_COO);
        vals[e] = nzval[e];
    }
}
// a buffer's pending entries (packed keys) into the scratch matrix of the batched folds: the same entries, their columns moved by
// `shift` (as it sits in a packed key; modulo 2^64) into the buffer's block of columns
__global__ __launch_bounds__(256) void shift_columns_k(BufTab b, u64 *__restrict__ keys_out, double *__restrict__ vals_out) {
    const int jb = buf_of_block(b, blockIdx.x);
    const u64 *keys_in = b.keys[jb];
    const double *vals_in = b.vals[jb];
    const i64 count = b.count[jb], first = (i64)(blockIdx.x - b.blk[jb]) * 4096, at = b.at[jb];
    const u64 shift = b.shift[jb];
    for (i64 i = first + threadIdx.x; i < min(count, first + 4096); i += 256) {
        keys_out[at + i] = keys_in[i] + shift;
        vals_out[at + i] = vals_in[i];
    }
}

// ---- the buffers' folds as ONE launch (every buffer holds an element batch as sorted items: esp_handle::LazyItems) ----------
// flags[k S + s] = 1 when segment s of buffer k holds items
__global__ __launch_bounds__(256) void pair_flags_k(const esplocal::MultiBuf *__restrict__ mb, int P, i64 S, u64 *__restrict__ flags) {
    const i64 g = (i64)blockIdx.x * 256 + threadIdx.x;
    if (g > (i64)P * S) return;
    u64 f = 0;
    if (g < (i64)P * S) {
        const int k = (int)(g / S);
        const i64 s = g - (i64)k * S;
        const i64 *seg = mb[k].seg;
        f = (seg && seg[s + 1] > seg[s]) ? 1ull : 0ull;
    }
    flags[g] = f;
}
// the non-empty pairs in (buffer, segment) order: pos = exclusive scan of the flags
__global__ __launch_bounds__(256) void pair_list_k(const esplocal::MultiBuf *__restrict__ mb, int P, i64 S, const u64 *__restrict__ pos,
                                                  u32 *__restrict__ vlist) {
    const i64 g = (i64)blockIdx.x * 256 + threadIdx.x;
    if (g >= (i64)P * S) return;
    const int k = (int)(g / S);
    const i64 s = g - (i64)k * S;
    const i64 *seg = mb[k].seg;
    if (seg && seg[s + 1] > seg[s]) vlist[pos[g]] = ((u32)k << esplocal::MULTI_SEG_BITS) | (u32)s;
}

// the longest segment the combine flush would meet with 2 / 4 / 8 neighbouring segments of the folds' plan joined into one
// (one thread per group of 8; out[0..2])
__global__ void coarse_max_k(const i64 *__restrict__ pstart, int P, i64 S, unsigned long long *__restrict__ out) {
    const i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    i64 m2 = 0, m4 = 0, m8 = 0;
    if (g * 8 < S) {
        i64 t[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const i64 s = g * 8 + j;
            t[j] = 0;
            if (s < S)
                for (int q = 0; q < P; q++) t[j] += pstart[(size_t)q * (size_t)(S + 1) + s + 1] - pstart[(size_t)q * (size_t)(S + 1) + s];
        }
#pragma unroll
        for (int j = 0; j < 8; j += 2) m2 = max(m2, t[j] + t[j + 1]);
        m4 = max(t[0] + t[1] + t[2] + t[3], t[4] + t[5] + t[6] + t[7]);
        m8 = t[0] + t[1] + t[2] + t[3] + t[4] + t[5] + t[6] + t[7];
    }
    m2 = (i64)esp_wave_max((u32)m2), m4 = (i64)esp_wave_max((u32)m4), m8 = (i64)esp_wave_max((u32)m8);  // (totals below 2^32: checked by the host)
    if ((threadIdx.x & 63) == 0) {
        atomicMax(out + 0, (unsigned long long)m2);
        atomicMax(out + 1, (unsigned long long)m4);
        atomicMax(out + 2, (unsigned long long)m8);
    }
}
// piece starts of the joined segments: every f-th entry of each piece's row (and its last)
__global__ void coarsen_pieces_k(const i64 *__restrict__ pstart, int P, i64 S, int f, i64 *__restrict__ out) {
    const i64 S2 = S / f;
    const i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (i64)P * (S2 + 1)) return;
    const i64 q = g / (S2 + 1), s2 = g - q * (S2 + 1);
    out[g] = pstart[(size_t)q * (size_t)(S + 1) + (size_t)(s2 * f)];
}

}  // namespace

__global__ void piece_totals_k(const i64 *__restrict__ pstart, int P, i64 nb, unsigned long long *__restrict__ maxlen,
                               unsigned long long *__restrict__ negative);  // shard.hip

// Base.sum when every non-empty buffer holds ONE element-level batch that stayed a list of sorted items, all with the same plan
// (key window, prefix bits, cell size, kind): the p folds are ONE launch of the fused bucket kernel over the non-empty
// (buffer, segment) pairs -- buffer-major, so the records of pair (k, s) lie behind those of every earlier pair -- which emits
// packed COO keys + values straight into dst's buffer and the number of records per pair; a scan of those numbers is the
// piece table of the combine flush, whose bucket kernel reads segment s as the concatenation of every buffer's records for
// it (the PIECES variants of the shard exchange: no CSC per buffer, no copy, no partition).  *served = false: not
// applicable or refused (nothing has happened: the general path below takes the buffers as they are).
static int32_t flush_sum_items(esp_handle *dst, esp_handle *const *xs, int p, int64_t *new_nnz, int32_t *pattern_changed, bool *served) {
    *served = false;
    const auto t_in = std::chrono::steady_clock::now();
    if (p < 2 || p > esplocal::MAX_PIECES || dst->count != 0 || windowed(dst) || dst->shard_user || dst->force_path != ESP_PATH_AUTO) return ESP_OK;
    const esp_handle *ref = nullptr;
    i64 total_in = 0;
    int pb_min = 1 << 20, pb_max = 0;
    for (int k = 0; k < p; k++) {
        const esp_handle *x = xs[k];
        if (x->count == 0) continue;
        const espelem::Args &el = x->lazy.el;
        if (!x->pre.valid || !x->lazy.on || x->lazy.src != 2 || x->pre.tail != 0 || x->pre.E != x->count || x->pre.key_bytes != 4 ||
            x->pre.mw_P != 0 || windowed(x) || x->shard_user || x->force_path != ESP_PATH_AUTO || !el.cellrec || (el.nloc != 3 && el.nloc != 4))
            return ESP_OK;
        if (x->kind_uniform != el.kind || x->kind_noted != x->count || (el.kind != ESP_UPDATE && el.kind != ESP_RAWUPDATE)) return ESP_OK;
        if (!ref) ref = x;
        const espelem::Args &e0 = ref->lazy.el;
        if (x->pre.K != ref->pre.K || x->pre.base != ref->pre.base || x->pre.span != ref->pre.span ||
            el.nloc != e0.nloc || (el.diag != nullptr) != (e0.diag != nullptr) || el.kind != e0.kind || x->L.rb != dst->L.rb)
            return ESP_OK;
        pb_min = std::min(pb_min, x->pre.pb), pb_max = std::max(pb_max, x->pre.pb);
        total_in += x->count;
    }
    if (!ref || ref->pre.base != dst->win_base || ref->pre.span != dst->win_span) return ESP_OK;
    // The buffers' COMMON plan is the coarsest of their own: every handle plans its item partition from its own history
    // (plan_prefix_bits: seen_spread), so two bands of one mesh may come with prefixes that differ by a bit -- the segments of the finer
    // plan are halves of the coarser one's, its items lie in the same order, and every 2^d-th entry of its segment table IS the
    // table of the coarser plan (a joined segment that outgrows the fused kernel is refused by it like any other: general path)
    if (pb_max - pb_min > 3) return ESP_OK;
    dst->last_sum_plan_bits[0] = pb_min, dst->last_sum_plan_bits[1] = pb_max;
    const int K = ref->pre.K, pb = pb_min, rem = K - pb, clb = rem - dst->L.rb;
    if (pb > esplocal::MULTI_SEG_BITS || clb < 0 || clb > esplocal::G3_CL_BITS || clb + dst->L.rb > 32 || dst->L.rb > 30 || rem > 32) return ESP_OK;
    const i64 S = (i64)1 << pb, PS = (i64)p * S;
    if (PS > ((i64)1 << 24) || total_in >= 0xFFFFFFF0ll) return ESP_OK;
    const espelem::Args &e0 = ref->lazy.el;
    // ---- tables: per-buffer arguments | the pair list; flags / positions
    std::vector<esplocal::MultiBuf> mb((size_t)p);
    if (pb_max > pb_min) CK(ensure(dst, dst->tseg, sizeof(i64) * (size_t)p * (size_t)(S + 1)));  // (dst has nothing pending: its tail table is free)
    for (int k = 0; k < p; k++) {
        const esp_handle *x = xs[k];
        esplocal::MultiBuf &b = mb[(size_t)k];
        memset(&b, 0, sizeof b);
        if (x->count == 0) continue;
        const espelem::Args &el = x->lazy.el;
        b.sorted = el.sorted_keys, b.seg = (const i64 *)x->seg[1].p, b.elmat = el.elmat, b.cellrec = el.cellrec;
        b.negate = el.negate, b.low = el.vrb + ESP_TAG_BITS;
        HIPCK(dst, hipStreamSynchronize(x->stream));  // (the buffer's item partition has run on ITS stream)
        if (x->pre.pb > pb) {  // a finer plan: its table at the common prefix
            i64 *coarse_tab = (i64 *)dst->tseg.p + (size_t)k * (size_t)(S + 1);
            hipLaunchKernelGGL(coarsen_pieces_k, dim3(grid_for(S + 1, 256)), dim3(256), 0, dst->stream, (const i64 *)x->seg[1].p, 1, (i64)1 << x->pre.pb,
                               1 << (x->pre.pb - pb), coarse_tab);
            b.seg = coarse_tab;
        }
    }
    const size_t o_list = 4096;
    CK(ensure(dst, dst->heads, o_list + sizeof(u32) * (size_t)PS));
    CK(ensure(dst, dst->hist, sizeof(u64) * (size_t)(PS + 1 + espscan::workspace_elems(PS + 1))));
    CK(ensure(dst, dst->misc, 256));
    esplocal::MultiBuf *d_mb = (esplocal::MultiBuf *)dst->heads.p;
    u32 *d_list = (u32 *)((char *)dst->heads.p + o_list);
    u64 *d_pos = (u64 *)dst->hist.p;
    HIPCK(dst, hipMemcpyAsync(d_mb, mb.data(), sizeof(esplocal::MultiBuf) * (size_t)p, hipMemcpyHostToDevice, dst->stream));
    {
        Span sp(dst, ESP_ST_SCAN);
        hipLaunchKernelGGL(pair_flags_k, dim3(grid_for(PS + 1, 256)), dim3(256), 0, dst->stream, d_mb, p, S, d_pos);
        int l = 1 + espscan::exclusive<u64, false>(dst->stream, d_pos, d_pos, PS + 1, d_pos + PS + 1);
        hipLaunchKernelGGL(pair_list_k, dim3(grid_for(PS, 256)), dim3(256), 0, dst->stream, d_mb, p, S, (const u64 *)d_pos, d_list);
        sp.add(l + 1);
    }
    HIPCK(dst, hipMemcpyAsync(dst->pin_scalar, d_pos + PS, 8, hipMemcpyDeviceToHost, dst->stream));
    HIPCK(dst, hipStreamSynchronize(dst->stream));  // (mb is read by the copy above)
    const i64 SV = (i64)dst->pin_scalar[0];
    if (SV <= 0 || SV > esplocal::MAX_GRID) return ESP_OK;
    // ---- the folds: one launch; records to dst's buffer, counts to the piece table
    CK(reserve_append(dst, total_in));
    const size_t o_ps = 256 * 8;
    CK(ensure(dst, dst->piecetab, o_ps + sizeof(i64) * (size_t)(p * (S + 1) + espscan::workspace_elems(p * (S + 1)))));
    char *TB = (char *)dst->piecetab.p;
    i64 *pstart = (i64 *)(TB + o_ps);
    HIPCK(dst, hipMemsetAsync(pstart, 0, sizeof(i64) * (size_t)(p * (S + 1)), dst->stream));
    const i64 n_gs = (SV >> 8) + 2;
    const i64 G = ((SV + 2 + n_gs + 511) & ~(i64)511) - (SV + 2);
    const i64 tick_at = SV + 2 + G + 256;
    CK(ensure(dst, dst->segout, sizeof(u64) * (size_t)(SV + 2 + G + 512)));
    u64 *status = (u64 *)dst->segout.p;
    HIPCK(dst, hipMemsetAsync(status, 0, sizeof(u64) * (size_t)(SV + 2 + G + 512), dst->stream));
    esplocal::Args a;
    memset(&a, 0, sizeof a);
    a.S = (int)SV;
    a.rem_bits = rem;
    a.base = dst->win_base;
    a.rb = dst->L.rb;
    a.cl_bits = clb;
    a.col_aligned = 1;
    a.mode = ESP_FLUSH_ROUTED;
    a.out_key = (u64 *)dst->keys.p;
    a.out_val = (double *)dst->vals.p;
    a.status = status;
    a.gstatus = status + SV + 2;
    a.ticket = (u32 *)(status + tick_at);
    a.err = (u32 *)(status + SV) + 1;
    a.maxrun_seen = (u32 *)(status + SV) + 2;
    a.total = -1;
    a.kind32 = (u32)e0.kind;
    a.k32_piece = -1;
    a.n_cols = dst->n;
    a.col_end = dst->n;
    a.kind_all = e0.kind;
    {
        Span sp(dst, ESP_ST_LOCAL);
        if (!esplocal::launch_group3_items_multi(e0.nloc, e0.diag != nullptr, d_list, d_mb, pstart, (int)S, (unsigned)SV, dst->stream, a)) return ESP_OK;
        sp.add(1);
    }
    HIPCK(dst, hipMemcpyAsync(dst->pin_scalar, status + (SV - 1), 24, hipMemcpyDeviceToHost, dst->stream));
    HIPCK(dst, hipStreamSynchronize(dst->stream));
    HIPCK(dst, hipGetLastError());
    const u32 err = (u32)(dst->pin_scalar[1] >> 32);
    if (err & 1u) FAIL(dst, ESP_ERR_HIP, "esp_flush_sum: look-back chain timed out inside the bucket kernel");
    if (err & (2u | 4u | 8u)) return ESP_OK;  // (a segment the fused kernel does not take: nothing has happened to the buffers)
    const i64 folded = (i64)(dst->pin_scalar[0] & esplocal::ST_VAL);
    if (folded == 0) return ESP_OK;
    const auto t_folded = std::chrono::steady_clock::now();
    // ---- the combine: piece starts = exclusive scan of the pairs' record counts (buffer-major = the order they were written in)
    unsigned long long *d_maxlen = (unsigned long long *)dst->misc.p + 24;
    HIPCK(dst, hipMemsetAsync(d_maxlen, 0, 16, dst->stream));
    {
        Span sp(dst, ESP_ST_SCAN);
        const i64 np = (i64)p * (S + 1);
        int l = espscan::exclusive<u64, false>(dst->stream, (const u64 *)pstart, (u64 *)pstart, np, (u64 *)pstart + np);
        hipLaunchKernelGGL(piece_totals_k, dim3(grid_for(S, 256)), dim3(256), 0, dst->stream, (const i64 *)pstart, p, S, d_maxlen, d_maxlen + 1);
        sp.add(l + 1);
    }
    std::vector<const void *> tab(192, nullptr);
    for (int k = 0; k < p; k++) tab[(size_t)k] = dst->keys.p, tab[(size_t)(p + k)] = dst->vals.p;
    HIPCK(dst, hipMemcpyAsync(TB, tab.data(), sizeof(void *) * 192, hipMemcpyHostToDevice, dst->stream));
    HIPCK(dst, hipMemcpyAsync(dst->pin_scalar, d_maxlen, 16, hipMemcpyDeviceToHost, dst->stream));
    HIPCK(dst, hipStreamSynchronize(dst->stream));
    i64 merged = (i64)dst->pin_scalar[0];
    if (merged > (i64)esplocal::CAP || dst->pin_scalar[1] != 0) return ESP_OK;
    // The folds' plan cuts the columns for ONE buffer's items; what p bands of a mesh leave per segment is a fraction of the
    // bucket kernel's capacity, and its time follows the number of segments: 2 / 4 / 8 neighbours are joined while the longest
    // joined segment still fits (the pairs' records lie buffer-major, a buffer's segments one behind the other: a joined piece
    // is contiguous) and the kernel still counts the segment's columns in LDS.
    int coarse = 1, rem_c = rem;
    if (S >= 64 && merged * 2 <= (i64)esplocal::CAP) {
        unsigned long long *d_cm = (unsigned long long *)dst->misc.p + 26;
        HIPCK(dst, hipMemsetAsync(d_cm, 0, 24, dst->stream));
        hipLaunchKernelGGL(coarse_max_k, dim3(grid_for((S + 7) / 8, 256)), dim3(256), 0, dst->stream, (const i64 *)pstart, p, S, d_cm);
        HIPCK(dst, hipMemcpyAsync(dst->pin_scalar, d_cm, 24, hipMemcpyDeviceToHost, dst->stream));
        HIPCK(dst, hipStreamSynchronize(dst->stream));
        for (int lf = 3; lf >= 1; lf--)
            if ((i64)dst->pin_scalar[lf - 1] <= (i64)esplocal::CAP && clb + lf <= esplocal::CL_MAX_BITS && rem + lf <= esplocal::MAX_REM_BITS) {
                coarse = 1 << lf;
                rem_c = rem + lf;
                merged = (i64)dst->pin_scalar[lf - 1];
                break;
            }
    }
    i64 S_c = S;
    if (coarse > 1) {
        S_c = S / coarse;
        const i64 np2 = (i64)p * (S_c + 1);
        CK(ensure(dst, dst->hist, sizeof(i64) * (size_t)np2));
        Span sp(dst, ESP_ST_SCAN);
        hipLaunchKernelGGL(coarsen_pieces_k, dim3(grid_for(np2, 256)), dim3(256), 0, dst->stream, (const i64 *)pstart, p, S, coarse, (i64 *)dst->hist.p);
        HIPCK(dst, hipMemcpyAsync(pstart, dst->hist.p, sizeof(i64) * (size_t)np2, hipMemcpyDeviceToDevice, dst->stream));
        sp.add(2);
    }
    note_kind(dst, ESP_COO, folded);
    dst->count = folded;
    pending_changed(dst);
    dst->part_assembled = true;
    dst->part_valid = false;
    dst->part_P = p, dst->part_me = 0, dst->part_shift = rem_c;
    dst->part_nb = (u32)S_c;
    dst->part_base = dst->win_base;
    dst->part_total = folded, dst->part_maxlen = merged, dst->part_own_lo = 0;
    dst->part_all_update = false;
    dst->part_own32 = false;
    const int32_t rc = esp_flush(dst, ESP_FLUSH_ROUTED, new_nnz, pattern_changed);
    if (rc != ESP_OK) return rc;
    dst->last_lazy_items = 2;  // (esp_debug_last_lazy_items: 2 = the folds of a Base.sum ran as one launch over item records)
    dst->last_sum_join = coarse;
    (void)hipStreamSynchronize(dst->stream);
    dst->last_sum_ms[0] = std::chrono::duration<double, std::milli>(t_folded - t_in).count();
    dst->last_sum_ms[1] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_folded).count();
    *served = true;
    return ESP_OK;
}

// Steps 1 + 2 as ONE flush (round 6): the non-empty buffers' pending entries go -- packed keys -- into a scratch handle
// (esp_handle::sumtmp) whose columns are the buffers' OCCUPIED column ranges side by side (buffer k's columns cmin_k .. cmax_k at
// off_k ..: no column is shared, and a buffer that fills a band of the matrix takes a band's worth of columns -- the plan sees dense
// columns, where the buffer's own flush planned for its whole, nearly empty key window and met thousands of columns per segment, the
// bucket kernel's slowest tier).  The scratch handle's flush folds every buffer by itself in one partition + one bucket-kernel pass;
// its CSC, read column after column, is the same list of folded entries in the same order that q flushes and q gathers produced.
// *used = false: not applicable (one buffer, windows / shards / test hooks on a buffer, too many key bits) -- the caller folds one by one.
static int32_t sum_batched_folds(esp_handle *dst, esp_handle *const *xs, int p, i64 *folded, bool *used) {
    *used = false;
    std::vector<int> idx;
    i64 total = 0;
    for (int k = 0; k < p; k++)
        if (xs[k]->count != 0) {
            const esp_handle *x = xs[k];
            if (windowed(x) || x->shard_user || x->part_assembled || x->force_path != ESP_PATH_AUTO) return ESP_OK;
            idx.push_back(k);
            total += x->count;
        }
    const int q = (int)idx.size();
    if (q < 2 || q > 64 || dst->force_path != ESP_PATH_AUTO || total >= 0xFFFFFFF0ll) return ESP_OK;
    // the buffers' packed keys where they can be read, and their column ranges (one launch each, one round trip for all)
    CK(ensure(dst, dst->misc, 256));
    DevBuf &rb_ = dst->sumrange;
    CK(ensure(dst, rb_, sizeof(unsigned long long) * 128));
    std::vector<unsigned long long> init(128);
    for (int j = 0; j < 64; j++) init[(size_t)2 * j] = ~0ull, init[(size_t)2 * j + 1] = 0ull;
    HIPCK(dst, hipMemcpyAsync(rb_.p, init.data(), sizeof(unsigned long long) * 128, hipMemcpyHostToDevice, dst->stream));
    const int colshift = dst->L.rb + ESP_TAG_BITS;
    BufTab bt;
    memset(&bt, 0, sizeof bt);
    bt.q = q;
    i64 blocks = 0;
    for (int j = 0; j < q; j++) {
        esp_handle *x = xs[(size_t)idx[(size_t)j]];
        int32_t rc = pending_materialize(x);
        if (rc == ESP_OK) rc = settle_offset(x);
        if (rc != ESP_OK) {
            dst->err = x->err;
            return rc;
        }
        HIPCK(dst, hipStreamSynchronize(x->stream));  // (the buffer's appends may still be on their way)
        bt.keys[j] = (const u64 *)x->keys.p;
        bt.vals[j] = (const double *)x->vals.p;
        bt.count[j] = x->count;
        bt.blk[j] = (u32)blocks;
        blocks += grid_for(x->count, 4096);
    }
    bt.blk[q] = (u32)blocks;
    hipLaunchKernelGGL(col_range_k, dim3((unsigned)blocks), dim3(256), 0, dst->stream, bt, colshift, (unsigned long long *)rb_.p);
    std::vector<unsigned long long> rng(128);
    HIPCK(dst, hipMemcpyAsync(rng.data(), rb_.p, sizeof(unsigned long long) * 128, hipMemcpyDeviceToHost, dst->stream));
    HIPCK(dst, hipStreamSynchronize(dst->stream));
    BlockTab tab;
    memset(&tab, 0, sizeof tab);
    tab.q = q;
    i64 nv = 0;
    for (int j = 0; j < q; j++) {
        const i64 c0 = (i64)rng[(size_t)2 * j], c1 = (i64)rng[(size_t)2 * j + 1];
        if (c0 < 0 || c1 >= dst->n || c0 > c1) return ESP_OK;  // (a key outside the matrix: the buffer's own flush reports it)
        tab.off[j] = nv;
        tab.cmin[j] = c0;
        nv += c1 - c0 + 1;
    }
    tab.off[q] = nv;
    {
        int cb = 1;
        while (cb < 62 && ((i64)1 << cb) < nv + nv / 4) cb++;
        if (dst->L.rb + cb > 62) return ESP_OK;  // (esp_create would refuse: more than 62 key bits)
    }
    // the scratch handle: at least nv columns (kept while nv stays between a quarter of its columns and all of them)
    esp_handle *t = dst->sumtmp;
    if (t && (t->m != dst->m || t->n < nv || t->n / 4 > nv)) {
        (void)esp_destroy(t);
        dst->sumtmp = t = nullptr;
    }
    if (!t) {
        if (esp_create(dst->m, nv + nv / 4, dst->device, total, &t) != ESP_OK) return ESP_OK;
        dst->sumtmp = t;
    }
    if (t->L.rb != dst->L.rb) return ESP_OK;
    auto fail_t = [&](int32_t rc) {
        dst->err = t->err;
        (void)esp_reset(t);
        return rc;
    };
    {
        int32_t rc = esp_reset(t);
        if (rc == ESP_OK) rc = esp_set_column_window(t, 1, nv);
        if (rc == ESP_OK) rc = reserve_append(t, total);
        if (rc != ESP_OK) return fail_t(rc);
    }
    i64 at = 0;
    for (int j = 0; j < q; j++) {
        esp_handle *x = xs[(size_t)idx[(size_t)j]];
        bt.shift[j] = (u64)(tab.off[j] - tab.cmin[j]) << colshift;  // (modulo 2^64: a block may move down as well as up)
        bt.at[j] = at;
        if (x->kind_uniform >= 0 && x->kind_noted == x->count) note_kind(t, x->kind_uniform, x->count);
        t->count += x->count;
        at += x->count;
    }
    {
        Span sp(dst, ESP_ST_APPEND);
        hipLaunchKernelGGL(shift_columns_k, dim3((unsigned)blocks), dim3(256), 0, t->stream, bt, (u64 *)t->keys.p, (double *)t->vals.p);
        sp.add(1);
    }
    HIPCK(dst, hipGetLastError());
    pending_changed(t);
    int64_t z = 0;
    {
        const int32_t rc = esp_flush(t, ESP_FLUSH_ROUTED, &z, nullptr);
        if (rc != ESP_OK) return fail_t(rc);
    }
    if (z > 0) {
        CK(reserve_append(dst, z));
        HIPCK(dst, hipStreamSynchronize(t->stream));
        Span sp(dst, ESP_ST_APPEND);
        hipLaunchKernelGGL(blocks_as_coo_k, dim3(grid_for(nv, 256)), dim3(256), 0, dst->stream, (const i64 *)t->colptr.p, (const i64 *)t->rowval.p,
                           (const double *)t->nzval.p, nv, dst->L, tab, (u64 *)dst->keys.p + dst->count, (double *)dst->vals.p + dst->count);
        sp.add(1);
        HIPCK(dst, hipGetLastError());
        note_kind(dst, ESP_COO, z);
        dst->count += z;
        pending_changed(dst);
    }
    *folded = z;
    *used = true;
    return ESP_OK;
}

// Base.sum(xmatrices, csc) (sparsematrixdilnkc.jl:397-435; flush! of GenericMTExtendableSparseMatrixCSC,
// genericmtextendablesparsematrixcsc.jl:45-51).  The reference lists the CSC's entries, then every buffer's -- each buffer
// holds ONE value per position, the fold of the calls it received -- and hands the triplets to sparse!(I, J, V, m, n, +):
// the value at (i,j) is ((csc + f_1) + f_2) + ... over the buffers that hold (i,j), in buffer order, zeros kept.  Here:
//   1. every buffer with pending entries is flushed BY ITSELF (its own matrix is empty: the flush is its fold, f_k);
//   2. the folded entries of all buffers are appended to dst as COO entries, buffer after buffer -- p pre-sorted runs;
//   3. ONE routed flush of dst: a COO entry that meets a stored position is added to it in call order, the others fold
//      first-as-it-is-then-added (fold.hpp) -- the sum above, bit for bit -- and new positions are joined in.
// Everything stays on the device: no CSC travels, whatever p is.  The buffers come back empty.
extern "C" int32_t esp_flush_sum(esp_handle *dst, esp_handle *const *xs, int32_t p, int64_t *new_nnz, int32_t *pattern_changed) {
    if (!dst || p < 0 || (p > 0 && !xs)) return ESP_ERR_INVALID;
    if (pattern_changed) *pattern_changed = 0;
    if (dst->count != 0) FAIL(dst, ESP_ERR_STATE, "esp_flush_sum: the destination has pending entries of its own (esp_flush it first)");
    i64 total = 0;
    for (int k = 0; k < p; k++) {
        esp_handle *x = xs[k];
        if (!x || x == dst) FAIL(dst, ESP_ERR_INVALID, "esp_flush_sum: buffer %d is NULL or the destination itself", k);
        if (x->m != dst->m || x->n != dst->n || x->device != dst->device) FAIL(dst, ESP_ERR_INVALID, "esp_flush_sum: buffer %d has another size or device", k);
        if (x->nnz != 0) FAIL(dst, ESP_ERR_STATE, "esp_flush_sum: buffer %d holds a matrix of its own (a buffer is an empty matrix + pending entries)", k);
        for (int q = 0; q < k; q++)
            if (xs[q] == x) FAIL(dst, ESP_ERR_INVALID, "esp_flush_sum: buffer %d is listed twice", k);
        total += x->count;
    }
    (void)hipSetDevice(dst->device);
    if (total == 0) {
        if (new_nnz) *new_nnz = dst->nnz;
        return ESP_OK;
    }
    const bool trace = esp_exp_env("ESP_SUM_TRACE") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_a = now();
    double t_b = t_a, t_c = t_a;
    i64 folded = 0;
    // Steps 1 - 3; every exit goes through the cleanup below: the call is all-or-nothing for dst (a failure leaves its stored
    // matrix as it was and nothing pending) and the buffers come back EMPTY whatever happened.
    bool by_items = false;
    auto run = [&]() -> int32_t {
        // 0. every buffer holds an element batch as sorted items with one common plan: the folds as ONE launch, the combine
        // flush over their records as pieces (flush_sum_items)
        CK(flush_sum_items(dst, xs, p, new_nnz, pattern_changed, &by_items));
        if (by_items) return ESP_OK;
        // 1. every buffer's own fold -- side by side on the host pool's threads (handle.hip: the threads of the bulk host copies; up to
        // eight with the caller's): distinct handles are independent (NOTES/round6.md section 1), and 16 pipelines of a dozen small
        // launches and two or three host round trips each overlap well: 16 buffers of 2 10^6 entries 8.8 -> 6.6 ms, of 2 10^5
        // entries 3.8 -> 1.7 ms (tools/r6_sum_threads.py).  (Rounds 4 / 5 had taken the threads away: what looked like a hazard of
        // concurrent flushes was the bucket kernel's missing barrier.)
        bool batched = false;
        CK(sum_batched_folds(dst, xs, p, &folded, &batched));
        dst->last_sum_batched = batched ? 1 : 0;
        if (batched) {
            t_b = t_c = now();
            return esp_flush(dst, ESP_FLUSH_ROUTED, new_nnz, pattern_changed);
        }
        {
            std::vector<int> idx;
            for (int k = 0; k < p; k++)
                if (xs[k]->count != 0) idx.push_back(k);
            std::vector<int32_t> rcs(idx.size(), ESP_OK);
            std::vector<int64_t> zs(idx.size(), 0);
            host_run_parts((int)idx.size(), [&](int j) {
                esp_handle *x = xs[(size_t)idx[(size_t)j]];
                (void)hipSetDevice(x->device);
                rcs[(size_t)j] = esp_flush(x, ESP_FLUSH_ROUTED, &zs[(size_t)j], nullptr);
            });
            for (size_t j = 0; j < idx.size(); j++) {
                if (rcs[j] != ESP_OK) {
                    dst->err = xs[(size_t)idx[j]]->err;
                    return rcs[j];
                }
                folded += zs[j];
            }
        }
        t_b = now();
        // 2. their entries behind one another in dst's buffer (dst's stream waits for each buffer's flush: esp_flush returned
        // after its last host round trip, but kernels of the buffer's stream may still run)
        if (folded > 0) {
            CK(reserve_append(dst, folded));
            i64 at = dst->count;
            for (int k = 0; k < p; k++) {
                esp_handle *x = xs[k];
                if (x->nnz == 0) continue;
                CK(fix_tail(x));
                HIPCK(dst, hipStreamSynchronize(x->stream));
                Span sp(dst, ESP_ST_APPEND);
                hipLaunchKernelGGL(csc_as_coo_k, dim3(grid_for(x->n, 256)), dim3(256), 0, dst->stream, (const i64 *)x->colptr.p, (const i64 *)x->rowval.p,
                                   (const double *)x->nzval.p, x->n, dst->L, (u64 *)dst->keys.p + at, (double *)dst->vals.p + at);
                sp.add(1);
                at += x->nnz;
            }
            HIPCK(dst, hipGetLastError());
            note_kind(dst, ESP_COO, folded);
            dst->count += folded;
            pending_changed(dst);
        }
        // 3. the one flush that meets the stored matrix
        t_c = now();
        return esp_flush(dst, ESP_FLUSH_ROUTED, new_nnz, pattern_changed);
    };
    const int32_t rc = run();
    if (!by_items) {
        (void)hipStreamSynchronize(dst->stream);
        dst->last_sum_ms[0] = t_b - t_a;
        dst->last_sum_ms[1] = now() - t_b;
    }
    if (trace) {
        (void)hipStreamSynchronize(dst->stream);
        fprintf(stderr, "esp_flush_sum: %d buffers %lld entries -> folds %.3f ms (%lld entries), gather %.3f ms, combine flush %.3f ms\n", p, (long long)total,
                t_b - t_a, (long long)folded, t_c - t_b, now() - t_c);
    }
    // (the buffers are consumed whatever happened: genericmtextendablesparsematrixcsc.jl:47-49 replaces them all)
    (void)hipStreamSynchronize(dst->stream);
    if (rc != ESP_OK && dst->count != 0) {  // (a failure behind the gather: the folded entries must not stay pending in dst)
        const std::string msg = dst->err;
        (void)esp_clear_pending(dst);
        dst->err = msg;
    }
    for (int k = 0; k < p; k++) (void)esp_reset(xs[k]);
    if (dst->sumtmp) (void)esp_reset(dst->sumtmp);  // (its CSC was read by dst's stream, which is idle now)
    return rc;
}

// nzval of the attached CSC := the caller's values (H2D of the values only: the pattern -- colptr, rowval -- is the one
// the handle holds since the caller's last esp_get_csc / esp_set_csc).  What a plug-in whose CSC stays on the device
// between flushes uploads instead of the whole matrix when only nonzeros(A) can have been edited on the host.
extern "C" int32_t esp_debug_last_sum_plan_bits(const esp_handle *h, int32_t *pb_min, int32_t *pb_max) {
    if (!h || !pb_min || !pb_max) return ESP_ERR_INVALID;
    *pb_min = h->last_sum_plan_bits[0];
    *pb_max = h->last_sum_plan_bits[1];
    return ESP_OK;
}
extern "C" int32_t esp_debug_last_sum_batched(const esp_handle *h, int32_t *batched) {
    if (!h || !batched) return ESP_ERR_INVALID;
    *batched = h->last_sum_batched;
    return ESP_OK;
}
extern "C" int32_t esp_debug_last_sum_ms(const esp_handle *h, double *folds_ms, double *combine_ms) {
    if (!h || !folds_ms || !combine_ms) return ESP_ERR_INVALID;
    *folds_ms = h->last_sum_ms[0];
    *combine_ms = h->last_sum_ms[1];
    return ESP_OK;
}

extern "C" int32_t esp_set_nzval(esp_handle *h, const double *nzval) {
    if (!h) return ESP_ERR_INVALID;
    if (h->nnz == 0) return ESP_OK;
    if (!nzval) return ESP_ERR_INVALID;
    (void)hipSetDevice(h->device);
    Span sp(h, ESP_ST_COPY);
    CK(h2d_pipelined(h, h->nzval.p, nzval, sizeof(double) * (size_t)h->nnz));
    sp.add(1);
    h->values_version++;
    return ESP_OK;
}
