// group3_items.hpp -- group3_k fed with ITEM records: the expansion of an item partition (femitems.hpp, elements.hpp) fused
// into the bucket kernel.
//
// An item partition (shuffled P1 FEM streams: one 8-byte record per (cell, vertex column)) used to end in an expansion
// kernel that stored every update once at its bucket position -- 12 B per update written, and read again by the bucket
// kernel one launch later: 3-D config 4 (1.19e9 updates) 14.3 GB each way, 5.0 + 6.6 ms.  The bucket of an update is a
// function of its item, and an item's W updates are a function of its record: the bucket kernel can form them itself.
// Here a segment reads its <= 4096 / W sorted item records (1.6 B per update), every thread turns one or two of them into
// their W updates -- SRC 1: the built-in generator's cell geometry (fem_column_of_cell), SRC 2: a gather from the caller's
// element matrices and the 64-byte cell records (elem_cells_k) -- straight into the LDS arrays group3_k sorts, and goes on
// as group3_k does (counting sort by local column, group_columns in its dense form, look-back, final stores).  The counting
// sort draws one slot range per ITEM (a fifth of the LDS atomics).  The append buffer is never written: the pending entries
// of such a batch are its sorted items (esp_handle::LazyItems); whoever needs them as entries other than this flush -- an
// append behind the batch, a clone, a flush over a stored pattern, a segment this kernel refuses -- runs the expansion
// kernel first (lazy_expand, produce.hip) and finds the handle as it always was.
//
// Order: inside a segment the items are in stream order (stable partition), an item's updates in call order; the sort key of
// update e of item j carries p = j W + e as its slot index, so (row, p) is (row, stream order) as in group3_k.
#pragma once
#include "elements.hpp"
#include "femitems.hpp"
#include "group3.hpp"

namespace esplocal {

struct ItemArgs {
    const u64 *sorted;      // the sorted item records (segment table: Args::seg_start, in UPDATES = items * W)
    int src;                // 1: generator (single-word records: column | cell), 2: element arrays (column | cell * nloc + jl)
    int low;                // record bits below the column: L.rb + 2 (src 1), vrb + 2 (src 2)
    espgen::FemArgs fem;    // src 1
    const double *elmat;    // src 2: Float64 nloc x nloc x ncells
    const char *cellrec;    // src 2: 64-byte cell records (rows as four u32 | the cell's diag values)
    int negate;             // src 2: op = '-'
    // MULTI (src 2): the launch's tickets are the non-empty (buffer, segment) pairs of several buffers
    const u32 *vlist;       // pair of ticket v: buffer << MULTI_SEG_BITS | segment
    const MultiBuf *mbuf;   // per buffer: sorted items, segment table, element matrices, cell records
    i64 *counts;            // records the pair emitted: counts[buffer * (S_real + 1) + segment]
    int S_real;             // segments per buffer
};

// NLOC = nodes per cell (3 / 4), DIAG: an item carries the diagonal's term (W = NLOC + 1)
// HITS: group3_k's re-assembly form (a ROUTED flush of additions over the pattern the same mesh built: the sums go to
// Args::hits_out, all-or-nothing -- bit 64 of Args::err) fed the same way: a time step of an instationary code never
// writes its updates anywhere
// MULTI: Base.sum over several buffers (esp_flush_sum): ticket v is the pair vlist[v] = (buffer, segment); the folded records of
// the pair go out as PACKED keys of kind COO + values at the look-back offset -- pair after pair, buffer-major: the pieces of the
// combine flush, which reads segment s as the concatenation of every buffer's records for it (no CSC per buffer, no partition)
template <int SRC, int NLOC, bool DIAG, bool HITS = false, bool MULTI = false>
__global__ __launch_bounds__(THREADS, 6) void group3_items_k(Args a, ItemArgs ia) {
    static_assert(!MULTI || (SRC == 2 && !HITS), "several buffers: element arrays, fresh folds");
    constexpr int W = NLOC + (DIAG ? 1 : 0);
    constexpr int NI = ITEMS;
    constexpr int CAPK = THREADS * NI;
    constexpr int MAXIT = CAPK / W;                           // items of a segment
    constexpr int IPT = (MAXIT + THREADS - 1) / THREADS;      // items per thread
    static_assert(SRC == 1 || SRC == 2, "generator or element arrays");
    static_assert(SRC == 2 || DIAG, "the generator's items carry the mass term");
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
    __shared__ u32 skey[CAPK];
    __shared__ double sval[CAPK];
    __shared__ u32 ccnt[(1 << G3_CL_BITS) + 4];
    __shared__ unsigned short ctot[1 << G3_CL_BITS];
    __shared__ u32 lw[16];
    __shared__ u64 s_dst;
    __shared__ int s_seg;
    __shared__ u32 s_early, s_rmin, s_rmax, s_done;
    __shared__ i64 s_win[66];

    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    constexpr int WIN = 64;
    const i64 w0 = max((i64)0, a.first + (i64)blockIdx.x - WIN / 2);
    if constexpr (!MULTI) {
        if (t <= WIN + 1 && w0 + t <= (i64)a.S) s_win[t] = a.seg_start[w0 + t];
    }
    if (t == 0) {
        s_seg = (int)atomicAdd(a.ticket, 1u);
        s_early = 0;
        s_done = 0;
        s_rmin = ~0u;
        s_rmax = 0u;
    }
    const int ncl = 1 << a.cl_bits;
    for (int q = t; q <= ncl; q += THREADS) ccnt[q] = 0;
    __syncthreads();
    const int vs = esp_uniform_i32(s_seg);  // the ticket: the segment's place in the look-back chain
    if (vs >= a.S) return;
    int s = vs, kb = 0;
    const u64 *sorted = ia.sorted;
    const double *elmat = ia.elmat;
    const char *cellrec = ia.cellrec;
    int negate = ia.negate, low = ia.low;
    const i64 *segtab = a.seg_start;
    if constexpr (MULTI) {
        const u32 v = (u32)esp_uniform_i32((int)ia.vlist[vs]);
        kb = (int)(v >> MULTI_SEG_BITS);
        s = (int)(v & ((1u << MULTI_SEG_BITS) - 1u));
        const MultiBuf &mb = ia.mbuf[kb];
        sorted = mb.sorted, elmat = mb.elmat, cellrec = mb.cellrec, negate = mb.negate, segtab = mb.seg;
        low = mb.low;
    }
    const u32 rowmask32 = a.rb >= 32 ? ~0u : ((1u << a.rb) - 1u);
    const u64 rowmask = (1ull << a.rb) - 1ull;
    const bool inwin = !MULTI && s >= w0 && (i64)s + 1 <= w0 + WIN + 1;
    const i64 beg = esp_uniform_i64(inwin ? s_win[s - w0] : segtab[s]);
    const i64 seg_end = esp_uniform_i64(inwin ? s_win[s - w0 + 1] : segtab[s + 1]);
    const int n = min((int)(seg_end - beg), MAXIT * W);
    const int nit = n / W;                     // (a segment holds whole items)
    const i64 ibeg = beg / W;
    if (a.total >= 0 && s == a.S - 1 && seg_end != a.total && t == 0) atomicOr(a.err, 2u);  // (an entry behind the last column)
    const u64 hi = ((u64)s << a.rem_bits) + a.base;
    const i64 c_lo = (i64)(hi >> a.rb);
    // ---- the items of the segment -> their updates: values to sval[j W + e], rows kept for the sort keys
    u32 rows[IPT][W];
    u32 lcol[IPT];
    unsigned short slot[IPT];
    {
        u32 rmin = ~0u, rmax = 0u;
#pragma unroll
        for (int i = 0; i < IPT; i++) {
            const int j = t + i * THREADS;
            lcol[i] = 0;
            slot[i] = 0;
#pragma unroll
            for (int e = 0; e < W; e++) rows[i][e] = 0;
            if (j < nit) {
                const u64 rec = sorted[ibeg + j];
                const i64 col0 = (i64)(rec >> low);
                lcol[i] = (u32)min(max(col0 - c_lo, (i64)0), (i64)(ncl - 1));
                double *vout = sval + j * W;
                if constexpr (SRC == 1) {
                    const i64 cell = (i64)(rec & ((1ull << low) - 1ull));
                    espgen::fem_column_of_cell(ia.fem, cell, col0 + 1, [&](int il, int jl, i64 row, double v) {
                        const int at = jl < 0 ? il : il + (il >= jl ? 1 : 0);
#pragma unroll
                        for (int e = 0; e < W; e++)
                            if (e == at) rows[i][e] = (u32)(row - 1);
                        vout[at] = v;
                    });
                } else {
                    const u32 p = (u32)(rec & ((1ull << low) - 1ull));
                    const u32 cell = p / (u32)NLOC, jl = p - cell * (u32)NLOC;
                    const double *em = elmat + (i64)p * NLOC;  // column jl of the cell's element matrix
                    const char *cr = cellrec + (i64)cell * 64;
                    const u32x4 rr = *reinterpret_cast<const u32x4 *>(cr);
                    double d = 0.0;
                    if constexpr (DIAG) d = *reinterpret_cast<const double *>(cr + 16 + 8 * jl);
                    double v[NLOC];
#pragma unroll
                    for (int il = 0; il < NLOC; il++) v[il] = em[il];
                    const u32 r[4] = {rr.x, rr.y, rr.z, rr.w};
                    if (negate) {
                        d = -d;
#pragma unroll
                        for (int il = 0; il < NLOC; il++) v[il] = -v[il];
                    }
                    // (call order of the column's entries: row il's term at il, +1 from the diagonal's row on -- the diagonal's
                    // term comes right before it: femtools.jl:64)
#pragma unroll
                    for (int il = 0; il < NLOC; il++) {
                        if constexpr (DIAG) {
                            const int at = il + ((u32)il > jl ? 1 : 0);
                            if ((u32)il == jl) {
#pragma unroll
                                for (int e = 0; e < W; e++)
                                    if (e == il) rows[i][e] = r[il], vout[e] = d;
#pragma unroll
                                for (int e = 0; e < W; e++)
                                    if (e == il + 1) rows[i][e] = r[il], vout[e] = v[il];
                            } else {
#pragma unroll
                                for (int e = 0; e < W; e++)
                                    if (e == at) rows[i][e] = r[il], vout[e] = v[il];
                            }
                        } else {
                            rows[i][il] = r[il];
                            vout[il] = v[il];
                        }
                    }
                }
                slot[i] = (unsigned short)atomicAdd(&ccnt[lcol[i]], (u32)W);
#pragma unroll
                for (int e = 0; e < W; e++) {
                    const u32 row = rows[i][e] & rowmask32;
                    rmin = min(rmin, row);
                    rmax = max(rmax, row);
                }
            }
        }
        rmin = ~esp_wave_max(~rmin);
        rmax = esp_wave_max(rmax);
        if (lane == 0 && nit > 0) {
            atomicMin(&s_rmin, rmin);
            atomicMax(&s_rmax, rmax);
        }
    }
    __syncthreads();
    u32 maxrun = 0;
    {
        // exclusive scan of the column counts (<= 256 columns: four per lane of the first wave) + the longest run
        if (w == 0) {
            u32 v[4], run = 0, mx = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int q = lane * 4 + j;
                const u32 x = q < ncl ? ccnt[q] : 0u;
                mx = max(mx, x);
                v[j] = run;
                run += x;
            }
            const u32 inc = esp_wave_scan_add(run);
            mx = esp_wave_max(mx);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int q = lane * 4 + j;
                if (q < ncl) ccnt[q] = inc - run + v[j];
            }
            if (lane == 0) {
                ccnt[ncl] = (u32)n;
                lw[8] = mx;
            }
        }
        __syncthreads();
        maxrun = lw[8];
    }
    if (t == 0 && maxrun > 16) atomicMax(a.maxrun_seen, maxrun);
    const u32 rmin = s_rmin;
    const int gmode = a.kind_all == ESP_UPDATE ? 1 : a.kind_all == ESP_RAWUPDATE ? 2 : 0;
    const int G = maxrun <= 16 ? 2 : maxrun <= 32 ? 4 : maxrun <= 64 ? 8 : 16;
    const bool shape_ok = maxrun <= 128 && ncl * G <= THREADS && gmode != 0 && a.stop_after == 0;
    const u32 span = s_rmax - rmin;
    const bool rows_ok = span < (1u << GROUP_ROW_BITS);
    const bool fits = n == 0 || (shape_ok && rows_ok);
    // (8: the segment is not this kernel's -- the host expands the items and runs the flush again with the other kernels)
    if (!fits && t == 0) atomicOr(a.err, !rows_ok ? (8u | 16u) : (8u | 32u));
    if constexpr (HITS) {
        if (n == 0 || !fits) {  // (no entries: then none of the segment's columns may hold a stored entry)
            bool bad = false;
            for (int q = t; q < ncl && n == 0; q += THREADS) {
                const i64 col = c_lo + q;
                bad = bad || (col < a.n_cols && a.csc.colptr[col + 1] != a.csc.colptr[col]);
            }
            if (bad) atomicOr(a.err, 64u);
            return;
        }
    }
    LbState lbs;
    lb_init(lbs, 0);
    bool dense = false;
    if (n > 0 && fits) {
#pragma unroll
        for (int i = 0; i < IPT; i++) {
            const int j = t + i * THREADS;
            if (j < nit) {
                const u32 at = ccnt[lcol[i]] + (u32)slot[i];
#pragma unroll
                for (int e = 0; e < W; e++) {
                    const u32 rel = (rows[i][e] & rowmask32) - rmin;
                    skey[at + e] = (rel << SUB_SHIFT) | (u32)(((j * W + e) << ESP_TAG_BITS) | (int)a.kind32);
                }
            }
        }
        __syncthreads();
        const DenseCtx dcx{&s_early, ctot, &lbs, vs, a.late_total ? nullptr : &s_done};
#define ESP_G3I_GO(GG)                                                                                                               \
    do {                                                                                                                             \
        if (gmode == 1)                                                                                                              \
            group_columns<GG, 8, CAPK, true, 1, true, u32, false, HITS>(a, skey, sval, ccnt, ncl, rmin, hi, rowmask, nullptr, &dcx, nullptr); \
        else                                                                                                                         \
            group_columns<GG, 8, CAPK, true, 2, true, u32, false, HITS>(a, skey, sval, ccnt, ncl, rmin, hi, rowmask, nullptr, &dcx, nullptr); \
    } while (0)
        if (G == 2)
            ESP_G3I_GO(2);
        else if (G == 4)
            ESP_G3I_GO(4);
        else if (G == 8)
            ESP_G3I_GO(8);
        else
            ESP_G3I_GO(16);
#undef ESP_G3I_GO
        dense = true;
    }
    if constexpr (HITS) return;  // (the sums are in hits_out: nothing to emit)
    __syncthreads();  // the records lie dense in skey / sval; the last wave is at the look-back
    if (!dense) {  // nothing to emit (an empty segment, or one the host will run again): the chain must still go on
        if (w == 0) {
            const u64 excl = lookback_wave(a, vs, 0u, lane);
            if (lane == 0) s_dst = excl;
        }
    } else if (w == WAVES - 1) {
        const u64 excl = lb_complete(a, lbs, vs, s_early, lane);
        if (lane == 0) s_dst = excl;
    }
    __syncthreads();
    const int total = dense ? (int)s_early : 0;
    const u64 dst = esp_uniform_u64(s_dst);
    if constexpr (MULTI) {  // the pair's records as packed COO keys + values; how many: for the combine flush's piece table
        for (int p = t; p < total; p += THREADS) {
            const u32 key = skey[p];
            const u64 col0 = (u64)c_lo + (u64)(key >> a.rb);
            a.out_key[dst + p] = (((col0 << a.rb) | (u64)(key & rowmask32)) << ESP_TAG_BITS) | (u64)ESP_COO;
            a.out_val[dst + p] = sval[p];
        }
        if (t == 0) ia.counts[(size_t)kb * (size_t)(ia.S_real + 1) + (size_t)s] = (i64)total;
        return;
    }
    // ---- coalesced stores + column-end marks (or colptr itself); a dense key is (local column << rb) | row
    const bool direct = a.colptr_out != nullptr;
    const i64 c_hi = direct ? min(c_lo + ((i64)1 << a.cl_bits), a.col_end) : c_lo;
    for (int p = t; p < total; p += THREADS) {
        const u32 key = skey[p];
        a.out_row[dst + p] = (i64)(key & rowmask32) + 1;
        a.out_val[dst + p] = sval[p];
        const i64 col = c_lo + (i64)(key >> a.rb);
        if (direct) {
            // first entry of its column: that column and the empty ones in front of it start here
            const i64 prev = p == 0 ? c_lo - 1 : c_lo + (i64)(skey[p - 1] >> a.rb);
            for (i64 c = max(prev + 1, c_lo); c <= min(col, c_hi - 1); c++) a.colptr_out[c] = (i64)(dst + (u64)p) + 1;
        } else if ((p == total - 1 || (skey[p + 1] >> a.rb) != (key >> a.rb)) && col < a.n_cols) {
            a.colend[col] = dst + (u64)p + 1;  // (a segment is a whole number of columns)
        }
    }
    if (direct) {  // the columns behind the last entry (all of them for an empty segment)
        const i64 after = total > 0 ? c_lo + (i64)(skey[total - 1] >> a.rb) + 1 : c_lo;
        for (i64 c = after + t; c < c_hi; c += THREADS) a.colptr_out[c] = (i64)(dst + (u64)total) + 1;
        if (s == a.S - 1 && t == 0) a.colptr_out[a.col_end] = (i64)(dst + (u64)total) + 1;
    }
}

}  // namespace esplocal
