// group3.hpp -- the group tier of the bucket kernel (local.hpp) as a kernel of its own with THREE workgroups per CU.
// The group tier sorts 32-bit keys (row - smallest row of the segment | slot | kind) in registers; local_k still moves
// the segment through an 8-byte LDS key array (32 KiB) beside the values (32 KiB) and the radix tier's counters (8 KiB):
// 73 KiB, two workgroups per CU -- and the tier's time follows the workgroups in flight (3-D FEM: one per CU 15.9 ms, two
// 9.7 ms).  Here the counting sort by column writes the 32-bit SORT keys (16 KiB; the segment's smallest row is known
// before the scatter: it is taken while the keys are loaded) and the dense form writes its records back as
// (local column << rb) | row: 52 KiB of LDS and at most 80 registers.
// Serves: a fresh matrix, 4-byte keys (one kind: UPDATE or RAWUPDATE), segments of at most 256 whole columns whose
// (local column, row) fits 32 bits, column runs of at most 128 entries in the shapes 2 / 4 / 8 / 16 lanes x 8 keys with all
// columns of the segment at once (dense form).  A segment outside that reports bit 8 of Args::err and emits nothing; the
// host then runs the flush again with local_k's group-tier kernels (a fresh-matrix flush has changed nothing).
#pragma once
#include "local.hpp"

namespace esplocal {

// NI = 6: segments of at most 3072 entries -- 39 KiB of LDS and 54 registers: FOUR workgroups per CU.  (Not instantiated: the
// headline's stencil segments -- runs of 12, two lanes x 8 keys per column -- take 1.63 ms this way with three workgroups per CU
// and 1.57 ms with four, against 1.56 ms in the register tiers of local_k's small variant on the same box.)
// WIDE: the rows of a segment may lie anywhere (a mesh numbered without locality): every entry's full row in a second
// LDS array (16 KiB more: two workgroups per CU), the run sorted twice (group_columns<..., WIDE>); rows may span 2^29.
// A segment the plain kernel refuses for its rows ALONE reports bit 16 beside bit 8: the host then takes this one.
// HITS: the re-assembly form (group_columns<..., HITS>): a ROUTED flush of additions over a stored pattern the same mesh built;
// the new values go to Args::hits_out, all-or-nothing (bit 64 of Args::err: some column is not what the batch covers).
// K64: the segment's keys are PACKED 8-byte keys of one known kind (Args::kind32) whose bits below the prefix fit 32 -- what the
// flush's own radix passes leave of a shuffled stream of caller-supplied triplets: the load turns them into the 32-bit form
template <int KEYS, int NI = ITEMS, bool WIDE = false, bool HITS = false, bool K64 = false>
__global__ __launch_bounds__(THREADS, WIDE ? 4 : NI == 6 ? 8 : 6) void group3_k(Args a) {
    static_assert(KEYS == 1 || KEYS == 2, "4-byte keys of one kind");
    static_assert(NI == ITEMS || NI == 6, "4096 or 3072 entries per segment");
    constexpr bool UPD = KEYS == 2;
    constexpr int CAPK = THREADS * NI;
    __shared__ u32 skey[CAPK];
    __shared__ double sval[CAPK];
    __shared__ u32 srow[WIDE ? CAPK : 1];
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
    if (t <= WIN + 1 && w0 + t <= (i64)a.S) s_win[t] = a.seg_start[w0 + t];
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
    const int s = esp_uniform_i32(s_seg);
    if (s >= a.S) return;
    const int wbase = w * (NI * ESP_WAVE) + lane;
    const u32 rowmask32 = a.rb >= 32 ? ~0u : ((1u << a.rb) - 1u);
    const u64 rowmask = (1ull << a.rb) - 1ull;
    const bool inwin = s >= w0 && (i64)s + 1 <= w0 + WIN + 1;
    const i64 beg = esp_uniform_i64(inwin ? s_win[s - w0] : a.seg_start[s]);
    const i64 seg_end = esp_uniform_i64(inwin ? s_win[s - w0 + 1] : a.seg_start[s + 1]);
#ifdef ESP_LOCAL_STAMPS
    if (a.stamps && t == 0) a.stamps[(size_t)s * 16 + 0] = wall_clock64();
#endif
    const int n = min((int)(seg_end - beg), CAPK);
    if (a.total >= 0 && s == a.S - 1 && seg_end != a.total && t == 0) atomicOr(a.err, 2u);  // (an entry behind the last column)
    const u64 hi = ((u64)s << a.rem_bits) + a.base;
    const i64 lbeg = n > 0 ? beg : max(beg - 1, (i64)0);
    const int nlast = n > 0 ? n - 1 : 0;
    u32 k[NI];
    {
        const u32 *k32 = reinterpret_cast<const u32 *>(a.keys_in);
        double vraw[NI];
        if constexpr (K64) {
#pragma unroll
            for (int i = 0; i < NI; i++) k[i] = (u32)((a.keys_in[lbeg + min(wbase + i * ESP_WAVE, nlast)] >> ESP_TAG_BITS) - hi);
        } else {
#pragma unroll
            for (int i = 0; i < NI; i++) k[i] = k32[lbeg + min(wbase + i * ESP_WAVE, nlast)];
        }
#pragma unroll
        for (int i = 0; i < NI; i++) vraw[i] = a.vals_in[lbeg + min(wbase + i * ESP_WAVE, nlast)];
        u32 rmin = ~0u, rmax = 0u;
#pragma unroll
        for (int i = 0; i < NI; i++) {
            const int p = wbase + i * ESP_WAVE;
            sval[p] = vraw[i];
            if (p < n) {
                const u32 row = k[i] & rowmask32;
                rmin = min(rmin, row);
                rmax = max(rmax, row);
            }
        }
        rmin = ~esp_wave_max(~rmin);
        rmax = esp_wave_max(rmax);
        if (lane == 0 && n > 0) {
            atomicMin(&s_rmin, rmin);
            atomicMax(&s_rmax, rmax);
        }
    }
#ifdef ESP_LOCAL_STAMPS
    if (a.stamps && t == 0) a.stamps[(size_t)s * 16 + 1] = wall_clock64();
#endif
    // ---- counting sort by local column
    unsigned short slot[NI];
#pragma unroll
    for (int i = 0; i < NI; i++) {
        slot[i] = 0;
        if (wbase + i * ESP_WAVE < n) slot[i] = (unsigned short)atomicAdd(&ccnt[min(k[i] >> a.rb, (u32)(ncl - 1))], 1u);
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
#ifdef ESP_LOCAL_STAMPS
    if (a.stamps && t == 0) a.stamps[(size_t)s * 16 + 2] = wall_clock64();
#endif
    if (t == 0 && maxrun > 16) atomicMax(a.maxrun_seen, maxrun);
    const u32 rmin = s_rmin;
    const int gmode = (UPD || a.kind_all == ESP_UPDATE) ? 1 : a.kind_all == ESP_RAWUPDATE ? 2 : 0;
    // the shape: G lanes x 8 keys per column, every column of the segment at once
    const int G = maxrun <= 16 ? 2 : maxrun <= 32 ? 4 : maxrun <= 64 ? 8 : 16;
    const bool shape_ok = maxrun <= 128 && ncl * G <= THREADS && gmode != 0 && a.stop_after == 0;
    const u32 span = s_rmax - rmin;
    const bool rows_ok = WIDE ? (span >> GROUP_ROW_BITS) < (1u << WIDE_HI_BITS) : span < (1u << GROUP_ROW_BITS);
    const bool fits = n == 0 || (shape_ok && rows_ok);
    // (8: the segment is not this kernel's; 16: ... some segment for its rows: the wide kernel may still take the flush)
    if (!fits && t == 0) atomicOr(a.err, (!WIDE && !rows_ok) ? (8u | 16u) : (8u | 32u));
    if constexpr (HITS) {
        if (n == 0 || !fits) {  // (no entries: then none of the segment's columns may hold a stored entry)
            const i64 c_lo0 = (i64)(hi >> a.rb);
            bool bad = false;
            for (int q = t; q < ncl && n == 0; q += THREADS) {
                const i64 col = c_lo0 + q;
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
        for (int i = 0; i < NI; i++) {
            const int p = wbase + i * ESP_WAVE;
            if (p < n) {
                const u32 rel = (k[i] & rowmask32) - rmin;
                skey[ccnt[min(k[i] >> a.rb, (u32)(ncl - 1))] + slot[i]] =
                    ((WIDE ? (rel & ((1u << GROUP_ROW_BITS) - 1u)) : rel) << SUB_SHIFT) | (u32)((p << ESP_TAG_BITS) | (int)a.kind32);
                if constexpr (WIDE) srow[p] = rel;
            }
        }
        __syncthreads();
        unsigned long long *gstamp = nullptr;
#ifdef ESP_LOCAL_STAMPS
        if (a.stamps) {
            gstamp = a.stamps + (size_t)s * 16;
            if (t == 0) gstamp[12] = wall_clock64();
        }
#endif
        const DenseCtx dcx{&s_early, ctot, &lbs, s, a.late_total ? nullptr : &s_done};
#define ESP_G3_GO(GG)                                                                                                                \
    do {                                                                                                                             \
        if (gmode == 1)                                                                                                              \
            group_columns<GG, 8, CAPK, true, 1, true, u32, WIDE, HITS>(a, skey, sval, ccnt, ncl, rmin, hi, rowmask, gstamp, &dcx, srow);   \
        else                                                                                                                         \
            group_columns<GG, 8, CAPK, true, 2, true, u32, WIDE, HITS>(a, skey, sval, ccnt, ncl, rmin, hi, rowmask, gstamp, &dcx, srow);   \
    } while (0)
        if (G == 2)
            ESP_G3_GO(2);
        else if (G == 4)
            ESP_G3_GO(4);
        else if (G == 8)
            ESP_G3_GO(8);
        else
            ESP_G3_GO(16);
#undef ESP_G3_GO
        dense = true;
    }
    if constexpr (HITS) return;  // (the sums are in hits_out: nothing to emit)
    __syncthreads();  // the records lie dense in skey / sval; the last wave is at the look-back
#ifdef ESP_LOCAL_STAMPS
    if (a.stamps && t == 0) a.stamps[(size_t)s * 16 + 4] = wall_clock64();
#endif
    if (!dense) {  // nothing to emit (an empty segment, or one the host will run again): the chain must still go on
        if (w == 0) {
            const u64 excl = lookback_wave(a, s, 0u, lane);
            if (lane == 0) s_dst = excl;
        }
    } else if (w == WAVES - 1) {
        const u64 excl = lb_complete(a, lbs, s, s_early, lane);
        if (lane == 0) s_dst = excl;
    }
    __syncthreads();
#ifdef ESP_LOCAL_STAMPS
    if (a.stamps && t == 0) a.stamps[(size_t)s * 16 + 6] = wall_clock64();
#endif
    const int total = dense ? (int)s_early : 0;
    const u64 dst = esp_uniform_u64(s_dst);
    // ---- coalesced stores + column-end marks (or colptr itself); a dense key is (local column << rb) | row
    const bool direct = a.colptr_out != nullptr;
    const i64 c_lo = (i64)(hi >> a.rb);
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
#ifdef ESP_LOCAL_STAMPS
    if (a.stamps && t == 0) a.stamps[(size_t)s * 16 + 7] = wall_clock64();
#endif
}

}  // namespace esplocal
