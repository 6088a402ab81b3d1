// merge.hpp -- merge-path join of the existing CSC with the sorted new entries (K5).
// Restates the 3-way column merge of Base.:+(lnk,csc) (sparsematrixlnk.jl:346-376) as one
// global merge of two key-sorted sequences; equal positions were already resolved by the
// fold (in-place update of csc.nzval), so the two inputs are disjoint here.
// Output tile = 2048 entries per workgroup: diagonal search in HBM (two per tile), the
// tile's inputs staged in LDS, per-thread merge-path in LDS.  HBM-bound:
// reads 12 B (col u32 + row i64... ) see DESIGN.md; writes 16 B per output entry.
#pragma once
#include "common.hpp"

namespace espmerge {

constexpr int THREADS = 256;
constexpr int ITEMS = 8;
constexpr int TILE = THREADS * ITEMS;

struct Args {
    const u32 *old_col;  // 0-based column of every stored entry
    const i64 *old_row;  // 1-based
    const double *old_val;
    i64 Z0;
    const u64 *new_key;  // (col0<<rb)|row0, strictly increasing
    const double *new_val;
    i64 Zn;
    int rb;
    i64 *out_row;  // 1-based
    double *out_val;
};

__device__ __forceinline__ u64 old_key(const Args &a, i64 p) {
    return ((u64)a.old_col[p] << a.rb) | (u64)(a.old_row[p] - 1);
}

// number of old entries among the first `diag` merged entries
__device__ __forceinline__ i64 diag_search(const Args &a, i64 diag) {
    i64 lo = diag > a.Zn ? diag - a.Zn : 0;
    i64 hi = diag < a.Z0 ? diag : a.Z0;
    while (lo < hi) {
        const i64 mid = lo + ((hi - lo) >> 1);
        // take old[mid] before new[diag-1-mid] ?
        if (old_key(a, mid) < a.new_key[diag - 1 - mid])
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

static __global__ __launch_bounds__(THREADS) void merge_k(Args a) {
    __shared__ u64 lk[TILE];
    __shared__ double lv[TILE];
    __shared__ i64 split[2];
    const int t = threadIdx.x;
    const i64 Zt = a.Z0 + a.Zn;
    const i64 d0 = (i64)blockIdx.x * TILE;
    const i64 d1 = min(Zt, d0 + (i64)TILE);
    if (t == 0) split[0] = diag_search(a, d0);
    if (t == 64) split[1] = diag_search(a, d1);
    __syncthreads();
    const i64 a0 = split[0], a1 = split[1];
    const i64 b0 = d0 - a0, b1 = d1 - a1;
    const int na = (int)(a1 - a0), nb = (int)(b1 - b0);
    for (int s = t; s < na; s += THREADS) {
        lk[s] = old_key(a, a0 + s);
        lv[s] = a.old_val[a0 + s];
    }
    for (int s = t; s < nb; s += THREADS) {
        lk[na + s] = a.new_key[b0 + s];
        lv[na + s] = a.new_val[b0 + s];
    }
    __syncthreads();
    const int ntile = na + nb;
    const int dt = min(t * ITEMS, ntile);
    // per-thread diagonal in LDS
    int lo = dt > nb ? dt - nb : 0, hi = dt < na ? dt : na;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (lk[mid] < lk[na + dt - 1 - mid])
            lo = mid + 1;
        else
            hi = mid;
    }
    int ia = lo, ib = dt - lo;
    const u64 rowmask = (1ull << a.rb) - 1ull;
#pragma unroll
    for (int q = 0; q < ITEMS; q++) {
        const int o = dt + q;
        if (o >= ntile) break;
        bool takeA;
        if (ia >= na)
            takeA = false;
        else if (ib >= nb)
            takeA = true;
        else
            takeA = lk[ia] < lk[na + ib];
        const int src = takeA ? ia : na + ib;
        a.out_row[d0 + o] = (i64)(lk[src] & rowmask) + 1;
        a.out_val[d0 + o] = lv[src];
        if (takeA)
            ia++;
        else
            ib++;
    }
}


// ---- column-tiled join ------------------------------------------------------------------------------------
// The same join without the per-entry column array (a scan over all stored entries) and without diagonal searches:
// a workgroup takes CT consecutive columns; every stored entry and every new entry of those columns finds its own
// place in the merged column: output index = own index + number of entries of the OTHER list in front of it
//   stored entry p of column c :  p + newstart[c] + #{new entries of c with a smaller row}
//   new entry q of column c    :  q + (colptr[c]-1) + #{stored entries of c with a smaller row}
// (newstart[c] = new entries in the columns before c: the exclusive max-scan of the bucket kernel's column-end marks.)
// The two lists of a column are disjoint (hits were folded into the stored values) and short, so the counts are
// binary searches of a few steps in cached memory; consecutive lanes take consecutive entries and write nearly
// consecutive places.  Reads 16 B per stored and per new entry + 16 B per column, writes 16 B per output entry.
constexpr int CT = 128;  // (measured at config 3: 64 -> 1.69, 128 -> 1.54, 256 -> 1.61, 512 -> 1.85, 1024 -> 1.98 ms)
struct ColArgs {
    const i64 *old_colptr;  // 1-based values, n + 1
    const i64 *old_row;     // 1-based
    const double *old_val;
    const u64 *newstart;    // n + 1
    const u64 *new_key;     // (col0 << rb) | row0, increasing
    const double *new_val;
    int rb;
    i64 c_begin, ncols;     // columns [c_begin, c_begin + ncols) hold every new entry
    i64 *out_row;
    double *out_val;
};

// The tile's new keys are staged in LDS (a tile of 256 stencil columns gains a few hundred; a tile with more than
// NEWCAP of them searches global memory), every thread handles MB entries per round with all their loads requested
// first, and the searches of a round's entries advance together (a binary search is a chain of dependent loads).
constexpr int NEWCAP = 1024, MB = 4;
static __global__ __launch_bounds__(THREADS) void colmerge_k(ColArgs a) {
    __shared__ i64 s_cp[CT + 1];
    __shared__ u64 s_ns[CT + 1];
    __shared__ u64 s_new[NEWCAP];
    const int t = threadIdx.x;
    const i64 c0 = a.c_begin + (i64)blockIdx.x * CT;
    const int nc = (int)min((i64)CT, a.c_begin + a.ncols - c0);
    for (int q = t; q <= nc; q += THREADS) {
        s_cp[q] = a.old_colptr[c0 + q] - 1;
        s_ns[q] = a.newstart[c0 + q];
    }
    __syncthreads();
    const i64 op0 = s_cp[0], op1 = s_cp[nc];
    const i64 np0 = (i64)s_ns[0], np1 = (i64)s_ns[nc];
    const bool staged = np1 - np0 <= (i64)NEWCAP;
    if (staged)
        for (i64 q = np0 + t; q < np1; q += THREADS) s_new[q - np0] = a.new_key[q];
    __syncthreads();
    const u64 rowmask = (1ull << a.rb) - 1ull;
    // ---- stored entries
    for (i64 base = op0; base < op1; base += (i64)MB * THREADS) {
        i64 row[MB], pos[MB];
        double val[MB];
#pragma unroll
        for (int i = 0; i < MB; i++) {
            const i64 p = base + (i64)i * THREADS + t;
            pos[i] = p < op1 ? p : -1;
            const i64 pc = p < op1 ? p : op1 - 1;  // (clamped: the loads need no branch)
            row[i] = a.old_row[pc];
            val[i] = a.old_val[pc];
        }
#pragma unroll
        for (int i = 0; i < MB; i++) {
            if (pos[i] < 0) continue;
            const i64 p = pos[i];
            int lo = 0, hi = nc;  // the column: largest c with s_cp[c] <= p
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (s_cp[mid] <= p)
                    lo = mid;
                else
                    hi = mid;
            }
            i64 b = (i64)s_ns[lo], e = (i64)s_ns[lo + 1];
            if (staged) {
                while (b < e) {  // new entries of the column with a smaller row
                    const i64 mid = b + ((e - b) >> 1);
                    if ((i64)(s_new[mid - np0] & rowmask) + 1 < row[i])
                        b = mid + 1;
                    else
                        e = mid;
                }
            } else {
                while (b < e) {
                    const i64 mid = b + ((e - b) >> 1);
                    if ((i64)(a.new_key[mid] & rowmask) + 1 < row[i])
                        b = mid + 1;
                    else
                        e = mid;
                }
            }
            const i64 out = p + b;  // = p + newstart[c] + count
            a.out_row[out] = row[i];
            a.out_val[out] = val[i];
        }
    }
    // ---- new entries
    for (i64 base = np0; base < np1; base += (i64)MB * THREADS) {
        i64 row[MB], b[MB], e[MB], pos[MB];
        double val[MB];
#pragma unroll
        for (int i = 0; i < MB; i++) {
            const i64 q = base + (i64)i * THREADS + t;
            pos[i] = q < np1 ? q : -1;
            const i64 qc = q < np1 ? q : np1 - 1;
            const u64 key = staged ? s_new[qc - np0] : a.new_key[qc];
            val[i] = a.new_val[qc];
            const int lc = (int)((i64)(key >> a.rb) - c0);
            row[i] = (i64)(key & rowmask) + 1;
            b[i] = s_cp[lc];
            e[i] = pos[i] >= 0 ? s_cp[lc + 1] : b[i];
        }
        // stored entries of the column with a smaller row: the MB searches step together
        bool more = true;
        while (more) {
            more = false;
            i64 mid[MB], r[MB];
#pragma unroll
            for (int i = 0; i < MB; i++) {
                mid[i] = b[i] + ((e[i] - b[i]) >> 1);
                r[i] = b[i] < e[i] ? a.old_row[mid[i]] : 0;
            }
#pragma unroll
            for (int i = 0; i < MB; i++) {
                if (b[i] < e[i]) {
                    if (r[i] < row[i])
                        b[i] = mid[i] + 1;
                    else
                        e[i] = mid[i];
                }
                more |= b[i] < e[i];
            }
        }
#pragma unroll
        for (int i = 0; i < MB; i++) {
            if (pos[i] < 0) continue;
            const i64 out = pos[i] + b[i];  // = q + (colptr[c]-1) + count
            a.out_row[out] = row[i];
            a.out_val[out] = val[i];
        }
    }
}

}  // namespace espmerge
