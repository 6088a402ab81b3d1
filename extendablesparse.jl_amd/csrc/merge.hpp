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
constexpr int CT = 128;  // (direct form, measured at config 3: 64 -> 1.69, 128 -> 1.54, 256 -> 1.61, 512 -> 1.85, 1024 -> 1.98 ms)
constexpr int CT_SCAP = 2048;  // stored entries of a tile the staged form has room for (256 columns, 4096: no change)
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
    i64 *out_colptr;        // the new colptr (old + new entries in front), or null: the caller finishes colptr itself
};

// Direct form of a tile (kept for tiles the staged form below has no room for): the tile's new keys are staged in LDS
// (a tile with more than NEWCAP of them searches global memory), every thread handles MB entries per round with all their
// loads requested first, and the searches of a round's entries advance together (a binary search is a chain of
// dependent loads).  Every entry is written straight to its place: consecutive lanes write nearly consecutive places, but
// the holes the stored entries leave are filled by other lanes much later -- the lines leave the L2 half-written and
// come back (config 3: 4.6 GB written for 2.4 GB of output).
constexpr int NEWCAP = 1024, MB = 4;
template <int CTV>
__device__ __forceinline__ void colmerge_direct(const ColArgs &a, i64 c0, int nc, const i64 *s_cp, const u64 *s_ns, const u64 *s_new, bool staged) {
    const int t = threadIdx.x;
    const i64 op0 = s_cp[0], op1 = s_cp[nc];
    const i64 np0 = (i64)s_ns[0], np1 = (i64)s_ns[nc];
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

// Staged form (the common tile: rows below 2^32, at most SCAP stored and NEWCAP new entries): the tile's merged ORDER is
// made in LDS first -- a 2-byte source index per output place -- and the output then leaves as whole lines, place by
// place, each lane gathering its entry from where the index points.
//   phase 1  stored entry sp: row -> s_row[sp] (the only read of old_row), place = sp + #new entries in front  (search in s_new)
//   phase 2  new entry j    :                                     place = j  + #stored entries in front        (search in s_row)
//   phase 3  place o: s_src[o] -> row from s_row / s_new, value from old_val / new_val (their only read) -> out[o]
template <int CTV, int SCAP>
static __global__ __launch_bounds__(THREADS) void colmerge_k(ColArgs a) {
    __shared__ i64 s_cp[CTV + 1];
    __shared__ u64 s_ns[CTV + 1];
    __shared__ u64 s_new[NEWCAP];
    __shared__ u32 s_row[SCAP];
    __shared__ unsigned short s_src[SCAP + NEWCAP];
    const int t = threadIdx.x;
    const i64 c0 = a.c_begin + (i64)blockIdx.x * CTV;
    const int nc = (int)min((i64)CTV, a.c_begin + a.ncols - c0);
    for (int q = t; q <= nc; q += THREADS) {
        s_cp[q] = a.old_colptr[c0 + q] - 1;
        s_ns[q] = a.newstart[c0 + q];
    }
    __syncthreads();
    if (a.out_colptr) {  // (the last tile writes the entry behind the last column too)
        const int last = c0 + nc == a.c_begin + a.ncols ? nc : nc - 1;
        for (int q = t; q <= last; q += THREADS) a.out_colptr[c0 + q] = s_cp[q] + 1 + (i64)s_ns[q];
    }
    const i64 op0 = s_cp[0], op1 = s_cp[nc];
    const i64 np0 = (i64)s_ns[0], np1 = (i64)s_ns[nc];
    const bool staged = np1 - np0 <= (i64)NEWCAP;
    if (staged)
        for (i64 q = np0 + t; q < np1; q += THREADS) s_new[q - np0] = a.new_key[q];
    const bool fast = staged && op1 - op0 <= (i64)SCAP && a.rb <= 32;
    if (!fast) {
        __syncthreads();
        colmerge_direct<CTV>(a, c0, nc, s_cp, s_ns, s_new, staged);
        return;
    }
    const int ns = (int)(op1 - op0), nn = (int)(np1 - np0);
    const u64 rowmask = (1ull << a.rb) - 1ull;
    // (the rows are requested before the barrier the staged new keys need)
    u32 row[(SCAP + THREADS - 1) / THREADS];
    constexpr int SR = (SCAP + THREADS - 1) / THREADS;
#pragma unroll
    for (int i = 0; i < SR; i++) {
        const int sp = i * THREADS + t;
        row[i] = (u32)(a.old_row[op0 + (sp < ns ? sp : (ns > 0 ? ns - 1 : 0))] - 1);
    }
    __syncthreads();
    // ---- phase 1
#pragma unroll
    for (int i = 0; i < SR; i++) {
        const int sp = i * THREADS + t;
        if (sp >= ns) break;
        s_row[sp] = row[i];
        const i64 p = op0 + sp;
        int lo = 0, hi = nc;  // the column: largest c with s_cp[c] <= p
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (s_cp[mid] <= p)
                lo = mid;
            else
                hi = mid;
        }
        int b = (int)((i64)s_ns[lo] - np0), e = (int)((i64)s_ns[lo + 1] - np0);
        while (b < e) {  // new entries of the column with a smaller row
            const int mid = b + ((e - b) >> 1);
            if ((u32)(s_new[mid] & rowmask) < row[i])
                b = mid + 1;
            else
                e = mid;
        }
        s_src[sp + b] = (unsigned short)sp;
    }
    __syncthreads();
    // ---- phase 2
    for (int j = t; j < nn; j += THREADS) {
        const u64 key = s_new[j];
        const int lc = (int)((i64)(key >> a.rb) - c0);
        const u32 r = (u32)(key & rowmask);
        int b = (int)(s_cp[lc] - op0), e = (int)(s_cp[lc + 1] - op0);
        while (b < e) {  // stored entries of the column with a smaller row
            const int mid = b + ((e - b) >> 1);
            if (s_row[mid] < r)
                b = mid + 1;
            else
                e = mid;
        }
        s_src[j + b] = (unsigned short)(0x8000u | (unsigned)j);
    }
    __syncthreads();
    // ---- phase 3
    const i64 ob = op0 + np0;
    const int no = ns + nn;
    for (int base = 0; base < no; base += MB * THREADS) {
        double val[MB];
        i64 r[MB];
#pragma unroll
        for (int i = 0; i < MB; i++) {
            const int o = base + i * THREADS + t;
            const unsigned src = s_src[o < no ? o : no - 1];
            if (src & 0x8000u) {
                const int j = (int)(src & 0x7FFFu);
                val[i] = a.new_val[np0 + j];
                r[i] = (i64)(s_new[j] & rowmask) + 1;
            } else {
                val[i] = a.old_val[op0 + src];
                r[i] = (i64)s_row[src] + 1;
            }
        }
#pragma unroll
        for (int i = 0; i < MB; i++) {
            const int o = base + i * THREADS + t;
            if (o < no) {
                a.out_row[ob + o] = r[i];
                a.out_val[ob + o] = val[i];
            }
        }
    }
}

}  // namespace espmerge
