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

__global__ __launch_bounds__(THREADS) void merge_k(Args a) {
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

}  // namespace espmerge
