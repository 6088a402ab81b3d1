// fold.hpp -- ordered segmented fold of duplicates on globally sorted entries (K3),
// compaction and colptr construction (K4).  General path: works for any segment
// length and any duplicate count; the LDS bucket kernel (local.hpp) is the fast path.
//
// Fold state machine per (col,row), entries taken in append order (the sort is stable):
//   absent  --SET v!=0--> present(v)          sparsematrixlnk.jl:184-188,196-199
//   absent  --UPDATE v!=0--> present(0+v)     sparsematrixlnk.jl:212-216,223-226
//   absent  --RAWUPDATE v--> present(0+v)     sparsematrixlnk.jl:239-243,249-250
//   absent  --SET/UPDATE 0--> absent          sparsematrixlnk.jl:196,223
//   present --SET v--> present(v)             sparsematrixlnk.jl:192-195
//   present --UPDATE/RAWUPDATE v--> present(old+v)   sparsematrixlnk.jl:219-221,246-247
//   absent  --COO v--> present(v), present --COO v--> present(old+v): sparse(I,J,V,m,n,+) of the COO
//                                             constructors (extendable.jl:92-104; SparseArrays stdlib)
// An (i,j) found in the CSC starts `present` with csc.nzval (ROUTED mode,
// extendable.jl:164-166,188-189,210-211) and is written back in place.
#pragma once
#include "common.hpp"

namespace espfold {

constexpr int THREADS = 256;

struct Csc {
    const i64 *colptr;  // n+1, 1-based values
    const i64 *rowval;  // 1-based
    double *nzval;
    i64 nnz;
};

__device__ __forceinline__ void fold_step(bool &present, double &acc, u32 kind, double v) {
    if (kind == ESP_SET) {
        if (present || v != 0.0) {
            present = true;
            acc = v;
        }
    } else if (present) {
        acc = acc + v;
    } else if (kind == ESP_COO) {
        present = true;
        acc = v;
    } else if (kind == ESP_RAWUPDATE || v != 0.0) {
        present = true;
        acc = 0.0 + v;
    }
}

// The same state machine without branches (selects only): used where 64 lanes run it in lock step on
// different column runs.  Bit-identical to fold_step (the unused sum of an absent, non-creating
// update is simply discarded).
__device__ __forceinline__ void fold_step_sel(bool &present, double &acc, u32 kind, double v) {
    const bool nz = v != 0.0;
    const bool set = kind == ESP_SET;
    const bool creates = set ? nz : (kind >= ESP_RAWUPDATE || nz);  // RAWUPDATE and COO always create
    const bool np = present || creates;
    const double sumv = (present ? acc : 0.0) + v;
    // (0.0 + v == v bit for bit unless v is -0.0: a COO entry starts from v itself)
    const double acc_add = np ? ((kind == ESP_COO && !present) ? v : sumv) : acc;
    const double acc_set = (present || nz) ? v : acc;
    acc = set ? acc_set : acc_add;
    present = np;
}

// The state machine when the entry is known to be an UPDATE: an absent position holds acc == +0.0, and
// +0.0 + v is v for a creating update and +0.0 for v == +-0.0 -- the sum can be taken unconditionally,
// bit for bit the result of fold_step.
__device__ __forceinline__ void fold_step_update(bool &present, double &acc, double v) {
    acc = acc + v;
    present = present || (v != 0.0);
}

// findindex(csc,i,j) (sparsematrixcsc.jl:7-23) with 0-based row0/col0; returns 0-based
// position in rowval/nzval or -1
__device__ __forceinline__ i64 csc_find(const Csc &c, i64 col0, i64 row0) {
    i64 lo = c.colptr[col0] - 1, hi = c.colptr[col0 + 1] - 1;
    const i64 want = row0 + 1;
    while (lo < hi) {
        i64 mid = lo + ((hi - lo) >> 1);
        if (c.rowval[mid] < want)
            lo = mid + 1;
        else
            hi = mid;
    }
    if (lo < c.colptr[col0 + 1] - 1 && c.rowval[lo] == want) return lo;
    return -1;
}

// one thread per sorted entry; segment heads fold their whole run left to right.
// flag[i]=1 when head i emits a new entry (value in fval[i]); flag has E+1 slots.
static __global__ __launch_bounds__(THREADS) void fold_k(const u64 *__restrict__ sk,
                                                  const double *__restrict__ sv, i64 E, Csc csc,
                                                  int rb, int mode, u32 *__restrict__ flag,
                                                  double *__restrict__ fval) {
    const i64 i = (i64)blockIdx.x * THREADS + threadIdx.x;
    if (i > E) return;
    if (i == E) {
        flag[i] = 0;
        return;
    }
    const u64 k = sk[i];
    const u64 key = k >> ESP_TAG_BITS;
    if (i > 0 && (sk[i - 1] >> ESP_TAG_BITS) == key) {
        flag[i] = 0;
        return;
    }
    const i64 col0 = (i64)(key >> rb), row0 = (i64)(key & ((1ull << rb) - 1ull));
    const i64 pos = csc.nnz > 0 ? csc_find(csc, col0, row0) : -1;
    bool present = (pos >= 0 && mode == ESP_FLUSH_ROUTED);
    double acc = present ? csc.nzval[pos] : 0.0;
    fold_step(present, acc, (u32)(k & ESP_TAG_MASK), sv[i]);
    for (i64 j = i + 1; j < E; j++) {
        const u64 kj = sk[j];
        if ((kj >> ESP_TAG_BITS) != key) break;
        fold_step(present, acc, (u32)(kj & ESP_TAG_MASK), sv[j]);
    }
    if (pos >= 0) {
        if (mode == ESP_FLUSH_ROUTED)
            csc.nzval[pos] = acc;
        else if (present)
            csc.nzval[pos] = csc.nzval[pos] + acc;  // csc operand first, sparsematrixlnk.jl:363
        flag[i] = 0;
    } else {
        flag[i] = present ? 1u : 0u;
        fval[i] = acc;
    }
}

// pos = exclusive scan of flag (E+1 entries).  Emits compacted entries; records for every
// column the number of emitted entries up to the end of its run (colend, zeroed before).
// FRESH: write the final CSC arrays (rowval 1-based); else write (key,val) of new entries.
template <bool FRESH>
static __global__ __launch_bounds__(THREADS) void compact_k(const u64 *__restrict__ sk,
                                                     const double *__restrict__ fval, i64 E,
                                                     const u32 *__restrict__ pos, int rb,
                                                     i64 *__restrict__ out_row,
                                                     u64 *__restrict__ out_key,
                                                     double *__restrict__ out_val,
                                                     u64 *__restrict__ colend) {
    const i64 i = (i64)blockIdx.x * THREADS + threadIdx.x;
    if (i >= E) return;
    const u64 key = sk[i] >> ESP_TAG_BITS;
    const u32 p = pos[i], pn = pos[i + 1];
    if (pn > p) {
        if (FRESH)
            out_row[p] = (i64)(key & ((1ull << rb) - 1ull)) + 1;
        else
            out_key[p] = key;
        out_val[p] = fval[i];
    }
    const u64 col = key >> rb;
    if (i == E - 1 || ((sk[i + 1] >> ESP_TAG_BITS) >> rb) != col) colend[col] = pn;
}

// colptr[c] = base(c) + scanned[c] + 1 ; scanned = exclusive max-scan of colend (n+1 entries)
static __global__ void colptr_finish_k(const u64 *__restrict__ scanned, const i64 *__restrict__ old_colptr,
                                i64 n1, i64 *__restrict__ colptr) {
    const i64 c = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n1) return;
    colptr[c] = (old_colptr ? old_colptr[c] : 1) + (i64)scanned[c];
}

// column index of every stored entry: heads[start(c)] = c for non-empty columns, then an
// exclusive max-scan over Z+1 slots gives colidx[p] at scanned[p+1]
// (columns [c0, c0+n): a shard's window, else all)
static __global__ void col_heads_k(const i64 *__restrict__ colptr, i64 c0, i64 n, u32 *__restrict__ heads) {
    const i64 c = c0 + (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= c0 + n) return;
    const i64 a = colptr[c] - 1, b = colptr[c + 1] - 1;
    if (b > a) heads[a] = (u32)c;
}

// dropzeros!: keep[k] = nzval != 0 (Z+1 slots)
static __global__ void nonzero_flags_k(const double *__restrict__ nzval, i64 Z, u32 *__restrict__ flag) {
    const i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (k > Z) return;
    flag[k] = (k < Z && nzval[k] != 0.0) ? 1u : 0u;
}
static __global__ void dropzeros_compact_k(const i64 *__restrict__ rowval, const double *__restrict__ nzval,
                                    i64 Z, const u32 *__restrict__ pos, i64 *__restrict__ out_row,
                                    double *__restrict__ out_val) {
    const i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= Z) return;
    if (pos[k + 1] > pos[k]) {
        out_row[pos[k]] = rowval[k];
        out_val[pos[k]] = nzval[k];
    }
}
static __global__ void dropzeros_colptr_k(const i64 *__restrict__ old_colptr, i64 n1,
                                   const u32 *__restrict__ pos, i64 *__restrict__ colptr) {
    const i64 c = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n1) return;
    colptr[c] = (i64)pos[old_colptr[c] - 1] + 1;
}

// single-entry lookup (getindex slow path)
static __global__ void getindex_k(Csc csc, i64 row0, i64 col0, double *out /* [value, found] */) {
    const i64 pos = csc.nnz > 0 ? csc_find(csc, col0, row0) : -1;
    out[0] = pos >= 0 ? csc.nzval[pos] : 0.0;
    out[1] = pos >= 0 ? 1.0 : 0.0;
}

// pattern hash partial sums: acc[0] over colptr, acc[1] over rowval (same formula as
// oracle/esparse_oracle.c:orc_csc_pattern_hash; integer adds commute -> deterministic)
static __global__ __launch_bounds__(THREADS) void pattern_hash_k(const i64 *__restrict__ colptr, i64 n1,
                                                          const i64 *__restrict__ rowval, i64 Z,
                                                          unsigned long long *__restrict__ acc) {
    __shared__ u64 red[2][THREADS / 64];
    u64 h1 = 0, h2 = 0;
    const i64 stride = (i64)gridDim.x * THREADS;
    for (i64 j = (i64)blockIdx.x * THREADS + threadIdx.x; j < n1; j += stride)
        h1 += esp_mix64((u64)colptr[j] + 0x9E3779B97F4A7C15ull * (u64)(j + 1));
    for (i64 k = (i64)blockIdx.x * THREADS + threadIdx.x; k < Z; k += stride)
        h2 += esp_mix64((u64)rowval[k] + 0x9E3779B97F4A7C15ull * (u64)(k + 1));
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        h1 += __shfl_down(h1, d, ESP_WAVE);
        h2 += __shfl_down(h2, d, ESP_WAVE);
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) {
        red[0][w] = h1;
        red[1][w] = h2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 a = 0, b = 0;
        for (int i = 0; i < THREADS / 64; i++) {
            a += red[0][i];
            b += red[1][i];
        }
        atomicAdd(&acc[0], (unsigned long long)a);
        atomicAdd(&acc[1], (unsigned long long)b);
    }
}

}  // namespace espfold
