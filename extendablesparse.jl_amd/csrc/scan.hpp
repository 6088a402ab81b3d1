// scan.hpp -- device-wide exclusive scan (add or max) used for partition offsets,
// compaction positions and colptr.  Three-phase (reduce / scan partials / apply):
// no inter-workgroup communication inside a launch, so no visibility protocol is
// needed (MI355X L2s are per-XCD).  HBM-bound: reads n, writes n (+ n/2048 partials).
#pragma once
#include "common.hpp"

namespace espscan {

constexpr int THREADS = 256;
constexpr int ITEMS = 8;
constexpr int CHUNK = THREADS * ITEMS;

template <typename T, bool MAX>
__device__ __forceinline__ T comb(T a, T b) {
    if (MAX) return a > b ? a : b;
    return a + b;
}

template <typename T, bool MAX>
__device__ __forceinline__ T wave_inclusive(T v, int lane) {
#pragma unroll
    for (int d = 1; d < ESP_WAVE; d <<= 1) {
        T o = __shfl_up(v, d, ESP_WAVE);
        if (lane >= d) v = comb<T, MAX>(o, v);
    }
    return v;
}

// block-wide exclusive scan of one value per thread; returns exclusive prefix, total in *total
template <typename T, bool MAX>
__device__ __forceinline__ T block_exclusive(T v, T *lds_wave /*THREADS/64*/, T *total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    T inc = wave_inclusive<T, MAX>(v, lane);
    if (lane == 63) lds_wave[w] = inc;
    __syncthreads();
    T carry = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < THREADS / 64; i++) {
        T x = lds_wave[i];
        if (i < w) carry = comb<T, MAX>(carry, x);
        tot = comb<T, MAX>(tot, x);
    }
    __syncthreads();
    T exc = __shfl_up(inc, 1, ESP_WAVE);
    if (lane == 0) exc = 0;
    *total = tot;
    return comb<T, MAX>(carry, exc);
}

template <typename T, bool MAX>
static __global__ __launch_bounds__(THREADS) void reduce_k(const T *__restrict__ in, i64 n,
                                                    T *__restrict__ partial) {
    __shared__ T lw[THREADS / 64];
    const i64 base = (i64)blockIdx.x * CHUNK;
    T acc = 0;
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
        i64 idx = base + k * THREADS + threadIdx.x;
        if (idx < n) acc = comb<T, MAX>(acc, in[idx]);
    }
    T tot;
    block_exclusive<T, MAX>(acc, lw, &tot);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// addend: added to every output (e.g. the 1 of a 1-based colptr), outside the scan operation
template <typename T, bool MAX>
static __global__ __launch_bounds__(THREADS) void apply_k(const T *in, T *out, i64 n,
                                                   const T *__restrict__ carry_in, T addend) {
    __shared__ T tile[CHUNK + ITEMS];
    __shared__ T lw[THREADS / 64];
    const i64 base = (i64)blockIdx.x * CHUNK;
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
        i64 idx = base + k * THREADS + threadIdx.x;
        tile[k * THREADS + threadIdx.x] = idx < n ? in[idx] : (T)0;
    }
    __syncthreads();
    T v[ITEMS];
    T run = 0;
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
        T x = tile[threadIdx.x * ITEMS + k];
        v[k] = run;
        run = comb<T, MAX>(run, x);
    }
    T tot;
    T pre = block_exclusive<T, MAX>(run, lw, &tot);
    if (carry_in) pre = comb<T, MAX>(pre, carry_in[blockIdx.x]);
#pragma unroll
    for (int k = 0; k < ITEMS; k++) tile[threadIdx.x * ITEMS + k] = comb<T, MAX>(pre, v[k]);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
        i64 idx = base + k * THREADS + threadIdx.x;
        if (idx < n) out[idx] = tile[k * THREADS + threadIdx.x] + addend;
    }
}

static inline i64 workspace_elems(i64 n) {
    i64 tot = 0;
    while (n > CHUNK) {
        n = ceil_div<i64>(n, CHUNK);
        tot += n;
    }
    return tot + 8;
}

// exclusive scan, in may equal out; ws holds workspace_elems(n) elements of T.
// returns the number of kernel launches
template <typename T, bool MAX>
static int exclusive(hipStream_t s, const T *in, T *out, i64 n, T *ws, T addend = 0) {
    if (n <= 0) return 0;
    i64 nb = ceil_div<i64>(n, CHUNK);
    if (nb == 1) {
        hipLaunchKernelGGL((apply_k<T, MAX>), dim3(1), dim3(THREADS), 0, s, in, out, n, (const T *)nullptr, addend);
        return 1;
    }
    hipLaunchKernelGGL((reduce_k<T, MAX>), dim3((unsigned)nb), dim3(THREADS), 0, s, in, n, ws);
    int l = 1 + exclusive<T, MAX>(s, ws, ws, nb, ws + nb);
    hipLaunchKernelGGL((apply_k<T, MAX>), dim3((unsigned)nb), dim3(THREADS), 0, s, in, out, n, (const T *)ws, addend);
    return l + 1;
}

}  // namespace espscan
