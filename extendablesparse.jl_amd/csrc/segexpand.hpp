// segexpand.hpp -- the LAST bits of an item partition done inside the expansion (round 4).
//
// The item partitions (femitems.hpp, elements.hpp) bring 8-byte item records down to the bucket kernel's segments with
// the flush's stable passes, then expand every item into its W updates.  The last pass often resolves only two or three
// bits (3-D P1 FEM at 10^7 DoF: 19 prefix bits = 8 + 8 + 3) and still costs a histogram and a scatter over all records.
// Here the passes stop while a segment still holds up to LCAP = 4096 items; ONE workgroup per segment then orders its
// records by the next `lbits` (1..3) bits itself -- a stable counting sort of at most 4096 records over at most 8 bins,
// ballots per 64-record chunk, a 512-counter scan, a 2-byte order array in LDS (the records themselves are read again
// through it: they lie in the caches) -- expands the items in that order, and writes the starts of its 2^lbits
// sub-segments (what the bucket kernel takes as its segment table) and their longest length.  One pass over the
// records (8 B read + 8 B written per item, its histogram 8 B) and ~25 small launches less.
#pragma once
#include "common.hpp"
#include "scan.hpp"

namespace espseg {

constexpr int THREADS = 256;  // (= espscan::THREADS: block_exclusive)
constexpr int LCAP = 4096;    // items per segment
constexpr int CHUNKS = LCAP / ESP_WAVE;
constexpr int MAXB = 3;

struct SegArgs {
    const u64 *recs;        // item records, partitioned down to S segments
    const i64 *seg_start;   // S + 1
    int S;
    int W;                  // updates per item
    int lbits;              // bits resolved here (1..MAXB)
    int lshift;             // the local digit = ((rec >> 2) - base) >> lshift, masked
    u64 base;               // key window base of the records' (virtual) layout
    i64 *sub_start;         // (S << lbits) + 1 entry offsets, in UPDATES (x W)
    unsigned long long *maxsub;  // atomicMax: longest sub-segment, in updates
    i64 total_items;
};

// LDS of the ordering step
struct SegLds {
    unsigned short ord[LCAP];
    u32 cnt[(1 << MAXB) * CHUNKS];  // bin-major: [bin][chunk]
    u32 lw[THREADS / ESP_WAVE];
    u32 binstart[(1 << MAXB) + 1];
};

// Orders the records of segment s (uniform per workgroup) by their local digit, stably; afterwards L.ord[q] = index (inside
// the segment) of the record at sorted position q, L.binstart[d] = first sorted position of digit d.  Returns n.
__device__ __forceinline__ int segment_order(const SegArgs &a, int s, SegLds &L, i64 *beg_out) {
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const i64 beg = esp_uniform_i64(a.seg_start[s]);
    const int n = (int)min((i64)LCAP, esp_uniform_i64(a.seg_start[s + 1]) - beg);
    *beg_out = beg;
    const int NB = 1 << a.lbits;
    const u32 mask = (u32)NB - 1u;
    for (int q = t; q < NB * CHUNKS; q += THREADS) L.cnt[q] = 0;
    __syncthreads();
    constexpr int ITER = LCAP / THREADS;  // 16 records per thread at most
    unsigned char dig[ITER];
    const u64 lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int k = 0; k < ITER; k++) {
        const int i = k * THREADS + t;  // chunk c = i / 64 = k * 4 + w
        const bool valid = i < n;
        u32 d = 0;
        if (valid) d = (u32)((((a.recs[beg + i] >> ESP_TAG_BITS) - a.base) >> a.lshift)) & mask;
        dig[k] = (unsigned char)d;
        if (k * THREADS < n) {  // (uniform)
            const int c = k * (THREADS / ESP_WAVE) + w;
            for (int b = 0; b < NB; b++) {
                const u64 m = __ballot(valid && d == (u32)b);
                if (lane == 0 && m) L.cnt[b * CHUNKS + c] = (u32)__popcll(m);
            }
        }
    }
    __syncthreads();
    // exclusive scan over (bin, chunk): two counters per thread
    {
        const int i0 = 2 * t, i1 = 2 * t + 1;
        const u32 c0 = i0 < NB * CHUNKS ? L.cnt[i0] : 0u, c1 = i1 < NB * CHUNKS ? L.cnt[i1] : 0u;
        u32 tot;
        const u32 ex = espscan::block_exclusive<u32, false>(c0 + c1, L.lw, &tot);
        if (i0 < NB * CHUNKS) L.cnt[i0] = ex;
        if (i1 < NB * CHUNKS) L.cnt[i1] = ex + c0;
    }
    __syncthreads();
    if (t <= NB) L.binstart[t] = t < NB ? L.cnt[t * CHUNKS] : (u32)n;
#pragma unroll
    for (int k = 0; k < ITER; k++) {
        const int i = k * THREADS + t;
        if (k * THREADS < n) {  // (uniform)
            const bool valid = i < n;
            const u32 d = dig[k];
            const int c = k * (THREADS / ESP_WAVE) + w;
            u64 m = 0;
            for (int b = 0; b < NB; b++) {
                const u64 mb = __ballot(valid && d == (u32)b);
                m = d == (u32)b ? mb : m;
            }
            if (valid) L.ord[L.cnt[d * CHUNKS + c] + (u32)__popcll(m & lt)] = (unsigned short)i;
        }
    }
    __syncthreads();
    // the segment table of the bucket kernel and its longest segment
    if (t < NB) {
        const i64 lo = beg + (i64)L.binstart[t], hi = beg + (i64)L.binstart[t + 1];
        a.sub_start[((i64)s << a.lbits) + t] = lo * a.W;
        if (hi > lo) atomicMax(a.maxsub, (unsigned long long)((hi - lo) * a.W));
    }
    if (s == a.S - 1 && t == 0) a.sub_start[(i64)a.S << a.lbits] = a.total_items * a.W;
    return n;
}

// `cnt` staged updates of ONE WAVE (keys KT, values) to the output: 16-byte stores where the position allows
template <typename KT>
__device__ __forceinline__ void copy_out_wave(const KT *lk, const double *lv, int cnt, KT *gk, double *gv) {
    const int lane = threadIdx.x & 63;
    typedef double dbl2 __attribute__((ext_vector_type(2)));
    {
        const int head = min(cnt, (int)(((uintptr_t)gv >> 3) & 1));
        if (lane < head) gv[lane] = lv[lane];
        const int pairs = (cnt - head) >> 1;
        dbl2 *g2 = reinterpret_cast<dbl2 *>(gv + head);
        for (int q = lane; q < pairs; q += ESP_WAVE) g2[q] = dbl2{lv[head + 2 * q], lv[head + 2 * q + 1]};
        if (lane == 0 && ((cnt - head) & 1)) gv[cnt - 1] = lv[cnt - 1];
    }
    if constexpr (sizeof(KT) == 4) {
        typedef u32 u32x4 __attribute__((ext_vector_type(4)));
        const int head = min(cnt, (int)((4 - (((uintptr_t)gk >> 2) & 3)) & 3));
        if (lane < head) gk[lane] = lk[lane];
        const int quads = (cnt - head) >> 2;
        u32x4 *g4 = reinterpret_cast<u32x4 *>(gk + head);
        for (int q = lane; q < quads; q += ESP_WAVE) g4[q] = u32x4{lk[head + 4 * q], lk[head + 4 * q + 1], lk[head + 4 * q + 2], lk[head + 4 * q + 3]};
        for (int q = head + 4 * quads + lane; q < cnt; q += ESP_WAVE) gk[q] = lk[q];
    } else {
        typedef unsigned long long ull2 __attribute__((ext_vector_type(2)));
        const int head = min(cnt, (int)(((uintptr_t)gk >> 3) & 1));
        if (lane < head) gk[lane] = lk[lane];
        const int pairs = (cnt - head) >> 1;
        ull2 *g2 = reinterpret_cast<ull2 *>(gk + head);
        for (int q = lane; q < pairs; q += ESP_WAVE) g2[q] = ull2{lk[head + 2 * q], lk[head + 2 * q + 1]};
        if (lane == 0 && ((cnt - head) & 1)) gk[cnt - 1] = lk[cnt - 1];
    }
}

// what one lane wrote to LDS is visible to the other lanes of its wave (no workgroup barrier: the wave runs in lock step)
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The expansion of segment blockIdx.x in the order segment_order gives.  Every WAVE takes rounds of 64 items by itself --
// emit(rec, lk + lane W, lv + lane W) forms the W updates of one item in the wave's own staging area, the round leaves as
// whole lines -- so that the waves of a workgroup do not wait for one another (with workgroup-wide rounds and two barriers
// per round the kernel took 40 % longer than the unordered expansion).  lk / lv: THREADS * W staged updates.
template <typename KT, typename Emit>
__device__ __forceinline__ void segment_expand(const SegArgs &a, SegLds &L, KT *lk, double *lv, KT *keys_out, double *vals_out, Emit emit) {
    const int s = blockIdx.x;
    i64 beg;
    const int n = segment_order(a, s, L, &beg);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, W = a.W;
    KT *wk = lk + w * (ESP_WAVE * W);
    double *wv = lv + w * (ESP_WAVE * W);
    for (int q0 = w * ESP_WAVE; q0 < n; q0 += THREADS) {
        const int q = q0 + lane;
        if (q < n) emit(a.recs[beg + (i64)L.ord[q]], wk + lane * W, wv + lane * W);
        wave_lds_sync();
        const int cnt = min(ESP_WAVE, n - q0) * W;
        const i64 e0 = (beg + q0) * (i64)W;
        copy_out_wave<KT>(wk, wv, cnt, keys_out + e0, vals_out + e0);
        wave_lds_sync();
    }
}

}  // namespace espseg
