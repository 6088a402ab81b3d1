// local.hpp -- LDS bucket kernel (K2 tail + K3): one workgroup takes one segment of the
// prefix-partitioned entry array (<= CAP entries, all sharing the top key bits), finishes
// the stable sort on the remaining key bits inside LDS, folds duplicates in append order and
// emits the distinct entries compacted at the head of the segment's slot in the scratch pair.
//
// HBM traffic: reads 16 B per appended entry once, writes 16 B per emitted entry once.
// LDS: packed sort keys (8 B) + values (8 B) per slot + wave digit counters; the sort moves
// only the packed key (remaining key bits | slot index | kind), values stay in place and are
// fetched through the slot index by the fold.  Stable LSD radix, 8-bit digits, ranking by
// 64-lane ballot matching (no LDS atomics, deterministic).
#pragma once
#include "common.hpp"
#include "fold.hpp"

namespace esplocal {

constexpr int THREADS = 512;
constexpr int WAVES = THREADS / ESP_WAVE;
constexpr int ITEMS = 8;
constexpr int CAP = THREADS * ITEMS;  // 4096 entries per segment
constexpr int IDX_BITS = 12;
constexpr int SUB_SHIFT = ESP_TAG_BITS + IDX_BITS;  // packed: sub << 14 | idx << 2 | kind
constexpr int MAX_REM_BITS = 64 - SUB_SHIFT;
static_assert((1 << IDX_BITS) == CAP, "slot index must cover the segment capacity");

struct Args {
    const u64 *keys_in;
    const double *vals_in;
    const i64 *seg_start;  // S+1
    int S;
    int rem_bits;  // key bits below the partition prefix (col/row bits, without the kind bits)
    int rb;
    espfold::Csc csc;
    int mode;
    u64 *out_keys;  // (col0<<rb | row0) of emitted entries, at seg_start[s] + q
    double *out_vals;
    u32 *seg_count;  // emitted entries per segment
};

__global__ __launch_bounds__(THREADS, 4) void local_k(Args a) {
    __shared__ u64 skey[CAP];
    __shared__ double sval[CAP];
    __shared__ u32 cnt[WAVES][256];
    __shared__ u32 lw[8];
    __shared__ u32 gcount[WAVES * ITEMS];

    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int s = blockIdx.x;
    const i64 beg = a.seg_start[s];
    const int n = (int)(a.seg_start[s + 1] - beg);
    if (n == 0) {
        if (t == 0) a.seg_count[s] = 0;
        return;
    }
    const u64 submask = a.rem_bits >= 64 ? ~0ull : ((1ull << a.rem_bits) - 1ull);
    const u64 hi = ((a.keys_in[beg] >> ESP_TAG_BITS) >> a.rem_bits) << a.rem_bits;  // shared prefix
    const int wbase = w * (ITEMS * ESP_WAVE) + lane;

    u64 k[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        const int p = wbase + i * ESP_WAVE;
        if (p < n) {
            const u64 key = a.keys_in[beg + p];
            sval[p] = a.vals_in[beg + p];
            k[i] = (((key >> ESP_TAG_BITS) & submask) << SUB_SHIFT) | ((u64)p << ESP_TAG_BITS) | (key & ESP_TAG_MASK);
        } else {
            k[i] = ~0ull;  // sorts behind every real entry (stable: real entries come first on ties)
        }
    }

    const u64 lt = (1ull << lane) - 1ull;
    for (int shift = SUB_SHIFT; shift < SUB_SHIFT + a.rem_bits; shift += 8) {
        const int bits = min(8, SUB_SHIFT + a.rem_bits - shift);
        const u32 dmask = (1u << bits) - 1u;
        // zero the wave counters
        for (int q = t; q < WAVES * 256; q += THREADS) (&cnt[0][0])[q] = 0;
        __syncthreads();
        unsigned short rank[ITEMS];
#pragma unroll
        for (int i = 0; i < ITEMS; i++) {
            const u32 d = (u32)(k[i] >> shift) & dmask;
            u64 m = ~0ull;
#pragma unroll
            for (int b = 0; b < 8; b++) {
                const bool bit = (d >> b) & 1u;
                const u64 bb = __ballot(bit);
                m &= bit ? bb : ~bb;
            }
            const u32 prev = cnt[w][d];
            rank[i] = (unsigned short)(prev + (u32)__popcll(m & lt));
            __builtin_amdgcn_wave_barrier();
            if ((m & lt) == 0) cnt[w][d] = prev + (u32)__popcll(m);
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
        // digit totals -> block-exclusive digit starts (+ per-wave bases) written back to cnt
        u32 c[WAVES];
        u32 tot = 0, inc = 0;
        if (t < 256) {
#pragma unroll
            for (int i = 0; i < WAVES; i++) {
                const u32 x = cnt[i][t];
                c[i] = tot;
                tot += x;
            }
            inc = tot;
#pragma unroll
            for (int dlt = 1; dlt < ESP_WAVE; dlt <<= 1) {
                const u32 o = __shfl_up(inc, dlt, ESP_WAVE);
                if (lane >= dlt) inc += o;
            }
            if (lane == 63) lw[w] = inc;
        }
        __syncthreads();
        if (t < 256) {
            u32 base = inc - tot;
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (i < w) base += lw[i];
#pragma unroll
            for (int i = 0; i < WAVES; i++) cnt[i][t] = base + c[i];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < ITEMS; i++) {
            const u32 d = (u32)(k[i] >> shift) & dmask;
            skey[cnt[w][d] + rank[i]] = k[i];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < ITEMS; i++) k[i] = skey[wbase + i * ESP_WAVE];
    }
    if (a.rem_bits <= 0) {
#pragma unroll
        for (int i = 0; i < ITEMS; i++) skey[wbase + i * ESP_WAVE] = k[i];
    }
    __syncthreads();

    // ---- ordered fold: run heads walk their run in LDS (append order inside a run)
    bool emit[ITEMS];
    double acc[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        const int q = wbase + i * ESP_WAVE;
        emit[i] = false;
        acc[i] = 0.0;
        if (q < n) {
            const u64 sub = k[i] >> SUB_SHIFT;
            const bool head = q == 0 || (skey[q - 1] >> SUB_SHIFT) != sub;
            if (head) {
                const u64 full = hi | sub;
                i64 pos = -1;
                if (a.csc.nnz > 0) pos = espfold::csc_find(a.csc, (i64)(full >> a.rb), (i64)(full & ((1ull << a.rb) - 1ull)));
                bool present = (pos >= 0 && a.mode == ESP_FLUSH_ROUTED);
                double x = present ? a.csc.nzval[pos] : 0.0;
                for (int j = q; j < n; j++) {
                    const u64 kj = skey[j];
                    if ((kj >> SUB_SHIFT) != sub) break;
                    espfold::fold_step(present, x, (u32)(kj & ESP_TAG_MASK), sval[(kj >> ESP_TAG_BITS) & (CAP - 1)]);
                }
                if (pos >= 0) {
                    if (a.mode == ESP_FLUSH_ROUTED)
                        a.csc.nzval[pos] = x;
                    else if (present)
                        a.csc.nzval[pos] = a.csc.nzval[pos] + x;
                } else if (present) {
                    emit[i] = true;
                    acc[i] = x;
                }
            }
        }
        const u64 bal = __ballot(emit[i]);
        if (lane == 0) gcount[w * ITEMS + i] = (u32)__popcll(bal);
    }
    __syncthreads();
    // exclusive scan over the WAVES*ITEMS (=64) group counts by wave 0
    if (w == 0) {
        const u32 c = gcount[lane];
        u32 inc = c;
#pragma unroll
        for (int dlt = 1; dlt < ESP_WAVE; dlt <<= 1) {
            const u32 o = __shfl_up(inc, dlt, ESP_WAVE);
            if (lane >= dlt) inc += o;
        }
        gcount[lane] = inc - c;
        if (lane == 63) a.seg_count[s] = inc;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        const u64 bal = __ballot(emit[i]);
        if (emit[i]) {
            const u32 e = gcount[w * ITEMS + i] + (u32)__popcll(bal & lt);
            a.out_keys[beg + e] = hi | (k[i] >> SUB_SHIFT);
            a.out_vals[beg + e] = acc[i];
        }
    }
}

// copy every segment's emitted run to its final place and mark column ends.
// FRESH: final CSC arrays (rowval 1-based); else compact (key,val) list of new entries.
template <bool FRESH>
__global__ __launch_bounds__(256) void gather_k(const u64 *__restrict__ tkeys, const double *__restrict__ tvals,
                                                const i64 *__restrict__ seg_start,
                                                const u64 *__restrict__ seg_out /* exclusive scan of seg_count, S+1 */,
                                                int rb, int col_aligned, i64 *__restrict__ out_row,
                                                u64 *__restrict__ out_key, double *__restrict__ out_val,
                                                u64 *__restrict__ colend) {
    const int s = blockIdx.x;
    const i64 src = seg_start[s];
    const i64 dst = (i64)seg_out[s];
    const int cnt = (int)(seg_out[s + 1] - seg_out[s]);
    const u64 rowmask = (1ull << rb) - 1ull;
    for (int q = threadIdx.x; q < cnt; q += 256) {
        const u64 key = tkeys[src + q];
        if (FRESH)
            out_row[dst + q] = (i64)(key & rowmask) + 1;
        else
            out_key[dst + q] = key;
        out_val[dst + q] = tvals[src + q];
        const u64 col = key >> rb;
        // last emitted entry of its column inside this segment; a later segment of the same
        // column (prefix finer than a column) overwrites with a larger value: use max
        if (q == cnt - 1 || (tkeys[src + q + 1] >> rb) != col) {
            if (col_aligned)
                colend[col] = (u64)(dst + q + 1);
            else
                atomicMax((unsigned long long *)&colend[col], (unsigned long long)(dst + q + 1));
        }
    }
}

__global__ void widen_counts_k(const u32 *__restrict__ in, i64 n, u64 *__restrict__ out) {
    const i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (g > n) return;
    out[g] = g < n ? (u64)in[g] : 0ull;
}

}  // namespace esplocal
