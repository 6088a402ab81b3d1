// radix.hpp -- stable radix partition pass on packed (col,row) keys (K2).
//
// One pass splits every segment of the entry array by one digit of the key, keeping
// the append order inside a digit (stable), which the ordered fold needs
// (SET/ADD order, sparsematrixlnk.jl:192-195,219-221).  A pass is
//   tile_hist_k  : per-tile digit counts            (reads 8 B/entry)
//   scan         : exclusive scan in (segment, digit, tile) order
//   scatter_k    : wave-match ranking, LDS reorder, coalesced run stores
//                                                    (reads 16 B, writes 16 B/entry)
// HBM-bound.  Tiles are 4096 entries (256 threads x 16): 64 KiB of LDS for the
// reorder buffer, two workgroups per CU.  Wave64 throughout: ballots are 64-bit.
#pragma once
#include "common.hpp"
#include "scan.hpp"

namespace espradix {

constexpr int THREADS = 256;
constexpr int WAVES = THREADS / ESP_WAVE;
constexpr int ITEMS = 16;
constexpr int TILE = THREADS * ITEMS;
constexpr int RADIX = 512;     // digits of up to 9 bits: thread t owns the digits 2t and 2t + 1 (t and t + THREADS in tile_hist_k)
constexpr int MAX_BITS = 9;

constexpr int XCDS = 8;  // L2s of an MI355X (scatter_k's tile order)
// grid of scatter_k for `tiles` tiles: a multiple of XCDS (workgroups whose tile does not exist leave at once)
static inline unsigned scatter_grid(long long tiles) { return (unsigned)(((tiles + XCDS - 1) / XCDS) * XCDS); }
// XCD-aware tile order: workgroups go round-robin to the 8 XCDs, each with an L2 of its own.  Workgroup i takes tile
// (i mod 8) * (grid / 8) + i / 8 (the grid is a multiple of 8, scatter_grid): an XCD works through a contiguous range of
// tiles, so what consecutive tiles write side by side -- the 128-byte pieces they append to a digit's output range, their
// 8-byte counts in the digit-major histogram -- meets in ONE L2 and leaves it as full lines (with tile = i the
// neighbours of a piece were written through seven other L2s: 3-D FEM 31 -> 22 ms for the scatter kernels)
__device__ __forceinline__ long long xcd_tile() {
    const long long per_xcd = (long long)gridDim.x / XCDS;
    return (long long)(blockIdx.x % XCDS) * per_xcd + (long long)(blockIdx.x / XCDS);
}
struct Pass {
    const u64 *keys_in;
    const double *vals_in;
    u64 *keys_out;
    double *vals_out;
    const i64 *seg_start;   // S+1 entry offsets (device)
    const i64 *tile_first;  // S+1 first tile of each segment (device)
    int S;
    int shift;  // digit = (((key >> 2) - base) >> shift) & ((1<<bits)-1)
    u64 base;   // key window: every (col,row) key lies in [base, base+span)
    u64 span;
    u32 *err;   // set when a key falls outside the window (digit clamped, no stray access)
    int bits;   // 1..9 (MAX_BITS)
    // owner mode (column-range shards): digit = floor(col0 * owner_P / owner_n), col0 = key >> colshift
    int owner_P;
    i64 owner_n;
    int colshift;
    u64 *hist;  // [tile_first[s]*R + d*ntiles_s + tile_in_seg]; scanned in place
    int keys_only = 0;  // the records are their keys (single-word item records, femitems.hpp): no value array is read or written
    // RAW: the FIRST pass over a caller's triplets, run while they are appended (append_first_pass, partition.hip): the keys are
    // formed from rows / cols on the fly -- no packed stream is written in stream order and read again; vals_in = the caller's values
    const i64 *raw_rows = nullptr, *raw_cols = nullptr;
    i64 raw_m = 0, raw_n = 0;
    int raw_rb = 0, raw_kind = 0, raw_negate = 0;
    unsigned long long *raw_err = nullptr;  // atomicMin: first entry (1-based) with an index outside the matrix
    // entries of ONE known kind: from the first pass that leaves at most 32 key bits below its prefix, the passes write those bits
    // as 4-byte keys (k32_out: keys_out is a u32 array) and read them as such (k32_in: keys_in is one; digit = key32 >> shift) --
    // 12 instead of 16 bytes per entry out of a pass, into the next one and into the bucket kernel
    int k32_out = 0;
    int k32_in = 0;
    int k32_rem = 0;  // key bits below the prefix after this pass
};

// the 4-byte key a pass writes for the entry whose key (packed, or 4-byte from the pass before) it holds as kk
template <bool K32IN>
__device__ __forceinline__ u32 narrow_key(const Pass &p, u64 kk) {
    const u64 low = p.k32_rem >= 32 ? 0xffffffffull : ((1ull << p.k32_rem) - 1ull);
    if constexpr (K32IN) return (u32)(kk & low);
    u64 kn = (kk >> ESP_TAG_BITS) - p.base;
    kn = kn < p.span ? kn : p.span - 1;  // (as digit_of: the histogram kernel has reported it)
    return (u32)(kn & low);
}

// CHECK: report keys outside the window (the histogram kernel sees every key of a pass with the
// same parameters, so the scatter kernel only clamps)
template <bool CHECK, bool K32IN = false>
__device__ __forceinline__ u32 digit_of(const Pass &p, u64 key, u32 mask) {
    if constexpr (K32IN) return (u32)(key >> p.shift) & mask;  // (a 4-byte key: the bits below the prefix, checked by an earlier pass)
    if (p.owner_P) return (u32)(((key >> p.colshift) * (u64)p.owner_P) / (u64)p.owner_n);
    u64 kn = (key >> ESP_TAG_BITS) - p.base;
    if (CHECK && kn >= p.span) *p.err = 1u;
    kn = kn < p.span ? kn : p.span - 1;
    return (u32)(kn >> p.shift) & mask;
}

// largest s with tile_first[s] <= tile, or -1 when tile is past the last segment
__device__ __forceinline__ int find_segment(const i64 *__restrict__ tile_first, int S, i64 tile) {
    if (tile >= tile_first[S]) return -1;
    int lo = 0, hi = S;  // invariant tile_first[lo] <= tile < tile_first[hi]
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (tile_first[mid] <= tile)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}

// RAW first pass (one segment: the whole batch): the digit of an entry follows from its column alone (the pass's shift does not
// reach into the row bits); columns outside the matrix are reported and counted under digit 0 (the host stops before anything
// is appended)
static __global__ __launch_bounds__(THREADS) void tile_hist_raw_k(Pass p) {
    __shared__ u32 cnt[RADIX];
    const int t = threadIdx.x;
    const i64 tile = xcd_tile();
    if (tile >= p.tile_first[1]) return;
    const i64 nts = p.tile_first[1];
    const i64 beg = tile * TILE;
    const i64 end = min(p.seg_start[1], beg + (i64)TILE);
    cnt[t] = 0;
    cnt[t + THREADS] = 0;
    __syncthreads();
    const u32 mask = (1u << p.bits) - 1u;
    // 16-byte loads (two columns per lane: counting does not care about order; a tile starts at a multiple of TILE entries, so
    // only the caller's base pointer decides the alignment), all of a thread's loads in flight before the first count
    auto count_one = [&](i64 col, i64 idx) {
        if (col < 1 || col > p.raw_n) {
            atomicMin(p.raw_err, (unsigned long long)idx + 1ull);
            col = 1;
        }
        atomicAdd(&cnt[digit_of<true>(p, ((u64)(col - 1) << p.raw_rb) << ESP_TAG_BITS, mask)], 1u);
    };
    if ((reinterpret_cast<uintptr_t>(p.raw_cols) & 15) == 0) {
        typedef long long ll2 __attribute__((ext_vector_type(2)));
        const ll2 *pc = reinterpret_cast<const ll2 *>(p.raw_cols + beg);
        const i64 npair = (end - beg) >> 1;
        ll2 c2[ITEMS / 2];
#pragma unroll
        for (int k = 0; k < ITEMS / 2; k++) {
            const i64 q = (i64)k * THREADS + t;
            c2[k] = q < npair ? pc[q] : ll2{1, 1};
        }
#pragma unroll
        for (int k = 0; k < ITEMS / 2; k++) {
            const i64 q = (i64)k * THREADS + t;
            if (q < npair) {
                count_one(c2[k].x, beg + 2 * q);
                count_one(c2[k].y, beg + 2 * q + 1);
            }
        }
        if (t == 0 && ((end - beg) & 1)) count_one(p.raw_cols[end - 1], end - 1);
    } else {
        i64 c[ITEMS];
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
            const i64 idx = beg + (i64)k * THREADS + t;
            c[k] = idx < end ? p.raw_cols[idx] : 1;
        }
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
            const i64 idx = beg + (i64)k * THREADS + t;
            if (idx < end) count_one(c[k], idx);
        }
    }
    __syncthreads();
    const int R = 1 << p.bits;
    if (t < R && cnt[t] != 0) p.hist[(i64)t * nts + tile] = cnt[t];
    if (t + THREADS < R && cnt[t + THREADS] != 0) p.hist[(i64)(t + THREADS) * nts + tile] = cnt[t + THREADS];
}

static __global__ __launch_bounds__(THREADS) void tile_hist_k(Pass p) {
    __shared__ u32 cnt[RADIX];
    const int t = threadIdx.x;
    const i64 tile = xcd_tile();
    const int s = p.S == 1 ? (tile < p.tile_first[1] ? 0 : -1) : find_segment(p.tile_first, p.S, tile);
    if (s < 0) return;
    const i64 tf = p.tile_first[s];
    const i64 nts = p.tile_first[s + 1] - tf;
    const i64 tin = tile - tf;
    const i64 beg = p.seg_start[s] + tin * TILE;
    const i64 end = min(p.seg_start[s + 1], beg + (i64)TILE);
    cnt[t] = 0;
    cnt[t + THREADS] = 0;
    __syncthreads();
    const u32 mask = (1u << p.bits) - 1u;
    // 16-byte loads (two keys per lane; counting does not care about order), all loads of a
    // thread in flight before the first count; a wave whose keys share one digit (the common
    // case on pre-sorted streams) issues ONE LDS add per load instead of 64 conflicting ones
    const i64 a0 = (beg + 1) & ~(i64)1;           // first 16-byte aligned key of the tile
    const i64 a1 = end & ~(i64)1;
    const i64 npair = a1 > a0 ? (a1 - a0) >> 1 : 0;
    typedef unsigned long long ull2 __attribute__((ext_vector_type(2)));
    const ull2 *pk = reinterpret_cast<const ull2 *>(p.keys_in + a0);
    ull2 kk[ITEMS / 2];
#pragma unroll
    for (int k = 0; k < ITEMS / 2; k++) {
        const i64 q = (i64)k * THREADS + t;
        kk[k] = q < npair ? pk[q] : ull2{~0ull, ~0ull};
    }
    const int lane = t & 63;
#pragma unroll
    for (int k = 0; k < ITEMS / 2; k++) {
        const bool valid = ((i64)k * THREADS + t) < npair;
        const u32 dA = valid ? digit_of<true>(p, kk[k].x, mask) : 0u, dB = valid ? digit_of<true>(p, kk[k].y, mask) : 0u;
        const u64 vm = __ballot(valid);
        if (vm) {
            const int fl = __builtin_ctzll(vm);
            const u32 d0 = (u32)__shfl((int)dA, fl, ESP_WAVE);
            const u64 same = __ballot(valid && dA == d0 && dB == d0);
            if (same == vm) {
                if (lane == fl) atomicAdd(&cnt[d0], 2u * (u32)__popcll(vm));
            } else if (valid) {
                atomicAdd(&cnt[dA], 1u);
                atomicAdd(&cnt[dB], 1u);
            }
        }
    }
    if (t == 0) {  // unaligned head / tail keys
        if (a0 > beg && beg < end) atomicAdd(&cnt[digit_of<true>(p, p.keys_in[beg], mask)], 1u);
        if (a1 < end && a1 >= a0) atomicAdd(&cnt[digit_of<true>(p, p.keys_in[a1], mask)], 1u);
    }
    __syncthreads();
    const int R = 1 << p.bits;
    // the histogram array is zeroed before the launch: only non-zero counts are stored (the
    // digit-major layout makes every store its own cache line)
    if (t < R && cnt[t] != 0) p.hist[tf * R + (i64)t * nts + tin] = cnt[t];
    if (t + THREADS < R && cnt[t + THREADS] != 0) p.hist[tf * R + (i64)(t + THREADS) * nts + tin] = cnt[t + THREADS];
}

// the same over 4-byte keys (Pass::k32_in): 8-byte loads of two keys
static __global__ __launch_bounds__(THREADS) void tile_hist32_k(Pass p) {
    __shared__ u32 cnt[RADIX];
    const int t = threadIdx.x;
    const i64 tile = xcd_tile();
    const int s = p.S == 1 ? (tile < p.tile_first[1] ? 0 : -1) : find_segment(p.tile_first, p.S, tile);
    if (s < 0) return;
    const i64 tf = p.tile_first[s];
    const i64 nts = p.tile_first[s + 1] - tf;
    const i64 tin = tile - tf;
    const i64 beg = p.seg_start[s] + tin * TILE;
    const i64 end = min(p.seg_start[s + 1], beg + (i64)TILE);
    cnt[t] = 0;
    cnt[t + THREADS] = 0;
    __syncthreads();
    const u32 mask = (1u << p.bits) - 1u;
    const u32 *k32 = reinterpret_cast<const u32 *>(p.keys_in);
    const i64 a0 = (beg + 1) & ~(i64)1;  // first 8-byte aligned key of the tile
    const i64 a1 = end & ~(i64)1;
    const i64 npair = a1 > a0 ? (a1 - a0) >> 1 : 0;
    const uint2 *pk = reinterpret_cast<const uint2 *>(k32 + a0);
    uint2 kk[ITEMS / 2];
#pragma unroll
    for (int k = 0; k < ITEMS / 2; k++) {
        const i64 q = (i64)k * THREADS + t;
        kk[k] = q < npair ? pk[q] : uint2{0u, 0u};
    }
    const int lane = t & 63;
#pragma unroll
    for (int k = 0; k < ITEMS / 2; k++) {
        const bool valid = ((i64)k * THREADS + t) < npair;
        const u32 dA = valid ? (kk[k].x >> p.shift) & mask : 0u, dB = valid ? (kk[k].y >> p.shift) & mask : 0u;
        const u64 vm = __ballot(valid);
        if (vm) {
            const int fl = __builtin_ctzll(vm);
            const u32 d0 = (u32)__shfl((int)dA, fl, ESP_WAVE);
            const u64 same = __ballot(valid && dA == d0 && dB == d0);
            if (same == vm) {
                if (lane == fl) atomicAdd(&cnt[d0], 2u * (u32)__popcll(vm));
            } else if (valid) {
                atomicAdd(&cnt[dA], 1u);
                atomicAdd(&cnt[dB], 1u);
            }
        }
    }
    if (t == 0) {  // unaligned head / tail keys
        if (a0 > beg && beg < end) atomicAdd(&cnt[(k32[beg] >> p.shift) & mask], 1u);
        if (a1 < end && a1 >= a0) atomicAdd(&cnt[(k32[a1] >> p.shift) & mask], 1u);
    }
    __syncthreads();
    const int R = 1 << p.bits;
    if (t < R && cnt[t] != 0) p.hist[tf * R + (i64)t * nts + tin] = cnt[t];
    if (t + THREADS < R && cnt[t + THREADS] != 0) p.hist[tf * R + (i64)(t + THREADS) * nts + tin] = cnt[t + THREADS];
}

// The tile is reordered through ONE 32 KiB LDS buffer, keys first, values second: 43 KiB of LDS and 144 VGPRs = three
// workgroups per CU (keys and values staged side by side: 73 KiB = two).  Worth 5 % on the shuffled FEM streams (2.6 ->
// 2.5 ms per pass over 2.4 10^8 entries): the pass is bound by its 128-byte write runs -- a tile of 4096 shuffled
// entries holds 16 per digit -- more than by the bytes it has in flight.
// NINE: digits of 9 bits (two per thread, the digit of an output slot recomputed from its key); else at most 8 bits (one
// digit per thread, the slot's digit kept in an LDS byte): the 8-bit passes of 3-D FEM lost 5 % in the general form.
// NOVAL: the records are 8-byte keys by themselves (Pass::keys_only): half the traffic of a pass
// RAW: keys and values from the caller's triplets (Pass::raw_*), see tile_hist_raw_k
// K32: 0 packed keys in and out; 1 packed keys in, 4-byte keys out (Pass::k32_out); 2 4-byte keys in and out (Pass::k32_in) --
// compile-time: the same choice at run time cost the keys-only passes 60 % (3-D FEM 3.2 -> 5.4 ms)
template <bool NINE, bool NOVAL = false, bool RAW = false, int K32 = 0>
static __global__ __launch_bounds__(THREADS, (NOVAL && !NINE) ? 4 : 3) void scatter_k(Pass p) {
    static_assert(!RAW || !NOVAL, "a raw pass moves values");
    // SLOTDIG: the digit of an output slot is recomputed from its key (NINE; NOVAL: without the byte per slot the kernel's LDS is
    // 39 KiB and FOUR workgroups share a CU); else it is kept in an LDS byte
    constexpr bool SLOTDIG = NINE || NOVAL;
    static_assert(K32 == 0 || (!RAW && !NOVAL), "4-byte keys: the flush's own passes over packed entries");
    constexpr int RDX = NINE ? RADIX : 256;
    __shared__ u64 lbuf[TILE];
    __shared__ u32 cnt[WAVES][RDX];
    __shared__ u32 dstart[RDX];
    __shared__ i64 goff[RDX];
    __shared__ u32 lw[WAVES];
    __shared__ unsigned char ldig[SLOTDIG ? 1 : TILE];

    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const i64 tile = xcd_tile();
    const int s = p.S == 1 ? (tile < p.tile_first[1] ? 0 : -1) : find_segment(p.tile_first, p.S, tile);
    if (s < 0) return;
    const i64 tf = p.tile_first[s];
    const i64 nts = p.tile_first[s + 1] - tf;
    const i64 tin = tile - tf;
    const i64 beg = p.seg_start[s] + tin * TILE;
    const i64 end = min(p.seg_start[s + 1], beg + (i64)TILE);
    const int ntile = (int)(end - beg);
    const u32 mask = (1u << p.bits) - 1u;
    const int R = 1 << p.bits;

#pragma unroll
    for (int i = 0; i < WAVES; i++) {
        cnt[i][t] = 0;
        if constexpr (NINE) cnt[i][t + THREADS] = 0;
    }

    // wave-striped arrangement: memory order == (wave, k, lane) order.  Keys AND values are
    // loaded up front: 32 independent 8-byte loads per lane in flight (the kernel runs at two
    // waves per SIMD, so the register file has room for them)
    u64 key[ITEMS];
    double val[ITEMS];
    const i64 wbase = beg + (i64)w * (ESP_WAVE * ITEMS) + lane;
    if constexpr (RAW) {
        i64 rr[ITEMS], cc[ITEMS];
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
            const i64 idx = wbase + k * ESP_WAVE;
            rr[k] = idx < end ? p.raw_rows[idx] : 1;
            cc[k] = idx < end ? p.raw_cols[idx] : 1;
        }
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
            const i64 idx = wbase + k * ESP_WAVE;
            i64 r = rr[k], c = cc[k];
            if (idx < end && (r < 1 || r > p.raw_m || c < 1 || c > p.raw_n)) {
                atomicMin(p.raw_err, (unsigned long long)idx + 1ull);
                r = 1, c = 1;  // (a valid key: the host stops before the batch is counted in)
            }
            key[k] = idx < end ? (((((u64)(c - 1) << p.raw_rb) | (u64)(r - 1)) << ESP_TAG_BITS) | (u64)p.raw_kind) : ~0ull;
        }
    } else {
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
            const i64 idx = wbase + k * ESP_WAVE;
            if constexpr (K32 == 2)
                key[k] = idx < end ? (u64) reinterpret_cast<const u32 *>(p.keys_in)[idx] : ~0ull;
            else
                key[k] = idx < end ? p.keys_in[idx] : ~0ull;
        }
    }
    if constexpr (!NOVAL) {
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
            const i64 idx = wbase + k * ESP_WAVE;
            val[k] = idx < end ? p.vals_in[idx] : 0.0;
            if constexpr (RAW) {
                if (p.raw_negate) val[k] = -val[k];
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < ITEMS; k++) val[k] = 0.0;
    }
    __syncthreads();

    // stable rank inside the wave by ballot matching; one-digit waves skip the 8 ballots
    unsigned short rank[ITEMS];
    unsigned short dig[ITEMS];
    const u64 lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
        const bool valid = (wbase + k * ESP_WAVE) < end;
        const u32 d = valid ? digit_of<false, K32 == 2>(p, key[k], mask) : 0u;
        dig[k] = (unsigned short)d;
        const u64 vm = __ballot(valid);
        u64 m = vm;
        if (vm) {
            const u32 d0 = (u32)__shfl((int)d, __builtin_ctzll(vm), ESP_WAVE);
            if (__ballot(valid && d == d0) != vm) {
#pragma unroll
                for (int b = 0; b < MAX_BITS; b++) {
                    if (b < p.bits) {  // (uniform: a pass of 6 bits pays 6 ballots)
                        const bool bit = (d >> b) & 1u;
                        const u64 bb = __ballot(bit);
                        m &= bit ? bb : ~bb;
                    }
                }
            }
        }
        u32 prev = 0;
        if (valid) prev = cnt[w][d];
        rank[k] = (unsigned short)(prev + (u32)__popcll(m & lt));
        __builtin_amdgcn_wave_barrier();
        if (valid && (m & lt) == 0) cnt[w][d] = prev + (u32)__popcll(m);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();

    // per-digit: exclusive prefix over the waves, tile total, LDS start, global offset (NINE: a thread owns two
    // neighbouring digits and the block scan runs over their sum)
    if constexpr (NINE) {
        const int d0 = 2 * t, d1 = 2 * t + 1;
        u32 tot0 = 0, tot1 = 0;
#pragma unroll
        for (int i = 0; i < WAVES; i++) {
            const u32 c0 = cnt[i][d0], c1 = cnt[i][d1];
            cnt[i][d0] = tot0;
            cnt[i][d1] = tot1;
            tot0 += c0;
            tot1 += c1;
        }
        u32 blocktot;
        const u32 ds = espscan::block_exclusive<u32, false>(tot0 + tot1, lw, &blocktot);
        dstart[d0] = ds;
        dstart[d1] = ds + tot0;
        if (d0 < R && tot0 != 0) goff[d0] = (i64)p.hist[tf * R + (i64)d0 * nts + tin] - (i64)ds;
        if (d1 < R && tot1 != 0) goff[d1] = (i64)p.hist[tf * R + (i64)d1 * nts + tin] - (i64)(ds + tot0);
    } else {
        u32 tot = 0;
        u32 c[WAVES];
#pragma unroll
        for (int i = 0; i < WAVES; i++) c[i] = cnt[i][t];
#pragma unroll
        for (int i = 0; i < WAVES; i++) {
            cnt[i][t] = tot;
            tot += c[i];
        }
        u32 blocktot;
        const u32 ds = espscan::block_exclusive<u32, false>(tot, lw, &blocktot);
        dstart[t] = ds;
        if (t < R && tot != 0) goff[t] = (i64)p.hist[tf * R + (i64)t * nts + tin] - (i64)ds;
    }
    __syncthreads();

#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
        const i64 idx = wbase + k * ESP_WAVE;
        if (idx < end) {
            const u32 d = dig[k];
            const u32 slot = dstart[d] + cnt[w][d] + rank[k];
            rank[k] = (unsigned short)slot;  // (kept for the values)
            lbuf[slot] = key[k];
            if constexpr (!SLOTDIG) ldig[slot] = (unsigned char)d;
        }
    }
    __syncthreads();
    double *lvals = reinterpret_cast<double *>(lbuf);
    if constexpr (SLOTDIG) {
        // (the digit of an output slot: from its key once, kept in a register for the values)
        unsigned short sdig[ITEMS];
#pragma unroll
        for (int j = 0; j < ITEMS; j++) {
            const int slot = t + j * THREADS;
            sdig[j] = 0;
            if (slot < ntile) {
                const u64 kk = lbuf[slot];
                const u32 d = digit_of<false, K32 == 2>(p, kk, mask);
                sdig[j] = (unsigned short)d;
                if constexpr (K32 != 0)
                    reinterpret_cast<u32 *>(p.keys_out)[goff[d] + slot] = narrow_key<K32 == 2>(p, kk);
                else
                    p.keys_out[goff[d] + slot] = kk;
            }
        }
        if constexpr (NOVAL) return;
        __syncthreads();  // (every key has been read)
#pragma unroll
        for (int k = 0; k < ITEMS; k++)
            if (wbase + k * ESP_WAVE < end) lvals[rank[k]] = val[k];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < ITEMS; j++) {
            const int slot = t + j * THREADS;
            if (slot < ntile) p.vals_out[goff[sdig[j]] + slot] = lvals[slot];
        }
    } else {
#pragma unroll 4
        for (int j = 0; j < ITEMS; j++) {
            const int slot = t + j * THREADS;
            if (slot < ntile) {
                if constexpr (K32 != 0)
                    reinterpret_cast<u32 *>(p.keys_out)[goff[ldig[slot]] + slot] = narrow_key<K32 == 2>(p, lbuf[slot]);
                else
                    p.keys_out[goff[ldig[slot]] + slot] = lbuf[slot];
            }
        }
        if constexpr (NOVAL) return;
        __syncthreads();  // (every key has been read)
#pragma unroll
        for (int k = 0; k < ITEMS; k++)
            if (wbase + k * ESP_WAVE < end) lvals[rank[k]] = val[k];
        __syncthreads();
#pragma unroll 4
        for (int j = 0; j < ITEMS; j++) {
            const int slot = t + j * THREADS;
            if (slot < ntile) p.vals_out[goff[ldig[slot]] + slot] = lvals[slot];
        }
    }
}

// after the scan: start of segment (s,d) = hist[tile_first[s]*R + d*ntiles_s]
static __global__ void new_segments_k(const u64 *__restrict__ hist, const i64 *__restrict__ seg_start,
                               const i64 *__restrict__ tile_first, int S, int bits,
                               i64 *__restrict__ new_seg_start, i64 total) {
    const i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const int R = 1 << bits;
    const i64 NS = (i64)S * R;
    if (g > NS) return;
    if (g == NS) {
        new_seg_start[g] = total;
        return;
    }
    const int s = (int)(g >> bits);
    const int d = (int)(g & (R - 1));
    const i64 nts = tile_first[s + 1] - tile_first[s];
    // empty segment: every (s,d) starts where the segment starts
    new_seg_start[g] = nts == 0 ? seg_start[s] : (i64)hist[tile_first[s] * R + (i64)d * nts];
}

// tiles per segment -> tile_first by exclusive scan (caller scans); also max segment length
static __global__ void seg_tiles_k(const i64 *__restrict__ seg_start, i64 S, i64 tile, u64 *__restrict__ ntiles,
                            unsigned long long *__restrict__ maxlen) {
    const i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    i64 len = 0;
    if (g < S) {
        len = seg_start[g + 1] - seg_start[g];
        ntiles[g] = (u64)((len + tile - 1) / tile);
    } else if (g == S) {
        ntiles[g] = 0;
    }
    // (one atomic per wave: 2^17 segments drawing on one word took 28 us where the rest of the kernel takes 5)
    if (maxlen) {
        u32 lo = (u32)len, hi = (u32)((u64)len >> 32);  // (segment lengths are below 2^32 in every flush: the high word decides first)
        const u32 mh = esp_wave_max(hi);
        lo = hi == mh ? lo : 0u;
        const u32 ml = esp_wave_max(lo);
        if ((threadIdx.x & 63) == 0 && (mh | ml)) atomicMax(maxlen, ((unsigned long long)mh << 32) | (unsigned long long)ml);
    }
}

// new_segments_k + seg_tiles_k + the scan of the tile counts in ONE launch, for a table of at most 2047 segments (the first pass
// of a flush: 128 .. 512 segments) -- four launches and a memset less; *maxlen is written, not maximised (nobody else writes
// it in this launch)
constexpr int SMALL_TABLE = 2048;
static __global__ __launch_bounds__(256) void small_table_k(const u64 *__restrict__ hist, const i64 *__restrict__ seg_start,
                                                        const i64 *__restrict__ tile_first, int S, int bits,
                                                        i64 *__restrict__ new_seg_start, i64 total, i64 tile,
                                                        u64 *__restrict__ new_tile_first, unsigned long long *__restrict__ maxlen) {
    __shared__ i64 st[SMALL_TABLE + 1];
    __shared__ u64 lw[4];
    __shared__ u64 wmax[4];
    const int t = threadIdx.x;
    const int R = 1 << bits;
    const int NS = S * R;  // (< SMALL_TABLE)
    for (int g = t; g <= NS; g += 256) {
        i64 v = total;
        if (g < NS) {
            const int s = g >> bits, d = g & (R - 1);
            const i64 nts = tile_first[s + 1] - tile_first[s];
            v = nts == 0 ? seg_start[s] : (i64)hist[tile_first[s] * R + (i64)d * nts];
        }
        st[g] = v;
        new_seg_start[g] = v;
    }
    __syncthreads();
    // tile counts of 8 consecutive segments per thread, exclusive scan over the block
    u64 cnt[8], run = 0, mx = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int g = t * 8 + k;
        u64 c = 0;
        if (g < NS) {
            const u64 len = (u64)(st[g + 1] - st[g]);
            mx = len > mx ? len : mx;
            c = (len + (u64)tile - 1) / (u64)tile;
        }
        cnt[k] = run;
        run += c;
    }
    u64 tot;
    const u64 pre = espscan::block_exclusive<u64, false>(run, lw, &tot);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int g = t * 8 + k;
        if (g <= NS) new_tile_first[g] = pre + cnt[k];
    }
    // the longest segment
    for (int o = 32; o; o >>= 1) {
        const u64 x = __shfl_xor(mx, o, 64);
        mx = x > mx ? x : mx;
    }
    if ((t & 63) == 0) wmax[t >> 6] = mx;
    __syncthreads();
    if (t == 0) {
        u64 m = wmax[0];
        for (int i = 1; i < 4; i++) m = wmax[i] > m ? wmax[i] : m;
        *maxlen = m;
    }
}

}  // namespace espradix
