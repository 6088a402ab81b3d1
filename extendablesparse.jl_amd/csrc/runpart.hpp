// runpart.hpp -- single-pass stable partition on up to 16 key bits for pre-sorted streams.
//
// Assembly loops emit their updates in an order with strong locality (a tile of 4096 consecutive
// stencil / natural-order FEM updates touches only a handful of the 65536 column blocks).  The
// classic partition needs two 8-bit passes (each reads and writes every entry) for 16 bits; here a
// tile is described by its few RUNS (digit, count) instead of a 65536-bin histogram:
//
//   run_hist_k     per tile: distinct digits and their counts (<= RMAX, else the flush falls
//                  back to the 8-bit passes), bucket totals by a few atomics; every digit also
//                  collects its own runs (tile, index in the tile, count)      (reads 8 B/entry)
//   run_coarse_k + run_rank_k   bucket starts (block scan + sums of 256 totals) and, one thread per
//                  digit, the output offset of each of its runs = bucket start + the entries of its
//                  runs from earlier tiles (a digit with more than DCAP runs: the run list of all
//                  tiles is ordered by the 8-bit pass kernels and scanned instead, host-driven)
//   run_scatter_k  per tile: stable rank of every entry inside its run, stores straight from
//                  registers to run offset + rank  (reads 16 B, writes 16 B/entry -- 12 B when all
//                  pending entries share one kind and <= 32 key bits remain: 4-byte keys)
//
// Stability: inside a bucket the runs are ordered by tile, inside a run the entries keep the
// (wave, item, lane) = memory order.  Deterministic: the only atomics are integer counters.
#pragma once
#include "common.hpp"
#include "radix.hpp"
#include "scan.hpp"

namespace esprun {

constexpr int THREADS = 256;
constexpr int WAVES = THREADS / ESP_WAVE;
constexpr int ITEMS = 16;
constexpr int TILE = THREADS * ITEMS;  // same tiles as espradix
constexpr int RMAX = 64;               // distinct digits a tile may hold on this path
constexpr u32 EMPTY = 0xFFFFFFFFu;
constexpr int DCAP = 32;               // runs one digit may collect in its own list (ranked path)

struct Args {
    const u64 *keys_in;
    const double *vals_in;
    u64 *keys_out;
    double *vals_out;
    i64 E;
    const i64 *chunk_start;  // C+1 buffer positions; chunk c = [chunk_start[c], chunk_start[c+1]), <= TILE entries
    int fixed_chunks;        // chunk c = [c*TILE, min(E, (c+1)*TILE)): the kernels compute the bounds instead of loading them
    int shift;  // digit = (((key >> 2) - base) >> shift), digits < nbuckets
    u64 base, span;
    u32 *err;        // key outside the window
    u32 *overflow;   // a tile holds more than RMAX distinct digits
    u32 *runs_d;     // [tile][RMAX] digits (distinct, any order)
    u32 *runs_c;     // [tile][RMAX] counts
    i64 *runs_off;   // [tile][RMAX] global output offset of the run
    u64 *nruns;      // [tile] (+1 slot for the scan)
    unsigned long long *bucket_count;  // [nbuckets + 1]
    // ranked path (run_rank_k): every digit collects its runs (chunk | index in the chunk | count) itself
    u32 *dcount;     // [nbuckets] runs of the digit
    u64 *dlist;      // [nbuckets][DCAP]
    int nruns_raw;   // nruns holds the counts themselves, not their scan
    // K32 scatter kernels: when the longest bucket fits the bucket kernel (*maxlen <= cap) the key goes out as 4
    // bytes -- its bits below the partition prefix, (key >> 2) - base masked to `shift` (<= 32) bits; the kind is the
    // same for all pending entries and travels outside the data
    const unsigned long long *maxlen;
    i64 cap;
    const u32 *flags;  // run_scatter_k leaves at once when flags[0], [1] or [3] is set (window error, too many
                       // digits in a chunk, too many runs of a digit): the host reads them while it runs
    // several key windows side by side (column shards, MULTI kernels): window r starts at key mw_base[r]
    // (ascending), holds mw_nb digits of width 2^shift; global digit = r * mw_nb + local digit
    int mw_P;
    u32 mw_nb;
    const u64 *mw_base;
};
constexpr int MW_MAX = 64;  // windows the MULTI kernels take

// (the kind bits ride along: one 64-bit subtraction and one shift per key; the histogram kernel
// checks the window -- the host rejects the flush before the scatter kernel runs -- and only then clamps)
__device__ __forceinline__ u32 digit16(const Args &a, u64 key, bool check) {
    u64 rel = key - (a.base << ESP_TAG_BITS);
    if (check && rel >= (a.span << ESP_TAG_BITS)) {
        *a.err = 1u;
        rel = 0;
    }
    return (u32)(rel >> (a.shift + ESP_TAG_BITS));
}

// MULTI: the window of a key -- almost always the one of the wave's first key (r0, found once per wave
// with scalar code), else a binary search in the LDS copy of mw_base -- then the digit inside it
struct WaveWin {
    u32 d0;      // first global digit of the window of the wave's first key
    u64 base4;   // its first key, as a packed key (kind bits = 0)
    u64 width4;  // packed keys up to the next window's first key
};
__device__ __forceinline__ WaveWin wave_window(const Args &a, const u64 *s_mw, u64 first_key) {
    const u64 kp = esp_uniform_u64(first_key) >> ESP_TAG_BITS;
    int r = 0;
    for (int step = MW_MAX / 2; step; step >>= 1) {
        const int c = r + step;
        if (c < a.mw_P && kp >= esp_uniform_u64(s_mw[c])) r = c;
    }
    const u64 b0 = esp_uniform_u64(s_mw[r]);
    // (a window ends where the next one starts; the last one is open -- up to the largest key, so that a key
    // BELOW the window, whose difference wraps around, still fails the width test)
    const u64 b4 = b0 << ESP_TAG_BITS;
    const u64 w4 = r + 1 < a.mw_P ? (esp_uniform_u64(s_mw[r + 1]) - b0) << ESP_TAG_BITS : ~0ull - b4;
    return WaveWin{(u32)r * a.mw_nb, b4, w4};
}
template <bool MULTI>
__device__ __forceinline__ u32 digit_mw(const Args &a, const u64 *s_mw, const WaveWin &ww, u64 key, bool check) {
    if constexpr (!MULTI) {
        return digit16(a, key, check);
    } else {
        // (the kind bits ride along, as in digit16: one subtraction, one compare, one shift)
        const u64 rel0 = key - ww.base4;
        if (rel0 < ww.width4) return ww.d0 + (u32)(rel0 >> (a.shift + ESP_TAG_BITS));
        const u64 kp = key >> ESP_TAG_BITS;
        int r = 0;
#pragma unroll
        for (int step = MW_MAX / 2; step; step >>= 1) {
            const int c = r + step;
            if (c < a.mw_P && kp >= s_mw[c]) r = c;
        }
        u64 dl = (kp - s_mw[r]) >> a.shift;
        dl = dl < (u64)a.mw_nb ? dl : (u64)a.mw_nb - 1;
        return (u32)r * a.mw_nb + (u32)dl;
    }
}

// digit of a key for producers that emit their chunk's run list themselves
__device__ __forceinline__ u32 run_digit(u64 key, u64 base, u64 span, int shift, u32 *err) {
    u64 kn = (key >> ESP_TAG_BITS) - base;
    if (kn >= span) *err = 1u;
    kn = kn < span ? kn : span - 1;
    return (u32)(kn >> shift);
}

// where a chunk's run list goes (shared by run_hist_k and the producers that emit it themselves)
struct RunSink {
    u32 *runs_d;
    u32 *runs_c;
    u64 *nruns;
    unsigned long long *bucket_count;
    u32 *overflow;
    u32 *dcount = nullptr;  // ranked path: the digit's own run list (nullptr: not collected)
    u64 *dlist = nullptr;
};

// Digit-major counting of one chunk by a whole workgroup: every wave walks the DISTINCT digits of its
// entries (a handful on a pre-sorted stream); for each one, NITEMS ballots count its entries.  The
// workgroup's table (rd/rc/over in LDS, initialised and barrier'd by the caller) collects the waves'
// results.  pend: bit k set = item k of this lane is an entry.
template <int NITEMS>
__device__ __forceinline__ void count_runs(const u32 (&dig)[NITEMS], u32 pend, i64 chunk, const RunSink &sink, u32 *rd,
                                           u32 *rc, u32 *over) {
    const int t = threadIdx.x, lane = t & 63;
    // counted entries are blanked (EMPTY never equals a digit): a hit test is one compare
    u32 dg[NITEMS];
#pragma unroll
    for (int k = 0; k < NITEMS; k++) dg[k] = ((pend >> k) & 1u) ? dig[k] : EMPTY;
    int trips = 0;
    bool stop = false;
#pragma unroll
    for (int k = 0; k < NITEMS; k++) {
        // digits first met at item k (items before k are fully counted)
        u64 m = stop ? 0ull : __ballot(dg[k] != EMPTY);
        while (m) {
            if (++trips > RMAX) {  // more distinct digits than a chunk may hold on this path
                if (lane == 0) *over = 1;
                stop = true;
                break;
            }
            const int fl = __builtin_ctzll(m);
            const u32 c0 = (u32)__builtin_amdgcn_readlane((int)dg[k], fl);
            u32 total = 0;
#pragma unroll
            for (int q = k; q < NITEMS; q++) {
                const bool hit = dg[q] == c0;
                total += (u32)__popcll(__ballot(hit));
                dg[q] = hit ? EMPTY : dg[q];
            }
            if (lane == 0) {  // open addressing in the workgroup's run table
                bool placed = false;
                int j = (int)((c0 * 0x9E3779B1u) >> 26) & (RMAX - 1);
                for (int probe = 0; probe < RMAX; probe++) {
                    const u32 old = atomicCAS(&rd[j], EMPTY, c0);
                    if (old == EMPTY || old == c0) {
                        atomicAdd(&rc[j], total);
                        placed = true;
                        break;
                    }
                    j = (j + 1) & (RMAX - 1);
                }
                if (!placed) *over = 1;
            }
            m = __ballot(dg[k] != EMPTY);
        }
    }
    __syncthreads();
    if (*over) {
        if (t == 0) {
            sink.nruns[chunk] = 0;
            atomicExch(sink.overflow, 1u);
        }
        return;
    }
    // the used table slots, densely (RMAX == one wave; no order among the runs of a chunk is needed: the
    // run list of all chunks is sorted by digit afterwards and a chunk's digits are distinct)
    static_assert(RMAX == ESP_WAVE, "the run table is compacted by one wave");
    if (t < RMAX) {
        const u32 x = rd[t];
        const u64 used = __ballot(x != EMPTY);
        if (x != EMPTY) {
            const int r = __popcll(used & ((1ull << t) - 1ull));
            sink.runs_d[chunk * RMAX + r] = x;
            sink.runs_c[chunk * RMAX + r] = rc[t];
            atomicAdd(&sink.bucket_count[x], (unsigned long long)rc[t]);
            if (sink.dlist) {  // (any order: run_rank_k orders a digit's few runs by chunk)
                const u32 slot = atomicAdd(&sink.dcount[x], 1u);
                if (slot < (u32)DCAP) sink.dlist[(size_t)x * DCAP + slot] = ((u64)chunk << 24) | ((u64)r << 16) | (u64)rc[t];
            }
        }
        if (t == 0) sink.nruns[chunk] = (u64)__popcll(used);
    }
}

template <bool MULTI>
__global__ __launch_bounds__(THREADS) void run_hist_k(Args a, i64 first_chunk) {
    __shared__ u32 rd[RMAX];
    __shared__ u32 rc[RMAX];
    __shared__ u32 over;
    __shared__ u64 s_mw[MULTI ? MW_MAX : 1];
    const int t = threadIdx.x;
    if (MULTI && t < a.mw_P) s_mw[t] = a.mw_base[t];
    const i64 chunk = first_chunk + blockIdx.x;
    const i64 beg = a.fixed_chunks ? chunk * TILE : a.chunk_start[chunk];
    const i64 end = a.fixed_chunks ? min(a.E, beg + (i64)TILE) : a.chunk_start[chunk + 1];
    // a stream that is not pre-sorted is recognised by the first workgroups: the rest leave at once
    // (the flag is read while the keys are in flight)
    const u32 give_up = __hip_atomic_load(a.overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t < RMAX) {
        rd[t] = EMPTY;
        rc[t] = 0;
    }
    if (t == 0) over = 0;
    u64 key[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
        const i64 idx = beg + k * THREADS + t;
        key[k] = idx < end ? a.keys_in[idx] : 0ull;
    }
    if (give_up != 0u) return;
    __syncthreads();
    u32 dig[ITEMS];
    u32 pend = 0;  // bit k: item k of this lane not yet counted
    WaveWin ww{0, 0, 0};
    if constexpr (MULTI) ww = wave_window(a, s_mw, key[0]);
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
        const bool valid = (beg + k * THREADS + t) < end;
        dig[k] = valid ? digit_mw<MULTI>(a, s_mw, ww, key[k], true) : 0u;
        pend |= valid ? (1u << k) : 0u;
    }
    const RunSink sink{a.runs_d, a.runs_c, a.nruns, a.bucket_count, a.overflow, a.dcount, a.dlist};
    count_runs<ITEMS>(dig, pend, chunk, sink, rd, rc, &over);
}

// chunk_start[first + i] = from + i*TILE (clamped to `to`): fixed-size chunks for a buffer range that
// came without run lists
__global__ void fixed_chunks_k(i64 *chunk_start, i64 first, i64 nchunks, i64 from, i64 to) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > nchunks) return;
    chunk_start[first + i] = min(to, from + i * (i64)TILE);
}

// dense run list in tile order: sortable records (key = digit << 2, payload = tile|j|count)
__global__ void run_pack_k(const u32 *__restrict__ runs_d, const u32 *__restrict__ runs_c,
                           const u64 *__restrict__ run_base /* exclusive scan of nruns, T+1 */, i64 T,
                           u64 *__restrict__ lk, double *__restrict__ lv) {
    const i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 tile = g / RMAX;
    const int j = (int)(g % RMAX);
    if (tile >= T) return;
    const i64 b = (i64)run_base[tile];
    const int nr = (int)((i64)run_base[tile + 1] - b);
    if (j >= nr) return;
    lk[b + j] = (u64)runs_d[g] << ESP_TAG_BITS;
    const u64 payload = ((u64)tile << 24) | ((u64)j << 16) | (u64)runs_c[g];
    lv[b + j] = __longlong_as_double((long long)payload);
}

__global__ void run_counts_k(const double *__restrict__ lv, i64 R, u64 *__restrict__ cnt) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > R) return;
    cnt[i] = i < R ? ((u64)__double_as_longlong(lv[i]) & 0xFFFFull) : 0ull;
}

// head[d] = scanned count at the first record of digit d
__global__ void run_heads_k(const u64 *__restrict__ lk, const u64 *__restrict__ sc, i64 R, u64 *__restrict__ head) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R) return;
    const u64 d = lk[i] >> ESP_TAG_BITS;
    if (i == 0 || (lk[i - 1] >> ESP_TAG_BITS) != d) head[d] = sc[i];
}

__global__ void run_offsets_k(const u64 *__restrict__ lk, const double *__restrict__ lv, const u64 *__restrict__ sc,
                              const u64 *__restrict__ head, const u64 *__restrict__ bucket_start, i64 R,
                              i64 *__restrict__ runs_off) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R) return;
    const u64 d = lk[i] >> ESP_TAG_BITS;
    const u64 payload = (u64)__double_as_longlong(lv[i]);
    const i64 tile = (i64)(payload >> 24);
    const int j = (int)((payload >> 16) & 0xFF);
    runs_off[tile * RMAX + j] = (i64)(bucket_start[d] + sc[i] - head[d]);
}

// Ranked path: two launches between the histogram and the scatter kernel.
//   run_coarse_k   coarse[b] = entries of the digits [256 b, 256 b + 256)
//   run_rank_k     one thread per digit d:
//     seg_out[d]  = entries of all digits before d (the workgroup sums the coarse totals before its own 256
//                   digits and scans its own), seg_out[nb] = all entries
//     runs_off    = seg_out[d] + the entries of the digit's runs from earlier chunks, for each of its runs
//                   (<= DCAP of them, a handful on a pre-sorted stream: quadratic walk over an LDS copy)
//     maxlen      = the longest bucket; *too_many = 1 when some digit has more than DCAP runs (the host
//                   then orders the run list with the radix passes instead)
__global__ __launch_bounds__(THREADS) void run_coarse_k(const unsigned long long *__restrict__ bucket_count, i64 nb,
                                                         u64 *__restrict__ coarse) {
    __shared__ u64 lw[WAVES];
    const i64 d = (i64)blockIdx.x * THREADS + threadIdx.x;
    u64 tot;
    espscan::block_exclusive<u64, false>(d < nb ? (u64)bucket_count[d] : 0ull, lw, &tot);
    if (threadIdx.x == 0) coarse[blockIdx.x] = tot;
}

__global__ __launch_bounds__(THREADS) void run_rank_k(const unsigned long long *__restrict__ bucket_count,
                                                       const u64 *__restrict__ coarse, const u32 *__restrict__ dcount,
                                                       const u64 *__restrict__ dlist, i64 nb, i64 *__restrict__ seg_out,
                                                       i64 *__restrict__ runs_off, unsigned long long *maxlen, u32 *too_many) {
    __shared__ u64 rows[DCAP * THREADS];  // [i][t]: run i of thread t's digit (64 KiB)
    u64 *lw = rows;                       // (the block scans are over before the rows are filled)
    const int t = threadIdx.x;
    const i64 d = (i64)blockIdx.x * THREADS + t;
    const u64 own = d < nb ? (u64)bucket_count[d] : 0ull;
    const u32 n = d < nb ? dcount[d] : 0u;
    u64 before = 0;
    for (i64 i = t; i < (i64)blockIdx.x; i += THREADS) before += coarse[i];
    u64 base, tot;
    espscan::block_exclusive<u64, false>(before, lw, &base);
    const u64 start = base + espscan::block_exclusive<u64, false>(own, lw, &tot);
    if (d <= nb) seg_out[d] = (i64)start;
    {
        u64 mx = own;
#pragma unroll
        for (int o = 32; o; o >>= 1) {
            const u64 x = __shfl_xor(mx, o, ESP_WAVE);
            mx = x > mx ? x : mx;
        }
        if ((t & 63) == 0 && mx) atomicMax(maxlen, (unsigned long long)mx);
    }
    if (n > (u32)DCAP) {
        *too_many = 1u;
        return;
    }
    const u64 *mine = dlist + (size_t)d * DCAP;
    for (u32 i = 0; i < n; i++) rows[i * THREADS + t] = mine[i];
    for (u32 i = 0; i < n; i++) {
        const u64 e = rows[i * THREADS + t];
        const u64 chunk = e >> 24;
        u64 off = 0;
        for (u32 k = 0; k < n; k++) {
            const u64 o = rows[k * THREADS + t];
            off += (o >> 24) < chunk ? (o & 0xFFFFull) : 0ull;
        }
        runs_off[chunk * RMAX + ((e >> 16) & 0xFFull)] = (i64)(start + off);
    }
}

template <bool MULTI, bool K32>
__global__ __launch_bounds__(THREADS) void run_scatter_k(Args a) {
    __shared__ u64 s_mw[MULTI ? MW_MAX : 1];
    __shared__ u32 hd[RMAX];  // open-addressing map digit -> run index of this tile
    __shared__ u32 hj[RMAX];
    __shared__ i64 roff[RMAX];
    __shared__ u32 cnt[WAVES][RMAX];
    __shared__ int s_nr;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const i64 tile = blockIdx.x;  // = chunk index
    const i64 beg = a.fixed_chunks ? tile * TILE : a.chunk_start[tile];
    const i64 end = a.fixed_chunks ? min(a.E, beg + (i64)TILE) : a.chunk_start[tile + 1];
    // run table of the tile (nruns holds the exclusive scan by now: nr = difference)
    if (a.flags && (a.flags[0] | a.flags[1] | a.flags[3]) != 0u) return;  // (uniform: the flush takes another path)
    if (t == 0) s_nr = a.nruns_raw ? (int)a.nruns[tile] : (int)(a.nruns[tile + 1] - a.nruns[tile]);
    if (t < RMAX) {
        roff[t] = a.runs_off[tile * RMAX + t];
        hd[t] = EMPTY;
    }
    if (t < WAVES * RMAX) (&cnt[0][0])[t] = 0;
    if (MULTI && t < a.mw_P) s_mw[t] = a.mw_base[t];
    u64 key[ITEMS];
    double val[ITEMS];
    const i64 wbase = beg + (i64)w * (ESP_WAVE * ITEMS) + lane;
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
        const i64 idx = wbase + k * ESP_WAVE;
        key[k] = idx < end ? a.keys_in[idx] : 0ull;
    }
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
        const i64 idx = wbase + k * ESP_WAVE;
        val[k] = idx < end ? a.vals_in[idx] : 0.0;
    }
    __syncthreads();
    const int nr = s_nr;
    if (t < nr) {  // build the map (digits of a tile are distinct)
        const u32 dd = a.runs_d[tile * RMAX + t];
        int j = (int)((dd * 0x9E3779B1u) >> 26) & (RMAX - 1);
        for (int probe = 0; probe < RMAX; probe++) {
            if (atomicCAS(&hd[j], EMPTY, dd) == EMPTY) {
                hj[j] = (u32)t;
                break;
            }
            j = (j + 1) & (RMAX - 1);
        }
    }
    __syncthreads();
    const u64 lt = (1ull << lane) - 1ull;
    unsigned short rank[ITEMS];
    unsigned char jrun[ITEMS];
    u32 dig[ITEMS];
    u32 pend = 0;
    WaveWin ww{0, 0, 0};
    if constexpr (MULTI) ww = wave_window(a, s_mw, key[0]);
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
        const bool valid = (wbase + k * ESP_WAVE) < end;
        dig[k] = valid ? digit_mw<MULTI>(a, s_mw, ww, key[k], false) : 0u;
        pend |= valid ? (1u << k) : 0u;
        rank[k] = 0;
        jrun[k] = 0;
    }
    // digit-major stable ranking: for each distinct digit of the wave, its entries are numbered in
    // (item, lane) order with a scalar running count -- no LDS counters inside the wave.  Ranked
    // entries are blanked, so the hit test is one compare; digits first met at item k only look at
    // items >= k.
#pragma unroll
    for (int k = 0; k < ITEMS; k++) dig[k] = ((pend >> k) & 1u) ? dig[k] : EMPTY;
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
        u64 m = __ballot(dig[k] != EMPTY);
        while (m) {
            const int fl = __builtin_ctzll(m);
            const u32 c0 = (u32)__builtin_amdgcn_readlane((int)dig[k], fl);
            int jj = 0;
            {
                int j = (int)((c0 * 0x9E3779B1u) >> 26) & (RMAX - 1);
                for (int probe = 0; probe < RMAX; probe++) {
                    if (hd[j] == c0) {
                        jj = (int)hj[j];
                        break;
                    }
                    j = (j + 1) & (RMAX - 1);
                }
            }
            u32 running = 0;
#pragma unroll
            for (int q = k; q < ITEMS; q++) {
                const bool hit = dig[q] == c0;
                const u64 mm = __ballot(hit);
                if (hit) {
                    rank[q] = (unsigned short)(running + (u32)__popcll(mm & lt));
                    jrun[q] = (unsigned char)jj;
                }
                running += (u32)__popcll(mm);
                dig[q] = hit ? EMPTY : dig[q];
            }
            if (lane == 0) cnt[w][jj] = running;
            m = __ballot(dig[k] != EMPTY);
        }
    }
    __syncthreads();
    // exclusive prefix over the waves, per run
    if (t < RMAX) {
        u32 run = 0;
#pragma unroll
        for (int i = 0; i < WAVES; i++) {
            const u32 x = cnt[i][t];
            cnt[i][t] = run;
            run += x;
        }
    }
    __syncthreads();
    bool short_keys = false;
    if constexpr (K32) short_keys = *a.maxlen <= (unsigned long long)a.cap;  // (uniform; the host applies the same rule)
    const u32 kmask = a.shift >= 32 ? 0xFFFFFFFFu : ((1u << a.shift) - 1u);
    const u64 base4 = a.base << ESP_TAG_BITS;
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
        const i64 idx = wbase + k * ESP_WAVE;
        if (idx < end) {
            const int jj = jrun[k];
            const i64 dst = roff[jj] + (i64)cnt[w][jj] + (i64)rank[k];
            if (K32 && short_keys)
                reinterpret_cast<u32 *>(a.keys_out)[dst] = (u32)((key[k] - base4) >> ESP_TAG_BITS) & kmask;
            else
                a.keys_out[dst] = key[k];
            a.vals_out[dst] = val[k];
        }
    }
}

}  // namespace esprun
