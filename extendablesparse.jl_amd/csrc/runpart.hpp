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
constexpr int PROBE_CHUNKS = 512;       // run_hist_k's first launch (2 M entries): a stream that is not pre-sorted shows there
constexpr int MAX_PB = 20;             // prefix bits of the run-based single pass / a producer-side partition (2^20 buckets: the run lists' arrays)
constexpr u32 EMPTY = 0xFFFFFFFFu;
constexpr int DCAP = 32;               // runs one digit may collect in its own list (ranked path)

struct Args {
    const u64 *keys_in;
    const double *vals_in;
    u64 *keys_out;
    double *vals_out;
    i64 E;
    // chunk c = [c*TILE, min(E, (c+1)*TILE)) of the pending buffer
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
    // raw source (esp_append_device / esp_append_host / esp_commit on an empty buffer: the append is the partition):
    // raw_cols != nullptr -- the keys are formed on the fly from 1-based (row, col) arrays, every entry of the kind
    // raw_kind, vals_in holds the caller's values (negated for op = SUB unless the kind is SET); nothing is read twice
    // and the packed stream is never written: run_hist_k reads the columns (8 B per entry), run_scatter_k rows, columns
    // and values (24 B) and stores key and value at their bucket position
    const i64 *raw_rows = nullptr, *raw_cols = nullptr;
    i64 raw_m = 0, raw_n = 0;
    int raw_rb = 0, raw_kind = 0, raw_negate = 0;
    unsigned long long *raw_err = nullptr;  // atomicMin(position + 1) of the first entry outside m x n (BoundsError)
    // VERIFY kernels (a caller that repeats its stream: the run lists of the previous assembly are used again, no count
    // pass): a tile whose digits or counts are not what its run list says stores nothing and raises *verify_err
    u32 *verify_err = nullptr;
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

// one wave's total for digit c0 goes into the workgroup's run table (open addressing; lane 0 calls it)
__device__ __forceinline__ void run_table_add(u32 *rd, u32 *rc, u32 *over, u32 c0, u32 total) {
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

// the workgroup's run table -> the chunk's run list, bucket totals and the digits' own run lists
// (call after a barrier that follows the last run_table_add)
__device__ __forceinline__ void emit_run_table(i64 chunk, const RunSink &sink, const u32 *rd, const u32 *rc, const u32 *over) {
    const int t = threadIdx.x;
    if (*over) {
        if (t == 0) {
            sink.nruns[chunk] = 0;
            atomicExch(sink.overflow, 1u);
        }
        return;
    }
    // the used table slots, densely (RMAX == one wave; no order among the runs of a chunk is needed: a chunk's
    // digits are distinct and the runs of a digit are ordered by chunk afterwards)
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

// Digit-major counting of one chunk by a whole workgroup: every wave walks the DISTINCT digits of its
// entries (a handful on a pre-sorted stream); for each one, NITEMS ballots count its entries.  The
// workgroup's table (rd/rc/over in LDS, initialised and barrier'd by the caller) collects the waves'
// results.  pend: bit k set = item k of this lane is an entry.
template <int NITEMS>
__device__ __forceinline__ void count_runs(const u32 (&dig)[NITEMS], u32 pend, i64 chunk, const RunSink &sink, u32 *rd,
                                           u32 *rc, u32 *over) {
    const int lane = threadIdx.x & 63;
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
            if (lane == 0) run_table_add(rd, rc, over, c0, total);
            m = __ballot(dg[k] != EMPTY);
        }
    }
    __syncthreads();
    emit_run_table(chunk, sink, rd, rc, over);
}

// The same for a producer's COUNT pass (the append is the partition, see PartOut below): a lane holds a few
// (digit, weight) items instead of its entries -- a stencil node knows that it sends 8 entries to its own column
// block and 2 to each of three others without forming them.  wt == 0: no item.
template <int NITEMS>
__device__ __forceinline__ void count_runs_weighted(const u32 (&dig)[NITEMS], const u32 (&wt)[NITEMS], i64 chunk,
                                                    const RunSink &sink, u32 *rd, u32 *rc, u32 *over) {
    const int lane = threadIdx.x & 63;
    u32 dg[NITEMS];
#pragma unroll
    for (int k = 0; k < NITEMS; k++) dg[k] = wt[k] ? dig[k] : EMPTY;
    int trips = 0;
    bool stop = false;
#pragma unroll
    for (int k = 0; k < NITEMS; k++) {
        u64 m = stop ? 0ull : __ballot(dg[k] != EMPTY);
        while (m) {
            if (++trips > RMAX) {
                if (lane == 0) *over = 1;
                stop = true;
                break;
            }
            const int fl = __builtin_ctzll(m);
            const u32 c0 = (u32)__builtin_amdgcn_readlane((int)dg[k], fl);
            u32 part = 0;
#pragma unroll
            for (int q = k; q < NITEMS; q++) {
                const bool hit = dg[q] == c0;
                part += hit ? wt[q] : 0u;
                dg[q] = hit ? EMPTY : dg[q];
            }
            part = esp_wave_sum(part);
            if (lane == 0) run_table_add(rd, rc, over, c0, part);
            m = __ballot(dg[k] != EMPTY);
        }
    }
    __syncthreads();
    emit_run_table(chunk, sink, rd, rc, over);
}

template <bool MULTI, bool RAW = false>
static __global__ __launch_bounds__(THREADS) void run_hist_k(Args a, i64 first_chunk) {
    __shared__ u32 rd[RMAX];
    __shared__ u32 rc[RMAX];
    __shared__ u32 over;
    __shared__ u64 s_mw[MULTI ? MW_MAX : 1];
    const int t = threadIdx.x;
    if (MULTI && t < a.mw_P) s_mw[t] = a.mw_base[t];
    const i64 chunk = first_chunk + blockIdx.x;
    const i64 beg = chunk * TILE;
    const i64 end = min(a.E, beg + (i64)TILE);
    // a stream that is not pre-sorted is recognised by the first workgroups: the rest leave at once
    // (the flag is read while the keys are in flight)
    const u32 give_up = __hip_atomic_load(a.overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // (the launch behind the probe -- partition.hip -- waits for the flag: its workgroups request nothing when the probe gave up)
    if (first_chunk > 0 && give_up != 0u) return;
    if (t < RMAX) {
        rd[t] = EMPTY;
        rc[t] = 0;
    }
    if (t == 0) over = 0;
    u64 key[ITEMS];
    // (RAW, the caller's columns 16-byte aligned: two columns per load -- counting does not care which lane holds which entry;
    // item k of a lane is then entry 2 ((k >> 1) THREADS + t) + (k & 1) of the chunk)
    bool vec = false;
    if constexpr (RAW) vec = (reinterpret_cast<uintptr_t>(a.raw_cols) & 15) == 0;
    auto idx_of = [&](int k) -> i64 { return vec ? beg + 2 * ((i64)(k >> 1) * THREADS + t) + (k & 1) : beg + (i64)k * THREADS + t; };
    if constexpr (RAW) {  // (the digit is a function of the column: row 0 stands in; a column outside the matrix is reported)
        i64 cc[ITEMS];
        if (vec) {
            typedef long long ll2 __attribute__((ext_vector_type(2)));
            const ll2 *pc = reinterpret_cast<const ll2 *>(a.raw_cols + beg);  // (beg is a multiple of TILE)
#pragma unroll
            for (int k = 0; k < ITEMS; k += 2) {
                const i64 i0 = idx_of(k);
                ll2 v{1, 1};
                if (i0 + 1 < end) v = pc[(i64)(k >> 1) * THREADS + t];
                else if (i0 < end) v.x = a.raw_cols[i0];
                cc[k] = v.x, cc[k + 1] = v.y;
            }
        } else {
#pragma unroll
            for (int k = 0; k < ITEMS; k++) {
                const i64 idx = idx_of(k);
                cc[k] = idx < end ? a.raw_cols[idx] : 1;
            }
        }
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
            const i64 idx = idx_of(k);
            i64 c = cc[k];
            if (idx < end && !(1 <= c && c <= a.raw_n)) {
                atomicMin(a.raw_err, (unsigned long long)(idx + 1));
                c = 1;
            }
            key[k] = ((u64)(c - 1) << a.raw_rb) << ESP_TAG_BITS;
        }
    } else {
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
            const i64 idx = beg + k * THREADS + t;
            key[k] = idx < end ? a.keys_in[idx] : 0ull;
        }
    }
    if (give_up != 0u) return;
    __syncthreads();
    u32 dig[ITEMS];
    u32 pend = 0;  // bit k: item k of this lane not yet counted
    WaveWin ww{0, 0, 0};
    if constexpr (MULTI) ww = wave_window(a, s_mw, key[0]);
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
        const bool valid = idx_of(k) < end;
        dig[k] = valid ? digit_mw<MULTI>(a, s_mw, ww, key[k], true) : 0u;
        pend |= valid ? (1u << k) : 0u;
    }
    const RunSink sink{a.runs_d, a.runs_c, a.nruns, a.bucket_count, a.overflow, a.dcount, a.dlist};
    count_runs<ITEMS>(dig, pend, chunk, sink, rd, rc, &over);
}

// dense run list in tile order: sortable records (key = digit << 2, payload = tile|j|count)
static __global__ void run_pack_k(const u32 *__restrict__ runs_d, const u32 *__restrict__ runs_c,
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

static __global__ void run_counts_k(const double *__restrict__ lv, i64 R, u64 *__restrict__ cnt) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > R) return;
    cnt[i] = i < R ? ((u64)__double_as_longlong(lv[i]) & 0xFFFFull) : 0ull;
}

// head[d] = scanned count at the first record of digit d
static __global__ void run_heads_k(const u64 *__restrict__ lk, const u64 *__restrict__ sc, i64 R, u64 *__restrict__ head) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R) return;
    const u64 d = lk[i] >> ESP_TAG_BITS;
    if (i == 0 || (lk[i - 1] >> ESP_TAG_BITS) != d) head[d] = sc[i];
}

static __global__ void run_offsets_k(const u64 *__restrict__ lk, const double *__restrict__ lv, const u64 *__restrict__ sc,
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
static __global__ __launch_bounds__(THREADS) void run_coarse_k(const unsigned long long *__restrict__ bucket_count, i64 nb,
                                                         u64 *__restrict__ coarse) {
    __shared__ u64 lw[WAVES];
    const i64 d = (i64)blockIdx.x * THREADS + threadIdx.x;
    u64 tot;
    espscan::block_exclusive<u64, false>(d < nb ? (u64)bucket_count[d] : 0ull, lw, &tot);
    if (threadIdx.x == 0) coarse[blockIdx.x] = tot;
}

static __global__ __launch_bounds__(THREADS) void run_rank_k(const unsigned long long *__restrict__ bucket_count,
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

// Digit-major stable ranking of one wave's entries: for each distinct digit of the wave, its entries are
// numbered in (item, lane) order with a scalar running count -- no LDS counters inside the wave.  Ranked
// entries are blanked (dig: EMPTY = no entry; destroyed), so the hit test is one compare; digits first met at
// item k only look at items >= k.  hd/hj: the tile's map digit -> run index; cnt_w[j] = the wave's entries in run j.
template <int NI>
__device__ __forceinline__ void rank_in_runs(u32 (&dig)[NI], const u32 *hd, const u32 *hj, u32 *cnt_w, int lane,
                                             unsigned short (&rank)[NI], unsigned char (&jrun)[NI], u32 *miss = nullptr) {
    const u64 lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int k = 0; k < NI; k++) {
        u64 m = __ballot(dig[k] != EMPTY);
        while (m) {
            const int fl = __builtin_ctzll(m);
            const u32 c0 = (u32)__builtin_amdgcn_readlane((int)dig[k], fl);
            int jj = 0;
            {
                bool found = false;
                int j = (int)((c0 * 0x9E3779B1u) >> 26) & (RMAX - 1);
                for (int probe = 0; probe < RMAX; probe++) {
                    if (hd[j] == c0) {
                        jj = (int)hj[j];
                        found = true;
                        break;
                    }
                    j = (j + 1) & (RMAX - 1);
                }
                if (miss && !found && lane == 0) *miss = 1u;  // (VERIFY: a digit the tile's run list does not know)
            }
            u32 running = 0;
#pragma unroll
            for (int q = k; q < NI; q++) {
                const bool hit = dig[q] == c0;
                const u64 mm = __ballot(hit);
                if (hit) {
                    rank[q] = (unsigned short)(running + (u32)__popcll(mm & lt));
                    jrun[q] = (unsigned char)jj;
                }
                running += (u32)__popcll(mm);
                dig[q] = hit ? EMPTY : dig[q];
            }
            if (lane == 0) cnt_w[jj] = running;
            m = __ballot(dig[k] != EMPTY);
        }
    }
}

// (tried for RAW: four workgroups per CU through __launch_bounds__ -- 128 VGPRs instead of the K32 store loop's 152 -- costs
// 76 bytes of scratch per lane)
template <bool MULTI, bool K32, bool RAW = false, bool VERIFY = false>
static __global__ __launch_bounds__(THREADS) void run_scatter_k(Args a) {
    __shared__ u32 s_bad;
    __shared__ u64 s_mw[MULTI ? MW_MAX : 1];
    __shared__ u32 hd[RMAX];  // open-addressing map digit -> run index of this tile
    __shared__ u32 hj[RMAX];
    __shared__ i64 roff[RMAX];
    __shared__ u32 cnt[WAVES][RMAX];
    __shared__ int s_nr;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const i64 tile = blockIdx.x;  // = chunk index
    const i64 beg = tile * TILE;
    const i64 end = min(a.E, beg + (i64)TILE);
    // run table of the tile (nruns holds the exclusive scan by now: nr = difference)
    if (a.flags && (a.flags[0] | a.flags[1] | a.flags[3]) != 0u) return;  // (uniform: the flush takes another path)
    if (t == 0) s_nr = a.nruns_raw ? (int)a.nruns[tile] : (int)(a.nruns[tile + 1] - a.nruns[tile]);
    if (t < RMAX) {
        roff[t] = a.runs_off[tile * RMAX + t];
        hd[t] = EMPTY;
    }
    if (t < WAVES * RMAX) (&cnt[0][0])[t] = 0;
    if (MULTI && t < a.mw_P) s_mw[t] = a.mw_base[t];
    if (VERIFY && t == 0) s_bad = 0u;
    u64 key[ITEMS];
    double val[ITEMS];
    const i64 wbase = beg + (i64)w * (ESP_WAVE * ITEMS) + lane;
    if constexpr (RAW) {
        // (see Args: the packed key of (row, col) of the one kind; indices outside the matrix are reported.  Columns and
        // rows first, the values once the keys are formed: three arrays of 16 loads each in flight would cost the kernel
        // a wave per SIMD)
        {
            i64 cc[ITEMS];
#pragma unroll
            for (int k = 0; k < ITEMS; k++) {
                const i64 idx = wbase + k * ESP_WAVE;
                cc[k] = idx < end ? a.raw_cols[idx] : 1;
            }
#pragma unroll
            for (int k = 0; k < ITEMS; k++) {
                const i64 idx = wbase + k * ESP_WAVE;
                i64 r = idx < end ? a.raw_rows[idx] : 1;
                if (!(1 <= cc[k] && cc[k] <= a.raw_n) || !(1 <= r && r <= a.raw_m)) {
                    atomicMin(a.raw_err, (unsigned long long)(idx + 1));
                    cc[k] = 1, r = 1;
                }
                key[k] = ((((u64)(cc[k] - 1) << a.raw_rb) | (u64)(r - 1)) << ESP_TAG_BITS) | (u64)a.raw_kind;
            }
        }
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
            const i64 idx = wbase + k * ESP_WAVE;
            const double v = idx < end ? a.vals_in[idx] : 0.0;
            val[k] = a.raw_negate ? -v : v;
        }
    } else {
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
            const i64 idx = wbase + k * ESP_WAVE;
            key[k] = idx < end ? a.keys_in[idx] : 0ull;
        }
        // (vals_in == nullptr: the records are single words -- the item records of an element batch, femitems.hpp / elements.hpp --
        // no value array is read or written)
        if (a.vals_in) {
#pragma unroll
            for (int k = 0; k < ITEMS; k++) {
                const i64 idx = wbase + k * ESP_WAVE;
                val[k] = idx < end ? a.vals_in[idx] : 0.0;
            }
        }
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
#pragma unroll
    for (int k = 0; k < ITEMS; k++) dig[k] = ((pend >> k) & 1u) ? dig[k] : EMPTY;
    rank_in_runs<ITEMS>(dig, hd, hj, cnt[w], lane, rank, jrun, VERIFY ? &s_bad : nullptr);
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
        if constexpr (VERIFY) {  // (the run list is the previous assembly's: every run must hold what it says)
            if (t < nr && run != a.runs_c[tile * RMAX + t]) s_bad = 1u;
        }
    }
    __syncthreads();
    if constexpr (VERIFY) {
        if (s_bad) {  // (uniform) not the stream the run lists were made for: nothing of this tile is stored
            if (t == 0) *a.verify_err = 1u;
            return;
        }
    }
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
            if (RAW || a.vals_in) a.vals_out[dst] = val[k];
        }
    }
}

// ---- the append IS the partition (device-side producers) ------------------------------------------------
// A producer that can count before it writes -- the stencil and FEM generators, a device-side triplet array --
// needs no partition pass at all: a COUNT launch of the producer (ALU only: no entry is formed or stored) fills
// the run lists of its chunks (count_runs_weighted -> RunSink), run_coarse_k + run_rank_k turn them into bucket
// starts and run offsets exactly as for run_scatter_k, and the WRITE launch stages a chunk's entries in LDS in
// stream order (as the plain producer does for its coalesced copy-out) and stores every entry straight to
// `run offset + stable rank inside the run` -- run_scatter_k's stores without its loads.  Chunk = workgroup
// of the producer, the same in both launches.  The pending buffer then holds the entries bucket by bucket
// (a stable permutation: entries of one (col,row) keep their append order), the flush starts at the bucket kernel.
struct PartOut {
    const u32 *runs_d;     // [chunk][RMAX]
    const i64 *runs_off;   // [chunk][RMAX]
    const u64 *nruns;      // [chunk] (raw counts)
    const u32 *flags;      // the write launch leaves at once when flags[0] (window), [1] (too many digits in a chunk)
                           // or [3] (too many runs of a digit) is set: the host then runs the plain producer
    const unsigned long long *maxlen;  // longest bucket (run_rank_k)
    i64 cap;               // a K32 launch leaves without a store when *maxlen > cap (the host applies the same rule)
    int k32;               // the batch has one known kind and <= 32 key bits lie below the prefix: 4-byte keys (the
                           // bits below the prefix) go out
    int s32;               // <= 32 key bits lie below the prefix: the staging area holds 4 bytes per key
    int shift;             // digit = ((key >> 2) - base) >> shift
    u64 base, span;
    u64 *keys_out;
    double *vals_out;
    i64 chunk_base;        // chunk index of the launch's first workgroup
    // several key windows side by side (column shards: window r = owner r's column range, see Args::mw_*): global digit
    // = r * mw_nb + digit inside window r; mw_P == 0: the one window [base, base + span)
    int mw_P;
    u32 mw_nb;
    const u64 *mw_base;    // first key of every window, ascending
    u64 mw_own_base, mw_own_width;  // ... and the own window's first key / key count (arguments: no load in the common case)
    // a shard's OWN window goes out as 4-byte keys (the bits below the bucket prefix, one kind for the batch), the
    // ranges it sends to other ranks as packed keys: own32 != 0, window mw_me; the 4-byte keys of bucket position p of
    // the own range [*own_lo, ...) live at ((u32 *)(keys_out + *own_lo))[p - *own_lo], inside the range's own bytes
    int own32;
    int mw_me;
    const i64 *own_lo;     // bucket start of the own window's first digit (written by run_rank_k)
};

// LDS tables of one producer workgroup (tile = chunk)
// 4-byte keys of a shard's own range: the key of batch position p sits at own_keys32(keys, own_lo)[p] -- inside the
// range's own bytes (8 per entry, 4 used), one slot further when own_lo is odd, so that an even p is 8-byte aligned and
// pairs of keys go out aligned together with pairs of values
__device__ __forceinline__ u32 *own_keys32(u64 *keys, i64 own_lo) { return reinterpret_cast<u32 *>(keys + own_lo) + (own_lo & 1) - own_lo; }
__device__ __forceinline__ const u32 *own_keys32(const u64 *keys, i64 own_lo) {
    return reinterpret_cast<const u32 *>(keys + own_lo) + (own_lo & 1) - own_lo;
}

template <int NWAVES>
struct TileLds {
    u32 cnt[NWAVES][RMAX];  // entries of wave w in run j of the tile
    i64 roff[RMAX];         // global offset of run j (run_rank_k)
    u32 lstart[RMAX + 1];   // first LDS slot of run j: the tile is staged run by run
    u64 rbase[RMAX];        // first key of run j's bucket (window base + digit << shift)
    u32 rown[RMAX];         // own32: run j lies in the shard's own window
    i64 own_lo;
    u32 all_own;            // own32: every run of the tile does (the usual tile of a slab-wise assembly)
};

// Where a thread's entries go inside the tile's LDS staging area.  A thread (a stencil node, a FEM cell) holds NQ
// items (digit, count): its entries for one column each, in stream order inside the item.  The staging area is laid
// out run by run (a run = the tile's entries of one digit), inside a run thread by thread, inside a thread item by
// item.  That is a stable order for the fold: entries of one (row,col) lie in ONE item of a thread and keep their
// order there, and of two threads the earlier one's come first -- entries of different columns may trade places
// freely (the bucket kernel sorts by column anyway).  slot[q] = first LDS slot of item q.  Two barriers.
// Returns false (uniform) when the launch has to leave without a store (see PartOut::flags).
// Everything a tile needs from global memory, requested as the FIRST thing its kernel does, in one round trip (a
// workgroup lives for a few microseconds: dependent loads in the middle of it would be a third of that).
struct TileLoads {
    u32 stop;                    // a flag is set: leave
    unsigned long long longest;  // longest bucket
    int nr;                      // runs of the tile
    u32 run_digit;               // lane j: digit of run j (garbage for j >= nr)
    i64 run_off;                 // lane j: global offset of run j
    i64 own_lo;                  // own32: first position of the shard's own range
};
__device__ __forceinline__ TileLoads tile_loads(const PartOut &p, i64 chunk) {
    const int lane = threadIdx.x & 63;
    TileLoads L;
    L.stop = p.flags[0] | p.flags[1] | p.flags[3];
    L.longest = *p.maxlen;
    L.nr = (int)p.nruns[chunk];
    L.run_digit = p.runs_d[chunk * RMAX + lane];
    L.run_off = p.runs_off[chunk * RMAX + lane];
    L.own_lo = p.own32 ? *p.own_lo : 0;
    return L;
}

// true: the launch leaves without a store (uniform) -- a flag is set, or it would store 4-byte keys (out32) and the
// longest bucket does not fit the bucket kernel.  Checked before the tile's entries are formed.
__device__ __forceinline__ bool tile_stop(const PartOut &p, const TileLoads &L, bool out32) {
    return L.stop != 0u || (out32 && L.longest > (unsigned long long)p.cap);
}

// The run table of a tile as every wave holds it in registers: lane j = run j (what TileLds holds in LDS: the copy-out loop
// walks the runs one after the other, and a table read from LDS per run -- four dependent round trips -- is a fifth of a
// finely cut tile's life; a readlane is not)
struct TileRegs {
    u32 lst;    // first LDS slot of run `lane` (lane >= nr: the tile's total)
    i64 roff;   // global offset of run `lane`
    u64 rbase;  // first key of run `lane`'s bucket
    u32 rown;   // own32: run `lane` lies in the shard's own window
};
// (out32: the launch stores 4-byte keys and leaves when the longest bucket does not fit the bucket kernel)
template <int NQ, int NWAVES>
__device__ __forceinline__ bool tile_slots(const PartOut &p, const TileLoads &L, const u32 (&dig)[NQ], const u32 (&wt)[NQ],
                                           u32 (&slot)[NQ], TileLds<NWAVES> &S, int *total_out, bool out32, TileRegs *regs = nullptr) {
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (L.stop != 0u || (out32 && L.longest > (unsigned long long)p.cap)) return false;
    const int nr = L.nr;
    const u32 my_run_digit = lane < nr ? L.run_digit : EMPTY;  // lane j holds the digit of run j
    const i64 run_off_raw = L.run_off;
    S.cnt[w][lane] = 0;
    u32 dg[NQ], pq[NQ], jq[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        dg[q] = wt[q] ? dig[q] : EMPTY;
        pq[q] = 0;
        jq[q] = 0;
    }
    int trips = 0;
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        u64 m = __ballot(dg[q] != EMPTY);
        while (m && ++trips <= RMAX) {
            const u32 c0 = (u32)__builtin_amdgcn_readlane((int)dg[q], __builtin_ctzll(m));
            const u64 hitrun = __ballot(my_run_digit == c0);
            const u32 j = hitrun ? (u32)__builtin_ctzll(hitrun) : 0u;  // (the COUNT launch listed every digit of the tile)
            u32 mine = 0;
#pragma unroll
            for (int r = q; r < NQ; r++) mine += dg[r] == c0 ? wt[r] : 0u;
            const u32 inc = esp_wave_scan_add(mine);
            u32 acc = inc - mine;
#pragma unroll
            for (int r = q; r < NQ; r++) {
                const bool hit = dg[r] == c0;
                pq[r] = hit ? acc : pq[r];
                jq[r] = hit ? j : jq[r];
                acc += hit ? wt[r] : 0u;
                dg[r] = hit ? EMPTY : dg[r];
            }
            if (lane == 63) S.cnt[w][j] = inc;  // the wave's total of this digit
            m = __ballot(dg[q] != EMPTY);
        }
    }
    __syncthreads();
    // every wave: lane j adds up run j over the waves; run starts = exclusive scan over the runs
    u32 tot = 0, before = 0;
#pragma unroll
    for (int i = 0; i < NWAVES; i++) {
        const u32 x = S.cnt[i][lane];
        before += i < w ? x : 0u;
        tot += x;
    }
    const u32 inc = esp_wave_scan_add(tot);
    const u32 lst = inc - tot;
    const u32 sb = lst + before;  // first slot of this wave's part of run `lane`
    const int total = __builtin_amdgcn_readlane((int)inc, 63);
    u64 rb = 0;
    u32 own = 0;
    if (lane < nr && (w == 0 || regs)) {
        if (p.mw_P) {
            const u32 r = my_run_digit / p.mw_nb;
            rb = ((int)r == p.mw_me ? p.mw_own_base : p.mw_base[r]) + ((u64)(my_run_digit - r * p.mw_nb) << p.shift);
            own = (p.own32 && (int)r == p.mw_me) ? 1u : 0u;
        } else {
            rb = p.base + ((u64)my_run_digit << p.shift);
        }
    }
    if (regs) *regs = TileRegs{lst, lane < nr ? run_off_raw : 0, rb, own};
    if (w == 0) {
        S.lstart[lane] = lst;
        if (lane == 63) S.lstart[RMAX] = inc;
        S.roff[lane] = lane < nr ? run_off_raw : 0;
        S.rbase[lane] = rb;
        S.rown[lane] = own;
        const u64 foreign = __ballot(lane < nr && !own);
        if (lane == 0) {
            S.own_lo = L.own_lo;
            S.all_own = (p.own32 && foreign == 0ull) ? 1u : 0u;
        }
    }
#pragma unroll
    for (int q = 0; q < NQ; q++) slot[q] = (u32)__shfl((int)sb, (int)jq[q], ESP_WAVE) + pq[q];
    *total_out = total;
    return true;
}

// staged tile (run by run, see tile_slots) -> its runs' places in the buffer; consecutive threads store consecutive
// entries of a run.  A barrier lies between the last staging store and this call.
// KT = u32: the staging area holds the LOW 32 bits of every (col,row) key -- the keys of a bucket span less than 2^32
// (shift <= 32), so key - first key of the bucket = (low bits - low bits of that first key) mod 2^32: OUT32 stores that
// difference (the 4-byte key the bucket kernel reads), else the packed key is put together again from it.
// KT = u64: packed keys staged and stored as they are.
template <typename KT, bool OUT32, int NT, int NWAVES>
__device__ __forceinline__ void copy_out_runs(const PartOut &p, const KT *lk, const double *lv, int total, const TileLds<NWAVES> &S,
                                              u32 kind, int lo = 0, int hi = 0x7FFFFFFF, u32 *k4_all = nullptr, const TileRegs *R = nullptr) {
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    const int t = threadIdx.x;
    // run by run (a handful per tile, hundreds of entries each): two entries per store instruction -- 8-byte / 16-byte
    // key stores and 16-byte value stores on the aligned body of the run, the odd entry at either end on its own
    for (int j = 0; j < RMAX; j++) {
        // (staging window [lo, hi) of the tile's slots: a tile that does not fit the staging area goes out in rounds;
        // slot q of the window sits at lk[q - lo])
        auto lane_u32 = [&](u32 x, int l) { return (u32)__builtin_amdgcn_readlane((int)x, l); };
        auto lane_u64 = [&](u64 x, int l) { return ((u64)lane_u32((u32)(x >> 32), l) << 32) | (u64)lane_u32((u32)x, l); };
        const int b0 = R ? (int)lane_u32(R->lst, j) : (int)S.lstart[j];
        const int e0 = R ? (j + 1 < ESP_WAVE ? (int)lane_u32(R->lst, j + 1) : total) : (int)S.lstart[j + 1];
        if (b0 >= total || b0 >= hi) break;
        const int bw = max(b0, lo), ew = min(e0, hi);
        const int len = ew - bw;
        if (len <= 0) continue;
        const int b = bw - lo;
        const i64 ro = (R ? (i64)lane_u64((u64)R->roff, j) : S.roff[j]) + (i64)(bw - b0);
        const u64 rb = R ? lane_u64(R->rbase, j) : S.rbase[j];
        // 4-byte keys for this run: every run (OUT32), or the runs of a shard's own window (own32; KT = u32 then)
        // (k4_all: OUT32 for a tile whose runs all lie in the shard's own window -- the keys go to own_keys32)
        bool run32 = OUT32;
        u32 *k4 = OUT32 && k4_all ? k4_all : reinterpret_cast<u32 *>(p.keys_out);
        const i64 k4off = 0;  // index of the 4-byte key of bucket position q: q - k4off
        if constexpr (sizeof(KT) == 4 && !OUT32) {
            if (R ? lane_u32(R->rown, j) != 0u : S.rown[j] != 0u) {
                run32 = true;
                k4 = own_keys32(p.keys_out, S.own_lo);
            }
        }
        auto key_of = [&](int q) -> u64 {  // what goes out for staged entry q: the 4-byte key, or the packed key
            if constexpr (sizeof(KT) == 4) {
                const u32 delta = (u32)lk[q] - (u32)rb;
                if (run32) return (u64)delta;
                return ((rb + (u64)delta) << ESP_TAG_BITS) | (u64)kind;
            } else {
                return (u64)lk[q];
            }
        };
        auto store1 = [&](int q, i64 dst) {
            if (run32)
                k4[dst - k4off] = (u32)key_of(q);
            else
                p.keys_out[dst] = key_of(q);
            p.vals_out[dst] = lv[q];
        };
        const int head = (int)(ro & 1);  // the run starts at an odd place: its first entry goes alone
        const int npair = (len - head) >> 1;
        // A run shorter than one round of the workgroup goes out through ONE wave (runs in turn: wave j mod NWAVES), so that the
        // short runs of a finely cut tile (segments of 64 columns: a dozen runs of 64 .. 256 pairs) leave side by side instead
        // of one after the other with three quarters of the threads idle.
        static_assert((NWAVES & (NWAVES - 1)) == 0, "waves of a producer workgroup: a power of two");
        const bool wide = npair >= NT;
        if (!wide && (j & (NWAVES - 1)) != (t >> 6)) continue;
        const int tq = wide ? t : (t & 63), tstep = wide ? NT : ESP_WAVE;
        if (tq == 0 && head) store1(b, ro);
        // (4-byte keys of the own window: the pair's key store is 8-byte aligned when its index is even)
        const bool pair32 = run32 && (((ro + head - k4off) & 1) == 0);
        for (int q = tq; q < npair; q += tstep) {
            const int e = b + head + 2 * q;
            const i64 dst = ro + head + 2 * q;  // even
            if (run32) {
                if (pair32) {
                    *reinterpret_cast<u32x2 *>(k4 + (dst - k4off)) = u32x2{(u32)key_of(e), (u32)key_of(e + 1)};
                } else {
                    k4[dst - k4off] = (u32)key_of(e);
                    k4[dst - k4off + 1] = (u32)key_of(e + 1);
                }
            } else {
                *reinterpret_cast<u64x2 *>(p.keys_out + dst) = u64x2{key_of(e), key_of(e + 1)};
            }
            *reinterpret_cast<f64x2 *>(p.vals_out + dst) = f64x2{lv[e], lv[e + 1]};
        }
        if (tq == tstep - 1 && ((len - head) & 1)) store1(b + len - 1, ro + len - 1);
    }
}

// digit of a column for the producers' launches (the digit must not reach into the row bits: shift >= rb); columns
// outside the key window raise *err.  Several windows: the window by binary search over their first keys (they cover
// every column), then the digit inside it.
__device__ __forceinline__ u32 column_digit(const PartOut &p, i64 col0, int rb, u32 *err) {
    const u64 key = (u64)col0 << rb;
    if (p.mw_P) {
        // (a rank mostly produces entries of its own column range: that window is tried first)
        const u64 rel_own = key - p.mw_own_base;
        if (rel_own < p.mw_own_width) {
            u64 dl = rel_own >> p.shift;
            dl = dl < (u64)p.mw_nb ? dl : (u64)p.mw_nb - 1;
            return (u32)p.mw_me * p.mw_nb + (u32)dl;
        }
        int r = 0;
#pragma unroll
        for (int step = MW_MAX / 2; step; step >>= 1) {
            const int c = r + step;
            if (c < p.mw_P && key >= p.mw_base[c]) r = c;
        }
        u64 dl = (key - p.mw_base[r]) >> p.shift;
        dl = dl < (u64)p.mw_nb ? dl : (u64)p.mw_nb - 1;
        return (u32)r * p.mw_nb + (u32)dl;
    }
    u64 kn = key - p.base;
    if (kn >= p.span) {
        if (err) *err = 1u;
        kn = 0;
    }
    return (u32)(kn >> p.shift);
}

// 4-byte keys of a bucket-ordered pending buffer back to packed keys (any call that reads or extends the pending
// entries other than the flush they were written for): one workgroup per bucket
// (buckets in a grid-stride loop: the flush's own passes may leave 2^24 segments, more than a launch has workgroups of this size --
// expand_keys_grid)
static __global__ __launch_bounds__(THREADS) void expand_keys_k(const u32 *__restrict__ k32, const i64 *__restrict__ seg_start, int shift,
                                                         u64 base, u32 kind, u64 *__restrict__ out, i64 nseg) {
    for (i64 s = blockIdx.x; s < nseg; s += gridDim.x) {
        const i64 b = seg_start[s], e = seg_start[s + 1];
        const u64 hi = ((u64)s << shift) + base;
        for (i64 i = b + threadIdx.x; i < e; i += THREADS) out[i] = ((hi + (u64)k32[i]) << ESP_TAG_BITS) | (u64)kind;
    }
}
static inline unsigned expand_keys_grid(i64 nseg) { return (unsigned)(nseg < ((i64)1 << 20) ? nseg : ((i64)1 << 20)); }

// the same for a shard's own window (digits [d0, d0 + nb) of the multi-window partition): 4-byte keys at
// own_keys32(keys, own_lo)[p] -> packed keys at out[p]
static __global__ __launch_bounds__(THREADS) void expand_own_keys_k(const u64 *__restrict__ keys, const i64 *__restrict__ seg_start, i64 d0,
                                                             int shift, u64 base, u32 kind, u64 *__restrict__ out) {
    const i64 s = blockIdx.x;
    const i64 own_lo = seg_start[d0];
    const i64 b = seg_start[d0 + s], e = seg_start[d0 + s + 1];
    const u32 *k32 = own_keys32(keys, own_lo);
    const u64 hi = ((u64)s << shift) + base;
    for (i64 i = b + threadIdx.x; i < e; i += THREADS) out[i] = ((hi + (u64)k32[i]) << ESP_TAG_BITS) | (u64)kind;
}

}  // namespace esprun
