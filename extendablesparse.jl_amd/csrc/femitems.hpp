// femitems.hpp -- shuffled P1 FEM streams (BASELINE config 4: random cell order): the producer partitions ITEMS, not
// updates, and stores every update once, at its final bucket position.
//
// A stream in random cell order defeats the run-based producer partition (runpart.hpp: a chunk of 128 cells touches
// ~500 column blocks, not a handful), and so far such a producer wrote its updates in stream order and the flush
// paid three histogram + scatter passes over them: 16 B written, then 3 x (8 B + 16 B read, 16 B written) per update.
// But a cell sends its (dim+1)(dim+2) updates to only dim+1 columns -- dim+2 updates each, one ITEM -- and the bucket
// of an update is a function of its column alone.  So:
//   fem_items_k  : one 16-byte record per item, in stream order: the column as a packed sort key | the cell's number
//                  and the vertex number (no update is formed: ALU + 16 B per item = 3.2 B per update)
//   partition    : the flush's own stable passes (radix.hpp) over the ITEM records down to the bucket kernel's segments --
//                  a fifth (3-D) or a quarter (2-D) of the records, each 16 B: what cost 120 B per update costs 24
//   fem_expand_k : sorted item g -> its dim+2 updates at entries [g (dim+2), (g+1)(dim+2)) of the append buffer, staged
//                  through LDS and stored with full-line coalesced stores; 4-byte keys (the bits below the segment
//                  prefix) when they fit: the buffer is written ONCE, bucket by bucket, 12 B per update
// The buffer is a stable permutation of the stream (inside a segment: items in stream order, an item's updates in call
// order -- all a (row,col) ever sees of its updates is their relative order), the segment table follows from the
// items' (x dim+2), and esp_flush starts at the bucket kernel: "the append is the partition" (esp_handle::PrePart).
// Price: the element matrix of a cell is computed once per vertex column (dim+1 times) instead of once.
#pragma once
#include "common.hpp"
#include "generators.hpp"
#include "segexpand.hpp"

namespace espitem {

constexpr int THREADS = 256;
constexpr int MAX_W = 5;  // 3-D: 5 updates per item

struct Args {
    espgen::FemArgs fem;
    i64 nitems;           // ncells * (dim + 1)
    u64 *ikeys;           // item records: packed key of (row 0, the item's column), kind bits zero
    double *ivals;        // ... | bit pattern of (cell << 2 | vertex)
    int single;           // single-word records: the cell's number sits in the key's row and kind bits (it fits when
                          // ncells <= 2^(rb+2)); no second array, the passes move 8 bytes per item (Pass::keys_only), and
                          // the expansion finds the vertex by comparing the column with the cell's nodes
    const u64 *sorted_keys;
    // fem_expand_k
    const double *sorted;  // the partitioned records' second halves
    int rem_bits;          // K32: key bits below the segment prefix
    u64 base;              // key window base
    u64 *keys_out;         // (K32: u32 keys)
    double *vals_out;
};

static __global__ __launch_bounds__(THREADS) void fem_items_k(Args a) {
    const i64 pos = (i64)blockIdx.x * THREADS + threadIdx.x;
    if (pos >= a.fem.ncells) return;
    i64 vx[4][3];
    i64 nodes[4];
    const i64 cell = (i64)espgen::fem_cell_at(a.fem, pos);
    espgen::fem_vertices(a.fem, cell, vx, nodes);
    const int ni = a.fem.dim + 1;
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (k < ni) {
            if (a.single) {
                a.ikeys[pos * ni + k] = esp_pack(a.fem.L, 1, nodes[k], 0) | (u64)cell;
            } else {
                a.ikeys[pos * ni + k] = esp_pack(a.fem.L, 1, nodes[k], 0);
                a.ivals[pos * ni + k] = __longlong_as_double((long long)(((u64)cell << 2) | (u64)k));  // (the cell, not its stream position: the expansion need not walk the permutation again)
            }
        }
}

// The same for a cell order whose walk is long (the host picks: 2^bits > 1.5 ncells).
// The cell of a stream position is a Feistel permutation of [0, 2^bits) walked until it lands below ncells (fem_cell_at).  The
// 2-D mesh of config 4 -- 2.0 10^7 cells in a domain of 2^26 -- walks 3.4 steps on average, and a wave that walks lane by lane
// waits for its slowest lane: 12 steps, 0.75 ms for the kernel.  So a wave walks ITEM_CELLS x 64 positions TOGETHER: after the
// first step the positions still outside wait in a queue (LDS, the wave's own: no workgroup barrier), which the wave takes 64 at a
// time, round after round, survivors packed to the queue's front -- the work follows the average, not the maximum.
constexpr int ITEM_CELLS = 8;                          // stream positions per lane
constexpr int ITEM_WAVE_CELLS = ITEM_CELLS * ESP_WAVE;  // ... per wave
static __global__ __launch_bounds__(THREADS) void fem_items_walk_k(Args a) {
    constexpr int WAVES = THREADS / ESP_WAVE;
    __shared__ u64 queue[WAVES][ITEM_WAVE_CELLS];  // x | owner << 48 (a cell number has at most 40 bits, its domain 42)
    __shared__ u64 cellof[WAVES][ITEM_WAVE_CELLS];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const u64 lt = (1ull << lane) - 1ull;
    const i64 wbase = ((i64)blockIdx.x * WAVES + w) * ITEM_WAVE_CELLS;
    const u64 nc = (u64)a.fem.ncells;
    const bool walk = a.fem.order_mode != 0 && a.fem.ncells >= 2;
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < ITEM_CELLS; i++) {
        const i64 pos = wbase + i * ESP_WAVE + lane;
        const bool go = walk && pos < a.fem.ncells;
        const u64 y = go ? espgen::fem_feistel(a.fem, (u64)pos) : (u64)pos;
        const bool out = go && y >= nc;
        if (!out) cellof[w][i * ESP_WAVE + lane] = y;
        const u64 bal = __ballot(out);
        if (out) queue[w][cnt + (int)__popcll(bal & lt)] = y | ((u64)(i * ESP_WAVE + lane) << 48);
        cnt += (int)__popcll(bal);
    }
    while (cnt > 0) {  // (wave-uniform)
        int kept = 0;
        for (int g = 0; g * ESP_WAVE < cnt; g++) {
            const int idx = g * ESP_WAVE + lane;
            const bool has = idx < cnt;
            __builtin_amdgcn_wave_barrier();
            const u64 e = has ? queue[w][idx] : 0ull;  // (read before this round's survivors are packed in front of it: kept <= g 64)
            __builtin_amdgcn_wave_barrier();
            const u64 own = e >> 48;
            const u64 y = has ? espgen::fem_feistel(a.fem, e & ((1ull << 48) - 1ull)) : 0ull;
            const bool out = has && y >= nc;
            if (has && !out) cellof[w][own] = y;
            const u64 bal = __ballot(out);
            if (out) queue[w][kept + (int)__popcll(bal & lt)] = y | (own << 48);
            kept += (int)__popcll(bal);
        }
        cnt = kept;
    }
    __builtin_amdgcn_wave_barrier();
    const int ni = a.fem.dim + 1;
#pragma unroll 1
    for (int i = 0; i < ITEM_CELLS; i++) {
        const i64 pos = wbase + i * ESP_WAVE + lane;
        if (pos >= a.fem.ncells) continue;
        i64 vx[4][3];
        i64 nodes[4];
        const i64 cell = (i64)cellof[w][i * ESP_WAVE + lane];
        espgen::fem_vertices(a.fem, cell, vx, nodes);
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (k < ni) {
                if (a.single) {
                    a.ikeys[pos * ni + k] = esp_pack(a.fem.L, 1, nodes[k], 0) | (u64)cell;
                } else {
                    a.ikeys[pos * ni + k] = esp_pack(a.fem.L, 1, nodes[k], 0);
                    a.ivals[pos * ni + k] = __longlong_as_double((long long)(((u64)cell << 2) | (u64)k));  // (the cell, not its stream position: the expansion need not walk the permutation again)
                }
            }
    }
}

// K32: 4-byte keys (the bits below the segment prefix; every entry is a RAWUPDATE), else packed keys
template <bool K32>
static __global__ __launch_bounds__(THREADS) void fem_expand_k(Args a) {
    typedef typename std::conditional<K32, u32, u64>::type KT;
    __shared__ KT lk[THREADS * MAX_W];
    __shared__ double lv[THREADS * MAX_W];
    const int t = threadIdx.x;
    const i64 g0 = (i64)blockIdx.x * THREADS, g = g0 + t;
    const int W = a.fem.dim + 2;
    if (g < a.nitems) {
        // the item: a cell and one of its vertex columns (single-word record: both in the key)
        i64 cell, icol;
        if (a.single) {
            const u64 rec = a.sorted_keys[g];
            const int low = a.fem.L.rb + ESP_TAG_BITS;
            cell = (i64)(rec & ((1ull << low) - 1ull));
            icol = (i64)(rec >> low) + 1;
        } else {
            cell = (i64)((u64)__double_as_longlong(a.sorted[g]) >> 2);
            icol = (i64)((a.sorted_keys[g] >> ESP_TAG_BITS) >> a.fem.L.rb) + 1;
        }
        const u64 lowmask = a.rem_bits >= 64 ? ~0ull : ((1ull << a.rem_bits) - 1ull);
        // the item's updates in call order: row il's term at il, +1 from the diagonal's row on (the mass term of the
        // diagonal comes right before it)
        espgen::fem_column_of_cell(a.fem, cell, icol, [&](int il, int jl, i64 row, double v) {
            const int at = t * W + (jl < 0 ? il : il + (il >= jl ? 1 : 0));
            if constexpr (K32)
                lk[at] = (u32)(((((u64)(icol - 1) << a.fem.L.rb) | (u64)(row - 1)) - a.base) & lowmask);
            else
                lk[at] = esp_pack(a.fem.L, row, icol, ESP_RAWUPDATE);
            lv[at] = v;
        });
    }
    __syncthreads();
    const int cnt = (int)min((i64)THREADS, a.nitems - g0) * W;
    if constexpr (K32) {
        // (whole lines, 16 bytes per lane: g0 * W is a multiple of 256 and both arrays start 256-byte aligned)
        typedef u32 u32x4 __attribute__((ext_vector_type(4)));
        typedef double dbl2 __attribute__((ext_vector_type(2)));
        u32 *gk = reinterpret_cast<u32 *>(a.keys_out) + g0 * W;
        const int kquad = cnt >> 2;
        for (int q = t; q < kquad; q += THREADS) reinterpret_cast<u32x4 *>(gk)[q] = u32x4{lk[4 * q], lk[4 * q + 1], lk[4 * q + 2], lk[4 * q + 3]};
        for (int q = 4 * kquad + t; q < cnt; q += THREADS) gk[q] = lk[q];
        double *gv = a.vals_out + g0 * W;
        const int vpair = cnt >> 1;
        for (int q = t; q < vpair; q += THREADS) reinterpret_cast<dbl2 *>(gv)[q] = dbl2{lv[2 * q], lv[2 * q + 1]};
        if (t == 0 && (cnt & 1)) gv[cnt - 1] = lv[cnt - 1];
    } else {
        espgen::copy_out_staged<THREADS>(lk, lv, cnt, a.keys_out + g0 * W, a.vals_out + g0 * W);
    }
}

// The expansion with the partition's last bits done inside it (segexpand.hpp): one workgroup per segment of up to 4096
// single-word item records orders them by the next lbits bits and expands them in that order; writes the sub-segment table.
template <bool K32>
static __global__ __launch_bounds__(espseg::THREADS) void fem_seg_expand_k(Args a, espseg::SegArgs sa) {
    typedef typename std::conditional<K32, u32, u64>::type KT;
    __shared__ espseg::SegLds L;
    __shared__ KT lk[espseg::THREADS * MAX_W];
    __shared__ double lv[espseg::THREADS * MAX_W];
    const int low = a.fem.L.rb + ESP_TAG_BITS;
    const u64 lowmask = a.rem_bits >= 64 ? ~0ull : ((1ull << a.rem_bits) - 1ull);
    espseg::segment_expand<KT>(sa, L, lk, lv, reinterpret_cast<KT *>(a.keys_out), a.vals_out, [&](u64 rec, KT *k, double *v) {
        const i64 cell = (i64)(rec & ((1ull << low) - 1ull));
        const i64 icol = (i64)(rec >> low) + 1;
        espgen::fem_column_of_cell(a.fem, cell, icol, [&](int il, int jl, i64 row, double val) {
            const int at = jl < 0 ? il : il + (il >= jl ? 1 : 0);
            if constexpr (K32)
                k[at] = (u32)(((((u64)(icol - 1) << a.fem.L.rb) | (u64)(row - 1)) - a.base) & lowmask);
            else
                k[at] = esp_pack(a.fem.L, row, icol, ESP_RAWUPDATE);
            v[at] = val;
        });
    });
}

// entries = items * (dim + 2): the segment table of the append buffer from the items'
static __global__ void scale_segments_k(const i64 *__restrict__ in, i64 n, i64 w, i64 *__restrict__ out) {
    const i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (g < n) out[g] = in[g] * w;
}

}  // namespace espitem
