// elements.hpp -- element-level append (esp_append_elements): the inner loops of a finite-element assembly
// (test/femtools.jl:62-69) for element data the CALLER holds in arrays -- connectivity, element matrices, the optional
// term on the diagonal -- with the item partition of femitems.hpp as its fast path.
//
//   for icell = 1:ncells, il = 1:nloc                      (femtools.jl:61-63)
//       i = cellnodes[il, icell]
//       update(A, diag[il, icell], i, i)                   (femtools.jl:64; only when the caller passes diag)
//       for jl = 1:nloc: update(A, elmat[il, jl, icell], i, cellnodes[jl, icell])    (femtools.jl:65-68)
//
// A cell sends its nloc * W updates (W = nloc + 1 with a diagonal term, else nloc) to only nloc columns, W each -- one
// ITEM per (cell, local column jl) -- and the bucket of an update is a function of its column.  So, on an empty buffer:
//   elem_items_k   : one 8-byte record per item, in stream order: the column in the key's column bits | the item's number
//                    p = cell * nloc + jl below them (a VIRTUAL key layout: as many low bits as the item numbers need --
//                    the partition only ever looks at column bits); reads the connectivity once (8 B per item), checks
//                    the node numbers (BoundsError) and that no cell names a node twice
//   partition      : the flush's own stable passes (radix.hpp, keys only) over the records down to the bucket kernel's segments
//   elem_expand_k  : sorted item g -> its W updates at entries [g W, (g+1) W) of the append buffer: the rows are the
//                    cell's nodes, the values column jl of the cell's element matrix (contiguous in Julia's layout:
//                    nloc * 8 B) and diag[jl] -- gathered; every address follows from the record alone, one round trip --
//                    staged through LDS, stored as whole lines, 4-byte keys when they fit.  The gathers are what the
//                    kernel is bound by (a 64-byte memory-side request each, at random addresses), so for cells of 3 or 4
//                    nodes the first kernel (elem_cells_k) leaves rows and diagonal terms of a cell in ONE 64-byte record
// The buffer is a stable permutation of the stream whenever the nodes of a cell are distinct (inside a segment: items in
// stream order, an item's updates in call order; two items of one cell never meet in a (row, column)), the segment table
// follows from the items' (x W), and esp_flush starts at the bucket kernel (esp_handle::PrePart).  A cell that names a node
// twice (its two items would interleave in the reference's call order), a non-empty buffer, a column window: the
// updates go out in stream order as packed keys (elem_stream_k) and the flush partitions them like any other stream.
#pragma once
#include "common.hpp"
#include "generators.hpp"
#include "segexpand.hpp"

namespace espelem {

constexpr int THREADS = 256;
constexpr int MAX_NLOC = 16;

struct Args {
    int nloc, W;            // nodes per cell; updates per item (nloc + 1 with a diagonal term)
    i64 ncells, nitems;     // nitems = ncells * nloc
    i64 cell_base;          // elem_stream_k: cells of the call in front of this launch (error reports)
    const i64 *cellnodes;   // Int64 nloc x ncells (Julia layout: the nodes of a cell lie together)
    const double *elmat;    // Float64 nloc x nloc x ncells: elmat[il + nloc * (jl + nloc * cell)]
    const double *diag;     // Float64 nloc x ncells or nullptr
    i64 lim;                // every node number must lie in 1 .. lim = min(m, n)
    KeyLayout L;            // the matrix's key layout
    int vrb;                // row bits of the VIRTUAL layout of the item records: record = (col0 << (vrb + 2)) | item number
    int kind, negate;
    unsigned long long *err;  // atomicMin: first offending cell + 1
    u32 *dup;                 // set when a cell names a node twice
    u64 *ikeys;               // item records, stream order
    const u64 *sorted_keys;   // ... partitioned
    int rem_bits;             // elem_expand_k, K32: key bits below the segment prefix (real layout)
    u64 base;                 // key window base
    u64 *keys_out;            // (K32: u32 keys)
    double *vals_out;
    char *cellrec;            // nloc 3 / 4: one 64-byte record per cell (rows as four u32 | the cell's diag values) -- see elem_cells_k
    u32 *colrange;            // {smallest, largest} 0-based node of the batch (atomicMin / atomicMax; matrices below 2^32 columns), or nullptr:
                              // the item partition plans for the columns the batch really touches (a band of a mesh: one partition's cells)
};

// item p = cell * nloc + jl: 32-bit arithmetic (the host keeps nitems below 2^32)
static __global__ __launch_bounds__(THREADS) void elem_items_k(Args a) {
    const i64 p = (i64)blockIdx.x * THREADS + threadIdx.x;
    if (p >= a.nitems) return;
    const u32 nloc = (u32)a.nloc;
    const u32 cell = (u32)p / nloc, jl = (u32)p - cell * nloc;
    i64 node = a.cellnodes[p];
    if (node < 1 || node > a.lim) {
        atomicMin(a.err, (unsigned long long)cell + 1ull);
        node = 1;  // (the record stays a valid key; the host stops before anything is expanded)
    }
    const i64 *cn = a.cellnodes + (i64)cell * nloc;
    bool twice = false;
    for (u32 k = 0; k < jl; k++) twice = twice || cn[k] == node;
    if (twice) *a.dup = 1u;
    a.ikeys[p] = ((u64)(node - 1) << (a.vrb + ESP_TAG_BITS)) | (u64)p;
}

// The same for cells of 3 or 4 nodes (P1 in 2-D / 3-D), one thread per CELL, which also leaves a 64-byte CELL RECORD for
// the expansion: the cell's rows as four u32 (0-based) | its diag values (four f64).  An item's expansion then fetches
// its rows and its diagonal term with ONE memory-side request instead of two: the gathers of elem_expand_k are what it is
// bound by (64-byte requests at random addresses: three per item cost 3.6 ms each at 3-D config-4 size), and the record
// costs this kernel 32 B more to read and 64 B to write per cell, in stream order.
template <int NLOC>
static __global__ __launch_bounds__(THREADS) void elem_cells_k(Args a) {
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
    typedef double dbl2 __attribute__((ext_vector_type(2)));
    typedef unsigned long long ull2 __attribute__((ext_vector_type(2)));
    const i64 c = (i64)blockIdx.x * THREADS + threadIdx.x;
    u32 lo = ~0u, hi = 0u;  // smallest / largest 0-based node of this thread's cell (neutral for a thread without one)
    if (c < a.ncells) {
    i64 nd[4] = {1, 1, 1, 1};
    double dg[4] = {0.0, 0.0, 0.0, 0.0};
    // (32 bytes per cell: two 16-byte loads where the caller's arrays are 16-byte aligned -- a view into a larger array need not be)
    const bool vec = NLOC == 4 && ((reinterpret_cast<unsigned long long>(a.cellnodes) | reinterpret_cast<unsigned long long>(a.diag)) & 15ull) == 0;
    if (vec) {
        const ull2 *cn = reinterpret_cast<const ull2 *>(a.cellnodes + c * 4);
        const ull2 n01 = cn[0], n23 = cn[1];
        nd[0] = (i64)n01.x, nd[1] = (i64)n01.y, nd[2] = (i64)n23.x, nd[3] = (i64)n23.y;
        if (a.diag) {
            const dbl2 *dp = reinterpret_cast<const dbl2 *>(a.diag + c * 4);
            const dbl2 d01 = dp[0], d23 = dp[1];
            dg[0] = d01.x, dg[1] = d01.y, dg[2] = d23.x, dg[3] = d23.y;
        }
    } else {
#pragma unroll
        for (int k = 0; k < NLOC; k++) nd[k] = a.cellnodes[c * NLOC + k];
        if (a.diag) {
#pragma unroll
            for (int k = 0; k < NLOC; k++) dg[k] = a.diag[c * NLOC + k];
        }
    }
    bool bad = false, twice = false;
#pragma unroll
    for (int k = 0; k < NLOC; k++) {
        if (nd[k] < 1 || nd[k] > a.lim) {
            bad = true;
            nd[k] = 1;
        }
#pragma unroll
        for (int q = 0; q < k; q++) twice = twice || nd[q] == nd[k];
    }
    if (bad) atomicMin(a.err, (unsigned long long)c + 1ull);
    if (twice && !bad) *a.dup = 1u;
    const int sh = a.vrb + ESP_TAG_BITS;
    const u64 p0 = (u64)c * NLOC;
    u64 rec[4];
#pragma unroll
    for (int k = 0; k < 4; k++) rec[k] = ((u64)(nd[k] - 1) << sh) | (p0 + (u64)k);
    if constexpr (NLOC == 4) {
        ull2 *o = reinterpret_cast<ull2 *>(a.ikeys + p0);  // (32-byte aligned)
        o[0] = ull2{rec[0], rec[1]};
        o[1] = ull2{rec[2], rec[3]};
    } else {
#pragma unroll
        for (int k = 0; k < NLOC; k++) a.ikeys[p0 + k] = rec[k];
    }
    char *cr = a.cellrec + c * 64;
    *reinterpret_cast<u32x4 *>(cr) = u32x4{(u32)(nd[0] - 1), (u32)(nd[1] - 1), (u32)(nd[2] - 1), (u32)(nd[3] - 1)};
    // (the whole 64-byte record, its unused tail included: full lines instead of masked ones)
    *reinterpret_cast<dbl2 *>(cr + 16) = dbl2{dg[0], dg[1]};
    *reinterpret_cast<dbl2 *>(cr + 32) = dbl2{dg[2], dg[3]};
    *reinterpret_cast<dbl2 *>(cr + 48) = dbl2{0.0, 0.0};
    lo = (u32)(nd[0] - 1), hi = lo;
#pragma unroll
    for (int k = 1; k < NLOC; k++) lo = min(lo, (u32)(nd[k] - 1)), hi = max(hi, (u32)(nd[k] - 1));
    }
    // (every lane of the wave is here: the DPP reductions need them all)
    // A SAMPLE of the batch -- the first wave of every 64th workgroup and of the last one -- tells the host what the cell order looks
    // like (atomics on one address cost the chip ~12 ns each: one per wave of a 2 10^7-cell batch would be 4 ms): the nodes of the
    // wave's 64 consecutive cells go into the batch's range (colrange[0 .. 1]: the item partition plans for the columns the batch
    // touches -- a hint: a sample's range misses at most a few thousand cells at either end), and colrange[2] counts the sampled waves
    // whose cells lie more than lim / 16 apart -- a shuffled cell order or a mesh numbered without locality: nearly all of them; a mesh in
    // its own order: none (the host's cue for the run-based pass).
    if (a.colrange && threadIdx.x < ESP_WAVE && ((blockIdx.x & 63u) == 0u || blockIdx.x == gridDim.x - 1)) {
        const u32 wlo = ~esp_wave_max(~lo), whi = esp_wave_max(hi);
        if ((threadIdx.x & 63) == 0 && wlo <= whi) {
            atomicMin(a.colrange, wlo);
            atomicMax(a.colrange + 1, whi);
            if ((i64)(whi - wlo) > (a.lim >> 4)) atomicAdd(a.colrange + 2, 1u);
        }
    }
}

// the diag values of every cell record again (esp_append_elements_again: the connectivity is the same, the element data new)
template <int NLOC>
static __global__ __launch_bounds__(THREADS) void elem_refresh_diag_k(const double *__restrict__ diag, i64 ncells, char *__restrict__ cellrec) {
    typedef double dbl2 __attribute__((ext_vector_type(2)));
    const i64 c = (i64)blockIdx.x * THREADS + threadIdx.x;
    if (c >= ncells) return;
    double dg[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < NLOC; k++) dg[k] = diag[c * NLOC + k];
    // (the whole 64-byte record goes out again, rows included: full lines instead of half-written ones)
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
    char *cr = cellrec + c * 64;
    const u32x4 rows = *reinterpret_cast<const u32x4 *>(cr);
    *reinterpret_cast<u32x4 *>(cr) = rows;
    *reinterpret_cast<dbl2 *>(cr + 16) = dbl2{dg[0], dg[1]};
    *reinterpret_cast<dbl2 *>(cr + 32) = dbl2{dg[2], dg[3]};
    *reinterpret_cast<dbl2 *>(cr + 48) = dbl2{0.0, 0.0};
}

// K32: 4-byte keys (the bits below the segment prefix; every entry has the batch's kind), else packed keys.
// NLOC > 0: the nodes per cell as a compile-time constant (3, 4), rows and the diagonal term from the cell record
// (elem_cells_k), the column of the element matrix from the caller's array; 0: any 1 .. MAX_NLOC, everything from the
// caller's arrays.  Dynamic LDS: THREADS * W values, then THREADS * W keys.
template <bool K32, int NLOC, bool CR = (NLOC > 0)>
static __global__ __launch_bounds__(THREADS) void elem_expand_k(Args a) {
    typedef typename std::conditional<K32, u32, u64>::type KT;
    typedef unsigned long long ull2 __attribute__((ext_vector_type(2)));
    typedef double dbl2 __attribute__((ext_vector_type(2)));
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ double elem_lds[];
    const int t = threadIdx.x;
    const int nloc = NLOC ? NLOC : a.nloc;
    const int W = a.W;
    double *lv = elem_lds;
    KT *lk = reinterpret_cast<KT *>(lv + THREADS * W);
    const i64 g0 = (i64)blockIdx.x * THREADS, g = g0 + t;
    if (g < a.nitems) {
        const u64 rec = a.sorted_keys[g];
        const int low = a.vrb + ESP_TAG_BITS;
        const u32 p = (u32)(rec & ((1ull << low) - 1ull));
        const u64 colpart = (rec >> low) << a.L.rb;
        const u32 cell = p / (u32)nloc, jl = p - cell * (u32)nloc;
        const double *em = a.elmat + (i64)p * nloc;  // column jl of the cell's element matrix
        const u64 lowmask = a.rem_bits >= 64 ? ~0ull : ((1ull << a.rem_bits) - 1ull);
        int at = t * W;
        auto put = [&](u64 row0, double v) {
            if (a.negate) v = -v;
            if constexpr (K32)
                lk[at] = (u32)(((colpart | row0) - a.base) & lowmask);
            else
                lk[at] = ((colpart | row0) << ESP_TAG_BITS) | (u64)a.kind;
            lv[at] = v;
            at++;
        };
        if constexpr (NLOC > 0 && CR) {
            const char *cr = a.cellrec + (i64)cell * 64;
            const u32x4 rr = *reinterpret_cast<const u32x4 *>(cr);
            const double d = a.diag ? *reinterpret_cast<const double *>(cr + 16 + 8 * jl) : 0.0;
            const u32 r[4] = {rr.x, rr.y, rr.z, rr.w};
            double v[NLOC];
#pragma unroll
            for (int il = 0; il < NLOC; il++) v[il] = em[il];
#pragma unroll
            for (int il = 0; il < NLOC; il++) {
                if (a.diag && (u32)il == jl) put((u64)r[il], d);  // (the diagonal's term comes right before the diagonal: femtools.jl:64)
                put((u64)r[il], v[il]);
            }
        } else if constexpr (NLOC > 0) {  // (no cell records: more than 2^32 rows, or the test hook)
            const i64 *cn = a.cellnodes + (i64)cell * NLOC;
            i64 r[NLOC];
            double v[NLOC];
#pragma unroll
            for (int il = 0; il < NLOC; il++) r[il] = cn[il];
#pragma unroll
            for (int il = 0; il < NLOC; il++) v[il] = em[il];
            const double d = a.diag ? a.diag[p] : 0.0;
#pragma unroll
            for (int il = 0; il < NLOC; il++) {
                if (a.diag && (u32)il == jl) put((u64)(r[il] - 1), d);
                put((u64)(r[il] - 1), v[il]);
            }
        } else {
            // (any cell size: all loads requested before the first use -- a loop of dependent round trips took 2.4 times as long)
            const i64 *cn = a.cellnodes + (i64)cell * nloc;
            i64 r[MAX_NLOC];
            double v[MAX_NLOC];
#pragma unroll
            for (int il = 0; il < MAX_NLOC; il++) r[il] = il < nloc ? cn[il] : 1;
#pragma unroll
            for (int il = 0; il < MAX_NLOC; il++) v[il] = il < nloc ? em[il] : 0.0;
            const double d = a.diag ? a.diag[p] : 0.0;
#pragma unroll
            for (int il = 0; il < MAX_NLOC; il++) {
                if (il < nloc) {
                    if (a.diag && (u32)il == jl) put((u64)(r[il] - 1), d);
                    put((u64)(r[il] - 1), v[il]);
                }
            }
        }
    }
    __syncthreads();
    const int cnt = (int)min((i64)THREADS, a.nitems - g0) * W;
    // whole lines: g0 * W is a multiple of 256, every array starts 256-byte aligned
    double *gv = a.vals_out + g0 * W;
    const int vpair = cnt >> 1;
    for (int q = t; q < vpair; q += THREADS) reinterpret_cast<dbl2 *>(gv)[q] = dbl2{lv[2 * q], lv[2 * q + 1]};
    if (t == 0 && (cnt & 1)) gv[cnt - 1] = lv[cnt - 1];
    if constexpr (K32) {
        u32 *gk = reinterpret_cast<u32 *>(a.keys_out) + g0 * W;
        const int kquad = cnt >> 2;
        for (int q = t; q < kquad; q += THREADS)
            reinterpret_cast<u32x4 *>(gk)[q] = u32x4{lk[4 * q], lk[4 * q + 1], lk[4 * q + 2], lk[4 * q + 3]};
        for (int q = 4 * kquad + t; q < cnt; q += THREADS) gk[q] = lk[q];
    } else {
        u64 *gk = a.keys_out + g0 * W;
        for (int q = t; q < vpair; q += THREADS) reinterpret_cast<ull2 *>(gk)[q] = ull2{lk[2 * q], lk[2 * q + 1]};
        if (t == 0 && (cnt & 1)) gk[cnt - 1] = lk[cnt - 1];
    }
}

// host side: the expansion's launch (elements.hip, and lazy_expand in produce.hip)
template <bool K32>
static inline void launch_expand(const Args &a, hipStream_t stream) {
    const dim3 grid((unsigned)((a.nitems + THREADS - 1) / THREADS)), block(THREADS);
    const size_t lds = (size_t)THREADS * (size_t)a.W * (sizeof(double) + (K32 ? sizeof(u32) : sizeof(u64)));
    if (a.nloc == 3 && a.cellrec)
        hipLaunchKernelGGL((elem_expand_k<K32, 3>), grid, block, lds, stream, a);
    else if (a.nloc == 4 && a.cellrec)
        hipLaunchKernelGGL((elem_expand_k<K32, 4>), grid, block, lds, stream, a);
    else if (a.nloc == 3)
        hipLaunchKernelGGL((elem_expand_k<K32, 3, false>), grid, block, lds, stream, a);
    else if (a.nloc == 4)
        hipLaunchKernelGGL((elem_expand_k<K32, 4, false>), grid, block, lds, stream, a);
    else
        hipLaunchKernelGGL((elem_expand_k<K32, 0>), grid, block, lds, stream, a);
}


// The expansion with the partition's last bits done inside it (segexpand.hpp), cells of 3 / 4 nodes with cell records: one
// workgroup per segment of up to 4096 item records orders them by the next lbits bits and expands them in that order.
template <bool K32, int NLOC>
static __global__ __launch_bounds__(espseg::THREADS) void elem_seg_expand_k(Args a, espseg::SegArgs sa) {
    typedef typename std::conditional<K32, u32, u64>::type KT;
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
    constexpr int WMAX = NLOC + 1;
    __shared__ espseg::SegLds L;
    __shared__ KT lk[espseg::THREADS * WMAX];
    __shared__ double lv[espseg::THREADS * WMAX];
    const int low = a.vrb + ESP_TAG_BITS;
    const u64 lowmask = a.rem_bits >= 64 ? ~0ull : ((1ull << a.rem_bits) - 1ull);
    espseg::segment_expand<KT>(sa, L, lk, lv, reinterpret_cast<KT *>(a.keys_out), a.vals_out, [&](u64 rec, KT *k, double *v) {
        const u32 p = (u32)(rec & ((1ull << low) - 1ull));
        const u64 colpart = (rec >> low) << a.L.rb;
        const u32 cell = p / (u32)NLOC, jl = p - cell * (u32)NLOC;
        const double *em = a.elmat + (i64)p * NLOC;
        const char *cr = a.cellrec + (i64)cell * 64;
        const u32x4 rr = *reinterpret_cast<const u32x4 *>(cr);
        const double d = a.diag ? *reinterpret_cast<const double *>(cr + 16 + 8 * jl) : 0.0;
        const u32 r[4] = {rr.x, rr.y, rr.z, rr.w};
        double ev[NLOC];
#pragma unroll
        for (int il = 0; il < NLOC; il++) ev[il] = em[il];
        int at = 0;
        auto put = [&](u64 row0, double val) {
            if (a.negate) val = -val;
            if constexpr (K32)
                k[at] = (u32)(((colpart | row0) - a.base) & lowmask);
            else
                k[at] = ((colpart | row0) << ESP_TAG_BITS) | (u64)a.kind;
            v[at] = val;
            at++;
        };
#pragma unroll
        for (int il = 0; il < NLOC; il++) {
            if (a.diag && (u32)il == jl) put((u64)r[il], d);
            put((u64)r[il], ev[il]);
        }
    });
}

// The updates in stream order, as packed keys (any buffer state; what the flush's own partition then takes): entry e of
// the launch = (cell, il, q), q = 0: the diagonal's term (with diag), then jl = 0 .. nloc-1.
static __global__ __launch_bounds__(THREADS) void elem_stream_k(Args a, i64 count) {
    const i64 e = (i64)blockIdx.x * THREADS + threadIdx.x;
    if (e >= count) return;
    const i64 per_cell = (i64)a.nloc * a.W;
    const i64 cell = e / per_cell;
    const int rem = (int)(e - cell * per_cell);
    const int il = rem / a.W, q = rem - il * a.W;
    const i64 *cn = a.cellnodes + cell * a.nloc;
    const i64 row = cn[il];
    i64 col;
    double v;
    if (a.diag && q == 0) {
        col = row;
        v = a.diag[cell * a.nloc + il];
    } else {
        const int jl = a.diag ? q - 1 : q;
        col = cn[jl];
        v = a.elmat[(cell * a.nloc + jl) * a.nloc + il];
    }
    if (row < 1 || row > a.lim || col < 1 || col > a.lim) {
        atomicMin(a.err, (unsigned long long)(a.cell_base + cell) + 1ull);
        return;
    }
    if (a.negate) v = -v;
    a.keys_out[e] = esp_pack(a.L, row, col, a.kind);
    a.vals_out[e] = v;
}

// ---- the arrays a caller of testassemble! holds, for the build's own Kuhn grid (tests and bench: test/femtools.jl:46-67) ----
struct MeshArgs {
    espgen::FemArgs fem;    // cells: dim, npd, ncells, seed, order_mode, bits, mq, h
    espgen::FemArgs nodes;  // the node renumbering as a second Feistel bijection: ncells = number of nodes, seed, bits; order_mode 0: natural
    i64 p0, p1;             // stream positions of the cells
    i64 *cellnodes;
    double *elmat, *diag;
};
static __global__ __launch_bounds__(THREADS) void fem_mesh_k(MeshArgs a) {
    const i64 q = (i64)blockIdx.x * THREADS + threadIdx.x;
    if (a.p0 + q >= a.p1) return;
    const int nloc = a.fem.dim + 1;
    espgen::fem_cell_updates(a.fem, a.p0 + q, [&](int il, int jl, i64 row, i64 col, double v) {
        (void)col;
        if (jl < 0) {
            a.cellnodes[q * nloc + il] = 1 + (i64)espgen::fem_cell_at(a.nodes, row - 1);
            if (a.diag) a.diag[q * nloc + il] = v;
        } else {
            a.elmat[(q * nloc + jl) * nloc + il] = v;
        }
    });
}

}  // namespace espelem
