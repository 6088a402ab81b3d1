// local.hpp -- LDS bucket kernel (K2 tail + K3 + K4 marks): one workgroup takes one segment of
// the prefix-partitioned entry array (<= CAP entries sharing the top key bits), finishes the
// stable sort on the remaining key bits inside LDS, folds duplicates in append order and writes
// the distinct entries to their FINAL position in the output (rowval/nzval of a fresh CSC, or
// the compact list of new entries for the merge join) with coalesced stores.
//
// HBM traffic: reads 16 B per appended entry once (12 B when the partition handed over 4-byte keys), writes
// 16 B per emitted entry once + 8 B per column (colptr itself for a fresh matrix, else column-end marks).
//
// Structure (one workgroup = 512 threads = 8 waves per segment, 2 workgroups per CU: 64 KiB of LDS each)
//   claim     : segments are claimed through an atomic ticket (start order = ticket order); the segment
//               bounds around the expected ticket are fetched while the atomic is in flight
//   load      : 8 B key + 8 B value per slot, all 16 loads of a thread in flight; branch-free transform
//               to the packed sort key (bits below the segment prefix | slot index | kind) in registers,
//               value to LDS.  KEYS 1/2: 4-byte keys (the bits below the prefix; one kind for all entries,
//               KEYS 2: that kind is UPDATE and the register tiers fold without decoding it).  PIECES variant (column shards): the segment is the concatenation of one
//               piece per source rank, read in place from the receive buffers
//   sort      : counting sort by local column (LDS atomics, unordered inside a column), then per column
//               run, chosen per segment from the longest run:
//                 <= 12 / 16 / 24 : one lane per column, the run in REGISTERS, merge-exchange network
//                         of 42 / 63 / 132 compare-exchanges = v_min_f64 + v_max_f64 (keys < 2^62 order
//                         like doubles); 24 lives in its own kernel instantiation (BIG)
//                 else  : stable 8-bit LSD radix on the varying key bits (ballot ranking), then a fold with
//                         one (col,row) group per thread
//               the slot index is the append order, so every tier yields the stable order
//   fold      : ordered left-to-right fold per (col,row) (espfold::fold_step); CSC hits are applied in
//               place (merge walk over the CSC column), misses become records.  Fresh matrix: the
//               number of records is known right after the sort, so the segment total is published
//               BEFORE the fold
//   look-back : decoupled look-back over the segment totals gives the global output offset (8-byte
//               {flag,value} granules, relaxed agent-scope atomics both sides -- MI355X L2s are per XCD;
//               spins are bounded).  The last wave -- idle while one lane per column folds -- owns it and
//               carries it across the following barriers: nothing before the final stores needs the
//               offset
//   compact   : records -> dense LDS prefix (ballot ranks)
//   store     : coalesced stores of rowval/nzval (or key/val); fresh matrix over whole-column segments: the
//               first entry of a column writes colptr for it and the empty columns before it (no marks, no
//               scan over the columns afterwards); else column-end marks
// Bound: instruction issue and barrier latency at 4 waves per SIMD (128 VGPRs), not HBM bytes; see DESIGN.md.
#pragma once
#include "local_args.hpp"

namespace esplocal {

// Batcher's merge-exchange sorting network for 16 keys (63 compare-exchanges), generated at
// compile time; used to sort one short column run per lane entirely in registers.
struct Net {
    int n;
    int a[160];
    int b[160];
};
constexpr Net make_net(int N) {
    Net r{};
    r.n = 0;
    for (int p = 1; p < N; p <<= 1)
        for (int kk = p; kk >= 1; kk >>= 1)
            for (int j = kk % p; j <= N - 1 - kk; j += 2 * kk)
                for (int i = 0; i <= (kk - 1 < N - j - kk - 1 ? kk - 1 : N - j - kk - 1); i++)
                    if ((i + j) / (2 * p) == (i + j + kk) / (2 * p)) {
                        r.a[r.n] = i + j;
                        r.b[r.n] = i + j + kk;
                        r.n++;
                    }
    return r;
}
template <int R>
struct NetOf {
    static constexpr Net net = make_net(R);
};
static_assert(NetOf<16>::net.n == 63, "merge-exchange network for 16 inputs has 63 comparators");
static_assert(NetOf<12>::net.n == 42, "merge-exchange network for 12 inputs has 42 comparators");
static_assert(NetOf<24>::net.n == 132, "merge-exchange network for 24 inputs has 132 comparators");


// Stable LSD radix tail on the packed keys held in registers (wave-striped arrangement):
// sorts on bits [SUB_SHIFT, SUB_SHIFT+rem_bits); result in k[] and skey[].
// Only the key bits in which the segment's entries actually DIFFER are sorted: (col,row) keys of one
// segment usually agree in a block of middle bits (rows near the diagonal: the high row bits), and a
// digit may be put together from two runs of varying bits, so such a block costs no pass.  Returns the
// number of passes (0: all entries share one (col,row); skey is not written then).
// lowshift: the lowest key bit that takes part (SUB_SHIFT: the (col,row) bits, entries arrive in append order; ESP_TAG_BITS:
// the slot index too -- entries may then arrive in any order)
__device__ __forceinline__ int radix_tail(u64 (&k)[ITEMS], u64 *skey, u32 (*cnt)[256], u32 *lw, int rem_bits, int t, int lane,
                                          int w, int wbase, int n, u64 *scratch /* 2*WAVES words */, int lowshift = SUB_SHIFT) {
    const u64 lt = (1ull << lane) - 1ull;
    // varying bits = AND ^ OR over the real entries
    u64 vand = ~0ull, vor = 0ull;
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        const bool real = wbase + i * ESP_WAVE < n;
        vand &= real ? k[i] : ~0ull;
        vor |= real ? k[i] : 0ull;
    }
#pragma unroll
    for (int dlt = 32; dlt > 0; dlt >>= 1) {
        vand &= __shfl_xor(vand, dlt, ESP_WAVE);
        vor |= __shfl_xor(vor, dlt, ESP_WAVE);
    }
    if (lane == 0) {
        scratch[w] = vand;
        scratch[WAVES + w] = vor;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < WAVES; i++) {
        vand &= scratch[i];
        vor |= scratch[WAVES + i];
    }
    const u64 sortmask = rem_bits >= 64 - lowshift ? ~0ull : (((u64)1 << rem_bits) - 1ull);
    u64 rest = ((vand ^ vor) >> lowshift) & sortmask;  // bit j: key bit lowshift + j varies
    int npass = 0;
    while (rest) {
        // a digit of up to 8 bits from the lowest one or two runs of varying bits (A below B)
        const int sA = __builtin_ctzll(rest);
        const int lenA = (int)__builtin_ctzll(~(rest >> sA) | ((u64)1 << 63));
        const int bA = min(lenA, 8);
        rest &= ~((((u64)1 << bA) - 1ull) << sA);
        int sB = 0, bB = 0;
        if (bA < 8 && rest) {
            sB = __builtin_ctzll(rest);
            const int lenB = (int)__builtin_ctzll(~(rest >> sB) | ((u64)1 << 63));
            bB = min(lenB, 8 - bA);
            rest &= ~((((u64)1 << bB) - 1ull) << sB);
        }
        const int shA = lowshift + sA, shB = lowshift + sB;
        const u32 mA = (1u << bA) - 1u, mB = (1u << bB) - 1u;
        npass++;
        for (int q = t; q < WAVES * 256; q += THREADS) (&cnt[0][0])[q] = 0;
        __syncthreads();
        unsigned short rank[ITEMS];
#pragma unroll
        for (int i = 0; i < ITEMS; i++) {
            const u32 d = ((u32)(k[i] >> shA) & mA) | (((u32)(k[i] >> shB) & mB) << bA);
            u64 m = ~0ull;
#pragma unroll
            for (int b = 0; b < 8; b++) {
                if (b < bA + bB) {  // (uniform: a digit of 5 varying bits pays 5 ballots)
                    const bool bit = (d >> b) & 1u;
                    const u64 bb = __ballot(bit);
                    m &= bit ? bb : ~bb;
                }
            }
            const u32 prev = cnt[w][d];
            rank[i] = (unsigned short)(prev + (u32)__popcll(m & lt));
            __builtin_amdgcn_wave_barrier();
            if ((m & lt) == 0) cnt[w][d] = prev + (u32)__popcll(m);
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
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
            const u32 d = ((u32)(k[i] >> shA) & mA) | (((u32)(k[i] >> shB) & mB) << bA);
            skey[cnt[w][d] + rank[i]] = k[i];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < ITEMS; i++) k[i] = skey[wbase + i * ESP_WAVE];
    }
    return npass;
}

// closes one (col,row) group of a column run: CSC hit -> in place, miss -> record at skey[rs+e]
__device__ __forceinline__ void close_group(const Args &a, u64 *skey, double *sval, int rs, int &e, i64 pos, bool present,
                                            double acc, u64 psub, u32 idx0) {
    if (pos >= 0) {
        if (a.mode == ESP_FLUSH_ROUTED)
            a.csc.nzval[pos] = acc;
        else if (present)
            a.csc.nzval[pos] = a.csc.nzval[pos] + acc;  // csc operand first, sparsematrixlnk.jl:363
    } else if (present) {
        skey[rs + e] = (psub << SUB_SHIFT) | ((u64)idx0 << ESP_TAG_BITS);
        sval[idx0] = acc;
        e++;
    }
}

// loads one column run (<= R entries at skey[rs..rs+len)) into registers and sorts it
template <int R>
__device__ __forceinline__ void sort_run_keys(const u64 *skey, int rs, int len, u64 (&x)[R]) {
    const int lastj = len > 0 ? len - 1 : 0;
#pragma unroll
    for (int j = 0; j < R; j++) x[j] = skey[rs + min(j, lastj)];  // all reads in flight
    // Sort keys of this tier are < 2^62 (rem_bits <= REG_MAX_REM): read as IEEE doubles they are positive,
    // finite and ordered like the integers (denormals are not flushed for f64), so a compare-exchange
    // is v_min_f64 + v_max_f64 -- two full-rate instructions instead of a 64-bit compare and four
    // selects.  Padding = the largest finite double.
    constexpr u64 PAD = 0x7FEFFFFFFFFFC003ull;  // (slot-index bits zero: a padded key reads value slot 0)
    double d[R];
#pragma unroll
    for (int j = 0; j < R; j++) d[j] = __longlong_as_double((long long)(j < len ? x[j] : PAD));
#pragma unroll
    for (int q = 0; q < NetOf<R>::net.n; q++) {
        const double lo = d[NetOf<R>::net.a[q]], hi2 = d[NetOf<R>::net.b[q]];
        double mn, mx;
        asm("v_min_f64 %0, %1, %2" : "=v"(mn) : "v"(lo), "v"(hi2));
        asm("v_max_f64 %0, %1, %2" : "=v"(mx) : "v"(lo), "v"(hi2));
        d[NetOf<R>::net.a[q]] = mn;
        d[NetOf<R>::net.b[q]] = mx;
    }
#pragma unroll
    for (int j = 0; j < R; j++) x[j] = (u64)__double_as_longlong(d[j]);
}
template <int R>
__device__ __forceinline__ void load_run_values(const double *sval, const u64 (&x)[R], double (&xv)[R]) {
#pragma unroll
    for (int j = 0; j < R; j++) xv[j] = sval[(x[j] >> ESP_TAG_BITS) & (CAP - 1)];  // (padding reads slot 0)
}
template <int R>
__device__ __forceinline__ void load_sorted_run(const u64 *skey, const double *sval, int rs, int len, u64 (&x)[R],
                                                double (&xv)[R]) {
    sort_run_keys<R>(skey, rs, len, x);
    load_run_values<R>(sval, x, xv);
}

// number of entries a sorted run will emit when nothing of it is in the CSC: a (col,row) group
// becomes present iff one of its updates is a RAWUPDATE / COO entry or has a non-zero value (fold_step)
// UPD: every entry is an UPDATE (the kernel was told so): no kind to decode
template <int R, bool UPD>
__device__ __forceinline__ u32 count_emitted(const u64 (&x)[R], const double (&xv)[R], int len) {
    u32 e = 0;
    bool any = false;
    u64 psub = 0;
#pragma unroll
    for (int j = 0; j <= R; j++) {
        const bool valid = j < R && j < len;
        const u64 sub = (j < R ? x[j] : NOREC) >> SUB_SHIFT;
        const bool fresh = j == 0 || !valid || sub != psub;
        if (fresh && j > 0 && j <= len) e += any ? 1u : 0u;
        if (valid) {
            if (fresh) {
                psub = sub;
                any = false;
            }
            if constexpr (UPD)
                any |= xv[j < R ? j : 0] != 0.0;
            else
                any |= ((u32)(x[j < R ? j : 0] & ESP_TAG_MASK) >= (u32)ESP_RAWUPDATE) || xv[j < R ? j : 0] != 0.0;
        }
    }
    return e;
}

// ordered fold of one sorted run held in registers; records go to skey[rs..), NOREC behind them
// NOCSC: the matrix holds no entries yet (fresh build): no position can hit the CSC
template <int R, bool NOCSC, bool UPD>
__device__ __forceinline__ void fold_run(const Args &a, u64 *skey, double *sval, const u64 (&x)[R], int rs, int len, u64 hi,
                                         u64 rowmask, i64 ccur = 0, i64 cend = 0) {
    if constexpr (NOCSC) {
        double xv[R];
        load_run_values<R>(sval, x, xv);
        int e = 0;
        bool present = false;
        double acc = 0.0;
        u64 prev = 0;  // first key of the running group
#pragma unroll
        for (int j = 0; j <= R; j++) {
            const bool valid = j < R && j < len;
            const u64 kj = j < R ? x[j] : NOREC;
            const bool fresh = j == 0 || !valid || (kj >> SUB_SHIFT) != (prev >> SUB_SHIFT);
            if (fresh && j > 0 && j <= len && present) {  // close the group: a record at the run's front
                const u32 idx0 = (u32)(prev >> ESP_TAG_BITS) & (CAP - 1);
                skey[rs + e] = (prev & ~(((u64)1 << SUB_SHIFT) - 1)) | ((u64)idx0 << ESP_TAG_BITS);
                sval[idx0] = acc;
                e++;
            }
            if (fresh) {
                prev = kj;
                present = false;
                acc = 0.0;
            }
            if (valid) {
                if constexpr (UPD)
                    espfold::fold_step_update(present, acc, xv[j < R ? j : 0]);
                else
                    espfold::fold_step_sel(present, acc, (u32)(kj & ESP_TAG_MASK), xv[j < R ? j : 0]);
            }
        }
#pragma unroll
        for (int j = 0; j < R; j++)
            if (j >= e && j < len) skey[rs + j] = NOREC;
        return;
    }
    // ---- existing CSC: the run's rows come in increasing order, the CSC column is walked with a cursor
    // (findindex, sparsematrixcsc.jl:7-23, restated as a merge walk; same result as the binary search).
    // (measured: fetching the column in one batch and merging in registers -- no dependent loads -- costs
    // more instructions than the walk saves in latency: 13.6 against 8.4 ms at config 3)
    // (the values are read from LDS where they are used: the walk's state leaves no room for R of them in registers)
    int e = 0;
    bool present = false;
    double acc = 0.0;
    u64 psub = 0;
    u32 idx0 = 0;
    i64 pos = -1;
    // (ccur, cend: the column's range in the CSC, requested by the caller before it sorted the run -- csc_column)
#pragma unroll
    for (int j = 0; j <= R; j++) {
        const bool valid = j < R && j < len;
        const u64 kj = j < R ? x[j] : NOREC;
        const u64 sub = kj >> SUB_SHIFT;
        const bool fresh = j == 0 || !valid || sub != psub;
        if (fresh && j > 0 && j <= len) close_group(a, skey, sval, rs, e, pos, present, acc, psub, idx0);
        if (valid) {
            if (fresh) {
                psub = sub;
                idx0 = (u32)(kj >> ESP_TAG_BITS) & (CAP - 1);
                pos = -1;
                if (ccur < cend) {
                    const i64 want = (i64)((hi + sub) & rowmask) + 1;
                    while (ccur < cend && a.csc.rowval[ccur] < want) ccur++;
                    if (ccur < cend && a.csc.rowval[ccur] == want) pos = ccur;
                }
                present = (pos >= 0 && a.mode == ESP_FLUSH_ROUTED);
                acc = present ? a.csc.nzval[pos] : 0.0;
            }
            const double vj = sval[(kj >> ESP_TAG_BITS) & (CAP - 1)];
            if constexpr (UPD)
                espfold::fold_step_update(present, acc, vj);
            else
                espfold::fold_step_sel(present, acc, (u32)(kj & ESP_TAG_MASK), vj);
        }
    }
#pragma unroll
    for (int j = 0; j < R; j++)
        if (j >= e && j < len) skey[rs + j] = NOREC;
}

// The same fold on a fresh matrix with the run's values in registers and the records going straight to their DENSE place
// (`at`: the segment's records in front of this run's -- the emitted counts were scanned before the fold): what the final
// stores read, no record slots, no NOREC marks, no compaction over the segment's slots afterwards.
template <int R, bool UPD>
__device__ __forceinline__ void fold_run_dense(u64 *skey, double *sval, const u64 (&x)[R], const double (&xv)[R], int len, u64 hi, u32 at) {
    bool present = false;
    double acc = 0.0;
    u64 prev = 0;  // first key of the running group
#pragma unroll
    for (int j = 0; j <= R; j++) {
        const bool valid = j < R && j < len;
        const u64 kj = j < R ? x[j] : NOREC;
        const bool fresh = j == 0 || !valid || (kj >> SUB_SHIFT) != (prev >> SUB_SHIFT);
        if (fresh && j > 0 && j <= len && present) {  // close the group
            skey[at] = hi + (prev >> SUB_SHIFT);
            sval[at] = acc;
            at++;
        }
        if (fresh) {
            prev = kj;
            present = false;
            acc = 0.0;
        }
        if (valid) {
            if constexpr (UPD)
                espfold::fold_step_update(present, acc, xv[j < R ? j : 0]);
            else
                espfold::fold_step_sel(present, acc, (u32)(kj & ESP_TAG_MASK), xv[j < R ? j : 0]);
        }
    }
}

// Ordered fold of one sorted run over a SHORT stored column (at most CSC_SHORT entries, rows below 2^32): the walk of
// fold_run is a chain of dependent global loads -- a row, then the value it hits, then the store, for every group of
// the run (stencil re-assembly: 14 round trips, 8 of a segment's 17 us).  Here the column's rows and values are
// requested at once (one round trip, consecutive lanes read consecutive memory) and held in registers under STATIC
// indices: the merge runs over the stored entries in an unrolled outer loop and consumes the run -- written back to
// LDS in sorted order, so that IT can be indexed dynamically -- in an inner loop.  Stores of hits are issued and not
// waited for.  Same results as fold_run (the same fold steps in the same order).
constexpr int CSC_SHORT = 8;
// WITHV: the values come with the rows (the host expects hits -- Args::expect_hits); else only the rows, and a hit fetches
// its value by itself: a tail of new positions reads no value at all (the regular kernel has the registers for both
// forms; in the small variant such a tail walks).
template <int R, bool UPD, bool WITHV>
__device__ __forceinline__ void fold_run_short_csc(const Args &a, u64 *skey, double *sval, const u64 (&x)[R], int rs, int len, u64 hi,
                                                   u64 rowmask, i64 c0, int n) {
    u32 r0[CSC_SHORT];
    double sv[WITHV ? CSC_SHORT : 1];
#pragma unroll
    for (int i = 0; i < CSC_SHORT; i++) {
        const i64 at = c0 + (i < n ? i : n - 1);
        r0[i] = (u32)(a.csc.rowval[at] - 1);
        if constexpr (WITHV) sv[i] = a.csc.nzval[at];
    }
#pragma unroll
    for (int j = 0; j < R; j++)
        if (j < len) skey[rs + j] = x[j];
    const bool routed = a.mode == ESP_FLUSH_ROUTED;
    int p = 0, e = 0;
    bool present = false;
    double acc = 0.0;
    u64 psub = 0;
    u32 idx0 = 0;
    i64 pos = -1;
    auto close = [&]() {
        if (pos >= 0) {
            if (routed)
                a.csc.nzval[pos] = acc;
            else if (present)
                a.csc.nzval[pos] = a.csc.nzval[pos] + acc;  // csc operand first, sparsematrixlnk.jl:363
        } else if (present) {
            skey[rs + e] = (psub << SUB_SHIFT) | ((u64)idx0 << ESP_TAG_BITS);
            sval[idx0] = acc;
            e++;
        }
    };
#pragma unroll
    for (int i = 0; i <= CSC_SHORT; i++) {
        // stored entry i (behind the column: nothing bounds the rows any more)
        const u64 ri = (i < CSC_SHORT && i < n) ? (u64)r0[i < CSC_SHORT ? i : 0] : ~0ull;
        double si = 0.0;
        if constexpr (WITHV) si = sv[i < CSC_SHORT ? i : 0];
        while (p < len) {
            const u64 kj = skey[rs + p];
            const u64 sub = kj >> SUB_SHIFT;
            const u64 row0 = (hi + sub) & rowmask;
            if (row0 > ri) break;
            if (p == 0 || sub != psub) {
                if (p > 0) close();
                psub = sub;
                idx0 = (u32)(kj >> ESP_TAG_BITS) & (CAP - 1);
                const bool hit = row0 == ri;
                pos = hit ? c0 + i : -1;
                present = hit && routed;
                if constexpr (WITHV)
                    acc = present ? si : 0.0;
                else
                    acc = present ? a.csc.nzval[c0 + i] : 0.0;
            }
            const double vj = sval[(kj >> ESP_TAG_BITS) & (CAP - 1)];
            if constexpr (UPD)
                espfold::fold_step_update(present, acc, vj);
            else
                espfold::fold_step_sel(present, acc, (u32)(kj & ESP_TAG_MASK), vj);
            p++;
        }
    }
    if (len > 0) close();
    for (int j = e; j < len; j++) skey[rs + j] = NOREC;
}

// The same fold for a run of ANY length that lies sorted in LDS (skey[rs .. rs+len)): the slow tier of the small
// kernel variant.  Records go to the run's front (a record is written behind the keys already read), NOREC behind them.
template <bool NOCSC, bool UPD>
__device__ __forceinline__ void fold_run_lds(const Args &a, u64 *skey, double *sval, int rs, int len, u64 hi, u64 rowmask) {
    int e = 0;
    bool present = false;
    double acc = 0.0;
    u64 psub = 0;
    u32 idx0 = 0;
    i64 pos = -1;
    i64 ccur = 0, cend = 0;
    if (!NOCSC && a.csc.nnz > 0 && len > 0) {
        const i64 col0 = (i64)((hi + (skey[rs] >> SUB_SHIFT)) >> a.rb);
        ccur = a.csc.colptr[col0] - 1;
        cend = a.csc.colptr[col0 + 1] - 1;
    }
    for (int j = 0; j <= len; j++) {
        const bool valid = j < len;
        const u64 kj = valid ? skey[rs + j] : NOREC;
        const u64 sub = kj >> SUB_SHIFT;
        const bool fresh = j == 0 || !valid || sub != psub;
        if (fresh && j > 0) close_group(a, skey, sval, rs, e, pos, present, acc, psub, idx0);
        if (valid) {
            if (fresh) {
                psub = sub;
                idx0 = (u32)(kj >> ESP_TAG_BITS) & (CAP - 1);
                pos = -1;
                if (ccur < cend) {
                    const i64 want = (i64)((hi + sub) & rowmask) + 1;
                    while (ccur < cend && a.csc.rowval[ccur] < want) ccur++;
                    if (ccur < cend && a.csc.rowval[ccur] == want) pos = ccur;
                }
                present = (pos >= 0 && a.mode == ESP_FLUSH_ROUTED);
                acc = present ? a.csc.nzval[pos] : 0.0;
            }
            const double v = sval[(kj >> ESP_TAG_BITS) & (CAP - 1)];
            if constexpr (UPD)
                espfold::fold_step_update(present, acc, v);
            else
                espfold::fold_step_sel(present, acc, (u32)(kj & ESP_TAG_MASK), v);
        }
    }
    for (int j = e; j < len; j++) skey[rs + j] = NOREC;
}

// ---- two-level look-back, run by ONE wave --------------------------------------------------------------
// A segment's output offset = the emitted entries of all segments with a lower ticket.  Segments are taken in groups
// of LB_GROUP = 256 consecutive tickets:
//   status[s]   = AGG | total of segment s                 (published as soon as the total is known)
//   gstatus[g]  = PRE | total of the groups 0..g            (published by the LAST segment of group g)
//   offset(s)   = gstatus[g-1] + the totals of the segments of its own group in front of it
// i.e. ONE round of polls of the own group (up to four 64-wide loads in flight together; its members hold
// neighbouring tickets: they publish at about the same time) and ONE granule of the previous group, whose last
// segment started 256 tickets earlier.  The chain that remains is over groups only: a link is one cross-XCD round
// trip (about 1.5 us), and there are S/256 of them -- with groups of 64 the 1024 links of a 256^3 flush added up to
// the kernel's whole run time once three workgroups per CU were resident (look-back 8.6 us per segment); a decoupled
// look-back over the group granules instead (64 granule loads per poll and segment) was slower still.  The linear
// decoupled look-back of round 1 walked back over every segment in flight (seven dependent polls).  Granules: 8 bytes
// {flag, value}, relaxed agent-scope atomics on both sides (MI355X L2s are per XCD), spins bounded (a timeout surfaces
// as ESP_ERR_HIP); a segment waits for lower tickets only, and tickets are drawn in start order: no deadlock.
// The ticket counter lives on a page of its own behind the granules (flush_local): every workgroup draws from it while
// others poll the granules, and on one line with them the draws cost the headline's bucket kernel 0.18 of 1.55 ms.
// (Round 4 also tried groups of 64 with unchained group TOTALS under a chain over super-groups of 64 groups -- two 64-wide
// loads per round, a lane asking only for what it has not seen: no faster for local_k, slower for the persistent wave_k;
// a poll interval of 8 or 30 instead of 2 x 64 cycles, the group granules on lines of their own, four copies of each: nothing.)
// WHEN the polls happen matters more than how: a round of polls in front of a barrier makes the polling wave -- and the
// workgroup -- a round trip late (HIP's barrier waits for a wave's outstanding loads): 1.30 -> 1.13 ms for the headline's
// kernel once the round right after the publication was gone; a round kept in flight ACROSS that barrier (a bare s_barrier for
// the polling wave), asked for 0.1 / 0.3 / 0.8 / 1.3 us after the publication: 1.27 / 1.21 / 1.13 / 1.14 ms -- polls that come back empty are worse than
// polls that are not made; the wave that owns the look-back starts polling behind the fold, when it is needed (a pause of
// 0.2 / 0.6 us in front of that first round: nothing / slower).
constexpr int LB_SHIFT = 8;

constexpr int LB_GROUP = 1 << LB_SHIFT;
#ifndef ESP_LB_SLEEP
#define ESP_LB_SLEEP 2  // (x 64 cycles between two rounds of polls)
#endif
struct LbState {
    u64 group_part;  // totals of the own group's segments in front of this one
    u64 prefix;      // gstatus of the previous group
    bool have_part, have_prefix, finished;
    u32 spins;
    u64 pend_v[LB_GROUP / 64], pend_gp;  // the answers of a round of polls (lb_issue), looked at by lb_consume
};
__device__ __forceinline__ void lb_init(LbState &st, int s) {
    st.group_part = 0;
    st.prefix = 0;
    st.have_part = (s & (LB_GROUP - 1)) == 0;  // first of its group: nobody in front
    st.have_prefix = (s >> LB_SHIFT) == 0;     // first group: no previous one
    st.finished = st.have_part && st.have_prefix;
    st.spins = 0;
}
__device__ __forceinline__ bool lb_last_of_group(const Args &a, int s) { return (s & (LB_GROUP - 1)) == LB_GROUP - 1 || s == a.S - 1; }
// one round of polls: all granule loads are in flight together (lanes beyond the segments in front: ready zeros)
__device__ __forceinline__ void lb_issue(const Args &a, LbState &st, int s, int lane) {
    const int g = s >> LB_SHIFT, gbase = g << LB_SHIFT, need = s - gbase;
#pragma unroll
    for (int c = 0; c < LB_GROUP / 64; c++) {
        st.pend_v[c] = ST_AGG;
        if (!st.have_part && c * 64 + lane < need)
            st.pend_v[c] = __hip_atomic_load(&a.status[gbase + c * 64 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    st.pend_gp = ST_PRE;
    if (!st.have_prefix && lane == 0) st.pend_gp = __hip_atomic_load(&a.gstatus[g - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// ... and what they brought; true: something new was learnt
__device__ __forceinline__ bool lb_consume(LbState &st) {
    bool progress = false;
    if (!st.have_part) {
        bool miss = false;
        u64 sum = 0;
#pragma unroll
        for (int c = 0; c < LB_GROUP / 64; c++) {
            miss = miss || (st.pend_v[c] >> 62) == 0;
            sum += st.pend_v[c] & ST_VAL;
        }
        if (__ballot(miss) == 0ull) {
            // (totals are < 2^32 each and 256 of them < 2^40: two 32-bit DPP sums of the per-lane sums' halves)
            const u32 lo = esp_wave_sum((u32)(sum & 0xFFFFFFull));
            const u32 hi = esp_wave_sum((u32)(sum >> 24));
            st.group_part = ((u64)hi << 24) + (u64)lo;
            st.have_part = true;
            progress = true;
        }
    }
    if (!st.have_prefix) {
        const u64 g0 = esp_uniform_u64(st.pend_gp);
        if ((g0 >> 62) != 0) {
            st.prefix = g0 & ST_VAL;
            st.have_prefix = true;
            progress = true;
        }
    }
    st.finished = st.have_part && st.have_prefix;
    return progress;
}
// at most `iters` polls (block: until resolved)
__device__ __forceinline__ void lb_poll(const Args &a, LbState &st, int s, int lane, u32 iters, bool block) {
    while (!st.finished && (block || iters-- > 0)) {
        lb_issue(a, st, s, lane);
        const bool progress = lb_consume(st);
        if (!st.finished && !progress) {
            if (++st.spins > SPIN_LIMIT) {
                if (lane == 0) atomicOr(a.err, 1u);
                st.finished = true;  // give up (the host reports the error)
                break;
            }
            __builtin_amdgcn_s_sleep(ESP_LB_SLEEP);
        }
    }
}
// the segment's total becomes visible to its group (call once, as early as the total is known)
__device__ __forceinline__ void lb_publish(const Args &a, LbState &st, int s, u32 total, int lane) {
    lb_init(st, s);
    if (lane == 0) __hip_atomic_store(&a.status[s], ST_AGG | (u64)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// resolves what is left, publishes the group's prefix if this is its last segment; returns the exclusive prefix
__device__ __forceinline__ u64 lb_complete(const Args &a, LbState &st, int s, u32 total, int lane) {
    lb_poll(a, st, s, lane, 0, true);
    const u64 excl = st.prefix + st.group_part;
    if (lane == 0) {
        if (lb_last_of_group(a, s))
            __hip_atomic_store(&a.gstatus[s >> LB_SHIFT], ST_PRE | ((excl + (u64)total) & ST_VAL), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // the last segment leaves the grand total where the host reads it (nobody polls its granule)
        if (s == a.S - 1) __hip_atomic_store(&a.status[s], ST_PRE | ((excl + (u64)total) & ST_VAL), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return excl;
}
__device__ __forceinline__ u64 lookback_wave(const Args &a, int s, u32 total, int lane) {
    LbState st;
    lb_publish(a, st, s, total, lane);
    return lb_complete(a, st, s, total, lane);
}
// a segment that emits nothing and is not the last of its group has no use for its offset: it publishes its zero total
// and leaves (the group's last segment always resolves, so the chain over the groups never breaks)
__device__ __forceinline__ bool lb_may_skip(const Args &a, int s) { return !lb_last_of_group(a, s); }

// Register tier of the bucket kernel: one lane per column, the whole run (<= R entries) in registers.
// Returns true when the look-back already ran (early publication of the segment total).
// *dense (fresh matrix, runs of at most 16): the records lie at their dense places already (fold_run_dense); wtot: WAVES LDS words
template <int R, bool FRESH, bool UPD, bool SM>
__device__ __forceinline__ bool reg_tier(const Args &a, u64 *skey, double *sval, const u32 *ccnt, int ncl, int s, u64 hi,
                                         u64 rowmask, u32 *s_early, LbState &lb, bool *dense = nullptr, u32 *wtot = nullptr) {
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (FRESH && a.csc.nnz == 0 && ncl <= THREADS - ESP_WAVE) {
        // Nothing can hit the CSC, so the number of entries a run emits is known right after
        // sorting.  The segment total is published BEFORE the fold and the last wave (idle in
        // this phase: one lane per column) runs the look-back while the others fold.
        u64 x[R];
        double xv[R];
        int rs = 0, len = 0;
        u32 ec = 0;
        if (t < ncl) {
            rs = (int)ccnt[t];
            len = (int)ccnt[t + 1] - rs;
            load_sorted_run<R>(skey, sval, rs, len, x, xv);
            ec = count_emitted<R, UPD>(x, xv, len);
        }
        // (the lanes' counts scanned: a run's records go straight to their dense place -- see fold_run_dense.  The small variant
        // has 80 registers: a run of 16 keys and values does not fit them across the barrier)
#ifdef ESP_NO_DENSE_REG
        constexpr bool DENSE_R = false;
#else
        constexpr bool DENSE_R = SM ? R <= 12 : R <= 16;
#endif
        const u32 ecl = ec;
        const u32 einc = esp_wave_scan_add(ecl);
        ec = (u32)__builtin_amdgcn_readlane((int)einc, 63);
        if (lane == 0) {
            if constexpr (DENSE_R) wtot[w] = ec;
            if (ec) atomicAdd(s_early, ec);
        }
#ifdef ESP_LOCAL_STAMPS
        if (a.stamps && t == 0) a.stamps[(size_t)s * 16 + 8] = wall_clock64();
#endif
        __syncthreads();
#ifdef ESP_LOCAL_STAMPS
        if (a.stamps && t == 0) a.stamps[(size_t)s * 16 + 9] = wall_clock64();
#endif
        if constexpr (DENSE_R) {
            // (every lane holds its run's keys and values in registers: the LDS arrays are free for the records)
            u32 at = einc - ecl;
#pragma unroll
            for (int i = 0; i < WAVES; i++) at += i < w ? wtot[i] : 0u;
            if (w == WAVES - 1) {
                // (no poll here: the wave would reach the barrier behind the fold a round trip late -- HIP's barrier waits for a
                // wave's outstanding loads -- and the whole workgroup with it: one immediate round of polls cost the headline's
                // bucket kernel 0.17 of 1.30 ms; the chain is resolved after the fold, lb_complete)
                lb_publish(a, lb, s, *s_early, lane);
            } else if (t < ncl) {
                fold_run_dense<R, UPD>(skey, sval, x, xv, len, hi, at);
            }
            *dense = true;
            return true;
        }
        if (w == WAVES - 1) {
            // publish, then poll the predecessors a bounded number of times: the chain is usually still
            // open when the others are done folding; the wave carries on with it between the following
            // barriers (nothing before the final stores needs the offset)
            // (measured: polling already while the other waves sort costs more in contention than the
            // shorter chain saves -- local 2.52 vs 2.33 ms)
            lb_publish(a, lb, s, *s_early, lane);
        } else if (t < ncl) {
            // (measured: keeping the sorted keys in registers across the barrier and re-reading
            // only the values beats writing the run back to LDS)
            fold_run<R, true, UPD>(a, skey, sval, x, rs, len, hi, rowmask);
#ifdef ESP_LOCAL_STAMPS
            if (a.stamps && t == 0) a.stamps[(size_t)s * 16 + 10] = wall_clock64();
#endif
        }
        return true;
    }
    for (int c = t; c < ncl; c += THREADS) {
        const int rs = (int)ccnt[c];
        const int len = (int)ccnt[c + 1] - rs;
        u64 x[R];
        // The fold of a run over a stored pattern starts a chain of dependent loads with the column's range.  The column
        // is known from any key of the run: the range is requested BEFORE the run is sorted and arrives while the
        // network runs.  (Requesting the first and the last line of the column's rows and values here as well, so that
        // the walk finds them in the cache: slower, 1.75 -> 1.88 ms for the re-assembly of the 256^3 stencil.)
        i64 ccur = 0, cend = 0;
        if (!FRESH && a.csc.nnz > 0 && len > 0) {
            const i64 col0 = (i64)((hi + (skey[rs] >> SUB_SHIFT)) >> a.rb);
            ccur = a.csc.colptr[col0] - 1;
            cend = a.csc.colptr[col0 + 1] - 1;
        }
        sort_run_keys<R>(skey, rs, len, x);
#ifndef ESP_NO_SHORT_CSC
        if (!FRESH && (!SM || a.expect_hits) && rowmask <= 0xFFFFFFFFull && ccur < cend && cend - ccur <= (i64)CSC_SHORT) {
            if (SM || a.expect_hits)
                fold_run_short_csc<R, UPD, true>(a, skey, sval, x, rs, len, hi, rowmask, ccur, (int)(cend - ccur));
            else if constexpr (!SM)
                fold_run_short_csc<R, UPD, false>(a, skey, sval, x, rs, len, hi, rowmask, ccur, (int)(cend - ccur));
        } else
#endif
            fold_run<R, FRESH, UPD>(a, skey, sval, x, rs, len, hi, rowmask, ccur, cend);  // (a FRESH launch has an empty CSC)
        // (over a stored pattern: did the run leave a record?  A segment whose updates all hit stored positions -- a
        // re-assembly -- emits nothing, and its workgroup leaves right behind the fold: see local_k)
        if (!FRESH && len > 0 && skey[rs] != NOREC) *s_early = 1u;
    }
    return false;
}

// ---- group tier: column runs of 17 .. 256 entries (P1 FEM: 24 per column in 2-D, 120 in 3-D) ---------------------
// After the counting sort by column every run is sorted by G = 2 / 4 / 8 / 16 neighbouring lanes, 16 keys per lane
// in REGISTERS: the lane's 16 keys by the 63-comparator network, then bitonic merges ACROSS the lanes of the group
// whose partner is always reachable by one DPP modifier (quad_perm for lane ^ 1 / ^ 2 / ^ 3, row_half_mirror for
// lane ^ 7, row_mirror for lane ^ 15, two bank-masked row shifts for lane ^ 4) -- no LDS traffic, no barrier inside
// the sort, and a wave sorts 64 / G columns at once.  Sort keys are 32 bits: (row - smallest row of the segment) in
// 18 bits | slot index | kind, i.e. a compare-exchange is v_min_u32 + v_max_u32 (in registers) or the same two on a
// DPP operand + a select (across lanes).  Against the 8-bit LSD radix passes over every varying key bit (three passes
// of ballots, LDS counters and five barriers each for 3-D FEM): 700 instead of about 4500 instructions per wave; the
// ordered fold follows in the same lanes (group_columns).
constexpr int GROUP_MAX = 256;    // longest column run the group tier takes (16 lanes x 16 keys)
constexpr int GROUP_ROW_BITS = 32 - SUB_SHIFT;  // 18: the rows of a segment must span less than 2^18

template <int CTRL>
__device__ __forceinline__ u32 dpp_perm(u32 x) {
    return (u32)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xf, 0xf, false);
}
// lane ^ 4 inside a row of 16: lanes 0-3 / 8-11 read four lanes up, lanes 4-7 / 12-15 four lanes down
__device__ __forceinline__ u32 dpp_xor4(u32 x) {
    int y = __builtin_amdgcn_update_dpp((int)x, (int)x, 0x104 /* row_shl:4 */, 0xf, 0x5, false);
    y = __builtin_amdgcn_update_dpp(y, (int)x, 0x114 /* row_shr:4 */, 0xf, 0xa, false);
    return (u32)y;
}
__device__ __forceinline__ u32 keep_side(u32 x, u32 y, bool lower) {
    const u32 lo = x < y ? x : y, hi = x < y ? y : x;
    return lower ? lo : hi;
}
// first step of a merge of two sorted blocks: element e against its mirror image in the block pair (partner lane
// through CTRL, register R - 1 - r); the lower block keeps the minima
template <int CTRL, int R>
__device__ __forceinline__ void group_mirror(u32 (&x)[R], bool lower) {
#pragma unroll
    for (int r = 0; r < R / 2; r++) {
        const u32 a = dpp_perm<CTRL>(x[R - 1 - r]), b = dpp_perm<CTRL>(x[r]);
        x[r] = keep_side(x[r], a, lower);
        x[R - 1 - r] = keep_side(x[R - 1 - r], b, lower);
    }
}
template <int CTRL, int R>
__device__ __forceinline__ void group_cross(u32 (&x)[R], bool lower) {
#pragma unroll
    for (int r = 0; r < R; r++) x[r] = keep_side(x[r], dpp_perm<CTRL>(x[r]), lower);
}
template <int R>
__device__ __forceinline__ void group_cross4(u32 (&x)[R], bool lower) {
#pragma unroll
    for (int r = 0; r < R; r++) x[r] = keep_side(x[r], dpp_xor4(x[r]), lower);
}
// the half cleaners inside a lane (distances R/2 .. 1 between registers)
template <int R>
__device__ __forceinline__ void group_clean(u32 (&x)[R]) {
#pragma unroll
    for (int j = R / 2; j >= 1; j >>= 1)
#pragma unroll
        for (int r = 0; r < R; r++)
            if ((r & j) == 0) {
                const u32 lo = x[r] < x[r | j] ? x[r] : x[r | j], hi = x[r] < x[r | j] ? x[r | j] : x[r];
                x[r] = lo;
                x[r | j] = hi;
            }
}
// sorts the R * G keys held by G neighbouring lanes (lane q of the group: sorted positions R q .. R q + R - 1).
// R = 16 keys per lane, or 8: twice the lanes per column -- a segment of 32 columns x 120 entries (3-D FEM) then keeps all
// eight waves of the workgroup busy instead of four, and a lane's 8 keys + 8 values leave room in the register file
template <int G, int R>
__device__ __forceinline__ void group_sort(u32 (&x)[R], int q) {
#pragma unroll
    for (int c = 0; c < NetOf<R>::net.n; c++) {
        const u32 lo = x[NetOf<R>::net.a[c]], hi = x[NetOf<R>::net.b[c]];
        x[NetOf<R>::net.a[c]] = lo < hi ? lo : hi;
        x[NetOf<R>::net.b[c]] = lo < hi ? hi : lo;
    }
    if constexpr (G >= 2) {
        group_mirror<0xB1, R>(x, (q & 1) == 0);  // quad_perm [1,0,3,2]: lane ^ 1
        group_clean<R>(x);
    }
    if constexpr (G >= 4) {
        group_mirror<0x1B, R>(x, (q & 2) == 0);  // quad_perm [3,2,1,0]: lane ^ 3
        group_cross<0xB1, R>(x, (q & 1) == 0);
        group_clean<R>(x);
    }
    if constexpr (G >= 8) {
        group_mirror<0x141, R>(x, (q & 4) == 0);  // row_half_mirror: lane ^ 7
        group_cross<0x4E, R>(x, (q & 2) == 0);    // quad_perm [2,3,0,1]: lane ^ 2
        group_cross<0xB1, R>(x, (q & 1) == 0);
        group_clean<R>(x);
    }
    if constexpr (G >= 16) {
        group_mirror<0x140, R>(x, (q & 8) == 0);  // row_mirror: lane ^ 15
        group_cross4<R>(x, (q & 4) == 0);
        group_cross<0x4E, R>(x, (q & 2) == 0);
        group_cross<0xB1, R>(x, (q & 1) == 0);
        group_clean<R>(x);
    }
}
// ---- sort + ordered fold of every column run of the segment by its group of G lanes -----------------------------
// in : skey[ccnt[c] .. ccnt[c+1]) packed keys of column c in any order, sval[slot] the values
// out: the records of column c at the front of its run (skey[rs + e] = sub << SUB_SHIFT | slot << 2, sval[slot] = value),
//      NOREC behind them -- what every tier of the bucket kernel leaves for the compaction
// After the sort a lane holds 16 consecutive sorted entries of its column in registers and gathers their values.
// PASS A folds, without a branch per entry, every (col,row) that STARTS among the lane's entries (running fold, reset at
// every first entry of a (col,row)); a (col,row) that ends in the lane is closed there.  A (col,row) that runs on into
// the next lanes -- the diagonal of a 3-D FEM column holds 48 updates: three or four lanes -- is handed over: PASS B,
// a few rounds: a lane whose first entries continue its neighbour's (col,row) takes the neighbour's accumulator
// through DPP once that is final, folds those entries behind it (left to right: the stream's order) and either closes
// the (col,row) or, when all its 16 entries belong to it, hands it on.  Everything stays in registers; LDS sees the
// gather, the records and nothing else; no barrier.
constexpr int GROUP_CL_BITS = 11;
// UPDATE / RAWUPDATE entries only: an absent position holds +0.0 and +0.0 + v is the value a creating update leaves
// (fold_step_update); a RAWUPDATE creates whatever its value
__device__ __forceinline__ void group_step(bool &present, double &acc, u32 kind, double v, bool adds, bool raws) {
    if (adds) {
        acc = acc + v;
        present = present || raws || v != 0.0;
    } else {
        espfold::fold_step_sel(present, acc, kind, v);
    }
}
// closes a (col,row): stored position -> in place; else a record whose value sits in the slot of entry `key`
template <bool FRESH>
__device__ __forceinline__ bool group_close(const Args &a, double *sval, i64 pos, bool present, double acc, u32 key) {
    if (!FRESH && pos >= 0) {
        if (a.mode == ESP_FLUSH_ROUTED)
            a.csc.nzval[pos] = acc;
        else if (present)
            a.csc.nzval[pos] = a.csc.nzval[pos] + acc;  // csc operand first, sparsematrixlnk.jl:363
        return false;
    }
    if (present) sval[(key >> ESP_TAG_BITS) & (CAP - 1)] = acc;  // (a slot of one of its own entries: nobody else's)
    return present;
}
__device__ __forceinline__ double dpp_shr1_f64(double x) {
    const long long b = __double_as_longlong(x);
    const u32 lo = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)b, 0x111, 0xf, 0xf, true);
    const u32 hi = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)(b >> 32), 0x111, 0xf, 0xf, true);
    return __longlong_as_double((long long)(((u64)hi << 32) | lo));
}
// MODE: 0 = entries of any kind; 1 = all UPDATE, 2 = all RAWUPDATE (known to the kernel's instantiation or to the host's
// bookkeeping): the fold is an addition, and on a fresh matrix pass A has no branch at all
// DENSE (fresh matrix, additions, every column of the segment has its group of lanes at once): the records stay in
// registers; the columns' record counts are scanned and every lane writes its records straight to the segment's dense
// output arrays in LDS -- no record slots, no NOREC marks, no generic compaction over the segment's 4096 slots afterwards --
// and the segment's total is published for the look-back right after the fold (the last wave owns it from there on).
// Returns true when it did so: the caller goes on at the final stores.
struct DenseCtx {
    u32 *s_early;          // LDS word, zeroed at the kernel's start: the segment's total
    unsigned short *ctot;  // LDS, one per column of the segment (<= DENSE_COLS)
    LbState *lb;
    int s;
    // RAWUPDATEs on a fresh matrix: a column emits one record per distinct row, known as soon as its run is SORTED -- the
    // segment's total is published then, by whichever wave counts its columns last (an LDS word counts the waves), and the
    // look-back's round trips run behind the gather and the fold instead of behind them (3-D FEM: the last wave used to wait
    // 2.3 us of a segment's 15 for totals that its neighbours had published 1.6 us earlier).  nullptr: after the fold.
    u32 *s_done = nullptr;
};
constexpr int DENSE_COLS = 512;
// KT = u32 (the three-workgroup kernel group3_k, dense form only): the LDS key array holds the 32-bit SORT keys already,
// and takes the records as (local column << rb) | row.
// WIDE (group3_k's second instantiation: rows of a segment spread over more than 2^18 -- a mesh whose node numbering has no
// locality): srow[slot] holds every entry's full row (relative to the segment's smallest), the sort keys its low 18 bits; the
// run is sorted TWICE with the same network -- by (low bits, slot), then by (high bits, rank of the low-bit group, slot): the
// keys are unique through their slot, so two unstable sorts compose like a stable two-digit LSD sort -- and a record takes
// its row from srow.  Rows may span 2^29.
constexpr int WIDE_GID_BITS = 7;                                   // rank of a low-bit group inside its column run (<= 128 entries)
constexpr int WIDE_HI_BITS = 32 - SUB_SHIFT - WIDE_GID_BITS;       // 11 high row bits in the second sort key
// HITS (group3_k's re-assembly form: a ROUTED flush of additions over a stored pattern that the SAME mesh built): the k-th
// (col,row) of a column's sorted run is the column's k-th stored entry and the run covers them all -- checked, entry by entry
// and count by count; a column where that does not hold raises bit 64 of Args::err.  The fold is the branch-free fresh-matrix
// one with the stored value as the start of every (col,row)'s sum; the sums go to a SECOND value array (Args::hits_out) at
// the stored positions: the host swaps the arrays when no flag came back, else nothing has happened and the flush takes
// the general kernels.  No records, no look-back, no output stores.
template <int G, int R, int CAPK, bool FRESH, int MODE, bool DENSE = false, typename KT = u64, bool WIDE = false, bool HITS = false>
__device__ __forceinline__ bool group_columns(const Args &a, KT *skey, double *sval, const u32 *ccnt, int ncl, u32 rmin, u64 hi,
                                              u64 rowmask, unsigned long long *stamp, const DenseCtx *dc = nullptr, const u32 *srow = nullptr) {
    static_assert(!DENSE || (FRESH && MODE != 0), "the dense form is the fresh-matrix addition fold");
    static_assert(!HITS || (DENSE && sizeof(KT) == 4), "the re-assembly form: group3_k only");
    constexpr bool K32L = sizeof(KT) == 4;
    static_assert(!K32L || DENSE, "32-bit LDS keys: dense form only");
    static_assert(!WIDE || (K32L && R * G <= (1 << WIDE_GID_BITS)), "wide rows: the three-workgroup kernel's shapes only");
    const int t = threadIdx.x, q = t & (G - 1), lane = t & (ESP_WAVE - 1);
    constexpr int CPB = THREADS / G;  // columns the workgroup takes at a time
    constexpr u32 LOWMASK = (1u << SUB_SHIFT) - 1u;
    constexpr bool raws = MODE == 2, adds = MODE != 0;
    for (int c0 = 0; c0 < ncl; c0 += CPB) {
        // (the whole wave has no column left: all 64 lanes leave together -- not in the dense form: its barriers are for all)
        if (!DENSE && c0 + (t & ~(ESP_WAVE - 1)) / G >= ncl) break;
        const int c = c0 + t / G;
        int rs = 0, len = 0;
        if (c < ncl) {
            rs = (int)ccnt[c];
            len = (int)ccnt[c + 1] - rs;
        }
        const int lastj = len > 0 ? len - 1 : 0;
        u32 x[R];
        // (the network does not care where an entry starts: the group's lanes read neighbouring slots -- with 16
        // consecutive slots per lane all lanes of a group would meet in one LDS bank)
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int p = r * G + q;
            if constexpr (K32L) {
                const u32 kk = skey[min(rs + min(p, lastj), CAPK - 1)];
                x[r] = p < len ? kk : ~0u;
            } else {
                const u64 kk = skey[min(rs + min(p, lastj), CAPK - 1)];  // (all reads in flight; clamped, never past the array)
                const u32 rel = (u32)((kk >> SUB_SHIFT) & rowmask) - rmin;
                x[r] = p < len ? ((rel << SUB_SHIFT) | ((u32)kk & LOWMASK)) : ~0u;
            }
        }
        group_sort<G, R>(x, q);
        if constexpr (WIDE) {
            // rank of every entry's low-bit group inside the run (the number of group starts up to it, minus one) ...
            const int nv1 = max(0, min(R, len - q * R));
            const u32 before1 = (u32)__builtin_amdgcn_update_dpp(0, (int)x[R - 1], 0x111 /* row_shr:1 */, 0xf, 0xf, true);
            u32 h1 = 0;
#pragma unroll
            for (int r = 0; r < R; r++) {
                const u32 prev = r == 0 ? before1 : x[r - 1];
                const bool hh = r < nv1 && ((r == 0 && q == 0) || (prev >> SUB_SHIFT) != (x[r] >> SUB_SHIFT));
                h1 |= hh ? 1u << r : 0u;
            }
            const u32 mine1 = (u32)__popc(h1);
            u32 inc1 = mine1;
            if constexpr (G >= 2) {
                const u32 o = (u32)__builtin_amdgcn_update_dpp(0, (int)inc1, 0x111, 0xf, 0xf, true);
                inc1 += q >= 1 ? o : 0u;
            }
            if constexpr (G >= 4) {
                const u32 o = (u32)__builtin_amdgcn_update_dpp(0, (int)inc1, 0x112, 0xf, 0xf, true);
                inc1 += q >= 2 ? o : 0u;
            }
            if constexpr (G >= 8) {
                const u32 o = (u32)__builtin_amdgcn_update_dpp(0, (int)inc1, 0x114, 0xf, 0xf, true);
                inc1 += q >= 4 ? o : 0u;
            }
            if constexpr (G >= 16) {
                const u32 o = (u32)__builtin_amdgcn_update_dpp(0, (int)inc1, 0x118, 0xf, 0xf, true);
                inc1 += q >= 8 ? o : 0u;
            }
            const u32 base1 = inc1 - mine1;
            // ... and the second sort: (high row bits, that rank, slot)
#pragma unroll
            for (int r = 0; r < R; r++) {
                const u32 gid = base1 + (u32)__popc(h1 & ((2u << r) - 1u)) - 1u;
                const u32 slot_kind = x[r] & LOWMASK;
                const u32 rowfull = srow[(x[r] >> ESP_TAG_BITS) & (CAP - 1)];
                x[r] = r < nv1 ? (((rowfull >> GROUP_ROW_BITS) << (SUB_SHIFT + WIDE_GID_BITS)) | ((gid & ((1u << WIDE_GID_BITS) - 1u)) << SUB_SHIFT) | slot_kind) : ~0u;
            }
            group_sort<G, R>(x, q);
        }
#ifdef ESP_LOCAL_STAMPS
        if (stamp && t == 0 && c0 == 0) stamp[13] = wall_clock64();
#endif
        double v[R];
#pragma unroll
        for (int r = 0; r < R; r++) v[r] = sval[(x[r] >> ESP_TAG_BITS) & (CAP - 1)];  // (padding reads the last slot)
        const int nv = max(0, min(R, len - q * R));               // this lane's entries
        const int nvn = q == G - 1 ? 0 : max(0, min(R, len - (q + 1) * R));  // the next lane's
        // first entry of its (col,row)?  (the entry in front of a lane's first one sits in the lane before it)
        const u32 before = (u32)__builtin_amdgcn_update_dpp(0, (int)x[R - 1], 0x111 /* row_shr:1 */, 0xf, 0xf, true);
        u32 heads = 0;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const u32 prev = r == 0 ? before : x[r - 1];
            const bool h = r < nv && ((r == 0 && q == 0) || (prev >> SUB_SHIFT) != (x[r] >> SUB_SHIFT));
            heads |= h ? 1u << r : 0u;
        }
        bool early = false;
        if constexpr (DENSE && raws && !HITS) {
            if (dc->s_done) {
                early = true;
                const u32 mine0 = (u32)__popc(heads);
                u32 inc0 = mine0;
                if constexpr (G >= 2) {
                    const u32 o = (u32)__builtin_amdgcn_update_dpp(0, (int)inc0, 0x111, 0xf, 0xf, true);
                    inc0 += q >= 1 ? o : 0u;
                }
                if constexpr (G >= 4) {
                    const u32 o = (u32)__builtin_amdgcn_update_dpp(0, (int)inc0, 0x112, 0xf, 0xf, true);
                    inc0 += q >= 2 ? o : 0u;
                }
                if constexpr (G >= 8) {
                    const u32 o = (u32)__builtin_amdgcn_update_dpp(0, (int)inc0, 0x114, 0xf, 0xf, true);
                    inc0 += q >= 4 ? o : 0u;
                }
                if constexpr (G >= 16) {
                    const u32 o = (u32)__builtin_amdgcn_update_dpp(0, (int)inc0, 0x118, 0xf, 0xf, true);
                    inc0 += q >= 8 ? o : 0u;
                }
                if (q == G - 1 && c < ncl) {
                    dc->ctot[c] = (unsigned short)inc0;
                    if (inc0) atomicAdd(dc->s_early, inc0);
                }
                // (the LDS unit takes a wave's operations in order: when this wave's count of the waves comes back as the
                // last one, every wave's column totals have been added)
                if (lane == 0 && atomicAdd(dc->s_done, 1u) == (u32)(WAVES - 1)) {
                    const u32 tot0 = atomicAdd(dc->s_early, 0u);
                    __hip_atomic_store(&a.status[dc->s], ST_AGG | (u64)tot0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        const int f = heads ? (int)__builtin_ctz(heads) : nv;  // entries in front of the lane's first own (col,row)
        const u32 next_head0 = (u32)__builtin_amdgcn_update_dpp(0, (int)(heads & 1u), 0x101 /* row_shl:1 */, 0xf, 0xf, true);
        // the (col,row) of the lane's last entry goes on in the next lane
        const bool open_end = nv == R && nvn > 0 && !next_head0;
        const u64 colbase = hi + ((u64)c << a.rb);
        u32 emit = 0;
        if constexpr (FRESH && MODE != 0) {
            // HITS: rank of the lane's first (col,row) inside the run, the stored value every (col,row) starts from
            i64 cstart = 0, cend = 0;
            u32 gbase = 0;
            double cst[R];
            if constexpr (HITS) {
                const i64 col = (i64)(colbase >> a.rb);
                if (c < ncl && col < a.n_cols) {
                    cstart = a.csc.colptr[col] - 1;
                    cend = a.csc.colptr[col + 1] - 1;
                }
                const u32 mineh = (u32)__popc(heads);
                u32 inch = mineh;
                if constexpr (G >= 2) {
                    const u32 o = (u32)__builtin_amdgcn_update_dpp(0, (int)inch, 0x111, 0xf, 0xf, true);
                    inch += q >= 1 ? o : 0u;
                }
                if constexpr (G >= 4) {
                    const u32 o = (u32)__builtin_amdgcn_update_dpp(0, (int)inch, 0x112, 0xf, 0xf, true);
                    inch += q >= 2 ? o : 0u;
                }
                if constexpr (G >= 8) {
                    const u32 o = (u32)__builtin_amdgcn_update_dpp(0, (int)inch, 0x114, 0xf, 0xf, true);
                    inch += q >= 4 ? o : 0u;
                }
                if constexpr (G >= 16) {
                    const u32 o = (u32)__builtin_amdgcn_update_dpp(0, (int)inch, 0x118, 0xf, 0xf, true);
                    inch += q >= 8 ? o : 0u;
                }
                gbase = inch - mineh;
                bool bad = q == G - 1 && c < ncl && (i64)inch != cend - cstart;  // (the run covers the column's stored entries, all of them)
#pragma unroll
                for (int r = 0; r < R; r++) {
                    cst[r] = 0.0;
                    if ((heads >> r) & 1u) {
                        const i64 pos = cstart + (i64)gbase + (i64)__popc(heads & ((1u << r) - 1u));
                        u32 rowrel;
                        if constexpr (WIDE)
                            rowrel = srow[(x[r] >> ESP_TAG_BITS) & (CAP - 1)];
                        else
                            rowrel = x[r] >> SUB_SHIFT;
                        const i64 row1 = (i64)(((colbase & rowmask) + (u64)rowrel + (u64)rmin)) + 1;
                        if (pos < cend && a.csc.rowval[pos] == row1)
                            cst[r] = a.csc.nzval[pos];
                        else
                            bad = true;
                    }
                }
                if (bad) atomicOr(a.err, 64u);
            }
            // ---- pass A, additions on a fresh matrix: running sums, restarted at every first entry of a (col,row); every
            // entry's slot takes the sum up to it -- the slot of a (col,row)'s LAST entry is its record's value slot
            double acc = 0.0;
            u32 np = 0, pres = 0;  // UPDATEs: a (col,row) is present once one of its values is not zero
#pragma unroll
            for (int r = 0; r < R; r++) {
                const bool is_head = (heads >> r) & 1u;
                if constexpr (HITS)
                    acc = (is_head ? cst[r] : acc) + v[r];  // (a stored position is present: old + v, whatever v is)
                else
                    acc = (is_head ? 0.0 : acc) + v[r];
                if constexpr (DENSE) {
                    v[r] = r >= f ? acc : v[r];  // (the values in front of the lane's first own (col,row) are pass B's)
                } else {
                    if (r < nv) sval[(x[r] >> ESP_TAG_BITS) & (CAP - 1)] = acc;
                }
                if constexpr (!raws) {
                    const u32 nz = v[r] != 0.0 ? 1u : 0u;
                    np = is_head ? nz : (np | nz);
                    pres |= np << r;
                }
            }
            // last entries of the (col,row)s that started in this lane: the entry in front of a first entry, and the
            // lane's last entry unless its (col,row) goes on
            const u32 valid = (1u << nv) - 1u;
            const u32 lastbit = (nv > 0 && !open_end) ? 1u << (nv - 1) : 0u;
            const u32 tails = ((heads >> 1) | lastbit) & valid & ~((1u << f) - 1u);
            emit = (raws || HITS) ? tails : (tails & pres);
            // ---- pass B
            double t_acc = acc;
            u32 t_present = (raws || HITS) ? 1u : np;
            bool final = f < R, need = f > 0;
            while (__ballot(need) != 0ull) {
                const double in_acc = dpp_shr1_f64(t_acc);
                const u32 in_present = (u32)__builtin_amdgcn_update_dpp(0, (int)t_present, 0x111, 0xf, 0xf, true);
                const bool in_final = __builtin_amdgcn_update_dpp(0, (int)final, 0x111, 0xf, 0xf, true) != 0;
                if (need && in_final) {
                    double acc2 = in_acc;
                    u32 p2 = in_present;
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        const double a3 = acc2 + v[r];
                        acc2 = r < f ? a3 : acc2;
                        if constexpr (!raws) p2 |= (r < f && v[r] != 0.0) ? 1u : 0u;
                    }
                    if (f == R && open_end) {  // every entry of the lane belongs to it and it goes on
                        t_acc = acc2, t_present = p2;
                        final = true;
                    } else if (p2) {
                        if constexpr (DENSE)
                            v[0] = acc2;
                        else
                            sval[(x[0] >> ESP_TAG_BITS) & (CAP - 1)] = acc2;
                        emit |= 1u;  // (entry 0 stands for it: the same row, a slot of its own, in front of the lane's other records)
                    }
                    need = false;
                }
            }
            if constexpr (HITS) {
                // the sums to the second value array, at the stored positions: the (col,row) that ends at entry r has the rank
                // gbase + (first entries up to r) - 1 -- also the one that came in from the lane before (no first entry up to r)
#pragma unroll
                for (int r = 0; r < R; r++) {
                    if ((emit >> r) & 1u) {
                        const i64 pos = cstart + (i64)gbase + (i64)__popc(heads & ((2u << r) - 1u)) - 1;
                        if (pos >= cstart && pos < cend) a.hits_out[pos] = v[r];
                    }
                }
                continue;  // (the next batch of columns, if the segment has more than the workgroup takes at a time: it has not)
            }
        } else {
            // ---- pass A
            double acc = 0.0;
            bool present = false;
            i64 pos = -1;
            // A re-assembly over the pattern the same mesh built: the k-th (col,row) of a column's sorted run IS the column's
            // k-th stored entry.  So every first entry of a (col,row) looks there first -- its rank among the run's (col,row)s
            // from a scan of the lanes' counts, one round trip (the column's start, then row and value side by side with the
            // neighbouring lanes') -- and searches the column only when that guess does not hold (new couplings, a column the
            // batch covers in part): the binary search per (col,row), four or five dependent loads each, took 11 of a 3-D FEM
            // segment's 27 us
            i64 cstart = 0, cend = 0;
            u32 gbase = 0;
            if constexpr (!FRESH) {
                if (a.csc.nnz > 0 && c < ncl) {
                    const i64 col = (i64)(colbase >> a.rb);
                    if (col < a.n_cols) {  // (the last segment's column block may reach past the matrix)
                        cstart = a.csc.colptr[col] - 1;
                        cend = a.csc.colptr[col + 1] - 1;
                    }
                }
                const u32 mineh = (u32)__popc(heads);
                u32 inch = mineh;
                if constexpr (G >= 2) {
                    const u32 o = (u32)__builtin_amdgcn_update_dpp(0, (int)inch, 0x111, 0xf, 0xf, true);
                    inch += q >= 1 ? o : 0u;
                }
                if constexpr (G >= 4) {
                    const u32 o = (u32)__builtin_amdgcn_update_dpp(0, (int)inch, 0x112, 0xf, 0xf, true);
                    inch += q >= 2 ? o : 0u;
                }
                if constexpr (G >= 8) {
                    const u32 o = (u32)__builtin_amdgcn_update_dpp(0, (int)inch, 0x114, 0xf, 0xf, true);
                    inch += q >= 4 ? o : 0u;
                }
                if constexpr (G >= 16) {
                    const u32 o = (u32)__builtin_amdgcn_update_dpp(0, (int)inch, 0x118, 0xf, 0xf, true);
                    inch += q >= 8 ? o : 0u;
                }
                gbase = inch - mineh;
            }
#pragma unroll
            for (int r = 0; r < R; r++) {
                const bool is_head = (heads >> r) & 1u;
                if constexpr (FRESH) {
                    acc = is_head ? 0.0 : acc;
                    present = is_head ? false : present;
                } else {
                    if (is_head) {
                        acc = 0.0, present = false, pos = -1;
                        if (a.csc.nnz > 0) {
                            const u64 full = colbase + (u64)((x[r] >> SUB_SHIFT) + rmin);
                            const i64 row1 = (i64)(full & rowmask) + 1;
                            const i64 guess = cstart + (i64)gbase + (i64)__popc(heads & ((1u << r) - 1u));
                            if (guess < cend && a.csc.rowval[guess] == row1)
                                pos = guess;
                            else
                                pos = espfold::csc_find(a.csc, (i64)(full >> a.rb), (i64)(full & rowmask));
                            present = pos >= 0 && a.mode == ESP_FLUSH_ROUTED;
                            acc = present ? a.csc.nzval[pos] : 0.0;
                        }
                    }
                }
                group_step(present, acc, x[r] & (u32)ESP_TAG_MASK, v[r], adds, raws);  // (entries in front of the first own one: discarded)
                const bool nexth = r == R - 1 ? false : ((heads >> (r + 1)) & 1u) != 0u;
                const bool tail = r >= f && r < nv && (r == nv - 1 ? !open_end : nexth);
                if (tail && group_close<FRESH>(a, sval, pos, present, acc, x[r])) emit |= 1u << r;
            }
            // ---- pass B: what the lane hands on (final at once when its last (col,row) started in the lane itself)
            double t_acc = acc;
            bool t_present = present;
            i64 t_pos = pos;
            bool final = f < R;
            bool need = f > 0;  // (q > 0: the first lane's first entry starts a (col,row))
            while (__ballot(need) != 0ull) {
                const double in_acc = dpp_shr1_f64(t_acc);
                const bool in_present = __builtin_amdgcn_update_dpp(0, (int)t_present, 0x111, 0xf, 0xf, true) != 0;
                const bool in_final = __builtin_amdgcn_update_dpp(0, (int)final, 0x111, 0xf, 0xf, true) != 0;
                i64 in_pos = -1;
                if constexpr (!FRESH) in_pos = (i64)__double_as_longlong(dpp_shr1_f64(__longlong_as_double((long long)t_pos)));
                if (need && in_final) {
                    double acc2 = in_acc;
                    bool present2 = in_present;
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        double a3 = acc2;
                        bool p3 = present2;
                        group_step(p3, a3, x[r] & (u32)ESP_TAG_MASK, v[r], adds, raws);
                        acc2 = r < f ? a3 : acc2;
                        present2 = r < f ? p3 : present2;
                    }
                    if (f == R && open_end) {  // every entry of the lane belongs to it and it goes on
                        t_acc = acc2, t_present = present2, t_pos = in_pos;
                        final = true;
                    } else if (group_close<FRESH>(a, sval, in_pos, present2, acc2, x[0])) {
                        emit |= 1u;  // (entry 0 stands for it: the same row, a slot of its own, in front of the lane's other records)
                    }
                    need = false;
                }
            }
        }
#ifdef ESP_LOCAL_STAMPS
        if (stamp && t == 0 && c0 == 0) stamp[14] = wall_clock64();
#endif
        __builtin_amdgcn_wave_barrier();  // (every key of the wave's runs has been read: their slots take the records)
        // records to the front of the run: exclusive scan of the lanes' record counts inside the group
        const u32 mine = (u32)__popc(emit);
        u32 inc = mine;
        if constexpr (G >= 2) {
            const u32 o = (u32)__builtin_amdgcn_update_dpp(0, (int)inc, 0x111, 0xf, 0xf, true);
            inc += q >= 1 ? o : 0u;
        }
        if constexpr (G >= 4) {
            const u32 o = (u32)__builtin_amdgcn_update_dpp(0, (int)inc, 0x112, 0xf, 0xf, true);
            inc += q >= 2 ? o : 0u;
        }
        if constexpr (G >= 8) {
            const u32 o = (u32)__builtin_amdgcn_update_dpp(0, (int)inc, 0x114, 0xf, 0xf, true);
            inc += q >= 4 ? o : 0u;
        }
        if constexpr (G >= 16) {
            const u32 o = (u32)__builtin_amdgcn_update_dpp(0, (int)inc, 0x118, 0xf, 0xf, true);
            inc += q >= 8 ? o : 0u;
        }
        const int total = (int)__shfl((int)inc, lane | (G - 1), ESP_WAVE);
        int e = (int)(inc - mine);
        const u64 colpart = (u64)c << a.rb;
        if constexpr (DENSE) {
            if (!early && q == G - 1 && c < ncl) {
                dc->ctot[c] = (unsigned short)total;
                if (total) atomicAdd(dc->s_early, (u32)total);
            }
            __syncthreads();  // every run of the segment is folded, its records in registers: skey / sval are free
#ifdef ESP_LOCAL_STAMPS
            if (stamp && t == 0) stamp[3] = wall_clock64();
#endif
            const int w = t >> 6;
            if (w == WAVES - 1) {  // the look-back starts here; the wave carries it on after its own dense writes
                if (early)
                    lb_init(*dc->lb, dc->s);  // (the total is out already)
                else
                    lb_publish(a, *dc->lb, dc->s, *dc->s_early, lane);
                // (no poll in front of a barrier, see reg_tier)
            }
            if (w == 0) {  // exclusive scan of the columns' record counts (<= 512 columns: eight per lane)
                u32 pre[DENSE_COLS / ESP_WAVE], run = 0;
#pragma unroll
                for (int j = 0; j < DENSE_COLS / ESP_WAVE; j++) {
                    const int cc = lane * (DENSE_COLS / ESP_WAVE) + j;
                    pre[j] = run;
                    run += cc < ncl ? (u32)dc->ctot[cc] : 0u;
                }
                const u32 base = esp_wave_scan_add(run) - run;
#pragma unroll
                for (int j = 0; j < DENSE_COLS / ESP_WAVE; j++) {
                    const int cc = lane * (DENSE_COLS / ESP_WAVE) + j;
                    if (cc < ncl) dc->ctot[cc] = (unsigned short)(base + pre[j]);
                }
            }
            __syncthreads();
#ifdef ESP_LOCAL_STAMPS
            if (stamp && t == 0) stamp[5] = wall_clock64();
#endif
            int d = (c < ncl ? (int)dc->ctot[c] : 0) + e;
#pragma unroll
            for (int r = 0; r < R; r++) {
                if ((emit >> r) & 1u) {
                    if constexpr (WIDE)
                        skey[d] = (u32)colpart | (srow[(x[r] >> ESP_TAG_BITS) & (CAP - 1)] + rmin);
                    else if constexpr (K32L)
                        skey[d] = (u32)colpart | ((x[r] >> SUB_SHIFT) + rmin);
                    else
                        skey[d] = hi + (colpart | (u64)((x[r] >> SUB_SHIFT) + rmin));
                    sval[d] = v[r];
                    d++;
                }
            }
            return true;
        }
        if constexpr (!K32L) {
#pragma unroll
            for (int r = 0; r < R; r++) {
                if ((emit >> r) & 1u) {
                    skey[rs + e] = ((colpart | (u64)((x[r] >> SUB_SHIFT) + rmin)) << SUB_SHIFT) | (u64)(x[r] & (LOWMASK & ~(u32)ESP_TAG_MASK));
                    e++;
                }
            }
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int p = q * R + r;
                if (p >= total && p < len) skey[rs + p] = NOREC;
            }
        }
    }
    return false;
}

// BIG: the kernel also carries the 24-input register tier.  It is a separate instantiation because the
// extra code costs the common path registers (measured: +25 % on the 12-input tier when both live in one
// kernel); the host picks it for a handle whose last flush met runs of 17..24 (a.maxrun_seen).
// KEYS (0 with PIECES): 0 = packed 8-byte keys; 1 = keys_in holds 4-byte keys -- the key bits below the segment's
// prefix -- and every entry has the kind a.kind32 (the run-based partition writes them when all pending entries
// share one kind); 2 = the same and that kind is UPDATE (an assembly loop of updateindex! calls): the register
// tiers fold without decoding a kind; 3 = packed keys whose kinds are all UPDATE (the pieces of a shard whose
// received blocks were checked): the same fold; 4 / 5 = pieces of which one holds 4-byte keys (5: all UPDATE -- the others' packed keys are narrowed to
// 4-byte ones as they are loaded and the kernel is the 4-byte-key one: it is bound by the instructions it issues, and decoding
// packed keys was a tenth of them); 6 / 7 = pieces
// that ALL hold 4-byte keys of the kind a.kind32 (a producer's batch and the tail behind it; 7: UPDATE)
// SMALL: segments of at most 3072 entries (6 per thread) over at most 256 columns, no radix tier: 51 KiB of LDS instead
// of 74, i.e. THREE workgroups per CU (measured at 256^3: one workgroup per CU 2.70 ms, two 1.70 ms, three 1.48 ms).
// The host picks it from what it knows before the flush; a segment whose column runs turn out longer than the register
// tiers take is served by a slow tier (one lane per column: insertion sort of its run in LDS, sequential fold), and
// the longest run it reports sends the handle's next flushes to the regular kernel.
// GRP: the instantiation that carries the GROUP tier (column runs of 17 .. 256: 2 .. 16 lanes per column) INSTEAD of the
// register tiers -- a kernel of its own for the same reason as BIG: with both families inlined the register allocation of
// the common phases paid for the widest of them (40 .. 120 bytes of scratch per lane in the fresh-matrix kernels, 200 in the
// stored-CSC ones).  The host picks it for a matrix whose columns hold more than 24 entries (the longest run its last
// flush met; without history: pending entries per column).  Runs of at most 16 go through the tier's 2-lane form.
// SHORTG (with GRP): the group-tier kernel for runs of at most 32 entries (2-D P1 FEM: 24 per column) -- four lanes x 8
// keys per column, so that the 128 columns of such a segment keep all eight waves busy where the 24-input register tier
// works in two of them; again an instantiation of its own (with the shapes for long runs in the same kernel the 3-D path
// paid 1.3 ms for their registers).  A segment with a longer run goes through the radix tier.
template <bool FRESH, bool PIECES, bool BIG, int KEYS, bool SMALL = false, bool GRP = false, bool SHORTG = false>
__global__ __launch_bounds__(THREADS, SMALL ? 6 : 4) void local_k(Args a) {
    static_assert(!GRP || (!PIECES && !BIG && !SMALL), "the group-tier kernel exists for the plain regular form only");
    static_assert(!SHORTG || GRP, "SHORTG is a form of the group-tier kernel");
    // (MIX = KEYS 5: every entry is an UPDATE, so the packed keys of the other pieces are narrowed to the 4-byte form as they are
    // loaded -- the bits below the segment's prefix, checked against the segment there -- and the kernel is the 4-byte-key one)
    constexpr bool MIX = KEYS == 5;
    constexpr bool K32 = KEYS == 1 || KEYS == 2 || KEYS == 6 || KEYS == 7 || MIX, UPD = KEYS == 2 || KEYS == 3 || KEYS == 5 || KEYS == 7,
                   P32 = KEYS == 4;
    static_assert(PIECES == (KEYS >= 3) || KEYS == 0, "KEYS 3 .. 7 are piece formats, 1 / 2 are not");
    constexpr int NI = SMALL ? 6 : ITEMS;
    constexpr int CAPK = THREADS * NI;
    __shared__ u64 skey[CAPK];
    __shared__ double sval[CAPK];
    // radix tail: cnt[WAVES][256]; column tiers: ccnt[CL_MAX+1] (same storage)
    __shared__ u32 cntraw[SMALL ? 256 + 64 : WAVES * 256 + 64];
    __shared__ u32 lw[16];
    __shared__ u32 gcount[WAVES * NI];
#ifdef ESP_LOCAL_PAD
    __shared__ char s_pad[ESP_LOCAL_PAD];  // (occupancy experiment: forces fewer workgroups per CU)
    if (threadIdx.x == 0 && a.S < 0) s_pad[0] = 1;
#endif
    __shared__ u64 s_dst;
    __shared__ int s_seg;
    __shared__ u32 s_early;
    __shared__ u32 s_rmin, s_rmax;  // group tier: smallest / largest row of the segment
    __shared__ i64 s_win[96];
    __shared__ u32 s_fine[16];  // fine partition: where bucket j of the segment starts (relative to the segment)
    // scratch of the radix tier, NOT inside s_win: between a wave's reads of s_win (its segment's bounds) and another wave's
    // first scratch store there is no workgroup barrier when the column tiers are skipped (cl_bits < 0) -- the radix tier used to
    // keep its two words per wave in s_win[0 .. 16) and a wave that was late met the other waves' AND / OR masks instead of its
    // segment's bounds whenever its ticket lay 17 .. 32 below its workgroup index (NOTES/round6.md: what corrupted a handle whose
    // flush ran beside another handle's)
    __shared__ u64 s_rtail[SMALL ? 1 : 2 * WAVES];
    u32(*cnt)[256] = reinterpret_cast<u32(*)[256]>(cntraw);
    u32 *ccnt = cntraw;

    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    // claim a segment in start order: every predecessor of a look-back chain has started
    // (a flush with more than MAX_GRID segments is issued as several launches; the ticket counter
    // and the look-back granules carry over from one launch to the next)
    // The segment starts around the expected ticket (launch offset + blockIdx) are fetched while the
    // ticket atomic is in flight: two dependent round trips become one (speed only -- a ticket
    // outside the window simply reloads).
    // (a FINE partition's table -- 4-byte keys only -- holds 2^fb entries per segment: the window covers fewer segments)
    const int fb = K32 ? a.fb : 0;
    const int WIN = 64 >> fb;
    const i64 w0 = max((i64)0, a.first + (i64)blockIdx.x - WIN / 2);
    if (!PIECES && t <= ((WIN + 1) << fb) && (w0 << fb) + t <= ((i64)a.S << fb)) s_win[t] = a.seg_start[(w0 << fb) + t];
    // (pieces: the same window over every source's row of piece starts -- WINP + 1 bounds per source in the 96 words -- and the
    // sources' array addresses, which do not depend on the ticket at all)
    const int WINP = PIECES ? min(64, 96 / max(a.npieces, 1) - 1) : 0;
    const i64 wp0 = max((i64)0, a.first + (i64)blockIdx.x - WINP / 2);
    __shared__ const u64 *p_k[PIECES ? MAX_PIECES : 1];
    __shared__ const double *p_v[PIECES ? MAX_PIECES : 1];
    if constexpr (PIECES) {
        if (WINP >= 4 && t < a.npieces * (WINP + 1)) {
            const int q = t / (WINP + 1), j = t - q * (WINP + 1);
            if (wp0 + j <= (i64)a.S) s_win[t] = a.pstart[(size_t)q * (size_t)(a.S + 1) + (size_t)(wp0 + j)];
        }
        if (t >= THREADS - 2 * MAX_PIECES) {  // (the last two waves: the first ones fetch the window)
            const int q = t - (THREADS - 2 * MAX_PIECES);
            if (q < a.npieces) p_k[q] = static_cast<const u64 *>(a.ptab[q]);
            if (q >= MAX_PIECES && q - MAX_PIECES < a.npieces) p_v[q - MAX_PIECES] = static_cast<const double *>(a.ptab[a.npieces + q - MAX_PIECES]);
        }
    }
    if (t == 0) {
        s_seg = (int)atomicAdd(a.ticket, 1u);
        s_early = 0;
        s_rmin = ~0u;
        s_rmax = 0u;
    }
    __syncthreads();
    const int s = esp_uniform_i32(s_seg);
    if (s >= a.S) return;
    const int wbase = w * (NI * ESP_WAVE) + lane;
    const u64 lt = (1ull << lane) - 1ull;
    const u64 rowmask = (1ull << a.rb) - 1ull;
    u64 k[NI];
    double vraw[NI];
    u64 hi;
    u64 bad = 0;
    int n;
    if constexpr (!PIECES) {
        const bool inwin = s >= w0 && (i64)s + 1 <= w0 + WIN + 1;
        const i64 beg = esp_uniform_i64(inwin ? s_win[(s - w0) << fb] : a.seg_start[(i64)s << fb]);
        const i64 seg_end = esp_uniform_i64(inwin ? s_win[(s - w0 + 1) << fb] : a.seg_start[((i64)s + 1) << fb]);
        if (fb > 0 && t < (1 << fb)) s_fine[t] = (u32)((inwin ? s_win[((s - w0) << fb) + t] : a.seg_start[((i64)s << fb) + t]) - beg);
#ifdef ESP_LOCAL_STAMPS
        if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)s * 16 + 0] = wall_clock64();
#endif
        n = (int)(seg_end - beg);
        if (a.total >= 0 && s == a.S - 1 && seg_end != a.total && t == 0) atomicOr(a.err, 2u);  // (an entry behind the last column)
        // shared prefix of the segment (window-relative), turned back into an absolute key prefix
        // (an empty segment has no key to take it from: its index is the prefix)
        if constexpr (K32)
            hi = ((u64)s << a.rem_bits) + a.base;
        else
            hi = n > 0 ? esp_uniform_u64(((((a.keys_in[beg] >> ESP_TAG_BITS) - a.base) >> a.rem_bits) << a.rem_bits) + a.base)
                       : ((u64)s << a.rem_bits) + a.base;
        // all 16 loads of a thread are issued before anything depends on them: the index is clamped so
        // that the loads need no branch (slots past the end re-read the last entry and are discarded)
        // (an empty segment reads its predecessor's last entry instead -- or entry 0 -- and ignores it: one
        // guard per load would cost two scalar instructions each)
        const i64 lbeg = n > 0 ? beg : max(beg - 1, (i64)0);
        const int nlast = n > 0 ? n - 1 : 0;
        if constexpr (K32) {
            const u32 *k32 = reinterpret_cast<const u32 *>(a.keys_in);
#pragma unroll
            for (int i = 0; i < NI; i++) k[i] = (u64)k32[lbeg + min(wbase + i * ESP_WAVE, nlast)];
        } else {
#pragma unroll
            for (int i = 0; i < NI; i++) k[i] = a.keys_in[lbeg + min(wbase + i * ESP_WAVE, nlast)];
        }
#pragma unroll
        for (int i = 0; i < NI; i++) vraw[i] = a.vals_in[lbeg + min(wbase + i * ESP_WAVE, nlast)];
    } else {
        // the segment's pieces, one per source rank (most segments of a slab-wise assembly have one)
        __shared__ i64 p_beg[MAX_PIECES];
        __shared__ int p_pre[MAX_PIECES + 1];
        // (every wave looks at the piece bounds itself: a segment with ONE non-empty piece -- all of them but those along a range
        // boundary -- needs no table in LDS, no scan and no second barrier; the test is the same in every wave)
        int len = 0;
        i64 pb0 = 0;
        if ((w == 0 || !a.pieces_dense) && lane < a.npieces) {
            const bool inwin = WINP >= 4 && s >= wp0 && (i64)s + 1 <= wp0 + WINP;
            const int at = lane * (WINP + 1) + (int)(s - wp0);
            pb0 = inwin ? s_win[at] : a.pstart[(size_t)lane * (size_t)(a.S + 1) + (size_t)s];
            len = (int)((inwin ? s_win[at + 1] : a.pstart[(size_t)lane * (size_t)(a.S + 1) + (size_t)s + 1]) - pb0);
        }
        const u64 nonempty = a.pieces_dense ? 0ull : __ballot(len > 0);
        const int single = __popcll(nonempty) == 1 ? (int)__builtin_ctzll(nonempty) : -1;
        i64 beg1 = 0;
        if (single >= 0) {
            n = esp_uniform_i32(__shfl(len, single, ESP_WAVE));
            beg1 = esp_uniform_i64(__shfl(pb0, single, ESP_WAVE));
        } else {
            if (w == 0) {
                if (lane < a.npieces) p_beg[lane] = pb0;
                int inc = len;
#pragma unroll
                for (int dlt = 1; dlt < ESP_WAVE; dlt <<= 1) {
                    const int o = __shfl_up(inc, dlt, ESP_WAVE);
                    if (lane >= dlt) inc += o;
                }
                if (lane < a.npieces) p_pre[lane] = inc - len;
                if (lane == 63) p_pre[a.npieces] = inc;
            }
            __syncthreads();
            n = p_pre[a.npieces];
        }
        n = min(n, CAPK);  // (the host checked the merged length; never index past the LDS arrays)
        hi = ((u64)s << a.rem_bits) + a.base;
        const int nlast = n > 0 ? n - 1 : 0;
        // FINE partition of a shard's producer: the own piece lies bucket by bucket, 2^fb buckets per segment, and its 4-byte keys
        // lack the bucket's number inside the segment (Args::fb, Args::own_fine) -- an entry's position in the piece tells it.
        // Lane j of every wave: where bucket j starts, relative to the piece (requested here, used behind the loads).
        const int pfb = (P32 || MIX) ? a.fb : 0;
        u32 fine_rel = 0;
        if (pfb > 0 && lane < (1 << pfb)) fine_rel = (u32)(a.own_fine[((i64)s << pfb) + lane] - a.own_fine[(i64)s << pfb]);
        auto sub_bucket = [&](u32 prel) -> u64 {
            u32 sub = 0;
#pragma unroll
            for (int j = 1; j < 16; j++) {
                const u32 f = (u32)__builtin_amdgcn_readlane((int)fine_rel, j);
                sub += (j < (1 << pfb) && prel >= f) ? 1u : 0u;
            }
            return (u64)sub;
        };
        const u64 lowmask = (((u64)1 << (a.rem_bits - 1)) << 1) - 1ull;
        // P32: piece a.k32_piece holds 4-byte keys -- the bits below the segment's prefix, kind a.kind32 (a shard's own
        // range, written by its producer) -- which become packed keys as they are loaded; the others are packed
        // (pointers read from a table are generic to the compiler: say that they are global memory, or every load of
        // the pieces is a flat load)
        typedef const u64 __attribute__((address_space(1))) *g_u64;
        typedef const u32 __attribute__((address_space(1))) *g_u32;
        typedef const double __attribute__((address_space(1))) *g_f64;
        if (single >= 0) {
            const g_u64 pk = (g_u64)p_k[single];
            const g_f64 pv = (g_f64)p_v[single];
            const i64 beg = beg1;
            const bool narrow = MIX && single != a.k32_piece;  // (packed keys that become 4-byte ones below)
            if constexpr (MIX) {
                if (narrow) {
#pragma unroll
                    for (int i = 0; i < NI; i++) k[i] = pk[beg + min(wbase + i * ESP_WAVE, nlast)];
                } else {
                    const g_u32 pk4 = (g_u32)(pk + a.k32_lo) + (a.k32_lo & 1) - a.k32_lo;  // (esprun::own_keys32)
#pragma unroll
                    for (int i = 0; i < NI; i++) k[i] = (u64)pk4[beg + min(wbase + i * ESP_WAVE, nlast)];
                }
            } else if constexpr (K32) {
                const g_u32 pk4 = (g_u32)pk;
#pragma unroll
                for (int i = 0; i < NI; i++) k[i] = (u64)pk4[beg + min(wbase + i * ESP_WAVE, nlast)];
            } else if (P32 && single == a.k32_piece) {
                // (the 4-byte key of position p of the piece sits at esprun::own_keys32(keys, k32_lo)[p])
                const g_u32 pk4 = (g_u32)(pk + a.k32_lo) + (a.k32_lo & 1) - a.k32_lo;  // (esprun::own_keys32)
#pragma unroll
                for (int i = 0; i < NI; i++)
                    k[i] = ((hi + (u64)pk4[beg + min(wbase + i * ESP_WAVE, nlast)]) << ESP_TAG_BITS) | (u64)a.kind32;
            } else {
#pragma unroll
                for (int i = 0; i < NI; i++) k[i] = pk[beg + min(wbase + i * ESP_WAVE, nlast)];
            }
#pragma unroll
            for (int i = 0; i < NI; i++) vraw[i] = pv[beg + min(wbase + i * ESP_WAVE, nlast)];
            if constexpr (MIX) {
                if (narrow && n > 0) {
#pragma unroll
                    for (int i = 0; i < NI; i++) {
                        const u64 rel = (k[i] >> ESP_TAG_BITS) - hi;
                        bad |= rel >> a.rem_bits;
                        k[i] = rel & lowmask;
                    }
                }
            }
            if constexpr (P32 || MIX) {
                if (pfb > 0 && single == a.k32_piece) {
#pragma unroll
                    for (int i = 0; i < NI; i++) {
                        const u64 sub = sub_bucket((u32)min(wbase + i * ESP_WAVE, nlast));
                        k[i] += MIX ? sub << 32 : sub << (32 + ESP_TAG_BITS);
                    }
                }
            }
        } else {
            g_u64 ak[NI];
            g_f64 av[NI];
            bool k4[NI];
            u32 prel4[NI];  // (position inside the piece: what tells an entry of the own piece its bucket)
#pragma unroll
            for (int i = 0; i < NI; i++) {
                const int p = min(wbase + i * ESP_WAVE, nlast);
                int q = 0;
                while (q + 1 < a.npieces && p >= p_pre[q + 1]) q++;
                const i64 at = p_beg[q] + (i64)(p - p_pre[q]);
                prel4[i] = (u32)(p - p_pre[q]);
                k4[i] = (P32 || MIX) && q == a.k32_piece;
                const g_u64 base_k = (g_u64)p_k[q];
                if constexpr (K32 && !MIX)
                    ak[i] = (g_u64)((g_u32)base_k + at);
                else
                    ak[i] = k4[i] ? (g_u64)((g_u32)(base_k + a.k32_lo) + ((a.k32_lo & 1) + at - a.k32_lo)) : base_k + at;
                av[i] = (g_f64)p_v[q] + at;
            }
#pragma unroll
            for (int i = 0; i < NI; i++) {
                if (n <= 0)
                    k[i] = 0ull;
                else if constexpr (K32 && !MIX)
                    k[i] = (u64) * (g_u32)ak[i];
                else if (k4[i])
                    k[i] = MIX ? (u64) * (g_u32)ak[i] : ((hi + (u64) * (g_u32)ak[i]) << ESP_TAG_BITS) | (u64)a.kind32;
                else
                    k[i] = *ak[i];
            }
#pragma unroll
            for (int i = 0; i < NI; i++) vraw[i] = n > 0 ? *av[i] : 0.0;
            if constexpr (MIX) {
                if (n > 0) {
#pragma unroll
                    for (int i = 0; i < NI; i++) {
                        const u64 rel = (k[i] >> ESP_TAG_BITS) - hi;
                        bad |= k4[i] ? 0ull : rel >> a.rem_bits;
                        k[i] = k4[i] ? k[i] : rel & lowmask;
                    }
                }
            }
            if constexpr (P32 || MIX) {
                if (pfb > 0 && n > 0) {
#pragma unroll
                    for (int i = 0; i < NI; i++) {
                        const u64 sub = k4[i] ? sub_bucket(prel4[i]) : 0ull;
                        k[i] += MIX ? sub << 32 : sub << (32 + ESP_TAG_BITS);
                    }
                }
            }
        }
    }
    const u64 hi4 = hi << ESP_TAG_BITS;  // the segment's prefix as it sits in a packed key
    // (LDS work that does not depend on the loads goes first: it runs while they are in flight)
    if (a.cl_bits >= 0)
        for (int q = t; q <= (1 << a.cl_bits); q += THREADS) ccnt[q] = 0;
    if constexpr (K32 && !PIECES) {
        if (fb > 0) {
            // the bucket's number inside the segment = the key bits [32, 32 + fb): an entry's position tells it (the entries lie
            // bucket by bucket)
            __syncthreads();  // (s_fine)
#pragma unroll
            for (int i = 0; i < NI; i++) {
                const u32 p = (u32)(wbase + i * ESP_WAVE);
                u32 sub = 0;
                for (int j = 1; j < (1 << fb); j++) sub += p >= s_fine[j] ? 1u : 0u;
                k[i] |= (u64)sub << 32;
            }
        }
    }
    // branch-free: slots past the end hold a copy of the last entry (clamped loads) and become NOREC
    const u64 relmask = (((u64)1 << (a.rem_bits + ESP_TAG_BITS - 1)) << 1) - 1ull;  // (rem_bits + 2 may be 64)
#pragma unroll
    for (int i = 0; i < NI; i++) {
        const int p = wbase + i * ESP_WAVE;
        const u64 key = k[i];
        sval[p] = vraw[i];
        // key - hi4 = (bits below the segment's prefix) << 2 | kind for every entry of the segment;
        // anything above is an entry outside the declared key window / the segment (reported to
        // the host, which rejects the flush)
        u64 kk;
        if constexpr (K32) {
            // (the partition masked the key to the bits below the prefix: nothing to check)
            kk = (key << SUB_SHIFT) | (u64)(u32)((p << ESP_TAG_BITS) | a.kind32);
        } else {
            // (... and is folded back into the segment: whatever follows -- LDS counters, CSC look-ups, column
            // marks -- stays inside its arrays; the flush fails anyway)
            const u64 rel = key - hi4;
            bad |= rel >> (a.rem_bits + ESP_TAG_BITS);
            kk = (((rel & relmask) & ~(u64)ESP_TAG_MASK) << IDX_BITS) | (u64)(u32)((p << ESP_TAG_BITS) | ((u32)key & (u32)ESP_TAG_MASK));
        }
        k[i] = p < n ? kk : NOREC;  // NOREC sorts behind every real entry
    }
    if (bad != 0 && n > 0) atomicOr(a.err, 2u);
#ifdef ESP_LOCAL_STAMPS
    if (a.stamps) { __syncthreads(); }
#endif
#ifdef ESP_LOCAL_STAMPS
    if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)s * 16 + 1] = wall_clock64();
#endif
    bool done = n == 0;
    bool refill = false;  // the group tier scattered the keys by column and gave up: the radix tier reads them back
    bool dense_done = false;  // the group tier's dense form wrote the segment's records to their dense LDS positions itself
    bool reg_ran = false;     // the register tier folded the segment (over a stored pattern it notes in s_early whether a record was left)
    bool lb_done = false;  // the look-back was started by the last wave (early publication, see the register tier)
    LbState lbs;
    lb_init(lbs, 0);
    if (a.stop_after == 1) done = true;

    if (!done && a.cl_bits >= 0) {
        // ---- column tiers: counting sort by local column with LDS atomics
        const int ncl = 1 << a.cl_bits;
        const int csh = SUB_SHIFT + a.rb;  // packed >> csh = local column
        __syncthreads();  // ccnt is zero (cleared while the loads were in flight)
        unsigned short slot[NI];
#pragma unroll
        for (int i = 0; i < NI; i++) {
            slot[i] = 0;
            if (wbase + i * ESP_WAVE < n) slot[i] = (unsigned short)atomicAdd(&ccnt[(u32)(k[i] >> csh)], 1u);
        }
        __syncthreads();
        // exclusive scan of the column counts (ncl <= 2048 -> <= 4 per thread) + longest run
        constexpr int PER = CL_MAX / THREADS;
        u32 v[PER];
        u32 run = 0, mx = 0, inc = 0;
        // (a wave whose columns all lie past ncl -- 7 of 8 waves when a segment is 256 columns -- only
        // reports zeros: the phase is bound by the instructions issued, not by the data)
        const bool wave_has = (t & ~(ESP_WAVE - 1)) * PER < ncl;
        if (wave_has) {
#pragma unroll
            for (int j = 0; j < PER; j++) {
                const int q = t * PER + j;
                const u32 x = q < ncl ? ccnt[q] : 0;
                mx = max(mx, x);
                v[j] = run;
                run += x;
            }
            inc = esp_wave_scan_add(run);
            mx = esp_wave_max(mx);
        }
        if (lane == 63) lw[w] = inc;
        if (lane == 0) lw[8 + w] = mx;
        __syncthreads();
        u32 base = inc - run, maxrun = 0;
#pragma unroll
        for (int i = 0; i < WAVES; i++) {
            if (i < w) base += lw[i];
            maxrun = max(maxrun, lw[8 + i]);
        }
        if (wave_has) {
#pragma unroll
            for (int j = 0; j < PER; j++) {
                const int q = t * PER + j;
                if (q < ncl) ccnt[q] = base + v[j];
            }
        }
        if (t == 0) ccnt[ncl] = (u32)n;
        __syncthreads();
#ifdef ESP_LOCAL_STAMPS
        if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)s * 16 + 2] = wall_clock64();
#endif
        if (a.stop_after == 2) done = true;
        // longer runs go to the radix tier (measured: from 25 entries per column on it beats an insertion sort of
        // the run by its lane 2.5 to 4 times; the insertion tier is gone)
        const int reg_max = BIG ? REG_RUN : 16;
        if (t == 0 && maxrun > 16) atomicMax(a.maxrun_seen, maxrun);  // (tells the host which kernel variant suits this matrix)
        if (!GRP && !done && maxrun <= reg_max && a.rem_bits <= REG_MAX_REM) {
#pragma unroll
            for (int i = 0; i < NI; i++)
                if (wbase + i * ESP_WAVE < n) skey[ccnt[(u32)(k[i] >> csh)] + slot[i]] = k[i];
            __syncthreads();
#ifdef ESP_LOCAL_STAMPS
            if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)s * 16 + 3] = wall_clock64();
#endif
            if (a.stop_after == 3) done = true;
            if (!done && maxrun <= (BIG ? REG_RUN : 16) && a.rem_bits <= REG_MAX_REM) {
                // one lane per column: the whole run in registers, sorting network + ordered fold; the
                // network is sized to the longest run of the segment (12 covers a 7-point stencil)
                // (fresh matrix: the tier leaves the records dense)
                reg_ran = true;
                if (maxrun <= 12)
                    lb_done = reg_tier<12, FRESH, UPD, SMALL>(a, skey, sval, ccnt, ncl, s, hi, rowmask, &s_early, lbs, &dense_done, lw);
                else if (!BIG || maxrun <= 16)
                    lb_done = reg_tier<16, FRESH, UPD, SMALL>(a, skey, sval, ccnt, ncl, s, hi, rowmask, &s_early, lbs, &dense_done, lw);
                else if constexpr (BIG)
                    lb_done = reg_tier<REG_RUN, FRESH, UPD, SMALL>(a, skey, sval, ccnt, ncl, s, hi, rowmask, &s_early, lbs);
                done = true;
            }
        } else if constexpr (GRP) {
            // ---- group tier: runs of up to 256 entries sorted by 2 .. 16 lanes each, keys in registers (see group_sort)
            if (!done && maxrun <= (SHORTG ? 32 : GROUP_MAX) && a.rb <= 30 && a.cl_bits <= GROUP_CL_BITS && !a.no_group) {
                u32 rmin = ~0u, rmax = 0u;
#pragma unroll
                for (int i = 0; i < NI; i++)
                    if (wbase + i * ESP_WAVE < n) {
                        const u32 row = (u32)(k[i] >> SUB_SHIFT) & (u32)rowmask;
                        rmin = min(rmin, row);
                        rmax = max(rmax, row);
                        skey[ccnt[(u32)(k[i] >> csh)] + slot[i]] = k[i];
                    }
                rmin = ~esp_wave_max(~rmin);
                rmax = esp_wave_max(rmax);
                if (lane == 0) {
                    atomicMin(&s_rmin, rmin);
                    atomicMax(&s_rmax, rmax);
                }
                __syncthreads();
                rmin = s_rmin;
                if (s_rmax - rmin < (1u << GROUP_ROW_BITS)) {
                    unsigned long long *gstamp = nullptr;
#ifdef ESP_LOCAL_STAMPS
                    if (a.stamps) {
                        gstamp = a.stamps + (size_t)s * 16;
                        if (t == 0) gstamp[12] = wall_clock64();
                    }
#endif
                    // (one kind for every entry, known to the kernel's instantiation or to the host: the fold is an addition)
                    const int gmode = (UPD || a.kind_all == ESP_UPDATE) ? 1 : a.kind_all == ESP_RAWUPDATE ? 2 : 0;
#define ESP_GROUP_GO(GG, RR)                                                                                         \
    do {                                                                                                             \
        if (gmode == 1)                                                                                              \
            group_columns<GG, RR, CAPK, FRESH, 1>(a, skey, sval, ccnt, ncl, rmin, hi, rowmask, gstamp);              \
        else if (gmode == 2)                                                                                         \
            group_columns<GG, RR, CAPK, FRESH, 2>(a, skey, sval, ccnt, ncl, rmin, hi, rowmask, gstamp);              \
        else                                                                                                         \
            group_columns<GG, RR, CAPK, FRESH, 0>(a, skey, sval, ccnt, ncl, rmin, hi, rowmask, gstamp);              \
    } while (0)
                    // (fresh matrix, one kind of additions, every column with its lanes at once: the dense form)
                    bool went_dense = false;
                    if constexpr (SHORTG) {
                        bool dense4 = false;
                        if constexpr (FRESH) {
                            if (gmode != 0 && ncl * 4 <= THREADS && a.stop_after == 0) {
                                const DenseCtx dcx{&s_early, reinterpret_cast<unsigned short *>(cntraw + 1024), &lbs, s};
                                if (gmode == 1)
                                    group_columns<4, 8, CAPK, true, 1, true>(a, skey, sval, ccnt, ncl, rmin, hi, rowmask, gstamp, &dcx);
                                else
                                    group_columns<4, 8, CAPK, true, 2, true>(a, skey, sval, ccnt, ncl, rmin, hi, rowmask, gstamp, &dcx);
                                dense4 = true;
                                dense_done = true;
                                lb_done = true;
                            }
                        }
                        if (dense4) {
                        } else if (ncl * 4 <= THREADS)
                            ESP_GROUP_GO(4, 8);
                        else
                            ESP_GROUP_GO(2, 16);
                        went_dense = true;  // (nothing below applies)
                    } else if constexpr (FRESH) {
                        if (gmode != 0 && maxrun > 64 && maxrun <= 128 && ncl * 16 <= THREADS && a.stop_after == 0) {
                            const DenseCtx dcx{&s_early, reinterpret_cast<unsigned short *>(cntraw + 1024), &lbs, s};
                            if (gmode == 1)
                                group_columns<16, 8, CAPK, true, 1, true>(a, skey, sval, ccnt, ncl, rmin, hi, rowmask, gstamp, &dcx);
                            else
                                group_columns<16, 8, CAPK, true, 2, true>(a, skey, sval, ccnt, ncl, rmin, hi, rowmask, gstamp, &dcx);
                            went_dense = true;
                            dense_done = true;
                            lb_done = true;
                        }
                    }
                    if (went_dense) {
                    } else if constexpr (SHORTG) {
                    } else if (maxrun <= 32)
                        ESP_GROUP_GO(2, 16);
                    else if (maxrun <= 64)
                        ESP_GROUP_GO(4, 16);
                    else if (maxrun <= 128 && ncl * 16 <= THREADS)
                        ESP_GROUP_GO(16, 8);  // (few long columns, 3-D FEM: 16 lanes x 8 keys keep every wave busy)
                    else if (maxrun <= 128)
                        ESP_GROUP_GO(8, 16);
                    else
                        ESP_GROUP_GO(16, 16);
#undef ESP_GROUP_GO
                    done = true;
                } else {
                    refill = true;  // (the rows of the segment lie too far apart for 32-bit sort keys: radix tier)
                }
            }
        } else if constexpr (SMALL) {
            if (!done) {
                // slow tier of the small variant (column runs longer than the register tiers take; the host sends the
                // handle's next flushes to the regular kernel): one lane per column sorts its run in LDS by insertion
                // -- keys are unique: sub-key | slot index | kind -- and folds it sequentially
#pragma unroll
                for (int i = 0; i < NI; i++)
                    if (wbase + i * ESP_WAVE < n) skey[ccnt[(u32)(k[i] >> csh)] + slot[i]] = k[i];
                __syncthreads();
                for (int c = t; c < ncl; c += THREADS) {
                    const int rs = (int)ccnt[c];
                    const int len = (int)ccnt[c + 1] - rs;
                    for (int i = 1; i < len; i++) {
                        const u64 x = skey[rs + i];
                        int j = i - 1;
                        while (j >= 0 && skey[rs + j] > x) {
                            skey[rs + j + 1] = skey[rs + j];
                            j--;
                        }
                        skey[rs + j + 1] = x;
                    }
                    fold_run_lds<FRESH, UPD>(a, skey, sval, rs, len, hi, rowmask);
                }
                done = true;
            }
        }
        __syncthreads();  // records are in place (or: ccnt storage is free for the radix counters)
        // Over a stored pattern, every update of the segment hit a stored position (a re-assembly: config 3's batch, a time step):
        // nothing to compact, no offset to resolve, nothing to store -- the segment publishes its zero total and leaves (the last
        // segment of a look-back group and of the flush resolve their chain as ever: lb_may_skip).
        if constexpr (!FRESH && !PIECES) {
            if (reg_ran && !lb_done && a.stop_after == 0 && s_early == 0u && lb_may_skip(a, s)) {
#ifdef ESP_LOCAL_STAMPS
                if (!a.stamps)
#endif
                {
                    if (t == 0) __hip_atomic_store(&a.status[s], ST_AGG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    return;
                }
            }
        }
    }

    if constexpr (!SMALL)
    if (!done) {
        // ---- radix tier (long runs / wide column ranges): stable LSD sort of all remaining bits
        // (after the group tier's scatter the registers no longer hold the keys -- the tier needs them for its own 48 --:
        // they come back from LDS in column order, and the sort takes the slot-index bits along to restore the append order)
        if (refill) {
#pragma unroll
            for (int i = 0; i < ITEMS; i++) k[i] = wbase + i * ESP_WAVE < n ? skey[wbase + i * ESP_WAVE] : NOREC;
            __syncthreads();
        }
        const int npass = radix_tail(k, skey, cnt, lw, refill ? a.rem_bits + IDX_BITS : a.rem_bits, t, lane, w, wbase, n,
                                     s_rtail, refill ? ESP_TAG_BITS : SUB_SHIFT);
        if (npass == 0) {
#pragma unroll
            for (int i = 0; i < ITEMS; i++) skey[wbase + i * ESP_WAVE] = k[i];
        }
        __syncthreads();
#ifdef ESP_LOCAL_STAMPS
        if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)s * 16 + 12] = wall_clock64();
#endif
        // ---- ordered fold, one (col,row) group per thread.  The walk of a group is sequential (append
        // order); with one thread per GROUP the phase takes as long as the longest group, not 8 x as long
        // (a thread that owns 8 slots walked up to 8 groups one after the other, and every wave waited for
        // its longest walk each time -- FEM diagonals hold ~50 duplicates).  Group heads are numbered by a
        // ballot scan over the slots; the list of head positions sits in the radix counters' storage.
        unsigned short *ghead = reinterpret_cast<unsigned short *>(cntraw);  // <= CAP heads (+1 end mark)
        static_assert(sizeof(cntraw) >= (CAP + 1) * sizeof(unsigned short), "head list fits the counter storage");
        bool ishead[ITEMS];
#pragma unroll
        for (int i = 0; i < ITEMS; i++) {
            const int q = wbase + i * ESP_WAVE;
            ishead[i] = q < n && (q == 0 || (skey[q - 1] >> SUB_SHIFT) != (k[i] >> SUB_SHIFT));
            const u64 bal = __ballot(ishead[i]);
            if (lane == 0) gcount[w * ITEMS + i] = (u32)__popcll(bal);
        }
        __syncthreads();
        {
            const u32 c = gcount[lane];  // (WAVES * ITEMS == 64 groups of slots: every wave scans them itself)
            u32 inc = c;
#pragma unroll
            for (int dlt = 1; dlt < ESP_WAVE; dlt <<= 1) {
                const u32 o = __shfl_up(inc, dlt, ESP_WAVE);
                if (lane >= dlt) inc += o;
            }
            const u32 exc = inc - c;
            const int G = (int)__shfl((int)inc, 63, ESP_WAVE);
#pragma unroll
            for (int i = 0; i < ITEMS; i++) {
                const u64 bal = __ballot(ishead[i]);
                const u32 gb = (u32)__shfl((int)exc, w * ITEMS + i, ESP_WAVE);
                if (ishead[i]) ghead[gb + (u32)__popcll(bal & lt)] = (unsigned short)(wbase + i * ESP_WAVE);
            }
            if (t == 0) ghead[G] = (unsigned short)n;
            __syncthreads();
            for (int g = t; g < G; g += THREADS) {
                const int q0 = ghead[g], q1 = ghead[g + 1];
                const u64 k0 = skey[q0];
                const u64 sub = k0 >> SUB_SHIFT;
                const u64 full = hi + sub;
                i64 pos = -1;
                if (a.csc.nnz > 0) pos = espfold::csc_find(a.csc, (i64)(full >> a.rb), (i64)(full & rowmask));
                bool present = (pos >= 0 && a.mode == ESP_FLUSH_ROUTED);
                double x = present ? a.csc.nzval[pos] : 0.0;
                for (int j = q0; j < q1; j++) {
                    const u64 kj = skey[j];
                    espfold::fold_step(present, x, (u32)(kj & ESP_TAG_MASK), sval[(kj >> ESP_TAG_BITS) & (CAP - 1)]);
                }
                // the group's slots belong to this thread alone: the record (or nothing) replaces them
                bool emit = false;
                if (pos >= 0) {
                    if (a.mode == ESP_FLUSH_ROUTED)
                        a.csc.nzval[pos] = x;
                    else if (present)
                        a.csc.nzval[pos] = a.csc.nzval[pos] + x;
                } else if (present) {
                    emit = true;
                }
                const u32 idx0 = (u32)(k0 >> ESP_TAG_BITS) & (CAP - 1);
                skey[q0] = emit ? ((k0 >> SUB_SHIFT << SUB_SHIFT) | ((u64)idx0 << ESP_TAG_BITS)) : NOREC;
                if (emit) sval[idx0] = x;
                for (int j = q0 + 1; j < q1; j++) skey[j] = NOREC;
            }
        }
        __syncthreads();
#ifdef ESP_LOCAL_STAMPS
        if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)s * 16 + 13] = wall_clock64();
#endif
    }
#ifdef ESP_LOCAL_STAMPS
    if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)s * 16 + 4] = wall_clock64();
#endif
    if (a.stop_after == 4) {
        if (t == 0 && s == a.S - 1) __hip_atomic_store(&a.status[s], ST_PRE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }

    // ---- compaction: records (skey[p] != NOREC, value in sval[idx]) -> dense prefix
    u64 rec[NI];
    double rv[NI];
    if (!dense_done) {
#pragma unroll
        for (int i = 0; i < NI; i++) {
            const int p = wbase + i * ESP_WAVE;
            rec[i] = (p < n && a.stop_after == 0) ? skey[p] : NOREC;
            rv[i] = rec[i] != NOREC ? sval[(rec[i] >> ESP_TAG_BITS) & (CAP - 1)] : 0.0;
            const u64 bal = __ballot(rec[i] != NOREC);
            if (lane == 0) gcount[w * NI + i] = (u32)__popcll(bal);
        }
        __syncthreads();
    } else {
#pragma unroll
        for (int i = 0; i < NI; i++) rec[i] = NOREC, rv[i] = 0.0;
    }
    if (dense_done) {
        // (the records lie dense already, the total is in s_early; the last wave is at the look-back)
    } else if (w == 0) {
#ifdef ESP_LOCAL_STAMPS
        if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)s * 16 + 5] = wall_clock64();
#endif
        const u32 c = lane < WAVES * NI ? gcount[lane] : 0u;
        const u32 inc = esp_wave_scan_add(c);
        if (lane < WAVES * NI) gcount[lane] = inc - c;
        const u32 total = (u32)__builtin_amdgcn_readlane((int)inc, 63);
        // ---- decoupled look-back (wave 0) unless the register tier already ran it
        if (!lb_done) {
            // A segment that emits nothing (re-assembly over the stored pattern: every update hit the CSC)
            // has no use for its offset: it publishes its (zero) total and leaves without waiting for its
            // predecessors.  Every 64th segment and the last one still resolve their chain and publish
            // the inclusive prefix, so nobody ever walks back more than 64 + the segments in flight.
            // (a segment that writes its columns' colptr itself needs its offset even then)
            if (total == 0 && lb_may_skip(a, s) && !(FRESH && a.colptr_out)) {
                if (lane == 0) {
                    __hip_atomic_store(&a.status[s], ST_AGG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    s_dst = 0;
                }
            } else {
                const u64 excl = lookback_wave(a, s, total, lane);
                if (lane == 0) s_dst = excl;
            }
        } else if (lane == 0 && total != s_early) {
            atomicOr(a.err, 4u);  // internal consistency: the early count must equal the folded count
        }
        if (lane == 0) lw[0] = total;
    }
    // (no poll of the look-back wave here: it would reach the barrier a round trip late, see reg_tier)
    if (!dense_done) __syncthreads();
#ifdef ESP_LOCAL_STAMPS
    if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)s * 16 + 6] = wall_clock64();
#endif
    const int total = dense_done ? (int)s_early : (int)lw[0];
    // dense prefix in LDS (all records and values are in registers: in-place is safe)
#pragma unroll
    for (int i = 0; i < NI; i++) {
        const u64 bal = __ballot(rec[i] != NOREC);
        if (rec[i] != NOREC) {
            const u32 e = gcount[w * NI + i] + (u32)__popcll(bal & lt);
            skey[e] = hi + (rec[i] >> SUB_SHIFT);
            sval[e] = rv[i];
        }
    }
    if (lb_done && w == WAVES - 1) {  // the rest of the look-back chain, then the inclusive prefix for the successors
        const u64 excl = lb_complete(a, lbs, s, s_early, lane);
        if (lane == 0) s_dst = excl;
#ifdef ESP_LOCAL_STAMPS
        if (a.stamps && lane == 0) a.stamps[(size_t)s * 16 + 11] = wall_clock64();
#endif
    }
    __syncthreads();
    const u64 dst = esp_uniform_u64(s_dst);
    // ---- coalesced stores + column-end marks (or colptr itself)
    const bool direct = FRESH && a.colptr_out != nullptr;
    // (direct: the segment's columns are [c_lo, c_hi); every colptr store is clamped to them -- a key outside the
    // declared window, which fails the flush, must not turn into a store outside the array)
    const i64 c_lo = (i64)(hi >> a.rb);
    const i64 c_hi = direct ? min(c_lo + ((i64)1 << a.cl_bits), a.col_end) : c_lo;
    for (int p = t; p < total; p += THREADS) {
        const u64 key = skey[p];
        if (FRESH)
            a.out_row[dst + p] = (i64)(key & rowmask) + 1;
        else
            a.out_key[dst + p] = key;
        a.out_val[dst + p] = sval[p];
        const u64 col = key >> a.rb;
        if (direct) {
            // first entry of its column: that column and the empty ones in front of it start here
            const i64 prev = p == 0 ? c_lo - 1 : (i64)(skey[p - 1] >> a.rb);
            for (i64 c = max(prev + 1, c_lo); c <= min((i64)col, c_hi - 1); c++) a.colptr_out[c] = (i64)(dst + (u64)p) + 1;
        } else if ((p == total - 1 || (skey[p + 1] >> a.rb) != col) && col < (u64)a.n_cols) {
            if (a.col_aligned)
                a.colend[col] = dst + (u64)p + 1;
            else
                atomicMax((unsigned long long *)&a.colend[col], (unsigned long long)(dst + (u64)p + 1));
        }
    }
    if (direct) {  // the columns behind the last entry (all of them for an empty segment)
        const i64 after = total > 0 ? max((i64)(skey[total - 1] >> a.rb) + 1, c_lo) : c_lo;
        for (i64 c = after + t; c < c_hi; c += THREADS) a.colptr_out[c] = (i64)(dst + (u64)total) + 1;
        if (s == a.S - 1 && t == 0) a.colptr_out[a.col_end] = (i64)(dst + (u64)total) + 1;
    }
#ifdef ESP_LOCAL_STAMPS
    if (a.stamps) {
        __syncthreads();
        if (threadIdx.x == 0) a.stamps[(size_t)s * 16 + 7] = wall_clock64();
    }
#endif
}

}  // namespace esplocal
