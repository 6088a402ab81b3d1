// local_args.hpp -- what the host side needs of the LDS bucket kernel (local.hpp): its capacities, its argument
// block and the launcher.  The kernel instantiations live in translation units of their own (local_*.hip), so that a
// change to the host logic does not recompile ~50 variants of the kernel and the variants compile in parallel.
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

constexpr int G3_CL_BITS = 8;   // group3_k: segments of at most 256 whole columns
constexpr int CL_MAX_BITS = 11;  // up to 2048 local columns counted in LDS
constexpr int CL_MAX = 1 << CL_MAX_BITS;
constexpr int REG_RUN = 24;   // longest column run sorted in registers
constexpr int REG_MAX_REM = 62 - SUB_SHIFT;  // ... when the packed sort keys stay below 2^62 (see load_sorted_run)
constexpr u64 NOREC = ~0ull;

// look-back status granule: [63:62] flag, [61:0] value
constexpr u64 ST_AGG = 1ull << 62;
constexpr u64 ST_PRE = 2ull << 62;
constexpr u64 ST_VAL = (1ull << 62) - 1ull;
constexpr u32 SPIN_LIMIT = 1u << 24;
constexpr i64 MAX_GRID = 1 << 22;  // workgroups per launch (HIP caps a grid at 2^32 threads)

struct Args {
    const u64 *keys_in;
    const double *vals_in;
    const i64 *seg_start;  // S+1
    int S;
    int rem_bits;  // key bits below the partition prefix (col/row bits, without the kind bits)
    u64 base;      // key window base: keys are sorted as (key>>2) - base
    int rb;
    int cl_bits;  // local column bits (rem_bits - rb) when 0..CL_MAX_BITS, else -1: radix tail only
    int col_aligned;  // a segment is a whole number of columns (column-end marks need no atomics)
    espfold::Csc csc;
    int mode;
    i64 *out_row;    // FRESH: rowval (1-based) of the new CSC
    u64 *out_key;    // !FRESH: (col0<<rb | row0) of new entries
    double *out_val;
    u64 *colend;     // per column: output index just past its last emitted entry (0 = none)
    u64 *status;     // S look-back granules, zeroed before launch
    u64 *gstatus;    // one more per group of 64 segments (see the look-back), zeroed as well
    u32 *ticket;     // zeroed before launch
    u32 *err;        // set to 1 if a look-back spin ran into its bound
    int stop_after;  // timing ablation only (0 = run everything)
    i64 total;       // >= 0: the launch ends before the table's last segment -- the last launched segment must end at this entry
    i64 first;       // ticket value the first workgroup of this launch is expected to draw
    unsigned long long *stamps;  // diagnostics (builds with -DESP_LOCAL_STAMPS only): 8 wall-clock stamps per segment
    // PIECES variant (column shards after the partitioned exchange): segment s is the concatenation, in
    // source-rank order, of one piece per source: entries [pstart[q*(S+1)+s], pstart[q*(S+1)+s+1]) of the
    // arrays ptab[q] (keys) / ptab[npieces+q] (values)
    int npieces;
    const i64 *pstart;
    const void *const *ptab;
    int pieces_dense;  // most segments hold entries of several pieces: their table goes through LDS at once (else every wave first looks
                       // whether the segment has ONE non-empty piece -- a shard's usual segment -- and then needs neither table nor barrier)
    u32 *maxrun_seen;  // longest column run any segment of this flush met (atomicMax)
    // FRESH kernels on whole-column segments that cover the flush's column range: the segment writes colptr (1-based)
    // for its own columns itself -- no column-end marks, no scan over the columns afterwards (nullptr: marks in colend)
    u32 kind32;  // K32 kernels: the kind of every entry; KEYS 4 / 5: of the entries of piece k32_piece
    int k32_piece;  // KEYS 4 / 5: the piece that holds 4-byte keys ...
    i64 k32_lo;     // ... from its position k32_lo on: key of position p at esprun::own_keys32(keys, k32_lo)[p]
    i64 *colptr_out;
    i64 n_cols;   // columns of the matrix (column-end marks of a failing flush -- keys outside the window -- stay inside colend)
    i64 col_end;  // end of the column range (colptr_out[col_end] = 1 + nnz comes from the last segment)
    int late_total;  // group3_k: the segment's total is published after the fold even when it is known after the sort (experiments: ESP_LATE_TOTAL)
    int no_group;  // test hook: column runs of more than 24 entries go to the radix tier, never to the group tier
    int kind_all;  // >= 0: every pending entry has this kind (the host's bookkeeping), whatever the key format says; else -1
    int expect_hits;  // the host expects most positions to be stored already: a short stored column is fetched with its values
    int fb;           // K32 kernels: the segment table is 2^fb times finer than the segments (esp_handle::PrePart::fb): segment s =
                      // table entries [s << fb, (s + 1) << fb], and the 4-byte key of an entry lacks the bucket's number inside the segment
                      // PIECES with a 4-byte-key piece (a shard's own range, KEYS 4 / 5): the same for THAT piece, whose fine table is
    const i64 *own_fine;  // ... this one: 2^fb entries per segment (+ 1), absolute positions like the piece's row of pstart
    double *hits_out; // group3_k's re-assembly form (HITS): the new values of the stored positions (a second nzval array)
};
constexpr int MAX_PIECES = 64;

// Base.sum over several buffers that all hold element batches as sorted items (sum.hip: flush_sum_items): ONE launch of the fused
// bucket kernel (group3_items.hpp, MULTI) folds every non-empty (buffer, segment) pair; per buffer it needs
struct MultiBuf {
    const u64 *sorted;     // the buffer's sorted item records
    const i64 *seg;        // its segment table (S + 1 entries, in UPDATES)
    const double *elmat;   // its element matrices
    const char *cellrec;   // its 64-byte cell records
    int negate;
    int low;               // its item records' bits below the column (virtual row bits + 2)
};
constexpr int MULTI_SEG_BITS = 20;  // a (buffer, segment) pair as buffer << 20 | segment

// which instantiation of local_k serves a flush (see the kernel's template parameters in local.hpp)
struct Variant {
    bool fresh;   // the matrix holds no entries: rowval/nzval of the new CSC are written directly
    bool pieces;  // segments are concatenations of per-source pieces
    bool big;     // carries the 24-input register tier
    bool small_variant;  // segments of at most 3072 entries over at most 256 columns: three workgroups per CU
    int keys;     // key format 0 .. 7
    bool grp = false;  // the group-tier kernel (column runs of more than 16 entries): regular form only
    bool shortg = false;  // ... its form for runs of at most 32 entries (four lanes x 8 keys per column)
    bool g3 = false;      // ... the group tier as a kernel of its own with three workgroups per CU (group3.hpp)
    bool g3hits = false;  // ... its re-assembly form: additions over a stored pattern the same mesh built (all-or-nothing, see group_columns)
    bool g3wide = false;  // ... its form for segments whose rows spread over more than 2^18 (two sorts per run, two workgroups per CU)
    bool g3k64 = false;   // ... fed packed 8-byte keys of one known kind whose bits below the prefix fit 32 (the radix passes' output)
};
// enqueues the kernel; false when the combination has no instantiation
bool launch(const Variant &v, unsigned grid, hipStream_t stream, const Args &a);
bool launch_regular(const Variant &v, unsigned grid, hipStream_t stream, const Args &a);        // local_a.hip
bool launch_small(const Variant &v, unsigned grid, hipStream_t stream, const Args &a);          // local_b.hip
bool launch_pieces_fresh(const Variant &v, unsigned grid, hipStream_t stream, const Args &a);   // local_c.hip
bool launch_pieces_stored(const Variant &v, unsigned grid, hipStream_t stream, const Args &a);  // local_d.hip
bool launch_pieces_small(const Variant &v, unsigned grid, hipStream_t stream, const Args &a);   // local_e.hip
bool launch_group(const Variant &v, unsigned grid, hipStream_t stream, const Args &a);          // local_f.hip
bool launch_group_short(const Variant &v, unsigned grid, hipStream_t stream, const Args &a);    // local_g.hip
bool launch_group3(const Variant &v, unsigned grid, hipStream_t stream, const Args &a);         // local_h.hip


}  // namespace esplocal
