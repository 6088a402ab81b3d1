// internal.hpp -- what the translation units of libesparse_hip share: the handle, the error / timing helpers, the plan
// and partition records, and the prototypes of the functions that cross file boundaries (hidden visibility: nothing
// of this is part of the C ABI, which is include/esparse_hip.h).
//   handle.hip     lifetime, buffers, append (esp_stage_begin / esp_commit / esp_append_*), CSC in and out, timing, debug
//   produce.hip    device-side producers (esp_generate_*): plain, producer-side partition (run lists), item partition
//   partition.hip  the plan (prefix bits, back-off), radix passes, run-based single pass, sort_msd
//   flush.hip      esp_flush: bucket kernel launch, join with a stored CSC, batch + tail, general path
//   shard.hip      column shards (esp_shard_*) and the group API (group.hpp: exchange policy + RCCL transport)
//   consumers.hip  what reads or edits the assembled CSC: getindex, dropzeros, pattern hash, mul!, Dirichlet, Jacobi / ILU0
//   local_*.hip    the instantiations of the bucket kernel (local.hpp; local_h.hip: group3.hpp, the group tier with three workgroups per CU;
//                  local_j.hip: group3_items.hpp, the same fed with the item records of an element-level batch)
#pragma once
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <cmath>
#include <thread>

#include "common.hpp"
#include "fold.hpp"
#include "generators.hpp"
#include "femitems.hpp"
#include "elements.hpp"
#include "local_args.hpp"
#include "merge.hpp"
#include "radix.hpp"
#include "runpart.hpp"
#include "scan.hpp"


// ------------------------------------------------------------------------ handle
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
};

struct TimedSpan {
    int stage;
    hipEvent_t a, b;
    int launches;
};

struct esp_handle {
    i64 m = 0, n = 0;
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    KeyLayout L{1, 1};
    std::string err;

    // COO append buffer
    DevBuf keys, vals;
    i64 cap = 0, count = 0;
    // ping-pong / scratch
    DevBuf keys2, vals2, hist, segs, colend, newkey, newval, heads, misc, seg[2], tilef[2], segcnt, segout, tseg, ttile;
    int force_path = 0, last_path = 0;
    DevBuf runbuf, chunkbuf;
    i64 chunk_cap = 0, hint = 0;
    int chunk_pb = 0;
    int runs_skip = 0, runs_penalty = 0;  // back-off after a stream turned out not to be pre-sorted
    bool g3_off = false;                  // a segment of this handle's matrix did not fit the three-workgroup group kernel: not tried again
    bool hits_off = false;                // the re-assembly form of the group kernel met a batch that was no re-assembly of the stored pattern: not tried again (until reset!)
    bool g3_wide = false;                 // ... for its rows alone (spread over more than 2^18): the kernel's wide form serves this handle
    bool debug_fail_bucket = false;       // esp_debug_fail_next_bucket_stage (test hook, one shot)
    double debug_plan_cap = 0.0;          // esp_debug_plan_cap: plan as if the bucket kernel took segments of this many entries (test hook)
    int last_group3 = 0;                  // the last flush's bucket kernel was group3_k (esp_debug_last_local_small reports 2)
    bool seen_hits = true;                // the last flush over a stored pattern mostly hit stored positions (re-assembly)
    int seen_maxrun = 0;                  // longest column run the bucket kernel met in the last flush
    int last_partition = 0;               // 1 = run-based single pass, 2 = 8-bit passes only, 4 = the producer's, 7 = shard pieces
    // Producer-side partition: a device-side producer appended to the empty buffer with its PART kernel (runpart.hpp,
    // "the append IS the partition"): the pending entries lie bucket by bucket -- a stable permutation of the stream --
    // and the flush starts at the bucket kernel.  Bucket starts: seg[1] (S + 1 entries).
    struct PrePart {
        bool valid = false;
        int K = 0, pb = 0;       // bits of the key window / of the prefix: S = 1 << pb buckets
        int key_bytes = 8;       // 4: `keys` holds u32 keys (the bits below the prefix); every entry has the kind `kind`
        int kind = 0;
        i64 E = 0, maxlen = 0;   // entries, longest bucket
        // FINE partition (round 6): the plan has `fb` more prefix bits than the bucket kernel needs, so that the bits below the
        // prefix fit 4-byte keys (322^3: 33 bits below the planned prefix; 400^3: 34); the flush's bucket kernel takes 2^fb
        // neighbouring buckets as ONE segment (maxlen_c: the longest of those) and tells an entry's bucket by its position.
        // To everybody else the batch is what its fine plan says: 2^pb buckets of 4-byte keys.
        int fb = 0;
        i64 maxlen_c = 0;
        i64 tail = 0;            // packed entries appended BEHIND the E bucket-ordered ones (count = E + tail)
        u64 base = 0, span = 0;  // the key window it was made for
        double Ee = 0.0;         // (plan_entries of the batch: spread bookkeeping)
        // column shards: the batch was partitioned by (owner, digit inside the owner's column range) -- what
        // esp_shard_partition produces; mw_P windows of mw_nb digits, digit width 2^mw_shift, plan made for mw_eps
        int mw_P = 0, mw_me = 0, mw_shift = 0;
        u32 mw_nb = 0;
        i64 mw_eps = 0;
        bool own32 = false;      // ... and the shard's OWN range holds 4-byte keys of kind `kind` (the sent ranges: packed)
        u64 plan_id = 0;         // ... the table build it came from (prepart_finish numbers them; a reused plan keeps its number)
    } pre;
    u64 plan_counter = 0;
    // esp_shard_partition over a producer's batch whose plan was REUSED: the owner ranges and per-digit counts are those of the
    // call that built the tables -- kept, so that nothing runs beside the PART launch (a second queue with three tiny operations
    // cost that launch 90 us of 500: NOTES/round6.md section 7)
    struct ShardOffsets {
        u64 plan_id = 0;
        int P = 0, me = 0;
        i64 eps = 0, NB = 0, E = 0;
        const void *cnt_at = nullptr;
        std::vector<i64> off;
    } shard_offsets;
    bool pre_keep = false;       // reserve_append: the append that follows goes behind the bucket-ordered batch
    // A batch of an item partition whose EXPANSION has not run (group3_items.hpp): `pre` describes it as if its updates lay
    // bucket by bucket in keys / vals -- they do not yet: the sorted item records lie in the keys array (the two ping-pong
    // halves of the item passes), the cell records of an element-level append in the vals array, and the flush's bucket
    // kernel forms the updates itself (fresh matrix, nothing appended behind the batch).  Everybody else -- an append behind
    // the batch, a clone, getindex, a shard call, a flush over a stored pattern, a segment the fused kernel refuses -- calls
    // lazy_expand() first (pending_materialize and reserve_append do), which runs the expansion kernel into the scratch pair,
    // swaps the pairs and leaves the handle as a producer-side partition always left it.  on implies pre.valid.
    struct LazyItems {
        bool on = false;
        bool armed = false;      // set by the item partition; the producer turns it into `on` beside pre.valid = true (after pending_changed)
        int src = 0;             // 1: the built-in generator's items (espitem), 2: an element-level append's (espelem)
        bool k32 = true;         // the expansion writes 4-byte keys (pre.key_bytes == 4)
        espitem::Args it;        // the expansion's argument block (sorted_keys set; keys_out / vals_out filled in by lazy_expand)
        espelem::Args el;
    } lazy;
    DevBuf lazy_hold;            // esp_append_elements_host: the uploaded element matrices of a batch that stayed a list of items (the fused
                                 // bucket kernel or lazy_expand gathers from them); released by the next flush / reset / upload
    int last_rebuild = 0;        // the last flush's tail rebuilt the matrix (flush_rebuild: the stored CSC as the first piece of a fresh flush)
    int last_lazy_items = 0;     // the last flush's bucket kernel formed its updates from item records (esp_debug_last_lazy_items)
    int last_sum_join = 0;     // esp_debug_last_sum_join
    int last_sum_plan_bits[2] = {0, 0};  // esp_debug_last_sum_plan_bits: smallest / largest prefix the buffers of the last joint esp_flush_sum had planned
    double last_sum_ms[2] = {0.0, 0.0};  // esp_debug_last_sum_ms: host wall-clock of the last esp_flush_sum's folds / gather + combine flush
    // The entries appended behind a batch over a STORED pattern were partitioned as they came (append_tail_partitioned):
    // pre.tail packed keys in bucket order of a plan of their own -- still a pending stream like any other (a stable
    // partition keeps every column's order), so whoever does not know about it loses nothing; esp_flush's split
    // starts their flush at the bucket kernel.  Segment starts in tseg.
    struct TailPart {
        bool valid = false;
        int K = 0, pb = 0;
        i64 T = 0, maxlen = 0;
        u64 base = 0, span = 0;
    } tailpart;
    // A caller that repeats its stream (a time-stepping code: the same triplets' positions every step): the run lists,
    // run offsets and bucket starts of the last append-is-the-partition of caller-supplied triplets (append_partitioned)
    // serve the next one -- no count pass over the columns; the scatter kernel checks every tile against its run list
    // and a stream that is not the same falls back to the full path.  Dropped by whatever rewrites the tables.
    struct RawPlan {
        bool valid = false;
        i64 count = 0, chunks = 0, maxlen = 0, maxlen_c = 0;
        int kind = 0, K = 0, pb = 0, key_bytes = 8, fb = 0;
        u64 base = 0, span = 0;
        double Ee = 0.0;
    } rawplan;
    // The same for the stencil generator (esp_generate_fdrand*): its run lists, run offsets and bucket starts are a function of
    // the grid, the node range and the plan alone -- not of seed, values or kind -- so an assembly that repeats the previous
    // one (a time loop: reset!, fdrand!, flush!) goes straight to the PART launch: no COUNT launch, no ranking launches, no
    // host round trip for their flags.  Dropped by whatever rewrites the tables (chunk_arrays, sort_msd, release_buffers).
    // A caller's triplets of one kind on an empty buffer that are NOT a pre-sorted stream (a shuffled assembly): the first radix
    // pass of their flush ran while they were appended (append_first_pass: keys formed from rows / cols on the fly, values read
    // where the caller holds them -- no packed copy in stream order is written and read again).  The buffer holds packed keys in
    // the order of that pass -- a stable permutation of the stream: to everybody who does not know, an ordinary pending buffer --
    // and sort_msd resumes behind it (segment / tile tables in seg[cur] / tilef[cur]).  Dropped by pending_changed.
    struct PrePass {
        bool valid = false;
        i64 count = 0, maxlen = 0;
        int K = 0, planned = 0, npass = 0, bits0 = 0, cur = 0, S = 0;
        u64 base = 0, span = 0;
        double Ee = 0.0;
    } prepass;
    struct GenPlan {
        bool valid = false;
        i64 nx = 0, ny = 0, nz = 0, g0 = 0, g1 = 0, E = 0;
        int kind = 0;
        u64 base = 0, span = 0;
        const void *keys_at = nullptr;  // (the arrays the stored PartOut wrote to: a reallocation drops the plan)
        esprun::PartOut out;
        PrePart pre;
    } genplan;
    // esp_elements_keep_plan: the item order, the cell records and the segment table of the last esp_append_elements on an
    // empty buffer (cells of 3 / 4 nodes) are kept in buffers of their own, so that esp_append_elements_again -- the same
    // connectivity, new element matrices: a time step of an instationary / nonlinear code -- goes straight to the expansion
    struct ElemPlan {
        bool keep = false, valid = false;
        int nloc = 0, W = 0, vrb = 0, rem_real = 0, K = 0, kind = 0;
        bool k32 = false, has_diag = false;
        i64 ncells = 0, S = 0, maxlen = 0;
        u64 base = 0, span = 0;
        DevBuf sorted, cellrec, segtab;
    } elemplan;
    // sort_msd over ITEM records (femitems.hpp): a segment may hold plan_cap records (the bucket kernel's capacity in
    // updates / updates per item), and a shuffled stream need not be tried as a pre-sorted one
    i64 plan_cap = 0;
    bool plan_try_runs = false;  // item records of an element batch whose cell order looked pre-sorted to elem_cells_k: sort_msd tries the run-based pass
    u64 plan_occ_span = 0;       // > 0: the records to partition occupy only this many keys of the window (the columns one band of a mesh touches): the plan counts them as that dense
    int plan_bits = 0;           // > 0: sort_msd resolves exactly this many prefix bits in its planned passes (item partitions whose expansion does the last bits itself: segexpand.hpp)
    // the pending entries start at this entry of keys/vals (behind a batch that esp_flush flushed by itself); else 0
    i64 pend_off = 0;
    bool item_mode = false;
    bool item_keys_only = false;  // ... whose records are single words (the value arrays are not touched)
    bool shard_user = false;     // the handle is driven through esp_shard_*: its flushes partition by owner first
    int last_shard_source = 0;   // esp_shard_partition: 1 = its own pass moved the entries, 2 = the producer had
    int last_local_small = 0;    // the last flush's bucket kernel was the small variant (3 workgroups per CU)
    // esp_shard_plan: the producers that find the buffer empty partition for the next esp_shard_partition(P, me, eps)
    struct ShardPlan {
        bool valid = false;
        int P = 0, me = 0;
        i64 eps = 0;
    } shard_plan;
    // column window of the pending entries (whole matrix by default)
    u64 win_base = 0, win_span = 0;
    // A shard works on its column range only.  When the window [wc0, wc1) (0-based columns) was
    // declared on an empty matrix and kept since, no entry can lie outside it: the per-column work of a
    // flush (colend clear, colptr scan) then runs over the window only -- with P shards the global colptr
    // has P times the columns a shard owns.  colptr[c] = 1 for c <= wc0 always; the part behind the window
    // (= nnz+1) is rewritten only when somebody needs the whole array (tail_stale).
    i64 wc0 = 0, wc1 = 0;
    bool win_excl = false, tail_stale = false;
    // reset! of an unwindowed matrix leaves colptr := 1 to whoever reads it next (fix_tail): the fresh flush that
    // normally follows rewrites every entry
    bool ones_pending = false;
    // device CSC (Julia layout) + spare set for rebuilds
    DevBuf colptr, rowval, nzval, rowval2, nzval2;
    DevBuf colptr2;  // the join writes the new colptr here (its tiles hold both summands), then the two are swapped
    i64 nnz = 0;
    bool csc_valid = false;  // colptr initialised
    // host staging (pinned) + device staging
    // pinned staging areas (+ their device mirrors): `stage` is the one esp_stage_begin hands to the caller
    // (its pointers stay valid until the caller asks for a larger one); `bulk` is private to esp_append_host
    struct StageArea {
        i64 cap = 0;
        i64 *rows = nullptr, *cols = nullptr;
        double *vals = nullptr;
        uint8_t *kinds = nullptr;
        DevBuf d_rows, d_cols, d_vals, d_kinds;
    } stage, bulk;
    // esp_commit of a staged chunk: the chunk is packed on the host (keys + values) into one of two pinned halves and leaves
    // for the append buffer asynchronously -- the caller refills its chunk while the transfer runs
    struct CommitPack {
        i64 cap = 0;  // entries per half
        u64 *keys[2] = {nullptr, nullptr};
        double *vals[2] = {nullptr, nullptr};
        hipEvent_t done[2] = {nullptr, nullptr};
        bool busy[2] = {false, false};
        int next = 0;
    } cpack;
    // two pinned bounce buffers + their events for transfers between PAGEABLE host memory and the device (d2h_pipelined,
    // h2d_pipelined, the narrowed downloads): made on first use, nothing on the device side
    struct Bounce {
        char *pin[2] = {nullptr, nullptr};
        size_t bytes = 0;  // each
        hipEvent_t ev[2] = {nullptr, nullptr};
    } bounce;
    unsigned long long *pin_scalar = nullptr;  // pinned, 8 slots
    u64 *pin_mw = nullptr;  // pinned source of prepart_begin's asynchronous upload of the window bases (<= MW_MAX)
    hipEvent_t pin_mw_done = nullptr;
    // kind bookkeeping of the pending batch: when every pending entry was appended with ONE known kind the run-based
    // partition hands the bucket kernel 4-byte keys (the key bits below the partition prefix) instead of packed keys
    i64 kind_noted = 0;     // pending entries appended with a single known kind
    int kind_uniform = -1;  // that kind; -1 none yet, -2 mixed / an append of unknown kinds (until the buffer is empty again)
    int last_key_bytes = 8;      // esp_debug_last_key_bytes
    // longest segment / average segment of the last bucket-path flush (0: not known): an assembly that repeats on
    // a handle with regular data (spread ~1.0x) is planned one partition bit tighter when the predicted longest
    // segment still fits the bucket kernel -- half-full segments cost that kernel up to 1.8x
    double seen_spread = 0.0;
    int last_fold_update = 0;    // the register tiers of the last flush ran their UPDATE-only fold
    bool part_own32 = false;       // the own range of the partitioned buffer holds 4-byte keys (kind part_kind32)
    int part_kind32 = 0;
    bool part_own_update = false;  // esp_shard_partition: every pending entry was appended as an UPDATE
    bool part_all_update = false;  // esp_shard_assemble: ... and so is every received entry (checked on the device)
    int last_run_order = 0;      // esp_debug_last_run_order
    int last_plan_reused = 0;    // the last append-is-the-partition of caller triplets used the previous assembly's run lists
    int last_colptr_direct = 0;  // the bucket kernel of the last flush wrote colptr itself
    hipStream_t aux = nullptr;   // second stream + event: small device-to-host reads beside a running kernel
    hipEvent_t aux_ev = nullptr;  // (created on first use, aux_ready)
    // shard cache
    bool shard_valid = false;
    int shard_P = 0;
    // partitioned exchange (esp_shard_partition / esp_shard_assemble)
    bool part_valid = false;      // the pending buffer is partitioned by (owner, digit); tables in parttab
    bool part_assembled = false;  // piece tables are built: the next flush runs the bucket kernel on them
    int part_P = 0, part_me = 0, part_shift = 0;
    u32 part_nb = 0;
    u64 part_base = 0, part_span = 0;
    i64 part_total = 0, part_maxlen = 0, part_own_lo = 0;
    int part_fb = 0;              // the own range came from a producer's FINE partition: 2^part_fb buckets of seg[1] per digit
    DevBuf parttab, piecetab;
    // esp_flush_sum's general path: the buffers' folds as ONE flush of a scratch handle with p n columns (buffer k's entries in the
    // columns [k n, (k + 1) n)); made on first use, destroyed with this handle
    esp_handle *sumtmp = nullptr;
    DevBuf sumrange;                       // ... the buffers' column ranges (device side of one round trip)
    int last_sum_batched = 0;              // the last esp_flush_sum folded its buffers in one flush of sumtmp
    DevBuf asmwork;                        // esp_shard_assemble's kernel: ticket | summary | per-workgroup partial results (zeroed once)
    unsigned long long *pin_asm = nullptr;  // ... and the pinned block its last workgroup writes the results to
    unsigned long long asm_seq = 0;
    // row-wise view of the device CSC for mul! (built on first use after a pattern change)
    unsigned long long pattern_version = 1, csr_version = 0;
    unsigned long long values_version = 1, csr_val_version = 0;  // nzval changed / row-wise copy of the values
    DevBuf csr_rowptr, csr_perm, csr_col, csr_tmp, csr_val, mul_x, mul_r;
    // timing
    bool timing = false;
    int timing_level = 2;
    std::vector<hipEvent_t> ev_pool;
    std::vector<TimedSpan> spans;
    esp_timing_t acc;
    hipEvent_t flush_a = nullptr, flush_b = nullptr;
};

extern thread_local std::string g_err;

// the pending entries changed: whatever was derived from them is stale
static inline void pending_changed(esp_handle *h) {
    if (h->count == 0) {
        h->kind_noted = 0;
        h->kind_uniform = -1;
    } else if (h->kind_noted != h->count) {
        h->kind_uniform = -2;  // (some entries came or went without note_kind: sticky until the buffer is empty)
    }
    h->shard_valid = false;
    h->part_valid = false;
    h->part_assembled = false;
    // (whoever changes a bucket-ordered buffer called pending_materialize first -- or appends behind it)
    if (h->pre.valid && h->pre_keep && h->count >= h->pre.E)
        h->pre.tail = h->count - h->pre.E;
    else
        h->pre.valid = false, h->lazy.on = false;
    h->prepass.valid = false;
    h->pre_keep = false;
    h->tailpart.valid = false;  // (append_tail_partitioned sets it after this call)
}

// set-up of a producer-side partition (prepart_* below, next to run_partition)
struct PartSetup {
    bool on = false;
    esprun::PartOut out;   // for the producer's PART kernel
    esprun::RunSink sink;  // for its COUNT kernel
    u32 *err = nullptr;    // window flag of the COUNT kernel
    int K = 0, pb = 0, kind = -1;
    int fb = 0;            // fine partition: pb holds fb bits more than the bucket kernel's segments need
    i64 E = 0, chunks = 0;
    double Ee = 0.0;
    i64 NB = 0;            // buckets (1 << pb, or shards * digits per shard)
    int mw_P = 0, mw_me = 0, mw_shift = 0;
    u32 mw_nb = 0;
    i64 mw_eps = 0;
    i64 *seg_out = nullptr, *runs_off = nullptr;
    const unsigned long long *bucket_count = nullptr;
    u64 *coarse = nullptr;
    const u32 *dcount = nullptr;
    const u64 *dlist = nullptr;
};

// call right before h->count grows by cnt entries that all carry `kind`
static inline void note_kind(esp_handle *h, int kind, i64 cnt) {
    if (h->count == 0 && h->kind_noted == 0 && h->kind_uniform == -1) h->kind_uniform = kind;
    else if (h->kind_uniform != kind) h->kind_uniform = -2;
    h->kind_noted += cnt;
}

#define FAIL(h, code, ...)                                   \
    do {                                                     \
        char _b[512];                                        \
        snprintf(_b, sizeof _b, __VA_ARGS__);                \
        if (h) (h)->err = _b;                                \
        g_err = _b;                                          \
        return (code);                                       \
    } while (0)

#define HIPCK(h, call)                                                                          \
    do {                                                                                        \
        hipError_t _e = (call);                                                                 \
        if (_e != hipSuccess)                                                                   \
            FAIL(h, _e == hipErrorOutOfMemory ? ESP_ERR_NOMEM : ESP_ERR_HIP, "%s failed: %s",   \
                 #call, hipGetErrorString(_e));                                                 \
    } while (0)

#define CK(...)                       \
    do {                              \
        int32_t _s = (__VA_ARGS__);   \
        if (_s != ESP_OK) return _s;  \
    } while (0)


// ---- records of the plan and the partition (partition.hip)
// ------------------------------------------------------------------------ flush
// MSD plan: partition on the top key bits until every segment fits the LDS bucket kernel.
// Returns local_ok=false when the general (global LSD + fold_k) path must be used instead.
struct Sorted {
    const u64 *sk;
    const double *sv;
    bool in_primary;  // data in h->keys/vals (true) or h->keys2/vals2 (false)
    int S;
    const i64 *seg_start;
    int rem_bits;
    bool local_ok;
    bool fits = false;  // every segment is within seg_cap (local_ok without the limit on the remaining key bits)
    bool all_update = false;  // PIECES: every entry of every piece is an UPDATE (esp_shard_assemble checked)
    int key_bytes = 8;  // 4: sk holds 32-bit keys (the bits below the prefix); every entry has the kind `kind`
    int fb = 0;         // 4-byte keys of a FINE partition: seg_start has (S << fb) + 1 entries, segment s = its buckets [s << fb, (s + 1) << fb)
    int k32_passes = 0;  // the 8-bit passes that wrote 4-byte keys (sort_msd): > 0 = the other pair holds no packed copy of the entries
    int kind = 0;
    i64 maxlen = esplocal::CAP;  // longest segment
    i64 total = -1;              // entries of all segments, if the caller knows (lets flush_local drop the segments behind the last column)
    int p32_piece = -1;          // PIECES: the piece that holds 4-byte keys of kind `kind` from position p32_lo on
    bool all32 = false;          // PIECES: EVERY piece holds 4-byte keys of kind `kind`
    i64 p32_lo = 0;
    // PIECES (partitioned shard exchange): segments are concatenations of per-source pieces
    int npieces = 0;
    const i64 *own_fine = nullptr;  // PIECES, fb > 0: the fine table of the piece p32_piece (local_args.hpp, Args::own_fine)
    bool pieces_dense = false;   // PIECES: most segments hold entries of several pieces (a batch and its tail, a stored slice and new entries)
    const i64 *pstart = nullptr;
    const void *const *ptab = nullptr;
    const esp_handle::LazyItems *lazy = nullptr;  // sk holds sorted ITEM records, seg_start counts their updates: the fused bucket kernel or nothing
    bool has_base = false;  // the segments' key base, if it is not the handle's window / shard range
    int expect_hits = -1;   // 1 / 0: the caller knows what the entries will mostly do over the stored pattern; -1: the handle's history
    u64 base = 0;
};

// ---- run lists of the pending entries (runpart.hpp) -------------------------------------------
// Persistent per-handle arrays: every chunk's runs, the digits' own run lists, the bucket totals.  They are
// filled by run_hist_k at flush time or by the COUNT launch of a producer whose append is the partition.
struct ChunkArrays {
    u32 *runs_d, *runs_c;
    u64 *nruns;
    unsigned long long *bucket_count;
    u32 *overflow;
    u32 *dcount;  // (directly in front of bucket_count: one memset clears both)
    u64 *dlist;
    u64 *coarse;  // totals of 256 digits each
    size_t clear_bytes;  // dcount .. bucket_count[NB]
};

// Single-pass partition on the top `pb` (9..20) bits of the key window, for pre-sorted streams
// (runpart.hpp).  *ok=false when some chunk holds too many distinct digits: nothing was moved and the
// caller uses the 8-bit passes.  On success kout/vout hold the partitioned entries, seg_out (NB+1
// entries, device) the bucket starts and tile_first_out the tile index of every bucket.
// several key windows side by side (column shards): see esprun::Args
struct MultiWin {
    int P;
    u32 nb;
    const u64 *d_base;
};

struct MwPlan {
    bool ok = false;
    int K = 0, shift = 0, pb = 0;
    u64 nb64 = 0;
    i64 NB = 0;
    // FINE partition of a producer's batch (33 .. 36 key bits below the plan's prefix -- the shards of 512^3 over 8 GPUs: 35): the
    // producer may cut every digit into 2^fb buckets (tables of NB << fb buckets, digit width 2^(shift - fb) = 2^32), so that its
    // OWN range holds 4-byte keys; the exchange, the piece tables and the bucket kernel's segments stay the plan's digits
    int fb = 0;
    std::vector<u64> base;
};

// mw != nullptr: buckets = mw->P * mw->nb (window r = digits [r*nb, (r+1)*nb)), pb = bits covering them
// triplets of one kind as the source of a partition (esprun::Args::raw_*)
struct RawSource {
    const i64 *rows, *cols;
    int kind, negate;
    unsigned long long *d_err;
};

#pragma GCC visibility push(hidden)
int32_t ensure(esp_handle *h, DevBuf &b, size_t need, bool keep = false);
void release(DevBuf &b);
void release_all(esp_handle *h);
hipEvent_t ev_get(esp_handle *h);
void timing_collect(esp_handle *h);
int32_t fix_tail(esp_handle *h);
int32_t init_empty_csc(esp_handle *h);
int32_t reserve_append(esp_handle *h, i64 add);
int32_t pack_device(esp_handle *h, const i64 *d_rows, const i64 *d_cols, const double *d_vals,
                           const uint8_t *d_kinds, int kind_all, int op, i64 count);
int32_t ensure_stage(esp_handle *h, esp_handle::StageArea &sa, i64 want);
int32_t ensure_bounce(esp_handle *h);
void par_memcpy(void *dst, const void *src, size_t bytes);
void host_run_parts(int parts, const std::function<void(int)> &fn);
int32_t d2h_pipelined(esp_handle *h, void *dst, const void *d_src, size_t bytes);
// pageable host memory -> device through the same two pinned bounce buffers (the host copy of chunk i+1 overlaps the transfer
// of chunk i); returns when the device holds the data
int32_t h2d_pipelined(esp_handle *h, void *d_dst, const void *src, size_t bytes);
i64 fd_offset_host(i64 nx, i64 ny, i64 nz, i64 g);
int32_t prepart_begin(esp_handle *h, i64 E, i64 chunks, int kind, PartSetup *ps);
int32_t prepart_rank(esp_handle *h, PartSetup *ps);
int32_t prepart_finish(esp_handle *h, PartSetup *ps, bool *took);
int32_t pending_materialize(esp_handle *h);
namespace esplocal {
bool launch_group3_items(const esp_handle::LazyItems &lz, unsigned grid, hipStream_t stream, const Args &a, bool hits = false);  // local_j.hip
bool launch_group3_items_multi(int nloc, bool diag, const u32 *vlist, const MultiBuf *mbuf, i64 *counts, int S_real, unsigned grid,
                               hipStream_t stream, const Args &a);  // local_j.hip
}
constexpr int32_t ESP_RETRY_EXPANDED = 1000;  // flush_local to esp_flush: expand the items (lazy_expand) and call again -- never leaves the library
int32_t lazy_expand(esp_handle *h);   // produce.hip: the expansion of a batch held as sorted items (esp_handle::LazyItems)
// may an item partition on this handle leave its batch unexpanded?  (kind: what its updates are; produce.hip)
bool lazy_items_wanted(const esp_handle *h, int kind);
int32_t settle_offset(esp_handle *h);
int32_t item_produce_fem(esp_handle *h, const espgen::FemArgs &fa, i64 E, bool *took);
int32_t partition_pass(esp_handle *h, espradix::Pass &p, i64 max_tiles);
int32_t sort_pending_lsd(esp_handle *h, const u64 **sk, const double **sv);
int32_t chunk_arrays(esp_handle *h, i64 Ccap, int pb, ChunkArrays *out, bool keep_plan = false);
double plan_entries(i64 E, int K, u64 span);
int plan_run_bits(i64 E, int K, u64 span);
int plan_prefix_bits(const esp_handle *h, i64 E, int K, double *Ee_out);
int plan_local_bits(esp_handle *h, i64 NI, int W, int K, int *sort_bits);
int window_bits(const esp_handle *h);
int32_t aux_ready(esp_handle *h);
int32_t run_partition(esp_handle *h, const u64 *kin, const double *vin, u64 *kout, double *vout, int K, int pb,
                             i64 *seg_out, u64 *tile_first_out, bool *tiles_ready, bool *ok, i64 *maxlen_out,
                             const MultiWin *mw = nullptr, int mw_shift = 0, bool allow_k32 = false, int *key_bytes_out = nullptr,
                             i64 E_in = -1, const RawSource *raw = nullptr);
int32_t append_first_pass(esp_handle *h, const i64 *d_rows, const i64 *d_cols, const double *d_vals, int kind, int op, i64 count, bool *took);  // partition.hip
int32_t append_tail_partitioned(esp_handle *h, const i64 *d_rows, const i64 *d_cols, const double *d_vals, int kind, int op, i64 count,
                                 bool *took);
int32_t append_partitioned(esp_handle *h, const i64 *d_rows, const i64 *d_cols, const double *d_vals, int kind, int op, i64 count,
                                  bool *took);
int32_t sort_msd(esp_handle *h, Sorted *out);
int32_t finish_csc(esp_handle *h, i64 Z0, i64 Zn, const u64 *new_key, const double *new_val);
int32_t prepare_outputs(esp_handle *h, i64 Z0, i64 Zn);
int32_t flush_local(esp_handle *h, const Sorted &st, int mode, i64 *Zn_out);
int32_t flush_global(esp_handle *h, int mode, i64 *Zn_out);
int32_t flush_pre_tail(esp_handle *h, int mode, i64 *Zn, bool *served);
int32_t build_csr(esp_handle *h);
int32_t dirichlet_call(esp_handle *h, uint8_t *marker, int32_t on_device, bool mark, double penalty);
int32_t diag_setup(esp_handle *h, double *inv, int64_t *idiag, int32_t on_device, const char *what);
int32_t shard_prepare(esp_handle *h, int P, espradix::Pass *out);
int32_t shard_offsets(esp_handle *h, int P, int64_t *offsets /* P+1 */);
#pragma GCC visibility pop

struct Span {
    esp_handle *h;
    int stage;
    hipEvent_t a = nullptr;
    int launches = 0;
    Span(esp_handle *hh, int st) : h(hh), stage(st) {
        // timing level 1 brackets the big kernels only: the ~20 tiny launches of the "scan" stage would cost
        // more in event records (two per span) than they run
        if (h->timing && (h->timing_level == 2 || (h->timing_level == 1 && st != ESP_ST_SCAN) ||
                          (h->timing_level == 3 && (st == ESP_ST_LOCAL || st == ESP_ST_FOLD)))) {
            a = ev_get(h);
            (void)hipEventRecord(a, h->stream);
        }
    }
    void add(int l) { launches += l; }
    ~Span() {
        if (h->timing && a) {
            hipEvent_t b = ev_get(h);
            (void)hipEventRecord(b, h->stream);
            h->spans.push_back({stage, a, b, launches});
            if (h->spans.size() > 2048) timing_collect(h);
        }
    }
};

// ------------------------------------------------------------------------ small kernels
static __global__ void set_i64_k(i64 *p, i64 a, i64 b, i64 c, i64 d) {
    p[0] = a;
    p[1] = b;
    p[2] = c;
    p[3] = d;
}
static __global__ void fill_i64_k(i64 *p, i64 n, i64 v) {
    const i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (g < n) p[g] = v;
}

// The 64-byte block of partition results (longest bucket + four flag words) goes to pinned HOST memory with plain
// stores: the host then needs neither a copy engine nor a blit kernel -- which may queue behind the kernel that fills
// the chip -- to read it, only the event recorded behind this launch.
static __global__ void publish_block_k(const unsigned long long *__restrict__ src, unsigned long long *__restrict__ host_dst) {
    if (threadIdx.x < 8) __hip_atomic_store(&host_dst[threadIdx.x], src[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// the longest run of 2^fb neighbouring buckets (one segment of a fine partition's flush: esp_handle::PrePart::fb)
static __global__ void coarse_seg_max_k(const i64 *__restrict__ seg, i64 S_coarse, int fb, unsigned long long *__restrict__ out) {
    const i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const u32 len = g < S_coarse ? (u32)min((i64)0xFFFFFFFFll, seg[(g + 1) << fb] - seg[g << fb]) : 0u;
    const u32 m = esp_wave_max(len);
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, (unsigned long long)m);
}

static inline unsigned grid_for(i64 n, int threads) { return (unsigned)std::max<i64>(1, ceil_div<i64>(n, threads)); }


// ---- small helpers
static inline bool windowed(const esp_handle *h) { return h->win_excl && (h->wc0 > 0 || h->wc1 < h->n); }

// first entry and number of entries of the per-column arrays (colptr, colend: n+1 entries) a flush touches
static inline void col_range(const esp_handle *h, i64 *c0, i64 *cnt) {
    *c0 = windowed(h) ? h->wc0 : 0;
    *cnt = windowed(h) ? h->wc1 - h->wc0 + 1 : h->n + 1;
}

// The digits of a partition cut the 2^K keys of the window's bit range, of which only `span` exist (a matrix
// with 2^k + 1 columns fills half of it): the plan counts the entries as if the empty part were filled as well,
// so that the occupied buckets come out at the planned fill.
// planned average fill of a segment (fraction of the bucket kernel's capacity); ESP_PLAN_FILL overrides (experiments)
static double plan_fill() {
    static const double f = [] {
        const char *e = esp_exp_env("ESP_PLAN_FILL");
        const double v = e ? atof(e) : 0.9;
        return v > 0.1 && v <= 1.0 ? v : 0.9;
    }();
    return f;
}

// records a segment of the partition may hold: the bucket kernel's capacity, or what an item partition says (plan_cap)
static inline i64 seg_cap(const esp_handle *h) { return h->plan_cap > 0 ? h->plan_cap : (i64)esplocal::CAP; }

// The plan of the partition by (owner, digit inside the owner's column range): every rank derives the same one from
// (n, P, entries_per_shard).  ok = false: small or odd problem (the plain exchange serves it).
static inline i64 shard_col0(i64 n, int P, int r) { return (i64)(((__int128)r * (__int128)n + P - 1) / P); }  // ceil(r*n/P)
static MwPlan shard_mw_plan(const esp_handle *h, int P, i64 entries_per_shard) {
    MwPlan m;
    m.base.resize((size_t)P);
    u64 maxspan = 1;
    for (int r = 0; r < P; r++) {
        const i64 c0 = shard_col0(h->n, P, r), c1 = shard_col0(h->n, P, r + 1);
        m.base[(size_t)r] = (u64)c0 << h->L.rb;
        maxspan = std::max(maxspan, (u64)(c1 - c0) << h->L.rb);
    }
    int K = 1;
    while (K < 62 && ((u64)1 << K) < maxspan) K++;
    const int pbw = plan_run_bits(std::max<i64>(entries_per_shard, 1), K, maxspan);
    if (pbw == 0 || K - pbw > esplocal::MAX_REM_BITS) return m;
    m.K = K;
    m.shift = K - pbw;
    m.nb64 = ((maxspan - 1) >> m.shift) + 1;
    m.NB = (i64)m.nb64 * P;
    if (m.NB > ((i64)1 << 24)) return m;
    m.pb = 1;
    while (((i64)1 << m.pb) < m.NB) m.pb++;
    if (m.shift > 32 && m.shift - 32 <= 4 && h->L.rb <= 32 && (m.NB << (m.shift - 32)) <= ((i64)1 << 24)) m.fb = m.shift - 32;
    m.ok = true;
    return m;
}

// in-place exclusive scan with its workspace in a handle buffer
// exclusive scan helpers with scratch carved from h->misc
template <typename T, bool MAX>
static int32_t scan_inplace(esp_handle *h, T *data, i64 n, DevBuf &ws, int *launches) {
    CK(ensure(h, ws, sizeof(T) * (size_t)espscan::workspace_elems(n)));
    *launches += espscan::exclusive<T, MAX>(h->stream, data, data, n, (T *)ws.p);
    return ESP_OK;
}
