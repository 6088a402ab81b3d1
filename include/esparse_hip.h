/*
 * esparse_hip.h -- C ABI of libesparse_hip.so, the MI355X (gfx950) sparse-assembly
 * backend for ExtendableSparse.jl's ExtendableSparseMatrix.
 *
 * What it replaces (paths relative to the reference repository):
 *   - the CPU insertion buffer SparseMatrixLNK  (src/matrix/sparsematrixlnk.jl:21-253)
 *     becomes a device-resident COO append buffer (packed u64 key + f64 value);
 *   - flush! / `lnk + csc`  (src/matrix/extendable.jl:248-255,
 *     src/matrix/sparsematrixlnk.jl:294-383) becomes a HIP pipeline: stable radix
 *     partition on (col,row) keys, LDS-staged ordered fold of duplicates,
 *     merge-path join with the existing CSC.
 * The entry points are what the reference's plugin slot for extension buffers
 * (src/matrix/abstractsparsematrixextension.jl:6-14, used by
 * src/matrix/genericextendablesparsematrixcsc.jl:14-92 and
 * src/matrix/genericmtextendablesparsematrixcsc.jl:16-114) needs from a foreign
 * buffer; INTEGRATION.md shows the Julia `ccall` shim that binds them.
 *
 * Conventions
 *   - every function returns an int32 status (0 = ok, <0 = esp_status); nothing
 *     throws or aborts across the boundary; esp_last_error() gives the message;
 *   - all matrix indices are 1-based Int64, exactly as Julia passes them;
 *   - output arrays are caller-allocated (Julia Vectors) after a size query;
 *   - one handle = one device + one HIP stream; a handle is not thread-safe
 *     (like one reference buffer per `tid`); distinct handles are independent and may be
 *     driven from different host threads concurrently -- fills, flushes, transfers, first
 *     use included (genericmtextendablesparsematrixcsc.jl:87-99: one buffer per task;
 *     tests/test_concurrent_handles.py; the corruption rounds 4 and 5 saw here was a
 *     missing barrier inside the bucket kernel, NOTES/round6.md section 1).  The library
 *     starts no threads of its own on the flush path, with one exception: when esp_flush_sum
 *     cannot fold its buffers in one flush (esp_debug_last_sum_batched 0: a column window or a
 *     test hook on a buffer) it folds them one by one, side by side on its host-copy pool (at
 *     most eight threads with the caller's; ESP_HOST_THREADS caps it);
 *   - element types: Float64 values, Int64 indices (the reference's default
 *     ExtendableSparseMatrix{Float64,Int64}); other Tv/Ti stay on the CPU path.
 */
#ifndef ESPARSE_HIP_H
#define ESPARSE_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct esp_handle esp_handle;

typedef enum {
    ESP_OK = 0,
    ESP_ERR_INVALID = -1,     /* bad argument / NULL handle                       */
    ESP_ERR_BOUNDS = -2,      /* (i,j) outside 1..m x 1..n  -> Julia BoundsError  */
    ESP_ERR_HIP = -3,         /* a HIP runtime call failed                         */
    ESP_ERR_NOMEM = -4,       /* device or pinned allocation failed                */
    ESP_ERR_UNSUPPORTED = -5, /* dimension / count outside the supported range     */
    ESP_ERR_STATE = -6,       /* call not valid in the handle's current state      */
    ESP_ERR_NODEVICE = -7     /* no usable GPU (the product never falls back)      */
} esp_status;

/* How an appended entry combines with what is already stored at (i,j).
 * SET       = Base.setindex!      (sparsematrixlnk.jl:178-201, extendable.jl:205-218)
 * UPDATE    = updateindex!(A,+..) (sparsematrixlnk.jl:210-228, extendable.jl:159-174)
 * RAWUPDATE = rawupdateindex!     (sparsematrixlnk.jl:237-253, extendable.jl:181-197)
 * COO       = a triplet of sparse(I,J,V,m,n,+), i.e. of the COO constructors (extendable.jl:85-104) and
 *             fdrand_coo (sprand.jl:134-185): always creates, the first value as it is, duplicates
 *             added in input order */
typedef enum { ESP_SET = 0, ESP_UPDATE = 1, ESP_RAWUPDATE = 2, ESP_COO = 3 } esp_kind;

/* `op` of updateindex!/rawupdateindex!.  Every call site of the reference passes `+`;
 * `-` is exact as `+` of the negated value.  Other functions stay on the CPU path. */
typedef enum { ESP_OP_ADD = 0, ESP_OP_SUB = 1 } esp_op;

/* Which reference operation esp_flush() evaluates:
 * ROUTED = flush!(ExtendableSparseMatrixCSC): every appended update of an (i,j) that
 *          is already in the CSC is applied to csc.nzval in call order
 *          (extendable.jl:164-166,188-189,210-211), the others fold in the buffer and
 *          are merged in (extendable.jl:248-255);
 * PLUS   = Base.:+(buffer, csc) (sparsematrixlnk.jl:294-383): the buffer folds on its
 *          own and equal positions give csc.nzval + buffer value (:363).            */
typedef enum { ESP_FLUSH_ROUTED = 0, ESP_FLUSH_PLUS = 1 } esp_flush_mode;

/* pipeline stages reported by esp_timing() */
enum {
    ESP_ST_APPEND = 0, /* pack / generator kernels (coalesced stores into the buffer) */
    ESP_ST_HIST = 1,   /* digit histograms of the radix partition                      */
    ESP_ST_SCAN = 2,   /* exclusive scans (offsets, colptr)                           */
    ESP_ST_SCATTER = 3,/* stable radix partition passes (global)                      */
    ESP_ST_LOCAL = 4,  /* LDS bucket sort + ordered fold + emit                       */
    ESP_ST_FOLD = 5,   /* ordered segmented fold on globally sorted entries           */
    ESP_ST_COLPTR = 6, /* column ends -> colptr                                       */
    ESP_ST_MERGE = 7,  /* merge-path join with the existing CSC                       */
    ESP_ST_COPY = 8,   /* H2D / D2H copies issued by this library                     */
    ESP_ST_COUNT = 9
};
typedef struct {
    double ms[ESP_ST_COUNT];        /* summed hipEvent time per stage since the last clear */
    int64_t launches[ESP_ST_COUNT]; /* kernel launches (or copies) per stage               */
    double flush_ms;                /* whole esp_flush calls (device time)                 */
    int64_t flushes;
} esp_timing_t;

/* ---- lifetime -------------------------------------------------------------------
 * esp_create: the constructor T_ext(m,n) of the plugin contract
 * (abstractsparsematrixextension.jl:8; SparseMatrixLNK{Tv,Ti}(m,n),
 * sparsematrixlnk.jl:75-77).  capacity_hint = expected number of appended entries. */
int32_t esp_create(int64_t m, int64_t n, int32_t device, int64_t capacity_hint, esp_handle **out);
int32_t esp_destroy(esp_handle *h);
/* Base.copy(ext) (extendable.jl:279-285): same CSC, same pending entries, same window; device-to-device */
int32_t esp_clone(esp_handle *h, esp_handle **out);
/* frees every device and pinned buffer of the handle NOW (the Generic wrappers drop their buffer after every flush!
 * -- genericextendablesparsematrixcsc.jl:34 -- and Julia's GC does not see device memory); the handle stays valid:
 * an empty matrix with an empty buffer.  Staging pointers from esp_stage_begin are invalid afterwards. */
int32_t esp_release_buffers(esp_handle *h);
const char *esp_last_error(const esp_handle *h);
const char *esp_version(void);
/* use an external HIP stream (hipStream_t) instead of the handle's own */
int32_t esp_set_stream(esp_handle *h, void *hip_stream);
int32_t esp_synchronize(esp_handle *h);
/* Base.size(ext) (abstractsparsematrixextension.jl:10) */
int32_t esp_size(const esp_handle *h, int64_t *m, int64_t *n);
/* packed key layout: key = ((col-1) << row_bits | (row-1)) << 2 | kind */
int32_t esp_key_layout(const esp_handle *h, int32_t *row_bits, int32_t *col_bits);

/* ---- append (replaces setindex!/updateindex!/rawupdateindex! on the buffer) -----
 * Host-fed: Julia fills a pinned chunk obtained from esp_stage_begin and commits it;
 * one ccall per chunk, not per entry.  kinds may be left untouched when kind_all>=0.
 * The chunk pointers stay valid until esp_stage_begin is called with a larger `want` (or esp_destroy);
 * no other call moves them (esp_append_host stages through an area of its own).  esp_commit returns
 * when the chunk may be refilled -- after it has been packed on the host (bounds are checked there: ESP_ERR_BOUNDS is
 * immediate and nothing of the chunk is appended) into one of two pinned halves; the transfer to the device runs while the
 * caller fills the chunk again. */
int32_t esp_stage_begin(esp_handle *h, int64_t want, int64_t **rows, int64_t **cols,
                        double **vals, uint8_t **kinds, int64_t *got);
int32_t esp_commit(esp_handle *h, int64_t count, int32_t kind_all, int32_t op);
/* bulk host arrays (COO constructor path extendable.jl:92-104): kinds==NULL -> kind_all */
int32_t esp_append_host(esp_handle *h, const int64_t *rows, const int64_t *cols,
                        const double *vals, const uint8_t *kinds, int32_t kind_all, int32_t op,
                        int64_t count);
/* the same for Int32 index arrays (ExtendableSparseMatrix{Float64,Int32}: extendable.jl:10-25 is generic in Ti; the CSC
 * the library hands back is Int64 all the same).  Both pack (row, col, kind) into keys on the host, whatever the index
 * type: 8-byte keys (16 bytes per entry over PCIe), or -- one kind for the batch (kinds == NULL) and at most 48 row + column
 * bits -- six-byte keys without the kind (14 bytes per entry), which a small kernel per chunk turns into packed keys on the
 * device */
int32_t esp_append_host_i32(esp_handle *h, const int32_t *rows, const int32_t *cols,
                            const double *vals, const uint8_t *kinds, int32_t kind_all, int32_t op,
                            int64_t count);
/* device-side producers: arrays already in HBM on this handle's device.  An index outside the matrix: ESP_ERR_BOUNDS, nothing is
 * appended; the message names AN offending entry -- the first of the batch when the batch was packed in stream order, the
 * smallest position among the tiles examined before the kernels gave up when the append was the partition */
int32_t esp_append_device(esp_handle *h, const int64_t *d_rows, const int64_t *d_cols,
                          const double *d_vals, const uint8_t *d_kinds, int32_t kind_all,
                          int32_t op, int64_t count);
/* entries already packed in this handle's key layout (used by the shard exchange) */
int32_t esp_append_packed(esp_handle *h, const uint64_t *d_keys, const double *d_vals,
                          int64_t count);
/* on-device update streams of the reference's workloads.  Called on an EMPTY buffer with a stream an assembly loop
 * emits (few column blocks per 256 nodes / 128 cells) the append is the partition: the entries are stored bucket by
 * bucket -- a stable permutation of the stream, so every (i,j) keeps its call order -- and esp_flush needs no partition
 * pass.  Any later append, clone or shard call first turns the batch back into an ordinary pending buffer.
 *
 * fdrand!(A,nx,ny,nz;update,rand) hot loop (src/matrix/sprand.jl:87-124); rand_mode
 * 0: ()->1, 1: 0.1+u (fdrand default, :232), 2: u; kind = ESP_UPDATE / ESP_RAWUPDATE;
 * testassemble! (test/femtools.jl:45-72) on a Kuhn grid with npd points per axis.    */
int32_t esp_generate_fdrand(esp_handle *h, int64_t nx, int64_t ny, int64_t nz, uint64_t seed,
                            int32_t rand_mode, int32_t kind);
int32_t esp_generate_fem(esp_handle *h, int32_t dim, int64_t npd, uint64_t seed,
                         int32_t order_mode);
/* the part of the fdrand! stream issued by the nodes [node_begin, node_end) (0-based l-1) of the
 * k,j,i loop nest: one rank's slab of a sharded assembly */
int32_t esp_generate_fdrand_range(esp_handle *h, int64_t nx, int64_t ny, int64_t nz, uint64_t seed,
                                  int32_t rand_mode, int32_t kind, int64_t node_begin,
                                  int64_t node_end);
/* Element-level append: the inner loops of a finite-element assembly (test/femtools.jl:61-69) as ONE call, for element data
 * the caller holds in arrays (Julia layouts, 1-based node numbers):
 *   for icell = 1:ncells, il = 1:nloc:  i = cellnodes[il,icell]
 *       diag != NULL:  update(A, diag[il,icell], i, i)                            (femtools.jl:64)
 *       for jl = 1:nloc:  update(A, elmat[il,jl,icell], i, cellnodes[jl,icell])   (femtools.jl:65-68)
 * with update = setindex! / updateindex! / rawupdateindex! as `kind` says (op as for esp_commit).  cellnodes: Int64
 * nloc x ncells (grid[CellNodes], femtools.jl:47), elmat: Float64 nloc x nloc x ncells (vol * S of femtools.jl:67), diag:
 * Float64 nloc x ncells or NULL; 1 <= nloc <= 16.  Bit-identical to the nloc * (nloc [+ 1]) per-entry calls per cell in
 * that order.  On an EMPTY buffer the library partitions one 8-byte record per (cell, local column) instead of the updates
 * and stores every update once, at its bucket position (the flush starts at the bucket kernel: esp_debug_last_partition 4);
 * a cell that names a node twice, a non-empty buffer or a column window: the updates are appended in stream order.
 * A node number outside 1..min(m,n): ESP_ERR_BOUNDS, nothing is appended.  The device form reads d_elmat (and, without cell
 * records, d_cellnodes / d_diag) AT FLUSH TIME: on a fresh matrix -- and over the pattern the same mesh built -- the batch stays a
 * list of sorted (cell, local column) items and the flush's bucket kernel forms the updates itself (csrc/group3_items.hpp: the
 * updates are never written to the append buffer).  The arrays must stay valid and unchanged until the handle's next esp_flush
 * (or esp_reset / esp_clear_pending / esp_destroy) has returned; the host form keeps its own device copy that long. */
int32_t esp_append_elements(esp_handle *h, int32_t nloc, int64_t ncells, const int64_t *d_cellnodes,
                            const double *d_elmat, const double *d_diag, int32_t kind, int32_t op);
int32_t esp_append_elements_host(esp_handle *h, int32_t nloc, int64_t ncells, const int64_t *cellnodes,
                                 const double *elmat, const double *diag, int32_t kind, int32_t op);
/* A time step of an instationary / nonlinear code assembles the SAME mesh again with new element matrices.
 * esp_elements_keep_plan(h, 1): from now on an esp_append_elements call on an empty buffer (cells of 3 or 4 nodes, the item
 * partition taken) keeps its plan -- the item order, the cell records, the segment table: 8 B per (cell, local column) + 64 B
 * per cell of device memory; on = 0 releases it.  esp_append_elements_again(h, d_elmat, d_diag, kind, op): the element loop
 * over the connectivity of that planned call (diag NULL iff it was then), on an empty buffer: no pass over the connectivity, no
 * item partition; the entries are bit for bit what esp_append_elements with the same cellnodes would append.  Without a plan:
 * ESP_ERR_STATE.  Together with the flush over the stored pattern (every update hits) a time step costs less than the
 * first assembly. */
int32_t esp_elements_keep_plan(esp_handle *h, int32_t on);
/* (d_elmat as for esp_append_elements: read at flush time, valid until the next esp_flush has returned) */
int32_t esp_append_elements_again(esp_handle *h, const double *d_elmat, const double *d_diag, int32_t kind, int32_t op);
int32_t esp_append_elements_again_host(esp_handle *h, const double *elmat, const double *diag, int32_t kind, int32_t op);
/* The element data of esp_generate_fem's grid as DEVICE arrays, for the cells at stream positions [cell_begin, cell_end):
 * what a caller of testassemble! holds (cellnodes) and computes per cell (elmat = vol * S, diag = 0.1 * vol / (dim+1);
 * femtools.jl:58-67) -- the producer of esp_append_elements' input in tests and bench.  node_mode 1: the nodes carry a
 * permuted numbering (a bijection of 1..n made from node_seed): nothing downstream can lean on grid arithmetic.
 * d_diag may be NULL. */
int32_t esp_generate_fem_mesh(esp_handle *h, int32_t dim, int64_t npd, uint64_t seed, int32_t order_mode,
                              int32_t node_mode, uint64_t node_seed, int64_t cell_begin, int64_t cell_end,
                              int64_t *d_cellnodes, double *d_elmat, double *d_diag);
/* number of appended, not yet flushed entries; >0 iff anything is pending
 * (the flush! gate of genericextendablesparsematrixcsc.jl:32 / nnznew :21) */
int32_t esp_pending(const esp_handle *h, int64_t *count);

/* ---- the CSC side ---------------------------------------------------------------
 * esp_set_csc: attach an existing SparseMatrixCSC (the `csc` operand of `ext + csc`). */
int32_t esp_set_csc(esp_handle *h, const int64_t *colptr, const int64_t *rowval,
                    const double *nzval, int64_t nnz);
/* THE flush: sort + ordered fold + join, result stays device-resident.
 * pattern_changed = 1 iff the CSC was rebuilt (the reference recomputes phash then). */
int32_t esp_flush(esp_handle *h, int32_t mode, int64_t *new_nnz, int32_t *pattern_changed);
/* Base.sum(xmatrices, csc) (sparsematrixdilnkc.jl:397-435) = flush! of GenericMTExtendableSparseMatrixCSC
 * (genericmtextendablesparsematrixcsc.jl:45-51) as ONE call: dst holds the CSC (esp_set_csc, or the result of its last
 * flush) and no pending entries; xs[0..p) are the partition buffers -- handles of the same size on the same device whose own
 * matrix is empty.  Result, bit for bit the reference's sparse!(I,J,V,m,n,+) over (csc entries, buffer 1's entries, ...):
 * at (i,j) ((csc + f_1) + f_2) + ... where f_k is the fold of the calls buffer k received at (i,j), buffers that do not
 * hold (i,j) skipped, zeros kept.  Every buffer is folded by itself on the device, the folds meet the stored matrix in ONE
 * routed flush of dst; no CSC travels between host and device whatever p is.  The buffers come back EMPTY (the reference
 * replaces them all, :47-49).  pattern_changed as for esp_flush. */
int32_t esp_flush_sum(esp_handle *dst, esp_handle *const *xs, int32_t p, int64_t *new_nnz,
                      int32_t *pattern_changed);
/* SparseArrays.nnz of the device CSC (does not flush) */
int32_t esp_nnz(const esp_handle *h, int64_t *nnz);
/* D2H into caller arrays: colptr (n+1), rowval (nnz), nzval (nnz); Julia layout.  Large destinations (> 8 MiB) get
 * madvise(MADV_HUGEPAGE) on their 2 MiB-aligned interior before the first byte lands -- a hint that changes no contents: a
 * vector the caller has just allocated is untouched memory, and its first touch otherwise costs ten times the copy */
int32_t esp_get_csc(esp_handle *h, int64_t *colptr, int64_t *rowval, double *nzval);
/* The same transfers for Int32 index arrays (SparseMatrixCSC{Float64,Int32}: extendable.jl:10-25 is generic in Ti).  The
 * device CSC stays Int64; colptr / rowval are narrowed (widened) on the device, so they cross PCIe as 4 bytes.
 * esp_get_csc_i32: ESP_ERR_UNSUPPORTED when m or nnz + 1 exceeds 2^31 - 1 (Julia's own Int32 CSC could not hold it). */
int32_t esp_set_csc_i32(esp_handle *h, const int32_t *colptr, const int32_t *rowval,
                        const double *nzval, int64_t nnz);
int32_t esp_get_csc_i32(esp_handle *h, int32_t *colptr, int32_t *rowval, double *nzval);
/* D2H of nzval only (pattern unchanged since the caller's last esp_get_csc) */
int32_t esp_get_nzval(esp_handle *h, double *nzval);
/* H2D of nzval only: the attached CSC keeps its pattern (the one of the caller's last esp_set_csc / esp_get_csc) and takes
 * these nnz values -- what a plug-in whose CSC stays on the device between flushes uploads when only nonzeros(A) can have
 * been edited on the host (the Generic wrappers edit cscmatrix.nzval in place: genericextendablesparsematrixcsc.jl:44-54) */
int32_t esp_set_nzval(esp_handle *h, const double *nzval);
/* device pointers of the resident CSC (Int64 1-based values), for device consumers; valid until the next esp_flush /
 * esp_reset / esp_set_csc of the handle (a flush that changes the pattern rotates all three arrays) */
int32_t esp_csc_device(esp_handle *h, const int64_t **d_colptr, const int64_t **d_rowval,
                       const double **d_nzval);
/* reset!(ext) (extendable.jl:269-272): empty CSC, empty buffer */
int32_t esp_reset(esp_handle *h);
/* drop the buffer only (a fresh T_ext(m,n), genericextendablesparsematrixcsc.jl:34) */
int32_t esp_clear_pending(esp_handle *h);
/* fdrand!'s zero!(A): nonzeros(A) .= 0 (sprand.jl:82) on the device CSC */
int32_t esp_zero_values(esp_handle *h);
/* dropzeros!(A) on the device CSC (test/test_updates.jl:19,23) */
int32_t esp_dropzeros(esp_handle *h, int64_t *new_nnz);
/* findindex(csc,i,j)+nzval[k] (sparsematrixcsc.jl:7-23) on the device CSC; found=0 if absent */
int32_t esp_getindex(esp_handle *h, int64_t i, int64_t j, double *value, int32_t *found);
/* getindex(buffer,i,j) (sparsematrixlnk.jl:151-171): the value the PENDING entries alone give (i,j) -- their ordered
 * fold -- 0 and found=0 if none creates an entry.  Slow by design (one pass over the pending keys per call); what
 * GenericExtendableSparseMatrixCSC's getindex falls back to for positions not yet in the CSC (genericext...:60-69).
 * More than 2048 pending updates of one position -> ESP_ERR_UNSUPPORTED (flush first). */
int32_t esp_pending_getindex(esp_handle *h, int64_t i, int64_t j, double *value, int32_t *found);
/* stand-in for phash(csc) (sparsematrixcsc.jl:74): a 64-bit function of colptr/rowval only */
int32_t esp_pattern_hash(esp_handle *h, uint64_t *hash);

/* mul!(r, A, x) on the device CSC (abstractextendablesparsematrixcsc.jl:179-181; the coloured loop of
 * genericmtextendablesparsematrixcsc.jl:124-143 visits the columns in the same order): r[i] = the sum of
 * A[i,j]*x[j] over the entries of row i in increasing column order, products and sums rounded
 * separately -- bit-identical to the reference's column loop (no atomics; a row-wise index of the CSC is
 * built on first use after a pattern change).  x has n, r has m elements; on_device != 0: both are
 * device pointers (a consumer that never leaves the GPU).  Pending entries -> ESP_ERR_STATE: flush first. */
int32_t esp_mul(esp_handle *h, const double *x, double *r, int32_t on_device);

/* mark_dirichlet(A; penalty) / eliminate_dirichlet!(A, marker) (sparsematrixcsc.jl:97-140) on the device CSC
 * of a square matrix: marker[i] = 1 iff A[i,i] >= penalty; A[:,i] = 0, A[i,:] = 0, A[i,i] = 1 for marked i.
 * marker has n bytes (Julia Vector{Bool}); on_device != 0: a device pointer.  Values only, the pattern stays. */
int32_t esp_mark_dirichlet(esp_handle *h, double penalty, uint8_t *marker, int32_t on_device);
int32_t esp_eliminate_dirichlet(esp_handle *h, const uint8_t *marker, int32_t on_device);

/* set-up of the point preconditioners on the device CSC of a square matrix (pending entries -> ESP_ERR_STATE).
 * esp_jacobi_setup = jacobi(A) (src/factorizations/jacobi.jl:5-12): invdiag[i] = 1 / A[i,i] (Inf where the diagonal
 * is not stored: getindex gives zero).  esp_ilu0_setup = ilu0(A) (src/factorizations/ilu0.jl:8-41): idiag[j] = 1-based
 * index of column j's diagonal entry in rowval/nzval, xdiag[j] = what the reference's loop leaves: 1 / nzval[idiag[j]]
 * (its `xdiag[i] -= ...` updates of rows i > j are overwritten by iteration i); a column without a stored diagonal
 * -> ESP_ERR_INVALID.  n entries each; on_device != 0: device pointers. */
int32_t esp_jacobi_setup(esp_handle *h, double *invdiag, int32_t on_device);
int32_t esp_ilu0_setup(esp_handle *h, double *xdiag, int64_t *idiag, int32_t on_device);

/* ---- column-range shards (multi-GPU, one process per GPU) ------------------------
 * owner(col) = floor((col-1)*nshards/n).  esp_shard_counts: pending entries per owner.
 * esp_shard_export: stable partition of the pending entries by owner into the caller's
 * device buffers (send buffer of the all-to-all); offsets has nshards+1 entries.       */
int32_t esp_shard_counts(esp_handle *h, int32_t nshards, int64_t *counts);
int32_t esp_shard_export(esp_handle *h, int32_t nshards, uint64_t *d_keys, double *d_vals,
                         int64_t *offsets);
/* In-place form of the exchange: what this rank owns itself is not copied twice.  After the counts
 * were exchanged the caller knows how many entries arrive from lower / higher ranks.  The pending
 * entries are partitioned by owner; the own chunk lands at its final position recv_lower, the other
 * chunks (compacted, owner order; send_offsets has nshards+1 entries, the own chunk counts 0) in a
 * send region returned as device pointers that stay valid until the next append or flush.  The
 * received chunks are then dropped in with esp_shard_exchange_place (lower ranks at position 0,
 * higher ranks behind the own chunk). */
int32_t esp_shard_exchange_begin(esp_handle *h, int32_t nshards, int32_t self, int64_t recv_lower,
                                 int64_t recv_higher, uint64_t **d_send_keys, double **d_send_vals,
                                 int64_t *send_offsets);
int32_t esp_shard_exchange_place(esp_handle *h, int64_t position, const uint64_t *d_keys,
                                 const double *d_vals, int64_t count);

/* Partitioned exchange (the fast path for streams an assembly loop emits; the calls above remain the
 * general one).  ONE stable pass partitions the pending entries by (owner, digit inside the owner's
 * column range) -- the owner split and the first partition pass of the local flush at once.
 *   esp_shard_partition: entries_per_shard = (global number of pending entries) / nshards, the SAME
 *     value on every rank (it fixes the digit width, which all ranks must agree on).  *ok = 0: not
 *     applicable (stream not pre-sorted, tiny problem, > 64 shards) -- use esp_shard_exchange_begin;
 *     the pending entries are intact (possibly permuted in a way that keeps the order per entry).
 *     *ok = 1: owner r's entries are d_keys/d_vals[entry_offsets[r] .. entry_offsets[r+1]) and its
 *     per-digit counts d_counts[r*digits .. (r+1)*digits) (device pointers, valid until the next
 *     append or flush).  Send every OTHER owner its range and its counts.  entry_offsets and d_counts are
 *     final on return; the kernel that moves the entries into d_keys/d_vals may still be running on the
 *     handle's stream (the callers' consensus round fits beside it): esp_synchronize(h) before reading them.
 *   esp_shard_assemble: device pointers of the blocks received from every source rank (index = source;
 *     the own index is ignored; recv_entries[q] = entries in block q, counts blocks hold `digits`
 *     values).  The buffers must stay alive until esp_flush returns: the bucket kernel reads a segment
 *     as the concatenation of one piece per source (rank order), nothing is copied.  *ok = 0: a merged
 *     segment is too long for the bucket kernel; the entries were copied into a plain pending buffer
 *     (rank order) instead and esp_flush works as usual.  *ok = 1: until that esp_flush no append is accepted
 *     (ESP_ERR_STATE): the pending entries live in the caller's receive buffers.
 * esp_set_column_window(own column range) must be in force, as for the other exchange. */
int32_t esp_shard_partition(esp_handle *h, int32_t nshards, int32_t self, int64_t entries_per_shard,
                            int32_t *ok, uint64_t **d_keys, double **d_vals, int64_t **d_counts,
                            int64_t *entry_offsets /* nshards+1 */, int64_t *digits_per_shard);
/* esp_shard_plan announces the next esp_shard_partition(nshards, self, entries_per_shard): a device-side producer
 * (esp_generate_*) that finds the buffer empty then writes every entry straight to its (owner, digit) bucket -- the
 * append is the partition, as on one GPU -- and that esp_shard_partition call, given exactly these arguments, has nothing
 * left to move.  entries_per_shard < 0: no announcement.  esp_group_flush does this for the flush after it. */
int32_t esp_shard_plan(esp_handle *h, int32_t nshards, int32_t self, int64_t entries_per_shard);
/* 1: the last esp_shard_partition moved the entries in a pass of its own, 2: the producer had partitioned them, 0: neither */
int32_t esp_debug_last_shard_source(const esp_handle *h, int32_t *kind);
int32_t esp_shard_assemble(esp_handle *h, const uint64_t *const *d_recv_keys,
                           const double *const *d_recv_vals, const int64_t *const *d_recv_counts,
                           const int64_t *recv_entries /* nshards */, int32_t *ok);

/* ---- esp_group: the sharded flush as ONE call per rank (one process per GPU) --------------------------
 * What GenericMTExtendableSparseMatrixCSC does with threads (one buffer per tid, flush! = Base.sum(xmatrices, csc):
 * src/matrix/genericmtextendablesparsematrixcsc.jl:45-51,87-99) across the GPUs of a node: every rank appends whatever
 * its part of the assembly loop produces to ITS handle (any column), esp_group_flush routes every pending entry to the
 * rank that owns its column and runs the local flush.  It drives the esp_shard_* calls above; the transport is RCCL
 * (grouped ncclSend/ncclRecv on the handle's stream, loaded with dlopen: no link-time dependency) or a callback table
 * of the host (MPI, a test harness).  Every esp_group_flush / _nnz / _get_csc is COLLECTIVE: all ranks call it.
 *
 *   rank 0:  esp_group_unique_id(id)  ->  the host broadcasts the 128 bytes (MPI_Bcast, a socket, Distributed.jl)
 *   all:     esp_create(m, n, device, hint, &h); esp_group_create(h, nranks, rank, id, &g)
 *   loop:    appends on h (esp_commit / esp_append_* / esp_generate_*);  esp_group_flush(g, mode, &local_nnz, &changed)
 *   result:  esp_group_nnz (global nnz, this rank's offset), esp_group_get_csc (own column range, global colptr values),
 *            or the device CSC of the handle (esp_csc_device) for consumers that stay on the GPU.                    */
typedef struct esp_group esp_group;
typedef struct {
    void *ctx;
    /* all-gather of `count` int64 per rank through HOST memory; recv holds nranks * count values, rank-major */
    int32_t (*allgather_i64)(void *ctx, const int64_t *send, int32_t count, int64_t *recv);
    /* all-to-all-v on DEVICE memory of the handle's device: for every peer q (the own index is ignored) send
     * send_bytes[q] bytes from send[q] and receive recv_bytes[q] bytes into recv[q]; must be complete, or ordered on
     * hip_stream (the handle's hipStream_t) in front of whatever is enqueued there next, when it returns */
    int32_t (*alltoallv_dev)(void *ctx, const void *const *send, const int64_t *send_bytes, void *const *recv,
                             const int64_t *recv_bytes, void *hip_stream);
} esp_comm_t;
int32_t esp_group_unique_id(uint8_t *id128 /* 128 bytes: an ncclUniqueId */);
/* the handle must be empty; its column window becomes the rank's own column range */
int32_t esp_group_create(esp_handle *h, int32_t nranks, int32_t rank, const uint8_t *id128, esp_group **out);
int32_t esp_group_create_comm(esp_handle *h, int32_t nranks, int32_t rank, const esp_comm_t *comm, esp_group **out);
int32_t esp_group_destroy(esp_group *g);   /* (the handle stays the caller's) */
int32_t esp_group_handle(esp_group *g, esp_handle **out);
const char *esp_group_last_error(const esp_group *g);
/* the rank's own columns, 1-based inclusive: owner(col) = floor((col-1)*nranks/n) */
int32_t esp_group_column_range(const esp_group *g, int64_t *col_lo, int64_t *col_hi);
int32_t esp_group_flush(esp_group *g, int32_t mode, int64_t *local_nnz, int32_t *pattern_changed);
int32_t esp_group_nnz(esp_group *g, int64_t *global_nnz, int64_t *nnz_before_me);
/* colptr_own: col_hi - col_lo + 2 entries = the GLOBAL colptr[col_lo .. col_hi + 1]; rowval, nzval: local_nnz entries */
int32_t esp_group_get_csc(esp_group *g, int64_t *colptr_own, int64_t *rowval, double *nzval);
/* kind: 1 = partitioned exchange (pre-sorted streams), 2 = in-place exchange; entries sent to other ranks */
int32_t esp_group_last_exchange(const esp_group *g, int32_t *kind, int64_t *sent_off_rank);
/* Test hook for one-GPU boxes (a single-rank group made with esp_group_create, i.e. over the library's own RCCL
 * transport): on = 1 -- every esp_group_flush sends the rank's own partitioned ranges to ITSELF through the all-to-all-v
 * (grouped ncclSend / ncclRecv, 1 GiB rounds, on the handle's stream), wipes them and restores them from what arrived,
 * and the flush's small agreements run as a real ncclAllGather on the second stream; on = 0 -- off; on < 0 -- query only.
 * bytes_last_flush: what travelled through RCCL in the last flush.  Results never change. */
int32_t esp_debug_group_loopback(esp_group *g, int32_t on, int64_t *bytes_last_flush);

/* promise: every pending entry of the following flushes has its column in [col_lo, col_hi]
 * (1-based); the partition then works on that window only.  Violations -> ESP_ERR_STATE from esp_flush: the
 * stored pattern is untouched and the batch stays pending (esp_clear_pending / esp_reset), but updates of the
 * batch that hit stored positions may already have been applied to their values. */
int32_t esp_set_column_window(esp_handle *h, int64_t col_lo, int64_t col_hi);

/* ---- measurement ---------------------------------------------------------------- */
/* on = 1: HIP events around the big kernels (append, hist, scatter, local, fold, colptr, merge, copy);
 * on = 2: also around the small launches of the "scan" stage (costs about 3 % of a 256^3 step);
 * on = 3: around the bucket kernel (general path: the fold kernel) only -- every bracketed kernel costs two event
 * records of about 6 us on the stream; 0: off */
int32_t esp_timing_enable(esp_handle *h, int32_t on);
int32_t esp_timing(esp_handle *h, esp_timing_t *out, int32_t clear);
/* Test hooks: esp_debug_force_path(h, path) pins ONE implementation where the library has two (results never change,
 * timings do); path 0 = automatic.  The numbers are part of the ABI (the tests and tools pass them). */
typedef enum {
    ESP_PATH_AUTO = 0,
    ESP_PATH_GENERAL = 2,            /* the general path: global LSD sort + global fold (no bucket kernel)                  */
    ESP_PATH_RADIX_TAIL_ONLY = 3,    /* bucket kernel with its radix tier only (no column tiers, no small variant)          */
    ESP_PATH_MANY_LAUNCHES = 4,      /* bucket kernel issued in launches of 64 workgroups (carry-over of the look-back state) */
    ESP_PATH_NO_RUN_PARTITION = 5,   /* never the run-based single-pass partition: 8/9-bit passes only; producers append in
                                        stream order                                                                        */
    ESP_PATH_SHARD_NOT_APPLICABLE = 11, /* esp_shard_partition reports "not applicable" (consensus fall-back of the exchange) */
    ESP_PATH_RUN_LIST_BY_RADIX = 12, /* the run-based partition orders its run list with radix passes (several small launches
                                        and a host round trip) instead of the one ranking kernel                            */
    ESP_PATH_COLPTR_BY_SCAN = 13,    /* fresh matrix: the bucket kernel marks column ends and a scan over all columns builds
                                        colptr (instead of every segment writing the colptr of its own columns)             */
    ESP_PATH_PACKED_KEYS = 14,       /* packed 8-byte keys for the bucket kernel always (never 4-byte keys)                  */
    ESP_PATH_GENERIC_FOLD = 15,      /* 4-byte keys but the generic fold (no UPDATE-only / one-kind variants)                */
    ESP_PATH_PRODUCER_STREAM_ORDER = 16, /* device-side producers and bulk appends always write in stream order (never the
                                        producer-side partition); the flush partitions                                      */
    ESP_PATH_MERGE_PATH_JOIN = 17,   /* the join with a stored CSC as a merge-path over a per-entry column array (a second
                                        implementation of the column-tiled join)                                            */
    ESP_PATH_NO_SMALL_VARIANT = 18,  /* never the small variant of the bucket kernel (see esp_debug_last_local_small)        */
    ESP_PATH_NO_BATCH_TAIL = 19,     /* an append behind a producer's bucket-ordered batch turns it back into packed keys
                                        (no "batch + tail" flush)                                                           */
    ESP_PATH_BATCH_TAIL_ONE_FLUSH = 22, /* a batch + tail over a stored pattern is ONE flush over two pieces (as on a fresh
                                        matrix) instead of two flushes                                                      */
    ESP_PATH_EIGHT_BIT_PASSES = 23,  /* partition passes of at most 8 bits (no 9-bit digits where they would save a pass)    */
    ESP_PATH_NO_GROUP_TIER = 24,     /* bucket kernel: column runs of more than 24 entries through the radix tier, never the
                                        group tier (2 .. 16 lanes per column)                                               */
    ESP_PATH_NO_ITEM_PARTITION = 25, /* esp_generate_fem in a shuffled order appends in stream order (no item partition)     */
    ESP_PATH_NO_BIG_VARIANT = 26,    /* never the bucket-kernel variant with the 24-input register tier                      */
    ESP_PATH_NO_APPEND_PARTITION = 27, /* esp_append_device / esp_commit of one kind on an empty buffer pack in stream order (the
                                        append is not the partition; esp_append_host always arrives in stream order)       */
    ESP_PATH_TWO_WORD_ITEMS = 28,    /* the item partition of esp_generate_fem moves 16-byte records (key | cell and vertex)
                                        even where the cell's number fits into the key                                      */
    ESP_PATH_TAIL_TO_FRONT = 29,     /* the entries behind a batch that was flushed by itself are copied to the front of
                                        the buffer before their partition (instead of being read where they lie)            */
    ESP_PATH_NO_GROUP3 = 30,         /* never the group-tier kernel with three workgroups per CU (group3_k)                  */
    ESP_PATH_LOCAL_BITS = 32,        /* item partitions (esp_generate_fem in a shuffled order, esp_append_elements): the passes
                                        stop up to three bits early and the expansion orders every segment of up to 4096 items
                                        by the last bits itself (segexpand.hpp) -- one pass less, a slower expansion: measured
                                        slower at config 4's sizes, so not the default                                      */
    ESP_PATH_NO_WIDE_GROUP3 = 33,    /* never the wide form of group3_k (rows of a segment spread over more than 2^18: full rows
                                        in LDS, every column run sorted twice); such segments go to local_k's kernels         */
    ESP_PATH_NO_HITS_KERNEL = 34,    /* never group3_k's re-assembly form (additions over a stored pattern the same mesh built:
                                        every (col,row) of a column run is the stored entry of its rank, sums to a second value
                                        array, all-or-nothing); such flushes take local_k's group-tier kernels               */
    ESP_PATH_LATE_TOTAL = 36,        /* group3_k publishes a segment's total for the look-back after the fold (instead of right
                                        after the sort, which a segment of RAWUPDATEs allows)                                */
    ESP_PATH_NO_CELL_RECORDS = 37,   /* esp_append_elements with cells of 3 / 4 nodes: no 64-byte cell records, the expansion gathers
                                        rows and diagonal terms from the caller's arrays (as it does for other cell sizes)     */
    ESP_PATH_HOST_KEYS8 = 38,        /* esp_append_host of one kind: packed eight-byte keys over PCIe (never six-byte keys)      */
    ESP_PATH_NO_REBUILD = 40,        /* the entries behind a re-assembly's batch never REBUILD the matrix (a fresh flush whose segments start with
                                        the stored entries of their columns as a first piece: no look-ups, no join): always the bucket kernel
                                        against the stored columns + the column-tiled join                                      */
    ESP_PATH_NO_LAZY_ITEMS = 39,     /* item partitions (esp_generate_fem in a shuffled order, esp_append_elements) always run their
                                        expansion at append time: the updates are stored bucket by bucket and the flush's bucket
                                        kernel reads them -- never the fused form that forms them from the sorted item records  */
    ESP_PATH_NO_FINE_PARTITION = 41, /* more than 32 key bits below the planned prefix (a stencil above 256^3): never the FINE partition
                                        (up to 4 more prefix bits, so that 4-byte keys serve, with the bucket kernel taking 2 .. 16
                                        neighbouring buckets as one segment): packed 8-byte keys as before round 6 -- for a
                                        shard's producer too (its own range: 2^fb buckets per digit of the exchange plan)      */
    ESP_PATH_NO_PLAN_REUSE = 31      /* esp_append_device / esp_commit of one kind on an empty buffer always count their columns
                                        (never the run lists of the previous, identical-looking batch)                      */
} esp_debug_path;
/* last_path reports which pipeline the last flush took (1 = LDS bucket path, 2 = general).
 * Environment: the product library reads ESP_HOST_THREADS (host threads the transfers from / to pageable host arrays --
 * esp_append_host, esp_set_csc / esp_set_nzval, esp_get_csc / esp_get_nzval, esp_append_elements_host -- may use beside the
 * caller's: they go through two pinned bounce buffers, the host copy of one chunk overlapping the PCIe transfer of the other), ESP_RCCL_LIB (path of the RCCL library esp_group_create loads) and, as test hooks, ESP_DEBUG_FORCE_PATH (the
 * path every new handle starts with) and ESP_DEBUG_FAIL_COMM_INIT (esp_group_create fails as with a broken fabric) --
 * nothing else: the switches of the measurement tools exist in a -DESP_EXPERIMENTS build only (csrc/common.hpp). */
int32_t esp_debug_force_path(esp_handle *h, int32_t path);
/* test hook: plan the partition as if the bucket kernel took segments of `cap` entries (many prefix bits -- the 9-bit passes,
 * several rounds -- at sizes a CPU oracle can follow); 0: off */
int32_t esp_debug_plan_cap(esp_handle *h, double cap);
/* test hook: the NEXT flush that reaches its bucket stage reports ESP_ERR_STATE there instead of running it (once) -- the batch
 * must stay pending, whatever the partition left in the buffers (4-byte keys in both pairs ...), and the next flush must finish it */
int32_t esp_debug_fail_next_bucket_stage(esp_handle *h);
int32_t esp_debug_last_path(const esp_handle *h, int32_t *path);
/* how the last run-based partition turned its run lists into offsets: 1 = one ranking kernel over per-digit run
 * lists, 2 = radix-ordered run list, 3 = ranking kernel given up (a digit with more runs than its list holds),
 * radix-ordered run list used */
int32_t esp_debug_last_run_order(const esp_handle *h, int32_t *kind);
/* 1 when the bucket kernel of the last flush wrote colptr itself (fresh matrix, full key window, segments = whole
 * blocks of at most 2048 columns), 0 when column-end marks + a scan over the columns did */
int32_t esp_debug_last_colptr_direct(const esp_handle *h, int32_t *direct);
/* bytes per key the bucket kernel of the last flush read: 4 when every pending entry had been appended with one
 * known kind, at most 32 key bits were left below the partition prefix and the run-based partition served the
 * flush (it then writes only those bits), else 8 (packed keys) */
int32_t esp_debug_last_key_bytes(const esp_handle *h, int32_t *bytes);
/* 1 when the register tiers of the last flush's bucket kernel ran the UPDATE-only fold (every entry known to be an
 * updateindex! call: the batch's bookkeeping, for a shard also a device check of the received blocks) */
int32_t esp_debug_last_fold_update(const esp_handle *h, int32_t *on);
/* 1 when the bucket kernel of the last flush was its small variant: segments of at most 3072 entries over at most 256
 * columns, no radix tier, 51 KiB of LDS = three workgroups per CU instead of two (column runs longer than its register
 * tiers take go through a slow tier and send the handle's next flushes to the regular kernel); esp_debug_force_path(18):
 * never.  2 when it was the group tier's kernel with three workgroups per CU (group3_k: long column runs on a fresh matrix,
 * 4-byte keys; esp_debug_force_path(30): never); 3 when it was that kernel's WIDE form (rows of a segment spread over more than
 * 2^18 -- a mesh numbered without locality: every run sorted twice; esp_debug_force_path(33): never); 4 / 5 when it was that
 * kernel's re-assembly form over a stored pattern (plain / wide; esp_debug_force_path(34): never) */
int32_t esp_debug_last_local_small(const esp_handle *h, int32_t *small);
/* 1 when the bucket kernel of the last flush formed its updates from the sorted ITEM records of an item partition
 * (esp_generate_fem in a shuffled order, esp_append_elements on an empty buffer of a fresh matrix): the expansion -- every
 * update stored once at its bucket position, read again by the bucket kernel -- never ran (csrc/group3_items.hpp);
 * esp_debug_force_path(39): never.  2 (on the destination of esp_flush_sum): the folds of all buffers ran as ONE launch over their
 * item records and the combine flush read the folded records as pieces (every buffer held an element batch with the same plan) */
int32_t esp_debug_last_lazy_items(const esp_handle *h, int32_t *on);
/* how many neighbouring segments of the folds' plan the combine flush of the last esp_flush_sum of that kind (last_lazy_items 2)
 * joined into one (1, 2, 4 or 8: as many as keep the longest joined segment within the bucket kernel's capacity); 0 otherwise */
int32_t esp_debug_last_sum_join(const esp_handle *h, int32_t *segments);
/* host wall-clock split of the destination's last esp_flush_sum, in ms: the buffers' folds (one launch over their item records, or
 * every buffer's own flush) | everything behind them (gather, combine flush, its completion) -- what a bench line reports beside
 * esp_debug_last_lazy_items / _sum_join so that a slow run can be told from a run that took the other path */
int32_t esp_debug_last_sum_ms(const esp_handle *h, double *folds_ms, double *combine_ms);
/* 1 when the destination's last esp_flush_sum folded its buffers -- not element batches with one plan: per-entry calls, mixed kinds,
 * anything -- in ONE flush of a scratch matrix with p n columns (buffer k in the columns [k n, (k + 1) n): no column is shared, every
 * buffer is folded by itself) instead of one flush per buffer; 0: one by one (a single buffer, a window or test hook on a buffer) */
int32_t esp_debug_last_sum_batched(const esp_handle *h, int32_t *batched);
/* the smallest / largest number of prefix bits the buffers of the destination's last joint esp_flush_sum (last_lazy_items 2) had planned
 * for their item partitions: every handle plans from its own history, the joint path takes the coarsest plan (test hook) */
int32_t esp_debug_last_sum_plan_bits(const esp_handle *h, int32_t *pb_min, int32_t *pb_max);
/* 1 when the last flush REBUILT the matrix for the entries behind a re-assembly's batch (a fresh flush whose segments start
 * with the stored entries of their columns: no look-ups against the stored columns, no join); esp_debug_force_path(40): never */
int32_t esp_debug_last_rebuild(const esp_handle *h, int32_t *on);
/* 1 when the last append-is-the-partition of caller-supplied triplets (esp_append_device / esp_commit of one kind on an
 * empty buffer) used the run lists of the previous assembly instead of counting its columns again: a batch of the same
 * length and kind is scattered straight away, every tile checked against its run list by the scatter kernel (a stream
 * that turns out different is partitioned the full way; results never change) */
/* ... likewise 1 when the last esp_generate_fdrand[_range] repeated the handle's previous call (same grid, node range and kind on
 * the same empty buffer: a time loop of reset! / fdrand! / flush!) and went straight to its PART launch: the run lists, run
 * offsets and bucket starts are a function of the grid and the plan, not of seed or values (esp_debug_force_path(31): never) */
int32_t esp_debug_last_plan_reused(const esp_handle *h, int32_t *reused);
/* which partition the last flush used: 1 = run-based single pass (pre-sorted stream), 2 = 8-bit passes,
 * 4 = none: the producer wrote every entry straight to its bucket (esp_generate_* on an empty buffer: a COUNT launch
 *     of the producer, then its stores go to `bucket start + stable rank`; the flush starts at the bucket kernel),
 * 5 = such a batch with entries appended behind it: the run-based pass over those entries only, the bucket kernel
 *     reads every segment as two pieces (batch, tail),
 * 6 = such a batch + tail over a stored pattern: the batch flushed by itself (as 4), then the tail as a flush of its own,
 * 7 = none: the segments came assembled from esp_shard_assemble,
 * 8 = as 6, and the tail had been partitioned as it was appended (one kind, a pre-sorted stream): both flushes start at the
 *     bucket kernel */
int32_t esp_debug_last_partition(const esp_handle *h, int32_t *kind);

#ifdef __cplusplus
}
#endif
#endif
