/*
 * esparse_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE)
 *
 * A plain-C restatement of the sparse-assembly hot path of ExtendableSparse.jl
 * (reference @ v1.5.1).  Only tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py may load this library; the product
 * (libesparse_hip.so) never links or calls it.
 *
 * PARITY PINNING.  The reference is Julia and no `julia` binary exists in the
 * build container or on the GPU box, and the reference stores no golden
 * vectors.  The oracle is therefore pinned against
 *   (1) the known-answer tests the reference's own test-suite holds for this
 *       path (test/test_updates.jl:12-24 nnz trace, test/test_fdrand.jl:22-53
 *       analytic rand=()->1 stencil, test/test_operations.jl:8-13
 *       csc+LNK(csc)==2csc, test/test_constructors.jl:26-31 LNK<->CSC round
 *       trip, test/test_assembly.jl:19-32 in-order accumulation `==`), and
 *   (2) an independent NumPy/SciPy restatement (tests/test_oracle.py).
 * It cannot be checked against outputs of the reference run here: that part
 * of parity is UNPINNED and says so in DESIGN.md.
 *
 * All indices are 1-based Int64, as Julia passes them.
 */
#ifndef ESPARSE_ORACLE_H
#define ESPARSE_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef int64_t i64;

/* op enum shared with the product ABI (include/esparse_hip.h) */
enum { ORC_OP_ADD = 0, ORC_OP_SUB = 1 };
/* entry kinds of an update stream */
enum { ORC_KIND_SET = 0, ORC_KIND_UPDATE = 1, ORC_KIND_RAWUPDATE = 2, ORC_KIND_PLUSEQ = 3 };
#define ORC_ERR_BOUNDS (-1)
#define ORC_ERR_MT_NEW_SETINDEX (-2)
#define ORC_ERR_MT_GETINDEX_PENDING (-3)

typedef struct orc_lnk orc_lnk;
typedef struct orc_csc orc_csc;
typedef struct orc_ext orc_ext;
typedef struct orc_mt orc_mt;

/* ---- SparseMatrixCSC (Julia stdlib container; helpers sparsematrixcsc.jl) */
orc_csc *orc_csc_new(i64 m, i64 n);
orc_csc *orc_csc_from_arrays(i64 m, i64 n, const i64 *colptr, const i64 *rowval,
                             const double *nzval);
void orc_csc_free(orc_csc *);
i64 orc_csc_m(const orc_csc *);
i64 orc_csc_n(const orc_csc *);
i64 orc_csc_nnz(const orc_csc *);
void orc_csc_copy_out(const orc_csc *, i64 *colptr, i64 *rowval, double *nzval);
i64 orc_csc_findindex(const orc_csc *, i64 i, i64 j);
int orc_csc_pattern_equal(const orc_csc *, const orc_csc *);
uint64_t orc_csc_pattern_hash(const orc_csc *);
i64 orc_csc_dropzeros(orc_csc *);
/* set-up of the point preconditioners: src/factorizations/jacobi.jl:5-12, ilu0.jl:8-41 */
void orc_csc_jacobi(const orc_csc *, double *invdiag);
i64 orc_csc_ilu0(const orc_csc *, double *xdiag, i64 *idiag);

/* ---- SparseMatrixLNK (sparsematrixlnk.jl) */
orc_lnk *orc_lnk_new(i64 m, i64 n);
orc_lnk *orc_lnk_from_csc(const orc_csc *);
void orc_lnk_free(orc_lnk *);
i64 orc_lnk_nnz(const orc_lnk *);
i64 orc_lnk_nentries(const orc_lnk *);
int orc_lnk_setindex(orc_lnk *, double v, i64 i, i64 j);
int orc_lnk_updateindex(orc_lnk *, int op, double v, i64 i, i64 j);
int orc_lnk_rawupdateindex(orc_lnk *, int op, double v, i64 i, i64 j);
int orc_lnk_getindex(const orc_lnk *, i64 i, i64 j, double *out);
orc_csc *orc_lnk_plus_csc(const orc_lnk *, const orc_csc *);

/* ---- ExtendableSparseMatrixCSC (extendable.jl) and the Generic wrapper */
orc_ext *orc_ext_new(i64 m, i64 n);
orc_ext *orc_ext_from_csc(const orc_csc *);
void orc_ext_free(orc_ext *);
int orc_ext_setindex(orc_ext *, double v, i64 i, i64 j);
int orc_ext_updateindex(orc_ext *, int op, double v, i64 i, i64 j);
int orc_ext_rawupdateindex(orc_ext *, int op, double v, i64 i, i64 j);
int orc_ext_getindex(const orc_ext *, i64 i, i64 j, double *out);
int orc_ext_flush(orc_ext *);           /* returns 1 if the CSC was rebuilt */
const orc_csc *orc_ext_csc(orc_ext *);  /* sparse(ext): flush, then the CSC */
i64 orc_ext_nnz(orc_ext *);             /* flushes first (abstractext..:80) */
i64 orc_ext_pending(const orc_ext *);   /* nnz(lnkmatrix) or 0 */
uint64_t orc_ext_phash(const orc_ext *);
i64 orc_ext_flush_count(const orc_ext *);
void orc_ext_reset(orc_ext *);
void orc_ext_zero_values(orc_ext *);    /* fdrand!'s zero! (sprand.jl:82) */
i64 orc_ext_dropzeros(orc_ext *);
/* apply a whole stream; returns 0 or the first error and its position */
/* mark_dirichlet(A;penalty) / eliminate_dirichlet!(A,marker) (sparsematrixcsc.jl:94-144) */
void orc_csc_mark_dirichlet(const orc_csc *, double penalty, uint8_t *marker);
void orc_csc_eliminate_dirichlet(orc_csc *, const uint8_t *marker);
/* mul!(r, A, x): r .= 0, column loop r[rows[i]] += vals[i]*x[col] (genericmtextendablesparsematrixcsc.jl:124-143) */
void orc_csc_mul(const orc_csc *, const double *x, double *r);
/* sparse(I,J,V,m,n,+) of the COO constructors (extendable.jl:92-104); NULL on an index outside m x n */
orc_csc *orc_sparse_coo(i64 m, i64 n, i64 count, const i64 *I, const i64 *J, const double *V);
int orc_ext_apply(orc_ext *, i64 count, const uint8_t *kinds, const i64 *I, const i64 *J,
                  const double *V, i64 *errpos);

/* ---- GenericMTExtendableSparseMatrixCSC + Base.sum(Vector{DILNKC},csc) */
orc_mt *orc_mt_new(i64 m, i64 n, i64 nparts);
void orc_mt_free(orc_mt *);
int orc_mt_setindex(orc_mt *, double v, i64 i, i64 j);
int orc_mt_updateindex(orc_mt *, int op, double v, i64 i, i64 j, i64 tid);
int orc_mt_rawupdateindex(orc_mt *, int op, double v, i64 i, i64 j, i64 tid);
int orc_mt_apply(orc_mt *, i64 count, const uint8_t *kinds, const i64 *I, const i64 *J, const double *V, i64 tid);
int orc_mt_getindex(const orc_mt *, i64 i, i64 j, double *out);
i64 orc_mt_nnznew(const orc_mt *);
int orc_mt_flush(orc_mt *);
const orc_csc *orc_mt_csc(orc_mt *);
void orc_mt_reset(orc_mt *);

/* ---- update streams */
double orc_uniform(uint64_t seed, uint64_t counter);
i64 orc_fdrand_count(i64 nx, i64 ny, i64 nz);
i64 orc_fdrand_nnz(i64 nx, i64 ny, i64 nz);
/* rand_mode: 0 -> ()->1 ; 1 -> 0.1+u (fdrand default) ; 2 -> u (fdrand! default) */
void orc_fdrand_stream(i64 nx, i64 ny, i64 nz, int rand_mode, uint64_t seed, i64 *I, i64 *J,
                       double *V);
/* fdrand!(A,nx,ny,nz;update,rand): zero!, hot loop, flush.  style = ORC_KIND_* */
int orc_fdrand_ext(orc_ext *, i64 nx, i64 ny, i64 nz, int rand_mode, uint64_t seed, int style);
/* CPU baseline: same thing with wall-clock split; returns nnz */
i64 orc_bench_fdrand(i64 nx, i64 ny, i64 nz, int style, double *t_insert_s, double *t_flush_s);
/* np threads, one buffer per thread (node slabs), COO merge: the shape of GenericMTExtendableSparseMatrixCSC */
i64 orc_bench_fdrand_mt(i64 nx, i64 ny, i64 nz, i64 np, double *t_insert_s, double *t_merge_s);

/* P1 FEM on a Kuhn-triangulated tensor grid (update pattern of test/femtools.jl:45-72) */
i64 orc_fem_ncells(int dim, i64 npd);
i64 orc_fem_nnodes(int dim, i64 npd);
i64 orc_fem_count(int dim, i64 npd);
uint64_t orc_fem_cell_at(i64 pos, i64 ncells, uint64_t seed, int order_mode);
void orc_fem_cell_nodes(int dim, i64 npd, i64 cell, i64 *nodes /* dim+1 */);
void orc_fem_stream(int dim, i64 npd, uint64_t seed, int order_mode, i64 *I, i64 *J, double *V);
/* the updates of the cells at stream positions [p0, p1): (p1-p0)*(dim+1)*(dim+2) triples */
void orc_fem_stream_range(int dim, i64 npd, uint64_t seed, int order_mode, i64 p0, i64 p1, i64 *I, i64 *J,
                          double *V);

/* element data of the cells at stream positions [p0, p1) as a caller of testassemble! holds it (femtools.jl:46-67):
 * cellnodes Int64 nloc x (p1-p0), elmat = vol*S Float64 nloc x nloc x (p1-p0), diag = 0.1*vol/(dim+1) nloc x (p1-p0)
 * (NULL: not wanted), Julia (column-major) layouts; node_mode 1: permuted node numbering */
void orc_fem_mesh_range(int dim, i64 npd, uint64_t seed, int order_mode, int node_mode, uint64_t node_seed, i64 p0,
                        i64 p1, i64 *cellnodes, double *elmat, double *diag);
/* the update calls of femtools.jl:62-69 for element data in arrays, as triplets in call order; returns their number */
i64 orc_elements_stream(int nloc, i64 ncells, const i64 *cellnodes, const double *elmat, const double *diag, i64 *I,
                        i64 *J, double *V);

#ifdef __cplusplus
}
#endif
#endif
