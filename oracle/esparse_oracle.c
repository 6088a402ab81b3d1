/*
 * esparse_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * Plain-C restatement of the reference algorithm for the sparse-assembly hot
 * path.  See esparse_oracle.h for who may use it and for the parity-pinning
 * statement.  Every function cites the reference lines it follows
 * (paths relative to /root/reference).  Arrays that mirror Julia Vectors are
 * kept 1-based (slot 0 unused) so the index arithmetic reads like the source.
 *
 * Build: see oracle/Makefile  (gcc -O2 -ffp-contract=off: no FMA contraction,
 * so the value streams are bit-identical with the device generators).
 */
#include "esparse_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------ utils */
static void *xmalloc(size_t n) {
    void *p = malloc(n ? n : 1);
    if (!p) abort();
    return p;
}
static void *xcalloc(size_t n, size_t s) {
    void *p = calloc(n ? n : 1, s);
    if (!p) abort();
    return p;
}
static void *xrealloc(void *q, size_t n) {
    void *p = realloc(q, n ? n : 1);
    if (!p) abort();
    return p;
}
static double apply_op(int op, double a, double b) {
    return op == ORC_OP_SUB ? a - b : a + b;
}

/* ------------------------------------------------------------------- CSC */
/* Julia's SparseMatrixCSC{Float64,Int64}: colptr (n+1), rowval, nzval, all
 * holding 1-based values.  C arrays here are 0-based containers of them.   */
struct orc_csc {
    i64 m, n;
    i64 *colptr; /* n+1 entries, colptr[0]==1 */
    i64 *rowval;
    double *nzval;
};

orc_csc *orc_csc_new(i64 m, i64 n) { /* spzeros(m,n) */
    orc_csc *c = xmalloc(sizeof *c);
    c->m = m;
    c->n = n;
    c->colptr = xmalloc(sizeof(i64) * (size_t)(n + 1));
    for (i64 j = 0; j <= n; j++) c->colptr[j] = 1;
    c->rowval = xmalloc(8);
    c->nzval = xmalloc(8);
    return c;
}
i64 orc_csc_m(const orc_csc *c) { return c->m; }
i64 orc_csc_n(const orc_csc *c) { return c->n; }
i64 orc_csc_nnz(const orc_csc *c) { return c->colptr[c->n] - 1; }

orc_csc *orc_csc_from_arrays(i64 m, i64 n, const i64 *colptr, const i64 *rowval,
                             const double *nzval) {
    orc_csc *c = xmalloc(sizeof *c);
    c->m = m;
    c->n = n;
    c->colptr = xmalloc(sizeof(i64) * (size_t)(n + 1));
    memcpy(c->colptr, colptr, sizeof(i64) * (size_t)(n + 1));
    i64 z = colptr[n] - 1;
    c->rowval = xmalloc(sizeof(i64) * (size_t)z);
    c->nzval = xmalloc(sizeof(double) * (size_t)z);
    if (z > 0) {
        memcpy(c->rowval, rowval, sizeof(i64) * (size_t)z);
        memcpy(c->nzval, nzval, sizeof(double) * (size_t)z);
    }
    return c;
}
static orc_csc *csc_clone(const orc_csc *c) {
    return orc_csc_from_arrays(c->m, c->n, c->colptr, c->rowval, c->nzval);
}
void orc_csc_free(orc_csc *c) {
    if (!c) return;
    free(c->colptr);
    free(c->rowval);
    free(c->nzval);
    free(c);
}
void orc_csc_copy_out(const orc_csc *c, i64 *colptr, i64 *rowval, double *nzval) {
    i64 z = orc_csc_nnz(c);
    memcpy(colptr, c->colptr, sizeof(i64) * (size_t)(c->n + 1));
    if (z > 0) {
        memcpy(rowval, c->rowval, sizeof(i64) * (size_t)z);
        memcpy(nzval, c->nzval, sizeof(double) * (size_t)z);
    }
}

/* findindex(csc,i,j): src/matrix/sparsematrixcsc.jl:7-23.
 * bounds check -> BoundsError; empty column -> 0; searchsortedfirst over
 * rowval[r1:r2]; returns the 1-based position in nzval or 0.               */
i64 orc_csc_findindex(const orc_csc *c, i64 i, i64 j) {
    if (!(1 <= i && i <= c->m && 1 <= j && j <= c->n)) return ORC_ERR_BOUNDS;
    i64 r1 = c->colptr[j - 1];
    i64 r2 = c->colptr[j] - 1;
    if (r1 > r2) return 0;
    /* searchsortedfirst(rowval, i, r1, r2): first index with rowval[idx] >= i */
    i64 lo = r1 - 1, hi = r2 + 1;
    while (lo < hi - 1) {
        i64 mid = lo + ((hi - lo) >> 1);
        if (c->rowval[mid - 1] < i)
            lo = mid;
        else
            hi = mid;
    }
    r1 = hi;
    if (r1 > r2 || c->rowval[r1 - 1] != i) return 0;
    return r1;
}

/* pattern_equal: src/matrix/sparsematrixcsc.jl:83-85 */
int orc_csc_pattern_equal(const orc_csc *a, const orc_csc *b) {
    if (a->n != b->n) return 0;
    if (memcmp(a->colptr, b->colptr, sizeof(i64) * (size_t)(a->n + 1))) return 0;
    i64 z = orc_csc_nnz(a);
    return z == 0 || !memcmp(a->rowval, b->rowval, sizeof(i64) * (size_t)z);
}

/* phash: src/matrix/sparsematrixcsc.jl:74 is hash((hash(colptr),hash(rowval)))
 * with Julia's Base.hash, which cannot be reproduced outside Julia.  The
 * restatement keeps the CONTRACT (a 64-bit function of colptr and rowval
 * only) with a position-keyed multiply-xorshift sum that the device can also
 * evaluate in parallel (same formula in csrc/, compared in the tests).     */
static uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
uint64_t orc_csc_pattern_hash(const orc_csc *c) {
    uint64_t h1 = 0, h2 = 0;
    for (i64 j = 0; j <= c->n; j++)
        h1 += mix64((uint64_t)c->colptr[j] + 0x9E3779B97F4A7C15ull * (uint64_t)(j + 1));
    i64 z = orc_csc_nnz(c);
    for (i64 k = 0; k < z; k++)
        h2 += mix64((uint64_t)c->rowval[k] + 0x9E3779B97F4A7C15ull * (uint64_t)(k + 1));
    return mix64(h1 ^ mix64(h2 + 0xD1B54A32D192ED03ull));
}

/* dropzeros!(csc) (Julia stdlib, reached through abstractext..:207-211);
 * used by test/test_updates.jl:19,23.  Removes stored entries == 0.        */
i64 orc_csc_dropzeros(orc_csc *c) {
    i64 w = 0, k = 0;
    for (i64 j = 0; j < c->n; j++) {
        i64 end = c->colptr[j + 1] - 1;
        c->colptr[j] = w + 1;
        for (; k < end; k++) {
            if (c->nzval[k] != 0.0) {
                c->rowval[w] = c->rowval[k];
                c->nzval[w] = c->nzval[k];
                w++;
            }
        }
    }
    c->colptr[c->n] = w + 1;
    return w;
}

/* ------------------------------------------------------------------- LNK */
/* struct: src/matrix/sparsematrixlnk.jl:21-68.  colptr = next pointer
 * (0 terminates), rowval (0 = empty head slot), nzval; slots 1..n are the
 * column heads; nentries starts at n.                                      */
struct orc_lnk {
    i64 m, n, nnz, nentries, len;
    i64 *colptr, *rowval;
    double *nzval;
};

/* ctor: sparsematrixlnk.jl:75-77 */
orc_lnk *orc_lnk_new(i64 m, i64 n) {
    orc_lnk *l = xmalloc(sizeof *l);
    l->m = m;
    l->n = n;
    l->nnz = 0;
    l->nentries = n;
    l->len = n;
    l->colptr = xcalloc((size_t)n + 1, sizeof(i64));
    l->rowval = xcalloc((size_t)n + 1, sizeof(i64));
    l->nzval = xcalloc((size_t)n + 1, sizeof(double));
    return l;
}
void orc_lnk_free(orc_lnk *l) {
    if (!l) return;
    free(l->colptr);
    free(l->rowval);
    free(l->nzval);
    free(l);
}
i64 orc_lnk_nnz(const orc_lnk *l) { return l->nnz; }
i64 orc_lnk_nentries(const orc_lnk *l) { return l->nentries; }

/* findindex: sparsematrixlnk.jl:120-135 -> (k,0) found, (0,k0) last slot */
static int lnk_findindex(const orc_lnk *l, i64 i, i64 j, i64 *k_out, i64 *k0_out) {
    if (!((1 <= i && i <= l->m) & (1 <= j && j <= l->n))) return ORC_ERR_BOUNDS;
    i64 k = j, k0 = j;
    while (k > 0) {
        if (l->rowval[k] == i) {
            *k_out = k;
            *k0_out = 0;
            return 0;
        }
        k0 = k;
        k = l->colptr[k];
    }
    *k_out = 0;
    *k0_out = k0;
    return 0;
}

/* getindex: sparsematrixlnk.jl:142-149 */
int orc_lnk_getindex(const orc_lnk *l, i64 i, i64 j, double *out) {
    i64 k, k0;
    if (lnk_findindex(l, i, j, &k, &k0)) return ORC_ERR_BOUNDS;
    *out = k == 0 ? 0.0 : l->nzval[k];
    return 0;
}

/* addentry!: sparsematrixlnk.jl:151-171 (growth x1.25 :154-159) */
static i64 lnk_addentry(orc_lnk *l, i64 i, i64 k0) {
    l->nentries += 1;
    if (l->len < l->nentries) {
        i64 newsize = (i64)ceil(5.0 * (double)l->nentries / 4.0);
        l->nzval = xrealloc(l->nzval, sizeof(double) * (size_t)(newsize + 1));
        l->rowval = xrealloc(l->rowval, sizeof(i64) * (size_t)(newsize + 1));
        l->colptr = xrealloc(l->colptr, sizeof(i64) * (size_t)(newsize + 1));
        l->len = newsize;
    }
    l->rowval[l->nentries] = i;
    l->colptr[l->nentries] = 0;
    l->colptr[k0] = l->nentries;
    l->nnz += 1;
    return l->nentries;
}

/* setindex!: sparsematrixlnk.jl:178-201 */
int orc_lnk_setindex(orc_lnk *l, double v, i64 i, i64 j) {
    if (!((1 <= i && i <= l->m) & (1 <= j && j <= l->n))) return ORC_ERR_BOUNDS;
    if (l->rowval[j] == 0 && v != 0.0) {
        l->rowval[j] = i;
        l->nzval[j] = v;
        l->nnz += 1;
        return 0;
    }
    i64 k, k0;
    lnk_findindex(l, i, j, &k, &k0);
    if (k > 0) {
        l->nzval[k] = v;
        return 0;
    }
    if (v != 0.0) {
        k = lnk_addentry(l, i, k0);
        l->nzval[k] = v;
    }
    return 0;
}

/* updateindex!: sparsematrixlnk.jl:210-228.  No bounds check of its own
 * before touching rowval[j]; the restatement checks so
 * that C never reads out of range -- callers reach it only after the
 * wrapper's findindex(csc) has thrown for bad indices.                     */
int orc_lnk_updateindex(orc_lnk *l, int op, double v, i64 i, i64 j) {
    if (!((1 <= i && i <= l->m) & (1 <= j && j <= l->n))) return ORC_ERR_BOUNDS;
    if (l->rowval[j] == 0 && v != 0.0) {
        l->rowval[j] = i;
        l->nzval[j] = apply_op(op, l->nzval[j], v);
        l->nnz += 1;
        return 0;
    }
    i64 k, k0;
    lnk_findindex(l, i, j, &k, &k0);
    if (k > 0) {
        l->nzval[k] = apply_op(op, l->nzval[k], v);
        return 0;
    }
    if (v != 0.0) {
        k = lnk_addentry(l, i, k0);
        l->nzval[k] = apply_op(op, 0.0, v);
    }
    return 0;
}

/* rawupdateindex!: sparsematrixlnk.jl:237-253 */
int orc_lnk_rawupdateindex(orc_lnk *l, int op, double v, i64 i, i64 j) {
    if (!((1 <= i && i <= l->m) & (1 <= j && j <= l->n))) return ORC_ERR_BOUNDS;
    if (l->rowval[j] == 0) {
        l->rowval[j] = i;
        l->nzval[j] = apply_op(op, l->nzval[j], v);
        l->nnz += 1;
        return 0;
    }
    i64 k, k0;
    lnk_findindex(l, i, j, &k, &k0);
    if (k > 0) {
        l->nzval[k] = apply_op(op, l->nzval[k], v);
    } else {
        k = lnk_addentry(l, i, k0);
        l->nzval[k] = apply_op(op, 0.0, v);
    }
    return 0;
}

/* SparseMatrixLNK(csc): sparsematrixlnk.jl:109-118 */
orc_lnk *orc_lnk_from_csc(const orc_csc *c) {
    orc_lnk *l = orc_lnk_new(c->m, c->n);
    for (i64 j = 1; j <= c->n; j++)
        for (i64 k = c->colptr[j - 1]; k <= c->colptr[j] - 1; k++)
            orc_lnk_setindex(l, c->nzval[k - 1], c->rowval[k - 1], j);
    return l;
}

typedef struct {
    i64 rowval;
    double nzval;
} colentry; /* ColEntry: sparsematrixlnk.jl:281-287 */

static int colentry_less(const void *a, const void *b) {
    i64 ra = ((const colentry *)a)->rowval, rb = ((const colentry *)b)->rowval;
    return (ra > rb) - (ra < rb);
}

/* Base.:+(lnk,csc): sparsematrixlnk.jl:294-383 -- THE flush kernel.
 * per column: gather list entries with rowval>0 (:330-338), sort by row
 * (:339), 3-way merge with the CSC column (:346-376): csc<lnk copy csc;
 * equal rows -> csc.nzval + lnk.nzval (csc operand first, :363); lnk-only
 * insert.  No zero dropping.  Result trimmed to inz-1 (:378-382).          */
orc_csc *orc_lnk_plus_csc(const orc_lnk *l, const orc_csc *c) {
    if (c->m != l->m || c->n != l->n) return NULL;
    i64 cscnnz = orc_csc_nnz(c);
    i64 xnnz = cscnnz + l->nnz;
    orc_csc *r = xmalloc(sizeof *r);
    r->m = c->m;
    r->n = c->n;
    r->colptr = xmalloc(sizeof(i64) * (size_t)(c->n + 1));
    r->rowval = xmalloc(sizeof(i64) * (size_t)xnnz);
    r->nzval = xmalloc(sizeof(double) * (size_t)xnnz);

    i64 lnk_maxcol = 0; /* :306-316 */
    for (i64 j = 1; j <= c->n; j++) {
        i64 lcol = 0, k = j;
        while (k > 0) {
            lcol++;
            k = l->colptr[k];
        }
        if (lcol > lnk_maxcol) lnk_maxcol = lcol;
    }
    colentry *col = xmalloc(sizeof(colentry) * (size_t)(lnk_maxcol + 1));

    i64 inz = 1;
    for (i64 j = 1; j <= c->n; j++) {
        i64 k = j, l_lnk_col = 0;
        while (k > 0) {
            if (l->rowval[k] > 0) {
                l_lnk_col++;
                col[l_lnk_col].rowval = l->rowval[k];
                col[l_lnk_col].nzval = l->nzval[k];
            }
            k = l->colptr[k];
        }
        if (l_lnk_col > 1) qsort(col + 1, (size_t)l_lnk_col, sizeof(colentry), colentry_less);

        r->colptr[j - 1] = inz;
        i64 jlnk = 1;
        i64 jcsc = c->colptr[j - 1];
        for (;;) {
            int in_csc = (cscnnz > 0) && (jcsc < c->colptr[j]); /* :323 */
            int in_lnk = (jlnk <= l_lnk_col);                   /* :325 */
            if (in_csc && ((in_lnk && c->rowval[jcsc - 1] < col[jlnk].rowval) || !in_lnk)) {
                r->rowval[inz - 1] = c->rowval[jcsc - 1];
                r->nzval[inz - 1] = c->nzval[jcsc - 1];
                jcsc++;
                inz++;
            } else if (in_csc && (in_lnk && c->rowval[jcsc - 1] == col[jlnk].rowval)) {
                r->rowval[inz - 1] = c->rowval[jcsc - 1];
                r->nzval[inz - 1] = c->nzval[jcsc - 1] + col[jlnk].nzval;
                jcsc++;
                inz++;
                jlnk++;
            } else if (in_lnk) {
                r->rowval[inz - 1] = col[jlnk].rowval;
                r->nzval[inz - 1] = col[jlnk].nzval;
                jlnk++;
                inz++;
            } else {
                break;
            }
        }
    }
    r->colptr[c->n] = inz;
    free(col);
    return r;
}

/* --------------------------------------------- ExtendableSparseMatrixCSC */
/* struct: src/matrix/extendable.jl:10-25 (cscmatrix, lazily created
 * lnkmatrix, phash).  The Generic wrapper
 * (genericextendablesparsematrixcsc.jl:1-92) routes identically; its flush
 * gate `nnz(x)>0` (:31-37) equals `lnk!=nothing && nnz(lnk)>0` here.       */
struct orc_ext {
    orc_csc *csc;
    orc_lnk *lnk; /* NULL == nothing */
    uint64_t phash;
    i64 flush_count;
};

orc_ext *orc_ext_new(i64 m, i64 n) { /* extendable.jl:39-41: phash = 0 */
    orc_ext *e = xmalloc(sizeof *e);
    e->csc = orc_csc_new(m, n);
    e->lnk = NULL;
    e->phash = 0;
    e->flush_count = 0;
    return e;
}
orc_ext *orc_ext_from_csc(const orc_csc *c) { /* extendable.jl:61-63 */
    orc_ext *e = xmalloc(sizeof *e);
    e->csc = csc_clone(c);
    e->lnk = NULL;
    e->phash = orc_csc_pattern_hash(c);
    e->flush_count = 0;
    return e;
}
void orc_ext_free(orc_ext *e) {
    if (!e) return;
    orc_csc_free(e->csc);
    orc_lnk_free(e->lnk);
    free(e);
}
static void ext_need_lnk(orc_ext *e) { /* extendable.jl:168-170 */
    if (!e->lnk) e->lnk = orc_lnk_new(e->csc->m, e->csc->n);
}

/* updateindex!: extendable.jl:159-174 */
int orc_ext_updateindex(orc_ext *e, int op, double v, i64 i, i64 j) {
    i64 k = orc_csc_findindex(e->csc, i, j);
    if (k < 0) return (int)k;
    if (k > 0) {
        e->csc->nzval[k - 1] = apply_op(op, e->csc->nzval[k - 1], v);
    } else {
        ext_need_lnk(e);
        orc_lnk_updateindex(e->lnk, op, v, i, j);
    }
    return 0;
}
/* rawupdateindex!: extendable.jl:181-197 */
int orc_ext_rawupdateindex(orc_ext *e, int op, double v, i64 i, i64 j) {
    i64 k = orc_csc_findindex(e->csc, i, j);
    if (k < 0) return (int)k;
    if (k > 0) {
        e->csc->nzval[k - 1] = apply_op(op, e->csc->nzval[k - 1], v);
    } else {
        ext_need_lnk(e);
        orc_lnk_rawupdateindex(e->lnk, op, v, i, j);
    }
    return 0;
}
/* setindex!: extendable.jl:205-218 */
int orc_ext_setindex(orc_ext *e, double v, i64 i, i64 j) {
    i64 k = orc_csc_findindex(e->csc, i, j);
    if (k < 0) return (int)k;
    if (k > 0) {
        e->csc->nzval[k - 1] = v;
    } else {
        ext_need_lnk(e);
        orc_lnk_setindex(e->lnk, v, i, j);
    }
    return 0;
}
/* getindex: extendable.jl:226-238 */
int orc_ext_getindex(const orc_ext *e, i64 i, i64 j, double *out) {
    i64 k = orc_csc_findindex(e->csc, i, j);
    if (k < 0) return (int)k;
    if (k > 0)
        *out = e->csc->nzval[k - 1];
    else if (!e->lnk)
        *out = 0.0;
    else
        return orc_lnk_getindex(e->lnk, i, j, out);
    return 0;
}
/* flush!: extendable.jl:248-255 */
int orc_ext_flush(orc_ext *e) {
    if (e->lnk && e->lnk->nnz > 0) {
        orc_csc *r = orc_lnk_plus_csc(e->lnk, e->csc);
        orc_csc_free(e->csc);
        e->csc = r;
        orc_lnk_free(e->lnk);
        e->lnk = NULL;
        e->phash = orc_csc_pattern_hash(e->csc);
        e->flush_count++;
        return 1;
    }
    return 0;
}
/* sparse(ext): extendable.jl:258-261 */
const orc_csc *orc_ext_csc(orc_ext *e) {
    orc_ext_flush(e);
    return e->csc;
}
/* nnz(ext): abstractextendablesparsematrixcsc.jl:80 (flush then nnz(csc)) */
i64 orc_ext_nnz(orc_ext *e) {
    orc_ext_flush(e);
    return orc_csc_nnz(e->csc);
}
i64 orc_ext_pending(const orc_ext *e) { return e->lnk ? e->lnk->nnz : 0; }
uint64_t orc_ext_phash(const orc_ext *e) { return e->phash; }
i64 orc_ext_flush_count(const orc_ext *e) { return e->flush_count; }
/* reset!: extendable.jl:269-272 (phash is NOT reset) */
void orc_ext_reset(orc_ext *e) {
    i64 m = e->csc->m, n = e->csc->n;
    orc_csc_free(e->csc);
    e->csc = orc_csc_new(m, n);
    orc_lnk_free(e->lnk);
    e->lnk = NULL;
}
/* zero!(A::ExtendableSparseMatrix): sprand.jl:76,82 -- nonzeros(A) flushes
 * (abstractext..:24 -> sparse -> flush!) then sets every stored value to 0 */
void orc_ext_zero_values(orc_ext *e) {
    orc_ext_flush(e);
    i64 z = orc_csc_nnz(e->csc);
    for (i64 k = 0; k < z; k++) e->csc->nzval[k] = 0.0;
}
i64 orc_ext_dropzeros(orc_ext *e) {
    orc_ext_flush(e);
    return orc_csc_dropzeros(e->csc);
}

int orc_ext_apply(orc_ext *e, i64 count, const uint8_t *kinds, const i64 *I, const i64 *J,
                  const double *V, i64 *errpos) {
    for (i64 t = 0; t < count; t++) {
        int rc;
        switch (kinds ? kinds[t] : ORC_KIND_UPDATE) {
        case ORC_KIND_SET:
            rc = orc_ext_setindex(e, V[t], I[t], J[t]);
            break;
        case ORC_KIND_UPDATE:
            rc = orc_ext_updateindex(e, ORC_OP_ADD, V[t], I[t], J[t]);
            break;
        case ORC_KIND_RAWUPDATE:
            rc = orc_ext_rawupdateindex(e, ORC_OP_ADD, V[t], I[t], J[t]);
            break;
        default: { /* A[i,j] += v  ==  setindex!(A, getindex(A,i,j)+v, i, j) */
            double old;
            rc = orc_ext_getindex(e, I[t], J[t], &old);
            if (!rc) rc = orc_ext_setindex(e, old + V[t], I[t], J[t]);
        }
        }
        if (rc) {
            if (errpos) *errpos = t;
            return rc;
        }
    }
    return 0;
}

/* ----------------------------------- GenericMTExtendableSparseMatrixCSC */
/* src/matrix/genericmtextendablesparsematrixcsc.jl:1-114 with
 * Tm = SparseMatrixDILNKC, whose insert rules (sparsematrixdilnkc.jl:184-237)
 * are the LNK rules above with Dict column heads, so orc_lnk serves as the
 * per-partition buffer.  flush! (:45-51) = Base.sum(xmatrices,csc)
 * (sparsematrixdilnkc.jl:397-435): I,J,V = CSC entries, then every buffer in
 * partition order, then SparseArrays.sparse!(I,J,V,m,n,+).  sparse! is Julia
 * stdlib (not under /root/reference, no pinned version): its published
 * behaviour is that duplicates are combined left to right in input order and
 * numerical zeros are kept; that is what is restated here (PARITY UNPINNED
 * for this function).                                       */
struct orc_mt {
    orc_csc *csc;
    i64 np;
    orc_lnk **x;
};
orc_mt *orc_mt_new(i64 m, i64 n, i64 np) {
    orc_mt *t = xmalloc(sizeof *t);
    t->csc = orc_csc_new(m, n);
    t->np = np;
    t->x = xmalloc(sizeof(orc_lnk *) * (size_t)np);
    for (i64 p = 0; p < np; p++) t->x[p] = orc_lnk_new(m, n);
    return t;
}
void orc_mt_free(orc_mt *t) {
    if (!t) return;
    for (i64 p = 0; p < t->np; p++) orc_lnk_free(t->x[p]);
    free(t->x);
    orc_csc_free(t->csc);
    free(t);
}
i64 orc_mt_nnznew(const orc_mt *t) { /* :84 */
    i64 s = 0;
    for (i64 p = 0; p < t->np; p++) s += t->x[p]->nnz;
    return s;
}
int orc_mt_setindex(orc_mt *t, double v, i64 i, i64 j) { /* :59-69 */
    i64 k = orc_csc_findindex(t->csc, i, j);
    if (k < 0) return (int)k;
    if (k > 0) {
        t->csc->nzval[k - 1] = v;
        return 0;
    }
    return ORC_ERR_MT_NEW_SETINDEX;
}
int orc_mt_getindex(const orc_mt *t, i64 i, i64 j, double *out) { /* :71-82 */
    i64 k = orc_csc_findindex(t->csc, i, j);
    if (k < 0) return (int)k;
    if (k > 0) {
        *out = t->csc->nzval[k - 1];
        return 0;
    }
    if (orc_mt_nnznew(t) == 0) {
        *out = 0.0;
        return 0;
    }
    return ORC_ERR_MT_GETINDEX_PENDING;
}
int orc_mt_rawupdateindex(orc_mt *t, int op, double v, i64 i, i64 j, i64 tid) { /* :87-99 */
    i64 k = orc_csc_findindex(t->csc, i, j);
    if (k < 0) return (int)k;
    if (k > 0)
        t->csc->nzval[k - 1] = apply_op(op, t->csc->nzval[k - 1], v);
    else
        orc_lnk_rawupdateindex(t->x[tid - 1], op, v, i, j);
    return 0;
}
int orc_mt_updateindex(orc_mt *t, int op, double v, i64 i, i64 j, i64 tid) { /* :102-114 */
    i64 k = orc_csc_findindex(t->csc, i, j);
    if (k < 0) return (int)k;
    if (k > 0)
        t->csc->nzval[k - 1] = apply_op(op, t->csc->nzval[k - 1], v);
    else
        orc_lnk_updateindex(t->x[tid - 1], op, v, i, j);
    return 0;
}
/* a batch of (raw)updateindex! calls with one tid: the loop a task of test/femtools.jl:88-107 runs over its partition */
int orc_mt_apply(orc_mt *t, i64 count, const uint8_t *kinds, const i64 *I, const i64 *J, const double *V, i64 tid) {
    for (i64 e = 0; e < count; e++) {
        const int k = kinds ? kinds[e] : ORC_KIND_RAWUPDATE;
        int rc;
        if (k == ORC_KIND_UPDATE)
            rc = orc_mt_updateindex(t, ORC_OP_ADD, V[e], I[e], J[e], tid);
        else if (k == ORC_KIND_RAWUPDATE)
            rc = orc_mt_rawupdateindex(t, ORC_OP_ADD, V[e], I[e], J[e], tid);
        else
            rc = orc_mt_setindex(t, V[e], I[e], J[e]);
        if (rc) return rc;
    }
    return 0;
}
/* flush!: :45-51 + Base.sum: sparsematrixdilnkc.jl:397-435 */
int orc_mt_flush(orc_mt *t) {
    i64 lnew = orc_mt_nnznew(t);
    i64 m = t->csc->m, n = t->csc->n;
    if (lnew > 0) {
        /* Combining left to right in (csc, x[0], x[1], ...) order is what
         * sparse!(I,J,V,m,n,+) does with that input order.  Implemented as
         * successive lnk+csc merges, whose equal-row branch is `csc + lnk`. */
        for (i64 p = 0; p < t->np; p++) {
            if (t->x[p]->nnz == 0) continue;
            orc_csc *r = orc_lnk_plus_csc(t->x[p], t->csc);
            orc_csc_free(t->csc);
            t->csc = r;
        }
    }
    for (i64 p = 0; p < t->np; p++) {
        orc_lnk_free(t->x[p]);
        t->x[p] = orc_lnk_new(m, n);
    }
    return lnew > 0;
}
const orc_csc *orc_mt_csc(orc_mt *t) {
    orc_mt_flush(t);
    return t->csc;
}
void orc_mt_reset(orc_mt *t) { /* :31-42 */
    i64 m = t->csc->m, n = t->csc->n;
    orc_csc_free(t->csc);
    t->csc = orc_csc_new(m, n);
    for (i64 p = 0; p < t->np; p++) {
        orc_lnk_free(t->x[p]);
        t->x[p] = orc_lnk_new(m, n);
    }
}

/* ------------------------------------------------- Dirichlet helpers */
/* mark_dirichlet / eliminate_dirichlet!: sparsematrixcsc.jl:94-144, statement for statement */
void orc_csc_mark_dirichlet(const orc_csc *c, double penalty, uint8_t *marker) {
    for (i64 i = 1; i <= c->n; i++) {
        marker[i - 1] = 0;
        for (i64 j = c->colptr[i - 1]; j < c->colptr[i]; j++)
            if (c->rowval[j - 1] == i && c->nzval[j - 1] >= penalty) marker[i - 1] = 1;
    }
}
void orc_csc_eliminate_dirichlet(orc_csc *c, const uint8_t *marker) {
    for (i64 i = 1; i <= c->n; i++) {
        if (marker[i - 1])
            for (i64 j = c->colptr[i - 1]; j < c->colptr[i]; j++) c->nzval[j - 1] = c->rowval[j - 1] == i ? 1.0 : 0.0;
        for (i64 j = c->colptr[i - 1]; j < c->colptr[i]; j++)
            if (c->rowval[j - 1] != i && marker[c->rowval[j - 1] - 1]) c->nzval[j - 1] = 0.0;
    }
}

/* ------------------------------------------------------------- mul! */
/* LinearAlgebra.mul!(r, ext, x): abstractextendablesparsematrixcsc.jl:179-181 forwards to the CSC; the
 * loop is the one spelled out in genericmtextendablesparsematrixcsc.jl:124-143 (with the default
 * single partition the columns are visited in increasing order): r .= 0; r[rows[i]] += vals[i]*x[col]. */
/* jacobi(A): src/factorizations/jacobi.jl:5-12 -- invdiag[i] = one(Tv) / A[i,i]; getindex of an absent position is zero */
void orc_csc_jacobi(const orc_csc *c, double *invdiag) {
    for (i64 i = 1; i <= c->n; i++) {
        i64 k = (i <= c->m) ? orc_csc_findindex(c, i, i) : 0;
        invdiag[i - 1] = 1.0 / (k > 0 ? c->nzval[k - 1] : 0.0);
    }
}
/* ilu0(A): src/factorizations/ilu0.jl:8-41, statement for statement.  xdiag is `undef` in the reference: the
 * `xdiag[i] -= ...` of iteration j works on memory that iteration i overwrites with `xdiag[i] = 1/nzval[idiag[i]]`
 * before anything reads it (here the array starts as zeros; the result is the same).  Returns 0, or j (1-based) of the
 * first column without a diagonal entry (the reference reads an undefined idiag[j] there). */
i64 orc_csc_ilu0(const orc_csc *c, double *xdiag, i64 *idiag) {
    const i64 n = c->n;
    const i64 *colptr = c->colptr, *rowval = c->rowval;
    const double *nzval = c->nzval;
    for (i64 j = 1; j <= n; j++) {
        idiag[j - 1] = 0;
        for (i64 k = colptr[j - 1]; k <= colptr[j] - 1; k++) {
            i64 i = rowval[k - 1];
            if (i == j) {
                idiag[j - 1] = k;
                break;
            }
        }
        if (idiag[j - 1] == 0) return j;
    }
    for (i64 j = 0; j < n; j++) xdiag[j] = 0.0;
    for (i64 j = 1; j <= n; j++) {
        xdiag[j - 1] = 1.0 / nzval[idiag[j - 1] - 1];
        for (i64 k = idiag[j - 1] + 1; k <= colptr[j] - 1; k++) {
            i64 i = rowval[k - 1];
            for (i64 l = colptr[i - 1]; l <= colptr[i] - 1; l++) {
                if (rowval[l - 1] == j) {
                    xdiag[i - 1] -= nzval[l - 1] * xdiag[j - 1] * nzval[k - 1];
                    break;
                }
            }
        }
    }
    return 0;
}

void orc_csc_mul(const orc_csc *c, const double *x, double *r) {
    for (i64 i = 0; i < c->m; i++) r[i] = 0.0;
    for (i64 col = 1; col <= c->n; col++)
        for (i64 k = c->colptr[col - 1]; k < c->colptr[col]; k++) r[c->rowval[k - 1] - 1] += c->nzval[k - 1] * x[col - 1];
}

/* ------------------------------------------------ sparse(I,J,V,m,n,+) */
/* The COO constructors ExtendableSparseMatrixCSC(I,J,V[,m,n]) (extendable.jl:92-104) and fdrand_coo
 * (sprand.jl:134-185) call SparseArrays.sparse(I,J,V,m,n[,+]) -- Julia stdlib, not under /root/reference,
 * no pinned version (PARITY UNPINNED).  Restated from its published behaviour: the
 * entries of a column in increasing row order, duplicates combined left to right in input order,
 * the first value taken as it is (no 0+v), numerical zeros kept, indices outside m x n rejected. */
typedef struct {
    i64 row, idx;
} coo_item;
static int coo_cmp(const void *a, const void *b) {
    const coo_item *x = a, *y = b;
    if (x->row != y->row) return x->row < y->row ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
}
orc_csc *orc_sparse_coo(i64 m, i64 n, i64 count, const i64 *I, const i64 *J, const double *V) {
    for (i64 t = 0; t < count; t++)
        if (I[t] < 1 || I[t] > m || J[t] < 1 || J[t] > n) return NULL;
    i64 *start = xmalloc(sizeof(i64) * (size_t)(n + 2));
    for (i64 j = 0; j <= n + 1; j++) start[j] = 0;
    for (i64 t = 0; t < count; t++) start[J[t] + 1]++;
    for (i64 j = 1; j <= n + 1; j++) start[j] += start[j - 1];
    coo_item *it = xmalloc(sizeof(coo_item) * (size_t)(count > 0 ? count : 1));
    i64 *fill = xmalloc(sizeof(i64) * (size_t)(n + 2));
    for (i64 j = 0; j <= n + 1; j++) fill[j] = start[j];
    for (i64 t = 0; t < count; t++) { /* stable counting sort by column */
        i64 at = fill[J[t]]++;
        it[at].row = I[t];
        it[at].idx = t;
    }
    orc_csc *c = orc_csc_new(m, n);
    free(c->rowval);
    free(c->nzval);
    c->rowval = xmalloc(sizeof(i64) * (size_t)(count > 0 ? count : 1));
    c->nzval = xmalloc(sizeof(double) * (size_t)(count > 0 ? count : 1));
    i64 z = 0;
    for (i64 j = 1; j <= n; j++) {
        i64 a = start[j], b = start[j + 1];
        qsort(it + a, (size_t)(b - a), sizeof(coo_item), coo_cmp);
        c->colptr[j - 1] = z + 1;
        for (i64 q = a; q < b; q++) {
            if (q > a && it[q].row == it[q - 1].row) {
                c->nzval[z - 1] = c->nzval[z - 1] + V[it[q].idx];
            } else {
                c->rowval[z] = it[q].row;
                c->nzval[z] = V[it[q].idx];
                z++;
            }
        }
    }
    c->colptr[n] = z + 1;
    free(start);
    free(fill);
    free(it);
    return c;
}

/* --------------------------------------------------------------- streams */
/* Counter-based uniform in [0,1): splitmix64 of (seed, counter), top 53 bits.
 * Same formula in csrc/ (device generators) -- values are bit-identical.   */
double orc_uniform(uint64_t seed, uint64_t counter) {
    uint64_t z = seed + (counter + 1) * 0x9E3779B97F4A7C15ull;
    z = mix64(z);
    return (double)(z >> 11) * 0x1.0p-53;
}
static double fd_rand(int mode, uint64_t seed, uint64_t counter) {
    if (mode == 0) return 1.0;                          /* rand = ()->1 (test_fdrand.jl:23) */
    if (mode == 1) return 0.1 + orc_uniform(seed, counter); /* sprand.jl:232 */
    return orc_uniform(seed, counter);                  /* sprand.jl:63 */
}

/* number of update calls of fdrand!: sprand.jl:100-124 */
i64 orc_fdrand_count(i64 nx, i64 ny, i64 nz) {
    i64 e = 0;
    e += 4 * (nx - 1) * ny * nz + (nx == 1 ? 1 : 2) * ny * nz;
    e += 4 * nx * (ny - 1) * nz + (ny > 2 ? 2 * nx * nz : 0);
    e += 4 * nx * ny * (nz - 1) + (nz > 2 ? 2 * nx * ny : 0);
    return e;
}
i64 orc_fdrand_nnz(i64 nx, i64 ny, i64 nz) {
    return nx * ny * nz + 2 * ((nx - 1) * ny * nz + nx * (ny - 1) * nz + nx * ny * (nz - 1));
}

typedef void (*fd_sink)(void *ctx, double v, i64 i, i64 j);

/* the update sequence of fdrand!: sprand.jl:87-126.  The random draw of
 * slot s (0..5) at node l uses counter 6*(l-1)+s, so the stream can also be
 * produced out of order (device generator).  `rand()*hy*hz/hx` is evaluated
 * left to right as Julia does.                                             */
/* nodes l_begin <= l < l_end of the loop nest (the whole nest: 1 .. nx*ny*nz+1) */
static void fdrand_walk_range(i64 nx, i64 ny, i64 nz, int mode, uint64_t seed, fd_sink sink, void *ctx, i64 l_begin,
                              i64 l_end) {
    double hx = 1.0 / (double)nx, hy = 1.0 / (double)ny, hz = 1.0 / (double)nz;
    i64 nxy = nx * ny, l = l_begin;
    if (l_begin >= l_end) return;
    /* the loop nest `for k, for j, for i` entered at node l_begin */
    const i64 k0 = (l_begin - 1) / nxy + 1, j0 = ((l_begin - 1) / nx) % ny + 1, i0 = (l_begin - 1) % nx + 1;
    for (i64 k = k0; k <= nz; k++)
        for (i64 j = (k == k0 ? j0 : 1); j <= ny; j++)
            for (i64 i = (k == k0 && j == j0 ? i0 : 1); i <= nx; i++) {
                if (l >= l_end) return;
                uint64_t c = 6 * (uint64_t)(l - 1);
                if (i < nx) {
                    double v = fd_rand(mode, seed, c + 0) * hy * hz / hx;
                    sink(ctx, -v, l, l + 1);
                    sink(ctx, -v, l + 1, l);
                    sink(ctx, v, l, l);
                    sink(ctx, v, l + 1, l + 1);
                }
                if (i == 1 || i == nx) sink(ctx, fd_rand(mode, seed, c + 1) * hy * hz, l, l);
                if (j < ny) {
                    double v = fd_rand(mode, seed, c + 2) * hx * hz / hy;
                    sink(ctx, -v, l, l + nx);
                    sink(ctx, -v, l + nx, l);
                    sink(ctx, v, l, l);
                    sink(ctx, v, l + nx, l + nx);
                }
                if (ny > 2 && (j == 1 || j == ny))
                    sink(ctx, fd_rand(mode, seed, c + 3) * hx * hz, l, l);
                if (k < nz) {
                    double v = fd_rand(mode, seed, c + 4) * hx * hy / hz;
                    sink(ctx, -v, l, l + nxy);
                    sink(ctx, -v, l + nxy, l);
                    sink(ctx, v, l, l);
                    sink(ctx, v, l + nxy, l + nxy);
                }
                if (nz > 2 && (k == 1 || k == nz))
                    sink(ctx, fd_rand(mode, seed, c + 5) * hx * hy, l, l);
                l++;
            }
}
static void fdrand_walk(i64 nx, i64 ny, i64 nz, int mode, uint64_t seed, fd_sink sink, void *ctx) {
    fdrand_walk_range(nx, ny, nz, mode, seed, sink, ctx, 1, nx * ny * nz + 1);
}

typedef struct {
    i64 *I, *J;
    double *V;
    i64 pos;
} coo_ctx;
static void coo_sink(void *c, double v, i64 i, i64 j) {
    coo_ctx *x = c;
    x->I[x->pos] = i;
    x->J[x->pos] = j;
    x->V[x->pos] = v;
    x->pos++;
}
void orc_fdrand_stream(i64 nx, i64 ny, i64 nz, int mode, uint64_t seed, i64 *I, i64 *J,
                       double *V) {
    coo_ctx c = {I, J, V, 0};
    fdrand_walk(nx, ny, nz, mode, seed, coo_sink, &c);
}

typedef struct {
    orc_ext *e;
    int style;
} ext_ctx;
static void ext_sink(void *c, double v, i64 i, i64 j) {
    ext_ctx *x = c;
    switch (x->style) {
    case ORC_KIND_UPDATE:
        orc_ext_updateindex(x->e, ORC_OP_ADD, v, i, j);
        break;
    case ORC_KIND_RAWUPDATE:
        orc_ext_rawupdateindex(x->e, ORC_OP_ADD, v, i, j);
        break;
    default: { /* update = (A,v,i,j)->A[i,j]+=v  (sprand.jl:62) */
        double old = 0.0;
        orc_ext_getindex(x->e, i, j, &old);
        orc_ext_setindex(x->e, old + v, i, j);
    }
    }
}
/* fdrand!(A,...): sprand.jl:58-126 on an ExtendableSparseMatrix */
int orc_fdrand_ext(orc_ext *e, i64 nx, i64 ny, i64 nz, int mode, uint64_t seed, int style) {
    i64 N = nx * ny * nz;
    if (e->csc->m != N || e->csc->n != N) return -1; /* "Matrix size mismatch" */
    orc_ext_zero_values(e);
    ext_ctx c = {e, style};
    fdrand_walk(nx, ny, nz, mode, seed, ext_sink, &c);
    orc_ext_flush(e);
    return 0;
}

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
/* CPU baseline of bench.py: fdrand(Float64,nx,ny,nz; matrixtype=
 * ExtendableSparseMatrix) (sprand.jl:226-256), single thread, timed split
 * into the insertion loop and flush!.  Values: rand mode 1, seed 0x5EED0002. */
i64 orc_bench_fdrand(i64 nx, i64 ny, i64 nz, int style, double *t_insert_s, double *t_flush_s) {
    i64 N = nx * ny * nz;
    double t0 = now_s();
    orc_ext *e = orc_ext_new(N, N);
    orc_ext_zero_values(e);
    ext_ctx c = {e, style};
    fdrand_walk(nx, ny, nz, 1, 0x5EED0002ull, ext_sink, &c);
    double t1 = now_s();
    orc_ext_flush(e);
    double t2 = now_s();
    i64 z = orc_csc_nnz(e->csc);
    orc_ext_free(e);
    if (t_insert_s) *t_insert_s = t1 - t0;
    if (t_flush_s) *t_flush_s = t2 - t1;
    return z;
}

/* Secondary CPU baseline of bench.py (cpu_baseline.mt): the shape of GenericMTExtendableSparseMatrixCSC with
 * Tm = SparseMatrixDILNKC.  np threads; thread p issues rawupdateindex!(A,+,v,i,j,tid=p) for the nodes of its slab
 * (genericmtextendablesparsematrixcsc.jl:87-99: the CSC is empty, so everything goes to xmatrices[tid]; orc_lnk
 * serves as the per-partition buffer, see orc_mt above), then flush! = Base.sum(xmatrices, csc)
 * (sparsematrixdilnkc.jl:397-435): every buffer's entries, partition by partition and column by column, into
 * I,J,V, then sparse!(I,J,V,m,n,+) -- serial, as in the reference.  Values: rand mode 1, seed 0x5EED0002. */
typedef struct {
    orc_lnk *x;
    i64 nx, ny, nz, l_begin, l_end;
} mt_job;
static void mt_sink(void *c, double v, i64 i, i64 j) { orc_lnk_rawupdateindex((orc_lnk *)c, ORC_OP_ADD, v, i, j); }
static void *mt_worker(void *arg) {
    mt_job *jb = arg;
    jb->x = orc_lnk_new(jb->nx * jb->ny * jb->nz, jb->nx * jb->ny * jb->nz); /* T_ext(m,n) of this partition */
    fdrand_walk_range(jb->nx, jb->ny, jb->nz, 1, 0x5EED0002ull, mt_sink, jb->x, jb->l_begin, jb->l_end);
    return NULL;
}
i64 orc_bench_fdrand_mt(i64 nx, i64 ny, i64 nz, i64 np, double *t_insert_s, double *t_merge_s) {
    i64 N = nx * ny * nz;
    if (np < 1) np = 1;
    double t0 = now_s();
    mt_job *jobs = xmalloc(sizeof(mt_job) * (size_t)np);
    pthread_t *th = xmalloc(sizeof(pthread_t) * (size_t)np);
    for (i64 p = 0; p < np; p++) {
        jobs[p].x = NULL;
        jobs[p].nx = nx;
        jobs[p].ny = ny;
        jobs[p].nz = nz;
        jobs[p].l_begin = 1 + N * p / np;
        jobs[p].l_end = 1 + N * (p + 1) / np;
        pthread_create(&th[p], NULL, mt_worker, &jobs[p]);
    }
    for (i64 p = 0; p < np; p++) pthread_join(th[p], NULL);
    double t1 = now_s();
    i64 lnew = 0;
    for (i64 p = 0; p < np; p++) lnew += jobs[p].x->nnz;
    i64 *I = xmalloc(sizeof(i64) * (size_t)(lnew > 0 ? lnew : 1)), *J = xmalloc(sizeof(i64) * (size_t)(lnew > 0 ? lnew : 1));
    double *V = xmalloc(sizeof(double) * (size_t)(lnew > 0 ? lnew : 1));
    i64 at = 0;
    for (i64 p = 0; p < np; p++) {
        const orc_lnk *l = jobs[p].x;
        for (i64 j = 1; j <= l->n; j++) {
            if (l->rowval[j] == 0) continue; /* empty head slot: the column is not in this buffer */
            for (i64 k = j; k > 0; k = l->colptr[k]) {
                I[at] = l->rowval[k];
                J[at] = j;
                V[at] = l->nzval[k];
                at++;
            }
        }
    }
    orc_csc *c = orc_sparse_coo(N, N, at, I, J, V);
    double t2 = now_s();
    i64 z = c ? orc_csc_nnz(c) : -1;
    orc_csc_free(c);
    free(I);
    free(J);
    free(V);
    for (i64 p = 0; p < np; p++) orc_lnk_free(jobs[p].x);
    free(jobs);
    free(th);
    if (t_insert_s) *t_insert_s = t1 - t0;
    if (t_merge_s) *t_merge_s = t2 - t1;
    return z;
}

/* ------------------------------------------------------------------- FEM */
/* Update pattern of testassemble! (test/femtools.jl:45-72): per simplex,
 * for il: rawupdateindex!(A,+,0.1*vol/(dim+1),i,i); for jl:
 * rawupdateindex!(A,+,vol*S[il,jl],i,j).  The grid (ExtendableGrids
 * simplexgrid) is an external package: connectivity is parity-unpinned and
 * generated here as a Kuhn triangulation of a tensor grid.     */
i64 orc_fem_nnodes(int dim, i64 npd) { return dim == 2 ? npd * npd : npd * npd * npd; }
i64 orc_fem_ncells(int dim, i64 npd) {
    i64 q = npd - 1;
    return dim == 2 ? 2 * q * q : 6 * q * q * q;
}
i64 orc_fem_count(int dim, i64 npd) { return orc_fem_ncells(dim, npd) * (dim + 1) * (dim + 2); }

static const int kuhn3[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
static const int kuhn2[2][2] = {{0, 1}, {1, 0}};

/* vertex lattice coordinates of cell `cell` (0-based) */
static void fem_cell_lattice(int dim, i64 npd, i64 cell, i64 vx[4][3]) {
    i64 q = npd - 1;
    int K = dim == 2 ? 2 : 6;
    i64 cube = cell / K;
    int s = (int)(cell % K);
    i64 b[3];
    b[0] = cube % q;
    b[1] = (cube / q) % q;
    b[2] = dim == 3 ? cube / (q * q) : 0;
    for (int d = 0; d < 3; d++) vx[0][d] = b[d];
    for (int k = 0; k < dim; k++) {
        int ax = dim == 2 ? kuhn2[s][k] : kuhn3[s][k];
        for (int d = 0; d < 3; d++) vx[k + 1][d] = vx[k][d];
        vx[k + 1][ax] += 1;
    }
}
void orc_fem_cell_nodes(int dim, i64 npd, i64 cell, i64 *nodes) {
    i64 vx[4][3];
    fem_cell_lattice(dim, npd, cell, vx);
    for (int k = 0; k <= dim; k++)
        nodes[k] = 1 + vx[k][0] + npd * (vx[k][1] + npd * vx[k][2]);
}

/* position -> cell permutation: 4-round Feistel on an even number of bits
 * with cycle walking (a bijection on [0,ncells)); order_mode 0 = identity. */
uint64_t orc_fem_cell_at(i64 pos, i64 ncells, uint64_t seed, int order_mode) {
    if (order_mode == 0 || ncells < 2) return (uint64_t)pos;
    int bits = 2;
    while (((uint64_t)1 << bits) < (uint64_t)ncells) bits += 2;
    int half = bits / 2;
    uint64_t mask = ((uint64_t)1 << half) - 1;
    uint64_t x = (uint64_t)pos;
    do {
        uint64_t L = x >> half, R = x & mask;
        for (int r = 0; r < 4; r++) {
            uint64_t f = mix64(R + seed + (uint64_t)(r + 1) * 0x9E3779B97F4A7C15ull) & mask;
            uint64_t t = L ^ f;
            L = R;
            R = t;
        }
        x = (L << half) | R;
    } while (x >= (uint64_t)ncells);
    return x;
}

/* local matrices: barycentric gradients from the edge matrix by cofactors
 * (fixed operation order, no pivoting, no FMA) -- same sequence in csrc/.  */
static void fem_local(int dim, i64 npd, i64 cell, i64 *nodes, double *vol_out, double S[4][4]) {
    i64 vx[4][3];
    fem_cell_lattice(dim, npd, cell, vx);
    double h = 1.0 / (double)(npd - 1);
    double X[4][3] = {{0.0}};
    for (int k = 0; k <= dim; k++) {
        nodes[k] = 1 + vx[k][0] + npd * (vx[k][1] + npd * vx[k][2]);
        for (int d = 0; d < 3; d++) X[k][d] = (double)vx[k][d] * h;
    }
    double G[4][3] = {{0}};
    double det;
    if (dim == 2) {
        double a = X[1][0] - X[0][0], b = X[2][0] - X[0][0]; /* J = [e1 e2], rows x,y */
        double c = X[1][1] - X[0][1], d = X[2][1] - X[0][1];
        det = a * d - b * c;
        G[1][0] = d / det;
        G[1][1] = -b / det;
        G[2][0] = -c / det;
        G[2][1] = a / det;
        G[0][0] = -(G[1][0] + G[2][0]);
        G[0][1] = -(G[1][1] + G[2][1]);
    } else {
        double J[3][3];
        for (int r = 0; r < 3; r++)
            for (int k = 0; k < 3; k++) J[r][k] = X[k + 1][r] - X[0][r];
        double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
        double c01 = J[1][0] * J[2][2] - J[1][2] * J[2][0];
        double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
        det = J[0][0] * c00 - J[0][1] * c01 + J[0][2] * c02;
        /* inverse = adj/det ; row k of the inverse = gradient of lambda_{k+1} */
        G[1][0] = c00 / det;
        G[1][1] = -(J[0][1] * J[2][2] - J[0][2] * J[2][1]) / det;
        G[1][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) / det;
        G[2][0] = -c01 / det;
        G[2][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) / det;
        G[2][2] = -(J[0][0] * J[1][2] - J[0][2] * J[1][0]) / det;
        G[3][0] = c02 / det;
        G[3][1] = -(J[0][0] * J[2][1] - J[0][1] * J[2][0]) / det;
        G[3][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) / det;
        for (int d = 0; d < 3; d++) G[0][d] = -((G[1][d] + G[2][d]) + G[3][d]);
    }
    *vol_out = fabs(det) / (dim == 2 ? 2.0 : 6.0); /* femtools.jl:23 */
    for (int il = 0; il <= dim; il++)               /* stiffness!: femtools.jl:34-43 */
        for (int jl = il; jl <= dim; jl++) {
            double s = 0.0;
            for (int k = 0; k < dim; k++) s += G[jl][k] * G[il][k];
            S[il][jl] = s;
            S[jl][il] = s;
        }
}

/* stream positions [p0, p1) in cells (the whole stream in chunks: bench-size digests) */
void orc_fem_stream_range(int dim, i64 npd, uint64_t seed, int order_mode, i64 p0, i64 p1, i64 *I, i64 *J,
                          double *V) {
    i64 nc = orc_fem_ncells(dim, npd);
    i64 pos = 0;
    for (i64 p = p0; p < p1; p++) {
        i64 cell = (i64)orc_fem_cell_at(p, nc, seed, order_mode);
        i64 nodes[4];
        double vol, S[4][4];
        fem_local(dim, npd, cell, nodes, &vol, S);
        for (int il = 0; il <= dim; il++) { /* femtools.jl:62-69 */
            I[pos] = nodes[il];
            J[pos] = nodes[il];
            V[pos] = 0.1 * vol / (double)(dim + 1);
            pos++;
            for (int jl = 0; jl <= dim; jl++) {
                I[pos] = nodes[il];
                J[pos] = nodes[jl];
                V[pos] = vol * S[il][jl];
                pos++;
            }
        }
    }
}

void orc_fem_stream(int dim, i64 npd, uint64_t seed, int order_mode, i64 *I, i64 *J, double *V) {
    orc_fem_stream_range(dim, npd, seed, order_mode, 0, orc_fem_ncells(dim, npd), I, J, V);
}

/* ---------------------------------------------------------- element-level assembly */
/* The arrays a caller of testassemble! (test/femtools.jl:45-72) holds for the cells at stream positions [p0, p1) of
 * the build's Kuhn grid: cellnodes (Int64 nloc x ncells, Julia layout -- grid[CellNodes], femtools.jl:47), and what
 * the loop body computes per cell before it updates the matrix: elmat[il,jl,c] = vol * S[il,jl] (femtools.jl:67),
 * diag[il,c] = 0.1 * vol / (dim+1) (femtools.jl:64).  node_mode 1: the nodes carry a permuted numbering (a Feistel
 * bijection of [0, nnodes) with node_seed), so that nothing downstream can lean on grid arithmetic.              */
void orc_fem_mesh_range(int dim, i64 npd, uint64_t seed, int order_mode, int node_mode, uint64_t node_seed, i64 p0,
                        i64 p1, i64 *cellnodes, double *elmat, double *diag) {
    i64 nc = orc_fem_ncells(dim, npd), nn = orc_fem_nnodes(dim, npd);
    int nloc = dim + 1;
    for (i64 p = p0; p < p1; p++) {
        i64 cell = (i64)orc_fem_cell_at(p, nc, seed, order_mode);
        i64 nodes[4];
        double vol, S[4][4];
        fem_local(dim, npd, cell, nodes, &vol, S);
        i64 q = p - p0;
        for (int il = 0; il < nloc; il++) {
            i64 nd = nodes[il];
            if (node_mode) nd = 1 + (i64)orc_fem_cell_at(nd - 1, nn, node_seed, 1);
            cellnodes[q * nloc + il] = nd;
            if (diag) diag[q * nloc + il] = 0.1 * vol / (double)(dim + 1);
            for (int jl = 0; jl < nloc; jl++) elmat[(q * nloc + jl) * nloc + il] = vol * S[il][jl];
        }
    }
}

/* The update calls of the assembly loop (test/femtools.jl:62-69) for element data held in arrays:
 *   for icell: for il: i = cellnodes[il,icell]; [update(A, diag[il,icell], i, i);]
 *                      for jl: j = cellnodes[jl,icell]; update(A, elmat[il,jl,icell], i, j)
 * as a triplet stream in call order; returns the number of triplets = ncells * nloc * (nloc + (diag != NULL)).   */
i64 orc_elements_stream(int nloc, i64 ncells, const i64 *cellnodes, const double *elmat, const double *diag, i64 *I,
                        i64 *J, double *V) {
    i64 pos = 0;
    for (i64 c = 0; c < ncells; c++)
        for (int il = 0; il < nloc; il++) {
            i64 i = cellnodes[c * nloc + il];
            if (diag) {
                I[pos] = i;
                J[pos] = i;
                V[pos] = diag[c * nloc + il];
                pos++;
            }
            for (int jl = 0; jl < nloc; jl++) {
                I[pos] = i;
                J[pos] = cellnodes[c * nloc + jl];
                V[pos] = elmat[(c * nloc + jl) * nloc + il];
                pos++;
            }
        }
    return pos;
}
