/*
 * abi_host.c -- a plain C99 host of libesparse_hip.so (TEST INFRASTRUCTURE).
 *
 * Julia is not in this image, so extendablesparse.jl_amd/julia/ESparseHIP.jl has never run.  This program is the
 * closest executable stand-in: it drives the C ABI of include/esparse_hip.h with exactly the call sequences the shim
 * makes, from a language that is not Python and through the header itself (gcc -std=c99 -pedantic), and checks every
 * result against an in-order accumulation done here in C (the semantics of test/test_assembly.jl:19-32):
 *
 *   A. esp_stage_begin / esp_commit chunk loop (ESparseHIP.jl: push! into the pinned chunk, one ccall per chunk) of a
 *      7-point stencil stream with duplicates, flush!(ROUTED), esp_get_csc into caller-owned arrays, a second assembly
 *      over the stored pattern (hits) with a few new positions;
 *   B. the Generic wrapper's buffer-per-flush life cycle (genericextendablesparsematrixcsc.jl:31-37): esp_set_csc ->
 *      esp_flush(PLUS) -> esp_get_csc -> esp_release_buffers, 200 times on ONE handle and 20 times with a fresh handle,
 *      with the device's free memory (hipMemGetInfo, resolved with dlsym: no HIP headers here) returning to its level;
 *   C. the group calls in the order of INTEGRATION.md section 3 (single rank, the library's own RCCL transport):
 *      esp_group_unique_id -> esp_group_create -> appends -> esp_group_flush -> esp_group_nnz -> esp_group_get_csc.
 *   D. flush! of the MT wrapper as the shim runs it (genericmtextendablesparsematrixcsc.jl:45-51 = Base.sum(xmatrices, csc)):
 *      p partition buffers filled through their staged chunks, ONE esp_flush_sum into the handle that keeps the CSC,
 *      esp_get_csc when the pattern changed, else esp_get_nzval; then nonzeros(A) .= 0 on the host (test_parallel.jl:71-92),
 *      esp_set_nzval (values only) and the same assembly again: every update now meets a stored position;
 *   E. esp_append_elements_host: the loops of test/femtools.jl:61-69 over cellnodes / elmat / diag arrays, checked against
 *      the per-entry rawupdateindex! calls in the same order.
 *   F. the field contract of HIPResidentSparseMatrixCSC (ESparseHIP.jl: host_csc! / push_edits!; the reference's consumers read
 *      A.cscmatrix right after flush!: factorizations/ilu0.jl:126-136, umfpack_lu.jl:18-27): flush!(ROUTED) -> esp_get_csc ->
 *      the host edits nzval (nonzeros(A) .= 0.5 on half the entries) -> esp_set_nzval in front of the next update -> updates
 *      that hit stored positions -> flush! (pattern_changed == 0) -> esp_get_nzval INTO the same vector; then a new position
 *      -> pattern_changed == 1 -> esp_get_csc again; the same matrix through the Int32 transfers (esp_append_host_i32,
 *      esp_get_csc_i32, esp_set_csc_i32 on a second handle -> esp_get_csc must return the Int64 arrays).
 *
 * Exit code 0 and "abi_host: ok" on success; any mismatch prints what differs and exits 1.
 * Built and run by tests/test_abi_host.py (compile-only without a GPU).
 */
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "esparse_hip.h"

#define CHECK(h, call)                                                                                   \
    do {                                                                                                 \
        int32_t rc_ = (call);                                                                            \
        if (rc_ != ESP_OK) {                                                                             \
            fprintf(stderr, "abi_host: %s -> %d (%s) at line %d\n", #call, (int)rc_, esp_last_error(h), __LINE__); \
            exit(1);                                                                                     \
        }                                                                                                \
    } while (0)
#define REQUIRE(cond, ...)                                      \
    do {                                                        \
        if (!(cond)) {                                          \
            fprintf(stderr, "abi_host: line %d: ", __LINE__);   \
            fprintf(stderr, __VA_ARGS__);                       \
            fprintf(stderr, "\n");                              \
            exit(1);                                            \
        }                                                       \
    } while (0)

/* ---- a tiny host model: dense accumulation in call order (n is small) ------------------------------------------- */
typedef struct {
    int64_t n;
    double *val;           /* n x n, column-major */
    unsigned char *present;
} dense_t;
static dense_t dense_new(int64_t n) {
    dense_t d;
    d.n = n;
    d.val = (double *)calloc((size_t)(n * n), sizeof(double));
    d.present = (unsigned char *)calloc((size_t)(n * n), 1);
    if (!d.val || !d.present) {
        fprintf(stderr, "abi_host: out of memory\n");
        exit(1);
    }
    return d;
}
static void dense_free(dense_t *d) {
    free(d->val);
    free(d->present);
}
/* updateindex!(A, +, v, i, j): sparsematrixlnk.jl:210-228 (nothing is created by a zero) */
static void dense_update(dense_t *d, int64_t i, int64_t j, double v) {
    size_t at = (size_t)((j - 1) * d->n + (i - 1));
    if (!d->present[at]) {
        if (v == 0.0) return;
        d->present[at] = 1;
        d->val[at] = 0.0 + v;
    } else {
        d->val[at] = d->val[at] + v;
    }
}
/* csc + buffer (sparsematrixlnk.jl:363): the buffer's folded value meets the stored one, csc operand first */
static void dense_plus(dense_t *d, int64_t i, int64_t j, double bufval) {
    size_t at = (size_t)((j - 1) * d->n + (i - 1));
    if (d->present[at]) {
        d->val[at] = d->val[at] + bufval;
    } else {
        d->present[at] = 1;
        d->val[at] = bufval;
    }
}
/* rawupdateindex!(A, +, v, i, j): sparsematrixlnk.jl:237-253 (always creates) */
static void dense_raw(dense_t *d, int64_t i, int64_t j, double v) {
    size_t at = (size_t)((j - 1) * d->n + (i - 1));
    if (!d->present[at]) {
        d->present[at] = 1;
        d->val[at] = 0.0 + v;
    } else {
        d->val[at] = d->val[at] + v;
    }
}
static int64_t dense_nnz(const dense_t *d) {
    int64_t z = 0, k;
    for (k = 0; k < d->n * d->n; k++) z += d->present[k];
    return z;
}
static void compare_csc(const dense_t *d, const int64_t *colptr, const int64_t *rowval, const double *nzval, int64_t nnz,
                        const char *what) {
    int64_t j, k, i;
    REQUIRE(colptr[0] == 1, "%s: colptr[1] = %lld", what, (long long)colptr[0]);
    REQUIRE(colptr[d->n] == nnz + 1, "%s: colptr[n+1] = %lld, nnz = %lld", what, (long long)colptr[d->n], (long long)nnz);
    REQUIRE(nnz == dense_nnz(d), "%s: nnz %lld, expected %lld", what, (long long)nnz, (long long)dense_nnz(d));
    for (j = 1; j <= d->n; j++) {
        int64_t seen = 0;
        REQUIRE(colptr[j] >= colptr[j - 1], "%s: colptr decreases at column %lld", what, (long long)j);
        for (k = colptr[j - 1]; k < colptr[j]; k++) {
            size_t at;
            i = rowval[k - 1];
            REQUIRE(i >= 1 && i <= d->n, "%s: row %lld in column %lld", what, (long long)i, (long long)j);
            if (k > colptr[j - 1]) REQUIRE(rowval[k - 2] < i, "%s: rows of column %lld not strictly increasing", what, (long long)j);
            at = (size_t)((j - 1) * d->n + (i - 1));
            REQUIRE(d->present[at], "%s: (%lld,%lld) is stored but was never updated", what, (long long)i, (long long)j);
            REQUIRE(memcmp(&d->val[at], &nzval[k - 1], sizeof(double)) == 0, "%s: value at (%lld,%lld): %.17g, expected %.17g", what,
                    (long long)i, (long long)j, nzval[k - 1], d->val[at]);
            seen++;
        }
        for (i = 1; i <= d->n; i++) seen -= d->present[(size_t)((j - 1) * d->n + (i - 1))];
        REQUIRE(seen == 0, "%s: column %lld misses entries", what, (long long)j);
    }
}

/* ---- the staged chunk of the shim ------------------------------------------------------------------------------- */
typedef struct {
    esp_handle *h;
    int64_t *rows, *cols;
    double *vals;
    uint8_t *kinds;
    int64_t cap, fill;
} stage_t;
static void stage_open(stage_t *s, esp_handle *h) {
    s->h = h;
    s->fill = 0;
    CHECK(h, esp_stage_begin(h, 1 << 12, &s->rows, &s->cols, &s->vals, &s->kinds, &s->cap));
    REQUIRE(s->cap >= 1, "stage capacity %lld", (long long)s->cap);
}
static void stage_commit(stage_t *s) {
    if (s->fill) CHECK(s->h, esp_commit(s->h, s->fill, -1, ESP_OP_ADD));
    s->fill = 0;
}
static void stage_push(stage_t *s, int kind, double v, int64_t i, int64_t j) {
    s->rows[s->fill] = i;
    s->cols[s->fill] = j;
    s->vals[s->fill] = v;
    s->kinds[s->fill] = (uint8_t)kind;
    if (++s->fill == s->cap) stage_commit(s);
}

static double lcg(uint64_t *st) {
    *st = *st * 6364136223846793005ull + 1442695040888963407ull;
    return (double)(*st >> 11) * (1.0 / 9007199254740992.0);
}

/* the 7-point stencil stream of fdrand! (sprand.jl:87-124) on an q x q x q grid, values from the generator above */
static void stencil_stream(int q, uint64_t seed, stage_t *st, dense_t *d) {
    int64_t i, j, k, dirs[3];
    uint64_t rng = seed;
    dirs[0] = 1;
    dirs[1] = q;
    dirs[2] = (int64_t)q * q;
    for (k = 0; k < q; k++)
        for (j = 0; j < q; j++)
            for (i = 0; i < q; i++) {
                int64_t l = 1 + i + q * (j + (int64_t)q * k), c[3];
                int a;
                c[0] = i, c[1] = j, c[2] = k;
                for (a = 0; a < 3; a++) {
                    if (c[a] < q - 1) { /* update_pair: sprand.jl:87-92 */
                        double v = 0.1 + lcg(&rng);
                        int64_t l2 = l + dirs[a];
                        stage_push(st, ESP_UPDATE, -v, l, l2), dense_update(d, l, l2, -v);
                        stage_push(st, ESP_UPDATE, -v, l2, l), dense_update(d, l2, l, -v);
                        stage_push(st, ESP_UPDATE, v, l, l), dense_update(d, l, l, v);
                        stage_push(st, ESP_UPDATE, v, l2, l2), dense_update(d, l2, l2, v);
                    }
                    if (c[a] == 0 || c[a] == q - 1) {
                        double v = 0.1 + lcg(&rng);
                        stage_push(st, ESP_UPDATE, v, l, l), dense_update(d, l, l, v);
                    }
                }
            }
}

typedef int (*mem_info_fn)(size_t *, size_t *);
static size_t device_free_bytes(mem_info_fn f) {
    size_t fr = 0, tot = 0;
    if (!f || f(&fr, &tot) != 0) return 0;
    return fr;
}

int main(void) {
    const int q = 9; /* 729 unknowns: the dense model stays small */
    const int64_t n = (int64_t)q * q * q;
    esp_handle *h = NULL;
    int64_t nnz = 0, *colptr, *rowval;
    int32_t changed = 0;
    double *nzval;
    dense_t D = dense_new(n);
    stage_t st;
    int64_t m_, n_;
    int rep;

    /* ---------------------------------------------------------------- A: staged chunks, flush!, get_csc, re-assembly */
    CHECK(NULL, esp_create(n, n, 0, 0, &h));
    CHECK(h, esp_size(h, &m_, &n_));
    REQUIRE(m_ == n && n_ == n, "esp_size");
    stage_open(&st, h);
    stencil_stream(q, 1u, &st, &D);
    stage_commit(&st);
    CHECK(h, esp_pending(h, &nnz));
    REQUIRE(nnz > 0, "pending entries after the commits");
    CHECK(h, esp_flush(h, ESP_FLUSH_ROUTED, &nnz, &changed));
    REQUIRE(changed == 1, "the first flush! builds the pattern");
    colptr = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
    rowval = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nnz + 64));
    nzval = (double *)malloc(sizeof(double) * (size_t)(nnz + 64));
    CHECK(h, esp_get_csc(h, colptr, rowval, nzval));
    compare_csc(&D, colptr, rowval, nzval, nnz, "A: fresh assembly");
    /* second assembly: every update hits the stored pattern (applied in place, no rebuild) ... */
    stencil_stream(q, 2u, &st, &D);
    stage_commit(&st);
    CHECK(h, esp_flush(h, ESP_FLUSH_ROUTED, &nnz, &changed));
    REQUIRE(changed == 0, "a re-assembly of hits must not rebuild the pattern");
    CHECK(h, esp_get_csc(h, colptr, rowval, nzval));
    compare_csc(&D, colptr, rowval, nzval, nnz, "A: re-assembly");
    /* ... then a few new couplings (and a zero update, which creates nothing) */
    stage_push(&st, ESP_UPDATE, 2.5, 1, n), dense_update(&D, 1, n, 2.5);
    stage_push(&st, ESP_UPDATE, -1.25, n, 1), dense_update(&D, n, 1, -1.25);
    stage_push(&st, ESP_UPDATE, 0.0, 2, n), dense_update(&D, 2, n, 0.0);
    stage_commit(&st);
    CHECK(h, esp_flush(h, ESP_FLUSH_ROUTED, &nnz, &changed));
    REQUIRE(changed == 1, "new positions rebuild the pattern");
    CHECK(h, esp_get_csc(h, colptr, rowval, nzval));
    compare_csc(&D, colptr, rowval, nzval, nnz, "A: new couplings");
    /* out-of-range index: BoundsError, nothing appended */
    stage_push(&st, ESP_UPDATE, 1.0, n + 1, 1);
    REQUIRE(esp_commit(h, st.fill, -1, ESP_OP_ADD) == ESP_ERR_BOUNDS, "an index outside the matrix must be refused");
    st.fill = 0;
    CHECK(h, esp_pending(h, &m_));
    REQUIRE(m_ == 0, "a refused chunk leaves nothing pending");

    /* ---------------------------------------------------------------- B: csc + buffer, 200 buffer life cycles */
    {
        void *hip = dlopen("libamdhip64.so", RTLD_NOW | RTLD_GLOBAL);
        mem_info_fn mem_info = (mem_info_fn)0;
        if (hip) { /* (POSIX's way around ISO C's missing object-to-function pointer conversion) */
            void *sym = dlsym(hip, "hipMemGetInfo");
            memcpy(&mem_info, &sym, sizeof sym);
        }
        size_t base_free = 0, low_free = (size_t)-1;
        esp_handle *x = NULL;
        int64_t z = 0;
        CHECK(NULL, esp_create(n, n, 0, 0, &x));
        for (rep = 0; rep < 200; rep++) {
            int64_t i = 1 + (rep * 37) % n, j = 1 + (rep * 101) % n;
            CHECK(x, esp_set_csc(x, colptr, rowval, nzval, nnz)); /* the wrapper's cscmatrix */
            stage_open(&st, x);
            stage_push(&st, ESP_RAWUPDATE, 1.0 + rep, i, j);
            stage_push(&st, ESP_RAWUPDATE, 0.5, i, j);
            dense_plus(&D, i, j, (0.0 + (1.0 + rep)) + 0.5); /* the buffer folds by itself first (rawupdateindex!: 0 + v, then + v) */
            stage_commit(&st);
            CHECK(x, esp_flush(x, ESP_FLUSH_PLUS, &z, &changed)); /* csc = x + csc */
            if (z > nnz) {
                rowval = (int64_t *)realloc(rowval, sizeof(int64_t) * (size_t)(z + 64));
                nzval = (double *)realloc(nzval, sizeof(double) * (size_t)(z + 64));
            }
            nnz = z;
            CHECK(x, esp_get_csc(x, colptr, rowval, nzval));
            CHECK(x, esp_release_buffers(x)); /* x = Tm(m,n): the old buffer is dropped NOW */
            if (rep == 3) base_free = device_free_bytes(mem_info);
            if (rep > 3) {
                size_t f = device_free_bytes(mem_info);
                if (f < low_free) low_free = f;
            }
        }
        compare_csc(&D, colptr, rowval, nzval, nnz, "B: 200 x (csc + buffer)");
        if (mem_info)
            REQUIRE(base_free - low_free < ((size_t)64 << 20), "device memory creeps: %zu bytes free after 3 cycles, %zu at the lowest",
                    base_free, low_free);
        CHECK(x, esp_destroy(x));
        for (rep = 0; rep < 20; rep++) { /* a fresh handle per flush (the shim before it reused one) */
            CHECK(NULL, esp_create(n, n, 0, 0, &x));
            CHECK(x, esp_set_csc(x, colptr, rowval, nzval, nnz));
            CHECK(x, esp_flush(x, ESP_FLUSH_PLUS, &z, &changed)); /* nothing pending: the flush! gate */
            REQUIRE(z == nnz && changed == 0, "an empty buffer leaves the CSC alone");
            CHECK(x, esp_destroy(x));
        }
        if (mem_info) {
            size_t f = device_free_bytes(mem_info);
            REQUIRE(base_free <= f + ((size_t)64 << 20), "handles leak device memory: %zu -> %zu bytes free", base_free, f);
        }
    }

    /* ---------------------------------------------------------------- D: Base.sum(xmatrices, csc) as one call */
    {
        enum { P = 5 };
        esp_handle *home = NULL, *xs[P];
        dense_t B[P];
        uint64_t rng = 77u;
        int64_t z = 0, k;
        int round, t;
        CHECK(NULL, esp_create(n, n, 0, 0, &home));
        CHECK(home, esp_set_csc(home, colptr, rowval, nzval, nnz)); /* the wrapper's cscmatrix (state of section B: D must run before C, which reuses rowval / nzval) */
        for (t = 0; t < P; t++) CHECK(NULL, esp_create(n, n, 0, 0, &xs[t]));
        for (round = 0; round < 2; round++) {
            for (t = 0; t < P; t++) B[t] = dense_new(n);
            for (t = 0; t < P; t++) {
                stage_open(&st, xs[t]);
                for (k = 0; k < 400; k++) { /* rawupdateindex!(A, +, v, i, j, tid): positions overlap between the buffers */
                    int64_t i = 1 + (int64_t)(lcg(&rng) * 40.0) + 13 * t, j = 1 + (int64_t)(lcg(&rng) * 60.0);
                    double v = lcg(&rng) - 0.5;
                    stage_push(&st, ESP_RAWUPDATE, v, i, j);
                    dense_raw(&B[t], i, j, v);
                }
                stage_commit(&st);
            }
            /* sparse!(I, J, V, m, n, +) over (csc entries, buffer 1, buffer 2, ...): sparsematrixdilnkc.jl:397-435 */
            for (t = 0; t < P; t++) {
                int64_t i, j;
                for (j = 1; j <= n; j++)
                    for (i = 1; i <= n; i++)
                        if (B[t].present[(size_t)((j - 1) * n + (i - 1))]) dense_plus(&D, i, j, B[t].val[(size_t)((j - 1) * n + (i - 1))]);
                dense_free(&B[t]);
            }
            CHECK(home, esp_flush_sum(home, xs, P, &z, &changed));
            for (t = 0; t < P; t++) {
                int64_t pend = -1, zz = -1;
                CHECK(xs[t], esp_pending(xs[t], &pend));
                CHECK(xs[t], esp_nnz(xs[t], &zz));
                REQUIRE(pend == 0 && zz == 0, "esp_flush_sum hands buffer %d back empty", t);
            }
            if (changed) {
                REQUIRE(round == 0, "the second round repeats the positions of the first");
                if (z > nnz) {
                    rowval = (int64_t *)realloc(rowval, sizeof(int64_t) * (size_t)(z + 64));
                    nzval = (double *)realloc(nzval, sizeof(double) * (size_t)(z + 64));
                }
                nnz = z;
                CHECK(home, esp_get_csc(home, colptr, rowval, nzval));
            } else {
                REQUIRE(z == nnz, "an unchanged pattern keeps nnz");
                CHECK(home, esp_get_nzval(home, nzval)); /* values only */
            }
            compare_csc(&D, colptr, rowval, nzval, nnz, round == 0 ? "D: Base.sum, new positions" : "D: Base.sum over the stored pattern");
            if (round == 0) { /* nonzeros(A) .= 0 on the host copy, then values only travel back */
                for (k = 0; k < nnz; k++) nzval[k] = 0.0;
                for (k = 0; k < n * n; k++)
                    if (D.present[k]) D.val[k] = 0.0;
                CHECK(home, esp_set_nzval(home, nzval));
                rng = 77u; /* the same calls again: every position is stored now */
            }
        }
        for (t = 0; t < P; t++) CHECK(xs[t], esp_destroy(xs[t]));
        CHECK(home, esp_destroy(home));
    }

    /* ---------------------------------------------------------------- C: the group calls of INTEGRATION.md section 3 */
    {
        uint8_t id[128];
        esp_handle *hs = NULL, *back = NULL;
        esp_group *g = NULL;
        int64_t lo = 0, hi = 0, z = 0, gz = 0, before = 0, *cp2;
        dense_t E = dense_new(n);
        CHECK(NULL, esp_group_unique_id(id));
        CHECK(NULL, esp_create(n, n, 0, 0, &hs));
        CHECK(hs, esp_group_create(hs, 1, 0, id, &g));
        CHECK(hs, esp_group_handle(g, &back));
        REQUIRE(back == hs, "esp_group_handle");
        CHECK(hs, esp_group_column_range(g, &lo, &hi));
        REQUIRE(lo == 1 && hi == n, "column range of the only rank");
        stage_open(&st, hs);
        stencil_stream(q, 3u, &st, &E);
        stage_commit(&st);
        CHECK(hs, esp_group_flush(g, ESP_FLUSH_ROUTED, &z, &changed));
        CHECK(hs, esp_group_nnz(g, &gz, &before));
        REQUIRE(gz == z && before == 0, "global nnz of a single rank");
        cp2 = (int64_t *)malloc(sizeof(int64_t) * (size_t)(hi - lo + 2));
        if (z > nnz) {
            rowval = (int64_t *)realloc(rowval, sizeof(int64_t) * (size_t)(z + 64));
            nzval = (double *)realloc(nzval, sizeof(double) * (size_t)(z + 64));
        }
        CHECK(hs, esp_group_get_csc(g, cp2, rowval, nzval));
        compare_csc(&E, cp2, rowval, nzval, z, "C: group flush");
        free(cp2);
        CHECK(hs, esp_group_destroy(g));
        CHECK(hs, esp_destroy(hs));
        dense_free(&E);
    }

    /* ---------------------------------------------------------------- E: element-level append from host arrays */
    {
        const int64_t nc = n - q - 1; /* cell c (0-based): the nodes c+1, c+2, c+1+q */
        int64_t *cn = (int64_t *)malloc(sizeof(int64_t) * 3 * (size_t)nc), c, z = 0;
        double *em = (double *)malloc(sizeof(double) * 9 * (size_t)nc), *dg = (double *)malloc(sizeof(double) * 3 * (size_t)nc);
        uint64_t rng = 5u;
        esp_handle *e = NULL;
        dense_t E = dense_new(n);
        int il, jl;
        REQUIRE(cn && em && dg, "out of memory");
        for (c = 0; c < nc; c++) {
            cn[3 * c] = c + 1, cn[3 * c + 1] = c + 2, cn[3 * c + 2] = c + 1 + q;
            for (il = 0; il < 9; il++) em[9 * c + il] = lcg(&rng) - 0.5;
            for (il = 0; il < 3; il++) dg[3 * c + il] = lcg(&rng);
        }
        for (c = 0; c < nc; c++) /* test/femtools.jl:61-69 */
            for (il = 0; il < 3; il++) {
                int64_t i = cn[3 * c + il];
                dense_raw(&E, i, i, dg[3 * c + il]);
                for (jl = 0; jl < 3; jl++) dense_raw(&E, i, cn[3 * c + jl], em[9 * c + 3 * jl + il]);
            }
        CHECK(NULL, esp_create(n, n, 0, 0, &e));
        CHECK(e, esp_append_elements_host(e, 3, nc, cn, em, dg, ESP_RAWUPDATE, ESP_OP_ADD));
        CHECK(e, esp_flush(e, ESP_FLUSH_ROUTED, &z, &changed));
        if (z > nnz) {
            rowval = (int64_t *)realloc(rowval, sizeof(int64_t) * (size_t)(z + 64));
            nzval = (double *)realloc(nzval, sizeof(double) * (size_t)(z + 64));
        }
        CHECK(e, esp_get_csc(e, colptr, rowval, nzval));
        compare_csc(&E, colptr, rowval, nzval, z, "E: esp_append_elements_host");
        cn[3 * (nc / 2) + 1] = n + 1; /* BoundsError: nothing is appended */
        REQUIRE(esp_append_elements_host(e, 3, nc, cn, em, dg, ESP_RAWUPDATE, ESP_OP_ADD) == ESP_ERR_BOUNDS, "a node outside the matrix");
        CHECK(e, esp_pending(e, &z));
        REQUIRE(z == 0, "a refused batch leaves nothing pending");
        CHECK(e, esp_destroy(e));
        dense_free(&E);
        free(cn);
        free(em);
        free(dg);
    }

    /* ---------------------------------------------------------------- F: A.cscmatrix after flush!, host edits, Int32 transfers */
    {
        esp_handle *f = NULL, *f2 = NULL;
        dense_t E = dense_new(n);
        int64_t z = 0, z2 = 0, k, j;
        int64_t *cp = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1)), *rv, *cp2, *rv2;
        int32_t *cp32, *rv32, *I32, *J32;
        double *nz, *nz2, *V;
        uint64_t rng = 11u;
        const int64_t T = 5 * n;
        CHECK(NULL, esp_create(n, n, 0, 0, &f));
        stage_open(&st, f);
        stencil_stream(q, 7u, &st, &E);
        stage_commit(&st);
        CHECK(f, esp_flush(f, ESP_FLUSH_ROUTED, &z, &changed)); /* flush!(A) */
        REQUIRE(changed == 1, "F: the first flush! builds the pattern");
        rv = (int64_t *)malloc(sizeof(int64_t) * (size_t)(z + 8));
        nz = (double *)malloc(sizeof(double) * (size_t)(z + 8));
        REQUIRE(cp && rv && nz, "out of memory");
        CHECK(f, esp_get_csc(f, cp, rv, nz)); /* A.cscmatrix: state STALE -> the whole matrix */
        compare_csc(&E, cp, rv, nz, z, "F: A.cscmatrix after flush!");
        for (j = 1; j <= n; j++) /* the host edits values of the matrix it was handed */
            for (k = cp[j - 1]; k < cp[j]; k++)
                if (k & 1) nz[k - 1] = 0.5, E.val[(size_t)((j - 1) * n + (rv[k - 1] - 1))] = 0.5;
        CHECK(f, esp_set_nzval(f, nz)); /* push_edits!: in front of the next update */
        stage_open(&st, f);
        stencil_stream(q, 8u, &st, &E); /* every update meets a stored position */
        stage_commit(&st);
        CHECK(f, esp_flush(f, ESP_FLUSH_ROUTED, &z2, &changed));
        REQUIRE(changed == 0 && z2 == z, "F: a re-assembly over the stored pattern changes no position");
        CHECK(f, esp_get_nzval(f, nz)); /* A.cscmatrix: state VALUES_STALE -> nzval only, into the vector handed out before */
        compare_csc(&E, cp, rv, nz, z, "F: host edits + hits, values-only download");
        stage_open(&st, f);
        stage_push(&st, ESP_RAWUPDATE, 2.25, 1, n), dense_raw(&E, 1, n, 2.25); /* a new position */
        stage_commit(&st);
        CHECK(f, esp_flush(f, ESP_FLUSH_ROUTED, &z2, &changed));
        REQUIRE(changed == 1 && z2 == z + 1, "F: a new position rebuilds the CSC");
        CHECK(f, esp_get_csc(f, cp, rv, nz));
        compare_csc(&E, cp, rv, nz, z2, "F: A.cscmatrix after a flush! that added a position");
        /* Ti = Int32: COO constructor through esp_append_host_i32, A.cscmatrix through esp_get_csc_i32 */
        dense_free(&E);
        E = dense_new(n);
        I32 = (int32_t *)malloc(sizeof(int32_t) * (size_t)T), J32 = (int32_t *)malloc(sizeof(int32_t) * (size_t)T);
        V = (double *)malloc(sizeof(double) * (size_t)T);
        REQUIRE(I32 && J32 && V, "out of memory");
        for (k = 0; k < T; k++) {
            I32[k] = 1 + (int32_t)(lcg(&rng) * (double)n), J32[k] = 1 + (int32_t)(lcg(&rng) * (double)n), V[k] = lcg(&rng) - 0.5;
            dense_raw(&E, I32[k], J32[k], V[k]);
        }
        CHECK(f, esp_reset(f));
        CHECK(f, esp_append_host_i32(f, I32, J32, V, NULL, ESP_RAWUPDATE, ESP_OP_ADD, T));
        CHECK(f, esp_flush(f, ESP_FLUSH_ROUTED, &z, &changed));
        cp32 = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n + 1)), rv32 = (int32_t *)malloc(sizeof(int32_t) * (size_t)(z + 8));
        nz2 = (double *)malloc(sizeof(double) * (size_t)(z + 8));
        cp2 = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1)), rv2 = (int64_t *)malloc(sizeof(int64_t) * (size_t)(z + 8));
        REQUIRE(cp32 && rv32 && nz2 && cp2 && rv2, "out of memory");
        CHECK(f, esp_get_csc_i32(f, cp32, rv32, nz2));
        for (k = 0; k <= n; k++) cp2[k] = cp32[k];
        for (k = 0; k < z; k++) rv2[k] = rv32[k];
        compare_csc(&E, cp2, rv2, nz2, z, "F: esp_append_host_i32 + esp_get_csc_i32");
        CHECK(NULL, esp_create(n, n, 0, 0, &f2)); /* A.cscmatrix = B with an Int32 matrix */
        CHECK(f2, esp_set_csc_i32(f2, cp32, rv32, nz2, z));
        memset(cp2, 0, sizeof(int64_t) * (size_t)(n + 1));
        memset(rv2, 0, sizeof(int64_t) * (size_t)z);
        memset(nz2, 0, sizeof(double) * (size_t)z);
        CHECK(f2, esp_nnz(f2, &z2));
        REQUIRE(z2 == z, "F: esp_set_csc_i32 nnz");
        CHECK(f2, esp_get_csc(f2, cp2, rv2, nz2));
        compare_csc(&E, cp2, rv2, nz2, z, "F: esp_set_csc_i32 + esp_get_csc");
        CHECK(f2, esp_destroy(f2));
        CHECK(f, esp_destroy(f));
        dense_free(&E);
        free(cp), free(rv), free(nz), free(cp32), free(rv32), free(nz2), free(cp2), free(rv2), free(I32), free(J32), free(V);
    }

    CHECK(h, esp_destroy(h));
    free(colptr);
    free(rowval);
    free(nzval);
    dense_free(&D);
    printf("abi_host: ok (%s)\n", esp_version());
    return 0;
}
