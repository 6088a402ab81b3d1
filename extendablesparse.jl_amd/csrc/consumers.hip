// consumers.hip -- libesparse_hip: readers and editors of the assembled CSC (see internal.hpp for the map of the translation units)
#include "internal.hpp"


extern "C" int32_t esp_dropzeros(esp_handle *h, int64_t *new_nnz) {
    if (!h) return ESP_ERR_INVALID;
    (void)hipSetDevice(h->device);
    CK(fix_tail(h));
    const i64 Z = h->nnz;
    if (Z == 0) {
        if (new_nnz) *new_nnz = 0;
        return ESP_OK;
    }
    if (Z >= 0xFFFFFFF0ll) FAIL(h, ESP_ERR_UNSUPPORTED, "dropzeros: nnz too large");
    CK(ensure(h, h->vals2, sizeof(u32) * (size_t)(Z + 1)));
    u32 *flag = (u32 *)h->vals2.p;
    hipLaunchKernelGGL(espfold::nonzero_flags_k, dim3(grid_for(Z + 1, 256)), dim3(256), 0, h->stream, (const double *)h->nzval.p, Z, flag);
    int l = 0;
    CK(scan_inplace<u32, false>(h, flag, Z + 1, h->hist, &l));
    HIPCK(h, hipMemcpyAsync(h->pin_scalar, flag + Z, 4, hipMemcpyDeviceToHost, h->stream));
    HIPCK(h, hipStreamSynchronize(h->stream));
    const i64 Zk = (i64) * (u32 *)h->pin_scalar;
    if (Zk != Z) {
        CK(ensure(h, h->rowval2, sizeof(i64) * (size_t)std::max<i64>(Zk, 1)));
        CK(ensure(h, h->nzval2, sizeof(double) * (size_t)std::max<i64>(Zk, 1)));
        hipLaunchKernelGGL(espfold::dropzeros_compact_k, dim3(grid_for(Z, 256)), dim3(256), 0, h->stream, (const i64 *)h->rowval.p,
                           (const double *)h->nzval.p, Z, flag, (i64 *)h->rowval2.p, (double *)h->nzval2.p);
        CK(ensure(h, h->colend, sizeof(i64) * (size_t)(h->n + 1)));
        hipLaunchKernelGGL(espfold::dropzeros_colptr_k, dim3(grid_for(h->n + 1, 256)), dim3(256), 0, h->stream, (const i64 *)h->colptr.p,
                           h->n + 1, flag, (i64 *)h->colend.p);
        std::swap(h->colptr, h->colend);
        std::swap(h->rowval, h->rowval2);
        std::swap(h->nzval, h->nzval2);
        h->nnz = Zk;
        h->pattern_version++, h->values_version++;
    }
    if (new_nnz) *new_nnz = h->nnz;
    return ESP_OK;
}

extern "C" int32_t esp_getindex(esp_handle *h, int64_t i, int64_t j, double *value, int32_t *found) {
    if (!h || !value) return ESP_ERR_INVALID;
    if (!(1 <= i && i <= h->m && 1 <= j && j <= h->n)) FAIL(h, ESP_ERR_BOUNDS, "BoundsError: (%lld,%lld) outside %lld x %lld", (long long)i, (long long)j, (long long)h->m, (long long)h->n);
    (void)hipSetDevice(h->device);
    CK(fix_tail(h));
    CK(ensure(h, h->misc, 256));
    espfold::Csc c{(const i64 *)h->colptr.p, (const i64 *)h->rowval.p, (double *)h->nzval.p, h->nnz};
    double *d_out = (double *)h->misc.p + 8;
    hipLaunchKernelGGL(espfold::getindex_k, dim3(1), dim3(1), 0, h->stream, c, i - 1, j - 1, d_out);
    HIPCK(h, hipMemcpyAsync(h->pin_scalar, d_out, 16, hipMemcpyDeviceToHost, h->stream));
    HIPCK(h, hipStreamSynchronize(h->stream));
    const double *r = (const double *)h->pin_scalar;
    *value = r[0];
    if (found) *found = r[1] != 0.0;
    return ESP_OK;
}

// ---- getindex(buffer, i, j): the value the pending entries alone give position (i,j) ---------------------------
// SparseMatrixLNK's getindex (sparsematrixlnk.jl:151-171) returns what the inserts so far left at (i,j), zero if
// there is no entry.  The device buffer holds the calls themselves: the matching ones are collected (buffer position,
// kind, value), ordered by position -- the call order, also in a bucket-ordered batch -- and folded by the state
// machine of fold.hpp.  A slow path by design (one pass over the pending keys per call): GenericExtendableSparseMatrixCSC
// reaches it for reads of positions that are not in the CSC yet (genericextendablesparsematrixcsc.jl:60-69).
constexpr int PENDING_MATCH_CAP = 2048;
__global__ void pending_matches_k(const u64 *__restrict__ keys, const double *__restrict__ vals, i64 E, u64 target,
                                  unsigned long long *__restrict__ count, u64 *__restrict__ mpos, double *__restrict__ mval) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 p = (i64)blockIdx.x * blockDim.x + threadIdx.x; p < E; p += stride) {
        const u64 k = keys[p];
        if ((k >> ESP_TAG_BITS) == target) {
            const unsigned long long at = atomicAdd(count, 1ull);
            if (at < (unsigned long long)PENDING_MATCH_CAP) {
                mpos[at] = ((u64)p << ESP_TAG_BITS) | (k & ESP_TAG_MASK);
                mval[at] = vals[p];
            }
        }
    }
}
__global__ __launch_bounds__(256) void pending_fold_k(const unsigned long long *__restrict__ count, const u64 *__restrict__ mpos,
                                                      const double *__restrict__ mval, double *__restrict__ out) {
    __shared__ u64 spos[PENDING_MATCH_CAP];
    __shared__ double sval[PENDING_MATCH_CAP];
    const int n = (int)min(*count, (unsigned long long)PENDING_MATCH_CAP);
    for (int q = threadIdx.x; q < n; q += 256) {  // rank sort by buffer position (positions are distinct)
        const u64 me = mpos[q];
        int r = 0;
        for (int o = 0; o < n; o++) r += mpos[o] < me ? 1 : 0;
        spos[r] = me;
        sval[r] = mval[q];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        bool present = false;
        double acc = 0.0;
        for (int q = 0; q < n; q++) espfold::fold_step(present, acc, (u32)(spos[q] & ESP_TAG_MASK), sval[q]);
        out[0] = present ? acc : 0.0;
        out[1] = present ? 1.0 : 0.0;
    }
}
extern "C" int32_t esp_pending_getindex(esp_handle *h, int64_t i, int64_t j, double *value, int32_t *found) {
    if (!h || !value) return ESP_ERR_INVALID;
    if (!(1 <= i && i <= h->m && 1 <= j && j <= h->n)) FAIL(h, ESP_ERR_BOUNDS, "BoundsError: (%lld,%lld) outside %lld x %lld", (long long)i, (long long)j, (long long)h->m, (long long)h->n);
    *value = 0.0;
    if (found) *found = 0;
    if (h->count == 0) return ESP_OK;
    (void)hipSetDevice(h->device);
    if (h->part_assembled) FAIL(h, ESP_ERR_STATE, "esp_pending_getindex: the pending entries are spread over shard pieces (flush first)");
    CK(pending_materialize(h));  // (packed keys)
    const size_t bytes = 64 + (sizeof(u64) + sizeof(double)) * (size_t)PENDING_MATCH_CAP;
    CK(ensure(h, h->heads, bytes));
    unsigned long long *cnt = (unsigned long long *)h->heads.p;
    double *d_out = (double *)h->heads.p + 2;
    u64 *mpos = (u64 *)((char *)h->heads.p + 64);
    double *mval = (double *)(mpos + PENDING_MATCH_CAP);
    HIPCK(h, hipMemsetAsync(cnt, 0, 64, h->stream));
    const u64 target = ((u64)(j - 1) << h->L.rb) | (u64)(i - 1);
    const unsigned grid = (unsigned)std::min<i64>(4096, std::max<i64>(1, ceil_div<i64>(h->count, 256)));
    hipLaunchKernelGGL(pending_matches_k, dim3(grid), dim3(256), 0, h->stream, (const u64 *)h->keys.p, (const double *)h->vals.p, h->count,
                       target, cnt, mpos, mval);
    hipLaunchKernelGGL(pending_fold_k, dim3(1), dim3(256), 0, h->stream, (const unsigned long long *)cnt, (const u64 *)mpos,
                       (const double *)mval, d_out);
    HIPCK(h, hipGetLastError());
    HIPCK(h, hipMemcpyAsync(h->pin_scalar, cnt, 32, hipMemcpyDeviceToHost, h->stream));
    HIPCK(h, hipStreamSynchronize(h->stream));
    if (h->pin_scalar[0] > (unsigned long long)PENDING_MATCH_CAP)
        FAIL(h, ESP_ERR_UNSUPPORTED, "esp_pending_getindex: more than %d pending updates of (%lld,%lld); flush first", PENDING_MATCH_CAP, (long long)i, (long long)j);
    const double *r = (const double *)(h->pin_scalar + 2);
    *value = r[0];
    if (found) *found = r[1] != 0.0;
    return ESP_OK;
}

extern "C" int32_t esp_pattern_hash(esp_handle *h, uint64_t *hash) {
    if (!h || !hash) return ESP_ERR_INVALID;
    (void)hipSetDevice(h->device);
    CK(fix_tail(h));
    CK(ensure(h, h->misc, 256));
    unsigned long long *acc = (unsigned long long *)h->misc.p + 16;
    HIPCK(h, hipMemsetAsync(acc, 0, 16, h->stream));
    const i64 work = std::max<i64>(h->n + 1, h->nnz);
    const unsigned grid = (unsigned)std::min<i64>(2048, std::max<i64>(1, ceil_div<i64>(work, espfold::THREADS)));
    hipLaunchKernelGGL(espfold::pattern_hash_k, dim3(grid), dim3(espfold::THREADS), 0, h->stream, (const i64 *)h->colptr.p, h->n + 1,
                       (const i64 *)h->rowval.p, h->nnz, acc);
    HIPCK(h, hipMemcpyAsync(h->pin_scalar, acc, 16, hipMemcpyDeviceToHost, h->stream));
    HIPCK(h, hipStreamSynchronize(h->stream));
    const u64 h1 = h->pin_scalar[0], h2 = h->pin_scalar[1];
    *hash = esp_mix64(h1 ^ esp_mix64(h2 + 0xD1B54A32D192ED03ull));
    return ESP_OK;
}


// ---- mul!(r, A, x) on the device CSC -------------------------------------------------------------
// LinearAlgebra.mul!(r, ext, x) (abstractextendablesparsematrixcsc.jl:179-181 -> SparseArrays; the
// coloured loop of genericmtextendablesparsematrixcsc.jl:124-143 visits the columns in the same
// order): r .= 0, then column by column r[rows[i]] += vals[i]*x[col].  Every r[i] is therefore the
// left-to-right sum over its row's entries in increasing column order, products and sums rounded
// separately.  The device reproduces exactly that with a row-wise view of the CSC: a stable sort of
// the entry indices by row (built once per pattern, values are gathered through it, so numeric
// re-assembly does not invalidate it) and one thread per row adding in column order.  No atomics:
// bit-identical to the reference loop.
__global__ void csr_keys_k(const i64 *__restrict__ colptr, const i64 *__restrict__ rowval, i64 n, u64 *__restrict__ key,
                           double *__restrict__ payload, u64 *__restrict__ colidx) {
    const i64 c = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    for (i64 p = colptr[c] - 1; p < colptr[c + 1] - 1; p++) {
        key[p] = (u64)(rowval[p] - 1) << ESP_TAG_BITS;
        payload[p] = __longlong_as_double((long long)p);
        colidx[p] = (u64)c;
    }
}
__global__ void csr_finish_k(const u64 *__restrict__ skey, const double *__restrict__ spayload, const u64 *__restrict__ colidx, i64 Z,
                             u32 *__restrict__ perm, u32 *__restrict__ tcol, u64 *__restrict__ rowend) {
    const i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= Z) return;
    const u64 p = (u64)__double_as_longlong(spayload[k]);
    perm[k] = (u32)p;
    tcol[k] = (u32)colidx[p];
    const u64 row = skey[k] >> ESP_TAG_BITS;
    if (k == Z - 1 || (skey[k + 1] >> ESP_TAG_BITS) != row) rowend[row + 1] = (u64)(k + 1);
}
// rowptr0 = exclusive-max-scanned row ends shifted by one: entries of row i = [rowptr0[i], rowptr0[i+1])
// row-wise copy of the values (refreshed when nzval changed: one gather per assembly, then every product
// of a solver loop streams it)
__global__ void csr_values_k(const u32 *__restrict__ perm, const double *__restrict__ nzval, i64 Z, double *__restrict__ rval) {
    const i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < Z) rval[k] = nzval[perm[k]];
}
__global__ __launch_bounds__(256) void spmv_rows_k(const u64 *__restrict__ rowptr0, const double *__restrict__ rval,
                                                   const u32 *__restrict__ tcol, const double *__restrict__ x, i64 m,
                                                   double *__restrict__ r) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    double acc = 0.0;  // r .= zero(eltype)
    const u64 b = rowptr0[i + 1], e = rowptr0[i + 2];
    for (u64 k = b; k < e; k++) acc = acc + rval[k] * x[tcol[k]];
    r[i] = acc;
}

int32_t build_csr(esp_handle *h) {
    const i64 Z = h->nnz, m = h->m;
    const i64 M2 = m + 2;
    CK(ensure(h, h->csr_rowptr, sizeof(u64) * (size_t)(M2 + espscan::workspace_elems(M2))));
    u64 *rowptr = (u64 *)h->csr_rowptr.p;
    HIPCK(h, hipMemsetAsync(rowptr, 0, sizeof(u64) * (size_t)M2, h->stream));
    if (Z > 0) {
        if (Z >= 0xFFFFFFF0ll || h->n >= 0xFFFFFFF0ll) FAIL(h, ESP_ERR_UNSUPPORTED, "esp_mul: the row-wise index holds 32-bit positions and columns");
        CK(ensure(h, h->csr_perm, sizeof(u32) * (size_t)Z));
        CK(ensure(h, h->csr_col, sizeof(u32) * (size_t)Z));
        // scratch: keys A/B, payload A/B, colidx
        CK(ensure(h, h->csr_tmp, sizeof(u64) * (size_t)Z * 5));
        u64 *kA = (u64 *)h->csr_tmp.p, *kB = kA + Z;
        double *vA = (double *)(kB + Z), *vB = vA + Z;
        u64 *colidx = (u64 *)(vB + Z);
        hipLaunchKernelGGL(csr_keys_k, dim3(grid_for(h->n, 256)), dim3(256), 0, h->stream, (const i64 *)h->colptr.p, (const i64 *)h->rowval.p,
                           h->n, kA, vA, colidx);
        CK(ensure(h, h->segs, sizeof(i64) * 8));
        CK(ensure(h, h->misc, 256));
        i64 *segs = (i64 *)h->segs.p;
        const i64 T = ceil_div<i64>(Z, espradix::TILE);
        hipLaunchKernelGGL(set_i64_k, dim3(1), dim3(1), 0, h->stream, segs, (i64)0, Z, (i64)0, T);
        u64 *ki = kA, *ko = kB;
        double *vi = vA, *vo = vB;
        for (int done = 0; done < h->L.rb; done += 8) {  // stable LSD sort by row: columns stay ascending
            espradix::Pass p;
            p.keys_in = ki;
            p.vals_in = vi;
            p.keys_out = ko;
            p.vals_out = vo;
            p.seg_start = segs;
            p.tile_first = segs + 2;
            p.S = 1;
            p.owner_P = 0;
            p.owner_n = 1;
            p.colshift = 0;
            p.base = 0;
            p.span = ~0ull;
            p.err = (u32 *)h->misc.p + 62;
            p.shift = done;
            p.bits = std::min(8, h->L.rb - done);
            CK(partition_pass(h, p, T));
            std::swap(ki, ko);
            std::swap(vi, vo);
        }
        hipLaunchKernelGGL(csr_finish_k, dim3(grid_for(Z, 256)), dim3(256), 0, h->stream, (const u64 *)ki, (const double *)vi,
                           (const u64 *)colidx, Z, (u32 *)h->csr_perm.p, (u32 *)h->csr_col.p, rowptr);
    }
    // rowptr[i+1] holds the end of row i (0 for empty rows): running maximum = start of the next row
    espscan::exclusive<u64, true>(h->stream, rowptr, rowptr, M2, rowptr + M2);
    HIPCK(h, hipGetLastError());
    if (h->csr_tmp.p) {  // the scratch is 5 arrays of nnz: not worth keeping
        HIPCK(h, hipStreamSynchronize(h->stream));
        (void)hipFree(h->csr_tmp.p);
        h->csr_tmp = DevBuf{};
    }
    h->csr_version = h->pattern_version;
    return ESP_OK;
}

extern "C" int32_t esp_mul(esp_handle *h, const double *x, double *r, int32_t on_device) {
    if (!h || !x || !r) return ESP_ERR_INVALID;
    if (h->count != 0) FAIL(h, ESP_ERR_STATE, "esp_mul: pending entries (flush first, like mul!(r, ext, x) does)");
    (void)hipSetDevice(h->device);
    if (!h->csc_valid) CK(init_empty_csc(h));
    CK(fix_tail(h));
    if (h->csr_version != h->pattern_version) {
        CK(build_csr(h));
        h->csr_val_version = 0;
    }
    if (h->csr_val_version != h->values_version && h->nnz > 0) {
        CK(ensure(h, h->csr_val, sizeof(double) * (size_t)h->nnz));
        hipLaunchKernelGGL(csr_values_k, dim3(grid_for(h->nnz, 256)), dim3(256), 0, h->stream, (const u32 *)h->csr_perm.p,
                           (const double *)h->nzval.p, h->nnz, (double *)h->csr_val.p);
        h->csr_val_version = h->values_version;
    }
    const double *dx = x;
    double *dr = r;
    if (!on_device) {
        CK(ensure(h, h->mul_x, sizeof(double) * (size_t)std::max<i64>(h->n, 1)));
        CK(ensure(h, h->mul_r, sizeof(double) * (size_t)std::max<i64>(h->m, 1)));
        HIPCK(h, hipMemcpyAsync(h->mul_x.p, x, sizeof(double) * (size_t)h->n, hipMemcpyHostToDevice, h->stream));
        dx = (const double *)h->mul_x.p;
        dr = (double *)h->mul_r.p;
    }
    if (h->m > 0)
        hipLaunchKernelGGL(spmv_rows_k, dim3(grid_for(h->m, 256)), dim3(256), 0, h->stream, (const u64 *)h->csr_rowptr.p,
                           (const double *)h->csr_val.p, (const u32 *)h->csr_col.p, dx, h->m, dr);
    HIPCK(h, hipGetLastError());
    if (!on_device) HIPCK(h, hipMemcpyAsync(r, dr, sizeof(double) * (size_t)h->m, hipMemcpyDeviceToHost, h->stream));
    HIPCK(h, hipStreamSynchronize(h->stream));
    return ESP_OK;
}

// ---- Dirichlet edits of the assembled CSC (sparsematrixcsc.jl:97-140) --------------------------------
// one thread per column, the same statements as the reference loops (order inside a column kept)
__global__ void mark_dirichlet_k(const i64 *__restrict__ colptr, const i64 *__restrict__ rowval, const double *__restrict__ nzval,
                                 i64 n, double penalty, uint8_t *__restrict__ marker) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint8_t d = 0;
    for (i64 j = colptr[i] - 1; j < colptr[i + 1] - 1; j++)
        if (rowval[j] == i + 1 && nzval[j] >= penalty) d = 1;
    marker[i] = d;
}
__global__ void eliminate_dirichlet_k(const i64 *__restrict__ colptr, const i64 *__restrict__ rowval, double *__restrict__ nzval, i64 n,
                                      const uint8_t *__restrict__ marker) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool mine = marker[i] != 0;
    for (i64 j = colptr[i] - 1; j < colptr[i + 1] - 1; j++) {
        const i64 r = rowval[j] - 1;
        double v = nzval[j];
        if (mine) v = r == i ? 1.0 : 0.0;                 // A[:,i] = 0, A[i,i] = 1
        if (r != i && marker[r] != 0) v = 0.0;            // A[r,:] = 0 for a marked row r
        nzval[j] = v;
    }
}
int32_t dirichlet_call(esp_handle *h, uint8_t *marker, int32_t on_device, bool mark, double penalty) {
    if (!h || !marker) return ESP_ERR_INVALID;
    if (h->m != h->n) FAIL(h, ESP_ERR_INVALID, "dirichlet: the matrix must be square");
    if (h->count != 0) FAIL(h, ESP_ERR_STATE, "dirichlet: pending entries (flush first)");
    (void)hipSetDevice(h->device);
    if (!h->csc_valid) CK(init_empty_csc(h));
    CK(fix_tail(h));
    const i64 n = h->n;
    uint8_t *dm = marker;
    if (!on_device) {
        CK(ensure(h, h->mul_x, (size_t)std::max<i64>(n, 1)));
        dm = (uint8_t *)h->mul_x.p;
        if (!mark) HIPCK(h, hipMemcpyAsync(dm, marker, (size_t)n, hipMemcpyHostToDevice, h->stream));
    }
    if (n > 0) {
        if (mark)
            hipLaunchKernelGGL(mark_dirichlet_k, dim3(grid_for(n, 256)), dim3(256), 0, h->stream, (const i64 *)h->colptr.p,
                               (const i64 *)h->rowval.p, (const double *)h->nzval.p, n, penalty, dm);
        else
            hipLaunchKernelGGL(eliminate_dirichlet_k, dim3(grid_for(n, 256)), dim3(256), 0, h->stream, (const i64 *)h->colptr.p,
                               (const i64 *)h->rowval.p, (double *)h->nzval.p, n, (const uint8_t *)dm);
        if (!mark) h->values_version++;
    }
    HIPCK(h, hipGetLastError());
    if (!on_device && mark) HIPCK(h, hipMemcpyAsync(marker, dm, (size_t)n, hipMemcpyDeviceToHost, h->stream));
    HIPCK(h, hipStreamSynchronize(h->stream));
    return ESP_OK;
}
extern "C" int32_t esp_mark_dirichlet(esp_handle *h, double penalty, uint8_t *marker, int32_t on_device) {
    return dirichlet_call(h, marker, on_device, true, penalty);
}
extern "C" int32_t esp_eliminate_dirichlet(esp_handle *h, const uint8_t *marker, int32_t on_device) {
    return dirichlet_call(h, const_cast<uint8_t *>(marker), on_device, false, 0.0);
}

// ---- set-up of the point preconditioners on the device CSC -------------------------------------
// jacobi(A) (factorizations/jacobi.jl:5-12): invdiag[i] = one(Tv) / A[i,i]; getindex of a position that is not stored
// gives zero, i.e. Inf.  ilu0(A) (factorizations/ilu0.jl:8-41): idiag[j] = index of the diagonal entry of column j in
// rowval/nzval; xdiag: iteration j of the reference's loop first sets xdiag[j] = 1/nzval[idiag[j]] and then updates
// xdiag[i] for rows i > j -- every such update is overwritten when iteration i sets xdiag[i] itself, so the loop leaves
// xdiag[j] = 1/nzval[idiag[j]] (restated literally in oracle/esparse_oracle.c: orc_ilu0).  One thread per column.
__global__ void diag_setup_k(espfold::Csc c, i64 n, double *__restrict__ inv, i64 *__restrict__ idiag, unsigned long long *__restrict__ missing) {
    const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const i64 pos = c.nnz > 0 ? espfold::csc_find(c, j, j) : -1;
    if (idiag) {
        idiag[j] = pos + 1;
        if (pos < 0) atomicMin(missing, (unsigned long long)(j + 1));
    }
    inv[j] = 1.0 / (pos >= 0 ? c.nzval[pos] : 0.0);
}
int32_t diag_setup(esp_handle *h, double *inv, int64_t *idiag, int32_t on_device, const char *what) {
    if (!h || !inv) return ESP_ERR_INVALID;
    if (h->m != h->n) FAIL(h, ESP_ERR_INVALID, "%s: the matrix must be square", what);
    if (h->count != 0) FAIL(h, ESP_ERR_STATE, "%s: pending entries (flush first)", what);
    (void)hipSetDevice(h->device);
    if (!h->csc_valid) CK(init_empty_csc(h));
    CK(fix_tail(h));
    const i64 n = h->n;
    if (n == 0) return ESP_OK;
    double *d_inv = inv;
    i64 *d_idiag = idiag;
    if (!on_device) {
        CK(ensure(h, h->mul_x, sizeof(double) * (size_t)n));
        d_inv = (double *)h->mul_x.p;
        if (idiag) {
            CK(ensure(h, h->mul_r, sizeof(i64) * (size_t)n));
            d_idiag = (i64 *)h->mul_r.p;
        }
    }
    CK(ensure(h, h->misc, 256));
    unsigned long long *d_missing = (unsigned long long *)h->misc.p + 20;
    h->pin_scalar[0] = ~0ull;
    HIPCK(h, hipMemcpyAsync(d_missing, h->pin_scalar, 8, hipMemcpyHostToDevice, h->stream));
    espfold::Csc c{(const i64 *)h->colptr.p, (const i64 *)h->rowval.p, (double *)h->nzval.p, h->nnz};
    hipLaunchKernelGGL(diag_setup_k, dim3(grid_for(n, 256)), dim3(256), 0, h->stream, c, n, d_inv, d_idiag, d_missing);
    HIPCK(h, hipGetLastError());
    HIPCK(h, hipMemcpyAsync(h->pin_scalar, d_missing, 8, hipMemcpyDeviceToHost, h->stream));
    if (!on_device) {
        HIPCK(h, hipMemcpyAsync(inv, d_inv, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, h->stream));
        if (idiag) HIPCK(h, hipMemcpyAsync(idiag, d_idiag, sizeof(i64) * (size_t)n, hipMemcpyDeviceToHost, h->stream));
    }
    HIPCK(h, hipStreamSynchronize(h->stream));
    if (idiag && h->pin_scalar[0] != ~0ull)
        FAIL(h, ESP_ERR_INVALID, "%s: column %llu has no stored diagonal entry (the reference reads an undefined idiag there)", what,
             (unsigned long long)h->pin_scalar[0]);
    return ESP_OK;
}
extern "C" int32_t esp_jacobi_setup(esp_handle *h, double *invdiag, int32_t on_device) {
    return diag_setup(h, invdiag, nullptr, on_device, "esp_jacobi_setup");
}
extern "C" int32_t esp_ilu0_setup(esp_handle *h, double *xdiag, int64_t *idiag, int32_t on_device) {
    if (!idiag) return ESP_ERR_INVALID;
    return diag_setup(h, xdiag, idiag, on_device, "esp_ilu0_setup");
}

