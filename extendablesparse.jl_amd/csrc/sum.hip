// sum.hip -- libesparse_hip: Base.sum(buffers, csc) as ONE call (esp_flush_sum), and the values-only upload (esp_set_nzval)
// of a plug-in whose CSC stays attached to its handle between flushes
#include "internal.hpp"

#include <chrono>

namespace {

// the entries of a device CSC as COO-kind pending entries (column-major = the CSC's own order: a pre-sorted stream)
__global__ __launch_bounds__(256) void csc_as_coo_k(const i64 *__restrict__ colptr, const i64 *__restrict__ rowval, const double *__restrict__ nzval,
                                                    i64 n, KeyLayout L, u64 *__restrict__ keys, double *__restrict__ vals) {
    const i64 c = (i64)blockIdx.x * 256 + threadIdx.x;
    if (c >= n) return;
    const i64 a = colptr[c] - 1, b = colptr[c + 1] - 1;
    for (i64 e = a; e < b; e++) {
        keys[e] = esp_pack(L, rowval[e], c + 1, ESP_COO);
        vals[e] = nzval[e];
    }
}

}  // namespace

// Base.sum(xmatrices, csc) (sparsematrixdilnkc.jl:397-435; flush! of GenericMTExtendableSparseMatrixCSC,
// genericmtextendablesparsematrixcsc.jl:45-51).  The reference lists the CSC's entries, then every buffer's -- each buffer
// holds ONE value per position, the fold of the calls it received -- and hands the triplets to sparse!(I, J, V, m, n, +):
// the value at (i,j) is ((csc + f_1) + f_2) + ... over the buffers that hold (i,j), in buffer order, zeros kept.  Here:
//   1. every buffer with pending entries is flushed BY ITSELF (its own matrix is empty: the flush is its fold, f_k);
//   2. the folded entries of all buffers are appended to dst as COO entries, buffer after buffer -- p pre-sorted runs;
//   3. ONE routed flush of dst: a COO entry that meets a stored position is added to it in call order, the others fold
//      first-as-it-is-then-added (fold.hpp) -- the sum above, bit for bit -- and new positions are joined in.
// Everything stays on the device: no CSC travels, whatever p is.  The buffers come back empty.
extern "C" int32_t esp_flush_sum(esp_handle *dst, esp_handle *const *xs, int32_t p, int64_t *new_nnz, int32_t *pattern_changed) {
    if (!dst || p < 0 || (p > 0 && !xs)) return ESP_ERR_INVALID;
    if (pattern_changed) *pattern_changed = 0;
    if (dst->count != 0) FAIL(dst, ESP_ERR_STATE, "esp_flush_sum: the destination has pending entries of its own (esp_flush it first)");
    i64 total = 0;
    for (int k = 0; k < p; k++) {
        esp_handle *x = xs[k];
        if (!x || x == dst) FAIL(dst, ESP_ERR_INVALID, "esp_flush_sum: buffer %d is NULL or the destination itself", k);
        if (x->m != dst->m || x->n != dst->n || x->device != dst->device) FAIL(dst, ESP_ERR_INVALID, "esp_flush_sum: buffer %d has another size or device", k);
        if (x->nnz != 0) FAIL(dst, ESP_ERR_STATE, "esp_flush_sum: buffer %d holds a matrix of its own (a buffer is an empty matrix + pending entries)", k);
        for (int q = 0; q < k; q++)
            if (xs[q] == x) FAIL(dst, ESP_ERR_INVALID, "esp_flush_sum: buffer %d is listed twice", k);
        total += x->count;
    }
    (void)hipSetDevice(dst->device);
    if (total == 0) {
        if (new_nnz) *new_nnz = dst->nnz;
        return ESP_OK;
    }
    const bool trace = esp_exp_env("ESP_SUM_TRACE") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_a = now();
    double t_b = t_a, t_c = t_a;
    i64 folded = 0;
    // Steps 1 - 3; every exit goes through the cleanup below: the call is all-or-nothing for dst (a failure leaves its stored
    // matrix as it was and nothing pending) and the buffers come back EMPTY whatever happened.
    auto run = [&]() -> int32_t {
        // 1. every buffer's own fold: the buffers are independent handles with streams of their own, so their flushes -- a
        // few dozen small launches and three or four host round trips each -- run side by side, one host thread per buffer
        // (as the reference's partitions are assembled by one task each: test/femtools.jl:88-110)
        {
            std::vector<int32_t> rcs((size_t)p, ESP_OK);
            std::vector<int64_t> zs((size_t)p, 0);
            const int WIDTH = 16;  // host threads at a time
            for (int k0 = 0; k0 < p; k0 += WIDTH) {
                std::vector<std::thread> th;
                for (int k = k0; k < std::min(p, k0 + WIDTH); k++) {
                    if (xs[k]->count == 0) continue;
                    th.emplace_back([&, k] { rcs[(size_t)k] = esp_flush(xs[k], ESP_FLUSH_ROUTED, &zs[(size_t)k], nullptr); });
                }
                for (auto &t : th) t.join();
            }
            for (int k = 0; k < p; k++) {
                if (rcs[(size_t)k] != ESP_OK) {
                    dst->err = xs[k]->err;
                    return rcs[(size_t)k];
                }
                folded += zs[(size_t)k];
            }
        }
        t_b = now();
        // 2. their entries behind one another in dst's buffer (dst's stream waits for each buffer's flush: esp_flush returned
        // after its last host round trip, but kernels of the buffer's stream may still run)
        if (folded > 0) {
            CK(reserve_append(dst, folded));
            i64 at = dst->count;
            for (int k = 0; k < p; k++) {
                esp_handle *x = xs[k];
                if (x->nnz == 0) continue;
                CK(fix_tail(x));
                HIPCK(dst, hipStreamSynchronize(x->stream));
                Span sp(dst, ESP_ST_APPEND);
                hipLaunchKernelGGL(csc_as_coo_k, dim3(grid_for(x->n, 256)), dim3(256), 0, dst->stream, (const i64 *)x->colptr.p, (const i64 *)x->rowval.p,
                                   (const double *)x->nzval.p, x->n, dst->L, (u64 *)dst->keys.p + at, (double *)dst->vals.p + at);
                sp.add(1);
                at += x->nnz;
            }
            HIPCK(dst, hipGetLastError());
            note_kind(dst, ESP_COO, folded);
            dst->count += folded;
            pending_changed(dst);
        }
        // 3. the one flush that meets the stored matrix
        t_c = now();
        return esp_flush(dst, ESP_FLUSH_ROUTED, new_nnz, pattern_changed);
    };
    const int32_t rc = run();
    if (trace) {
        (void)hipStreamSynchronize(dst->stream);
        fprintf(stderr, "esp_flush_sum: %d buffers %lld entries -> folds %.3f ms (%lld entries), gather %.3f ms, combine flush %.3f ms\n", p, (long long)total,
                t_b - t_a, (long long)folded, t_c - t_b, now() - t_c);
    }
    // (the buffers are consumed whatever happened: genericmtextendablesparsematrixcsc.jl:47-49 replaces them all)
    (void)hipStreamSynchronize(dst->stream);
    if (rc != ESP_OK && dst->count != 0) {  // (a failure behind the gather: the folded entries must not stay pending in dst)
        const std::string msg = dst->err;
        (void)esp_clear_pending(dst);
        dst->err = msg;
    }
    for (int k = 0; k < p; k++) (void)esp_reset(xs[k]);
    return rc;
}

// nzval of the attached CSC := the caller's values (H2D of the values only: the pattern -- colptr, rowval -- is the one
// the handle holds since the caller's last esp_get_csc / esp_set_csc).  What a plug-in whose CSC stays on the device
// between flushes uploads instead of the whole matrix when only nonzeros(A) can have been edited on the host.
extern "C" int32_t esp_set_nzval(esp_handle *h, const double *nzval) {
    if (!h) return ESP_ERR_INVALID;
    if (h->nnz == 0) return ESP_OK;
    if (!nzval) return ESP_ERR_INVALID;
    (void)hipSetDevice(h->device);
    Span sp(h, ESP_ST_COPY);
    HIPCK(h, hipMemcpyAsync(h->nzval.p, nzval, sizeof(double) * (size_t)h->nnz, hipMemcpyHostToDevice, h->stream));
    HIPCK(h, hipStreamSynchronize(h->stream));
    sp.add(1);
    h->values_version++;
    return ESP_OK;
}
