// generators.hpp -- append-path kernels (K1): pack host/device (i,j,v) triples into the
// COO buffer, and on-device producers of the reference's update streams.
// All of them write the buffer in the exact order of the sequential reference loop, so
// the ordered fold reproduces the CPU accumulation order bit for bit.
// Compiled with -ffp-contract=off: no FMA contraction, values match the oracle's bits.
#pragma once
#include "common.hpp"
#include "runpart.hpp"

namespace espgen {

// Optional: the producer also emits the run list of its chunk (runpart.hpp), so that the flush can
// skip the histogram kernel (one full read of the keys).  runs.runs_d == nullptr switches it off.
struct Fused {
    esprun::RunSink runs;
    i64 *chunk_start;  // absolute buffer position of every chunk (+ the end of the last one)
    i64 chunk_base;    // index of this launch's first chunk
    i64 buf_base;      // buffer position of the first entry this launch writes
    int shift;         // digit = (((key >> 2) - base) >> shift)
    u64 base, span;
    u32 *err;
};

constexpr int THREADS = 256;

// (rows, cols, vals, kinds) -> packed keys.  Out-of-range indices raise *err (first bad
// position + 1) and nothing of the batch is committed by the host side
// (BoundsError, sparsematrixcsc.jl:8-10).
__global__ __launch_bounds__(THREADS) void pack_k(const i64 *__restrict__ rows,
                                                  const i64 *__restrict__ cols,
                                                  const double *__restrict__ vals,
                                                  const uint8_t *__restrict__ kinds, int kind_all,
                                                  int negate, i64 count, i64 m, i64 n, KeyLayout L,
                                                  u64 *__restrict__ keys, double *__restrict__ out,
                                                  unsigned long long *__restrict__ err, i64 index_base) {
    const i64 g = (i64)blockIdx.x * THREADS + threadIdx.x;
    if (g >= count) return;
    const i64 r = rows[g], c = cols[g];
    int kind = kinds ? (int)kinds[g] : kind_all;
    if (!(1 <= r && r <= m && 1 <= c && c <= n) || kind < 0 || kind > 3) {
        atomicMin(err, (unsigned long long)(index_base + g + 1));  // first offending entry of the whole batch
        return;
    }
    double v = vals[g];
    if (negate && kind != ESP_SET) v = -v;
    keys[g] = esp_pack(L, r, c, kind);
    out[g] = v;
}

// ---- fdrand! stream (src/matrix/sprand.jl:87-124) -----------------------------------
struct FdArgs {
    i64 nx, ny, nz;
    double hx, hy, hz;
    u64 seed;
    int rand_mode;
    int kind;
    i64 total;  // number of updates of the generated node range
    i64 g_begin, g_end;  // node range [g_begin, g_end) (0-based l-1) of the loop nest
    i64 off_begin;       // stream position of node g_begin
    Fused fused;
    KeyLayout L;
    u64 *keys;
    double *vals;
};

__device__ __forceinline__ double fd_rand(int mode, u64 seed, u64 ctr) {
    if (mode == 0) return 1.0;
    if (mode == 1) return 0.1 + esp_uniform(seed, ctr);
    return esp_uniform(seed, ctr);
}

// number of update calls issued by nodes that precede node (i,j,k) (1-based) in the
// k,j,i loop nest -- closed form, so every node writes at its exact stream position
__device__ __forceinline__ i64 fd_offset(const FdArgs &a, i64 i, i64 j, i64 k, i64 *cy_out,
                                         i64 *cz_out) {
    const i64 nx = a.nx, ny = a.ny, nz = a.nz;
    const i64 CX = 4 * (nx - 1) + (nx == 1 ? 1 : 2);
    const i64 CY = 4 * (ny - 1) + (ny > 2 ? 2 : 0);
    const i64 PX = 4 * (i - 1) + (i > 1 ? 1 : 0);
    const i64 PY = 4 * (j - 1) + ((ny > 2 && j > 1) ? 1 : 0);
    const i64 PZ = 4 * (k - 1) + ((nz > 2 && k > 1) ? 1 : 0);
    const i64 cy = (j < ny ? 4 : 0) + ((ny > 2 && (j == 1 || j == ny)) ? 1 : 0);
    const i64 cz = (k < nz ? 4 : 0) + ((nz > 2 && (k == 1 || k == nz)) ? 1 : 0);
    *cy_out = cy;
    *cz_out = cz;
    return (k - 1) * (ny * CX + nx * CY) + nx * ny * PZ + (j - 1) * CX + nx * PY +
           (j - 1) * nx * cz + PX + (i - 1) * (cy + cz);
}

constexpr int FD_MAX_PER_NODE = 15;  // 3 pairs x 4 + 3 boundary terms

__device__ __forceinline__ void fd_put(const FdArgs &a, u64 *lk, double *lv, int &o, double v, i64 row, i64 col) {
    lk[o] = esp_pack(a.L, row, col, a.kind);
    lv[o] = v;
    o++;
}
__device__ __forceinline__ void fd_pair(const FdArgs &a, u64 *lk, double *lv, int &o, double v, i64 l, i64 l2) {
    fd_put(a, lk, lv, o, -v, l, l2);  // update_pair: sprand.jl:87-92
    fd_put(a, lk, lv, o, -v, l2, l);
    fd_put(a, lk, lv, o, v, l, l);
    fd_put(a, lk, lv, o, v, l2, l2);
}

// One workgroup = 256 consecutive nodes.  Every node knows the exact position of its updates in
// the sequential stream (closed form), the workgroup's updates form one contiguous range of the
// buffer: they are staged in LDS and written with coalesced 16-byte stores (K1: "coalesced HBM
// stores on the append path").  Writes 16 B per update, reads nothing.
__global__ __launch_bounds__(THREADS) void fdrand_k(FdArgs a) {
    __shared__ u64 lk[THREADS * FD_MAX_PER_NODE];
    __shared__ double lv[THREADS * FD_MAX_PER_NODE];
    __shared__ u32 rd[esprun::RMAX];
    __shared__ u32 rc[esprun::RMAX];
    __shared__ u32 rover;
    if (threadIdx.x < esprun::RMAX) {
        rd[threadIdx.x] = esprun::EMPTY;
        rc[threadIdx.x] = 0;
    }
    if (threadIdx.x == 0) rover = 0;
    int my_first = 0, my_count = 0;
    const i64 N = a.g_end;
    const i64 g0 = a.g_begin + (i64)blockIdx.x * THREADS;
    const i64 g = g0 + threadIdx.x;  // node l-1
    i64 cy, cz;
    // stream position of the first node of this workgroup and of the next one
    const i64 i0 = g0 % a.nx + 1, j0 = (g0 / a.nx) % a.ny + 1, k0 = g0 / (a.nx * a.ny) + 1;
    const i64 off0 = fd_offset(a, i0, j0, k0, &cy, &cz) - a.off_begin;
    if (g < N) {
        const i64 i = g % a.nx + 1, j = (g / a.nx) % a.ny + 1, k = g / (a.nx * a.ny) + 1;
        int o = (int)(fd_offset(a, i, j, k, &cy, &cz) - a.off_begin - off0);
        my_first = o;
        const i64 l = g + 1;
        const u64 c = 6ull * (u64)g;
        if (i < a.nx) fd_pair(a, lk, lv, o, fd_rand(a.rand_mode, a.seed, c + 0) * a.hy * a.hz / a.hx, l, l + 1);
        if (i == 1 || i == a.nx) fd_put(a, lk, lv, o, fd_rand(a.rand_mode, a.seed, c + 1) * a.hy * a.hz, l, l);
        if (j < a.ny) fd_pair(a, lk, lv, o, fd_rand(a.rand_mode, a.seed, c + 2) * a.hx * a.hz / a.hy, l, l + a.nx);
        if (a.ny > 2 && (j == 1 || j == a.ny)) fd_put(a, lk, lv, o, fd_rand(a.rand_mode, a.seed, c + 3) * a.hx * a.hz, l, l);
        if (k < a.nz) fd_pair(a, lk, lv, o, fd_rand(a.rand_mode, a.seed, c + 4) * a.hx * a.hy / a.hz, l, l + a.nx * a.ny);
        if (a.nz > 2 && (k == 1 || k == a.nz)) fd_put(a, lk, lv, o, fd_rand(a.rand_mode, a.seed, c + 5) * a.hx * a.hy, l, l);
        my_count = o - my_first;
    }
    // total of the workgroup = position of the first node of the next workgroup (or the stream end)
    i64 off1;
    const i64 g1 = g0 + THREADS;
    if (g1 >= N) {
        off1 = a.total;
    } else {
        const i64 i1 = g1 % a.nx + 1, j1 = (g1 / a.nx) % a.ny + 1, k1 = g1 / (a.nx * a.ny) + 1;
        off1 = fd_offset(a, i1, j1, k1, &cy, &cz) - a.off_begin;
    }
    const int cnt = (int)(off1 - off0);
    __syncthreads();
    if (a.fused.runs.runs_d) {  // run list of this workgroup's chunk (every thread owns <= 15 entries)
        const Fused &f = a.fused;
        u32 dig[FD_MAX_PER_NODE];
        u32 pend = 0;
#pragma unroll
        for (int q = 0; q < FD_MAX_PER_NODE; q++) {
            const bool valid = q < my_count;
            dig[q] = valid ? esprun::run_digit(lk[my_first + q], f.base, f.span, f.shift, f.err) : 0u;
            pend |= valid ? (1u << q) : 0u;
        }
        const i64 chunk = f.chunk_base + blockIdx.x;
        if (threadIdx.x == 0) {
            f.chunk_start[chunk] = f.buf_base + off0;
            if (blockIdx.x == gridDim.x - 1) f.chunk_start[chunk + 1] = f.buf_base + off1;
        }
        esprun::count_runs<FD_MAX_PER_NODE>(dig, pend, chunk, f.runs, rd, rc, &rover);
    }
    // coalesced copy-out; 16-byte stores on the aligned body
    u64 *gk = a.keys + off0;
    double *gv = a.vals + off0;
    const int head = (int)(((uintptr_t)gk >> 3) & 1);  // first element not 16-byte aligned
    if (threadIdx.x == 0 && head && cnt > 0) {
        gk[0] = lk[0];
        gv[0] = lv[0];
    }
    const int npair = (cnt - head) >> 1;
    typedef unsigned long long ull2 __attribute__((ext_vector_type(2)));
    typedef double dbl2 __attribute__((ext_vector_type(2)));
    ull2 *gk2 = reinterpret_cast<ull2 *>(gk + head);
    dbl2 *gv2 = reinterpret_cast<dbl2 *>(gv + head);
    for (int q = threadIdx.x; q < npair; q += THREADS) {
        gk2[q] = ull2{lk[head + 2 * q], lk[head + 2 * q + 1]};
        gv2[q] = dbl2{lv[head + 2 * q], lv[head + 2 * q + 1]};
    }
    if (threadIdx.x == 0 && ((cnt - head) & 1)) {
        gk[cnt - 1] = lk[cnt - 1];
        gv[cnt - 1] = lv[cnt - 1];
    }
}

// ---- P1 FEM stream (test/femtools.jl:45-72) on a Kuhn-triangulated tensor grid --------
struct FemArgs {
    int dim;
    i64 npd, ncells;
    u64 seed;
    int order_mode;
    int bits;  // even bit count of the Feistel domain
    double h;
    KeyLayout L;
    u64 *keys;
    double *vals;
};

__device__ __forceinline__ u64 fem_cell_at(const FemArgs &a, i64 pos) {
    if (a.order_mode == 0 || a.ncells < 2) return (u64)pos;
    const int half = a.bits / 2;
    const u64 mask = (1ull << half) - 1ull;
    u64 x = (u64)pos;
    do {
        u64 Lh = x >> half, R = x & mask;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const u64 f = esp_mix64(R + a.seed + (u64)(r + 1) * 0x9E3779B97F4A7C15ull) & mask;
            const u64 tt = Lh ^ f;
            Lh = R;
            R = tt;
        }
        x = (Lh << half) | R;
    } while (x >= (u64)a.ncells);
    return x;
}

// One workgroup = FEM_CELLS consecutive stream positions; the (dim+1)(dim+2) updates of a cell are
// staged in LDS and the workgroup's contiguous range is written with coalesced 16-byte stores.
constexpr int FEM_CELLS = 128;
__global__ __launch_bounds__(FEM_CELLS) void fem_k(FemArgs a) {
    __shared__ u64 lk[FEM_CELLS * 20];
    __shared__ double lv[FEM_CELLS * 20];
    const i64 p0 = (i64)blockIdx.x * FEM_CELLS;
    const i64 p = p0 + threadIdx.x;
    const int per = (a.dim + 1) * (a.dim + 2);
    if (p < a.ncells) {
    const i64 cell = (i64)fem_cell_at(a, p);
    const int dim = a.dim;
    const int K = dim == 2 ? 2 : 6;
    const i64 q = a.npd - 1;
    const i64 cube = cell / K;
    const int s = (int)(cell % K);
    i64 vx[4][3];
    vx[0][0] = cube % q;
    vx[0][1] = (cube / q) % q;
    vx[0][2] = dim == 3 ? cube / (q * q) : 0;
    // Kuhn simplices: vertex k+1 = vertex k + e_{perm[k]}
    const int perm3[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
    const int perm2[2][2] = {{0, 1}, {1, 0}};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        if (k < dim) {
            const int ax = dim == 2 ? perm2[s][k & 1] : perm3[s][k];
#pragma unroll
            for (int d = 0; d < 3; d++) vx[k + 1][d] = vx[k][d] + (d == ax ? 1 : 0);
        }
    }
    i64 nodes[4];
    double X[4][3];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (k <= dim) {
            nodes[k] = 1 + vx[k][0] + a.npd * (vx[k][1] + a.npd * vx[k][2]);
#pragma unroll
            for (int d = 0; d < 3; d++) X[k][d] = (double)vx[k][d] * a.h;
        }
    }
    double G[4][3];
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
        for (int d = 0; d < 3; d++) G[k][d] = 0.0;
    double det;
    if (dim == 2) {
        const double aa = X[1][0] - X[0][0], bb = X[2][0] - X[0][0];
        const double cc = X[1][1] - X[0][1], dd = X[2][1] - X[0][1];
        det = aa * dd - bb * cc;
        G[1][0] = dd / det;
        G[1][1] = -bb / det;
        G[2][0] = -cc / det;
        G[2][1] = aa / det;
        G[0][0] = -(G[1][0] + G[2][0]);
        G[0][1] = -(G[1][1] + G[2][1]);
    } else {
        double J[3][3];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int k = 0; k < 3; k++) J[r][k] = X[k + 1][r] - X[0][r];
        const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
        const double c01 = J[1][0] * J[2][2] - J[1][2] * J[2][0];
        const double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
        det = J[0][0] * c00 - J[0][1] * c01 + J[0][2] * c02;
        G[1][0] = c00 / det;
        G[1][1] = -(J[0][1] * J[2][2] - J[0][2] * J[2][1]) / det;
        G[1][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) / det;
        G[2][0] = -c01 / det;
        G[2][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) / det;
        G[2][2] = -(J[0][0] * J[1][2] - J[0][2] * J[1][0]) / det;
        G[3][0] = c02 / det;
        G[3][1] = -(J[0][0] * J[2][1] - J[0][1] * J[2][0]) / det;
        G[3][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) / det;
#pragma unroll
        for (int d = 0; d < 3; d++) G[0][d] = -((G[1][d] + G[2][d]) + G[3][d]);
    }
    const double vol = fabs(det) / (dim == 2 ? 2.0 : 6.0);
    double S[4][4];
#pragma unroll
    for (int il = 0; il < 4; il++)
#pragma unroll
        for (int jl = 0; jl < 4; jl++) {
            if (il <= dim && jl >= il && jl <= dim) {
                double sacc = 0.0;
#pragma unroll
                for (int k = 0; k < 3; k++)
                    if (k < dim) sacc += G[jl][k] * G[il][k];
                S[il][jl] = sacc;
                S[jl][il] = sacc;
            }
        }
    int o = threadIdx.x * per;
#pragma unroll
    for (int il = 0; il < 4; il++) {
        if (il <= dim) {
            lk[o] = esp_pack(a.L, nodes[il], nodes[il], ESP_RAWUPDATE);
            lv[o] = 0.1 * vol / (double)(dim + 1);
            o++;
#pragma unroll
            for (int jl = 0; jl < 4; jl++) {
                if (jl <= dim) {
                    lk[o] = esp_pack(a.L, nodes[il], nodes[jl], ESP_RAWUPDATE);
                    lv[o] = vol * S[il][jl];
                    o++;
                }
            }
        }
    }
    }  // p < ncells
    __syncthreads();
    const i64 ncell_blk = min((i64)FEM_CELLS, a.ncells - p0);
    const int cnt = (int)ncell_blk * per;
    u64 *gk = a.keys + p0 * per;
    double *gv = a.vals + p0 * per;
    const int head = (int)(((uintptr_t)gk >> 3) & 1);
    if (threadIdx.x == 0 && head && cnt > 0) {
        gk[0] = lk[0];
        gv[0] = lv[0];
    }
    const int npair = (cnt - head) >> 1;
    typedef unsigned long long ull2 __attribute__((ext_vector_type(2)));
    typedef double dbl2 __attribute__((ext_vector_type(2)));
    ull2 *gk2 = reinterpret_cast<ull2 *>(gk + head);
    dbl2 *gv2 = reinterpret_cast<dbl2 *>(gv + head);
    for (int q = threadIdx.x; q < npair; q += FEM_CELLS) {
        gk2[q] = ull2{lk[head + 2 * q], lk[head + 2 * q + 1]};
        gv2[q] = dbl2{lv[head + 2 * q], lv[head + 2 * q + 1]};
    }
    if (threadIdx.x == 0 && cnt > 0 && ((cnt - head) & 1)) {
        gk[cnt - 1] = lk[cnt - 1];
        gv[cnt - 1] = lv[cnt - 1];
    }
}

}  // namespace espgen
