// generators.hpp -- append-path kernels (K1): pack host/device (i,j,v) triples into the
// COO buffer, and on-device producers of the reference's update streams.
// All of them write the buffer in the exact order of the sequential reference loop, so
// the ordered fold reproduces the CPU accumulation order bit for bit.
// Compiled with -ffp-contract=off: no FMA contraction, values match the oracle's bits.
#pragma once
#include <type_traits>

#include "common.hpp"
#include "runpart.hpp"

namespace espgen {

constexpr int THREADS = 256;

// (rows, cols, vals, kinds) -> packed keys.  Out-of-range indices raise *err (first bad
// position + 1) and nothing of the batch is committed by the host side
// (BoundsError, sparsematrixcsc.jl:8-10).
static __global__ __launch_bounds__(THREADS) void pack_k(const i64 *__restrict__ rows,
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
    // nx*ny*nz < 2^32: node -> (i,j,k) with two multiply-high instead of 64-bit divisions (fd_node)
    int fast;
    u64 magic_nx, magic_nxny;
    esprun::PartOut part;  // the PART kernels: the append is the partition (runpart.hpp)
    KeyLayout L;
    u64 *keys;
    double *vals;
};

__device__ __forceinline__ double fd_rand(int mode, u64 seed, u64 ctr) {
    if (mode == 0) return 1.0;
    if (mode == 1) return 0.1 + esp_uniform(seed, ctr);
    return esp_uniform(seed, ctr);
}
// draw k (0..5) of a node whose draw 0 has z0 = seed + (6 g + 1) * golden: the same bits as fd_rand(mode, seed, 6 g + k)
__device__ __forceinline__ double fd_rand_z(int mode, u64 z0, int k) {
    if (mode == 0) return 1.0;
    const double u = esp_uniform_z(z0 + (u64)k * ESP_GOLDEN);
    return mode == 1 ? 0.1 + u : u;
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

// n / d for n, d < 2^32 with magic = 2^64 / d + 1 (host: fd_magic)
__device__ __forceinline__ u32 fast_div(u32 n, u64 magic) { return (u32)__umul64hi(magic, (u64)n); }
// node g (0-based l-1) -> (i,j,k), 1-based
__device__ __forceinline__ void fd_node(const FdArgs &a, i64 g, i64 *i, i64 *j, i64 *k) {
    if (a.fast) {
        const u32 g32 = (u32)g, nx = (u32)a.nx, nxny = (u32)(a.nx * a.ny);
        const u32 k0 = fast_div(g32, a.magic_nxny);
        const u32 rem = g32 - k0 * nxny;
        const u32 j0 = fast_div(rem, a.magic_nx);
        *i = (i64)(rem - j0 * nx) + 1;
        *j = (i64)j0 + 1;
        *k = (i64)k0 + 1;
    } else {
        *i = g % a.nx + 1;
        *j = (g / a.nx) % a.ny + 1;
        *k = g / (a.nx * a.ny) + 1;
    }
}

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
// stream: they are staged in LDS in stream order.  Returns the range [off0, off0 + cnt) of the stream.
__device__ __forceinline__ void fd_stage(const FdArgs &a, u64 *lk, double *lv, i64 *off0_out, int *cnt_out) {
    const i64 N = a.g_end;
    const i64 g0 = a.g_begin + (i64)blockIdx.x * THREADS;
    const i64 g = g0 + threadIdx.x;  // node l-1
    i64 cy, cz;
    // stream position of the first node of this workgroup and of the next one
    i64 i0, j0, k0;
    fd_node(a, g0, &i0, &j0, &k0);
    const i64 off0 = fd_offset(a, i0, j0, k0, &cy, &cz) - a.off_begin;
    if (g < N) {
        i64 i, j, k;
        fd_node(a, g, &i, &j, &k);
        int o = (int)(fd_offset(a, i, j, k, &cy, &cz) - a.off_begin - off0);
        const i64 l = g + 1;
        const u64 c = 6ull * (u64)g;
        if (i < a.nx) fd_pair(a, lk, lv, o, fd_rand(a.rand_mode, a.seed, c + 0) * a.hy * a.hz / a.hx, l, l + 1);
        if (i == 1 || i == a.nx) fd_put(a, lk, lv, o, fd_rand(a.rand_mode, a.seed, c + 1) * a.hy * a.hz, l, l);
        if (j < a.ny) fd_pair(a, lk, lv, o, fd_rand(a.rand_mode, a.seed, c + 2) * a.hx * a.hz / a.hy, l, l + a.nx);
        if (a.ny > 2 && (j == 1 || j == a.ny)) fd_put(a, lk, lv, o, fd_rand(a.rand_mode, a.seed, c + 3) * a.hx * a.hz, l, l);
        if (k < a.nz) fd_pair(a, lk, lv, o, fd_rand(a.rand_mode, a.seed, c + 4) * a.hx * a.hy / a.hz, l, l + a.nx * a.ny);
        if (a.nz > 2 && (k == 1 || k == a.nz)) fd_put(a, lk, lv, o, fd_rand(a.rand_mode, a.seed, c + 5) * a.hx * a.hy, l, l);
    }
    // total of the workgroup = position of the first node of the next workgroup (or the stream end)
    i64 off1;
    const i64 g1 = g0 + THREADS;
    if (g1 >= N) {
        off1 = a.total;
    } else {
        i64 i1, j1, k1;
        fd_node(a, g1, &i1, &j1, &k1);
        off1 = fd_offset(a, i1, j1, k1, &cy, &cz) - a.off_begin;
    }
    *off0_out = off0;
    *cnt_out = (int)(off1 - off0);
}

// coalesced copy-out of a staged range to its place in the stream; 16-byte stores on the aligned body
template <int NT>
__device__ __forceinline__ void copy_out_staged(const u64 *lk, const double *lv, int cnt, u64 *gk, double *gv) {
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
    for (int q = threadIdx.x; q < npair; q += NT) {
        gk2[q] = ull2{lk[head + 2 * q], lk[head + 2 * q + 1]};
        gv2[q] = dbl2{lv[head + 2 * q], lv[head + 2 * q + 1]};
    }
    if (threadIdx.x == 0 && cnt > 0 && ((cnt - head) & 1)) {
        gk[cnt - 1] = lk[cnt - 1];
        gv[cnt - 1] = lv[cnt - 1];
    }
}

// The stream goes to the buffer as it is (coalesced 16-byte stores: K1, "coalesced HBM stores on the append path";
// writes 16 B per update, reads nothing).
static __global__ __launch_bounds__(THREADS) void fdrand_k(FdArgs a) {
    __shared__ u64 lk[THREADS * FD_MAX_PER_NODE];
    __shared__ double lv[THREADS * FD_MAX_PER_NODE];
    i64 off0;
    int cnt;
    fd_stage(a, lk, lv, &off0, &cnt);
    __syncthreads();
    copy_out_staged<THREADS>(lk, lv, cnt, a.keys + off0, a.vals + off0);
}

// What a node sends where: 2 updates per pair it starts to its own column and 2 to the partner's, its boundary terms
// to its own (fd_pair / fd_put above) -- four (column, count) items; px/py/pz: the node starts a pair in x/y/z
struct FdItems {
    u32 dig[4], wt[4];
    bool px, py, pz, bx, by, bz;
    i64 i, j, k;
};
__device__ __forceinline__ void fd_items(const FdArgs &a, i64 g, FdItems &it, u32 *err) {
#pragma unroll
    for (int q = 0; q < 4; q++) it.dig[q] = it.wt[q] = 0;
    it.px = it.py = it.pz = it.bx = it.by = it.bz = false;
    if (g >= a.g_end) return;
    fd_node(a, g, &it.i, &it.j, &it.k);
    it.px = it.i < a.nx;
    it.py = it.j < a.ny;
    it.pz = it.k < a.nz;
    it.bx = it.i == 1 || it.i == a.nx;
    it.by = a.ny > 2 && (it.j == 1 || it.j == a.ny);
    it.bz = a.nz > 2 && (it.k == 1 || it.k == a.nz);
    const esprun::PartOut &p = a.part;
    it.wt[0] = 2u * ((u32)it.px + (u32)it.py + (u32)it.pz) + (u32)it.bx + (u32)it.by + (u32)it.bz;
    it.wt[1] = 2u * (u32)it.px;
    it.wt[2] = 2u * (u32)it.py;
    it.wt[3] = 2u * (u32)it.pz;
    it.dig[0] = esprun::column_digit(p, g, a.L.rb, err);
    if (it.px) it.dig[1] = esprun::column_digit(p, g + 1, a.L.rb, err);
    if (it.py) it.dig[2] = esprun::column_digit(p, g + a.nx, a.L.rb, err);
    if (it.pz) it.dig[3] = esprun::column_digit(p, g + a.nx * a.ny, a.L.rb, err);
}

// COUNT launch of the stencil producer: no update is formed.  Same chunks (workgroups of 256 nodes) as the PART launch.
static __global__ __launch_bounds__(THREADS) void fd_count_k(FdArgs a, esprun::RunSink sink, u32 *err) {
    __shared__ u32 rd[esprun::RMAX];
    __shared__ u32 rc[esprun::RMAX];
    __shared__ u32 over;
    if (threadIdx.x < esprun::RMAX) {
        rd[threadIdx.x] = esprun::EMPTY;
        rc[threadIdx.x] = 0;
    }
    if (threadIdx.x == 0) over = 0;
    // a stream that is not pre-sorted is recognised by the first workgroups: the rest leave at once (the flag is
    // requested first and looked at after the node's items are formed: its round trip is not waited for)
    const u32 give_up = __hip_atomic_load(sink.overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    FdItems it;
    fd_items(a, a.g_begin + (i64)blockIdx.x * THREADS + threadIdx.x, it, err);
    if (give_up != 0u) return;
    __syncthreads();
    esprun::count_runs_weighted<4>(it.dig, it.wt, a.part.chunk_base + blockIdx.x, sink, rd, rc, &over);
}

// PART launch: every update goes straight to its bucket -- the flush needs no partition pass.  The workgroup's updates
// are staged in LDS run by run (esprun::tile_slots: a node's updates for one column lie together, in call order) and
// every run is copied to `run offset` with consecutive stores.  S32: the staging area holds the low 32 bits of a key
// (46 KiB of LDS: 3 workgroups per CU; buckets narrower than 2^32 keys), OUT32: 4-byte keys go out (12 B per update).
// Staging capacity of the PART launch: an interior tile holds 256 * 12 = 3072 updates, a tile along the domain boundary
// up to 15 per node.  With room for 3168 (S32: 12 B each) the kernel needs 40 KiB of LDS -- FOUR workgroups per CU -- and
// the few tiles above that go out in two rounds (staged window [lo, lo + FD_STAGE) of the tile's slots).
constexpr int FD_STAGE = 3168;
template <bool S32, bool OUT32>
static __global__ __launch_bounds__(THREADS) void fdrand_part_k(FdArgs a) {
    static_assert(S32 || !OUT32, "4-byte keys come from a 4-byte staging area");
    typedef typename std::conditional<S32, u32, u64>::type KT;
    __shared__ KT lk[FD_STAGE];
    __shared__ double lv[FD_STAGE];
    __shared__ esprun::TileLds<THREADS / ESP_WAVE> S;
    const esprun::PartOut &p = a.part;
    const esprun::TileLoads tl = esprun::tile_loads(p, p.chunk_base + blockIdx.x);
    // (the flags are looked at by tile_slots: the node's items are formed while the tile's table is on its way)
    const i64 g = a.g_begin + (i64)blockIdx.x * THREADS + threadIdx.x;
    FdItems it;
    fd_items(a, g, it, nullptr);  // (the COUNT launch checked the window)
    u32 slot[4];
    int total;
    // (leaves -- uniformly -- when a flag is set or 4-byte keys do not apply: the host issues the plain producer instead)
    esprun::TileRegs tr;
    if (!esprun::tile_slots<4, THREADS / ESP_WAVE>(p, tl, it.dig, it.wt, slot, S, &total, OUT32, &tr)) return;
    const int rb = a.L.rb;
    for (int lo = 0; lo < total; lo += FD_STAGE) {  // (one round, but for a boundary tile)
        if (lo > 0) __syncthreads();                // (the previous round's copy has read the staging area)
        auto put = [&](u32 at, double v, i64 row, i64 col) {
            at -= (u32)lo;
            if (at >= (u32)FD_STAGE) return;
            if constexpr (S32) {  // (the low 32 bits of (col-1) << rb | (row-1), in 32-bit arithmetic)
                lk[at] = (rb < 32 ? (u32)(col - 1) << rb : 0u) | (u32)(row - 1);
            } else {
                const u64 kp = ((u64)(col - 1) << rb) | (u64)(row - 1);
                lk[at] = (kp << ESP_TAG_BITS) | (u64)a.kind;
            }
            lv[at] = v;
        };
        // update_pair (sprand.jl:87-92): (l,l2) (l2,l) (l,l) (l2,l2) -- column l2 gets the first and the last one
        auto pair = [&](u32 &own, u32 other, double v, i64 l, i64 l2) {
            put(other, -v, l, l2);
            put(own++, -v, l2, l);
            put(own++, v, l, l);
            put(other + 1, v, l2, l2);
        };
        if (g < a.g_end) {
            const i64 l = g + 1;
            const u64 z0 = a.seed + (6ull * (u64)g + 1ull) * ESP_GOLDEN;  // (draw k of this node: counter 6 g + k)
            const int md = a.rand_mode;
            u32 own = slot[0];
            if (it.px) pair(own, slot[1], fd_rand_z(md, z0, 0) * a.hy * a.hz / a.hx, l, l + 1);
            if (it.bx) put(own++, fd_rand_z(md, z0, 1) * a.hy * a.hz, l, l);
            if (it.py) pair(own, slot[2], fd_rand_z(md, z0, 2) * a.hx * a.hz / a.hy, l, l + a.nx);
            if (it.by) put(own++, fd_rand_z(md, z0, 3) * a.hx * a.hz, l, l);
            if (it.pz) pair(own, slot[3], fd_rand_z(md, z0, 4) * a.hx * a.hy / a.hz, l, l + a.nx * a.ny);
            if (it.bz) put(own++, fd_rand_z(md, z0, 5) * a.hx * a.hy, l, l);
        }
        __syncthreads();
        if constexpr (S32 && !OUT32) {
            if (S.all_own) {  // (a shard's usual tile: the 4-byte-key copy loop of the unsharded producer)
                esprun::copy_out_runs<KT, true, THREADS, THREADS / ESP_WAVE>(p, lk, lv, total, S, (u32)a.kind, lo, lo + FD_STAGE,
                                                                           esprun::own_keys32(p.keys_out, S.own_lo), &tr);
                continue;
            }
        }
        esprun::copy_out_runs<KT, OUT32, THREADS, THREADS / ESP_WAVE>(p, lk, lv, total, S, (u32)a.kind, lo, lo + FD_STAGE, nullptr, &tr);
    }
}

// ---- P1 FEM stream (test/femtools.jl:45-72) on a Kuhn-triangulated tensor grid --------
struct FemArgs {
    int dim;
    i64 npd, ncells;
    u64 seed;
    int order_mode;
    int bits;  // even bit count of the Feistel domain
    u64 mq;    // fem_fill_magic: ceil(2^64 / (npd - 1)) when every cell number and (npd - 1)^2 fit 32 bits, else 0
    double h;
    KeyLayout L;
    esprun::PartOut part;  // part.on: the append is the partition (runpart.hpp)
    u64 *keys;
    double *vals;
};

// one application of the 4-round Feistel permutation of [0, 2^bits)
__device__ __forceinline__ u64 fem_feistel(const FemArgs &a, u64 x) {
    const int half = a.bits / 2;
    const u64 mask = (1ull << half) - 1ull;
    u64 Lh = x >> half, R = x & mask;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const u64 f = esp_mix64(R + a.seed + (u64)(r + 1) * 0x9E3779B97F4A7C15ull) & mask;
        const u64 tt = Lh ^ f;
        Lh = R;
        R = tt;
    }
    return (Lh << half) | R;
}
// the cell at stream position pos: the permutation, walked until it lands inside [0, ncells) (cycle walking)
__device__ __forceinline__ u64 fem_cell_at(const FemArgs &a, i64 pos) {
    if (a.order_mode == 0 || a.ncells < 2) return (u64)pos;
    u64 x = (u64)pos;
    do x = fem_feistel(a, x);
    while (x >= (u64)a.ncells);
    return x;
}

// x / d for 32-bit x and d through one multiplication by m = ceil(2^64 / d) (exact for every 32-bit x: the error term
// x (m d - 2^64) / (d 2^64) stays below 1 / d): the integer divisions of the vertex arithmetic below cost over a
// hundred instructions each as 64-bit divisions -- a third of a cell's work
__host__ __device__ inline u64 fem_magic(u64 d) { return d > 1 ? ~0ull / d + 1ull : 0ull; }
__device__ __forceinline__ u32 fem_div32(u32 x, u64 m) {
    return (u32)__umul64hi(m, (u64)x);
}
static inline void fem_fill_magic(struct FemArgs &a);

// vertices of a cell of the Kuhn triangulation: grid coordinates vx[k][d] and node numbers (1-based)
__device__ __forceinline__ void fem_vertices(const FemArgs &a, i64 cell, i64 (&vx)[4][3], i64 (&nodes)[4]) {
    const int dim = a.dim;
    const int K = dim == 2 ? 2 : 6;
    const i64 q = a.npd - 1;
    int s;
    if (a.mq) {  // (32-bit arithmetic, divisions by multiplication: the same integers)
        const u32 c32 = (u32)cell;
        const u32 cube = dim == 2 ? c32 >> 1 : c32 / 6u;
        s = (int)(c32 - cube * (u32)K);
        const u32 t1 = q > 1 ? fem_div32(cube, a.mq) : cube;
        const u32 t2 = q > 1 ? fem_div32(t1, a.mq) : t1;
        vx[0][0] = (i64)(cube - t1 * (u32)q);
        vx[0][1] = (i64)(t1 - t2 * (u32)q);
        vx[0][2] = dim == 3 ? (i64)t2 : 0;
    } else {
        const i64 cube = cell / K;
        s = (int)(cell % K);
        vx[0][0] = cube % q;
        vx[0][1] = (cube / q) % q;
        vx[0][2] = dim == 3 ? cube / (q * q) : 0;
    }
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
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (k <= dim) nodes[k] = 1 + vx[k][0] + a.npd * (vx[k][1] + a.npd * vx[k][2]);
}
static inline void fem_fill_magic(FemArgs &a) {
    const u64 q = (u64)(a.npd - 1);
    a.mq = ((u64)a.ncells < (1ull << 32) && q * q < (1ull << 32)) ? fem_magic(q) : 0ull;
}

// One workgroup = FEM_CELLS consecutive stream positions; the (dim+1)(dim+2) updates of a cell are
// staged in LDS in stream order.
constexpr int FEM_CELLS = 128;
constexpr int FEM_MAX_PER_CELL = 20;
// The updates of the cell at stream position p, in call order (test/femtools.jl:62-69): for every local row il the mass
// term on the diagonal, then the row of the element matrix.  emit(il, jl, row, col, v); jl = -1 for the mass term.
// ONE copy of the element arithmetic for every producer (fixed operation order, no FMA: the oracle's sequence).
template <typename F>
__device__ __forceinline__ void fem_updates_of_cell(const FemArgs &a, i64 cell, F emit);
template <typename F>
__device__ __forceinline__ void fem_cell_updates(const FemArgs &a, i64 p, F emit) {
    fem_updates_of_cell(a, (i64)fem_cell_at(a, p), emit);
}
// n / d for several numerators over ONE denominator, bit for bit the IEEE quotient: the compiler's f64 division is
// div_scale x 2, rcp, four fma refining the reciprocal, a product, a remainder fma, div_fmas, div_fixup -- of which the
// reciprocal and its refinement depend on d alone whenever div_scale does not scale (both operands well inside the normal
// range: what a cell of a mesh gives).  SharedDiv keeps them; quot() is the product, the exact remainder and the final fma --
// the very operations div_fmas / div_fixup perform on unscaled, finite operands.  Operands outside [2^-400, 2^400] (or a zero
// / non-finite denominator): the plain division.  The nine gradient components of a tetrahedron share their determinant.
struct SharedDiv {
    double d, r;
    bool fast;
    __device__ __forceinline__ static bool mid(double x) {
        const double ax = fabs(x);
        return ax > 0x1.0p-400 && ax < 0x1.0p400;
    }
    __device__ __forceinline__ explicit SharedDiv(double den) : d(den) {
        fast = mid(den);
        const double r0 = __builtin_amdgcn_rcp(den);
        const double e0 = fma(-den, r0, 1.0);
        const double r1 = fma(r0, e0, r0);
        const double e1 = fma(-den, r1, 1.0);
        r = fma(r1, e1, r1);
    }
    __device__ __forceinline__ double quot(double n) const {
        if (fast && (n == 0.0 || mid(n))) {
            const double q0 = n * r;
            const double rem = fma(-d, q0, n);
            return fma(rem, r, q0);
        }
        return n / d;
    }
};

// vertices' node numbers, gradients of the P1 basis functions and the volume of one cell
__device__ __forceinline__ void fem_cell_geometry(const FemArgs &a, i64 cell, i64 (&nodes)[4], double (&G)[4][3], double *vol_out) {
    const int dim = a.dim;
    i64 vx[4][3];
    fem_vertices(a, cell, vx, nodes);
    double X[4][3];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (k <= dim) {
#pragma unroll
            for (int d = 0; d < 3; d++) X[k][d] = (double)vx[k][d] * a.h;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
        for (int d = 0; d < 3; d++) G[k][d] = 0.0;
    double det;
    if (dim == 2) {
        const double aa = X[1][0] - X[0][0], bb = X[2][0] - X[0][0];
        const double cc = X[1][1] - X[0][1], dd = X[2][1] - X[0][1];
        det = aa * dd - bb * cc;
        const SharedDiv by(det);
        G[1][0] = by.quot(dd);
        G[1][1] = by.quot(-bb);
        G[2][0] = by.quot(-cc);
        G[2][1] = by.quot(aa);
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
        const SharedDiv by(det);
        G[1][0] = by.quot(c00);
        G[1][1] = by.quot(-(J[0][1] * J[2][2] - J[0][2] * J[2][1]));
        G[1][2] = by.quot(J[0][1] * J[1][2] - J[0][2] * J[1][1]);
        G[2][0] = by.quot(-c01);
        G[2][1] = by.quot(J[0][0] * J[2][2] - J[0][2] * J[2][0]);
        G[2][2] = by.quot(-(J[0][0] * J[1][2] - J[0][2] * J[1][0]));
        G[3][0] = by.quot(c02);
        G[3][1] = by.quot(-(J[0][0] * J[2][1] - J[0][1] * J[2][0]));
        G[3][2] = by.quot(J[0][0] * J[1][1] - J[0][1] * J[1][0]);
#pragma unroll
        for (int d = 0; d < 3; d++) G[0][d] = -((G[1][d] + G[2][d]) + G[3][d]);
    }
    *vol_out = fabs(det) / (dim == 2 ? 2.0 : 6.0);
}
template <typename F>
__device__ __forceinline__ void fem_updates_of_cell(const FemArgs &a, i64 cell, F emit) {
    const int dim = a.dim;
    i64 nodes[4];
    double G[4][3];
    double vol;
    fem_cell_geometry(a, cell, nodes, G, &vol);
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
#pragma unroll
    for (int il = 0; il < 4; il++) {
        if (il <= dim) {
            emit(il, -1, nodes[il], nodes[il], 0.1 * vol / (double)(dim + 1));
#pragma unroll
            for (int jl = 0; jl < 4; jl++)
                if (jl <= dim) emit(il, jl, nodes[il], nodes[jl], vol * S[il][jl]);
        }
    }
}
// The updates of ONE vertex column of the cell (the item partition's expansion: femitems.hpp), the same values bit for bit:
// S[il][jl] = sum_k G[max(il,jl)][k] * G[min(il,jl)][k] in the order k = 0, 1, 2 -- a product does not depend on the order of
// its factors -- so only the dim + 1 scalar products of the column are formed.  emit(il, jl, row, v): jl = -1 for the mass
// term (row il = jl's own), else the column's local index; in call order of the column's entries.
template <typename F>
__device__ __forceinline__ void fem_column_of_cell(const FemArgs &a, i64 cell, i64 icol, F emit) {
    const int dim = a.dim;
    i64 nodes[4];
    double G[4][3];
    double vol;
    fem_cell_geometry(a, cell, nodes, G, &vol);
    int jv = 0;
#pragma unroll
    for (int k = 1; k < 4; k++)
        if (k <= dim && nodes[k] == icol) jv = k;
    double Gj[3];
#pragma unroll
    for (int d = 0; d < 3; d++) Gj[d] = jv == 0 ? G[0][d] : jv == 1 ? G[1][d] : jv == 2 ? G[2][d] : G[3][d];
#pragma unroll
    for (int il = 0; il < 4; il++) {
        if (il <= dim) {
            if (il == jv) emit(il, -1, nodes[il], 0.1 * vol / (double)(dim + 1));
            double sacc = 0.0;
#pragma unroll
            for (int k = 0; k < 3; k++)
                if (k < dim) sacc += Gj[k] * G[il][k];
            emit(il, jv, nodes[il], vol * sacc);
        }
    }
}

// slot == nullptr: stream order from LDS position o0 on (packed keys); else entry (il,jl) goes to the column's item
// (esprun::tile_slots) and KT = u32 stages the key bits below the bucket prefix
template <typename KT>
__device__ __forceinline__ void fem_stage(const FemArgs &a, KT *lk, double *lv, int o0, const u32 *slot, i64 p) {
    if (p >= a.ncells) return;
    int o = o0;
    fem_cell_updates(a, p, [&](int il, int jl, i64 row, i64 col, double v) {
        // (within a column's item: row il's term at il, +1 from the diagonal's row on -- the mass term comes right before it)
        const int at = !slot ? o : jl < 0 ? (int)slot[il] + il : (int)slot[jl] + il + (il >= jl ? 1 : 0);
        if constexpr (sizeof(KT) == 4)
            lk[at] = (u32)(((u64)(col - 1) << a.L.rb) | (u64)(row - 1));  // (low 32 bits: see esprun::copy_out_runs)
        else
            lk[at] = esp_pack(a.L, row, col, ESP_RAWUPDATE);
        lv[at] = v;
        o++;
    });
}

static __global__ __launch_bounds__(FEM_CELLS) void fem_k(FemArgs a) {
    __shared__ u64 lk[FEM_CELLS * FEM_MAX_PER_CELL];
    __shared__ double lv[FEM_CELLS * FEM_MAX_PER_CELL];
    const i64 p0 = (i64)blockIdx.x * FEM_CELLS;
    const int per = (a.dim + 1) * (a.dim + 2);
    fem_stage(a, lk, lv, threadIdx.x * per, nullptr, p0 + threadIdx.x);
    __syncthreads();
    const int cnt = (int)min((i64)FEM_CELLS, a.ncells - p0) * per;
    copy_out_staged<FEM_CELLS>(lk, lv, cnt, a.keys + p0 * per, a.vals + p0 * per);
}

// the vertex columns of a cell as (digit, count) items: every vertex column receives dim+2 updates (one per row of
// the element matrix, one more for the diagonal's mass term)
__device__ __forceinline__ void fem_items(const FemArgs &a, i64 pos, u32 (&dig)[4], u32 (&wt)[4], u32 *err) {
#pragma unroll
    for (int k = 0; k < 4; k++) dig[k] = wt[k] = 0;
    if (pos >= a.ncells) return;
    i64 vx[4][3];
    i64 nodes[4];
    fem_vertices(a, (i64)fem_cell_at(a, pos), vx, nodes);
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (k <= a.dim) {
            wt[k] = (u32)(a.dim + 2);
            dig[k] = esprun::column_digit(a.part, nodes[k] - 1, a.L.rb, err);
        }
}

// COUNT launch of the FEM producer (same chunks -- workgroups of 128 cells -- as the PART launch)
static __global__ __launch_bounds__(FEM_CELLS) void fem_count_k(FemArgs a, esprun::RunSink sink, u32 *err) {
    __shared__ u32 rd[esprun::RMAX];
    __shared__ u32 rc[esprun::RMAX];
    __shared__ u32 over;
    if (threadIdx.x < esprun::RMAX) {
        rd[threadIdx.x] = esprun::EMPTY;
        rc[threadIdx.x] = 0;
    }
    if (threadIdx.x == 0) over = 0;
    if (__hip_atomic_load(sink.overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;  // (see fd_count_k)
    __syncthreads();
    u32 dig[4], wt[4];
    fem_items(a, (i64)blockIdx.x * FEM_CELLS + threadIdx.x, dig, wt, err);
    esprun::count_runs_weighted<4>(dig, wt, a.part.chunk_base + blockIdx.x, sink, rd, rc, &over);
}

// PART launch of the FEM producer (see fdrand_part_k): the updates of a cell for vertex column jl lie together, in
// call order: row il's term at il (+1 from the diagonal's row on: the mass term comes right before the diagonal)
template <bool S32, bool OUT32>
static __global__ __launch_bounds__(FEM_CELLS) void fem_part_k(FemArgs a) {
    typedef typename std::conditional<S32, u32, u64>::type KT;
    __shared__ KT lk[FEM_CELLS * FEM_MAX_PER_CELL];
    __shared__ double lv[FEM_CELLS * FEM_MAX_PER_CELL];
    __shared__ esprun::TileLds<FEM_CELLS / ESP_WAVE> S;
    const esprun::PartOut &p = a.part;
    const esprun::TileLoads tl = esprun::tile_loads(p, p.chunk_base + blockIdx.x);
    if (esprun::tile_stop(p, tl, OUT32)) return;
    const i64 pos = (i64)blockIdx.x * FEM_CELLS + threadIdx.x;
    u32 dig[4], wt[4], slot[4];
    fem_items(a, pos, dig, wt, nullptr);
    int total;
    if (!esprun::tile_slots<4, FEM_CELLS / ESP_WAVE>(p, tl, dig, wt, slot, S, &total, OUT32)) return;
    fem_stage(a, lk, lv, 0, slot, pos);
    __syncthreads();
    esprun::copy_out_runs<KT, OUT32, FEM_CELLS, FEM_CELLS / ESP_WAVE>(p, lk, lv, total, S, (u32)ESP_RAWUPDATE);
}

}  // namespace espgen
