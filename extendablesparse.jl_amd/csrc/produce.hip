// produce.hip -- libesparse_hip: device-side producers (esp_generate_*) (see internal.hpp for the map of the translation units)
#include "internal.hpp"

// stream position of node g (0-based) in the k,j,i loop nest: host copy of espgen::fd_offset
i64 fd_offset_host(i64 nx, i64 ny, i64 nz, i64 g) {
    const i64 N = nx * ny * nz;
    if (g >= N) {
        i64 E = 4 * (nx - 1) * ny * nz + (nx == 1 ? 1 : 2) * ny * nz;
        E += 4 * nx * (ny - 1) * nz + (ny > 2 ? 2 * nx * nz : 0);
        E += 4 * nx * ny * (nz - 1) + (nz > 2 ? 2 * nx * ny : 0);
        return E;
    }
    const i64 i = g % nx + 1, j = (g / nx) % ny + 1, k = g / (nx * ny) + 1;
    const i64 CX = 4 * (nx - 1) + (nx == 1 ? 1 : 2);
    const i64 CY = 4 * (ny - 1) + (ny > 2 ? 2 : 0);
    const i64 PX = 4 * (i - 1) + (i > 1 ? 1 : 0);
    const i64 PY = 4 * (j - 1) + ((ny > 2 && j > 1) ? 1 : 0);
    const i64 PZ = 4 * (k - 1) + ((nz > 2 && k > 1) ? 1 : 0);
    const i64 cy = (j < ny ? 4 : 0) + ((ny > 2 && (j == 1 || j == ny)) ? 1 : 0);
    const i64 cz = (k < nz ? 4 : 0) + ((nz > 2 && (k == 1 || k == nz)) ? 1 : 0);
    return (k - 1) * (ny * CX + nx * CY) + nx * ny * PZ + (j - 1) * CX + nx * PY + (j - 1) * nx * cz + PX + (i - 1) * (cy + cz);
}

extern "C" int32_t esp_generate_fdrand_range(esp_handle *h, int64_t nx, int64_t ny, int64_t nz, uint64_t seed, int32_t rand_mode,
                                             int32_t kind, int64_t node_begin, int64_t node_end) {
    if (!h) return ESP_ERR_INVALID;
    if (nx < 1 || ny < 1 || nz < 1) FAIL(h, ESP_ERR_INVALID, "fdrand: bad grid");
    const i64 N = nx * ny * nz;
    if (h->m != N || h->n != N) FAIL(h, ESP_ERR_INVALID, "Matrix size mismatch");  // sprand.jl:66-68
    if (kind != ESP_UPDATE && kind != ESP_RAWUPDATE && kind != ESP_COO) FAIL(h, ESP_ERR_INVALID, "fdrand: kind must be UPDATE, RAWUPDATE or COO");
    if (rand_mode < 0 || rand_mode > 2) FAIL(h, ESP_ERR_INVALID, "fdrand: rand_mode");
    if (node_begin < 0 || node_end > N || node_begin > node_end) FAIL(h, ESP_ERR_INVALID, "fdrand: node range");
    if (node_begin == node_end) return ESP_OK;
    (void)hipSetDevice(h->device);
    const i64 off_b = fd_offset_host(nx, ny, nz, node_begin);
    const i64 E = fd_offset_host(nx, ny, nz, node_end) - off_b;
    CK(reserve_append(h, E));
    espgen::FdArgs a;
    a.nx = nx;
    a.ny = ny;
    a.nz = nz;
    a.hx = 1.0 / (double)nx;
    a.hy = 1.0 / (double)ny;
    a.hz = 1.0 / (double)nz;
    a.seed = seed;
    a.rand_mode = rand_mode;
    a.kind = kind;
    a.total = E;
    a.g_begin = node_begin;
    a.g_end = node_end;
    a.off_begin = off_b;
    // (magic = 2^64 / d + 1: n / d = high half of magic * n for n, d < 2^32, d >= 2)
    a.fast = (N < ((i64)1 << 32) && nx >= 2) ? 1 : 0;
    a.magic_nx = a.fast ? ~0ull / (u64)nx + 1ull : 0;
    a.magic_nxny = a.fast ? ~0ull / (u64)(nx * ny) + 1ull : 0;
    a.L = h->L;
    a.keys = (u64 *)h->keys.p + h->count;
    a.vals = (double *)h->vals.p + h->count;
    CK(ensure(h, h->misc, 256));
    const dim3 grid(grid_for(node_end - node_begin, espgen::THREADS)), block(espgen::THREADS);
    // the append is the partition when the buffer is empty and the stream is one an assembly loop emits: COUNT launch
    // (ALU only), two tiny ranking launches, then every update goes straight to its bucket
    // ... and when the call repeats the handle's last one (same grid, node range and kind on the same empty buffer: a time
    // loop) the tables of that call still stand: straight to the PART launch (esp_handle::GenPlan; force_path 31: never)
    PartSetup ps;
    bool took = false;
    const esp_handle::GenPlan &gp = h->genplan;
    const bool reuse = gp.valid && gp.nx == nx && gp.ny == ny && gp.nz == nz && gp.g0 == node_begin && gp.g1 == node_end && gp.kind == kind &&
                       gp.E == E && h->count == 0 && h->force_path == ESP_PATH_AUTO && !h->part_assembled &&
                       gp.base == h->win_base && gp.span == h->win_span && gp.keys_at == h->keys.p && h->runs_skip == 0 &&
                       // (a shard: the plan was made for the exchange the caller announced -- esp_shard_plan -- and that one only)
                       (h->shard_user ? (h->shard_plan.valid && gp.pre.mw_P == h->shard_plan.P && gp.pre.mw_me == h->shard_plan.me &&
                                         gp.pre.mw_eps == h->shard_plan.eps)
                                      : gp.pre.mw_P == 0);
    h->last_plan_reused = reuse ? 1 : 0;
    if (reuse) {
        a.part = gp.out;
        a.part.keys_out = (u64 *)h->keys.p;
        a.part.vals_out = (double *)h->vals.p;
        // (the 64-byte block the PART launch looks at -- longest bucket, which it only compares with the bucket kernel's capacity, and
        // the flag words: flush_pre_tail, flush_rebuild, a shard's assemble and a failed flush all write there; the plan's own
        // longest bucket fitted, or the plan would not have been kept)
        HIPCK(h, hipMemsetAsync((unsigned long long *)h->misc.p + 24, 0, 64, h->stream));
        Span sp(h, ESP_ST_APPEND);
        if (gp.out.k32)
            hipLaunchKernelGGL((espgen::fdrand_part_k<true, true>), grid, block, 0, h->stream, a);
        else if (gp.out.s32)
            hipLaunchKernelGGL((espgen::fdrand_part_k<true, false>), grid, block, 0, h->stream, a);
        else
            hipLaunchKernelGGL((espgen::fdrand_part_k<false, false>), grid, block, 0, h->stream, a);
        sp.add(1);
        h->pre = gp.pre;
        h->pre.valid = false;  // (set below, once the entries are counted in)
        took = true;
    } else {
        CK(prepart_begin(h, E, (i64)grid.x, kind, &ps));
        a.part = ps.out;
    }
    if (ps.on) {
        {
            Span sp(h, ESP_ST_HIST);
            hipLaunchKernelGGL(espgen::fd_count_k, grid, block, 0, h->stream, a, ps.sink, ps.err);
            sp.add(1);
        }
        CK(prepart_rank(h, &ps));
        {
            Span sp(h, ESP_ST_APPEND);
            if (ps.out.k32)
                hipLaunchKernelGGL((espgen::fdrand_part_k<true, true>), grid, block, 0, h->stream, a);
            else if (ps.out.s32)
                hipLaunchKernelGGL((espgen::fdrand_part_k<true, false>), grid, block, 0, h->stream, a);
            else
                hipLaunchKernelGGL((espgen::fdrand_part_k<false, false>), grid, block, 0, h->stream, a);
            sp.add(1);
        }
        CK(prepart_finish(h, &ps, &took));
        if (took) {  // the tables stand until somebody rewrites them: the next identical call starts at the PART launch
            esp_handle::GenPlan &np = h->genplan;
            np.nx = nx, np.ny = ny, np.nz = nz, np.g0 = node_begin, np.g1 = node_end, np.E = E, np.kind = kind;
            np.base = h->win_base, np.span = h->win_span;
            np.keys_at = h->keys.p;
            np.out = ps.out;
            np.pre = h->pre;
            np.valid = true;
        }
    }
    if (!took) {  // stream order (the PART launch left without a store when the stream turned out not to be pre-sorted)
        Span sp(h, ESP_ST_APPEND);
        hipLaunchKernelGGL(espgen::fdrand_k, grid, block, 0, h->stream, a);
        sp.add(1);
    }
    HIPCK(h, hipGetLastError());
    note_kind(h, kind, E);
    h->count += E;
    pending_changed(h);
    if (took) h->pre.valid = true;  // (else: whatever pending_changed left -- an earlier batch with this call as its tail)
    return ESP_OK;
}

extern "C" int32_t esp_generate_fdrand(esp_handle *h, int64_t nx, int64_t ny, int64_t nz, uint64_t seed,
                                       int32_t rand_mode, int32_t kind) {
    if (!h) return ESP_ERR_INVALID;
    return esp_generate_fdrand_range(h, nx, ny, nz, seed, rand_mode, kind, 0, nx * ny * nz);
}



extern "C" int32_t esp_generate_fem(esp_handle *h, int32_t dim, int64_t npd, uint64_t seed, int32_t order_mode) {
    if (!h) return ESP_ERR_INVALID;
    if ((dim != 2 && dim != 3) || npd < 2) FAIL(h, ESP_ERR_INVALID, "fem: dim must be 2 or 3 and npd >= 2");
    const i64 nn = dim == 2 ? npd * npd : npd * npd * npd;
    if (h->m != nn || h->n != nn) FAIL(h, ESP_ERR_INVALID, "Matrix size mismatch");
    (void)hipSetDevice(h->device);
    const i64 q = npd - 1;
    const i64 nc = dim == 2 ? 2 * q * q : 6 * q * q * q;
    const i64 E = nc * (dim + 1) * (dim + 2);
    CK(reserve_append(h, E));
    espgen::FemArgs a;
    a.dim = dim;
    a.npd = npd;
    a.ncells = nc;
    a.seed = seed;
    a.order_mode = order_mode;
    int bits = 2;
    while (((u64)1 << bits) < (u64)nc) bits += 2;
    a.bits = bits;
    espgen::fem_fill_magic(a);
    a.h = 1.0 / (double)(npd - 1);
    a.L = h->L;
    a.keys = (u64 *)h->keys.p + h->count;
    a.vals = (double *)h->vals.p + h->count;
    CK(ensure(h, h->misc, 256));
    const dim3 grid(grid_for(nc, espgen::FEM_CELLS)), block(espgen::FEM_CELLS);
    PartSetup ps;
    CK(prepart_begin(h, E, (i64)grid.x, ESP_RAWUPDATE, &ps));  // (a shuffled cell order fails the COUNT launch's digit limit)
    a.part = ps.out;
    bool took = false;
    if (ps.on) {
        {
            Span sp(h, ESP_ST_HIST);
            hipLaunchKernelGGL(espgen::fem_count_k, grid, block, 0, h->stream, a, ps.sink, ps.err);
            sp.add(1);
        }
        CK(prepart_rank(h, &ps));
        {
            Span sp(h, ESP_ST_APPEND);
            if (ps.out.k32)
                hipLaunchKernelGGL((espgen::fem_part_k<true, true>), grid, block, 0, h->stream, a);
            else if (ps.out.s32)
                hipLaunchKernelGGL((espgen::fem_part_k<true, false>), grid, block, 0, h->stream, a);
            else
                hipLaunchKernelGGL((espgen::fem_part_k<false, false>), grid, block, 0, h->stream, a);
            sp.add(1);
        }
        CK(prepart_finish(h, &ps, &took));
    }
    // a shuffled stream: the producer partitions its ITEMS and stores every update at its bucket position (femitems.hpp)
    if (!took) CK(item_produce_fem(h, a, E, &took));
    if (!took) {
        Span sp(h, ESP_ST_APPEND);
        hipLaunchKernelGGL(espgen::fem_k, grid, block, 0, h->stream, a);
        sp.add(1);
    }
    HIPCK(h, hipGetLastError());
    note_kind(h, ESP_RAWUPDATE, E);
    h->count += E;
    pending_changed(h);
    if (took) h->pre.valid = true, h->lazy.on = h->lazy.armed;  // (else: whatever pending_changed left -- an earlier batch with this call as its tail)
    h->lazy.armed = false;
    return ESP_OK;
}


// ---- producer-side partition: host side (the kernels: runpart.hpp "the append IS the partition") ----------
// prepart_begin: a device-side producer is about to append E entries in `chunks` chunks (= its workgroups, at most
// esprun::TILE entries each) -- kind >= 0: all of that kind.  ps->on = true when the append can be the partition:
// empty buffer, a prefix of 9..20 bits that brings the buckets under the bucket kernel's capacity without reaching
// into the row bits, handle not driven through esp_shard_* (its flush partitions by owner first).  Clears the tables.
// force_path 16: never.
int32_t prepart_begin(esp_handle *h, i64 E, i64 chunks, int kind, PartSetup *ps) {
    ps->on = false;
    ps->fb = 0;
    memset(&ps->out, 0, sizeof ps->out);
    if (h->count != 0 || E <= esplocal::CAP || chunks >= ((i64)1 << 38)) return ESP_OK;
    if (h->force_path == ESP_PATH_GENERAL || h->force_path == ESP_PATH_NO_RUN_PARTITION || h->force_path == ESP_PATH_RUN_LIST_BY_RADIX || h->force_path == ESP_PATH_PRODUCER_STREAM_ORDER) return ESP_OK;
    if (h->runs_skip > 0) return ESP_OK;  // (the handle's last streams were not pre-sorted: back-off, see sort_msd)
    int K, pb, shift;
    i64 NB;
    double Ee = 0.0;
    MwPlan mw;
    if (h->shard_user) {
        // a shard: partition by (owner, digit inside the owner's column range) -- what the next esp_shard_partition
        // would do in a pass of its own -- if the caller announced that call (esp_shard_plan)
        if (!h->shard_plan.valid || h->force_path == ESP_PATH_SHARD_NOT_APPLICABLE) return ESP_OK;
        const int P = h->shard_plan.P;
        if (P > esprun::MW_MAX || P > esplocal::MAX_PIECES || (double)h->n * (double)P >= 9.0e18) return ESP_OK;
        mw = shard_mw_plan(h, P, h->shard_plan.eps);
        if (!mw.ok || mw.shift < h->L.rb) return ESP_OK;
        K = mw.K, pb = mw.pb, shift = mw.shift, NB = mw.NB;
        // (FINE partition: 2^fb buckets per digit of the plan -- 4-byte keys for the own range; force_path 41 / 14: never)
        if (mw.fb > 0 && kind >= 0 && h->force_path != ESP_PATH_NO_FINE_PARTITION && h->force_path != ESP_PATH_PACKED_KEYS) {
            ps->fb = mw.fb;
            pb += mw.fb, shift -= mw.fb, NB <<= mw.fb;
        }
        const size_t o_cnt = 256 * 8;  // (the table layout of esp_shard_partition: window bases | owner offsets | counts)
        CK(ensure(h, h->parttab, o_cnt + sizeof(i64) * (size_t)(NB + 1)));
        // (the copy is asynchronous: its source lives in the handle, not in this function's frame -- and a flush that
        // still reads an earlier table from the same vector has been synchronised by then: every flush ends in a wait)
        if (!h->pin_mw) {
            HIPCK(h, hipHostMalloc((void **)&h->pin_mw, sizeof(u64) * esprun::MW_MAX, hipHostMallocDefault));
            HIPCK(h, hipEventCreateWithFlags(&h->pin_mw_done, hipEventDisableTiming));
        } else {
            HIPCK(h, hipEventSynchronize(h->pin_mw_done));  // (the previous upload has read the buffer)
        }
        memcpy(h->pin_mw, mw.base.data(), sizeof(u64) * (size_t)P);
        HIPCK(h, hipMemcpyAsync(h->parttab.p, h->pin_mw, sizeof(u64) * (size_t)P, hipMemcpyHostToDevice, h->stream));
        HIPCK(h, hipEventRecord(h->pin_mw_done, h->stream));
    } else {
        K = window_bits(h);
        pb = plan_prefix_bits(h, E, K, &Ee);
        shift = K - pb;
        if (pb <= 8 || pb > 20 || shift < h->L.rb || shift > esplocal::MAX_REM_BITS) return ESP_OK;
        // FINE partition: 33 .. 36 key bits below the planned prefix (a stencil above 256^3) and one known kind -- up to four
        // more prefix bits make the rest fit 4-byte keys; the flush's bucket kernel still takes the PLANNED segments (2^fb
        // buckets each: its time follows the number of segments, NOTES/round5.md section 9).  force_path 41 / 14: never.
        if (kind >= 0 && shift > 32 && shift - 32 <= 4 && pb + (shift - 32) <= esprun::MAX_PB && shift - (shift - 32) >= h->L.rb &&
            h->force_path != ESP_PATH_NO_FINE_PARTITION && h->force_path != ESP_PATH_PACKED_KEYS) {
            ps->fb = shift - 32;
            pb += ps->fb;
            shift = 32;
        }
        NB = (i64)1 << pb;
    }
    ChunkArrays ca;
    CK(chunk_arrays(h, chunks + 64, pb, &ca));
    CK(ensure(h, h->runbuf, sizeof(i64) * (size_t)chunks * esprun::RMAX));
    CK(ensure(h, h->seg[1], sizeof(i64) * (size_t)(NB + 1)));
    CK(ensure(h, h->misc, 256));
    CK(aux_ready(h));
    unsigned long long *d_maxlen = (unsigned long long *)h->misc.p + 24;
    u32 *flags = (u32 *)h->misc.p + 60;
    HIPCK(h, hipMemsetAsync(d_maxlen, 0, 64, h->stream));  // maxlen .. the four flag words
    HIPCK(h, hipMemsetAsync(ca.dcount, 0, ca.clear_bytes, h->stream));
    ps->sink = esprun::RunSink{ca.runs_d, ca.runs_c, ca.nruns, ca.bucket_count, flags + 1, ca.dcount, ca.dlist};
    ps->err = flags;
    ps->K = K;
    ps->pb = pb;
    ps->kind = kind;
    ps->E = E;
    ps->chunks = chunks;
    ps->Ee = Ee;
    ps->NB = NB;
    if (mw.ok) {
        ps->mw_P = h->shard_plan.P;
        ps->mw_me = h->shard_plan.me;
        ps->mw_shift = shift;                        // (the tables' digits: the plan's, or 2^fb times finer)
        ps->mw_nb = (u32)(mw.nb64 << ps->fb);
        ps->mw_eps = h->shard_plan.eps;
    }
    ps->seg_out = (i64 *)h->seg[1].p;
    ps->runs_off = (i64 *)h->runbuf.p;
    ps->bucket_count = ca.bucket_count;
    ps->coarse = ca.coarse;
    ps->dcount = ca.dcount;
    ps->dlist = ca.dlist;
    esprun::PartOut &o = ps->out;
    o.runs_d = ca.runs_d;
    o.runs_off = ps->runs_off;
    o.nruns = ca.nruns;
    o.flags = flags;
    o.maxlen = d_maxlen;
    o.cap = esplocal::CAP;
    // (a shard's ranges travel to other ranks as packed keys: 4-byte keys only without windows)
    o.k32 = (!mw.ok && kind >= 0 && h->force_path != ESP_PATH_PACKED_KEYS && shift <= 32) ? 1 : 0;
    o.s32 = shift <= 32 ? 1 : 0;
    // (a shard's own range never leaves the GPU: 4-byte keys there -- force_path 14: packed keys everywhere)
    o.own32 = (mw.ok && kind >= 0 && h->force_path != ESP_PATH_PACKED_KEYS && shift <= 32) ? 1 : 0;
    o.mw_me = ps->mw_me;
    o.own_lo = (const i64 *)h->seg[1].p + (size_t)ps->mw_me * (size_t)ps->mw_nb;
    o.mw_P = mw.ok ? ps->mw_P : 0;
    o.mw_nb = ps->mw_nb;
    o.mw_base = (const u64 *)h->parttab.p;
    if (mw.ok) {
        o.mw_own_base = mw.base[(size_t)ps->mw_me];
        o.mw_own_width = ps->mw_me + 1 < ps->mw_P ? mw.base[(size_t)ps->mw_me + 1] - o.mw_own_base : ~0ull - o.mw_own_base;
    }
    o.shift = shift;
    o.base = h->win_base;
    o.span = h->win_span;
    o.keys_out = (u64 *)h->keys.p;
    o.vals_out = (double *)h->vals.p;
    o.chunk_base = 0;
    ps->on = true;
    return ESP_OK;
}
// between the COUNT and the PART launch: bucket starts and run offsets (the two ranking launches of run_partition)
int32_t prepart_rank(esp_handle *h, PartSetup *ps) {
    const i64 NB = ps->NB;
    unsigned long long *d_maxlen = (unsigned long long *)h->misc.p + 24;
    u32 *flags = (u32 *)h->misc.p + 60;
    {
        Span sp(h, ESP_ST_SCAN);
        const unsigned g = (unsigned)grid_for(NB + 1, esprun::THREADS);
        hipLaunchKernelGGL(esprun::run_coarse_k, dim3(g), dim3(esprun::THREADS), 0, h->stream, ps->bucket_count, NB, ps->coarse);
        hipLaunchKernelGGL(esprun::run_rank_k, dim3(g), dim3(esprun::THREADS), 0, h->stream, ps->bucket_count, (const u64 *)ps->coarse,
                           ps->dcount, ps->dlist, NB, ps->seg_out, ps->runs_off, d_maxlen, flags + 3);
        sp.add(2);
        if (ps->fb > 0) {  // (word 1 of the block: the longest SEGMENT of the flush, 2^fb buckets)
            const i64 Sc = NB >> ps->fb;
            hipLaunchKernelGGL(coarse_seg_max_k, dim3(grid_for(Sc, 256)), dim3(256), 0, h->stream, (const i64 *)ps->seg_out, Sc, ps->fb, d_maxlen + 1);
            sp.add(1);
        }
    }
    // (the host reads the flags while the PART launch runs)
    hipLaunchKernelGGL(publish_block_k, dim3(1), dim3(64), 0, h->stream, (const unsigned long long *)d_maxlen, h->pin_scalar);
    HIPCK(h, hipEventRecord(h->aux_ev, h->stream));
    return ESP_OK;
}
// after the PART launch was issued: *took = false when it left without a store (window error, a chunk with too many
// digits, a digit with too many runs: the caller issues the plain producer); else the handle's buffer is bucket-ordered
int32_t prepart_finish(esp_handle *h, PartSetup *ps, bool *took) {
    *took = false;
    HIPCK(h, hipEventSynchronize(h->aux_ev));  // (publish_block_k has written the block to pin_scalar)
    const u32 f_err = (u32)h->pin_scalar[6], f_over = (u32)(h->pin_scalar[6] >> 32), f_many = (u32)(h->pin_scalar[7] >> 32);
    if (f_err | f_over | f_many) {
        // (flags[0] stays set for nobody: the plain producer follows and the flush's own partition checks the window)
        HIPCK(h, hipMemsetAsync((u32 *)h->misc.p + 60, 0, 16, h->stream));
        if (f_over | f_many) {  // not a pre-sorted stream: neither this handle's producers nor its next flushes try again soon
            h->runs_penalty = std::min(16, 2 * h->runs_penalty + 1);
            h->runs_skip = h->runs_penalty + 1;  // (+1: the flush of this very batch)
        }
        return ESP_OK;
    }
    if (ps->out.k32 && (i64)h->pin_scalar[0] > (i64)esplocal::CAP) return ESP_OK;  // (the K32 launch left without a store)
    esp_handle::PrePart &pp = h->pre;
    pp.K = ps->K;
    pp.pb = ps->pb;
    pp.maxlen = (i64)h->pin_scalar[0];
    pp.key_bytes = ps->out.k32 ? 4 : 8;
    // (a fine partition whose joined segments outgrow the bucket kernel is flushed bucket by bucket, like any other batch)
    pp.maxlen_c = ps->fb > 0 ? (i64)h->pin_scalar[1] : 0;
    pp.fb = (ps->fb > 0 && ps->out.k32 && pp.maxlen_c <= (i64)esplocal::CAP) ? ps->fb : 0;
    if (ps->mw_P > 0) pp.fb = ps->fb;  // (a shard: the plan's digits are what the exchange speaks, whatever their longest one is)
    pp.kind = ps->kind;
    pp.E = ps->E;
    pp.tail = 0;
    pp.base = h->win_base;
    pp.span = h->win_span;
    pp.Ee = ps->Ee;
    pp.mw_P = ps->mw_P;
    pp.mw_me = ps->mw_me;
    pp.mw_shift = ps->mw_shift;
    pp.mw_nb = ps->mw_nb;
    pp.mw_eps = ps->mw_eps;
    pp.own32 = ps->out.own32 != 0;
    pp.plan_id = ++h->plan_counter;
    *took = true;  // (the caller sets pre.valid once the entries are counted in)
    return ESP_OK;
}
// A bucket-ordered pending buffer is a valid pending buffer -- a stable permutation of the stream -- once its keys are
// packed keys again: every call that reads or extends the pending entries other than the flush they were written for
int32_t pending_materialize(esp_handle *h) {
    if (!h->pre.valid) return ESP_OK;
    CK(lazy_expand(h));  // (a batch still held as sorted items: its updates first)
    h->pre.valid = false;
    h->part_own32 = false;
    if (h->pre.mw_P > 0 && h->pre.own32 && h->count > 0) {
        // a shard's batch: the own range holds 4-byte keys -> packed keys; the other ranges are copied as they are
        const esp_handle::PrePart &pp = h->pre;
        const i64 nb = (i64)pp.mw_nb, d0 = (i64)pp.mw_me * nb;
        CK(ensure(h, h->keys2, std::max(h->keys.bytes, sizeof(u64) * (size_t)h->count)));
        std::vector<i64> lohi(2);
        HIPCK(h, hipMemcpyAsync(&lohi[0], (const i64 *)h->seg[1].p + d0, sizeof(i64), hipMemcpyDeviceToHost, h->stream));
        HIPCK(h, hipMemcpyAsync(&lohi[1], (const i64 *)h->seg[1].p + d0 + nb, sizeof(i64), hipMemcpyDeviceToHost, h->stream));
        HIPCK(h, hipStreamSynchronize(h->stream));
        const i64 lo = lohi[0], hi_ = lohi[1], E = pp.E;
        Span sp(h, ESP_ST_COPY);
        if (lo > 0) HIPCK(h, hipMemcpyAsync(h->keys2.p, h->keys.p, sizeof(u64) * (size_t)lo, hipMemcpyDeviceToDevice, h->stream));
        if (E > hi_)
            HIPCK(h, hipMemcpyAsync((u64 *)h->keys2.p + hi_, (const u64 *)h->keys.p + hi_, sizeof(u64) * (size_t)(E - hi_), hipMemcpyDeviceToDevice, h->stream));
        const u64 base = (u64)shard_col0(h->n, pp.mw_P, pp.mw_me) << h->L.rb;
        hipLaunchKernelGGL(esprun::expand_own_keys_k, dim3((unsigned)nb), dim3(esprun::THREADS), 0, h->stream, (const u64 *)h->keys.p,
                           (const i64 *)h->seg[1].p, d0, pp.mw_shift, base, (u32)pp.kind, (u64 *)h->keys2.p);
        sp.add(3);
        HIPCK(h, hipGetLastError());
        std::swap(h->keys, h->keys2);
        h->cap = (i64)std::min(h->keys.bytes / sizeof(u64), h->vals.bytes / sizeof(double));
        return ESP_OK;
    }
    if (h->pre.key_bytes != 4 || h->count == 0) return ESP_OK;
    const esp_handle::PrePart &pp = h->pre;
    CK(ensure(h, h->keys2, std::max(h->keys.bytes, sizeof(u64) * (size_t)h->count)));
    Span sp(h, ESP_ST_COPY);
    hipLaunchKernelGGL(esprun::expand_keys_k, dim3(esprun::expand_keys_grid((i64)1 << pp.pb)), dim3(esprun::THREADS), 0, h->stream, (const u32 *)h->keys.p,
                       (const i64 *)h->seg[1].p, pp.K - pp.pb, pp.base, (u32)pp.kind, (u64 *)h->keys2.p, (i64)1 << pp.pb);
    sp.add(1);
    if (h->count > pp.E) {  // (the packed entries behind the batch)
        HIPCK(h, hipMemcpyAsync((u64 *)h->keys2.p + pp.E, (const u64 *)h->keys.p + pp.E, sizeof(u64) * (size_t)(h->count - pp.E),
                                hipMemcpyDeviceToDevice, h->stream));
        sp.add(1);
    }
    HIPCK(h, hipGetLastError());
    std::swap(h->keys, h->keys2);
    h->cap = (i64)std::min(h->keys.bytes / sizeof(u64), h->vals.bytes / sizeof(double));
    return ESP_OK;
}


// ---- batches held as sorted items (esp_handle::LazyItems, group3_items.hpp) -----------------------------------------
// The fused bucket kernel serves a FRESH matrix whose flush would take group3_k in its plain form: additions of one kind
// (UPDATE / RAWUPDATE), cells of 3 or 4 nodes, rows below 2^30.  Whether the matrix is still fresh at flush time nobody knows
// here; a handle that already holds a matrix assembles over it (re-assembly kernels read expanded entries), so its batches
// are expanded at once.  force_path 39 (ESP_PATH_NO_LAZY_ITEMS): never; any other forced path but 36: never either.
bool lazy_items_wanted(const esp_handle *h, int kind) {
    if (h->force_path != ESP_PATH_AUTO && h->force_path != ESP_PATH_LATE_TOTAL) return false;
    if (kind != ESP_UPDATE && kind != ESP_RAWUPDATE) return false;
    if (h->count != 0 || windowed(h) || h->shard_user) return false;
    // (over a stored pattern: only when the handle's last flush over it hit -- the re-assembly form of the fused kernel)
    if (h->nnz != 0 && !(h->seen_hits && !h->hits_off)) return false;
    if (h->g3_off || h->g3_wide) return false;  // (the handle's segments are not group3_k's plain form's)
    if (h->L.rb > 30) return false;
    return true;
}
// The expansion that was put off: sorted items (in the keys array) -> updates, bucket by bucket, into the scratch pair; then
// the pairs trade places.  Afterwards the handle is what item_produce_fem / elements_by_items used to leave.
int32_t lazy_expand(esp_handle *h) {
    if (!h->lazy.on) return ESP_OK;
    h->lazy.on = false;
    if (!h->pre.valid) return ESP_OK;
    const i64 E = h->pre.E;
    CK(ensure(h, h->keys2, std::max(h->keys.bytes, sizeof(u64) * (size_t)E)));
    CK(ensure(h, h->vals2, std::max(h->vals.bytes, sizeof(double) * (size_t)E)));
    {
        Span sp(h, ESP_ST_APPEND);
        if (h->lazy.src == 1) {
            espitem::Args a = h->lazy.it;
            a.keys_out = (u64 *)h->keys2.p;
            a.vals_out = (double *)h->vals2.p;
            const dim3 grid(grid_for(a.nitems, espitem::THREADS)), block(espitem::THREADS);
            if (h->lazy.k32)
                hipLaunchKernelGGL(espitem::fem_expand_k<true>, grid, block, 0, h->stream, a);
            else
                hipLaunchKernelGGL(espitem::fem_expand_k<false>, grid, block, 0, h->stream, a);
        } else {
            espelem::Args a = h->lazy.el;
            a.keys_out = (u64 *)h->keys2.p;
            a.vals_out = (double *)h->vals2.p;
            if (h->lazy.k32)
                espelem::launch_expand<true>(a, h->stream);
            else
                espelem::launch_expand<false>(a, h->stream);
        }
        sp.add(1);
    }
    HIPCK(h, hipGetLastError());
    std::swap(h->keys, h->keys2);
    std::swap(h->vals, h->vals2);
    h->cap = (i64)std::min(h->keys.bytes / sizeof(u64), h->vals.bytes / sizeof(double));
    return ESP_OK;
}

// Shuffled FEM stream on an empty buffer (femitems.hpp): item records -> the flush's own partition passes over them ->
// every update stored once at its bucket position; the handle is left as after a producer-side partition (h->pre).
// *took = false: not applicable, the caller appends in stream order.
int32_t item_produce_fem(esp_handle *h, const espgen::FemArgs &fa, i64 E, bool *took) {
    *took = false;
    if (h->count != 0 || E <= esplocal::CAP || windowed(h) || h->shard_user) return ESP_OK;
    // (test hooks that pin another path: 2 general, 5 / 12 / 16 partition flavours, 19 plain pending buffer, 25 this one off)
    if (h->force_path == ESP_PATH_GENERAL || h->force_path == ESP_PATH_NO_RUN_PARTITION || h->force_path == ESP_PATH_RUN_LIST_BY_RADIX || h->force_path == ESP_PATH_PRODUCER_STREAM_ORDER || h->force_path == ESP_PATH_NO_BATCH_TAIL || h->force_path == ESP_PATH_NO_ITEM_PARTITION)
        return ESP_OK;
    const int W = fa.dim + 2, ni = fa.dim + 1;
    const i64 NI = fa.ncells * ni;
    if (NI >= 0xFFFFFFF0ll || (fa.ncells >> 40) != 0 || 2 * NI > E) return ESP_OK;
    // two ping-pong pairs of item records inside the flush's scratch pair (sized for E updates: E / W items each)
    CK(ensure(h, h->keys2, sizeof(u64) * (size_t)E));
    CK(ensure(h, h->vals2, sizeof(double) * (size_t)E));
    espitem::Args a;
    a.fem = fa;
    a.nitems = NI;
    a.ikeys = (u64 *)h->keys2.p;
    a.ivals = (double *)h->vals2.p;
    // single-word records when the cell's number fits below the column bits of the key (28: test hook, never)
    a.single = ((u64)fa.ncells <= ((u64)1 << (h->L.rb + ESP_TAG_BITS)) && h->force_path != ESP_PATH_TWO_WORD_ITEMS) ? 1 : 0;
    a.sorted_keys = nullptr;
    // the batch stays a list of sorted items and the flush's bucket kernel forms the updates (group3_items.hpp): the item
    // records then ping-pong inside the KEYS array (the pending buffer the updates would fill), the scratch pair stays free
    // for the flush's output
    const bool lazy = a.single && lazy_items_wanted(h, ESP_RAWUPDATE) && h->keys.bytes >= 2 * sizeof(u64) * (size_t)NI;
    if (lazy) a.ikeys = (u64 *)h->keys.p;
    const int K = window_bits(h);
    // single-word records: the passes may stop a few bits early, the expansion orders every segment by the last bits itself
    // (segexpand.hpp); a second attempt with the passes alone when a segment does not fit that
    int sort_bits = 0;
    const int lbits0 = (a.single && !lazy) ? plan_local_bits(h, NI, W, K, &sort_bits) : 0;
    Sorted st;
    int lbits = 0;
    i64 maxlen_updates = 0;
    bool done = false;
    for (int attempt = lbits0 > 0 ? 0 : 1; attempt < 2 && !done; attempt++) {
        lbits = attempt == 0 ? lbits0 : 0;
        {
            Span sp(h, ESP_ST_APPEND);
            // (a shuffled order whose permutation domain is much larger than the mesh: the waves walk it together)
            if (fa.order_mode != 0 && fa.ncells >= 2 && ((u64)1 << fa.bits) > (u64)fa.ncells + (u64)fa.ncells / 2)
                hipLaunchKernelGGL(espitem::fem_items_walk_k, dim3(grid_for(fa.ncells, espitem::THREADS * espitem::ITEM_CELLS)), dim3(espitem::THREADS), 0, h->stream, a);
            else
                hipLaunchKernelGGL(espitem::fem_items_k, dim3(grid_for(fa.ncells, espitem::THREADS)), dim3(espitem::THREADS), 0, h->stream, a);
            sp.add(1);
        }
        // the flush's partition over the items: a temporary view of the handle (sort_msd reads count, keys/vals, keys2/vals2)
        const DevBuf k0 = h->keys, v0 = h->vals, k2 = h->keys2, v2 = h->vals2;
        const i64 count0 = h->count;
        const double spread0 = h->seen_spread;
        h->keys.p = a.ikeys, h->keys.bytes = sizeof(u64) * (size_t)NI;
        h->vals.p = a.ivals, h->vals.bytes = sizeof(double) * (size_t)NI;
        h->keys2.p = a.ikeys + NI, h->keys2.bytes = sizeof(u64) * (size_t)NI;
        h->vals2.p = a.ivals + NI, h->vals2.bytes = sizeof(double) * (size_t)NI;
        h->count = NI;
        h->plan_cap = lbits ? (i64)espseg::LCAP : (i64)esplocal::CAP / W;
        h->plan_bits = lbits ? sort_bits : 0;
        h->item_mode = true;
        h->item_keys_only = a.single != 0;
        st = Sorted();
        const int32_t rc = sort_msd(h, &st);
        h->keys = k0, h->vals = v0, h->keys2 = k2, h->vals2 = v2;
        h->count = count0;
        h->plan_cap = 0;
        h->plan_bits = 0;
        h->item_mode = false;
        h->item_keys_only = false;
        if (rc != ESP_OK) return rc;
        if (lbits) lbits = std::min(lbits, st.rem_bits - h->L.rb);  // (the sub-segments are whole columns)
        const bool usable = st.local_ok && st.S >= 2 && st.rem_bits - std::max(lbits, 0) >= h->L.rb &&
                            (attempt == 0 ? lbits >= 1 : st.maxlen * W <= (i64)esplocal::CAP);
        if (!usable) {
            h->seen_spread = spread0;
            if (attempt == 1) return ESP_OK;  // (no segment table the bucket kernel takes: the plain producer and the flush's own passes)
            continue;
        }
        const int rem_final = st.rem_bits - lbits;
        const bool k32 = rem_final <= 32 && h->force_path != ESP_PATH_PACKED_KEYS;
        a.sorted = st.sv;
        a.sorted_keys = st.sk;
        a.rem_bits = rem_final;
        a.base = h->win_base;
        a.keys_out = (u64 *)h->keys.p;
        a.vals_out = (double *)h->vals.p;
        const i64 S_final = (i64)st.S << lbits;
        if (lbits && st.seg_start == (const i64 *)h->seg[1].p) {
            // (the passes left their table in the array the sub-segment table goes to -- which grows, and is written while the
            // coarse entries are still read: the coarse table moves aside first.  Found with released buffers poisoned.)
            CK(ensure(h, h->segout, sizeof(i64) * (size_t)(st.S + 1)));
            HIPCK(h, hipMemcpyAsync(h->segout.p, st.seg_start, sizeof(i64) * (size_t)(st.S + 1), hipMemcpyDeviceToDevice, h->stream));
            HIPCK(h, hipStreamSynchronize(h->stream));
            st.seg_start = (const i64 *)h->segout.p;
        }
        CK(ensure(h, h->seg[1], sizeof(i64) * (size_t)(S_final + 1)));
        if (lbits) {
            unsigned long long *d_maxsub = (unsigned long long *)h->misc.p + 2;
            HIPCK(h, hipMemsetAsync(d_maxsub, 0, 8, h->stream));
            espseg::SegArgs sa;
            sa.recs = st.sk;
            sa.seg_start = st.seg_start;
            sa.S = st.S;
            sa.W = W;
            sa.lbits = lbits;
            sa.lshift = rem_final;
            sa.base = h->win_base;
            sa.sub_start = (i64 *)h->seg[1].p;
            sa.maxsub = d_maxsub;
            sa.total_items = NI;
            {
                Span sp(h, ESP_ST_APPEND);
                if (k32)
                    hipLaunchKernelGGL(espitem::fem_seg_expand_k<true>, dim3((unsigned)st.S), dim3(espseg::THREADS), 0, h->stream, a, sa);
                else
                    hipLaunchKernelGGL(espitem::fem_seg_expand_k<false>, dim3((unsigned)st.S), dim3(espseg::THREADS), 0, h->stream, a, sa);
                sp.add(1);
            }
            HIPCK(h, hipMemcpyAsync(h->pin_scalar, d_maxsub, 8, hipMemcpyDeviceToHost, h->stream));
            HIPCK(h, hipStreamSynchronize(h->stream));
            maxlen_updates = (i64)h->pin_scalar[0];
            if (maxlen_updates > (i64)esplocal::CAP) {  // (a sub-segment the bucket kernel does not take: the passes alone)
                h->seen_spread = spread0;
                continue;
            }
        } else {
            Span sp(h, ESP_ST_APPEND);
            const dim3 grid(grid_for(NI, espitem::THREADS)), block(espitem::THREADS);
            if (lazy) {  // (the expansion is the flush's business now -- or lazy_expand's)
                h->lazy.src = 1;
                h->lazy.k32 = k32;
                h->lazy.it = a;
            } else if (k32)
                hipLaunchKernelGGL(espitem::fem_expand_k<true>, grid, block, 0, h->stream, a);
            else
                hipLaunchKernelGGL(espitem::fem_expand_k<false>, grid, block, 0, h->stream, a);
            hipLaunchKernelGGL(espitem::scale_segments_k, dim3(grid_for((i64)st.S + 1, 256)), dim3(256), 0, h->stream, st.seg_start, (i64)st.S + 1,
                               (i64)W, (i64 *)h->seg[1].p);
            sp.add(2);
            maxlen_updates = st.maxlen * W;
        }
        done = true;
    }
    if (!done) return ESP_OK;
    h->lazy.armed = lazy;  // (-> lazy.on beside pre.valid, once the caller has counted the entries in)
    st.rem_bits -= lbits;
    HIPCK(h, hipGetLastError());
    esp_handle::PrePart &pp = h->pre;
    pp.K = K;
    pp.pb = K - st.rem_bits;
    pp.maxlen = maxlen_updates;
    pp.key_bytes = (st.rem_bits <= 32 && h->force_path != ESP_PATH_PACKED_KEYS) ? 4 : 8;
    pp.kind = ESP_RAWUPDATE;
    pp.E = E;
    pp.tail = 0;
    pp.base = h->win_base;
    pp.span = h->win_span;
    pp.Ee = plan_entries(E, K, h->win_span);
    pp.mw_P = 0, pp.mw_me = 0, pp.mw_shift = 0, pp.mw_nb = 0, pp.mw_eps = 0;
    pp.own32 = false;
    pp.fb = 0, pp.maxlen_c = 0;
    *took = true;  // (the caller sets pre.valid once the entries are counted in)
    return ESP_OK;
}

