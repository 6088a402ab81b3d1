// partition.hip -- libesparse_hip: the plan, the radix passes, the run-based single pass, sort_msd (see internal.hpp for the map of the translation units)
#include "internal.hpp"

// ------------------------------------------------------------------------ sort
// One stable partition pass over S segments (device arrays seg_start/tile_first).
// max_tiles bounds the grid; the scanned histogram stays in h->hist.
int32_t partition_pass(esp_handle *h, espradix::Pass &p, i64 max_tiles) {
    const int R = 1 << p.bits;
    const i64 hn = max_tiles * R;
    const size_t hist_bytes = sizeof(u64) * (size_t)(hn + espscan::workspace_elems(hn));
    CK(ensure(h, h->hist, hist_bytes));
    p.hist = (u64 *)h->hist.p;
    HIPCK(h, hipMemsetAsync(p.hist, 0, sizeof(u64) * (size_t)hn, h->stream));
    {
        Span sp(h, ESP_ST_HIST);
        if (p.raw_cols)
            hipLaunchKernelGGL(espradix::tile_hist_raw_k, dim3(espradix::scatter_grid(max_tiles)), dim3(espradix::THREADS), 0, h->stream, p);
        else if (p.k32_in)
            hipLaunchKernelGGL(espradix::tile_hist32_k, dim3(espradix::scatter_grid(max_tiles)), dim3(espradix::THREADS), 0, h->stream, p);
        else
            hipLaunchKernelGGL(espradix::tile_hist_k, dim3(espradix::scatter_grid(max_tiles)), dim3(espradix::THREADS), 0, h->stream, p);
        sp.add(1);
    }
    {
        Span sp(h, ESP_ST_SCAN);
        sp.add(espscan::exclusive<u64, false>(h->stream, p.hist, p.hist, hn, p.hist + hn));
    }
    {
        Span sp(h, ESP_ST_SCATTER);
        if (p.raw_cols && p.bits > 8)
            hipLaunchKernelGGL((espradix::scatter_k<true, false, true>), dim3(espradix::scatter_grid(max_tiles)), dim3(espradix::THREADS), 0, h->stream, p);
        else if (p.raw_cols)
            hipLaunchKernelGGL((espradix::scatter_k<false, false, true>), dim3(espradix::scatter_grid(max_tiles)), dim3(espradix::THREADS), 0, h->stream, p);
        else if (p.keys_only && p.bits > 8)
            hipLaunchKernelGGL((espradix::scatter_k<true, true>), dim3(espradix::scatter_grid(max_tiles)), dim3(espradix::THREADS), 0, h->stream, p);
        else if (p.keys_only)
            hipLaunchKernelGGL((espradix::scatter_k<false, true>), dim3(espradix::scatter_grid(max_tiles)), dim3(espradix::THREADS), 0, h->stream, p);
        else if (p.k32_out && p.bits > 8 && p.k32_in)
            hipLaunchKernelGGL((espradix::scatter_k<true, false, false, 2>), dim3(espradix::scatter_grid(max_tiles)), dim3(espradix::THREADS), 0, h->stream, p);
        else if (p.k32_out && p.bits > 8)
            hipLaunchKernelGGL((espradix::scatter_k<true, false, false, 1>), dim3(espradix::scatter_grid(max_tiles)), dim3(espradix::THREADS), 0, h->stream, p);
        else if (p.k32_out && p.k32_in)
            hipLaunchKernelGGL((espradix::scatter_k<false, false, false, 2>), dim3(espradix::scatter_grid(max_tiles)), dim3(espradix::THREADS), 0, h->stream, p);
        else if (p.k32_out)
            hipLaunchKernelGGL((espradix::scatter_k<false, false, false, 1>), dim3(espradix::scatter_grid(max_tiles)), dim3(espradix::THREADS), 0, h->stream, p);
        else if (p.bits > 8)
            hipLaunchKernelGGL((espradix::scatter_k<true>), dim3(espradix::scatter_grid(max_tiles)), dim3(espradix::THREADS), 0, h->stream, p);
        else
            hipLaunchKernelGGL((espradix::scatter_k<false>), dim3(espradix::scatter_grid(max_tiles)), dim3(espradix::THREADS), 0, h->stream, p);
        sp.add(1);
    }
    return ESP_OK;
}

// full stable LSD sort of the pending entries on their (col,row) bits.  Result in *sk/*sv.
int32_t sort_pending_lsd(esp_handle *h, const u64 **sk, const double **sv) {
    const i64 E = h->count;
    CK(ensure(h, h->keys2, sizeof(u64) * (size_t)E));
    CK(ensure(h, h->vals2, sizeof(double) * (size_t)E));
    CK(ensure(h, h->segs, sizeof(i64) * 8));
    CK(ensure(h, h->misc, 256));
    const i64 T = ceil_div<i64>(E, espradix::TILE);
    i64 *segs = (i64 *)h->segs.p;
    hipLaunchKernelGGL(set_i64_k, dim3(1), dim3(1), 0, h->stream, segs, (i64)0, E, (i64)0, T);
    u64 *kin = (u64 *)h->keys.p, *kout = (u64 *)h->keys2.p;
    double *vin = (double *)h->vals.p, *vout = (double *)h->vals2.p;
    const int K = h->L.sort_bits();
    for (int done = 0; done < K; done += 8) {
        espradix::Pass p;
        p.keys_in = kin;
        p.vals_in = vin;
        p.keys_out = kout;
        p.vals_out = vout;
        p.seg_start = segs;
        p.tile_first = segs + 2;
        p.S = 1;
        p.owner_P = 0;
        p.owner_n = 1;
        p.colshift = 0;
        p.base = 0;
        p.span = ~0ull;
        p.err = (u32 *)h->misc.p + 60;
        p.shift = done;
        p.bits = std::min(8, K - done);
        CK(partition_pass(h, p, T));
        std::swap(kin, kout);
        std::swap(vin, vout);
    }
    HIPCK(h, hipGetLastError());
    *sk = kin;
    *sv = vin;
    // make keys/vals the scratch pair for the caller: after an odd number of passes the sorted
    // data lives in keys2/vals2
    if (kin != (u64 *)h->keys.p) {
        std::swap(h->keys, h->keys2);
        std::swap(h->vals, h->vals2);
        // capacities may differ: keep cap consistent with the smaller of the two
        h->cap = (i64)std::min(h->keys.bytes / sizeof(u64), h->vals.bytes / sizeof(double));
    }
    return ESP_OK;
}



int32_t chunk_arrays(esp_handle *h, i64 Ccap, int pb, ChunkArrays *out, bool keep_plan) {
    if (!keep_plan) h->rawplan.valid = false;  // (whoever asks for the run tables is about to rewrite them)
    h->genplan.valid = false;
    const i64 NB = (i64)1 << pb;
    const i64 RM = Ccap * esprun::RMAX;
    size_t off = 0;
    auto carve = [&](size_t bytes) {
        size_t o = off;
        off += (bytes + 255) & ~(size_t)255;
        return o;
    };
    const size_t o_rd = carve(sizeof(u32) * (size_t)RM), o_rc = carve(sizeof(u32) * (size_t)RM);
    const size_t o_nr = carve(sizeof(u64) * (size_t)(Ccap + 1 + espscan::workspace_elems(Ccap + 1)));
    const size_t o_dc = carve(sizeof(u32) * (size_t)NB);
    const size_t o_bc = carve(sizeof(u64) * (size_t)(NB + 1 + espscan::workspace_elems(NB + 1)));
    const size_t o_ov = carve(64);
    const size_t o_dl = carve(sizeof(u64) * (size_t)NB * esprun::DCAP);
    const size_t o_co = carve(sizeof(u64) * (size_t)(NB / 256 + 2));
    if (h->chunkbuf.bytes < off || h->chunk_cap != Ccap || h->chunk_pb != pb) {
        CK(ensure(h, h->chunkbuf, off));
        h->chunk_cap = Ccap;
        h->chunk_pb = pb;
    }
    char *B = (char *)h->chunkbuf.p;
    out->runs_d = (u32 *)(B + o_rd);
    out->runs_c = (u32 *)(B + o_rc);
    out->nruns = (u64 *)(B + o_nr);
    out->bucket_count = (unsigned long long *)(B + o_bc);
    out->overflow = (u32 *)(B + o_ov);
    out->dcount = (u32 *)(B + o_dc);
    out->dlist = (u64 *)(B + o_dl);
    out->coarse = (u64 *)(B + o_co);
    // (a multiple of 256 bytes -- the runtime splits an odd-sized memset into two launches; what follows the totals
    // inside their carve is scan workspace)
    out->clear_bytes = ((o_bc + sizeof(u64) * (size_t)(NB + 1) + 255) & ~(size_t)255) - o_dc;
    return ESP_OK;
}

double plan_entries(i64 E, int K, u64 span) {
    const double full = std::ldexp(1.0, K);
    return span > 0 && (double)span < full ? (double)E * full / (double)span : (double)E;
}
// bits the run-based pass would resolve for E pending entries in a K-bit key window holding `span` keys (0: not used)
int plan_run_bits(i64 E, int K, u64 span) {
    int planned = 0;
    if (E > esplocal::CAP) {
        const double target = plan_fill() * esplocal::CAP, Ee = plan_entries(E, K, span);
        while (planned < K && Ee / (double)((i64)1 << planned) > target) planned++;
    }
    return planned > 8 ? std::min(planned, 20) : 0;
}
// prefix bits that bring the segments of E pending entries under the bucket kernel's capacity at the planned fill,
// corrected by what the handle's last flush saw (seen_spread)
int plan_prefix_bits(const esp_handle *h, i64 E, int K, double *Ee_out) {
    int planned = 0;
    // (see plan_entries: the window fills only part of its 2^K keys -- and a batch that touches only part of the window's columns,
    // plan_occ_span, is as dense as if the rest of the window held the same)
    const double Ee = plan_entries(E, K, h->plan_occ_span > 0 && h->plan_occ_span < h->win_span ? h->plan_occ_span : h->win_span);
    if (E > seg_cap(h)) {
        double target = plan_fill() * (double)seg_cap(h);
        // (test hook: plan as if the bucket kernel took segments of this many entries -- many prefix bits, i.e. the 9-bit
        // passes, at sizes a CPU oracle can follow)
        if (h->debug_plan_cap > 0.0) target = std::min(target, std::max(8.0, h->debug_plan_cap));  // (esp_debug_plan_cap)
        while (planned < K && Ee / (double)((i64)1 << planned) > target) planned++;
    }
    if (planned > 0 && h->seen_spread > 0.0 && h->seen_spread < 2.0 &&
        Ee / (double)((i64)1 << (planned - 1)) * h->seen_spread <= 0.98 * (double)seg_cap(h))
        planned--;  // (see seen_spread; a wrong guess costs one further pass and corrects itself)
    // ... and irregular data (the longest segment well above the average) gets the bits up front that the last flush
    // had to add in a further pass
    for (int extra = 0; extra < 3 && planned > 0 && planned < K && h->seen_spread >= 1.0 && h->seen_spread < 8.0 &&
                        Ee / (double)((i64)1 << planned) * h->seen_spread > (double)seg_cap(h);
         extra++)
        planned++;
    *Ee_out = Ee;
    return planned;
}
// An item partition of NI records (W updates each) whose expansion resolves the last bits itself (segexpand.hpp): the
// number of those bits (0: not worth it / not possible) and, in *sort_bits, what the passes in front of it resolve.
// The final prefix is what the classic plan gives for segments of CAP / W items; the passes stop up to three bits
// earlier when a segment then still fits the expansion's local sort with room to spare.
int plan_local_bits(esp_handle *h, i64 NI, int W, int K, int *sort_bits) {
    *sort_bits = 0;
    // (measured at config 4's sizes, round 4: the pass it saves is worth 1.3 ms at 3-D, the ordering step and the per-segment
    // rounds cost the expansion 1.6 -- 18.3 against 17.7 ms, elements 24.7 against 23.9, 2-D 5.2 against 4.7: it stays a test
    // hook, esp_debug_force_path(ESP_PATH_LOCAL_BITS), until a mesh shape shows up where the last pass resolves a single bit)
    if (h->force_path != ESP_PATH_LOCAL_BITS) return 0;
    const i64 cap0 = h->plan_cap;
    h->plan_cap = (i64)esplocal::CAP / W;
    double Ee = 0.0;
    const int total = plan_prefix_bits(h, NI, K, &Ee);
    h->plan_cap = cap0;
    for (int b = espseg::MAXB; b >= 1; b--) {
        const int sb = total - b;
        if (sb >= 4 && Ee / (double)((i64)1 << sb) <= 0.8 * (double)espseg::LCAP) {
            *sort_bits = sb;
            return b;
        }
    }
    return 0;
}
int window_bits(const esp_handle *h) {
    int K = 1;
    while (K < 62 && ((u64)1 << K) < h->win_span) K++;
    return K;
}


int32_t aux_ready(esp_handle *h) {
    if (!h->aux) {
        // highest priority: its few tiny launches must get workgroup slots WHILE a kernel that fills the chip runs
        // on the main stream (at equal priority they were seen to start only after that kernel had drained)
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        HIPCK(h, hipStreamCreateWithPriority(&h->aux, hipStreamNonBlocking, greatest));
    }
    if (!h->aux_ev) HIPCK(h, hipEventCreateWithFlags(&h->aux_ev, hipEventDisableTiming));
    return ESP_OK;
}



int32_t run_partition(esp_handle *h, const u64 *kin, const double *vin, u64 *kout, double *vout, int K, int pb,
                             i64 *seg_out, u64 *tile_first_out, bool *tiles_ready, bool *ok, i64 *maxlen_out,
                             const MultiWin *mw, int mw_shift, bool allow_k32, int *key_bytes_out,
                             i64 E_in, const RawSource *raw) {
    const i64 E = E_in >= 0 ? E_in : h->count;  // (E_in: the entries behind a producer's batch, flush_pre_tail; a raw batch)
    const i64 NB = mw ? (i64)mw->P * (i64)mw->nb : (i64)1 << pb;
    CK(ensure(h, h->misc, 256));
    const i64 C = ceil_div<i64>(E, esprun::TILE);
    ChunkArrays ca;
    CK(chunk_arrays(h, C + 64, pb, &ca));
    const i64 RM = C * esprun::RMAX;
    // scratch of this call
    size_t off = 0;
    auto carve = [&](size_t bytes) {
        size_t o = off;
        off += (bytes + 255) & ~(size_t)255;
        return o;
    };
    const size_t o_ro = carve(sizeof(i64) * (size_t)RM);
    const size_t o_hd = carve(sizeof(u64) * (size_t)NB);
    const size_t o_lk = carve(sizeof(u64) * (size_t)RM), o_lk2 = carve(sizeof(u64) * (size_t)RM);
    const size_t o_lv = carve(sizeof(double) * (size_t)RM), o_lv2 = carve(sizeof(double) * (size_t)RM);
    const size_t o_sc = carve(sizeof(u64) * (size_t)(RM + 1 + espscan::workspace_elems(RM + 1)));
    CK(ensure(h, h->runbuf, off));
    char *B = (char *)h->runbuf.p;
    esprun::Args a;
    a.keys_in = kin;
    a.vals_in = vin;
    a.keys_out = kout;
    a.vals_out = vout;
    a.E = E;
    a.shift = mw ? mw_shift : K - pb;
    a.base = h->win_base;
    a.span = h->win_span;
    a.mw_P = mw ? mw->P : 0;
    a.mw_nb = mw ? mw->nb : 0;
    a.mw_base = mw ? mw->d_base : nullptr;
    if (raw) {
        a.raw_rows = raw->rows, a.raw_cols = raw->cols;
        a.raw_m = h->m, a.raw_n = h->n;
        a.raw_rb = h->L.rb, a.raw_kind = raw->kind, a.raw_negate = raw->negate;
        a.raw_err = raw->d_err;
    }
    a.err = (u32 *)h->misc.p + 60;
    a.overflow = ca.overflow;
    a.runs_d = ca.runs_d;
    a.runs_c = ca.runs_c;
    a.runs_off = (i64 *)(B + o_ro);
    a.nruns = ca.nruns;
    a.bucket_count = ca.bucket_count;
    u64 *nruns = a.nruns, *bstart = (u64 *)ca.bucket_count, *head = (u64 *)(B + o_hd);
    u64 *lk = (u64 *)(B + o_lk), *lk2 = (u64 *)(B + o_lk2), *sc = (u64 *)(B + o_sc);
    double *lv = (double *)(B + o_lv), *lv2 = (double *)(B + o_lv2);
    // flags[0] window error, [1] a chunk with too many digits, [2] (run-list sort passes), [3] a digit with too
    // many runs; maxlen in front of them: one 64-byte block the CALLER zeroed
    unsigned long long *d_maxlen = (unsigned long long *)h->misc.p + 24;
    u32 *flags = (u32 *)h->misc.p + 60;
    a.dcount = nullptr;
    a.dlist = nullptr;
    a.nruns_raw = 0;
    a.flags = nullptr;
    a.maxlen = nullptr;
    a.cap = 0;
    if (key_bytes_out) *key_bytes_out = 8;
    *tiles_ready = false;
    // ranked: every digit collects its own runs, ONE kernel turns them into run offsets (run_rank_k), the
    // scatter kernel follows without a host round trip (force_path 12: the radix-ordered run list instead)
    const bool ranked = h->force_path != ESP_PATH_RUN_LIST_BY_RADIX;
    h->last_run_order = 2;
    if (ranked) CK(aux_ready(h));
    {
        a.overflow = flags + 1;
        if (ranked) {
            a.dcount = ca.dcount;
            a.dlist = ca.dlist;
            HIPCK(h, hipMemsetAsync(ca.dcount, 0, ca.clear_bytes, h->stream));
        } else {
            HIPCK(h, hipMemsetAsync(bstart, 0, sizeof(u64) * (size_t)(NB + 1), h->stream));
        }
        Span sp(h, ESP_ST_HIST);
        // Two launches: a PROBE over the first chunks, then the rest -- whose workgroups look at the overflow flag BEFORE they
        // request their keys (run_hist_k: first_chunk > 0).  A stream that is not pre-sorted is recognised by the probe, and what
        // used to cost a full pass over the keys (2.2 ms for 1.2 10^9 shuffled triplets, every time the back-off lets the handle try
        // again) costs the probe.
        const i64 C0 = C > 2 * esprun::PROBE_CHUNKS ? (i64)esprun::PROBE_CHUNKS : C;
        for (i64 c0 = 0; c0 < C; c0 = c0 == 0 ? C0 : C) {
            const unsigned g = (unsigned)(c0 == 0 ? C0 : C - C0);
            if (raw)
                hipLaunchKernelGGL((esprun::run_hist_k<false, true>), dim3(g), dim3(esprun::THREADS), 0, h->stream, a, c0);
            else if (mw)
                hipLaunchKernelGGL((esprun::run_hist_k<true>), dim3(g), dim3(esprun::THREADS), 0, h->stream, a, c0);
            else
                hipLaunchKernelGGL((esprun::run_hist_k<false>), dim3(g), dim3(esprun::THREADS), 0, h->stream, a, c0);
            sp.add(1);
        }
    }
    if (ranked) {
        {
            Span sp(h, ESP_ST_SCAN);
            const unsigned g = (unsigned)grid_for(NB + 1, esprun::THREADS);
            u64 *coarse = ca.coarse;
            hipLaunchKernelGGL(esprun::run_coarse_k, dim3(g), dim3(esprun::THREADS), 0, h->stream, (const unsigned long long *)ca.bucket_count,
                               NB, coarse);
            hipLaunchKernelGGL(esprun::run_rank_k, dim3(g), dim3(esprun::THREADS), 0, h->stream, (const unsigned long long *)ca.bucket_count,
                               (const u64 *)coarse, (const u32 *)ca.dcount, (const u64 *)ca.dlist, NB, seg_out, a.runs_off, d_maxlen,
                               flags + 3);
            sp.add(2);
        }
        // the host reads the flags on the second stream while the scatter kernel (which leaves at once when one
        // of them is set) already runs
        hipLaunchKernelGGL(publish_block_k, dim3(1), dim3(64), 0, h->stream, (const unsigned long long *)d_maxlen, h->pin_scalar);
        HIPCK(h, hipEventRecord(h->aux_ev, h->stream));
        a.nruns_raw = 1;
        a.flags = flags;
        // 4-byte keys for the bucket kernel: one kind for all pending entries, <= 32 key bits below the prefix, no
        // further pass (force_path 14: packed keys always)
        const bool k32 = allow_k32 && key_bytes_out && !mw && h->force_path != ESP_PATH_PACKED_KEYS &&
                         (raw || (h->kind_uniform >= 0 && h->kind_noted == h->count)) && a.shift <= 32;
        a.maxlen = d_maxlen;
        a.cap = esplocal::CAP;
        {
            Span sp(h, ESP_ST_SCATTER);
            if (raw && k32)
                hipLaunchKernelGGL((esprun::run_scatter_k<false, true, true>), dim3((unsigned)C), dim3(esprun::THREADS), 0, h->stream, a);
            else if (raw)
                hipLaunchKernelGGL((esprun::run_scatter_k<false, false, true>), dim3((unsigned)C), dim3(esprun::THREADS), 0, h->stream, a);
            else if (mw)
                hipLaunchKernelGGL((esprun::run_scatter_k<true, false>), dim3((unsigned)C), dim3(esprun::THREADS), 0, h->stream, a);
            else if (k32)
                hipLaunchKernelGGL((esprun::run_scatter_k<false, true>), dim3((unsigned)C), dim3(esprun::THREADS), 0, h->stream, a);
            else
                hipLaunchKernelGGL((esprun::run_scatter_k<false, false>), dim3((unsigned)C), dim3(esprun::THREADS), 0, h->stream, a);
            sp.add(1);
        }
        HIPCK(h, hipEventSynchronize(h->aux_ev));  // (publish_block_k has written the block to pin_scalar)
        const u32 f_err = (u32)h->pin_scalar[6], f_over = (u32)(h->pin_scalar[6] >> 32), f_many = (u32)(h->pin_scalar[7] >> 32);
        if (f_over) {
            *ok = false;
            return ESP_OK;
        }
        if (f_err) FAIL(h, ESP_ERR_STATE, "esp_flush: a pending entry lies outside the declared column window (partition)");
        h->last_run_order = f_many ? 3 : 1;
        if (!f_many) {
            if (key_bytes_out) *key_bytes_out = (k32 && (i64)h->pin_scalar[0] <= (i64)esplocal::CAP) ? 4 : 8;
            *maxlen_out = (i64)h->pin_scalar[0];
            HIPCK(h, hipGetLastError());
            *ok = true;
            return ESP_OK;
        }
        // some digit has more runs than its list holds (nothing was moved): order the run list with the radix passes
        a.nruns_raw = 0;
        a.flags = nullptr;
    }
    HIPCK(h, hipMemsetAsync(nruns + C, 0, sizeof(u64), h->stream));
    {
        Span sp(h, ESP_ST_SCAN);
        sp.add(espscan::exclusive<u64, false>(h->stream, nruns, nruns, C + 1, nruns + C + 1));
        sp.add(espscan::exclusive<u64, false>(h->stream, bstart, bstart, NB + 1, bstart + NB + 1));
    }
    // bucket starts are final here: tiles per bucket and the longest bucket come with the same sync
    {
        Span sp(h, ESP_ST_SCAN);
        HIPCK(h, hipMemcpyAsync(seg_out, bstart, sizeof(i64) * (size_t)(NB + 1), hipMemcpyDeviceToDevice, h->stream));
        HIPCK(h, hipMemsetAsync(d_maxlen, 0, 8, h->stream));
        hipLaunchKernelGGL(espradix::seg_tiles_k, dim3(grid_for(NB + 1, 256)), dim3(256), 0, h->stream, (const i64 *)seg_out, NB,
                           (i64)espradix::TILE, tile_first_out, d_maxlen);
        sp.add(1 + espscan::exclusive<u64, false>(h->stream, tile_first_out, tile_first_out, NB + 1, tile_first_out + NB + 1));
    }
    *tiles_ready = true;
    HIPCK(h, hipMemcpyAsync(h->pin_scalar, a.overflow, 4, hipMemcpyDeviceToHost, h->stream));
    HIPCK(h, hipMemcpyAsync(h->pin_scalar + 1, nruns + C, 8, hipMemcpyDeviceToHost, h->stream));
    HIPCK(h, hipMemcpyAsync(h->pin_scalar + 2, d_maxlen, 8, hipMemcpyDeviceToHost, h->stream));
    HIPCK(h, hipMemcpyAsync(h->pin_scalar + 3, a.err, 4, hipMemcpyDeviceToHost, h->stream));
    HIPCK(h, hipStreamSynchronize(h->stream));
    if ((u32)h->pin_scalar[0]) {
        *ok = false;
        return ESP_OK;
    }
    if ((u32)h->pin_scalar[3]) FAIL(h, ESP_ERR_STATE, "esp_flush: a pending entry lies outside the declared column window (partition)");
    const i64 R = (i64)h->pin_scalar[1];
    *maxlen_out = (i64)h->pin_scalar[2];
    {
        Span sp(h, ESP_ST_SCAN);
        hipLaunchKernelGGL(esprun::run_pack_k, dim3(grid_for(RM, 256)), dim3(256), 0, h->stream, (const u32 *)a.runs_d, (const u32 *)a.runs_c,
                           (const u64 *)nruns, C, lk, lv);
        sp.add(1);
    }
    // stable sort of the run list by digit with the ordinary 8-bit passes (chunk order is kept)
    {
        CK(ensure(h, h->segs, sizeof(i64) * 8));
        i64 *segs = (i64 *)h->segs.p;
        const i64 TR = ceil_div<i64>(R, espradix::TILE);
        hipLaunchKernelGGL(set_i64_k, dim3(1), dim3(1), 0, h->stream, segs, (i64)0, R, (i64)0, TR);
        u64 *ki = lk, *ko = lk2;
        double *vi = lv, *vo = lv2;
        for (int done = 0; done < pb; done += 8) {
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
            p.bits = std::min(8, pb - done);
            CK(partition_pass(h, p, TR));
            std::swap(ki, ko);
            std::swap(vi, vo);
        }
        lk = ki;
        lv = vi;
    }
    {
        Span sp(h, ESP_ST_SCAN);
        hipLaunchKernelGGL(esprun::run_counts_k, dim3(grid_for(R + 1, 256)), dim3(256), 0, h->stream, (const double *)lv, R, sc);
        sp.add(1 + espscan::exclusive<u64, false>(h->stream, sc, sc, R + 1, sc + R + 1));
        hipLaunchKernelGGL(esprun::run_heads_k, dim3(grid_for(R, 256)), dim3(256), 0, h->stream, (const u64 *)lk, (const u64 *)sc, R, head);
        hipLaunchKernelGGL(esprun::run_offsets_k, dim3(grid_for(R, 256)), dim3(256), 0, h->stream, (const u64 *)lk, (const double *)lv,
                           (const u64 *)sc, (const u64 *)head, (const u64 *)bstart, R, a.runs_off);
        sp.add(2);
    }
    {
        Span sp(h, ESP_ST_SCATTER);
        if (raw)
            hipLaunchKernelGGL((esprun::run_scatter_k<false, false, true>), dim3((unsigned)C), dim3(esprun::THREADS), 0, h->stream, a);
        else if (mw)
            hipLaunchKernelGGL((esprun::run_scatter_k<true, false>), dim3((unsigned)C), dim3(esprun::THREADS), 0, h->stream, a);
        else
            hipLaunchKernelGGL((esprun::run_scatter_k<false, false>), dim3((unsigned)C), dim3(esprun::THREADS), 0, h->stream, a);
        sp.add(1);
    }
    HIPCK(h, hipGetLastError());
    *ok = true;
    return ESP_OK;
}

// esp_append_device / esp_commit (kind_all >= 0) on an EMPTY buffer, all entries of one kind (NOT esp_append_host, which packs its
// keys on the host and sends them in stream order: the flush partitions them): the run-based single pass
// (runpart.hpp) reads the caller's triplets directly -- a count pass over the columns, then one kernel that reads rows,
// columns and values and stores key and value at their bucket position (4-byte keys when they fit).  What used to be
// pack (24 B read, 16 B written) + histogram (8 B) + scatter (16 B + 12 B) per entry is 8 B + 24 B read, 12 B written,
// and the flush starts at the bucket kernel (h->pre, as after a device-side producer).  *took = false: the stream is
// no pre-sorted one (or the plan does not apply): nothing was appended, the caller packs in stream order.
int32_t append_partitioned(esp_handle *h, const i64 *d_rows, const i64 *d_cols, const double *d_vals, int kind, int op, i64 count,
                                  bool *took) {
    *took = false;
    if (h->count != 0 || count <= esplocal::CAP || h->shard_user || h->runs_skip > 0) return ESP_OK;
    // (test hooks that pin another path; 27: this one off)
    if (h->force_path == ESP_PATH_GENERAL || h->force_path == ESP_PATH_NO_RUN_PARTITION || h->force_path == ESP_PATH_RUN_LIST_BY_RADIX || h->force_path == ESP_PATH_PRODUCER_STREAM_ORDER || h->force_path == ESP_PATH_NO_BATCH_TAIL || h->force_path == ESP_PATH_NO_APPEND_PARTITION)
        return ESP_OK;
    const int K = window_bits(h);
    double Ee = 0.0;
    int pb = plan_prefix_bits(h, count, K, &Ee);
    int shift = K - pb;
    if (pb <= 8 || pb > 20 || shift < h->L.rb || shift > esplocal::MAX_REM_BITS) return ESP_OK;
    // (FINE partition, as for the device-side producers -- prepart_begin: up to four more prefix bits bring the rest into
    // 4-byte keys, the flush's bucket kernel takes 2^fb buckets as one segment)
    int fb = 0;
    i64 maxlen_c = 0;
    if (shift > 32 && shift - 32 <= 4 && pb + (shift - 32) <= esprun::MAX_PB && h->L.rb <= 32 && h->force_path != ESP_PATH_NO_FINE_PARTITION &&
        h->force_path != ESP_PATH_PACKED_KEYS) {
        fb = shift - 32;
        pb += fb;
        shift = 32;
    }
    CK(reserve_append(h, count));
    const i64 NB = (i64)1 << pb;
    CK(ensure(h, h->seg[1], sizeof(i64) * (size_t)(NB + 1)));
    CK(ensure(h, h->tilef[1], sizeof(u64) * (size_t)(NB + 1 + espscan::workspace_elems(NB + 1))));
    CK(ensure(h, h->misc, 256));
    unsigned long long *d_err = (unsigned long long *)h->misc.p;
    const RawSource raw{d_rows, d_cols, kind, (op == ESP_OP_SUB && kind != ESP_SET) ? 1 : 0, d_err};
    bool tr = false, ok = false;
    i64 ml = count;
    int kb = 8;
    // the run lists of the previous assembly, when this batch looks like a repetition of it (the scatter kernel checks
    // every tile; force_path 31: never)
    const esp_handle::RawPlan rp = h->rawplan;
    bool reused = false;
    if (rp.valid && rp.count == count && rp.kind == kind && rp.K == K && rp.pb == pb && rp.base == h->win_base && rp.span == h->win_span &&
        h->force_path != ESP_PATH_NO_PLAN_REUSE) {
        const i64 C = ceil_div<i64>(count, esprun::TILE);
        ChunkArrays ca;
        CK(chunk_arrays(h, C + 64, pb, &ca, /*keep_plan=*/true));
        if (C == rp.chunks && h->runbuf.p) {
            // error word | verify flag, then the longest bucket and the four flag words as the ranking kernel left them
            h->pin_scalar[0] = ~0ull, h->pin_scalar[1] = 0ull;
            HIPCK(h, hipMemcpyAsync(d_err, h->pin_scalar, 16, hipMemcpyHostToDevice, h->stream));
            unsigned long long *d_maxlen = (unsigned long long *)h->misc.p + 24;
            hipLaunchKernelGGL(set_i64_k, dim3(1), dim3(1), 0, h->stream, (i64 *)d_maxlen, rp.maxlen, (i64)0, (i64)0, (i64)0);
            hipLaunchKernelGGL(set_i64_k, dim3(1), dim3(1), 0, h->stream, (i64 *)d_maxlen + 4, (i64)0, (i64)0, (i64)0, (i64)0);
            esprun::Args a;
            memset(&a, 0, sizeof a);
            a.vals_in = d_vals;
            a.keys_out = (u64 *)h->keys.p;
            a.vals_out = (double *)h->vals.p;
            a.E = count;
            a.shift = K - pb;
            a.base = h->win_base, a.span = h->win_span;
            a.raw_rows = d_rows, a.raw_cols = d_cols;
            a.raw_m = h->m, a.raw_n = h->n;
            a.raw_rb = h->L.rb, a.raw_kind = kind, a.raw_negate = raw.negate;
            a.raw_err = d_err;
            a.err = (u32 *)h->misc.p + 60;
            a.runs_d = ca.runs_d, a.runs_c = ca.runs_c, a.nruns = ca.nruns;
            a.runs_off = (i64 *)h->runbuf.p;
            a.nruns_raw = 1;
            a.maxlen = d_maxlen;
            a.cap = esplocal::CAP;
            a.verify_err = (u32 *)(d_err + 1);
            {
                Span sp(h, ESP_ST_SCATTER);
                if (rp.key_bytes == 4)
                    hipLaunchKernelGGL((esprun::run_scatter_k<false, true, true, true>), dim3((unsigned)C), dim3(esprun::THREADS), 0, h->stream, a);
                else
                    hipLaunchKernelGGL((esprun::run_scatter_k<false, false, true, true>), dim3((unsigned)C), dim3(esprun::THREADS), 0, h->stream, a);
                sp.add(1);
            }
            HIPCK(h, hipMemcpyAsync(h->pin_scalar, d_err, 16, hipMemcpyDeviceToHost, h->stream));
            HIPCK(h, hipStreamSynchronize(h->stream));
            HIPCK(h, hipGetLastError());
            if (h->pin_scalar[0] != ~0ull) {
                h->rawplan.valid = false;
                FAIL(h, ESP_ERR_BOUNDS, "BoundsError: entry %llu of the batch has an index outside %lld x %lld (or a bad kind)",
                     (unsigned long long)h->pin_scalar[0], (long long)h->m, (long long)h->n);
            }
            if ((u32)h->pin_scalar[1] == 0u) {
                reused = true;
                ok = true, ml = rp.maxlen, kb = rp.key_bytes;
                Ee = rp.Ee;
                fb = rp.fb, maxlen_c = rp.maxlen_c;
            } else {
                h->rawplan.valid = false;  // (another stream: the full path below, which makes a plan of its own)
            }
        }
    }
    if (!reused) {
    h->pin_scalar[0] = ~0ull;
    HIPCK(h, hipMemcpyAsync(d_err, h->pin_scalar, 8, hipMemcpyHostToDevice, h->stream));
    HIPCK(h, hipMemsetAsync((unsigned long long *)h->misc.p + 24, 0, 64, h->stream));  // maxlen .. the four flag words (run_partition)
    CK(run_partition(h, nullptr, d_vals, (u64 *)h->keys.p, (double *)h->vals.p, K, pb, (i64 *)h->seg[1].p, (u64 *)h->tilef[1].p, &tr, &ok, &ml,
                     nullptr, 0, /*allow_k32=*/true, &kb, count, &raw));
    HIPCK(h, hipMemcpyAsync(h->pin_scalar, d_err, 8, hipMemcpyDeviceToHost, h->stream));
    HIPCK(h, hipStreamSynchronize(h->stream));
    if (h->pin_scalar[0] != ~0ull)
        FAIL(h, ESP_ERR_BOUNDS, "BoundsError: entry %llu of the batch has an index outside %lld x %lld (or a bad kind)",
             (unsigned long long)h->pin_scalar[0], (long long)h->m, (long long)h->n);
    if (!ok) {  // not a pre-sorted stream: neither this handle's appends nor its next flushes try again soon
        h->runs_penalty = std::min(16, 2 * h->runs_penalty + 1);
        h->runs_skip = h->runs_penalty + 1;
        return ESP_OK;
    }
    if (fb > 0 && kb == 4) {  // the longest SEGMENT the flush will meet (2^fb buckets); one that outgrows the bucket kernel: bucket by bucket
        unsigned long long *d_cm = (unsigned long long *)h->misc.p + 25;
        HIPCK(h, hipMemsetAsync(d_cm, 0, 8, h->stream));
        hipLaunchKernelGGL(coarse_seg_max_k, dim3(grid_for(NB >> fb, 256)), dim3(256), 0, h->stream, (const i64 *)h->seg[1].p, NB >> fb, fb, d_cm);
        HIPCK(h, hipMemcpyAsync(h->pin_scalar, d_cm, 8, hipMemcpyDeviceToHost, h->stream));
        HIPCK(h, hipStreamSynchronize(h->stream));
        maxlen_c = (i64)h->pin_scalar[0];
        if (maxlen_c > (i64)esplocal::CAP) fb = 0;
    } else {
        fb = 0;
    }
    // (the tables of this batch serve the next one that looks the same -- only the ranked flavour leaves them complete)
    esp_handle::RawPlan &np = h->rawplan;
    np.valid = h->last_run_order == 1;
    np.count = count, np.chunks = ceil_div<i64>(count, esprun::TILE), np.maxlen = ml;
    np.kind = kind, np.K = K, np.pb = pb, np.key_bytes = kb;
    np.fb = fb, np.maxlen_c = maxlen_c;
    np.base = h->win_base, np.span = h->win_span, np.Ee = Ee;
    }
    h->last_plan_reused = reused ? 1 : 0;
    h->runs_penalty = 0;
    esp_handle::PrePart &pp = h->pre;
    pp.K = K;
    pp.pb = pb;
    pp.maxlen = ml;
    pp.key_bytes = kb;
    pp.kind = kind;
    pp.E = count;
    pp.tail = 0;
    pp.base = h->win_base;
    pp.span = h->win_span;
    pp.Ee = Ee;
    pp.mw_P = 0, pp.mw_me = 0, pp.mw_shift = 0, pp.mw_nb = 0, pp.mw_eps = 0;
    pp.own32 = false;
    pp.fb = (kb == 4) ? fb : 0, pp.maxlen_c = maxlen_c;
    note_kind(h, kind, count);
    h->count += count;
    pending_changed(h);
    h->pre.valid = true;
    *took = true;
    return ESP_OK;
}

// The same for an append BEHIND a bucket-ordered batch when a pattern is stored (a re-assembly whose mesh gained
// couplings: esp_flush flushes the batch by itself and then what came behind it): the caller's triplets go through the
// run-based single pass with a plan of their own and land behind the batch as PACKED keys in bucket order -- to
// everybody else they are the packed tail they would have been (a stable partition keeps every column's order);
// esp_flush's split finds the segment starts in tseg and starts the tail's flush at the bucket kernel.  Instead of
// pack (24 B read, 16 B written) + histogram + scatter (16 B + 16 B): 8 B + 24 B read, 16 B written per entry.
int32_t append_tail_partitioned(esp_handle *h, const i64 *d_rows, const i64 *d_cols, const double *d_vals, int kind, int op, i64 count,
                                bool *took) {
    *took = false;
    const esp_handle::PrePart &pp0 = h->pre;
    if (!pp0.valid || pp0.tail != 0 || pp0.mw_P != 0 || h->count != pp0.E || h->nnz == 0 || count <= esplocal::CAP || h->shard_user ||
        h->runs_skip > 0 || windowed(h))
        return ESP_OK;
    if (h->force_path == ESP_PATH_GENERAL || h->force_path == ESP_PATH_NO_RUN_PARTITION || h->force_path == ESP_PATH_RUN_LIST_BY_RADIX || h->force_path == ESP_PATH_PRODUCER_STREAM_ORDER || h->force_path == ESP_PATH_NO_BATCH_TAIL || h->force_path == ESP_PATH_NO_APPEND_PARTITION || h->force_path == ESP_PATH_BATCH_TAIL_ONE_FLUSH || h->force_path == ESP_PATH_TAIL_TO_FRONT)
        return ESP_OK;
    const int K = window_bits(h);
    double Ee = 0.0;
    // (short stored columns: the tail's flush rebuilds the matrix with the stored entries as the first piece of every segment
    // -- flush_rebuild --, so a segment must hold its columns' stored entries too, and suit the small variant of the bucket kernel)
    const bool rebuild = h->force_path == ESP_PATH_AUTO && (double)h->nnz <= 10.0 * (double)h->n && (double)(h->nnz + count) <= 14.0 * (double)h->n;
    const i64 cap0 = h->plan_cap;
    if (rebuild) h->plan_cap = 6 * esplocal::THREADS;
    int pb = plan_prefix_bits(h, rebuild ? count + h->nnz : count, K, &Ee);
    h->plan_cap = cap0;
    if (rebuild && (pb > 20 || K - pb < h->L.rb)) pb = plan_prefix_bits(h, count, K, &Ee);  // (no such plan: the tail by itself)
    const int shift = K - pb;
    if (pb <= 8 || pb > 20 || shift < h->L.rb || shift > esplocal::MAX_REM_BITS) return ESP_OK;
    CK(reserve_append(h, count));
    if (!h->pre_keep) return ESP_OK;  // (the batch went back to packed keys: an ordinary append)
    const i64 E0 = h->count;
    const i64 NB = (i64)1 << pb;
    CK(ensure(h, h->tseg, sizeof(i64) * (size_t)(NB + 1)));
    CK(ensure(h, h->ttile, sizeof(u64) * (size_t)(NB + 1 + espscan::workspace_elems(NB + 1))));
    CK(ensure(h, h->misc, 256));
    unsigned long long *d_err = (unsigned long long *)h->misc.p;
    h->pin_scalar[0] = ~0ull;
    HIPCK(h, hipMemcpyAsync(d_err, h->pin_scalar, 8, hipMemcpyHostToDevice, h->stream));
    HIPCK(h, hipMemsetAsync((unsigned long long *)h->misc.p + 24, 0, 64, h->stream));  // maxlen .. the four flag words (run_partition)
    const RawSource raw{d_rows, d_cols, kind, (op == ESP_OP_SUB && kind != ESP_SET) ? 1 : 0, d_err};
    bool tr = false, ok = false;
    i64 ml = count;
    int kb = 8;
    CK(run_partition(h, nullptr, d_vals, (u64 *)h->keys.p + E0, (double *)h->vals.p + E0, K, pb, (i64 *)h->tseg.p, (u64 *)h->ttile.p, &tr, &ok, &ml,
                     nullptr, 0, /*allow_k32=*/false, &kb, count, &raw));
    HIPCK(h, hipMemcpyAsync(h->pin_scalar, d_err, 8, hipMemcpyDeviceToHost, h->stream));
    HIPCK(h, hipStreamSynchronize(h->stream));
    if (h->pin_scalar[0] != ~0ull) {
        h->pre_keep = false;  // (nothing was appended behind the batch after all)
        FAIL(h, ESP_ERR_BOUNDS, "BoundsError: entry %llu of the batch has an index outside %lld x %lld (or a bad kind)",
             (unsigned long long)h->pin_scalar[0], (long long)h->m, (long long)h->n);
    }
    if (!ok || kb != 8) {  // no pre-sorted stream: nothing was appended, the caller packs in stream order
        h->runs_penalty = std::min(16, 2 * h->runs_penalty + 1);
        h->runs_skip = h->runs_penalty + 1;
        h->pre_keep = false;
        return ESP_OK;
    }
    note_kind(h, kind, count);
    h->count += count;
    pending_changed(h);
    esp_handle::TailPart &tp = h->tailpart;
    tp.K = K, tp.pb = pb, tp.T = count, tp.maxlen = ml;
    tp.base = h->win_base, tp.span = h->win_span;
    tp.valid = h->pre.valid && h->pre.tail == count;
    *took = true;
    return ESP_OK;
}

// the planned passes of sort_msd over E entries: prefix bits and number of passes
static void plan_passes(const esp_handle *h, i64 E, int K, int *planned_out, int *npass_out, double *Ee_out) {
    int planned = plan_prefix_bits(h, E, K, Ee_out);
    const bool fixed_bits = h->plan_bits > 0;
    if (fixed_bits) planned = std::min(h->plan_bits, K);  // (an item partition that leaves the last bits to its expansion)
    // one bit short of a whole number of 8-bit passes: an average fill of up to 95 % is worth
    // trying with one pass less (the longest segment is checked after the planned passes and a
    // further pass is added only if a segment really overflows)
    if (!fixed_bits && planned > 8 && planned % 8 == 1 && *Ee_out / (double)((i64)1 << (planned - 1)) <= 0.95 * (double)seg_cap(h)) planned--;
    // (digits of 9 bits only where they save a whole pass -- 17 or 18 bits in two passes: a tile then holds 8 entries per
    // digit instead of 16; force_path 23: never)
    const int npass8 = (planned + 7) / 8, npass9 = (planned + espradix::MAX_BITS - 1) / espradix::MAX_BITS;
    *npass_out = (npass9 < npass8 && h->force_path != ESP_PATH_EIGHT_BIT_PASSES) ? npass9 : npass8;
    *planned_out = planned;
}

// esp_append_device / esp_commit of ONE kind on an EMPTY buffer whose stream is no pre-sorted one (append_partitioned gave up: a
// shuffled assembly handed over as triplets): the FIRST radix pass of the flush runs now, straight from the caller's arrays --
// a histogram over the columns, then a scatter that forms key and value and stores them in the order of that pass.  What used to
// be pack (24 B read, 16 B written) + histogram (8 B) + scatter (16 B + 16 B) per entry is 8 B + 24 B read, 16 B written; the
// buffer holds packed keys -- a stable permutation of the stream: an ordinary pending buffer to whoever does not know -- and
// sort_msd resumes behind the pass (esp_handle::PrePass).  *took = false: not applicable, the caller packs in stream order.
int32_t append_first_pass(esp_handle *h, const i64 *d_rows, const i64 *d_cols, const double *d_vals, int kind, int op, i64 count, bool *took) {
    *took = false;
    if (h->count != 0 || count < ((i64)1 << 18) || count >= 0xFFFFFFF0ll || h->shard_user || windowed(h) || h->force_path != ESP_PATH_AUTO) return ESP_OK;
    const int K = window_bits(h);
    int planned = 0, npass = 0;
    double Ee = 0.0;
    plan_passes(h, count, K, &planned, &npass, &Ee);
    if (planned < 8 || npass < 1) return ESP_OK;
    const int bits0 = (planned + npass - 1) / npass;
    if (K - bits0 < h->L.rb) return ESP_OK;  // (the digit must follow from the column alone)
    CK(reserve_append(h, count));
    h->rawplan.valid = false;
    h->genplan.valid = false;
    CK(ensure(h, h->misc, 256));
    unsigned long long *d_err = (unsigned long long *)h->misc.p;
    unsigned long long *d_maxlen = (unsigned long long *)h->misc.p + 24;
    u32 *d_werr = (u32 *)h->misc.p + 60;
    h->pin_scalar[0] = ~0ull;
    HIPCK(h, hipMemcpyAsync(d_err, h->pin_scalar, 8, hipMemcpyHostToDevice, h->stream));
    HIPCK(h, hipMemsetAsync(d_maxlen, 0, 64, h->stream));
    const i64 T = ceil_div<i64>(count, espradix::TILE);
    CK(ensure(h, h->seg[0], sizeof(i64) * 4));
    CK(ensure(h, h->tilef[0], sizeof(u64) * (size_t)(4 + espscan::workspace_elems(4))));
    {
        Span sp(h, ESP_ST_SCAN);
        hipLaunchKernelGGL(set_i64_k, dim3(1), dim3(1), 0, h->stream, (i64 *)h->seg[0].p, (i64)0, count, (i64)0, (i64)0);
        hipLaunchKernelGGL(set_i64_k, dim3(1), dim3(1), 0, h->stream, (i64 *)h->tilef[0].p, (i64)0, T, (i64)0, (i64)0);
        sp.add(2);
    }
    espradix::Pass p;
    p.keys_in = nullptr;
    p.vals_in = d_vals;
    p.keys_out = (u64 *)h->keys.p;
    p.vals_out = (double *)h->vals.p;
    p.seg_start = (const i64 *)h->seg[0].p;
    p.tile_first = (const i64 *)h->tilef[0].p;
    p.S = 1;
    p.owner_P = 0;
    p.owner_n = 1;
    p.colshift = 0;
    p.base = h->win_base;
    p.span = h->win_span;
    p.err = d_werr;
    p.bits = bits0;
    p.shift = K - bits0;
    p.keys_only = 0;
    p.raw_rows = d_rows, p.raw_cols = d_cols;
    p.raw_m = h->m, p.raw_n = h->n;
    p.raw_rb = h->L.rb, p.raw_kind = kind, p.raw_negate = (op == ESP_OP_SUB && kind != ESP_SET) ? 1 : 0;
    p.raw_err = d_err;
    CK(partition_pass(h, p, T));
    const int S2 = 1 << bits0;
    CK(ensure(h, h->seg[1], sizeof(i64) * (size_t)(S2 + 1)));
    CK(ensure(h, h->tilef[1], sizeof(u64) * (size_t)(S2 + 1 + espscan::workspace_elems(S2 + 1))));
    {  // (at most 512 segments: one launch for the table, its tile counts and the longest segment)
        Span sp(h, ESP_ST_SCAN);
        static_assert((1 << espradix::MAX_BITS) < espradix::SMALL_TABLE, "the first pass's table is a small one");
        hipLaunchKernelGGL(espradix::small_table_k, dim3(1), dim3(256), 0, h->stream, (const u64 *)h->hist.p, (const i64 *)h->seg[0].p,
                           (const i64 *)h->tilef[0].p, 1, bits0, (i64 *)h->seg[1].p, count, (i64)espradix::TILE, (u64 *)h->tilef[1].p, d_maxlen);
        sp.add(1);
    }
    HIPCK(h, hipMemcpyAsync(h->pin_scalar, d_err, 8, hipMemcpyDeviceToHost, h->stream));
    HIPCK(h, hipMemcpyAsync(h->pin_scalar + 1, d_maxlen, 8, hipMemcpyDeviceToHost, h->stream));
    HIPCK(h, hipStreamSynchronize(h->stream));
    HIPCK(h, hipGetLastError());
    if (h->pin_scalar[0] != ~0ull)
        FAIL(h, ESP_ERR_BOUNDS, "BoundsError: entry %llu of the batch has an index outside %lld x %lld (or a bad kind)",
             (unsigned long long)h->pin_scalar[0], (long long)h->m, (long long)h->n);
    note_kind(h, kind, count);
    h->count += count;
    pending_changed(h);
    esp_handle::PrePass &pp = h->prepass;
    pp.count = count, pp.maxlen = (i64)h->pin_scalar[1];
    pp.K = K, pp.planned = planned, pp.npass = npass, pp.bits0 = bits0, pp.cur = 1, pp.S = S2;
    pp.base = h->win_base, pp.span = h->win_span, pp.Ee = Ee;
    pp.valid = true;
    *took = true;
    return ESP_OK;
}

int32_t sort_msd(esp_handle *h, Sorted *out) {
    h->rawplan.valid = false;  // (the segment tables are rewritten)
    h->genplan.valid = false;
    const esp_handle::PrePass pre0 = h->prepass;
    h->prepass.valid = false;
    const i64 E = h->count;
    // sort bits of the key window: (key>>2) - win_base lies in [0, win_span)
    int K = 1;
    while (K < 62 && ((u64)1 << K) < h->win_span) K++;
    CK(ensure(h, h->keys2, sizeof(u64) * (size_t)E));
    CK(ensure(h, h->vals2, sizeof(double) * (size_t)E));
    CK(ensure(h, h->misc, 256));
    unsigned long long *d_maxlen = (unsigned long long *)h->misc.p + 24;
    u32 *d_werr = (u32 *)h->misc.p + 60;
    HIPCK(h, hipMemsetAsync(d_maxlen, 0, 64, h->stream));  // maxlen .. the four flag words (run_partition)

    double Ee = 0.0;
    const bool fixed_bits = h->plan_bits > 0;
    int planned_run = plan_prefix_bits(h, E, K, &Ee);  // the run-based pass takes up to 20 bits at once: no need to be tight
    if (fixed_bits) planned_run = std::min(h->plan_bits, K);
    int planned = 0, npass = 0;
    plan_passes(h, E, K, &planned, &npass, &Ee);
    // the first pass ran while the entries were appended (append_first_pass): its tables stand, the loop resumes behind it
    const bool resume = pre0.valid && pre0.count == E && h->pend_off == 0 && pre0.K == K && pre0.base == h->win_base && pre0.span == h->win_span &&
                        !h->item_mode && !fixed_bits && h->force_path == ESP_PATH_AUTO && !h->shard_user;
    if (resume) planned = pre0.planned, npass = pre0.npass, Ee = pre0.Ee;

    int cur = 0, S = 1, done = 0;
    CK(ensure(h, h->seg[0], sizeof(i64) * 4));
    CK(ensure(h, h->tilef[0], sizeof(u64) * (size_t)(4 + espscan::workspace_elems(4))));
    const i64 T = ceil_div<i64>(E, espradix::TILE);
    bool tiles_ready = false;  // seg[cur] has its tile table in tilef[cur] (only the 8-bit passes need one)
    // (pend_off: the pending entries start behind a batch that was flushed by itself -- esp_flush)
    u64 *kin = (u64 *)h->keys.p + h->pend_off, *kout = (u64 *)h->keys2.p;
    double *vin = (double *)h->vals.p + h->pend_off, *vout = (double *)h->vals2.p;
    i64 maxlen = E;
    bool ok = true;
    int pass_idx = 0;
    int npass_eff = npass;
    h->last_partition = 2;
    // pre-sorted streams: the first (up to) 16 bits in ONE pass (runpart.hpp)
    int passes_here = 0;
    bool k32_written = false;  // the buffer the last pass wrote holds 4-byte keys
    if (resume) {
        cur = pre0.cur, S = pre0.S, done = pre0.bits0, pass_idx = 1;
        tiles_ready = true;
        maxlen = pre0.maxlen;
        if (h->runs_skip > 0) h->runs_skip--;  // (the back-off of the run-based attempt counts flushes)
    }
    // (item records that are single words -- an element batch in a mesh's natural cell order is as pre-sorted as any assembly loop's
    // stream -- take it too, keys only; records with a value word stay with the radix passes)
    const bool items_1w = h->item_mode && h->item_keys_only && h->plan_try_runs;
    if (planned_run > 8 && h->force_path != ESP_PATH_NO_RUN_PARTITION && (!h->item_mode || items_1w) && !resume) {
        if (h->runs_skip > 0) {
            h->runs_skip--;
        } else {
            const int pb = std::min(planned_run, 20);
            const int S2 = 1 << pb;
            CK(ensure(h, h->seg[1], sizeof(i64) * (size_t)(S2 + 1)));
            CK(ensure(h, h->tilef[1], sizeof(u64) * (size_t)(S2 + 1 + espscan::workspace_elems(S2 + 1))));
            bool took = false;
            i64 ml = E;
            bool tr = false;
            int kb = 8;
            CK(run_partition(h, kin, items_1w ? nullptr : vin, kout, items_1w ? nullptr : vout, K, pb, (i64 *)h->seg[1].p, (u64 *)h->tilef[1].p, &tr,
                             &took, &ml, nullptr, 0, /*allow_k32=*/pb >= planned_run && !h->item_mode, &kb));
            if (took) {
                tiles_ready = tr;
                out->key_bytes = kb;
                out->kind = h->kind_uniform;
            }
            if (took) {
                std::swap(kin, kout);
                std::swap(vin, vout);
                cur = 1;
                S = S2;
                done = pb;
                planned = std::max(planned_run, pb);
                npass_eff = (planned - pb + 7) / 8;
                h->last_partition = 1;
                h->runs_penalty = 0;
                maxlen = ml;
            } else {
                h->runs_penalty = std::min(16, 2 * h->runs_penalty + 1);
                h->runs_skip = h->runs_penalty;
            }
        }
    }
    // (the passes ended in 4-byte keys and the bucket kernel cannot take the result: packed keys again, into the other pair, whose
    // entries the last pass has consumed)
    auto unpack32 = [&]() -> int32_t {
        Span sp(h, ESP_ST_COPY);
        hipLaunchKernelGGL(esprun::expand_keys_k, dim3(esprun::expand_keys_grid(S)), dim3(esprun::THREADS), 0, h->stream, (const u32 *)kin,
                           (const i64 *)h->seg[cur].p, K - done, h->win_base, (u32)h->kind_uniform, kout, (i64)S);
        HIPCK(h, hipMemcpyAsync(vout, vin, sizeof(double) * (size_t)E, hipMemcpyDeviceToDevice, h->stream));
        sp.add(2);
        std::swap(kin, kout);
        std::swap(vin, vout);
        out->key_bytes = 8;
        k32_written = false;
        return ESP_OK;
    };
    for (;;) {
        int bits;
        if (pass_idx < npass_eff) {
            // spread the planned bits evenly over the planned passes
            bits = (planned - done + (npass_eff - pass_idx) - 1) / (npass_eff - pass_idx);
        } else {
            if (maxlen <= seg_cap(h)) break;
            if (done >= K || done >= 24) {
                ok = false;
                break;
            }
            // just enough further bits to bring the longest segment under the capacity
            bits = 1;
            while (bits < 8 && (double)maxlen / (double)(1 << bits) > 0.8 * (double)seg_cap(h)) bits++;
            bits = std::min(bits, K - done);
        }
        if (!tiles_ready) {
            Span sp(h, ESP_ST_SCAN);
            if (S == 1) {  // the whole buffer as one segment
                hipLaunchKernelGGL(set_i64_k, dim3(1), dim3(1), 0, h->stream, (i64 *)h->seg[0].p, (i64)0, E, (i64)0, (i64)0);
                hipLaunchKernelGGL(set_i64_k, dim3(1), dim3(1), 0, h->stream, (i64 *)h->tilef[0].p, (i64)0, T, (i64)0, (i64)0);
                sp.add(2);
            } else {  // (segments of the run-based pass, whose ranked flavour leaves the tile table to its user)
                u64 *tf = (u64 *)h->tilef[cur].p;
                HIPCK(h, hipMemsetAsync(d_maxlen, 0, 8, h->stream));
                hipLaunchKernelGGL(espradix::seg_tiles_k, dim3(grid_for((i64)S + 1, 256)), dim3(256), 0, h->stream, (const i64 *)h->seg[cur].p,
                                   (i64)S, (i64)espradix::TILE, tf, d_maxlen);
                sp.add(1 + espscan::exclusive<u64, false>(h->stream, tf, tf, (i64)S + 1, tf + S + 1));
            }
            tiles_ready = true;
        }
        espradix::Pass p;
        p.keys_in = kin;
        p.vals_in = vin;
        p.keys_out = kout;
        p.vals_out = vout;
        p.seg_start = (const i64 *)h->seg[cur].p;
        p.tile_first = (const i64 *)h->tilef[cur].p;
        p.S = S;
        p.owner_P = 0;
        p.owner_n = 1;
        p.colshift = 0;
        p.base = h->win_base;
        p.span = h->win_span;
        p.err = d_werr;
        p.bits = bits;
        p.shift = K - done - bits;
        p.keys_only = h->item_mode && h->item_keys_only ? 1 : 0;
        // a stream of ONE known kind: from the first pass that leaves at most 32 key bits below its prefix on, 4-byte keys (those
        // bits) out of a pass, into the next and into the bucket kernel (12 bytes per entry instead of 16; the bucket kernel takes
        // its 4-byte-key forms).  Any force_path: packed keys throughout
        const int rem_out = K - done - bits;
        const bool k32_now = k32_written || (!h->item_mode && !fixed_bits && h->force_path == ESP_PATH_AUTO && h->pend_off == 0 &&
                                             h->kind_uniform >= 0 && h->kind_noted == h->count && !windowed(h) && !h->shard_user &&
                                             rem_out <= 32 && rem_out <= esplocal::MAX_REM_BITS && out->key_bytes != 4);
        p.k32_in = k32_written ? 1 : 0;
        p.k32_out = k32_now ? 1 : 0;
        p.k32_rem = rem_out;
        const i64 max_tiles = S == 1 ? T : T + S;
        CK(partition_pass(h, p, max_tiles));
        if (k32_now) {
            out->key_bytes = 4;
            out->kind = h->kind_uniform;
            out->k32_passes += 1;
            k32_written = true;
        }
        const int S2 = S << bits;
        CK(ensure(h, h->seg[1 - cur], sizeof(i64) * (size_t)(S2 + 1)));
        CK(ensure(h, h->tilef[1 - cur], sizeof(u64) * (size_t)(S2 + 1 + espscan::workspace_elems(S2 + 1))));
        if (S2 < espradix::SMALL_TABLE) {  // (a small table: segments, tile counts, their scan and the longest segment in one launch)
            Span sp(h, ESP_ST_SCAN);
            hipLaunchKernelGGL(espradix::small_table_k, dim3(1), dim3(256), 0, h->stream, (const u64 *)h->hist.p, (const i64 *)h->seg[cur].p,
                               (const i64 *)h->tilef[cur].p, S, bits, (i64 *)h->seg[1 - cur].p, E, (i64)espradix::TILE, (u64 *)h->tilef[1 - cur].p,
                               d_maxlen);
            sp.add(1);
        } else {
            Span sp(h, ESP_ST_SCAN);
            hipLaunchKernelGGL(espradix::new_segments_k, dim3(grid_for(S2 + 1, 256)), dim3(256), 0, h->stream, (const u64 *)h->hist.p,
                               (const i64 *)h->seg[cur].p, (const i64 *)h->tilef[cur].p, S, bits, (i64 *)h->seg[1 - cur].p, E);
            HIPCK(h, hipMemsetAsync(d_maxlen, 0, 8, h->stream));
            u64 *tf = (u64 *)h->tilef[1 - cur].p;
            hipLaunchKernelGGL(espradix::seg_tiles_k, dim3(grid_for(S2 + 1, 256)), dim3(256), 0, h->stream, (const i64 *)h->seg[1 - cur].p,
                               (i64)S2, (i64)espradix::TILE, tf, d_maxlen);
            sp.add(2 + espscan::exclusive<u64, false>(h->stream, tf, tf, S2 + 1, tf + S2 + 1));
        }
        std::swap(kin, kout);
        std::swap(vin, vout);
        cur = 1 - cur;
        S = S2;
        done += bits;
        pass_idx++;
        passes_here++;
        if (pass_idx >= npass_eff) {
            // (the longest segment and the window flag lie in one 64-byte block: one copy, one round trip)
            HIPCK(h, hipMemcpyAsync(h->pin_scalar, d_maxlen, 64, hipMemcpyDeviceToHost, h->stream));
            HIPCK(h, hipStreamSynchronize(h->stream));
            maxlen = (i64)h->pin_scalar[0];
        }
    }
    if (k32_written && !(ok && maxlen <= seg_cap(h))) CK(unpack32());  // (the general path sorts packed keys)
    if (S == 1 && !tiles_ready)  // no pass at all: the buffer is the one segment
        hipLaunchKernelGGL(set_i64_k, dim3(1), dim3(1), 0, h->stream, (i64 *)h->seg[0].p, (i64)0, E, (i64)0, (i64)0);
    HIPCK(h, hipGetLastError());
    if (passes_here > 0) {  // the partition passes clamp and report keys outside the window (d_werr: word 12 of the block read above)
        if ((u32)h->pin_scalar[6]) FAIL(h, ESP_ERR_STATE, "esp_flush: a pending entry lies outside the declared column window (partition)");
    }
    if (out->key_bytes == 4 && !k32_written && (passes_here > 0 || cur != 1))
        FAIL(h, ESP_ERR_STATE, "esp_flush: internal error (4-byte keys met a further partition pass)");
    out->sk = kin;
    out->sv = vin;
    out->in_primary = (kin == (u64 *)h->keys.p + h->pend_off);
    out->S = S;
    out->total = E;
    out->seg_start = (const i64 *)h->seg[cur].p;
    out->rem_bits = K - done;
    out->local_ok = ok && (K - done) <= esplocal::MAX_REM_BITS && maxlen <= seg_cap(h);
    out->fits = ok && maxlen <= seg_cap(h);
    out->maxlen = maxlen;
    h->seen_spread = (done > 0 && Ee > 0.0) ? (double)maxlen * std::ldexp(1.0, done) / Ee : 0.0;
    return ESP_OK;
}

