// flush.hip -- libesparse_hip: esp_flush and what it drives (see internal.hpp for the map of the translation units)
#include "internal.hpp"

bool esplocal::launch(const Variant &v, unsigned grid, hipStream_t stream, const Args &a) {
    if (v.g3) return launch_group3(v, grid, stream, a);
    if (v.grp) return v.shortg ? launch_group_short(v, grid, stream, a) : launch_group(v, grid, stream, a);
    if (v.pieces) return v.small_variant ? launch_pieces_small(v, grid, stream, a) : v.fresh ? launch_pieces_fresh(v, grid, stream, a) : launch_pieces_stored(v, grid, stream, a);
    return v.small_variant ? launch_small(v, grid, stream, a) : launch_regular(v, grid, stream, a);
}


// colend (u64, n+1, zero-initialised, filled with column ends) -> colptr; merges with the old
// CSC when there is one.  New entries are in h->newkey/h->newval (Z0>0) or already in
// h->rowval/h->nzval (Z0==0).
int32_t finish_csc(esp_handle *h, i64 Z0, i64 Zn, const u64 *new_key, const double *new_val) {
    const i64 N1 = h->n + 1;
    u64 *colend = (u64 *)h->colend.p;
    i64 c0, ccnt;  // the columns this flush can have touched (a shard's window, else all)
    col_range(h, &c0, &ccnt);
    if (windowed(h)) h->tail_stale = h->wc1 < h->n;
    if (Z0 == 0) {
        // colptr = 1 + exclusive max-scan of the column ends, written by the scan's last pass
        Span sp(h, ESP_ST_COLPTR);
        sp.add(espscan::exclusive<u64, true>(h->stream, colend + c0, (u64 *)h->colptr.p + c0, ccnt, colend + N1, (u64)1));
        if (!windowed(h)) h->ones_pending = false;  // (every entry of colptr was written)
        h->nnz = Zn;
        h->pattern_version++, h->values_version++;
        return ESP_OK;
    }
    const i64 Zt = Z0 + Zn;
    bool colptr_done = false;
    {
        Span sp(h, ESP_ST_COLPTR);
        sp.add(espscan::exclusive<u64, true>(h->stream, colend + c0, colend + c0, ccnt, colend + N1));
    }
    CK(ensure(h, h->rowval2, sizeof(i64) * (size_t)Zt));
    CK(ensure(h, h->nzval2, sizeof(double) * (size_t)Zt));
    if (h->force_path == ESP_PATH_MERGE_PATH_JOIN) {
        // (test hook: the merge-path join over a per-entry column array, kept as a second implementation of the same join)
        if (Z0 >= 0xFFFFFFF0ll) FAIL(h, ESP_ERR_UNSUPPORTED, "esp_flush: CSC too large for the 32-bit column index");
        {
            Span sp(h, ESP_ST_COLPTR);
            const i64 hn = Z0 + 1;  // column index of every stored entry
            CK(ensure(h, h->heads, sizeof(u32) * (size_t)(hn + espscan::workspace_elems(hn))));
            u32 *heads = (u32 *)h->heads.p;
            HIPCK(h, hipMemsetAsync(heads, 0, sizeof(u32) * (size_t)hn, h->stream));
            hipLaunchKernelGGL(espfold::col_heads_k, dim3(grid_for(ccnt - 1, 256)), dim3(256), 0, h->stream, (const i64 *)h->colptr.p, c0, ccnt - 1,
                               heads);
            sp.add(1 + espscan::exclusive<u32, true>(h->stream, heads, heads, hn, heads + hn));
        }
        Span sp(h, ESP_ST_MERGE);
        espmerge::Args a;
        a.old_col = (const u32 *)h->heads.p + 1;
        a.old_row = (const i64 *)h->rowval.p;
        a.old_val = (const double *)h->nzval.p;
        a.Z0 = Z0;
        a.new_key = new_key;
        a.new_val = new_val;
        a.Zn = Zn;
        a.rb = h->L.rb;
        a.out_row = (i64 *)h->rowval2.p;
        a.out_val = (double *)h->nzval2.p;
        hipLaunchKernelGGL(espmerge::merge_k, dim3(grid_for(Zt, espmerge::TILE)), dim3(espmerge::THREADS), 0, h->stream, a);
        sp.add(1);
    } else {
        // column-tiled join: every stored and every new entry finds its own place in its merged column.  The stored
        // entries outside the flush's column range (a shard's window) keep their order: in front of the range they
        // stay where they are, behind it they move up by Zn.
        Span sp(h, ESP_ST_MERGE);
        const i64 ncols = ccnt - 1;
        if (windowed(h)) {
            // (win_excl: nothing is stored outside the window)
        } else if (c0 != 0 || ncols != h->n) {
            FAIL(h, ESP_ERR_STATE, "esp_flush: internal error (column range of the join)");
        }
        espmerge::ColArgs a;
        a.old_colptr = (const i64 *)h->colptr.p;
        a.old_row = (const i64 *)h->rowval.p;
        a.old_val = (const double *)h->nzval.p;
        a.newstart = (const u64 *)colend;
        a.new_key = new_key;
        a.new_val = new_val;
        a.rb = h->L.rb;
        a.c_begin = c0;
        a.ncols = ncols;
        a.out_row = (i64 *)h->rowval2.p;
        a.out_val = (double *)h->nzval2.p;
        a.out_colptr = nullptr;
        if (!windowed(h) && h->force_path != ESP_PATH_MERGE_PATH_JOIN) {
            // (its tiles hold both summands of the new colptr: written into a second array, swapped below)
            CK(ensure(h, h->colptr2, sizeof(i64) * (size_t)(h->n + 1)));
            a.out_colptr = (i64 *)h->colptr2.p;
            colptr_done = true;
        }
        hipLaunchKernelGGL((espmerge::colmerge_k<espmerge::CT, espmerge::CT_SCAP>), dim3(grid_for(ncols, espmerge::CT)), dim3(espmerge::THREADS), 0,
                           h->stream, a);
        sp.add(1);
    }
    if (colptr_done) {
        std::swap(h->colptr, h->colptr2);
    } else {
        Span sp(h, ESP_ST_COLPTR);
        hipLaunchKernelGGL(espfold::colptr_finish_k, dim3(grid_for(ccnt, 256)), dim3(256), 0, h->stream, (const u64 *)colend + c0,
                           (const i64 *)h->colptr.p + c0, ccnt, (i64 *)h->colptr.p + c0);
        sp.add(1);
    }
    std::swap(h->rowval, h->rowval2);
    std::swap(h->nzval, h->nzval2);
    h->nnz = Zt;
    h->pattern_version++, h->values_version++;
    return ESP_OK;
}

int32_t prepare_outputs(esp_handle *h, i64 Z0, i64 Zn) {
    const i64 N1 = h->n + 1;
    CK(ensure(h, h->colend, sizeof(u64) * (size_t)(N1 + espscan::workspace_elems(N1))));
    HIPCK(h, hipMemsetAsync(h->colend.p, 0, sizeof(u64) * (size_t)N1, h->stream));
    if (Z0 == 0) {
        CK(ensure(h, h->rowval, sizeof(i64) * (size_t)Zn));
        CK(ensure(h, h->nzval, sizeof(double) * (size_t)Zn));
    } else {
        CK(ensure(h, h->newkey, sizeof(u64) * (size_t)Zn));
        CK(ensure(h, h->newval, sizeof(double) * (size_t)Zn));
    }
    return ESP_OK;
}

// fast path: LDS bucket kernel over the MSD segments; writes the final arrays itself
int32_t flush_local(esp_handle *h, const Sorted &st, int mode, i64 *Zn_out) {
    const i64 Z0 = h->nnz;
    const i64 N1 = h->n + 1;
    // Segments behind the last column hold nothing (a matrix of 10^7 columns fills 60 % of the 2^24 its column bits span:
    // 40 % of the segments, each of which would still draw a ticket, resolve its offset and leave): the launch ends at
    // the segment of the last column; that segment checks that every entry lies in front of its end (Args::total).
    int S = st.S;
    i64 total_check = -1;
    if (st.npieces == 0 && st.total >= 0 && st.seg_start && st.rem_bits >= h->L.rb && st.rem_bits - h->L.rb < 40 && S > 1) {
        const int clb0 = st.rem_bits - h->L.rb;
        // (the segments cut the key window: its first column, and the column behind its last)
        const i64 c_first = (i64)(h->win_base >> h->L.rb), c_last = (i64)((h->win_base + h->win_span) >> h->L.rb);
        const i64 need = ceil_div<i64>(c_last - c_first, (i64)1 << clb0);
        if (need >= 1 && need < (i64)S) {
            S = (int)need;
            total_check = st.total;
        }
    }
    // esp_flush normalised the buffers: data in keys/vals, scratch pair = keys2/vals2
    u64 *tk = (u64 *)h->keys2.p;
    double *tv = (double *)h->vals2.p;
    // look-back granules: one per segment | error flag | longest run | one per group of 256 segments, rounded up to a 4 KiB
    // page; behind them a page of its own for the ticket counter (every workgroup draws from it while others poll the
    // granules: on one line with them the draws cost the headline's bucket kernel 0.18 of 1.55 ms) -- one memset clears
    // everything
    const i64 n_gs = ((i64)S >> 8) + 2;  // (local.hpp: LB_SHIFT = 8)
    const i64 G = (((i64)S + 2 + n_gs + 511) & ~(i64)511) - (S + 2);  // (granules behind status[S + 1])
    const i64 tick_at = S + 2 + G + 256;
    CK(ensure(h, h->segout, sizeof(u64) * (size_t)(S + 2 + G + 512)));
    u64 *status = (u64 *)h->segout.p;
    const size_t status_bytes = sizeof(u64) * (size_t)(S + 2 + G + 512);
    HIPCK(h, hipMemsetAsync(status, 0, status_bytes, h->stream));
    CK(ensure(h, h->colend, sizeof(u64) * (size_t)(N1 + espscan::workspace_elems(N1))));
    esplocal::Args a;
    memset(&a, 0, sizeof a);
    const char *stop_env = esp_exp_env("ESP_LOCAL_STOP");  // (experiments build only: ablation of the bucket kernel)
    // A fresh matrix whose segments are whole blocks of <= CL_MAX columns that start at the first column of the
    // range this flush can touch (all columns, or the column window of a shard) and cover it: every segment writes
    // the colptr of its own columns (no column-end marks, no memset and no scan over the columns).
    // force_path 13: marks + scan.
    bool direct = false;
    const i64 col_begin = windowed(h) ? h->wc0 : 0, col_end = windowed(h) ? h->wc1 : h->n;
    {
        const int clb = st.rem_bits - h->L.rb;
        const u64 seg_base = st.has_base ? st.base : st.npieces > 0 ? h->part_base : h->win_base;
        direct = Z0 == 0 && clb >= 0 && clb <= esplocal::CL_MAX_BITS && h->force_path != ESP_PATH_RADIX_TAIL_ONLY && h->force_path != ESP_PATH_COLPTR_BY_SCAN && !stop_env &&
                 seg_base == ((u64)col_begin << h->L.rb) && col_begin + ((i64)S << clb) >= col_end;
    }
    h->last_colptr_direct = direct ? 1 : 0;
    if (!direct) {
        i64 c0, cnt;
        col_range(h, &c0, &cnt);
        HIPCK(h, hipMemsetAsync((u64 *)h->colend.p + c0, 0, sizeof(u64) * (size_t)cnt, h->stream));
    }
    // (a failed flush must not leave a half-written colptr behind)
    auto restore_colptr = [&]() {
        if (direct) {
            hipLaunchKernelGGL(fill_i64_k, dim3(grid_for(N1, 256)), dim3(256), 0, h->stream, (i64 *)h->colptr.p, N1, (i64)1);
            h->tail_stale = false;
            h->ones_pending = false;
        }
    };
    // The small variant of the bucket kernel (3 workgroups per CU) serves segments of at most 3072 entries over at most
    // 256 columns whose column runs the register tiers take; a segment with longer runs (it cannot know before it counts)
    // goes through the variant's slow tier, and what it reports sends the next flushes to the regular kernel.
    // force_path 18: never.
    bool small_variant = false;
    {
        const int clb = st.rem_bits - h->L.rb;
        // (what the handle's last flush saw decides; a handle without history tries it when the columns hold few entries
        // on average -- a stencil's 12, not a 3-D FEM mesh's 120)
        const double per_col = (double)h->count / (double)std::max<i64>(col_end - col_begin, 1);
        // (runs of 17..24 want the 24-input network, which does not fit the variant's 80 registers: measured 4.7 against
        // 4.0 ms on 2-D FEM)
        const bool runs_fit = h->seen_maxrun > 0 ? h->seen_maxrun <= 16 : per_col <= 16.0;
        small_variant = st.maxlen <= 6 * esplocal::THREADS && clb >= 0 && clb <= 8 &&
                        st.rem_bits <= esplocal::REG_MAX_REM && runs_fit && h->force_path != ESP_PATH_RADIX_TAIL_ONLY &&
                        h->force_path != ESP_PATH_NO_SMALL_VARIANT && !stop_env;
    }
    h->last_local_small = small_variant ? 1 : 0;
    std::function<int32_t(bool)> launch_all;
    bool used_g3 = false;
    bool want_wide = h->g3_wide && h->force_path != ESP_PATH_NO_WIDE_GROUP3;
    {
        Span sp(h, ESP_ST_LOCAL);
        a.kind32 = (u32)((st.key_bytes == 4 || st.p32_piece >= 0 || st.all32) ? st.kind : 0);
        a.k32_piece = st.p32_piece;
        a.k32_lo = st.p32_lo;
        h->last_key_bytes = (st.p32_piece >= 0 || st.all32) ? 4 : st.key_bytes;
        a.fb = ((st.npieces == 0 && st.key_bytes == 4) || (st.npieces > 0 && st.p32_piece >= 0 && st.own_fine)) ? st.fb : 0;
        a.own_fine = a.fb > 0 ? st.own_fine : nullptr;
        a.colptr_out = direct ? (i64 *)h->colptr.p : nullptr;
        a.late_total = h->force_path == ESP_PATH_LATE_TOTAL ? 1 : 0;  // 36: test hook, group3_k publishes a segment's total after the fold
        a.no_group = h->force_path == ESP_PATH_NO_GROUP_TIER ? 1 : 0;  // 24: test hook, long column runs through the radix tier
        // (every pending entry was noted with one kind; pieces of other ranks carry kinds this handle has not seen)
        a.kind_all = (st.npieces == 0 && h->kind_uniform >= 0 && h->kind_noted == h->count && h->force_path != ESP_PATH_GENERIC_FOLD) ? h->kind_uniform : -1;
        a.expect_hits = st.expect_hits >= 0 ? st.expect_hits : (h->seen_hits ? 1 : 0);
        a.col_end = col_end;
        a.n_cols = h->n;
        a.keys_in = st.sk;
        a.vals_in = st.sv;
        a.seg_start = st.seg_start;
        a.S = S;
        a.rem_bits = st.rem_bits;
        a.base = st.has_base ? st.base : st.npieces > 0 ? h->part_base : h->win_base;
        a.rb = h->L.rb;
        {
            const int clb = st.rem_bits - h->L.rb;
            a.cl_bits = (clb >= 0 && clb <= esplocal::CL_MAX_BITS && h->force_path != ESP_PATH_RADIX_TAIL_ONLY) ? clb : -1;
        }
        // a segment is a whole number of columns when the prefix does not reach into the row bits
        a.col_aligned = st.rem_bits >= h->L.rb ? 1 : 0;
        a.csc = espfold::Csc{(const i64 *)h->colptr.p, (const i64 *)h->rowval.p, (double *)h->nzval.p, Z0};
        a.mode = mode;
        a.out_row = (i64 *)tk;
        a.out_key = tk;
        a.out_val = tv;
        a.colend = (u64 *)h->colend.p;
        a.status = status;
        a.gstatus = status + S + 2;
        a.npieces = st.npieces;
        a.pieces_dense = st.pieces_dense ? 1 : 0;
        a.total = total_check;
        a.pstart = st.pstart;
        a.ptab = st.ptab;
        a.ticket = (u32 *)(status + tick_at);
        a.err = (u32 *)(status + S) + 1;
        a.maxrun_seen = (u32 *)(status + S) + 2;  // (zeroed with the granules)
        {
            a.stop_after = stop_env ? atoi(stop_env) : 0;
            a.stamps = nullptr;
            a.hits_out = nullptr;
            if (esp_exp_env("ESP_LOCAL_STAMPS")) {  // diagnostics (experiments build): per-segment phase stamps, dumped to a file
                CK(ensure(h, h->heads, sizeof(u64) * (size_t)S * 16));
                HIPCK(h, hipMemsetAsync(h->heads.p, 0, sizeof(u64) * (size_t)S * 16, h->stream));
                a.stamps = (unsigned long long *)h->heads.p;
            }
        }
        const i64 max_grid = h->force_path == ESP_PATH_MANY_LAUNCHES ? 64 : esplocal::MAX_GRID;  // 4: test hook, many launches
        launch_all = [&, max_grid](bool allow_g3) -> int32_t {
        used_g3 = false;
        for (i64 first = 0; first < S; first += max_grid) {
            const unsigned grid = (unsigned)std::min<i64>(max_grid, S - first);
            a.first = first;
            // (key format: 0 packed, 1 four-byte keys of one kind, 2 four-byte keys that are all UPDATEs)
            // (3: packed keys whose kinds are all UPDATE -- the pieces of a shard)
            // (4 / 5: pieces of which one -- a shard's own range -- holds 4-byte keys; 5: everything is an UPDATE)
            // (6 / 7: pieces that all hold 4-byte keys of one kind -- a producer's batch and its tail; 7: UPDATE)
            const bool generic = h->force_path == ESP_PATH_GENERIC_FOLD;
            const int keys = st.npieces > 0 ? (st.all32 ? (st.kind == ESP_UPDATE && !generic ? 7 : 6)
                                               : st.p32_piece >= 0 ? (st.all_update && !generic ? 5 : 4)
                                                                   : (st.all_update && !generic ? 3 : 0))
                                            : st.key_bytes != 4 ? 0 : (st.kind == ESP_UPDATE && !generic ? 2 : 1);
            h->last_fold_update = (keys == 2 || keys == 3 || keys == 5 || keys == 7) ? 1 : 0;  // (local.hpp: UPD)
            // the longest column run: what the handle's last flush met, or -- no history -- the pending entries per column (a
            // P1 mesh in 2-D: 24, in 3-D: 120; a wrong guess costs that one flush the radix tier)
            const double longest = h->seen_maxrun > 0 ? (double)h->seen_maxrun : (double)h->count / (double)std::max<i64>(col_end - col_begin, 1);
            // (the group-tier kernel for runs of more than 16 entries -- up to 32: its four-lane form; 24: test hook, never)
            const bool grp = st.npieces == 0 && !small_variant && keys <= 2 && longest > 16.0 && h->force_path != ESP_PATH_NO_GROUP_TIER &&
                             h->force_path != ESP_PATH_RADIX_TAIL_ONLY;
            // (else the variant with the 24-input register tier for runs of 17 .. 24 -- shard pieces; 26: test hook, never)
            const bool big = longest > 16.0 && longest <= (double)esplocal::REG_RUN && h->force_path != ESP_PATH_NO_BIG_VARIANT;
            esplocal::Variant var{Z0 == 0, st.npieces > 0, big && !small_variant && !grp, small_variant, keys};
            var.shortg = grp && longest <= 32.0;
            var.grp = grp;
            // (the group tier with three workgroups per CU -- group3.hpp: a fresh matrix, 4-byte keys of one kind, segments of at
            // most 256 whole columns whose (local column, row) fits 32 bits, runs of at most 128 entries; a segment it does not
            // take makes the flush run again with the kernels above; 30: test hook, never)
            var.g3 = allow_g3 && grp && st.fb == 0 && Z0 == 0 && (keys == 1 || keys == 2) && a.cl_bits >= 0 && a.cl_bits <= esplocal::G3_CL_BITS &&
                     a.cl_bits + a.rb <= 32 && a.rb <= 30 && longest <= 128.0 && !h->g3_off && !a.no_group && !a.stop_after &&
                     h->force_path != ESP_PATH_NO_GROUP3;
            // (packed keys of ONE known adding kind whose bits below the prefix fit 32 -- a shuffled stream of triplets after the
            // flush's own passes: the same kernel, the keys narrowed as they are loaded; not its wide form)
            const bool k64_ok = keys == 0 && st.key_bytes == 8 && st.npieces == 0 && st.rem_bits <= 32 && !want_wide &&
                                (a.kind_all == ESP_UPDATE || a.kind_all == ESP_RAWUPDATE) && h->force_path == ESP_PATH_AUTO;
            if (!var.g3 && k64_ok && allow_g3 && grp && Z0 == 0 && a.cl_bits >= 0 && a.cl_bits <= esplocal::G3_CL_BITS && a.cl_bits + a.rb <= 32 &&
                a.rb <= 30 && longest <= 128.0 && !h->g3_off && !a.no_group && !a.stop_after) {
                var.g3 = var.g3k64 = true;
                a.kind32 = (u32)a.kind_all;
            }
            var.g3wide = var.g3 && want_wide;
            used_g3 = used_g3 || var.g3;
            if (st.lazy) {  // sorted ITEM records: group3_k's fused form (group3_items.hpp) or nothing
                if (!var.g3 || var.g3wide || !esplocal::launch_group3_items(*st.lazy, grid, h->stream, a)) return ESP_RETRY_EXPANDED;
                continue;
            }
            if (!esplocal::launch(var, grid, h->stream, a)) FAIL(h, ESP_ERR_STATE, "esp_flush: internal error (no bucket kernel for this flush)");
        }
        return ESP_OK;
        };
    }
    auto read_back = [&]() -> int32_t {
        HIPCK(h, hipMemcpyAsync(h->pin_scalar, status + (S - 1), 24, hipMemcpyDeviceToHost, h->stream));  // last granule | ticket, err | maxrun
        HIPCK(h, hipMemcpyAsync(h->pin_scalar + 3, (u32 *)h->misc.p + 60, 4, hipMemcpyDeviceToHost, h->stream));
        HIPCK(h, hipStreamSynchronize(h->stream));
        return ESP_OK;
    };
    auto reset_launch_state = [&]() -> int32_t {
        HIPCK(h, hipMemsetAsync(status, 0, status_bytes, h->stream));
        if (!direct) {
            i64 c0, cnt;
            col_range(h, &c0, &cnt);
            HIPCK(h, hipMemsetAsync((u64 *)h->colend.p + c0, 0, sizeof(u64) * (size_t)cnt, h->stream));
        }
        return ESP_OK;
    };
    // A re-assembly over the pattern the same mesh built (a ROUTED flush of 4-byte-key additions with long column runs, and
    // the handle's last flush over the pattern hit): group3_k's re-assembly form -- every (col,row) of a column's sorted run
    // IS the column's stored entry of the same rank; the sums go to a second value array, swapped in when no column
    // objected (bit 64) -- else nothing has happened and the general kernels below take the flush.  force_path 34: never.
    {
        const bool generic = h->force_path == ESP_PATH_GENERIC_FOLD;
        const int keys = st.key_bytes != 4 ? 0 : (st.kind == ESP_UPDATE && !generic ? 2 : 1);
        const double longest = h->seen_maxrun > 0 ? (double)h->seen_maxrun : (double)h->count / (double)std::max<i64>(col_end - col_begin, 1);
        const bool hits_expected = st.expect_hits >= 0 ? st.expect_hits == 1 : h->seen_hits;
        bool try_hits = Z0 > 0 && st.fb == 0 && mode == ESP_FLUSH_ROUTED && st.npieces == 0 && (keys == 1 || keys == 2) && (st.kind == ESP_UPDATE || st.kind == ESP_RAWUPDATE) &&
                        a.kind_all == st.kind && a.cl_bits >= 0 && a.cl_bits <= esplocal::G3_CL_BITS && a.cl_bits + a.rb <= 32 && a.rb <= 30 &&
                        longest > 16.0 && longest <= 128.0 && hits_expected && !h->hits_off && !a.no_group && !a.stop_after && !small_variant &&
                        h->wc0 == 0 && h->wc1 == h->n &&  // (the segments cover every column: what they do not write into the second array would be lost)
                        h->force_path != ESP_PATH_NO_GROUP_TIER && h->force_path != ESP_PATH_RADIX_TAIL_ONLY && h->force_path != ESP_PATH_NO_GROUP3 &&
                        h->force_path != ESP_PATH_NO_HITS_KERNEL && h->force_path != ESP_PATH_MANY_LAUNCHES && (i64)S <= esplocal::MAX_GRID;
        for (int attempt = 0; attempt < 2 && try_hits; attempt++) {
            CK(ensure(h, h->nzval2, sizeof(double) * (size_t)Z0));
            a.hits_out = (double *)h->nzval2.p;
            a.first = 0;
            esplocal::Variant var{false, false, false, false, keys};
            var.grp = var.g3 = var.g3hits = true;
            var.g3wide = want_wide;
            {
                Span sp(h, ESP_ST_LOCAL);
                if (st.lazy) {  // (sorted item records: the fused form -- plain rows only -- or the caller expands them)
                    if (want_wide || !esplocal::launch_group3_items(*st.lazy, (unsigned)S, h->stream, a, true)) return ESP_RETRY_EXPANDED;
                } else if (!esplocal::launch(var, (unsigned)S, h->stream, a))
                    FAIL(h, ESP_ERR_STATE, "esp_flush: internal error (no re-assembly kernel for this flush)");
                sp.add(1);
            }
            CK(read_back());
            const u32 e = (u32)(h->pin_scalar[1] >> 32);
            // (accepted only when NOTHING objected: no refused segment (8), no column that is not the stored one (64), no look-back
            // error (1 / 2 / 4), no entry outside a declared window -- anything else falls through to the general kernels' checks)
            if ((e & (1u | 2u | 4u | 8u | 64u)) == 0u && (u32)h->pin_scalar[3] == 0u) {
                std::swap(h->nzval, h->nzval2);
                h->last_group3 = want_wide ? 4 : 3;
                h->seen_maxrun = (int)(u32)(h->pin_scalar[2] & 0xFFFFFFFFull);
                h->seen_hits = true;
                h->last_lazy_items = st.lazy ? 1 : 0;
                *Zn_out = 0;
                return ESP_OK;
            }
            CK(reset_launch_state());
            if (st.lazy) return ESP_RETRY_EXPANDED;  // (nothing has happened: the entries, then whatever form takes them)
            if ((e & 8u) && (e & 16u) && !(e & 32u) && !(e & 64u) && !want_wide && h->force_path != ESP_PATH_NO_WIDE_GROUP3) {
                want_wide = true;  // (refused for its rows alone: once more in the wide form)
                h->g3_wide = true;
                continue;
            }
            h->hits_off = true;  // (not a re-assembly of the stored pattern, or shapes the kernel does not take: not tried again)
            try_hits = false;
        }
    }
    {
        Span sp(h, ESP_ST_LOCAL);
        CK(launch_all(true));
        sp.add(1);
    }
    CK(read_back());
    h->last_lazy_items = 0;
    h->last_sum_join = 0;
    if (st.lazy) {
        // a segment the fused kernel refuses (a column run above 128, rows spread over more than 2^18): nothing of a
        // fresh-matrix flush has taken effect -- the caller expands the items and comes back with the entries
        if ((u32)(h->pin_scalar[1] >> 32) & 8u) return ESP_RETRY_EXPANDED;
        h->last_lazy_items = 1;
    }
    if (used_g3 && !want_wide && h->force_path != ESP_PATH_NO_WIDE_GROUP3) {
        const u32 e = (u32)(h->pin_scalar[1] >> 32);
        if ((e & 8u) && (e & 16u) && !(e & 32u)) {
            // every segment the kernel refused, it refused for its ROWS alone (spread over more than 2^18: a mesh numbered
            // without locality): nothing of a fresh-matrix flush has taken effect -- once more with the kernel's wide form
            // (full rows in LDS, every run sorted twice), which serves this handle from now on
            want_wide = true;
            h->g3_wide = true;
            CK(reset_launch_state());
            {
                Span sp(h, ESP_ST_LOCAL);
                CK(launch_all(true));
                sp.add(1);
            }
            CK(read_back());
        }
    }
    if (used_g3 && ((u32)(h->pin_scalar[1] >> 32) & 8u)) {
        // a segment the three-workgroup group kernel does not take (a longer run, rows too far apart): nothing of a
        // fresh-matrix flush has taken effect -- once more with the general kernels, and they serve this handle from now on
        h->g3_off = true;
        HIPCK(h, hipMemsetAsync(status, 0, status_bytes, h->stream));
        if (!direct) {
            i64 c0, cnt;
            col_range(h, &c0, &cnt);
            HIPCK(h, hipMemsetAsync((u64 *)h->colend.p + c0, 0, sizeof(u64) * (size_t)cnt, h->stream));
        }
        {
            Span sp(h, ESP_ST_LOCAL);
            CK(launch_all(false));
            sp.add(1);
        }
        CK(read_back());
    }
    h->last_group3 = used_g3 ? (want_wide ? 2 : 1) : 0;

    if ((u32)h->pin_scalar[3]) {
        restore_colptr();
        FAIL(h, ESP_ERR_STATE, "esp_flush: a pending entry lies outside the declared column window");
    }
    if (a.stamps) {
        std::vector<u64> st((size_t)S * 16);
        HIPCK(h, hipMemcpy(st.data(), a.stamps, sizeof(u64) * st.size(), hipMemcpyDeviceToHost));
        if (FILE *f = fopen(esp_exp_env("ESP_LOCAL_STAMPS"), "wb")) {
            fwrite(st.data(), sizeof(u64), st.size(), f);
            fclose(f);
        }
    }
    if (direct && !windowed(h)) h->ones_pending = false;  // (the bucket kernel wrote every entry of colptr)
    const u32 lookback_err = (u32)(h->pin_scalar[1] >> 32);
    h->seen_maxrun = (int)(u32)(h->pin_scalar[2] >> 0 & 0xFFFFFFFFull);
    if (lookback_err & 7u) restore_colptr();
    if (lookback_err & 2u) FAIL(h, ESP_ERR_STATE, "esp_flush: a pending entry lies outside the declared column window (bucket)");
    if (lookback_err & 1u) FAIL(h, ESP_ERR_HIP, "esp_flush: look-back chain timed out inside the bucket kernel");
    if (lookback_err & 4u) FAIL(h, ESP_ERR_HIP, "esp_flush: internal error (early segment total differs from the folded total)");
    const i64 Zn = (i64)(h->pin_scalar[0] & esplocal::ST_VAL);
    *Zn_out = Zn;
    // (history for the next flush over a stored pattern: fewer than a quarter of the entries opened a new position)
    if (Z0 > 0 && st.expect_hits < 0 && st.total > 0) h->seen_hits = Zn * 4 <= st.total;
    if (a.stop_after || Zn == 0) return ESP_OK;
    if (Z0 == 0) {
        // the scratch pair now holds rowval/nzval: rotate the buffers instead of copying.  The old
        // (empty) CSC arrays become the next flush's scratch pair: bring them to the same capacity
        // once, so that the rotation never shrinks the scratch pair (a 10 GB hipMalloc per flush
        // costs more than the flush itself)
        CK(ensure(h, h->rowval, h->keys2.bytes));
        CK(ensure(h, h->nzval, h->vals2.bytes));
        std::swap(h->rowval, h->keys2);
        std::swap(h->nzval, h->vals2);
        if (direct) {  // colptr is complete (behind a column window it is refreshed lazily, as after the scan)
            if (windowed(h)) h->tail_stale = h->wc1 < h->n;
            h->nnz = Zn;
            h->pattern_version++, h->values_version++;
            return ESP_OK;
        }
        return finish_csc(h, 0, Zn, nullptr, nullptr);
    }
    return finish_csc(h, Z0, Zn, (const u64 *)tk, (const double *)tv);
}

// general path: finish with a full stable LSD sort and the global fold (any run length)
int32_t flush_global(esp_handle *h, int mode, i64 *Zn_out) {
    const i64 E = h->count;
    const i64 Z0 = h->nnz;
    const u64 *sk;
    const double *sv;
    CK(sort_pending_lsd(h, &sk, &sv));  // sorted data now in h->keys/h->vals; keys2/vals2 are scratch
    u32 *flag = (u32 *)h->vals2.p;        // E+1 u32 fits in E doubles (E>=1)
    double *fval = (double *)h->keys2.p;  // E doubles
    espfold::Csc csc{(const i64 *)h->colptr.p, (const i64 *)h->rowval.p, (double *)h->nzval.p, Z0};
    {
        Span sp(h, ESP_ST_FOLD);
        hipLaunchKernelGGL(espfold::fold_k, dim3(grid_for(E + 1, espfold::THREADS)), dim3(espfold::THREADS), 0, h->stream, sk, sv, E,
                           csc, h->L.rb, mode, flag, fval);
        sp.add(1);
    }
    {
        Span sp(h, ESP_ST_SCAN);
        int l = 0;
        CK(scan_inplace<u32, false>(h, flag, E + 1, h->hist, &l));
        sp.add(l);
    }
    HIPCK(h, hipMemcpyAsync(h->pin_scalar, flag + E, 4, hipMemcpyDeviceToHost, h->stream));
    HIPCK(h, hipStreamSynchronize(h->stream));
    const i64 Zn = (i64) * (u32 *)h->pin_scalar;
    *Zn_out = Zn;
    if (Zn == 0) return ESP_OK;
    CK(prepare_outputs(h, Z0, Zn));
    {
        Span sp(h, ESP_ST_FOLD);
        if (Z0 == 0)
            hipLaunchKernelGGL((espfold::compact_k<true>), dim3(grid_for(E, espfold::THREADS)), dim3(espfold::THREADS), 0, h->stream, sk,
                               (const double *)fval, E, (const u32 *)flag, h->L.rb, (i64 *)h->rowval.p, (u64 *)nullptr,
                               (double *)h->nzval.p, (u64 *)h->colend.p);
        else
            hipLaunchKernelGGL((espfold::compact_k<false>), dim3(grid_for(E, espfold::THREADS)), dim3(espfold::THREADS), 0, h->stream, sk,
                               (const double *)fval, E, (const u32 *)flag, h->L.rb, (i64 *)nullptr, (u64 *)h->newkey.p,
                               (double *)h->newval.p, (u64 *)h->colend.p);
        sp.add(1);
    }
    return finish_csc(h, Z0, Zn, (const u64 *)h->newkey.p, (const double *)h->newval.p);
}

__global__ void piece_totals_k(const i64 *__restrict__ pstart, int P, i64 nb, unsigned long long *__restrict__ maxlen,
                               unsigned long long *__restrict__ negative);

// The pending entries start at pend_off of the buffer (behind a batch that was flushed by itself): move them to the front,
// in chunks of at most pend_off entries (source and destination of one copy never overlap).  On failure they are dropped:
// the batch in front of them is in the matrix already and must not be applied again.
int32_t settle_offset(esp_handle *h) {
    const i64 off = h->pend_off, T = h->count;
    if (off == 0) return ESP_OK;
    h->pend_off = 0;
    Span sp(h, ESP_ST_COPY);
    for (i64 at = 0; at < T; at += off) {
        const i64 c = std::min(off, T - at);
        hipError_t e1 = hipMemcpyAsync((u64 *)h->keys.p + at, (const u64 *)h->keys.p + off + at, sizeof(u64) * (size_t)c, hipMemcpyDeviceToDevice, h->stream);
        if (e1 == hipSuccess)
            e1 = hipMemcpyAsync((double *)h->vals.p + at, (const double *)h->vals.p + off + at, sizeof(double) * (size_t)c, hipMemcpyDeviceToDevice, h->stream);
        if (e1 != hipSuccess) {
            h->count = 0;
            pending_changed(h);
            FAIL(h, ESP_ERR_HIP, "esp_flush: %s while moving the entries behind a flushed batch; they were dropped", hipGetErrorString(e1));
        }
        sp.add(2);
    }
    return ESP_OK;
}

// bucket starts of a tail sorted by its prefix digit: first position whose digit is >= d, for d = 0 .. NB
__global__ void tail_bucket_starts_k(const u64 *__restrict__ keys, i64 T, u64 base, u64 span, int shift, i64 NB, i64 *__restrict__ out) {
    const i64 d = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (d > NB) return;
    i64 lo = 0, hi = T;
    while (lo < hi) {
        const i64 mid = (lo + hi) >> 1;
        u64 kn = (keys[mid] >> ESP_TAG_BITS) - base;
        kn = kn < span ? kn : span - 1;
        if ((i64)(kn >> shift) < d)
            lo = mid + 1;
        else
            hi = mid;
    }
    out[d] = lo;
}

// A producer's bucket-ordered batch with packed entries appended behind it (a re-assembly whose mesh gained a few
// couplings: the generator's batch, then the new positions): the tail alone goes through the run-based partition with
// the batch's prefix bits, and the bucket kernel reads every segment as two pieces -- the batch's bucket (4-byte keys
// when the producer wrote them), then the tail's.  Stream order is kept: the tail's entries come after the batch's.
int32_t flush_pre_tail(esp_handle *h, int mode, i64 *Zn, bool *served) {
    *served = false;
    const esp_handle::PrePart pp = h->pre;
    const i64 E0 = pp.E, T = pp.tail, NB = (i64)1 << pp.pb;
    CK(ensure(h, h->newkey, sizeof(u64) * (size_t)T));
    CK(ensure(h, h->newval, sizeof(double) * (size_t)T));
    CK(ensure(h, h->seg[0], sizeof(i64) * (size_t)(NB + 1)));
    CK(ensure(h, h->tilef[0], sizeof(u64) * (size_t)(NB + 1 + espscan::workspace_elems(NB + 1))));
    CK(ensure(h, h->misc, 256));
    unsigned long long *d_maxlen = (unsigned long long *)h->misc.p + 24;
    HIPCK(h, hipMemsetAsync(d_maxlen, 0, 64, h->stream));  // maxlen .. the four flag words (run_partition)
    bool took = false, tiles = false;
    i64 ml = 0;
    // (a tail of the batch's kind goes out as 4-byte keys too: the bucket kernel then reads nothing but such keys)
    const bool tail32 = pp.key_bytes == 4 && h->kind_uniform == pp.kind && h->kind_noted == h->count;
    int tail_bytes = 8;
    CK(run_partition(h, (const u64 *)h->keys.p + E0, (const double *)h->vals.p + E0, (u64 *)h->newkey.p, (double *)h->newval.p, pp.K,
                     pp.pb, (i64 *)h->seg[0].p, (u64 *)h->tilef[0].p, &tiles, &took, &ml, nullptr, 0, tail32, &tail_bytes, T));
    if (!took) {
        // no pre-sorted stream (a few entries spread over many buckets, say): stable 8-bit passes over the prefix bits of the
        // tail alone, least significant first; the last one lands in newkey/newval (keys2/vals2, the bucket kernel's
        // output, serve as the other half of the ping-pong until then)
        CK(ensure(h, h->keys2, sizeof(u64) * (size_t)(E0 + T)));
        CK(ensure(h, h->vals2, sizeof(double) * (size_t)(E0 + T)));
        CK(ensure(h, h->segs, sizeof(i64) * 8));
        i64 *segs = (i64 *)h->segs.p;
        const i64 TR = ceil_div<i64>(T, espradix::TILE);
        hipLaunchKernelGGL(set_i64_k, dim3(1), dim3(1), 0, h->stream, segs, (i64)0, T, (i64)0, TR);
        HIPCK(h, hipMemsetAsync(d_maxlen, 0, 64, h->stream));
        const int npass = (pp.pb + 7) / 8;
        const u64 *ki = (const u64 *)h->keys.p + E0;
        const double *vi = (const double *)h->vals.p + E0;
        bool to_new = (npass & 1) != 0;
        for (int done = 0; done < pp.pb; done += 8) {
            espradix::Pass p;
            p.keys_in = ki;
            p.vals_in = vi;
            p.keys_out = to_new ? (u64 *)h->newkey.p : (u64 *)h->keys2.p;
            p.vals_out = to_new ? (double *)h->newval.p : (double *)h->vals2.p;
            p.seg_start = segs;
            p.tile_first = segs + 2;
            p.S = 1;
            p.owner_P = 0;
            p.owner_n = 1;
            p.colshift = 0;
            p.base = h->win_base;
            p.span = h->win_span;
            p.err = (u32 *)h->misc.p + 60;
            p.shift = pp.K - pp.pb + done;
            p.bits = std::min(8, pp.pb - done);
            CK(partition_pass(h, p, TR));
            ki = p.keys_out;
            vi = p.vals_out;
            to_new = !to_new;
        }
        {
            Span sp(h, ESP_ST_SCAN);
            hipLaunchKernelGGL(tail_bucket_starts_k, dim3(grid_for(NB + 1, 256)), dim3(256), 0, h->stream, (const u64 *)h->newkey.p, T,
                               h->win_base, h->win_span, pp.K - pp.pb, NB, (i64 *)h->seg[0].p);
            sp.add(1);
        }
        HIPCK(h, hipMemcpyAsync(h->pin_scalar, (u32 *)h->misc.p + 60, 4, hipMemcpyDeviceToHost, h->stream));
        HIPCK(h, hipStreamSynchronize(h->stream));
        if ((u32)h->pin_scalar[0]) FAIL(h, ESP_ERR_STATE, "esp_flush: a pending entry lies outside the declared column window (partition)");
        h->last_run_order = 0;
    }
    // pointer table (keys | values) | piece starts: batch, tail
    const size_t o_ps = 256 * 8;
    CK(ensure(h, h->piecetab, o_ps + sizeof(i64) * 2 * (size_t)(NB + 1)));
    char *TB = (char *)h->piecetab.p;
    const void *tab[4] = {h->keys.p, h->newkey.p, h->vals.p, h->newval.p};
    i64 *pstart = (i64 *)(TB + o_ps);
    HIPCK(h, hipMemcpyAsync(TB, tab, sizeof(tab), hipMemcpyHostToDevice, h->stream));
    HIPCK(h, hipMemcpyAsync(pstart, h->seg[1].p, sizeof(i64) * (size_t)(NB + 1), hipMemcpyDeviceToDevice, h->stream));
    HIPCK(h, hipMemcpyAsync(pstart + (NB + 1), h->seg[0].p, sizeof(i64) * (size_t)(NB + 1), hipMemcpyDeviceToDevice, h->stream));
    HIPCK(h, hipMemsetAsync(d_maxlen, 0, 16, h->stream));
    {
        Span sp(h, ESP_ST_SCAN);
        hipLaunchKernelGGL(piece_totals_k, dim3(grid_for(NB, 256)), dim3(256), 0, h->stream, (const i64 *)pstart, 2, NB, d_maxlen, d_maxlen + 1);
        sp.add(1);
    }
    HIPCK(h, hipMemcpyAsync(h->pin_scalar, d_maxlen, 8, hipMemcpyDeviceToHost, h->stream));
    HIPCK(h, hipStreamSynchronize(h->stream));  // (tab is read by the copy above)
    const i64 merged = (i64)h->pin_scalar[0];
    if (merged > (i64)esplocal::CAP) return ESP_OK;
    HIPCK(h, hipMemsetAsync((u32 *)h->misc.p + 60, 0, 4, h->stream));
    Sorted st;
    st.sk = (const u64 *)h->keys.p;
    st.sv = (const double *)h->vals.p;
    st.in_primary = true;
    st.S = (int)NB;
    st.seg_start = nullptr;
    st.rem_bits = pp.K - pp.pb;
    st.local_ok = true;
    st.npieces = 2;
    st.all_update = h->kind_uniform == ESP_UPDATE && h->kind_noted == h->count;
    st.ptab = (const void *const *)TB;
    st.pstart = pstart;
    st.pieces_dense = true;
    st.maxlen = merged;
    st.has_base = true;
    st.base = h->win_base;
    if (pp.key_bytes == 4 && tail_bytes == 4) {
        st.all32 = true;
        st.kind = pp.kind;
    } else if (pp.key_bytes == 4) {
        st.p32_piece = 0;
        st.p32_lo = 0;
        st.kind = pp.kind;
    }
    CK(ensure(h, h->keys2, sizeof(u64) * (size_t)(E0 + T)));
    CK(ensure(h, h->vals2, sizeof(double) * (size_t)(E0 + T)));
    CK(flush_local(h, st, mode, Zn));
    *served = true;
    return ESP_OK;
}

// ---- a routed flush that REBUILDS the matrix: the stored CSC as the first piece of every segment ------------------------
// A flush over a stored pattern whose entries mostly open NEW positions (the entries behind a re-assembly's batch: a mesh
// that gained couplings) used to run the bucket kernel against the stored columns (look-ups that miss), emit the new
// entries and join them with the stored matrix in a pass of its own (colmerge_k): the stored rows read twice, the new
// entries written and read again.  But a stored entry is, to the ordered fold, nothing but an entry that came FIRST and
// always creates -- the COO kind (fold.hpp: the first value as it is, whatever follows added or set in call order).  So the
// flush runs as a FRESH one whose segments are two pieces (the PIECES variants the shard exchange uses): the stored
// entries of the segment's columns -- contiguous in the CSC, their keys formed once (csc_keys_k), their values read where
// they lie -- then the pending entries of the segment.  The bucket kernel writes the new rowval / nzval / colptr itself: no
// look-ups, no join.  Config 3's tail (33 M entries behind 117 M stored): tail kernel 0.75 + join 1.23 + scan 0.09 ms -> one
// kernel over 150 M entries.  Taken for short columns only (the PIECES variants have no group tier), whole-column segments,
// ROUTED mode (csc + buffer folds the buffer by itself first); a merged segment above the kernel's capacity: not served,
// the caller takes the look-up + join path.  force_path 40: never.
// keys of the stored entries, entry-parallel (coalesced reads and stores): a workgroup takes 256 columns, their colptr slice in
// LDS, every entry finds its column by a binary search there.  K32: the 32 key bits below the segment prefix (rem_bits <= 32:
// (col << rb | row) mod 2^rem_bits -- what the bucket kernel's P32 piece format reads), else packed keys of kind COO.
template <bool K32>
__global__ __launch_bounds__(256) void csc_keys_k(const i64 *__restrict__ colptr, const i64 *__restrict__ rowval, i64 n, KeyLayout L, int rem_bits,
                                                  void *__restrict__ keys_out) {
    __shared__ i64 cp[257];
    const int t = threadIdx.x;
    const i64 c0 = (i64)blockIdx.x * 256;
    for (int q = t; q <= 256; q += 256) cp[q] = colptr[min(c0 + q, n)] - 1;
    __syncthreads();
    const i64 e0 = cp[0], e1 = cp[256];
    const u64 lowmask = rem_bits >= 64 ? ~0ull : ((1ull << rem_bits) - 1ull);
    for (i64 e = e0 + t; e < e1; e += 256) {
        int lo = 0, hi = 256;  // invariant cp[lo] <= e < cp[hi]
#pragma unroll
        for (int step = 0; step < 8; step++) {
            const int mid = (lo + hi) >> 1;
            if (cp[mid] <= e)
                lo = mid;
            else
                hi = mid;
        }
        const u64 kk = ((u64)(c0 + lo) << L.rb) | (u64)(rowval[e] - 1);
        if constexpr (K32)
            static_cast<u32 *>(keys_out)[e] = (u32)(kk & lowmask);
        else
            static_cast<u64 *>(keys_out)[e] = (kk << ESP_TAG_BITS) | (u64)ESP_COO;
    }
}
// start of segment s in the stored CSC: the first entry of column s << clb (columns behind the last: nnz)
__global__ void csc_piece_starts_k(const i64 *__restrict__ colptr, i64 n, int clb, i64 S, i64 *__restrict__ pstart) {
    const i64 s = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (s > S) return;
    const i64 c = (s << clb) < n ? (s << clb) : n;
    pstart[s] = colptr[c] - 1;
}
int32_t flush_rebuild(esp_handle *h, const Sorted &st, int mode, i64 *Znew, bool *served) {
    *served = false;
    const i64 Z0 = h->nnz, T = st.total;
    if (mode != ESP_FLUSH_ROUTED || Z0 == 0 || T <= 0 || st.npieces != 0 || !st.seg_start || st.key_bytes != 8 || st.has_base) return ESP_OK;
    if (windowed(h) || h->shard_user || h->win_base != 0 || h->force_path != ESP_PATH_AUTO) return ESP_OK;
    const int clb = st.rem_bits - h->L.rb;
    if (clb < 0 || clb > esplocal::CL_MAX_BITS || st.S < 2 || ((i64)st.S << clb) < h->n) return ESP_OK;
    if (Z0 + T >= 0xFFFFFFF0ll) return ESP_OK;
    // (short columns only: the register tiers of the PIECES variants; longer runs would go through their radix tier)
    if ((double)Z0 > 10.0 * (double)h->n || (double)(Z0 + T) > 14.0 * (double)h->n) return ESP_OK;
    CK(fix_tail(h));
    const i64 S = st.S;
    CK(ensure(h, h->newkey, sizeof(u64) * (size_t)Z0));
    const size_t o_ps = 256 * 8;
    CK(ensure(h, h->piecetab, o_ps + sizeof(i64) * 2 * (size_t)(S + 1)));
    CK(ensure(h, h->misc, 256));
    char *TB = (char *)h->piecetab.p;
    i64 *pstart = (i64 *)(TB + o_ps);
    unsigned long long *d_maxlen = (unsigned long long *)h->misc.p + 24;
    const void *tab[4] = {h->newkey.p, st.sk, h->nzval.p, st.sv};
    const bool k32 = st.rem_bits <= 32;  // (the stored piece as 4-byte keys of kind COO: the bucket kernel's P32 piece format)
    {
        Span sp(h, ESP_ST_MERGE);
        if (k32)
            hipLaunchKernelGGL(csc_keys_k<true>, dim3(grid_for(h->n, 256)), dim3(256), 0, h->stream, (const i64 *)h->colptr.p, (const i64 *)h->rowval.p,
                               h->n, h->L, st.rem_bits, h->newkey.p);
        else
            hipLaunchKernelGGL(csc_keys_k<false>, dim3(grid_for(h->n, 256)), dim3(256), 0, h->stream, (const i64 *)h->colptr.p, (const i64 *)h->rowval.p,
                               h->n, h->L, st.rem_bits, h->newkey.p);
        hipLaunchKernelGGL(csc_piece_starts_k, dim3(grid_for(S + 1, 256)), dim3(256), 0, h->stream, (const i64 *)h->colptr.p, h->n, clb, S, pstart);
        sp.add(2);
    }
    HIPCK(h, hipMemcpyAsync(TB, tab, sizeof(tab), hipMemcpyHostToDevice, h->stream));
    HIPCK(h, hipMemcpyAsync(pstart + (S + 1), st.seg_start, sizeof(i64) * (size_t)(S + 1), hipMemcpyDeviceToDevice, h->stream));
    HIPCK(h, hipMemsetAsync(d_maxlen, 0, 16, h->stream));
    hipLaunchKernelGGL(piece_totals_k, dim3(grid_for(S, 256)), dim3(256), 0, h->stream, (const i64 *)pstart, 2, S, d_maxlen, d_maxlen + 1);
    HIPCK(h, hipMemcpyAsync(h->pin_scalar, d_maxlen, 16, hipMemcpyDeviceToHost, h->stream));
    HIPCK(h, hipStreamSynchronize(h->stream));  // (tab is read by the copy above)
    const i64 merged = (i64)h->pin_scalar[0];
    if (merged > (i64)esplocal::CAP || h->pin_scalar[1] != 0) return ESP_OK;  // (the caller's look-up + join path takes any segment)
    HIPCK(h, hipMemsetAsync((u32 *)h->misc.p + 60, 0, 4, h->stream));
    Sorted sp2 = st;
    sp2.seg_start = nullptr;
    sp2.npieces = 2;
    sp2.all_update = false;
    sp2.ptab = (const void *const *)TB;
    sp2.pstart = pstart;
    sp2.pieces_dense = true;
    sp2.maxlen = merged;
    sp2.total = -1;
    sp2.has_base = true;
    sp2.base = h->win_base;
    sp2.expect_hits = -1;
    if (k32) {
        sp2.p32_piece = 0;
        sp2.p32_lo = 0;
        sp2.kind = ESP_COO;
    }
    CK(ensure(h, h->keys2, sizeof(u64) * (size_t)(Z0 + T)));
    CK(ensure(h, h->vals2, sizeof(double) * (size_t)(Z0 + T)));
    // the bucket kernel writes colptr itself: into the second array (a failed flush leaves the stored matrix as it was)
    CK(ensure(h, h->colptr2, sizeof(i64) * (size_t)(h->n + 1)));
    std::swap(h->colptr, h->colptr2);
    h->nnz = 0;  // (a fresh matrix to everything below: the stored entries are pending entries of piece 0 now)
    i64 Ztot = 0;
    const int32_t rc = flush_local(h, sp2, mode, &Ztot);
    if (rc != ESP_OK) {
        std::swap(h->colptr, h->colptr2);
        h->nnz = Z0;
        h->tail_stale = false;
        h->ones_pending = false;
        return rc;
    }
    *Znew = Ztot - Z0;
    *served = true;
    return ESP_OK;
}

extern "C" int32_t esp_flush(esp_handle *h, int32_t mode, int64_t *new_nnz, int32_t *pattern_changed) {
    if (!h) return ESP_ERR_INVALID;
    if (mode != ESP_FLUSH_ROUTED && mode != ESP_FLUSH_PLUS) FAIL(h, ESP_ERR_INVALID, "esp_flush: mode");
    (void)hipSetDevice(h->device);
    if (pattern_changed) *pattern_changed = 0;
    i64 E = h->count;
    if (E == 0) {
        if (new_nnz) *new_nnz = h->nnz;
        return ESP_OK;
    }
    if (E >= 0xFFFFFFF0ll) FAIL(h, ESP_ERR_UNSUPPORTED, "esp_flush: %lld pending entries exceed the 2^32 limit of one flush", (long long)E);
    hipEvent_t fa = nullptr;
    if (h->timing) {
        fa = ev_get(h);
        (void)hipEventRecord(fa, h->stream);
    }
    i64 Zn = 0;
    h->last_rebuild = 0;
    bool use_local = h->force_path != ESP_PATH_GENERAL;
    if (h->ones_pending && windowed(h)) CK(fix_tail(h));  // (cannot happen: a window is declared through fix_tail)
    if (h->pre.valid) {  // the producer's partition serves this flush if nothing changed since (appends BEHIND it may have)
        const esp_handle::PrePart &pp = h->pre;
        const bool usable = use_local && !h->part_assembled && pp.mw_P == 0 && pp.E + pp.tail == E && pp.base == h->win_base &&
                            pp.span == h->win_span && pp.maxlen <= (i64)esplocal::CAP && pp.K - pp.pb <= esplocal::MAX_REM_BITS;
        if (!usable) CK(pending_materialize(h));
    }
    // a batch still held as sorted items goes to the fused bucket kernel only on a fresh matrix with nothing behind it
    // (... or over the pattern the same mesh built, when the handle's last flush over it hit: the re-assembly form)
    if (h->pre.valid && h->lazy.on &&
        (h->pre.tail != 0 || !use_local || windowed(h) || h->shard_user || h->part_assembled ||
         (h->nnz != 0 && !(h->seen_hits && !h->hits_off && mode == ESP_FLUSH_ROUTED))))
        CK(lazy_expand(h));
    bool served = false, split = false, tail_direct = false;
    i64 Zsplit = 0;  // new entries of the batch's own flush
    if (h->pre.valid && h->pre.tail > 0 && h->nnz > 0 && mode == ESP_FLUSH_ROUTED && h->force_path != ESP_PATH_BATCH_TAIL_ONE_FLUSH) {
        // Batch + tail over a stored pattern (a re-assembly whose mesh gained couplings): the batch by itself -- its buckets
        // fit the small variant of the bucket kernel, and a batch of hits emits nothing and needs no join -- then the tail
        // as a flush of its own.  A ROUTED flush may be cut at any stream position: flush! between two calls of an
        // ExtendableSparseMatrix never changes a result (extendable.jl:159-255: a call either hits the CSC or goes to the
        // buffer, which the flush adds as it is).  Not so csc + buffer (ESP_FLUSH_PLUS): the buffer is folded by itself
        // first.  force_path 22: one flush over two pieces, as on a fresh matrix.
        const esp_handle::PrePart pp = h->pre;
        Sorted st;
        st.sk = (const u64 *)h->keys.p;
        st.sv = (const double *)h->vals.p;
        st.in_primary = true;
        st.S = 1 << pp.pb;
        st.total = pp.E;
        st.seg_start = (const i64 *)h->seg[1].p;
        st.rem_bits = pp.K - pp.pb;
        st.local_ok = true;
        st.key_bytes = pp.key_bytes;
        st.kind = pp.kind;
        st.maxlen = pp.maxlen;
        st.expect_hits = 1;  // (the batch repeats the stream that built the pattern; what is new comes behind it)
        CK(ensure(h, h->keys2, sizeof(u64) * (size_t)E));
        CK(ensure(h, h->vals2, sizeof(double) * (size_t)E));
        CK(flush_local(h, st, mode, &Zsplit));  // (on failure everything is still pending)
        // the tail is the pending buffer now: packed keys BEHIND the batch's -- the partition reads them where they lie
        // (pend_off; settle_offset moves them to the front for the paths that expect them there)
        const i64 E0 = pp.E, T = pp.tail;
        const bool one_kind = h->kind_uniform >= 0 && h->kind_noted == h->count;
        const esp_handle::TailPart tp = h->tailpart;  // (the tail may have been partitioned as it was appended)
        h->count = T;
        h->pend_off = E0;
        if (h->force_path == ESP_PATH_TAIL_TO_FRONT) CK(settle_offset(h));
        h->pre.valid = false;
        h->kind_noted = one_kind ? T : 0;
        if (!one_kind) h->kind_uniform = -2;
        h->shard_valid = h->part_valid = false;
        h->values_version++;
        E = T;
        split = true;
        if (tp.valid && tp.T == T && h->pend_off == E0 && tp.base == h->win_base && tp.span == h->win_span && tp.maxlen <= seg_cap(h) &&
            tp.K - tp.pb <= esplocal::MAX_REM_BITS) {
            // ... then its flush starts at the bucket kernel, reading the packed keys where they lie
            Sorted st2;
            st2.sk = (const u64 *)h->keys.p + E0;
            st2.sv = (const double *)h->vals.p + E0;
            st2.in_primary = true;
            st2.S = 1 << tp.pb;
            st2.total = T;
            st2.seg_start = (const i64 *)h->tseg.p;
            st2.rem_bits = tp.K - tp.pb;
            st2.local_ok = true;
            st2.maxlen = tp.maxlen;
            st2.expect_hits = 0;
            // (the tail opens new positions: the rebuild -- stored entries as the first piece of a fresh flush -- where it applies)
            bool rebuilt = false;
            int32_t rc = flush_rebuild(h, st2, mode, &Zn, &rebuilt);
            h->last_rebuild = rebuilt ? 1 : 0;
            if (rc == ESP_OK && !rebuilt) rc = flush_local(h, st2, mode, &Zn);
            if (rc != ESP_OK) {
                (void)settle_offset(h);
                return rc;
            }
            served = true;
            tail_direct = true;
        }
        // (the tail gets a plan of its own: fewer, fuller segments -- with the batch's 2^16 buckets, small variant and 4-byte
        // keys included, its bucket kernel took 1.4 instead of 0.9 ms at config 3: time follows the number of segments)
    }
    if (h->pre.valid && h->pre.tail > 0) {
        // batch + tail: only the tail is partitioned, the bucket kernel reads every segment as two pieces
        CK(flush_pre_tail(h, mode, &Zn, &served));
        if (!served) CK(pending_materialize(h));  // (a merged segment is too long, or the tail is no pre-sorted stream)
    }
    if (served) {
        h->last_partition = tail_direct ? 8 : 5;
    } else if (h->pre.valid) {
        const esp_handle::PrePart &pp = h->pre;
        Sorted st;
        st.sk = (const u64 *)h->keys.p;
        st.sv = (const double *)h->vals.p;
        st.in_primary = true;
        st.S = 1 << pp.pb;
        st.total = pp.E;
        st.seg_start = (const i64 *)h->seg[1].p;
        st.rem_bits = pp.K - pp.pb;
        st.local_ok = true;
        st.key_bytes = pp.key_bytes;
        st.kind = pp.kind;
        st.maxlen = pp.maxlen;
        if (pp.fb > 0 && pp.key_bytes == 4 && !h->lazy.on) {
            // a FINE partition: the bucket kernel takes 2^fb neighbouring buckets as one segment (PrePart::fb)
            st.fb = pp.fb;
            st.S = 1 << (pp.pb - pp.fb);
            st.rem_bits = pp.K - pp.pb + pp.fb;
            st.maxlen = pp.maxlen_c;
        }
        CK(ensure(h, h->keys2, sizeof(u64) * (size_t)E));
        CK(ensure(h, h->vals2, sizeof(double) * (size_t)E));
        int32_t rc_local;
        if (h->lazy.on) {
            st.sk = h->lazy.src == 1 ? h->lazy.it.sorted_keys : h->lazy.el.sorted_keys;
            st.sv = nullptr;
            st.lazy = &h->lazy;
            rc_local = flush_local(h, st, mode, &Zn);
            if (rc_local == ESP_RETRY_EXPANDED) {  // (not the fused kernel's flush after all: the entries, then the usual way)
                CK(lazy_expand(h));
                st.sk = (const u64 *)h->keys.p;
                st.sv = (const double *)h->vals.p;
                st.lazy = nullptr;
                CK(ensure(h, h->keys2, sizeof(u64) * (size_t)E));
                CK(ensure(h, h->vals2, sizeof(double) * (size_t)E));
                rc_local = flush_local(h, st, mode, &Zn);
            }
            // (a failed flush leaves the batch pending as it was: still items, or expanded by the retry)
        } else {
            rc_local = flush_local(h, st, mode, &Zn);
        }
        CK(rc_local);
        h->last_partition = 4;
        h->runs_penalty = 0;
        h->seen_spread = pp.Ee > 0.0 ? (double)pp.maxlen * std::ldexp(1.0, pp.pb) / pp.Ee : 0.0;
    } else if (h->part_assembled) {
        // partitioned shard exchange: the segments are already formed (esp_shard_assemble)
        CK(ensure(h, h->misc, 256));
        HIPCK(h, hipMemsetAsync((u32 *)h->misc.p + 60, 0, 4, h->stream));
        Sorted st;
        char *T = (char *)h->piecetab.p;
        st.sk = (const u64 *)h->keys.p;
        st.sv = (const double *)h->vals.p;
        st.in_primary = true;
        st.S = (int)h->part_nb;
        st.seg_start = nullptr;
        st.rem_bits = h->part_shift;
        st.local_ok = true;
        st.npieces = h->part_P;
        st.all_update = h->part_all_update;
        st.ptab = (const void *const *)T;
        st.pstart = (const i64 *)(T + 256 * 8);
        st.maxlen = h->part_maxlen;
        if (h->part_own32) {
            st.p32_piece = h->part_me;
            st.p32_lo = h->part_own_lo;
            st.kind = h->part_kind32;
            if (h->part_fb > 0) {  // (the own range lies bucket by bucket of the producer's FINE partition)
                st.fb = h->part_fb;
                st.own_fine = (const i64 *)h->seg[1].p + (size_t)h->part_me * ((size_t)h->part_nb << h->part_fb);
            }
        }
        CK(ensure(h, h->keys2, sizeof(u64) * (size_t)std::max<i64>(h->part_total, 1)));
        CK(ensure(h, h->vals2, sizeof(double) * (size_t)std::max<i64>(h->part_total, 1)));
        CK(flush_local(h, st, mode, &Zn));
        h->last_partition = 7;
    } else if (use_local) {
        Sorted st;
        const i64 off0 = h->pend_off;
        {
            const int32_t rc = sort_msd(h, &st);
            if (rc != ESP_OK) {
                (void)settle_offset(h);
                return rc;
            }
        }
        if (!st.in_primary) {  // keep "pending data lives in keys/vals" true for the general path
            std::swap(h->keys, h->keys2);
            std::swap(h->vals, h->vals2);
            h->cap = (i64)std::min(h->keys.bytes / sizeof(u64), h->vals.bytes / sizeof(double));
            st.in_primary = true;
            h->pend_off = 0;  // (the partition wrote from the front of the other pair)
        }
        if (st.local_ok) {
            if (split) st.expect_hits = 0;
            int32_t rc;
            if (h->debug_fail_bucket) {  // (esp_debug_fail_next_bucket_stage)
                h->debug_fail_bucket = false;
                rc = [&]() -> int32_t { FAIL(h, ESP_ERR_STATE, "esp_flush: the bucket stage was told to fail (esp_debug_fail_next_bucket_stage)"); }();
            } else {
                rc = flush_local(h, st, mode, &Zn);
            }
            if (rc != ESP_OK && st.key_bytes == 4 && st.k32_passes >= 2) {
                // (two or more passes moved 4-byte keys: the scratch pair holds the input of the last one, 4-byte keys as well --
                // the packed keys are rebuilt there from the partitioned ones)
                hipLaunchKernelGGL(esprun::expand_keys_k, dim3(esprun::expand_keys_grid(st.S)), dim3(esprun::THREADS), 0, h->stream, (const u32 *)st.sk,
                                   st.seg_start, st.rem_bits, h->win_base, (u32)st.kind, (u64 *)h->keys2.p, (i64)st.S);
                (void)hipMemcpyAsync(h->vals2.p, st.sv, sizeof(double) * (size_t)st.total, hipMemcpyDeviceToDevice, h->stream);
                (void)hipStreamSynchronize(h->stream);
            }
            if (rc != ESP_OK && st.key_bytes == 4) {
                // the batch stays pending: its packed keys are intact in the scratch pair (the partition wrote the 4-byte
                // keys into the other one)
                std::swap(h->keys, h->keys2);
                std::swap(h->vals, h->vals2);
                h->cap = (i64)std::min(h->keys.bytes / sizeof(u64), h->vals.bytes / sizeof(double));
                h->pend_off = off0;
            }
            if (rc != ESP_OK) {
                (void)settle_offset(h);
                return rc;
            }
        } else {
            use_local = false;
        }
    }
    if (!use_local && !h->part_assembled) CK(settle_offset(h));
    if (!use_local && !h->part_assembled) CK(flush_global(h, mode, &Zn));
    if (split) h->last_partition = tail_direct ? 8 : 6;
    h->last_path = (use_local || h->part_assembled) ? 1 : 2;
    if ((Zn > 0 || Zsplit > 0) && pattern_changed) *pattern_changed = 1;
    h->values_version++;  // (hits were applied in place)
    HIPCK(h, hipGetLastError());
    h->count = 0;
    h->pend_off = 0;
    pending_changed(h);
    if (h->lazy_hold.p) {  // (the uploaded element matrices of esp_append_elements_host: their batch is flushed)
        HIPCK(h, hipStreamSynchronize(h->stream));
        release(h->lazy_hold);
    }
    if (h->timing && fa) {
        hipEvent_t fb = ev_get(h);
        (void)hipEventRecord(fb, h->stream);
        h->spans.push_back({-1, fa, fb, 0});
    }
    if (new_nnz) *new_nnz = h->nnz;
    return ESP_OK;
}

