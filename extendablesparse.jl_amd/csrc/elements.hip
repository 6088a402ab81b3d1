// elements.hip -- libesparse_hip: element-level append (esp_append_elements*, esp_generate_fem_mesh); kernels in elements.hpp
#include "internal.hpp"
#include "elements.hpp"

namespace {

// The item partition (elements.hpp) on an empty buffer.  *took = false: not applicable (or a cell names a node twice):
// nothing was appended, the caller writes the updates in stream order.
int32_t elements_by_items(esp_handle *h, espelem::Args a, i64 E, bool *took) {
    *took = false;
    if (h->count != 0 || E <= esplocal::CAP || windowed(h) || h->shard_user) return ESP_OK;
    // (test hooks that pin another path: 2 general, 5 / 12 / 16 partition flavours, 19 plain pending buffer, 25 the item partition off)
    if (h->force_path == ESP_PATH_GENERAL || h->force_path == ESP_PATH_NO_RUN_PARTITION || h->force_path == ESP_PATH_RUN_LIST_BY_RADIX ||
        h->force_path == ESP_PATH_PRODUCER_STREAM_ORDER || h->force_path == ESP_PATH_NO_BATCH_TAIL || h->force_path == ESP_PATH_NO_ITEM_PARTITION)
        return ESP_OK;
    const int W = a.W;
    const i64 NI = a.nitems;
    if (NI >= 0xFFFFFFF0ll || 2 * NI > E || (i64)esplocal::CAP / W < 64) return ESP_OK;
    // the records' virtual layout: the item's number below the column bits
    const int vrb = std::max(h->L.rb, bits_for(NI) - ESP_TAG_BITS);
    if (vrb + h->L.cb + ESP_TAG_BITS > 64) return ESP_OK;
    CK(reserve_append(h, E));
    // two ping-pong arrays of item records inside the flush's scratch array (sized for E updates: E / W items each)
    CK(ensure(h, h->keys2, sizeof(u64) * (size_t)E));
    CK(ensure(h, h->vals2, sizeof(double) * (size_t)E));
    CK(ensure(h, h->misc, 256));
    unsigned long long *d_err = (unsigned long long *)h->misc.p;
    u32 *d_dup = (u32 *)((unsigned long long *)h->misc.p + 1);
    a.vrb = vrb;
    a.err = d_err;
    a.dup = d_dup;
    a.ikeys = (u64 *)h->keys2.p;
    // cells of 3 or 4 nodes: a 64-byte record per cell (rows as u32, diag values) in the value half of the scratch pair,
    // which the keys-only passes never touch (force_path 37: test hook, never -- the expansion gathers from the caller's arrays)
    const bool cellrec = (a.nloc == 3 || a.nloc == 4) && h->m <= ((i64)1 << 32) && (size_t)a.ncells * 64 <= h->vals2.bytes &&
                         h->force_path != ESP_PATH_NO_CELL_RECORDS;
    a.cellrec = cellrec ? (char *)h->vals2.p : nullptr;
    // the batch stays a list of sorted items and the flush's bucket kernel forms the updates (group3_items.hpp): item records
    // and cell records then live in the pending buffer's own arrays (which the updates would fill), the scratch pair stays
    // free for the flush's output
    const bool lazy = cellrec && lazy_items_wanted(h, a.kind) && h->keys.bytes >= 2 * sizeof(u64) * (size_t)NI &&
                      h->vals.bytes >= (size_t)a.ncells * 64;
    if (lazy) {
        a.ikeys = (u64 *)h->keys.p;
        a.cellrec = (char *)h->vals.p;
    }
    const int Kv = bits_for(std::max<i64>(h->n, 1)) + vrb;  // bits of the records' virtual key window (n << vrb keys)
    const u64 vspan = (u64)std::max<i64>(h->n, 1) << vrb;
    // with cell records the passes may stop a few bits early: the expansion orders every segment by the last bits itself
    // (segexpand.hpp); a second attempt with the passes alone when a segment does not fit that
    int sort_bits = 0;
    int lbits0 = 0;
    if (cellrec && !lazy) {
        const u64 span0 = h->win_span, base0 = h->win_base;
        h->win_base = 0, h->win_span = vspan;
        int Kw = 1;
        while (Kw < 62 && ((u64)1 << Kw) < vspan) Kw++;
        lbits0 = plan_local_bits(h, NI, W, Kw, &sort_bits);
        h->win_base = base0, h->win_span = span0;
    }
    (void)Kv;
    Sorted st;
    int lbits = 0;
    i64 maxlen_updates = 0;
    int rem_real = 0;
    bool k32 = false, done = false;
    for (int attempt = lbits0 > 0 ? 0 : 1; attempt < 2 && !done; attempt++) {
        lbits = attempt == 0 ? lbits0 : 0;
        h->pin_scalar[0] = ~0ull;
        h->pin_scalar[1] = 0ull;
        h->pin_scalar[2] = 0ull, h->pin_scalar[3] = 0ull;
        h->pin_scalar[4] = 0x00000000FFFFFFFFull;  // (colrange: smallest node in the low word, largest in the high one)
        h->pin_scalar[5] = 0ull;                   // (... and the count of wide waves)
        HIPCK(h, hipMemcpyAsync(d_err, h->pin_scalar, 48, hipMemcpyHostToDevice, h->stream));
        a.colrange = (cellrec && h->n < ((i64)1 << 32)) ? (u32 *)((unsigned long long *)h->misc.p + 4) : nullptr;
        {
            Span sp(h, ESP_ST_APPEND);
            if (cellrec && a.nloc == 3)
                hipLaunchKernelGGL(espelem::elem_cells_k<3>, dim3(grid_for(a.ncells, espelem::THREADS)), dim3(espelem::THREADS), 0, h->stream, a);
            else if (cellrec)
                hipLaunchKernelGGL(espelem::elem_cells_k<4>, dim3(grid_for(a.ncells, espelem::THREADS)), dim3(espelem::THREADS), 0, h->stream, a);
            else
                hipLaunchKernelGGL(espelem::elem_items_k, dim3(grid_for(NI, espelem::THREADS)), dim3(espelem::THREADS), 0, h->stream, a);
            sp.add(1);
        }
        // the flush's partition over the items: a temporary view of the handle (sort_msd reads count, keys, keys2 and the
        // key window -- here the records' virtual one)
        const DevBuf k0 = h->keys, v0 = h->vals, k2 = h->keys2, v2 = h->vals2;
        const i64 count0 = h->count;
        const double spread0 = h->seen_spread;
        const u64 span0 = h->win_span, base0 = h->win_base;
        h->keys.p = a.ikeys, h->keys.bytes = sizeof(u64) * (size_t)NI;
        h->keys2.p = a.ikeys + NI, h->keys2.bytes = sizeof(u64) * (size_t)NI;
        h->vals.bytes = std::max(h->vals.bytes, sizeof(double) * (size_t)NI);  // (keys-only passes never touch the value arrays)
        h->vals2.bytes = std::max(h->vals2.bytes, sizeof(double) * (size_t)NI);
        h->count = NI;
        h->win_base = 0;
        h->win_span = vspan;
        h->plan_cap = lbits ? (i64)espseg::LCAP : (i64)esplocal::CAP / W;
        h->plan_bits = lbits ? sort_bits : 0;
        h->plan_occ_span = 0;
        h->plan_try_runs = false;
        if (a.colrange && !lbits) {
            // the columns the batch touches (one round trip: it saves the pass -- a dozen launches -- a plan made for the whole window
            // needs when the batch is a band of the mesh: one partition's cells, 1 / p of the columns at p times the density)
            HIPCK(h, hipMemcpyAsync(h->pin_scalar, (unsigned long long *)h->misc.p + 4, 16, hipMemcpyDeviceToHost, h->stream));
            HIPCK(h, hipStreamSynchronize(h->stream));
            const u64 lo = h->pin_scalar[0] & 0xFFFFFFFFull, hi = h->pin_scalar[0] >> 32;
            if (lo <= hi) h->plan_occ_span = (hi - lo + 1) << vrb;
            // (the run-based single pass only for a cell order that looks like a mesh's own: fewer than an eighth of the waves wide)
            const u64 blocks = (u64)grid_for(a.ncells, espelem::THREADS);
            const u64 wide = h->pin_scalar[1] & 0xFFFFFFFFull, sampled = (blocks + 63) / 64 + 1;  // (elem_cells_k: one wave of every 64th workgroup + the last)
            h->plan_try_runs = wide * 8 <= sampled;
        }
        h->item_mode = true;
        h->item_keys_only = true;
        st = Sorted();
        const int32_t rc = sort_msd(h, &st);
        h->keys = k0, h->vals = v0, h->keys2 = k2, h->vals2 = v2;
        h->count = count0;
        h->win_base = base0, h->win_span = span0;
        h->plan_cap = 0;
        h->plan_bits = 0;
        h->plan_occ_span = 0;
        h->plan_try_runs = false;
        h->item_mode = false;
        h->item_keys_only = false;
        if (rc != ESP_OK) return rc;
        HIPCK(h, hipMemcpyAsync(h->pin_scalar, d_err, 16, hipMemcpyDeviceToHost, h->stream));
        HIPCK(h, hipStreamSynchronize(h->stream));
        if (h->pin_scalar[0] != ~0ull)
            FAIL(h, ESP_ERR_BOUNDS, "BoundsError: cell %llu of the batch names a node outside 1..%lld", (unsigned long long)h->pin_scalar[0],
                 (long long)a.lim);
        if ((u32)h->pin_scalar[1] != 0u) {  // a cell with a repeated node: its items would interleave in call order
            h->seen_spread = spread0;
            return ESP_OK;
        }
        if (lbits) lbits = std::min(lbits, st.rem_bits - vrb);  // (the sub-segments are whole columns)
        rem_real = st.rem_bits - lbits - (vrb - h->L.rb);
        const bool usable = st.fits && st.S >= 2 && st.rem_bits - std::max(lbits, 0) >= vrb && rem_real <= esplocal::MAX_REM_BITS &&
                            (attempt == 0 ? lbits >= 1 : st.maxlen * W <= (i64)esplocal::CAP);
        if (!usable) {
            h->seen_spread = spread0;
            if (attempt == 1) return ESP_OK;  // (no segment table the bucket kernel takes)
            continue;
        }
        k32 = rem_real <= 32 && h->force_path != ESP_PATH_PACKED_KEYS;
        a.sorted_keys = st.sk;
        a.rem_bits = rem_real;
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
            sa.lshift = st.rem_bits - lbits;  // (in the records' virtual layout)
            sa.base = 0;
            sa.sub_start = (i64 *)h->seg[1].p;
            sa.maxsub = d_maxsub;
            sa.total_items = NI;
            {
                Span sp(h, ESP_ST_APPEND);
                const dim3 grid((unsigned)st.S), block(espseg::THREADS);
                if (a.nloc == 3 && k32)
                    hipLaunchKernelGGL((espelem::elem_seg_expand_k<true, 3>), grid, block, 0, h->stream, a, sa);
                else if (a.nloc == 3)
                    hipLaunchKernelGGL((espelem::elem_seg_expand_k<false, 3>), grid, block, 0, h->stream, a, sa);
                else if (k32)
                    hipLaunchKernelGGL((espelem::elem_seg_expand_k<true, 4>), grid, block, 0, h->stream, a, sa);
                else
                    hipLaunchKernelGGL((espelem::elem_seg_expand_k<false, 4>), grid, block, 0, h->stream, a, sa);
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
            if (lazy) {  // (the expansion is the flush's business now -- or lazy_expand's)
                h->lazy.src = 2;
                h->lazy.k32 = k32;
                h->lazy.el = a;
            } else if (k32)
                espelem::launch_expand<true>(a, h->stream);
            else
                espelem::launch_expand<false>(a, h->stream);
            hipLaunchKernelGGL(espitem::scale_segments_k, dim3(grid_for((i64)st.S + 1, 256)), dim3(256), 0, h->stream, st.seg_start, (i64)st.S + 1,
                               (i64)W, (i64 *)h->seg[1].p);
            sp.add(2);
            maxlen_updates = st.maxlen * W;
        }
        done = true;
    }
    if (!done) return ESP_OK;
    const int K = window_bits(h);
    HIPCK(h, hipGetLastError());
    esp_handle::PrePart &pp = h->pre;
    pp.K = K;
    pp.pb = K - rem_real;
    pp.maxlen = maxlen_updates;
    pp.key_bytes = k32 ? 4 : 8;
    pp.kind = a.kind;
    pp.E = E;
    pp.tail = 0;
    pp.base = h->win_base;
    pp.span = h->win_span;
    pp.Ee = plan_entries(E, K, h->win_span);
    pp.mw_P = 0, pp.mw_me = 0, pp.mw_shift = 0, pp.mw_nb = 0, pp.mw_eps = 0;
    pp.own32 = false;
    pp.fb = 0, pp.maxlen_c = 0;
    // esp_elements_keep_plan: item order, cell records and segment table into buffers of their own (the flush is about to
    // take the scratch pair they lie in) for esp_append_elements_again
    esp_handle::ElemPlan &ep = h->elemplan;
    ep.valid = false;
    if (ep.keep && cellrec && lbits == 0) {
        const i64 S_final = (i64)st.S;
        CK(ensure(h, ep.sorted, sizeof(u64) * (size_t)NI));
        CK(ensure(h, ep.cellrec, (size_t)a.ncells * 64));
        CK(ensure(h, ep.segtab, sizeof(i64) * (size_t)(S_final + 1)));
        Span sp(h, ESP_ST_COPY);
        HIPCK(h, hipMemcpyAsync(ep.sorted.p, st.sk, sizeof(u64) * (size_t)NI, hipMemcpyDeviceToDevice, h->stream));
        HIPCK(h, hipMemcpyAsync(ep.cellrec.p, a.cellrec, (size_t)a.ncells * 64, hipMemcpyDeviceToDevice, h->stream));
        HIPCK(h, hipMemcpyAsync(ep.segtab.p, h->seg[1].p, sizeof(i64) * (size_t)(S_final + 1), hipMemcpyDeviceToDevice, h->stream));
        sp.add(3);
        ep.nloc = a.nloc, ep.W = W, ep.vrb = vrb, ep.rem_real = rem_real, ep.K = K, ep.kind = a.kind;
        ep.k32 = k32, ep.has_diag = a.diag != nullptr;
        ep.ncells = a.ncells, ep.S = S_final, ep.maxlen = maxlen_updates;
        ep.base = h->win_base, ep.span = h->win_span;
        ep.valid = true;
    }
    h->lazy.armed = lazy;  // (-> lazy.on beside pre.valid, once the caller has counted the entries in)
    *took = true;  // (the caller sets pre.valid once the entries are counted in)
    return ESP_OK;
}

}  // namespace

// A handle that keeps the plan of its element-level appends (item order, cell records, segment table: 8 B per item + 64 B
// per cell of device memory) for esp_append_elements_again.  on = 0: off, the buffers are released.
extern "C" int32_t esp_elements_keep_plan(esp_handle *h, int32_t on) {
    if (!h) return ESP_ERR_INVALID;
    h->elemplan.keep = on != 0;
    if (!on) {
        (void)hipSetDevice(h->device);
        // (a pending batch of esp_append_elements_again that stayed a list of items reads the plan's item order and cell records
        // at flush time: its updates are formed now, before the plan's buffers go)
        if (h->lazy.on && h->lazy.src == 2 &&
            (h->lazy.el.sorted_keys == (const u64 *)h->elemplan.sorted.p || h->lazy.el.cellrec == (char *)h->elemplan.cellrec.p))
            CK(lazy_expand(h));
        HIPCK(h, hipStreamSynchronize(h->stream));
        h->elemplan.valid = false;
        release(h->elemplan.sorted);
        release(h->elemplan.cellrec);
        release(h->elemplan.segtab);
    }
    return ESP_OK;
}

// The element loop again over the SAME connectivity as the handle's last planned esp_append_elements (a time step of an
// instationary / nonlinear code: the mesh stays, the element matrices change): no pass over the connectivity, no item
// partition -- the diag values of the cell records are refreshed and the expansion runs over the kept item order.  The
// entries are, bit for bit, what esp_append_elements(h, nloc, ncells, the same cellnodes, d_elmat, d_diag, kind, op) appends.
extern "C" int32_t esp_append_elements_again(esp_handle *h, const double *d_elmat, const double *d_diag, int32_t kind, int32_t op) {
    if (!h || !d_elmat) return ESP_ERR_INVALID;
    if (kind < 0 || kind > 3) FAIL(h, ESP_ERR_INVALID, "append: kind %d invalid", kind);
    if (op != ESP_OP_ADD && op != ESP_OP_SUB) FAIL(h, ESP_ERR_UNSUPPORTED, "append: op %d not supported on the device path", op);
    const esp_handle::ElemPlan &ep = h->elemplan;
    if (!ep.valid) FAIL(h, ESP_ERR_STATE, "esp_append_elements_again: the handle keeps no plan (esp_elements_keep_plan, then esp_append_elements on an empty buffer)");
    if ((d_diag != nullptr) != ep.has_diag) FAIL(h, ESP_ERR_INVALID, "esp_append_elements_again: the planned call %s a diagonal term", ep.has_diag ? "had" : "had no");
    if (h->count != 0 || windowed(h) || h->shard_user || ep.base != h->win_base || ep.span != h->win_span)
        FAIL(h, ESP_ERR_STATE, "esp_append_elements_again: the buffer must be empty (flush first) and the key window the planned call's");
    (void)hipSetDevice(h->device);
    espelem::Args a;
    memset(&a, 0, sizeof a);
    a.nloc = ep.nloc;
    a.W = ep.W;
    a.ncells = ep.ncells;
    a.nitems = ep.ncells * ep.nloc;
    a.elmat = d_elmat;
    a.diag = d_diag;
    a.lim = std::min(h->m, h->n);
    a.L = h->L;
    a.vrb = ep.vrb;
    a.kind = kind;
    a.negate = (op == ESP_OP_SUB && kind != ESP_SET) ? 1 : 0;
    a.cellrec = (char *)ep.cellrec.p;
    a.sorted_keys = (const u64 *)ep.sorted.p;
    a.rem_bits = ep.rem_real;
    a.base = h->win_base;
    const i64 E = a.nitems * a.W;
    CK(reserve_append(h, E));
    a.keys_out = (u64 *)h->keys.p;
    a.vals_out = (double *)h->vals.p;
    CK(ensure(h, h->seg[1], sizeof(i64) * (size_t)(ep.S + 1)));
    bool lazy = false;
    {
        Span sp(h, ESP_ST_APPEND);
        if (d_diag) {
            const dim3 grid(grid_for(ep.ncells, espelem::THREADS)), block(espelem::THREADS);
            if (ep.nloc == 3)
                hipLaunchKernelGGL(espelem::elem_refresh_diag_k<3>, grid, block, 0, h->stream, d_diag, ep.ncells, a.cellrec);
            else
                hipLaunchKernelGGL(espelem::elem_refresh_diag_k<4>, grid, block, 0, h->stream, d_diag, ep.ncells, a.cellrec);
        }
        // (the batch may stay a list of items -- the kept order IS the sorted item list: the fused bucket kernel's re-assembly
        // form forms the updates at flush time, lazy_expand for everybody else)
        lazy = lazy_items_wanted(h, kind);
        if (lazy) {
            h->lazy.src = 2;
            h->lazy.k32 = ep.k32;
            h->lazy.el = a;
        } else if (ep.k32)
            espelem::launch_expand<true>(a, h->stream);
        else
            espelem::launch_expand<false>(a, h->stream);
        sp.add(2);
    }
    HIPCK(h, hipMemcpyAsync(h->seg[1].p, ep.segtab.p, sizeof(i64) * (size_t)(ep.S + 1), hipMemcpyDeviceToDevice, h->stream));
    HIPCK(h, hipGetLastError());
    h->rawplan.valid = false;  // (seg[1] is rewritten)
    h->genplan.valid = false;
    esp_handle::PrePart &pp = h->pre;
    pp.K = ep.K;
    pp.pb = ep.K - ep.rem_real;
    pp.maxlen = ep.maxlen;
    pp.key_bytes = ep.k32 ? 4 : 8;
    pp.kind = kind;
    pp.E = E;
    pp.tail = 0;
    pp.base = h->win_base;
    pp.span = h->win_span;
    pp.Ee = plan_entries(E, ep.K, h->win_span);
    pp.mw_P = 0, pp.mw_me = 0, pp.mw_shift = 0, pp.mw_nb = 0, pp.mw_eps = 0;
    pp.own32 = false;
    pp.fb = 0, pp.maxlen_c = 0;
    note_kind(h, kind, E);
    h->count += E;
    pending_changed(h);
    h->pre.valid = true;
    h->lazy.on = lazy;
    return ESP_OK;
}

extern "C" int32_t esp_append_elements(esp_handle *h, int32_t nloc, int64_t ncells, const int64_t *d_cellnodes, const double *d_elmat,
                                       const double *d_diag, int32_t kind, int32_t op) {
    if (!h) return ESP_ERR_INVALID;
    if (nloc < 1 || nloc > espelem::MAX_NLOC) FAIL(h, ESP_ERR_UNSUPPORTED, "esp_append_elements: %d nodes per cell (1..%d)", nloc, espelem::MAX_NLOC);
    if (ncells < 0 || (ncells > 0 && (!d_cellnodes || !d_elmat))) FAIL(h, ESP_ERR_INVALID, "esp_append_elements: bad arguments");
    if (kind < 0 || kind > 3) FAIL(h, ESP_ERR_INVALID, "append: kind %d invalid", kind);
    if (op != ESP_OP_ADD && op != ESP_OP_SUB) FAIL(h, ESP_ERR_UNSUPPORTED, "append: op %d not supported on the device path", op);
    if (ncells == 0) return ESP_OK;
    // any new element-level append supersedes the kept plan: only a call that takes the item partition below makes one again
    // (a call that leaves early -- non-empty buffer, column window, a repeated node, ESP_ERR_BOUNDS -- must not leave the plan of
    // an EARLIER mesh behind for esp_append_elements_again)
    h->elemplan.valid = false;
    (void)hipSetDevice(h->device);
    espelem::Args a;
    memset(&a, 0, sizeof a);
    a.nloc = nloc;
    a.W = nloc + (d_diag ? 1 : 0);
    a.ncells = ncells;
    if (ncells > (((i64)1 << 40) / nloc)) FAIL(h, ESP_ERR_UNSUPPORTED, "esp_append_elements: too many cells for one call");
    a.nitems = ncells * nloc;
    a.cellnodes = d_cellnodes;
    a.elmat = d_elmat;
    a.diag = d_diag;
    a.lim = std::min(h->m, h->n);
    a.L = h->L;
    a.kind = kind;
    a.negate = (op == ESP_OP_SUB && kind != ESP_SET) ? 1 : 0;
    const i64 E = a.nitems * a.W;
    bool took = false;
    CK(elements_by_items(h, a, E, &took));
    if (!took) {
        // stream order: packed keys behind whatever is pending
        CK(reserve_append(h, E));
        CK(ensure(h, h->misc, 256));
        unsigned long long *d_err = (unsigned long long *)h->misc.p;
        h->pin_scalar[0] = ~0ull;
        HIPCK(h, hipMemcpyAsync(d_err, h->pin_scalar, 8, hipMemcpyHostToDevice, h->stream));
        a.err = d_err;
        a.keys_out = (u64 *)h->keys.p + h->count;
        a.vals_out = (double *)h->vals.p + h->count;
        {
            Span sp(h, ESP_ST_APPEND);
            const i64 per_cell = (i64)a.nloc * a.W;
            const i64 cpl = std::max<i64>(1, ((i64)1 << 30) / per_cell);  // (cells per launch: the grid stays below 2^32 threads)
            for (i64 c0 = 0; c0 < ncells; c0 += cpl) {
                espelem::Args b = a;
                const i64 cnt = std::min(cpl, ncells - c0) * per_cell;
                b.cell_base = c0;
                b.cellnodes = a.cellnodes + c0 * a.nloc;
                b.elmat = a.elmat + c0 * a.nloc * a.nloc;
                b.diag = a.diag ? a.diag + c0 * a.nloc : nullptr;
                b.keys_out = a.keys_out + c0 * per_cell;
                b.vals_out = a.vals_out + c0 * per_cell;
                hipLaunchKernelGGL(espelem::elem_stream_k, dim3(grid_for(cnt, espelem::THREADS)), dim3(espelem::THREADS), 0, h->stream, b, cnt);
                sp.add(1);
            }
        }
        HIPCK(h, hipMemcpyAsync(h->pin_scalar, d_err, 8, hipMemcpyDeviceToHost, h->stream));
        HIPCK(h, hipStreamSynchronize(h->stream));
        HIPCK(h, hipGetLastError());
        if (h->pin_scalar[0] != ~0ull) {
            h->pre_keep = false;
            FAIL(h, ESP_ERR_BOUNDS, "BoundsError: cell %llu of the batch names a node outside 1..%lld",
                 (unsigned long long)h->pin_scalar[0], (long long)a.lim);
        }
    }
    note_kind(h, kind, E);
    h->count += E;
    pending_changed(h);
    if (took) h->pre.valid = true, h->lazy.on = h->lazy.armed;
    h->lazy.armed = false;
    return ESP_OK;
}

// host arrays: uploaded as they are (three plain copies), then the device form
extern "C" int32_t esp_append_elements_host(esp_handle *h, int32_t nloc, int64_t ncells, const int64_t *cellnodes, const double *elmat,
                                            const double *diag, int32_t kind, int32_t op) {
    if (!h) return ESP_ERR_INVALID;
    if (nloc < 1 || nloc > espelem::MAX_NLOC) FAIL(h, ESP_ERR_UNSUPPORTED, "esp_append_elements: %d nodes per cell (1..%d)", nloc, espelem::MAX_NLOC);
    if (ncells < 0 || (ncells > 0 && (!cellnodes || !elmat))) FAIL(h, ESP_ERR_INVALID, "esp_append_elements: bad arguments");
    if (ncells == 0) return ESP_OK;
    (void)hipSetDevice(h->device);
    if (h->lazy_hold.p) {  // (the matrices of an earlier batch: whatever still gathers from them runs now -- an append behind a batch of items expands it)
        CK(lazy_expand(h));
        HIPCK(h, hipStreamSynchronize(h->stream));
        release(h->lazy_hold);
    }
    DevBuf dn, de, dd;
    const size_t bn = sizeof(i64) * (size_t)ncells * (size_t)nloc, be = sizeof(double) * (size_t)ncells * (size_t)nloc * (size_t)nloc;
    int32_t rc = ensure(h, dn, bn);
    if (rc == ESP_OK) rc = ensure(h, de, be);
    if (rc == ESP_OK && diag) rc = ensure(h, dd, bn);
    if (rc == ESP_OK) {
        Span sp(h, ESP_ST_COPY);  // (pageable arrays: through the pinned bounce buffers, handle.hip)
        rc = h2d_pipelined(h, dn.p, cellnodes, bn);
        if (rc == ESP_OK) rc = h2d_pipelined(h, de.p, elmat, be);
        if (rc == ESP_OK && diag) rc = h2d_pipelined(h, dd.p, diag, bn);
        sp.add(diag ? 3 : 2);
    }
    if (rc == ESP_OK)
        rc = esp_append_elements(h, nloc, ncells, (const i64 *)dn.p, (const double *)de.p, diag ? (const double *)dd.p : nullptr, kind, op);
    (void)hipStreamSynchronize(h->stream);  // (the kernels have read the temporaries)
    release(dn);
    if (rc == ESP_OK && h->lazy.on) {
        // the batch stayed a list of items: the fused bucket kernel (or lazy_expand) gathers the values from the uploaded
        // element matrices at flush time -- they live on in the handle (rows and diagonal terms are in the cell records)
        h->lazy_hold = de;
        de = DevBuf();
    }
    release(de);
    release(dd);
    return rc;
}

// host arrays: uploaded as they are, then the device form
extern "C" int32_t esp_append_elements_again_host(esp_handle *h, const double *elmat, const double *diag, int32_t kind, int32_t op) {
    if (!h || !elmat) return ESP_ERR_INVALID;
    const esp_handle::ElemPlan &ep = h->elemplan;
    if (!ep.valid) FAIL(h, ESP_ERR_STATE, "esp_append_elements_again: the handle keeps no plan (esp_elements_keep_plan, then esp_append_elements on an empty buffer)");
    (void)hipSetDevice(h->device);
    if (h->lazy_hold.p) {  // (the matrices of an earlier batch: see esp_append_elements_host)
        CK(lazy_expand(h));
        HIPCK(h, hipStreamSynchronize(h->stream));
        release(h->lazy_hold);
    }
    DevBuf de, dd;
    const size_t bn = sizeof(double) * (size_t)ep.ncells * (size_t)ep.nloc, be = bn * (size_t)ep.nloc;
    int32_t rc = ensure(h, de, be);
    if (rc == ESP_OK && diag) rc = ensure(h, dd, bn);
    if (rc == ESP_OK) {
        Span sp(h, ESP_ST_COPY);
        rc = h2d_pipelined(h, de.p, elmat, be);
        if (rc == ESP_OK && diag) rc = h2d_pipelined(h, dd.p, diag, bn);
        sp.add(diag ? 2 : 1);
    }
    if (rc == ESP_OK) rc = esp_append_elements_again(h, (const double *)de.p, diag ? (const double *)dd.p : nullptr, kind, op);
    (void)hipStreamSynchronize(h->stream);  // (the kernels have read the temporaries)
    if (rc == ESP_OK && h->lazy.on) {  // (the batch stayed a list of items: the element matrices live on until its flush)
        h->lazy_hold = de;
        de = DevBuf();
    }
    release(de);
    release(dd);
    return rc;
}

// The element data of the build's Kuhn grid as device arrays (the producer of cellnodes / elmat / diag for tests and bench)
extern "C" int32_t esp_generate_fem_mesh(esp_handle *h, int32_t dim, int64_t npd, uint64_t seed, int32_t order_mode, int32_t node_mode,
                                         uint64_t node_seed, int64_t cell_begin, int64_t cell_end, int64_t *d_cellnodes, double *d_elmat,
                                         double *d_diag) {
    if (!h) return ESP_ERR_INVALID;
    if ((dim != 2 && dim != 3) || npd < 2) FAIL(h, ESP_ERR_INVALID, "fem: dim must be 2 or 3 and npd >= 2");
    const i64 q = npd - 1;
    const i64 nc = dim == 2 ? 2 * q * q : 6 * q * q * q;
    const i64 nn = dim == 2 ? npd * npd : npd * npd * npd;
    if (cell_begin < 0 || cell_end > nc || cell_begin > cell_end || !d_cellnodes || !d_elmat) FAIL(h, ESP_ERR_INVALID, "fem mesh: bad cell range or NULL output");
    if (cell_begin == cell_end) return ESP_OK;
    (void)hipSetDevice(h->device);
    espelem::MeshArgs a;
    memset(&a, 0, sizeof a);
    a.fem.dim = dim;
    a.fem.npd = npd;
    a.fem.ncells = nc;
    a.fem.seed = seed;
    a.fem.order_mode = order_mode;
    int bits = 2;
    while (((u64)1 << bits) < (u64)nc) bits += 2;
    a.fem.bits = bits;
    espgen::fem_fill_magic(a.fem);
    a.fem.h = 1.0 / (double)(npd - 1);
    a.fem.L = h->L;
    a.nodes = a.fem;
    a.nodes.ncells = nn;
    a.nodes.seed = node_seed;
    a.nodes.order_mode = node_mode ? 1 : 0;
    bits = 2;
    while (((u64)1 << bits) < (u64)nn) bits += 2;
    a.nodes.bits = bits;
    a.p0 = cell_begin;
    a.p1 = cell_end;
    a.cellnodes = d_cellnodes;
    a.elmat = d_elmat;
    a.diag = d_diag;
    hipLaunchKernelGGL(espelem::fem_mesh_k, dim3(grid_for(cell_end - cell_begin, espelem::THREADS)), dim3(espelem::THREADS), 0, h->stream, a);
    HIPCK(h, hipGetLastError());
    return ESP_OK;
}
