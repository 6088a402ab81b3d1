// shard.hip -- libesparse_hip: column shards and the group API (see internal.hpp for the map of the translation units)
#include "internal.hpp"

// ------------------------------------------------------------------------ shards
// Column-range shards: owner(col) = floor((col-1)*P/n).  Both calls share one
// histogram + scan of the pending entries by owner; the export is one stable partition pass, so
// every destination receives its entries in this shard's append order.
int32_t shard_prepare(esp_handle *h, int P, espradix::Pass *out) {
    if (P < 1 || P > 256) FAIL(h, ESP_ERR_INVALID, "shards: nshards must be in 1..256");
    h->shard_user = true;
    CK(pending_materialize(h));
    h->part_valid = h->part_assembled = false;  // (its tables share scratch arrays with this path)
    h->genplan.valid = h->rawplan.valid = false;  // (seg[1] and the histogram arrays are rewritten: a kept producer plan would read them)
    if ((double)h->n * (double)P >= 9.0e18) FAIL(h, ESP_ERR_UNSUPPORTED, "shards: n*nshards overflows");
    const i64 E = h->count;
    int bits = 1;
    while ((1 << bits) < P) bits++;
    const i64 T = std::max<i64>(1, ceil_div<i64>(E, espradix::TILE));
    CK(ensure(h, h->seg[0], sizeof(i64) * 4));
    CK(ensure(h, h->tilef[0], sizeof(u64) * (size_t)(4 + espscan::workspace_elems(4))));
    espradix::Pass p;
    p.keys_in = (const u64 *)h->keys.p;
    p.vals_in = (const double *)h->vals.p;
    p.keys_out = nullptr;
    p.vals_out = nullptr;
    p.seg_start = (const i64 *)h->seg[0].p;
    p.tile_first = (const i64 *)h->tilef[0].p;
    p.S = 1;
    p.shift = 0;
    p.bits = bits;
    p.base = 0;
    p.span = ~0ull;
    p.err = nullptr;
    p.owner_P = P;
    p.owner_n = h->n;
    p.colshift = ESP_TAG_BITS + h->L.rb;
    const int R = 1 << bits;
    const i64 hn = T * R;
    CK(ensure(h, h->hist, sizeof(u64) * (size_t)(hn + espscan::workspace_elems(hn))));
    p.hist = (u64 *)h->hist.p;
    if (!(h->shard_valid && h->shard_P == P)) {
        hipLaunchKernelGGL(set_i64_k, dim3(1), dim3(1), 0, h->stream, (i64 *)h->seg[0].p, (i64)0, E, (i64)0, (i64)0);
        hipLaunchKernelGGL(set_i64_k, dim3(1), dim3(1), 0, h->stream, (i64 *)h->tilef[0].p, (i64)0, ceil_div<i64>(E, espradix::TILE), (i64)0, (i64)0);
        HIPCK(h, hipMemsetAsync(p.hist, 0, sizeof(u64) * (size_t)hn, h->stream));
        if (E > 0) {
            Span sp(h, ESP_ST_HIST);
            hipLaunchKernelGGL(espradix::tile_hist_k, dim3(espradix::scatter_grid(T)), dim3(espradix::THREADS), 0, h->stream, p);
            sp.add(1);
        }
        {
            Span sp(h, ESP_ST_SCAN);
            sp.add(espscan::exclusive<u64, false>(h->stream, p.hist, p.hist, hn, p.hist + hn));
        }
        // owner offsets = scanned value of (digit, tile 0); R+1 entries into seg[1]
        CK(ensure(h, h->seg[1], sizeof(i64) * (size_t)(R + 1)));
        hipLaunchKernelGGL(espradix::new_segments_k, dim3(grid_for(R + 1, 256)), dim3(256), 0, h->stream, (const u64 *)p.hist,
                           (const i64 *)h->seg[0].p, (const i64 *)h->tilef[0].p, 1, bits, (i64 *)h->seg[1].p, E);
        HIPCK(h, hipGetLastError());
        h->shard_valid = true;
        h->shard_P = P;
    }
    *out = p;
    return ESP_OK;
}

int32_t shard_offsets(esp_handle *h, int P, int64_t *offsets /* P+1 */) {
    std::vector<i64> tmp((size_t)P + 1);
    HIPCK(h, hipMemcpyAsync(tmp.data(), h->seg[1].p, sizeof(i64) * (size_t)(P + 1), hipMemcpyDeviceToHost, h->stream));
    HIPCK(h, hipStreamSynchronize(h->stream));
    for (int d = 0; d <= P; d++) offsets[d] = tmp[(size_t)d];
    offsets[P] = h->count;  // digits >= P never occur
    return ESP_OK;
}

extern "C" int32_t esp_shard_counts(esp_handle *h, int32_t nshards, int64_t *counts) {
    if (!h || !counts) return ESP_ERR_INVALID;
    (void)hipSetDevice(h->device);
    espradix::Pass p;
    CK(shard_prepare(h, nshards, &p));
    std::vector<int64_t> off((size_t)nshards + 1);
    CK(shard_offsets(h, nshards, off.data()));
    for (int d = 0; d < nshards; d++) counts[d] = off[(size_t)d + 1] - off[(size_t)d];
    return ESP_OK;
}

extern "C" int32_t esp_shard_export(esp_handle *h, int32_t nshards, uint64_t *d_keys, double *d_vals, int64_t *offsets) {
    if (!h || !offsets) return ESP_ERR_INVALID;
    if (h->count > 0 && (!d_keys || !d_vals)) return ESP_ERR_INVALID;
    (void)hipSetDevice(h->device);
    espradix::Pass p;
    CK(shard_prepare(h, nshards, &p));
    CK(shard_offsets(h, nshards, offsets));
    if (h->count > 0) {
        p.keys_out = (u64 *)d_keys;
        p.vals_out = d_vals;
        Span sp(h, ESP_ST_SCATTER);
        hipLaunchKernelGGL((espradix::scatter_k<false>), dim3(espradix::scatter_grid(ceil_div<i64>(h->count, espradix::TILE))), dim3(espradix::THREADS), 0,
                           h->stream, p);
        sp.add(1);
        HIPCK(h, hipGetLastError());
        HIPCK(h, hipStreamSynchronize(h->stream));
    }
    return ESP_OK;
}

// In-place exchange (avoids copying what a rank owns itself).  The pending entries are partitioned by
// owner: this rank's own chunk goes straight to its final place behind the `recv_lower` entries it
// will receive from lower ranks, the other chunks go, compacted in owner order, to a send region
// behind the new pending area of the same buffers.  The caller exchanges the send region (RCCL) and
// drops the received chunks in with esp_shard_exchange_place.
__global__ void add_digit_delta_k(u64 *hist, i64 T, int R, const i64 *__restrict__ delta) {
    const i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= T * R) return;
    hist[g] += (u64)delta[g / T];
}

extern "C" int32_t esp_shard_exchange_begin(esp_handle *h, int32_t nshards, int32_t self, int64_t recv_lower,
                                            int64_t recv_higher, uint64_t **d_send_keys, double **d_send_vals,
                                            int64_t *send_offsets) {
    if (!h || !d_send_keys || !d_send_vals || !send_offsets) return ESP_ERR_INVALID;
    if (self < 0 || self >= nshards || recv_lower < 0 || recv_higher < 0) FAIL(h, ESP_ERR_INVALID, "esp_shard_exchange_begin: arguments");
    (void)hipSetDevice(h->device);
    espradix::Pass p;
    CK(shard_prepare(h, nshards, &p));
    std::vector<int64_t> off((size_t)nshards + 1);
    CK(shard_offsets(h, nshards, off.data()));
    const i64 E = h->count;
    const i64 own = off[(size_t)self + 1] - off[(size_t)self];
    const i64 others = E - own;
    const i64 newcount = recv_lower + own + recv_higher;
    const i64 SR = newcount;  // send region starts behind the new pending area
    CK(ensure(h, h->keys2, sizeof(u64) * (size_t)(SR + others + 1)));
    CK(ensure(h, h->vals2, sizeof(double) * (size_t)(SR + others + 1)));
    const int R = 1 << p.bits;
    std::vector<i64> delta((size_t)R, 0);
    i64 sent = 0;
    for (int d = 0; d < nshards; d++) {
        const i64 start = off[(size_t)d], cnt = off[(size_t)d + 1] - start;
        send_offsets[d] = sent;
        if (d == self) {
            delta[(size_t)d] = recv_lower - start;
        } else {
            delta[(size_t)d] = SR + sent - start;
            sent += cnt;
        }
    }
    send_offsets[nshards] = sent;
    if (E > 0) {
        const i64 T = ceil_div<i64>(E, espradix::TILE);
        CK(ensure(h, h->seg[1], sizeof(i64) * (size_t)(R + 1)));
        HIPCK(h, hipMemcpyAsync(h->seg[1].p, delta.data(), sizeof(i64) * (size_t)R, hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(add_digit_delta_k, dim3(grid_for(T * R, 256)), dim3(256), 0, h->stream, p.hist, T, R, (const i64 *)h->seg[1].p);
        p.keys_out = (u64 *)h->keys2.p;
        p.vals_out = (double *)h->vals2.p;
        Span sp(h, ESP_ST_SCATTER);
        hipLaunchKernelGGL((espradix::scatter_k<false>), dim3(espradix::scatter_grid(T)), dim3(espradix::THREADS), 0, h->stream, p);
        sp.add(1);
        HIPCK(h, hipGetLastError());
    }
    HIPCK(h, hipStreamSynchronize(h->stream));
    // the partitioned buffers become the pending buffers
    std::swap(h->keys, h->keys2);
    std::swap(h->vals, h->vals2);
    h->cap = (i64)std::min(h->keys.bytes / sizeof(u64), h->vals.bytes / sizeof(double));
    h->count = newcount;
    h->kind_noted = 0;  // entries of other ranks, with kinds of their own, are about to be placed among these
    pending_changed(h);
    if (h->count > 0) h->kind_uniform = -2;
    *d_send_keys = (uint64_t *)h->keys.p + SR;
    *d_send_vals = (double *)h->vals.p + SR;
    return ESP_OK;
}

extern "C" int32_t esp_shard_exchange_place(esp_handle *h, int64_t position, const uint64_t *d_keys, const double *d_vals,
                                            int64_t count) {
    if (!h || position < 0 || count < 0 || position + count > h->count) return ESP_ERR_INVALID;
    if (count == 0) return ESP_OK;
    (void)hipSetDevice(h->device);
    h->kind_uniform = -2;  // (foreign entries: kinds unknown to this handle's bookkeeping)
    h->kind_noted = 0;
    Span sp(h, ESP_ST_COPY);
    HIPCK(h, hipMemcpyAsync((u64 *)h->keys.p + position, d_keys, sizeof(u64) * (size_t)count, hipMemcpyDeviceToDevice, h->stream));
    HIPCK(h, hipMemcpyAsync((double *)h->vals.p + position, d_vals, sizeof(double) * (size_t)count, hipMemcpyDeviceToDevice, h->stream));
    sp.add(2);
    HIPCK(h, hipStreamSynchronize(h->stream));
    return ESP_OK;
}


// ---- partitioned exchange -------------------------------------------------------------------
// The owner partition of esp_shard_exchange_begin and the first partition pass of the local flush are
// ONE pass here: every rank partitions its pending entries by (owner, digit inside the owner's key
// window) with the run-based single pass, sends every other owner its range together with the
// per-digit counts, and the bucket kernel of the receiving rank reads a segment as the concatenation
// of one piece per source rank (rank order, source order inside: the same deterministic order as one
// buffer fed the ranks' streams in turn).  Nothing is moved a second time and the own range is
// never copied.
__global__ void gather_stride_k(const i64 *__restrict__ src, i64 stride, int count, i64 *__restrict__ dst) {
    const int i = threadIdx.x;
    if (i < count) dst[i] = src[(size_t)i * (size_t)stride];
}

// (fb: the table is 2^fb times finer than the digits -- a producer's FINE partition)
__global__ void diff_counts_k(const i64 *__restrict__ bstart, i64 NB, int fb, i64 *__restrict__ cnt) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < NB) cnt[i] = bstart[(i + 1) << fb] - bstart[i << fb];
}

// esp_shard_assemble's ONE launch.  Workgroups [0, P): source q's row of piece starts, pstart[q][0 .. nb] = exclusive scan of
// its per-digit counts; workgroups [P, P + G): the merged length of 1024 segments each (straight from the counts: no scan
// needed), their longest one and whether a length is negative -- and their part of row `me`, a copy of the bucket starts of
// the own range (absolute positions in the partitioned buffer).  Workgroup 0 also stores the pointer table the bucket kernel reads.  The
// workgroup that finishes last (a ticket that it sets back to zero) gathers the partial results and writes them to PINNED
// host memory: the host reads them behind the launch, without a copy.  (Round 5 ran this as a pointer-table upload, two
// memsets, a device-to-device copy, two kernels and two downloads.)
struct AsmArgs {
    const void *tab[192];    // keys | values (from P) | counts (from 128) of every source
    const void **T;          // device copy of tab
    const i64 *own_bstart;   // own range of the bucket starts (nb + 1)
    i64 *pstart;             // P rows of nb + 1
    unsigned long long *work;  // [0] ticket | [8, 8 + P + 1) summary | [80, 80 + G) longest segment of a workgroup | [80 + G, 80 + 2 G) negative lengths
    const unsigned long long *other;  // kinds_check_k's flag (nullptr: no received block was looked at)
    unsigned long long *host;  // pinned: [0] seq | [1] longest merged segment | [2] negative lengths | [3] other kinds | [4, 4 + P + 1) summary
    unsigned long long seq;
    i64 nb;
    int fb;  // own_bstart holds 2^fb entries per digit (a producer's FINE partition)
    int P, me, G;
};
constexpr int ASM_WORK_WORDS = 80 + 2 * 1024;  // (G <= 1024: at most 2^20 digits per shard)
// summary[q] = entries of source q (q == me: of the own range), summary[P] = start of the own range
__global__ __launch_bounds__(1024) void assemble_k(AsmArgs a) {
    __shared__ i64 lw[2][16];
    __shared__ u32 s_last;
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
    const i64 nb = a.nb;
    unsigned long long *summary = a.work + 8;
    auto put = [](unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto get = [](const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    if (b == 0 && t < 192) a.T[t] = a.tab[t];
    if (b < a.P) {
        const int q = b;
        i64 *out = a.pstart + (size_t)q * (size_t)(nb + 1);
        if (q == a.me) {  // (the row itself is copied by the workgroups below, 1024 digits each: one workgroup would take 50 us for 2^16)
            if (t == 0) {
                put(&summary[q], (unsigned long long)(a.own_bstart[nb << a.fb] - a.own_bstart[0]));
                put(&summary[a.P], (unsigned long long)a.own_bstart[0]);
            }
        } else {
            const i64 *c = static_cast<const i64 *>(a.tab[128 + q]);
            i64 carry = 0;  // (the same in every thread: all of them add up the 16 wave totals of a round)
            int buf = 0;
            for (i64 b0 = 0; b0 < nb; b0 += 1024, buf ^= 1) {
                const i64 d = b0 + t;
                const i64 x = d < nb ? c[d] : 0;
                i64 inc = x;
#pragma unroll
                for (int dlt = 1; dlt < 64; dlt <<= 1) {
                    const i64 o = __shfl_up(inc, dlt, 64);
                    if (lane >= dlt) inc += o;
                }
                if (lane == 63) lw[buf][w] = inc;
                __syncthreads();  // (one barrier per round: the wave totals alternate between two buffers)
                i64 pre = carry, tot = 0;
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const i64 v = lw[buf][i];
                    pre += i < w ? v : 0;
                    tot += v;
                }
                if (d < nb) out[d] = pre + inc - x;
                carry += tot;
            }
            if (t == 0) {
                out[nb] = carry;
                put(&summary[q], (unsigned long long)carry);
            }
        }
    } else {
        const i64 d = (i64)(b - a.P) * 1024 + t;
        i64 tot = 0;
        bool neg = false;
        if (d < nb) {
            const i64 o0 = a.own_bstart[d << a.fb], o1 = a.own_bstart[(d + 1) << a.fb];
            i64 *own_row = a.pstart + (size_t)a.me * (size_t)(nb + 1);
            own_row[d] = o0;
            if (d == nb - 1) own_row[nb] = o1;
            for (int q = 0; q < a.P; q++) {
                const i64 len = q == a.me ? o1 - o0 : static_cast<const i64 *>(a.tab[128 + q])[d];
                neg |= len < 0;
                tot += len;
            }
        }
#pragma unroll
        for (int o = 32; o; o >>= 1) {
            const i64 x = __shfl_xor(tot, o, 64);
            tot = x > tot ? x : tot;
        }
        const u64 negs = __ballot(neg);
        if (lane == 0) {
            lw[0][w] = tot;
            lw[1][w] = (i64)__popcll(negs);
        }
        __syncthreads();
        if (t == 0) {
            i64 mx = 0, ng = 0;
            for (int i = 0; i < 16; i++) {
                mx = lw[0][i] > mx ? lw[0][i] : mx;
                ng += lw[1][i];
            }
            put(&a.work[80 + (b - a.P)], (unsigned long long)mx);
            put(&a.work[80 + a.G + (b - a.P)], (unsigned long long)ng);
        }
    }
    // ---- the workgroup that finishes last publishes
    __threadfence();
    __syncthreads();
    if (t == 0) s_last = atomicAdd(a.work, 1ull) == (unsigned long long)(gridDim.x - 1) ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    i64 mx = 0, ng = 0;
    for (int i = t; i < a.G; i += 1024) {
        const i64 v = (i64)get(&a.work[80 + i]);
        mx = v > mx ? v : mx;
        ng += (i64)get(&a.work[80 + a.G + i]);
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) {
        const i64 x = __shfl_xor(mx, o, 64);
        mx = x > mx ? x : mx;
        ng += __shfl_xor(ng, o, 64);
    }
    if (lane == 0) {
        lw[0][w] = mx;
        lw[1][w] = ng;
    }
    __syncthreads();
    auto host_put = [](unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); };
    if (t <= a.P) host_put(&a.host[4 + t], get(&summary[t]));
    if (t == 0) {
        mx = 0, ng = 0;
        for (int i = 0; i < 16; i++) {
            mx = lw[0][i] > mx ? lw[0][i] : mx;
            ng += lw[1][i];
        }
        host_put(&a.host[1], (unsigned long long)mx);
        host_put(&a.host[2], (unsigned long long)ng);
        host_put(&a.host[3], a.other ? *a.other : 0ull);
        host_put(&a.host[0], a.seq);  // (the host reads behind the launch: a block that still holds another launch's number is an error)
        put(a.work, 0ull);  // (the ticket, for the next launch)
    }
}

// *other = 1 when a received entry is not an UPDATE (the UPDATE-only fold of the bucket kernel is then not used)
__global__ void kinds_check_k(const u64 *__restrict__ keys, i64 count, unsigned long long *__restrict__ other) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const bool bad = i < count && (u32)(keys[i] & ESP_TAG_MASK) != (u32)ESP_UPDATE;
    if (__ballot(bad) != 0ull && (threadIdx.x & 63) == 0) atomicOr(other, 1ull);
}

// merged length of every segment; longest one (a batch and its tail, a stored slice and its new entries, Base.sum: flush.hip, sum.hip)
__global__ void piece_totals_k(const i64 *__restrict__ pstart, int P, i64 nb, unsigned long long *__restrict__ maxlen,
                               unsigned long long *__restrict__ negative) {
    const i64 d = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    i64 tot = 0;
    bool neg = false;
    if (d < nb)
        for (int q = 0; q < P; q++) {
            const i64 len = pstart[(size_t)q * (size_t)(nb + 1) + d + 1] - pstart[(size_t)q * (size_t)(nb + 1) + d];
            neg |= len < 0;
            tot += len;
        }
    if (neg) atomicAdd(negative, 1ull);
#pragma unroll
    for (int o = 32; o; o >>= 1) {  // one atomic per wave
        const i64 x = __shfl_xor(tot, o, 64);
        tot = x > tot ? x : tot;
    }
    if ((threadIdx.x & 63) == 0 && tot > 0) atomicMax(maxlen, (unsigned long long)tot);
}

extern "C" int32_t esp_shard_plan(esp_handle *h, int32_t nshards, int32_t self, int64_t entries_per_shard) {
    if (!h) return ESP_ERR_INVALID;
    h->shard_user = true;
    h->shard_plan.valid = nshards >= 1 && self >= 0 && self < nshards && entries_per_shard >= 0;
    h->shard_plan.P = nshards;
    h->shard_plan.me = self;
    h->shard_plan.eps = entries_per_shard;
    return ESP_OK;
}
extern "C" int32_t esp_debug_last_rebuild(const esp_handle *h, int32_t *on) {
    if (!h || !on) return ESP_ERR_INVALID;
    *on = h->last_rebuild;
    return ESP_OK;
}
extern "C" int32_t esp_debug_last_lazy_items(const esp_handle *h, int32_t *on) {
    if (!h || !on) return ESP_ERR_INVALID;
    *on = h->last_lazy_items;
    return ESP_OK;
}
extern "C" int32_t esp_debug_last_sum_join(const esp_handle *h, int32_t *segments) {
    if (!h || !segments) return ESP_ERR_INVALID;
    *segments = h->last_sum_join;
    return ESP_OK;
}
extern "C" int32_t esp_debug_last_local_small(const esp_handle *h, int32_t *small) {
    if (!h || !small) return ESP_ERR_INVALID;
    *small = h->last_group3 == 4 ? 5 : h->last_group3 == 3 ? 4 : h->last_group3 == 2 ? 3 : h->last_group3 ? 2 : h->last_local_small;
    return ESP_OK;
}
extern "C" int32_t esp_debug_last_shard_source(const esp_handle *h, int32_t *kind) {
    if (!h || !kind) return ESP_ERR_INVALID;
    *kind = h->last_shard_source;
    return ESP_OK;
}

extern "C" int32_t esp_shard_partition(esp_handle *h, int32_t nshards, int32_t self, int64_t entries_per_shard, int32_t *ok,
                                       uint64_t **d_keys, double **d_vals, int64_t **d_counts, int64_t *entry_offsets,
                                       int64_t *digits_per_shard) {
    if (!h || !ok || !d_keys || !d_vals || !d_counts || !entry_offsets || !digits_per_shard) return ESP_ERR_INVALID;
    *ok = 0;
    const int P = nshards;
    if (P < 1 || self < 0 || self >= P || entries_per_shard < 0) FAIL(h, ESP_ERR_INVALID, "esp_shard_partition: arguments");
    (void)hipSetDevice(h->device);
    h->shard_user = true;
    const i64 E = h->count;
    // the producer already partitioned this very batch for this very call (esp_shard_plan): nothing to move
    const bool from_producer = h->pre.valid && h->pre.mw_P == nshards && h->pre.mw_me == self && h->pre.mw_eps == entries_per_shard &&
                               h->pre.E == E && h->pre.key_bytes == 8 && h->force_path != ESP_PATH_SHARD_NOT_APPLICABLE;
    if (!from_producer) CK(pending_materialize(h));
    // (a producer's batch whose own range holds 4-byte keys stays described by `pre` until esp_shard_assemble hands it
    // to the bucket kernel: every other reader of the pending keys goes through pending_materialize)
    h->part_own32 = from_producer && h->pre.own32;
    h->part_kind32 = h->pre.kind;
    if (!h->part_own32) h->pre.valid = false;
    h->last_shard_source = 0;
    h->part_valid = h->part_assembled = false;
    h->part_own_update = h->kind_uniform == ESP_UPDATE && h->kind_noted == h->count;
    if (P > esprun::MW_MAX || P > esplocal::MAX_PIECES || h->force_path == ESP_PATH_SHARD_NOT_APPLICABLE) return ESP_OK;  // caller uses the plain exchange
    if ((double)h->n * (double)P >= 9.0e18) FAIL(h, ESP_ERR_UNSUPPORTED, "shards: n*nshards overflows");
    if (E >= 0xFFFFFFF0ll) FAIL(h, ESP_ERR_UNSUPPORTED, "esp_shard_partition: too many pending entries");
    // every rank derives the same plan from (n, P, entries_per_shard)
    const MwPlan plan = shard_mw_plan(h, P, entries_per_shard);
    if (!plan.ok) return ESP_OK;  // small or odd problem: plain exchange
    const std::vector<u64> &base = plan.base;
    const int K = plan.K, shift = plan.shift, pb = plan.pb;
    const u64 nb64 = plan.nb64;
    const i64 NB = plan.NB;
    // (a producer's FINE partition: its tables are 2^fb times finer than the plan's digits; counts, owner ranges and everything
    // behind this call speak the plan's digits)
    const int fbp = from_producer ? h->pre.fb : 0;
    if (from_producer && (fbp > plan.fb || (fbp > 0 && fbp != plan.fb) || h->pre.mw_shift != shift - fbp || h->pre.mw_nb != (u32)(nb64 << fbp)))
        FAIL(h, ESP_ERR_STATE, "esp_shard_partition: internal error (the producer's plan differs)");
    // tables: bases (<= 64 u64) | owner offsets (<= 65 i64) | counts (NB i64)
    const size_t o_cnt = 256 * 8;
    CK(ensure(h, h->parttab, o_cnt + sizeof(i64) * (size_t)((NB << fbp) + 1)));
    char *T = (char *)h->parttab.p;
    // (a producer's batch: prepart_begin wrote the window bases; the copy would queue behind the PART launch)
    if (!from_producer) HIPCK(h, hipMemcpyAsync(T, base.data(), sizeof(u64) * (size_t)P, hipMemcpyHostToDevice, h->stream));
    i64 *cnt = (i64 *)(T + o_cnt);
    CK(ensure(h, h->seg[1], sizeof(i64) * (size_t)((NB << fbp) + 1)));
    CK(ensure(h, h->tilef[1], sizeof(u64) * (size_t)(NB + 1 + espscan::workspace_elems(NB + 1))));
    i64 *bstart = (i64 *)h->seg[1].p;
    std::vector<i64> off((size_t)P + 1, 0);
    if (from_producer) {
        h->last_shard_source = 2;  // (bucket starts in seg[1], entries in place: the PART launch wrote them)
        h->last_run_order = 0;
        h->shard_valid = false;
    } else if (E > 0) {
        h->last_shard_source = 1;
        CK(ensure(h, h->keys2, sizeof(u64) * (size_t)E));
        CK(ensure(h, h->vals2, sizeof(double) * (size_t)E));
        CK(ensure(h, h->misc, 256));
        HIPCK(h, hipMemsetAsync((unsigned long long *)h->misc.p + 24, 0, 64, h->stream));  // maxlen .. flag words
        MultiWin mw{P, (u32)nb64, (const u64 *)T};
        bool took = false, tiles = false;
        i64 ml = 0;
        CK(run_partition(h, (const u64 *)h->keys.p, (const double *)h->vals.p, (u64 *)h->keys2.p, (double *)h->vals2.p, K, pb, bstart,
                         (u64 *)h->tilef[1].p, &tiles, &took, &ml, &mw, shift));
        if (!took) return ESP_OK;  // not a pre-sorted stream: plain exchange
        std::swap(h->keys, h->keys2);
        std::swap(h->vals, h->vals2);
        h->cap = (i64)std::min(h->keys.bytes / sizeof(u64), h->vals.bytes / sizeof(double));
        h->shard_valid = false;
    } else {
        HIPCK(h, hipMemsetAsync(bstart, 0, sizeof(i64) * (size_t)(NB + 1), h->stream));
    }
    // The counts and owner ranges follow from the bucket starts, which the ranking kernel wrote BEFORE the scatter
    // kernel started: they are produced on the second stream (it already waits for the ranking kernel) and this call
    // returns while the entries are still being moved -- the caller's consensus round runs beside the scatter
    // kernel; esp_synchronize() before the key/value arrays are read.
    // (a producer's batch: the bucket starts were final behind ITS ranking kernel, the PART launch may still run)
    // (a producer's batch over a REUSED plan: the tables are those of the call that built them, and so are the counts and
    // the owner ranges this call derived from them then -- nothing to launch, nothing to wait for)
    esp_handle::ShardOffsets &so = h->shard_offsets;
    const bool kept = from_producer && h->pre.plan_id != 0 && so.plan_id == h->pre.plan_id && so.P == P && so.me == self &&
                      so.eps == entries_per_shard && so.NB == NB && so.E == E && so.cnt_at == (const void *)cnt &&
                      (int)so.off.size() == P + 1 && h->force_path == ESP_PATH_AUTO;
    if (kept) {
        off = so.off;
    } else {
        hipStream_t qs = (E > 0 && (h->last_run_order == 1 || from_producer) && h->aux && h->aux_ev) ? h->aux : h->stream;
        if (qs == h->aux) HIPCK(h, hipStreamWaitEvent(h->aux, h->aux_ev, 0));  // (recorded right behind the ranking kernel)
        hipLaunchKernelGGL(diff_counts_k, dim3(grid_for(NB, 256)), dim3(256), 0, qs, (const i64 *)bstart, NB, fbp, cnt);
        // owner ranges = bucket starts at every multiple of nb
        i64 *d_off = (i64 *)(T + 64 * 8);
        hipLaunchKernelGGL(gather_stride_k, dim3(1), dim3(128), 0, qs, (const i64 *)bstart, (i64)(nb64 << fbp), P + 1, d_off);
        HIPCK(h, hipMemcpyAsync(off.data(), d_off, sizeof(i64) * (size_t)(P + 1), hipMemcpyDeviceToHost, qs));
        HIPCK(h, hipStreamSynchronize(qs));
        so.plan_id = from_producer ? h->pre.plan_id : 0;  // (0: the tables of this call are nobody's plan)
        so.P = P, so.me = self, so.eps = entries_per_shard, so.NB = NB, so.E = E;
        so.cnt_at = cnt;
        so.off = off;
    }
    for (int r = 0; r <= P; r++) entry_offsets[r] = off[(size_t)r];
    *digits_per_shard = (int64_t)nb64;
    *d_keys = (uint64_t *)h->keys.p;
    *d_vals = (double *)h->vals.p;
    *d_counts = cnt;
    h->part_valid = true;
    h->part_P = P;
    h->part_me = self;
    h->part_shift = shift;
    h->part_nb = (u32)nb64;
    h->part_fb = fbp;
    h->part_base = base[(size_t)self];
    h->part_span = (u64)(shard_col0(h->n, P, self + 1) - shard_col0(h->n, P, self)) << h->L.rb;
    *ok = 1;
    return ESP_OK;
}

extern "C" int32_t esp_shard_assemble(esp_handle *h, const uint64_t *const *d_recv_keys, const double *const *d_recv_vals,
                                      const int64_t *const *d_recv_counts, const int64_t *recv_entries, int32_t *ok) {
    if (!h || !d_recv_keys || !d_recv_vals || !d_recv_counts || !recv_entries || !ok) return ESP_ERR_INVALID;
    *ok = 0;
    if (!h->part_valid) FAIL(h, ESP_ERR_STATE, "esp_shard_assemble: no partitioned pending buffer (esp_shard_partition first; no appends in between)");
    (void)hipSetDevice(h->device);
    const int P = h->part_P, me = h->part_me;
    const i64 nb = (i64)h->part_nb;
    const int fb = h->part_fb;  // (a producer's FINE partition: 2^fb table entries per digit)
    const i64 *bstart = (const i64 *)h->seg[1].p + (size_t)me * ((size_t)nb << fb);  // own range of the bucket starts
    // pointer table (keys | values | counts of every source) | piece starts
    const size_t o_ps = 256 * 8;
    CK(ensure(h, h->piecetab, o_ps + sizeof(i64) * (size_t)P * (size_t)(nb + 1)));
    char *T = (char *)h->piecetab.p;
    AsmArgs aa;
    memset(&aa, 0, sizeof aa);
    i64 total_recv = 0;
    bool look = false;
    for (int q = 0; q < P; q++) {
        if (q == me) {
            aa.tab[(size_t)q] = h->keys.p;
            aa.tab[(size_t)P + q] = h->vals.p;
        } else {
            if (recv_entries[q] < 0 || (recv_entries[q] > 0 && (!d_recv_keys[q] || !d_recv_vals[q])) || !d_recv_counts[q])
                FAIL(h, ESP_ERR_INVALID, "esp_shard_assemble: received block %d", q);
            aa.tab[(size_t)q] = d_recv_keys[q];
            aa.tab[(size_t)P + q] = d_recv_vals[q];
            aa.tab[128 + (size_t)q] = d_recv_counts[q];
            total_recv += recv_entries[q];
            look |= h->part_own_update && recv_entries[q] > 0;
        }
    }
    i64 *pstart = (i64 *)(T + o_ps);
    CK(ensure(h, h->misc, 256));
    unsigned long long *d_other = (unsigned long long *)h->misc.p + 26;
    if (!h->pin_asm) HIPCK(h, hipHostMalloc((void **)&h->pin_asm, sizeof(unsigned long long) * 80, hipHostMallocDefault));
    if (h->asmwork.bytes == 0) {  // (the ticket starts at zero and every launch leaves it there)
        CK(ensure(h, h->asmwork, sizeof(unsigned long long) * (size_t)ASM_WORK_WORDS));
        HIPCK(h, hipMemsetAsync(h->asmwork.p, 0, sizeof(unsigned long long) * (size_t)ASM_WORK_WORDS, h->stream));
    }
    const int G = (int)grid_for(nb, 1024);
    if (G > 1024) FAIL(h, ESP_ERR_UNSUPPORTED, "esp_shard_assemble: %lld digits per shard", (long long)nb);
    {
        Span sp(h, ESP_ST_SCAN);
        if (look) {  // (the received blocks are the cross-shard pairs only: a few launches over little data)
            HIPCK(h, hipMemsetAsync(d_other, 0, 8, h->stream));
            for (int q = 0; q < P; q++)
                if (q != me && recv_entries[q] > 0) {
                    hipLaunchKernelGGL(kinds_check_k, dim3(grid_for(recv_entries[q], 256)), dim3(256), 0, h->stream, (const u64 *)d_recv_keys[q],
                                       (i64)recv_entries[q], d_other);
                    sp.add(1);
                }
        }
        aa.T = (const void **)T;
        aa.own_bstart = bstart;
        aa.pstart = pstart;
        aa.work = (unsigned long long *)h->asmwork.p;
        aa.other = look ? d_other : nullptr;
        aa.host = h->pin_asm;
        aa.seq = ++h->asm_seq;
        aa.nb = nb;
        aa.fb = fb;
        aa.P = P, aa.me = me, aa.G = G;
        hipLaunchKernelGGL(assemble_k, dim3((unsigned)(P + G)), dim3(1024), 0, h->stream, aa);
        sp.add(1);
    }
    HIPCK(h, hipStreamSynchronize(h->stream));
    HIPCK(h, hipGetLastError());
    if (h->pin_asm[0] != aa.seq) {
        (void)hipMemsetAsync(h->asmwork.p, 0, 8, h->stream);
        FAIL(h, ESP_ERR_STATE, "esp_shard_assemble: internal error (the table kernel did not publish its results)");
    }
    const unsigned long long mx[3] = {h->pin_asm[1], h->pin_asm[2], h->pin_asm[3]};
    std::vector<i64> last((size_t)P + 1);
    for (int q = 0; q <= P; q++) last[(size_t)q] = (i64)h->pin_asm[4 + q];
    for (int q = 0; q < P; q++)
        if (q != me && last[(size_t)q] != recv_entries[q])
            FAIL(h, ESP_ERR_STATE, "esp_shard_assemble: block from shard %d holds %lld entries, its digit counts sum to %lld", q,
                 (long long)recv_entries[q], (long long)last[(size_t)q]);
    if (mx[1]) FAIL(h, ESP_ERR_STATE, "esp_shard_assemble: negative digit count in a received block");
    h->part_all_update = h->part_own_update && mx[2] == 0;
    const i64 own_n = last[(size_t)me];
    const i64 own[2] = {last[(size_t)P], last[(size_t)P] + own_n};
    const i64 total = own_n + total_recv;
    if (total >= 0xFFFFFFF0ll) FAIL(h, ESP_ERR_UNSUPPORTED, "esp_shard_assemble: too many entries for one flush");
    if ((i64)mx[0] > esplocal::CAP) {
        // a merged segment does not fit the bucket kernel: hand the entries over as a plain pending
        // buffer (lower ranks, own range, higher ranks) -- the next flush partitions it as usual
        if (h->part_own32) {  // (packed keys for the own range first)
            h->count = h->pre.E;
            CK(pending_materialize(h));
        }
        CK(ensure(h, h->keys2, sizeof(u64) * (size_t)std::max<i64>(total, 1)));
        CK(ensure(h, h->vals2, sizeof(double) * (size_t)std::max<i64>(total, 1)));
        i64 at = 0;
        Span sp(h, ESP_ST_COPY);
        for (int q = 0; q < P; q++) {
            const u64 *sk = q == me ? (const u64 *)h->keys.p + own[0] : d_recv_keys[q];
            const double *sv = q == me ? (const double *)h->vals.p + own[0] : d_recv_vals[q];
            const i64 c = q == me ? own_n : recv_entries[q];
            if (c > 0) {
                HIPCK(h, hipMemcpyAsync((u64 *)h->keys2.p + at, sk, sizeof(u64) * (size_t)c, hipMemcpyDeviceToDevice, h->stream));
                HIPCK(h, hipMemcpyAsync((double *)h->vals2.p + at, sv, sizeof(double) * (size_t)c, hipMemcpyDeviceToDevice, h->stream));
                sp.add(2);
            }
            at += c;
        }
        HIPCK(h, hipStreamSynchronize(h->stream));
        std::swap(h->keys, h->keys2);
        std::swap(h->vals, h->vals2);
        h->cap = (i64)std::min(h->keys.bytes / sizeof(u64), h->vals.bytes / sizeof(double));
        h->count = total;
        h->kind_noted = 0;  // (the received entries carry kinds of their own)
        pending_changed(h);
        if (h->count > 0) h->kind_uniform = -2;
        return ESP_OK;
    }
    h->count = total;
    h->part_total = total;
    h->part_maxlen = (i64)mx[0];
    h->part_own_lo = own[0];
    h->pre.valid = false;  // (the batch is the bucket kernel's now; part_own32 says how its own range is stored)
    h->part_assembled = true;
    *ok = 1;
    return ESP_OK;
}


// ------------------------------------------------------------------------ groups (one process per GPU)
#include "group.hpp"

