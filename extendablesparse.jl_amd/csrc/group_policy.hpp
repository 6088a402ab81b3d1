// group_policy.hpp -- the exchange POLICY of a sharded flush, host-only (no HIP, no RCCL, no handle internals).
//
// esp_group_flush = [agree on how to exchange] -> [move every entry to the rank that owns its column] -> local flush.
// Everything that decides the communication pattern lives here and is taken from all-gathered data, so that all ranks
// always take the same branch: the consensus "is every rank's stream pre-sorted?", the back-off after a "no", the
// digit width of the next partitioned exchange (entries per shard), send / receive offsets, which bytes go to whom.
// What the policy asks of its shard -- partition, export, place, assemble, flush -- is the ShardOps table: the product
// binds it to the esp_shard_* calls on the device (group.hpp), tests/group_policy_test.cpp to a host model run by
// 2 / 3 / 8 threads-as-ranks under -fsanitize=address,undefined.  The transport is the esp_comm_t of the C ABI.
//
// Reference shape: flush! of GenericMTExtendableSparseMatrixCSC (genericmtextendablesparsematrixcsc.jl:45-51: every
// partition's buffer ends in ONE matrix, buffers in tid order), with tid = rank.
#pragma once
#include <stdint.h>

#include <algorithm>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/esparse_hip.h"

namespace espgroup {

typedef int64_t i64;

// keys, values and digit counts are 8-byte records on both sides of the exchange
constexpr i64 REC = 8;

struct ShardOps {
    void *ctx = nullptr;
    int32_t (*pending)(void *ctx, i64 *count) = nullptr;
    // one partition pass: ranges [eoff[q], eoff[q+1]) of keys / vals go to owner q, counts[q * nb + d] = entries of
    // digit d of owner q; *ok = 0: not applicable (the stream is not pre-sorted, the plan does not fit): nothing moved
    int32_t (*partition)(void *ctx, int P, int me, i64 eps, int32_t *ok, void **keys, void **vals, void **counts, i64 *eoff, i64 *nb) = nullptr;
    // announces the next flush's partition(P, me, eps) to the producers (eps < 0: none)
    int32_t (*plan)(void *ctx, int P, int me, i64 eps) = nullptr;
    // the received blocks become pieces of the local segments (entry q of the tables: from rank q; the own one ignored)
    int32_t (*assemble)(void *ctx, const void *const *rk, const void *const *rv, const void *const *rc, const i64 *recv_entries, int32_t *ok) = nullptr;
    int32_t (*counts)(void *ctx, int P, i64 *counts) = nullptr;
    // stable partition by owner; the own chunk lands at position `lower` of the new pending buffer, [soff[q], soff[q+1])
    // of sk / sv go to owner q
    int32_t (*exchange_begin)(void *ctx, int P, int me, i64 lower, i64 higher, void **sk, void **sv, i64 *soff) = nullptr;
    int32_t (*exchange_place)(void *ctx, i64 position, const void *keys, const void *vals, i64 count) = nullptr;
    // receive buffers for nrecv entries (+ ncounts digit counts), alive until the local flush has read them
    int32_t (*recv_buffers)(void *ctx, i64 nrecv, i64 ncounts, void **rk, void **rv, void **rc) = nullptr;
    int32_t (*flush)(void *ctx, int32_t mode, i64 *local_nnz, int32_t *pattern_changed) = nullptr;
    void *(*stream)(void *ctx) = nullptr;  // what the transport orders its transfers on
    // test hook (may be null): after the partition / the owner split, before anything is sent
    int32_t (*after_split)(void *ctx, int partitioned, void *keys, void *vals, i64 entries) = nullptr;
};

struct Policy {
    int P = 1, me = 0;
    esp_comm_t comm{};
    ShardOps ops;
    i64 eps = -1;             // entries per shard of the previous flush: fixes the digit width of the partitioned exchange
    int part_skip = 0, part_penalty = 0;
    int last_exchange = 0;    // 1 partitioned, 2 in place
    i64 sent_off_rank = 0;
    i64 local_nnz = 0;
    std::vector<i64> nnz_offsets;  // P + 1, valid when offsets_valid
    bool offsets_valid = false;
    std::string err;

    void init(int nranks, int rank) {
        P = nranks, me = rank;
        nnz_offsets.assign((size_t)nranks + 1, 0);
    }
    int32_t fail(int32_t code, const char *what, int32_t inner) {
        char b[256];
        snprintf(b, sizeof b, "esp_group: %s failed (%d)", what, (int)inner);
        if (err.empty()) err = b;
        return code;
    }
#define ESPG_CK(call, what)                                  \
    do {                                                     \
        const int32_t s_ = (call);                           \
        if (s_ != ESP_OK) return fail(s_, what, s_);         \
    } while (0)

    // all_gather of a small int64 vector -> P x len (same on every rank)
    int32_t gather(const std::vector<i64> &mine, std::vector<i64> *all) {
        all->assign(mine.size() * (size_t)P, 0);
        ESPG_CK(comm.allgather_i64(comm.ctx, mine.data(), (int32_t)mine.size(), all->data()), "all-gather");
        return ESP_OK;
    }
    int32_t exchange(const std::vector<const void *> &sp, const std::vector<i64> &sb, const std::vector<void *> &rp, const std::vector<i64> &rb) {
        if (P == 1) return ESP_OK;
        ESPG_CK(comm.alltoallv_dev(comm.ctx, sp.data(), sb.data(), rp.data(), rb.data(), ops.stream ? ops.stream(ops.ctx) : nullptr), "all-to-all-v");
        return ESP_OK;
    }

    // One partition pass per rank (owner split + first pass of the local flush), ranges and per-digit counts to the
    // owners, pieces assembled without a copy.  *done = false when the ranks agreed to use the in-place exchange for this
    // flush (some rank's stream is not pre-sorted, or the plan does not apply).
    int32_t exchange_partitioned(bool *done) {
        *done = false;
        if (part_skip > 0) {
            part_skip--;
            return ESP_OK;
        }
        std::vector<i64> all;
        if (eps < 0) {  // first flush: the digit width comes from the global number of entries
            i64 mine_n = 0;
            ESPG_CK(ops.pending(ops.ctx, &mine_n), "pending count");
            ESPG_CK(gather({mine_n}, &all), "all-gather");
            i64 sum = 0;
            for (i64 x : all) sum += x;
            eps = (sum + P - 1) / P;
        }
        int32_t ok = 0;
        void *dk = nullptr, *dv = nullptr, *dc = nullptr;
        std::vector<i64> eoff((size_t)P + 1, 0);
        i64 nb = 0;
        ESPG_CK(ops.partition(ops.ctx, P, me, eps, &ok, &dk, &dv, &dc, eoff.data(), &nb), "partition");
        if (ok && ops.after_split) ESPG_CK(ops.after_split(ops.ctx, 1, dk, dv, eoff[(size_t)P]), "loop-back");
        std::vector<i64> mine((size_t)P + 1, 0);
        mine[0] = ok ? 1 : 0;
        for (int r = 0; r < P; r++) mine[(size_t)r + 1] = ok ? eoff[(size_t)r + 1] - eoff[(size_t)r] : 0;
        ESPG_CK(gather(mine, &all), "all-gather");
        const size_t W = (size_t)P + 1;
        bool all_ok = true;
        i64 total = 0;
        for (int q = 0; q < P; q++) {
            all_ok = all_ok && all[(size_t)q * W] != 0;
            for (int r = 0; r < P; r++) total += all[(size_t)q * W + 1 + (size_t)r];
        }
        if (!all_ok) {  // plain exchange now and for the next few flushes (the pending entries are intact)
            part_penalty = std::min(16, 2 * part_penalty + 1);
            part_skip = part_penalty;
            eps = -1;
            (void)ops.plan(ops.ctx, P, me, -1);
            return ESP_OK;
        }
        part_penalty = 0;
        eps = total ? (total + P - 1) / P : -1;
        // the producers of the NEXT assembly partition for the next flush's partition themselves (the append is the
        // partition, as on one GPU): every rank knows the same entries-per-shard from this flush's all-gather
        (void)ops.plan(ops.ctx, P, me, eps);
        std::vector<i64> in_x((size_t)P, 0), out_x((size_t)P, 0), ro((size_t)P + 1, 0);
        i64 sent = 0;
        for (int r = 0; r < P; r++) {
            in_x[(size_t)r] = r == me ? 0 : mine[(size_t)r + 1];
            out_x[(size_t)r] = r == me ? 0 : all[(size_t)r * W + 1 + (size_t)me];
            sent += in_x[(size_t)r];
            ro[(size_t)r + 1] = ro[(size_t)r] + out_x[(size_t)r];
        }
        const i64 nrecv = ro[(size_t)P];
        void *rkeys = nullptr, *rvals = nullptr, *rcnts = nullptr;
        ESPG_CK(ops.recv_buffers(ops.ctx, nrecv, (i64)P * std::max<i64>(nb, 1), &rkeys, &rvals, &rcnts), "receive buffers");
        // three grouped exchanges -- keys, values, counts -- each with every pair's send and receive, ordered on the
        // shard's stream behind the partition: no host synchronisation
        std::vector<const void *> sp((size_t)P, nullptr);
        std::vector<void *> rp((size_t)P, nullptr);
        std::vector<i64> sb((size_t)P, 0), rb((size_t)P, 0);
        for (int pass = 0; pass < 3; pass++) {
            for (int q = 0; q < P; q++) {
                if (q == me) continue;
                if (pass == 0) {
                    sp[(size_t)q] = (const char *)dk + REC * eoff[(size_t)q], sb[(size_t)q] = REC * in_x[(size_t)q];
                    rp[(size_t)q] = (char *)rkeys + REC * ro[(size_t)q], rb[(size_t)q] = REC * out_x[(size_t)q];
                } else if (pass == 1) {
                    sp[(size_t)q] = (const char *)dv + REC * eoff[(size_t)q], sb[(size_t)q] = REC * in_x[(size_t)q];
                    rp[(size_t)q] = (char *)rvals + REC * ro[(size_t)q], rb[(size_t)q] = REC * out_x[(size_t)q];
                } else {
                    sp[(size_t)q] = (const char *)dc + REC * (i64)q * nb, sb[(size_t)q] = REC * nb;
                    rp[(size_t)q] = (char *)rcnts + REC * (i64)q * nb, rb[(size_t)q] = REC * nb;
                }
            }
            ESPG_CK(exchange(sp, sb, rp, rb), "all-to-all-v");
        }
        std::vector<const void *> rk((size_t)P, nullptr), rv((size_t)P, nullptr), rc((size_t)P, nullptr);
        for (int q = 0; q < P; q++) {
            rk[(size_t)q] = (const char *)rkeys + REC * ro[(size_t)q];
            rv[(size_t)q] = (const char *)rvals + REC * ro[(size_t)q];
            rc[(size_t)q] = (const char *)rcnts + REC * (i64)q * nb;
        }
        int32_t ok2 = 0;
        ESPG_CK(ops.assemble(ops.ctx, rk.data(), rv.data(), rc.data(), out_x.data(), &ok2), "assemble");  // (ok2 = 0: plain pending buffer instead)
        sent_off_rank = sent;
        last_exchange = 1;
        *done = true;
        return ESP_OK;
    }

    // any stream: stable partition by owner, the own chunk stays where it is
    int32_t exchange_inplace() {
        std::vector<i64> cnt((size_t)P, 0), all;
        ESPG_CK(ops.counts(ops.ctx, P, cnt.data()), "owner counts");
        ESPG_CK(gather(cnt, &all), "all-gather");
        std::vector<i64> in_x((size_t)P, 0), out_x((size_t)P, 0), ro((size_t)P + 1, 0);
        i64 lower = 0, higher = 0, sent = 0;
        for (int q = 0; q < P; q++) {
            in_x[(size_t)q] = q == me ? 0 : cnt[(size_t)q];
            out_x[(size_t)q] = q == me ? 0 : all[(size_t)q * (size_t)P + (size_t)me];
            (q < me ? lower : higher) += out_x[(size_t)q];
            sent += in_x[(size_t)q];
            ro[(size_t)q + 1] = ro[(size_t)q] + out_x[(size_t)q];
        }
        void *sk = nullptr, *sv = nullptr;
        std::vector<i64> soff((size_t)P + 1, 0);
        ESPG_CK(ops.exchange_begin(ops.ctx, P, me, lower, higher, &sk, &sv, soff.data()), "owner split");
        if (ops.after_split) ESPG_CK(ops.after_split(ops.ctx, 0, nullptr, nullptr, 0), "loop-back");
        const i64 nrecv = lower + higher;
        void *rkeys = nullptr, *rvals = nullptr, *rcnts = nullptr;
        ESPG_CK(ops.recv_buffers(ops.ctx, nrecv, 0, &rkeys, &rvals, &rcnts), "receive buffers");
        std::vector<const void *> sp((size_t)P, nullptr);
        std::vector<void *> rp((size_t)P, nullptr);
        std::vector<i64> sb((size_t)P, 0), rb((size_t)P, 0);
        for (int pass = 0; pass < 2; pass++) {
            for (int q = 0; q < P; q++) {
                if (q == me) continue;
                sb[(size_t)q] = REC * in_x[(size_t)q];
                rb[(size_t)q] = REC * out_x[(size_t)q];
                sp[(size_t)q] = (const char *)(pass == 0 ? sk : sv) + REC * soff[(size_t)q];
                rp[(size_t)q] = (char *)(pass == 0 ? rkeys : rvals) + REC * ro[(size_t)q];
            }
            ESPG_CK(exchange(sp, sb, rp, rb), "all-to-all-v");
        }
        ESPG_CK(ops.exchange_place(ops.ctx, 0, rkeys, rvals, lower), "place");
        ESPG_CK(ops.exchange_place(ops.ctx, lower + cnt[(size_t)me], (const char *)rkeys + REC * lower, (const char *)rvals + REC * lower, higher), "place");
        sent_off_rank = sent;
        last_exchange = 2;
        return ESP_OK;
    }

    // COLLECTIVE: every rank of the group calls it (like flush! of the MT wrapper it is the synchronisation point)
    int32_t flush(int32_t mode, i64 *local_nnz_out, int32_t *pattern_changed) {
        err.clear();
        bool done = false;
        const int32_t s1 = exchange_partitioned(&done);
        if (s1 != ESP_OK) return s1;
        if (!done) {
            const int32_t s2 = exchange_inplace();
            if (s2 != ESP_OK) return s2;
        }
        i64 z = 0;
        ESPG_CK(ops.flush(ops.ctx, mode, &z, pattern_changed), "local flush");  // (returns after the receive buffers were read)
        local_nnz = z;
        offsets_valid = false;
        if (local_nnz_out) *local_nnz_out = z;
        return ESP_OK;
    }

    // COLLECTIVE on first use after a flush: the ranks' nnz -> every shard's offset into the global colptr
    int32_t offsets() {
        if (offsets_valid) return ESP_OK;
        std::vector<i64> all;
        ESPG_CK(gather({local_nnz}, &all), "all-gather");
        nnz_offsets[0] = 0;
        for (int q = 0; q < P; q++) nnz_offsets[(size_t)q + 1] = nnz_offsets[(size_t)q] + all[(size_t)q];
        offsets_valid = true;
        return ESP_OK;
    }
#undef ESPG_CK
};

}  // namespace espgroup
