// group_policy_test.cpp -- the exchange policy of esp_group_flush (csrc/group_policy.hpp: the very code the library runs)
// on the CPU: 2 / 3 / 8 ranks as threads, a host model of a shard behind the ShardOps table, an in-process transport
// behind esp_comm_t.  Built with -fsanitize=address,undefined by tests/test_group_policy.py.  TEST INFRASTRUCTURE.
// `--procs P`: the same scenario with every rank a PROCESS of its own (fork) and a transport over Unix socket pairs, every
// message framed with (operation, sequence number, byte count): ranks that issued their collectives in different orders,
// or with different sizes, meet a frame they do not expect -- what threads sharing one address space cannot show.
//
// Checked after every collective flush, for every rank and column: what the rank flushed for that column is the
// concatenation, in RANK order, of what every rank appended for it in its own append order (= Base.sum(xmatrices, csc)
// of GenericMTExtendableSparseMatrixCSC with tid = rank, genericmtextendablesparsematrixcsc.jl:45-51); and the policy's
// decisions -- partitioned / in-place exchange, back-off after a "not pre-sorted", entries per shard, bytes sent off
// rank, nnz offsets -- are the same on all ranks and what the streams call for.
#include <pthread.h>
#include <stdint.h>
#include <sys/socket.h>
#include <sys/types.h>
#include <sys/wait.h>
#include <unistd.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <vector>

#include "../extendablesparse.jl_amd/csrc/group_policy.hpp"

typedef int64_t i64;
typedef uint64_t u64;

#define REQUIRE(cond, ...)                                              \
    do {                                                                \
        if (!(cond)) {                                                  \
            fprintf(stderr, "group_policy_test: line %d: ", __LINE__);  \
            fprintf(stderr, __VA_ARGS__);                               \
            fprintf(stderr, "\n");                                      \
            abort();                                                    \
        }                                                               \
    } while (0)

// ---- in-process transport -----------------------------------------------------------------------------------------
struct Hub {
    int P;
    pthread_barrier_t bar;
    std::vector<const i64 *> ag_send;
    std::vector<int32_t> ag_count;
    std::vector<const void *const *> a2a_send;
    std::vector<const i64 *> a2a_sbytes;
    explicit Hub(int p) : P(p), ag_send((size_t)p), ag_count((size_t)p), a2a_send((size_t)p), a2a_sbytes((size_t)p) { pthread_barrier_init(&bar, nullptr, (unsigned)p); }
    ~Hub() { pthread_barrier_destroy(&bar); }
};
struct RankComm {
    Hub *hub;
    int me;
};
static int32_t hub_allgather(void *ctx, const int64_t *send, int32_t count, int64_t *recv) {
    RankComm *c = static_cast<RankComm *>(ctx);
    Hub *h = c->hub;
    h->ag_send[(size_t)c->me] = send;
    h->ag_count[(size_t)c->me] = count;
    pthread_barrier_wait(&h->bar);
    for (int q = 0; q < h->P; q++) {
        REQUIRE(h->ag_count[(size_t)q] == count, "all-gather: rank %d brings %d values, rank %d brings %d", q, h->ag_count[(size_t)q], c->me, count);
        memcpy(recv + (size_t)q * (size_t)count, h->ag_send[(size_t)q], sizeof(int64_t) * (size_t)count);
    }
    pthread_barrier_wait(&h->bar);
    return ESP_OK;
}
static int32_t hub_alltoallv(void *ctx, const void *const *send, const int64_t *send_bytes, void *const *recv, const int64_t *recv_bytes, void *) {
    RankComm *c = static_cast<RankComm *>(ctx);
    Hub *h = c->hub;
    h->a2a_send[(size_t)c->me] = send;
    h->a2a_sbytes[(size_t)c->me] = send_bytes;
    pthread_barrier_wait(&h->bar);
    for (int q = 0; q < h->P; q++) {
        if (q == c->me) continue;
        const i64 theirs = h->a2a_sbytes[(size_t)q][c->me];
        REQUIRE(theirs == recv_bytes[q], "all-to-all-v: rank %d sends %lld bytes to rank %d, which expects %lld", q, (long long)theirs, c->me,
                (long long)recv_bytes[q]);
        if (theirs > 0) memcpy(recv[q], h->a2a_send[(size_t)q][c->me], (size_t)theirs);
    }
    pthread_barrier_wait(&h->bar);
    return ESP_OK;
}

// ---- transport between processes: one Unix socket pair per pair of ranks ----------------------------------------------
struct SockComm {
    int P, me;
    std::vector<int> fd;  // fd[q]: this rank's end of the pair (me, q)
    uint64_t seq = 0;     // collectives issued so far: must agree between the two ends of every frame
};
struct Frame {
    uint32_t op;  // 1 all-gather, 2 all-to-all-v
    uint32_t from;
    uint64_t seq;
    int64_t bytes;
};
static void sock_write(int fd, const void *p, size_t n) {
    const char *c = static_cast<const char *>(p);
    while (n > 0) {
        const ssize_t w = write(fd, c, n);
        REQUIRE(w > 0, "socket write failed");
        c += w, n -= (size_t)w;
    }
}
static void sock_read(int fd, void *p, size_t n) {
    char *c = static_cast<char *>(p);
    while (n > 0) {
        const ssize_t r = read(fd, c, n);
        REQUIRE(r > 0, "socket read failed (the peer is gone?)");
        c += r, n -= (size_t)r;
    }
}
// pairwise, lower peers first, the lower rank of a pair sends first: no cycle of waiting ranks (messages of any size)
static void sock_pair_exchange(SockComm *c, int q, uint32_t op, const void *send, int64_t sbytes, void *recv, int64_t rbytes) {
    const Frame mine{op, (uint32_t)c->me, c->seq, sbytes};
    Frame theirs{};
    auto out = [&]() {
        sock_write(c->fd[(size_t)q], &mine, sizeof mine);
        if (sbytes > 0) sock_write(c->fd[(size_t)q], send, (size_t)sbytes);
    };
    auto in = [&]() {
        sock_read(c->fd[(size_t)q], &theirs, sizeof theirs);
        REQUIRE(theirs.op == op && theirs.seq == c->seq && theirs.from == (uint32_t)q,
                "rank %d expects operation %u number %llu from rank %d, got operation %u number %llu from rank %u: the ranks' collectives are out of step", c->me,
                op, (unsigned long long)c->seq, q, theirs.op, (unsigned long long)theirs.seq, theirs.from);
        REQUIRE(theirs.bytes == rbytes, "rank %d expects %lld bytes from rank %d, which sends %lld", c->me, (long long)rbytes, q, (long long)theirs.bytes);
        if (rbytes > 0) sock_read(c->fd[(size_t)q], recv, (size_t)rbytes);
    };
    if (c->me < q) {
        out();
        in();
    } else {
        in();
        out();
    }
}
static int32_t sock_allgather(void *ctx, const int64_t *send, int32_t count, int64_t *recv) {
    SockComm *c = static_cast<SockComm *>(ctx);
    const int64_t bytes = (int64_t)sizeof(int64_t) * count;
    memcpy(recv + (size_t)c->me * (size_t)count, send, (size_t)bytes);
    for (int q = 0; q < c->P; q++)
        if (q != c->me) sock_pair_exchange(c, q, 1u, send, bytes, recv + (size_t)q * (size_t)count, bytes);
    c->seq++;
    return ESP_OK;
}
static int32_t sock_alltoallv(void *ctx, const void *const *send, const int64_t *send_bytes, void *const *recv, const int64_t *recv_bytes, void *) {
    SockComm *c = static_cast<SockComm *>(ctx);
    for (int q = 0; q < c->P; q++)
        if (q != c->me) sock_pair_exchange(c, q, 2u, send[q], send_bytes[q], recv[q], recv_bytes[q]);
    c->seq++;
    return ESP_OK;
}

// ---- host model of a shard ------------------------------------------------------------------------------------------
// an entry = (column 0-based as the key, a value that names its source rank and its position in that rank's stream)
struct Shard {
    int P = 1, me = 0;
    i64 n = 0;                      // columns of the matrix
    std::vector<u64> keys;          // pending, in append order (or: partitioned / exchanged)
    std::vector<double> vals;
    std::vector<u64> sk;            // send buffers of the in-place exchange
    std::vector<double> sv;
    std::vector<i64> cnts;          // digit counts of the partitioned exchange
    std::vector<u64> rk;            // receive buffers
    std::vector<double> rv;
    std::vector<i64> rc;
    i64 nb = 0, part_total = 0;
    std::vector<i64> part_eoff;
    bool partitioned = false;
    i64 plan_eps = -2;
    std::map<u64, std::vector<double>> stored;  // column -> flushed values in order
    int partitions_tried = 0, plans_seen = 0;

    i64 col0(int r) const { return (i64)(((__int128)r * (__int128)n + P - 1) / P); }  // ceil(r n / P): first column of rank r
    int owner(u64 c) const {
        int r = (int)((__int128)c * P / n);
        while (r + 1 < P && (i64)c >= col0(r + 1)) r++;
        while (r > 0 && (i64)c < col0(r)) r--;
        return r;
    }
    static const int NB = 4;  // digits per owner in the model
    int digit(u64 c, int q) const {
        const i64 lo = col0(q), hi = col0(q + 1);
        const i64 w = std::max<i64>(1, (hi - lo + NB - 1) / NB);
        return (int)std::min<i64>(NB - 1, ((i64)c - lo) / w);
    }
    bool presorted() const {  // an assembly loop's stream: few changes of the bucket (owner, digit)
        int runs = 0;
        for (size_t i = 1; i < keys.size(); i++) {
            const int qa = owner(keys[i]), qb = owner(keys[i - 1]);
            runs += qa != qb || digit(keys[i], qa) != digit(keys[i - 1], qb);
        }
        return runs <= 96;
    }
};
static int32_t s_pending(void *c, i64 *count) {
    *count = (i64)static_cast<Shard *>(c)->keys.size();
    return ESP_OK;
}
static int32_t s_plan(void *c, int, int, i64 eps) {
    Shard *s = static_cast<Shard *>(c);
    s->plan_eps = eps;
    s->plans_seen++;
    return ESP_OK;
}
static int32_t s_partition(void *c, int P, int me, i64 eps, int32_t *ok, void **k, void **v, void **cnt, i64 *eoff, i64 *nb) {
    Shard *s = static_cast<Shard *>(c);
    REQUIRE(P == s->P && me == s->me && eps >= 0, "partition(%d, %d, %lld)", P, me, (long long)eps);
    s->partitions_tried++;
    *ok = s->presorted() ? 1 : 0;
    *nb = Shard::NB;
    if (!*ok) return ESP_OK;
    const size_t E = s->keys.size();
    std::vector<size_t> order(E);
    for (size_t i = 0; i < E; i++) order[i] = i;
    auto bucket = [&](size_t i) { const int q = s->owner(s->keys[i]); return q * Shard::NB + s->digit(s->keys[i], q); };
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return bucket(a) < bucket(b); });
    std::vector<u64> nk(E);
    std::vector<double> nv(E);
    s->cnts.assign((size_t)P * Shard::NB, 0);
    for (size_t i = 0; i < E; i++) {
        nk[i] = s->keys[order[i]], nv[i] = s->vals[order[i]];
        s->cnts[(size_t)bucket(order[i])]++;
    }
    s->keys.swap(nk), s->vals.swap(nv);
    s->part_eoff.assign((size_t)P + 1, 0);
    for (int q = 0; q < P; q++) {
        i64 t = 0;
        for (int d = 0; d < Shard::NB; d++) t += s->cnts[(size_t)q * Shard::NB + (size_t)d];
        s->part_eoff[(size_t)q + 1] = s->part_eoff[(size_t)q] + t;
        eoff[q + 1] = s->part_eoff[(size_t)q + 1];
    }
    eoff[0] = 0;
    s->partitioned = true;
    *k = s->keys.data(), *v = s->vals.data(), *cnt = s->cnts.data();
    return ESP_OK;
}
static int32_t s_recv_buffers(void *c, i64 nrecv, i64 ncounts, void **rk, void **rv, void **rc) {
    Shard *s = static_cast<Shard *>(c);
    s->rk.assign((size_t)std::max<i64>(nrecv, 1), ~0ull);
    s->rv.assign((size_t)std::max<i64>(nrecv, 1), -1.0);
    s->rc.assign((size_t)std::max<i64>(ncounts, 1), -1);
    *rk = s->rk.data(), *rv = s->rv.data(), *rc = s->rc.data();
    return ESP_OK;
}
static int32_t s_assemble(void *c, const void *const *rk, const void *const *rv, const void *const *rc, const i64 *recv_entries, int32_t *ok) {
    Shard *s = static_cast<Shard *>(c);
    REQUIRE(s->partitioned, "assemble without a partition");
    // segment (= digit of the own range) by segment: one piece per source, in rank order
    std::vector<u64> nk;
    std::vector<double> nv;
    std::vector<i64> at((size_t)s->P, 0);  // read position inside every source's block
    for (int q = 0; q < s->P; q++) {
        if (q == s->me) continue;
        i64 t = 0;
        for (int d = 0; d < Shard::NB; d++) t += static_cast<const i64 *>(rc[q])[d];
        REQUIRE(t == recv_entries[q], "block of rank %d: %lld entries, digit counts sum to %lld", q, (long long)recv_entries[q], (long long)t);
    }
    i64 own_at = s->part_eoff[(size_t)s->me];
    for (int d = 0; d < Shard::NB; d++)
        for (int q = 0; q < s->P; q++) {
            const i64 cnt = q == s->me ? s->cnts[(size_t)s->me * Shard::NB + (size_t)d] : static_cast<const i64 *>(rc[q])[d];
            for (i64 e = 0; e < cnt; e++) {
                if (q == s->me) {
                    nk.push_back(s->keys[(size_t)own_at]), nv.push_back(s->vals[(size_t)own_at]);
                    own_at++;
                } else {
                    nk.push_back(static_cast<const u64 *>(rk[q])[at[(size_t)q]]), nv.push_back(static_cast<const double *>(rv[q])[at[(size_t)q]]);
                    at[(size_t)q]++;
                }
            }
        }
    s->keys.swap(nk), s->vals.swap(nv);
    s->partitioned = false;
    *ok = 1;
    return ESP_OK;
}
static int32_t s_counts(void *c, int P, i64 *counts) {
    Shard *s = static_cast<Shard *>(c);
    for (int q = 0; q < P; q++) counts[q] = 0;
    for (u64 k : s->keys) counts[s->owner(k)]++;
    return ESP_OK;
}
static int32_t s_exchange_begin(void *c, int P, int me, i64 lower, i64 higher, void **sk, void **sv, i64 *soff) {
    Shard *s = static_cast<Shard *>(c);
    // (a rank whose own partition went through while another rank's did not: its buffer is partitioned -- a stable
    // permutation of its stream, every column's entries still in append order -- and goes through the plain exchange)
    s->partitioned = false;
    std::vector<i64> cnt((size_t)P, 0);
    for (u64 k : s->keys) cnt[(size_t)s->owner(k)]++;
    soff[0] = 0;
    for (int q = 0; q < P; q++) soff[q + 1] = soff[q] + cnt[(size_t)q];
    s->sk.assign(s->keys.size() + 1, 0), s->sv.assign(s->keys.size() + 1, 0.0);
    std::vector<i64> at(soff, soff + P);
    for (size_t i = 0; i < s->keys.size(); i++) {
        const int q = s->owner(s->keys[i]);
        s->sk[(size_t)at[(size_t)q]] = s->keys[i], s->sv[(size_t)at[(size_t)q]] = s->vals[i];
        at[(size_t)q]++;
    }
    const i64 own = cnt[(size_t)me];
    std::vector<u64> nk((size_t)(lower + own + higher), ~0ull);
    std::vector<double> nv((size_t)(lower + own + higher), -1.0);
    for (i64 e = 0; e < own; e++) nk[(size_t)(lower + e)] = s->sk[(size_t)(soff[me] + e)], nv[(size_t)(lower + e)] = s->sv[(size_t)(soff[me] + e)];
    s->keys.swap(nk), s->vals.swap(nv);
    *sk = s->sk.data(), *sv = s->sv.data();
    return ESP_OK;
}
static int32_t s_exchange_place(void *c, i64 position, const void *keys, const void *vals, i64 count) {
    Shard *s = static_cast<Shard *>(c);
    REQUIRE(position >= 0 && position + count <= (i64)s->keys.size(), "place [%lld, %lld) in a buffer of %zu", (long long)position,
            (long long)(position + count), s->keys.size());
    if (count > 0) {
        memcpy(s->keys.data() + position, keys, sizeof(u64) * (size_t)count);
        memcpy(s->vals.data() + position, vals, sizeof(double) * (size_t)count);
    }
    return ESP_OK;
}
static int32_t s_flush(void *c, int32_t, i64 *local_nnz, int32_t *changed) {
    Shard *s = static_cast<Shard *>(c);
    REQUIRE(!s->partitioned, "local flush of a partitioned buffer that was never assembled");
    for (size_t i = 0; i < s->keys.size(); i++) {
        REQUIRE(s->keys[i] != ~0ull, "rank %d flushes a slot nobody filled", s->me);
        REQUIRE(s->owner(s->keys[i]) == s->me, "rank %d flushes column %llu of rank %d", s->me, (unsigned long long)s->keys[i], s->owner(s->keys[i]));
        s->stored[s->keys[i]].push_back(s->vals[i]);
    }
    s->keys.clear(), s->vals.clear();
    *local_nnz = (i64)s->stored.size();
    if (changed) *changed = 1;
    return ESP_OK;
}

// ---- the scenario ----------------------------------------------------------------------------------------------------
struct Round {
    const char *name;
    int shuffled_rank;   // -1: every rank's stream is pre-sorted
    int empty_rank;      // -1: none
    int expect;          // exchange every rank must report: 1 partitioned, 2 in place
};
struct Job {
    Hub *hub;
    SockComm *sock = nullptr;  // --procs: the transport between processes instead of the hub
    int P, me;
    i64 n;
    std::vector<Round> rounds;
    std::vector<std::vector<std::pair<u64, double>>> *streams;  // [round * P + rank]
    std::vector<int> kinds;       // what this rank saw per round
    std::vector<i64> sent, nnz_before, nnz_total;
    Shard shard;
};
static double tag(int rank, int round, i64 pos) { return (double)rank * 1.0e9 + (double)round * 1.0e7 + (double)pos; }

static void *run_rank(void *arg) {
    Job *j = static_cast<Job *>(arg);
    RankComm rc{j->hub, j->me};
    espgroup::Policy pol;
    pol.init(j->P, j->me);
    pol.comm.ctx = &rc;
    pol.comm.allgather_i64 = hub_allgather;
    pol.comm.alltoallv_dev = hub_alltoallv;
    if (j->sock) {
        pol.comm.ctx = j->sock;
        pol.comm.allgather_i64 = sock_allgather;
        pol.comm.alltoallv_dev = sock_alltoallv;
    }
    Shard &s = j->shard;
    s.P = j->P, s.me = j->me, s.n = j->n;
    espgroup::ShardOps &o = pol.ops;
    o.ctx = &s;
    o.pending = s_pending, o.partition = s_partition, o.plan = s_plan, o.assemble = s_assemble, o.counts = s_counts;
    o.exchange_begin = s_exchange_begin, o.exchange_place = s_exchange_place, o.recv_buffers = s_recv_buffers, o.flush = s_flush;
    for (size_t r = 0; r < j->rounds.size(); r++) {
        const auto &st = (*j->streams)[r * (size_t)j->P + (size_t)j->me];
        for (const auto &e : st) s.keys.push_back(e.first), s.vals.push_back(e.second);
        i64 z = 0;
        int32_t ch = 0;
        const int32_t rc2 = pol.flush(ESP_FLUSH_ROUTED, &z, &ch);
        REQUIRE(rc2 == ESP_OK, "rank %d round %zu: flush -> %d (%s)", j->me, r, rc2, pol.err.c_str());
        j->kinds.push_back(pol.last_exchange);
        j->sent.push_back(pol.sent_off_rank);
        REQUIRE(pol.offsets() == ESP_OK, "offsets");
        j->nnz_before.push_back(pol.nnz_offsets[(size_t)j->me]);
        j->nnz_total.push_back(pol.nnz_offsets[(size_t)j->P]);
    }
    return nullptr;
}

// procs: every rank a forked process over socket pairs; each child checks its own shard and reports its per-round decisions to
// the parent through a pipe, the parent checks what must agree across the ranks
static void scenario(int P, bool procs = false) {
    const i64 n = 4000 * (i64)P + 37;
    // slab, slab, one rank shuffled (consensus: in place; back-off 1), slab (skipped: in place), slab (partitioned again),
    // one rank empty, two shuffled rounds in a row (back-off 1, then 3)
    std::vector<Round> rounds = {{"slab", -1, -1, 1}, {"slab again", -1, -1, 1}, {"last rank shuffled", P - 1, -1, 2}, {"slab (back-off)", -1, -1, 2},
                                 {"slab", -1, -1, 1}, {"rank 0 empty", -1, 0, 1}, {"rank 0 shuffled", 0, -1, 2}, {"slab (back-off)", -1, -1, 2},
                                 {"slab", -1, -1, 1}};
    Shard ref;
    ref.P = P, ref.n = n;
    std::vector<std::vector<std::pair<u64, double>>> streams(rounds.size() * (size_t)P);
    u64 rng = 88172645463325252ull + (u64)P;
    auto next = [&]() { rng ^= rng << 13, rng ^= rng >> 7, rng ^= rng << 17; return rng; };
    for (size_t r = 0; r < rounds.size(); r++)
        for (int q = 0; q < P; q++) {
            auto &st = streams[r * (size_t)P + (size_t)q];
            if (rounds[r].empty_rank == q) continue;
            const i64 lo = ref.col0(q), hi = ref.col0(q + 1);
            const i64 cnt = 3000 + 200 * q;
            for (i64 e = 0; e < cnt; e++) {
                // the rank's own slab in ascending order with duplicates, now and then a column of a neighbour (cross-slab pairs)
                i64 c = lo + (e * (hi - lo)) / cnt;
                if (next() % 16 == 0) c = std::min<i64>(n - 1, std::max<i64>(0, c + (i64)(next() % 21) - 10));
                if (next() % 256 == 0 && q + 1 < P) c = std::min<i64>(n - 1, hi + (i64)(next() % 8));
                st.push_back({(u64)c, tag(q, (int)r, e)});
            }
            if (rounds[r].shuffled_rank == q) {
                for (size_t i = st.size(); i > 1; i--) std::swap(st[i - 1].first, st[(size_t)(next() % i)].first);  // columns in random order
                for (size_t i = 0; i < st.size(); i++) st[i].first = (u64)(next() % (u64)n);
            }
        }
    // ---- expected: per column, round by round, the ranks' entries in rank order, each rank's in its append order
    std::map<u64, std::vector<double>> want;
    for (size_t r = 0; r < rounds.size(); r++)
        for (int q = 0; q < P; q++)
            for (const auto &e : streams[r * (size_t)P + (size_t)q]) want[e.first].push_back(e.second);
    Hub hub(P);
    std::vector<Job> jobs((size_t)P);
    for (int q = 0; q < P; q++) {
        jobs[(size_t)q].hub = &hub, jobs[(size_t)q].P = P, jobs[(size_t)q].me = q, jobs[(size_t)q].n = n;
        jobs[(size_t)q].rounds = rounds, jobs[(size_t)q].streams = &streams;
    }
    if (procs) {
        // socket pairs for every pair of ranks, a report pipe per rank; then one process per rank
        std::vector<std::vector<int>> fds((size_t)P, std::vector<int>((size_t)P, -1));
        for (int a = 0; a < P; a++)
            for (int b = a + 1; b < P; b++) {
                int sv[2];
                REQUIRE(socketpair(AF_UNIX, SOCK_STREAM, 0, sv) == 0, "socketpair");
                fds[(size_t)a][(size_t)b] = sv[0], fds[(size_t)b][(size_t)a] = sv[1];
            }
        std::vector<int> rep_r((size_t)P), rep_w((size_t)P);
        std::vector<pid_t> pid((size_t)P);
        for (int q = 0; q < P; q++) {
            int pp[2];
            REQUIRE(pipe(pp) == 0, "pipe");
            rep_r[(size_t)q] = pp[0], rep_w[(size_t)q] = pp[1];
        }
        fflush(stdout);
        for (int q = 0; q < P; q++) {
            pid[(size_t)q] = fork();
            REQUIRE(pid[(size_t)q] >= 0, "fork");
            if (pid[(size_t)q] == 0) {
                for (int a = 0; a < P; a++)
                    for (int b = 0; b < P; b++)
                        if (a != q && fds[(size_t)a][(size_t)b] >= 0) close(fds[(size_t)a][(size_t)b]);  // (the other ranks' ends)
                SockComm sc{P, q, fds[(size_t)q]};
                Job &j = jobs[(size_t)q];
                j.sock = &sc;
                run_rank(&j);
                // this rank's own shard against what every rank appended for its columns
                size_t mine = 0;
                for (const auto &kv : want) mine += ref.owner(kv.first) == q;
                REQUIRE(j.shard.stored.size() == mine, "P = %d rank %d: %zu columns flushed, %zu appended for it", P, q, j.shard.stored.size(), mine);
                for (const auto &kv : j.shard.stored) {
                    REQUIRE(ref.owner(kv.first) == q, "column %llu stored on rank %d", (unsigned long long)kv.first, q);
                    const auto it = want.find(kv.first);
                    REQUIRE(it != want.end() && it->second == kv.second, "P = %d: column %llu on rank %d: order or content of its %zu updates differs", P,
                            (unsigned long long)kv.first, q, kv.second.size());
                }
                std::vector<i64> rep;
                for (size_t r = 0; r < rounds.size(); r++) {
                    rep.push_back(j.kinds[r]), rep.push_back(j.sent[r]), rep.push_back(j.nnz_before[r]), rep.push_back(j.nnz_total[r]);
                }
                sock_write(rep_w[(size_t)q], rep.data(), sizeof(i64) * rep.size());
                _exit(0);
            }
        }
        for (int a = 0; a < P; a++)
            for (int b = 0; b < P; b++)
                if (fds[(size_t)a][(size_t)b] >= 0) close(fds[(size_t)a][(size_t)b]);
        for (int q = 0; q < P; q++) {
            int st = 0;
            REQUIRE(waitpid(pid[(size_t)q], &st, 0) == pid[(size_t)q] && WIFEXITED(st) && WEXITSTATUS(st) == 0, "P = %d: the process of rank %d failed (status %d)", P, q,
                    st);
            std::vector<i64> rep(4 * rounds.size());
            sock_read(rep_r[(size_t)q], rep.data(), sizeof(i64) * rep.size());
            close(rep_r[(size_t)q]), close(rep_w[(size_t)q]);
            Job &j = jobs[(size_t)q];
            for (size_t r = 0; r < rounds.size(); r++) {
                j.kinds.push_back((int)rep[4 * r]), j.sent.push_back(rep[4 * r + 1]), j.nnz_before.push_back(rep[4 * r + 2]), j.nnz_total.push_back(rep[4 * r + 3]);
            }
        }
    } else {
        std::vector<pthread_t> th((size_t)P);
        for (int q = 0; q < P; q++) pthread_create(&th[(size_t)q], nullptr, run_rank, &jobs[(size_t)q]);
        for (int q = 0; q < P; q++) pthread_join(th[(size_t)q], nullptr);
    }
    size_t seen = procs ? want.size() : 0;  // (--procs: every child has checked its own shard)
    for (int q = 0; q < P && !procs; q++) {
        for (const auto &kv : jobs[(size_t)q].shard.stored) {
            REQUIRE(ref.owner(kv.first) == q, "column %llu stored on rank %d", (unsigned long long)kv.first, q);
            const auto it = want.find(kv.first);
            REQUIRE(it != want.end(), "column %llu was never appended", (unsigned long long)kv.first);
            REQUIRE(it->second == kv.second, "P = %d: column %llu on rank %d: order or content of its %zu updates differs", P, (unsigned long long)kv.first, q,
                    kv.second.size());
            seen++;
        }
    }
    REQUIRE(seen == want.size(), "P = %d: %zu columns flushed, %zu appended", P, seen, want.size());
    for (size_t r = 0; r < rounds.size(); r++) {
        i64 sent_total = 0, expect_sent = 0;
        for (int q = 0; q < P; q++) {
            REQUIRE(jobs[(size_t)q].kinds[r] == rounds[r].expect, "P = %d round %zu (%s): rank %d took exchange %d, expected %d", P, r, rounds[r].name, q,
                    jobs[(size_t)q].kinds[r], rounds[r].expect);
            sent_total += jobs[(size_t)q].sent[r];
            for (const auto &e : streams[r * (size_t)P + (size_t)q]) expect_sent += ref.owner(e.first) != q;
            REQUIRE(jobs[(size_t)q].nnz_total[r] == jobs[0].nnz_total[r], "global nnz differs between ranks");
            if (q > 0) REQUIRE(jobs[(size_t)q].nnz_before[r] >= jobs[(size_t)q - 1].nnz_before[r], "nnz offsets not monotone");
        }
        REQUIRE(sent_total == expect_sent, "P = %d round %zu: %lld entries sent off rank, %lld cross the shard boundaries", P, r, (long long)sent_total,
                (long long)expect_sent);
    }
    printf("group_policy_test: P = %d %s ok (%zu rounds, %zu columns)\n", P, procs ? "processes" : "threads", rounds.size(), want.size());
}

int main(int argc, char **argv) {
    if (argc >= 3 && strcmp(argv[1], "--procs") == 0) {
        const int P = atoi(argv[2]);
        REQUIRE(P >= 1 && P <= 16, "--procs P: 1 .. 16");
        scenario(P, true);
        printf("group_policy_test: ok\n");
        return 0;
    }
    scenario(1);
    scenario(2);
    scenario(3);
    scenario(8);
    printf("group_policy_test: ok\n");
    return 0;
}
