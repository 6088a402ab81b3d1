"""CPU, world_size 2, gloo: the column-shard exchange (counts, all-to-all-v, ordering, colptr stitch)
of tests/sharded_model.py (the Python model of the sharded flush) with a CPU stand-in for the local device operations.
The stand-in lives HERE (tests may use the oracle); the product backend is HipShardBackend."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def bits_for(extent):
    b = 1
    while (1 << b) < extent:
        b += 1
    return b


class CpuShardBackend:
    """numpy/oracle stand-in with the same packed-key layout as the device buffer."""

    def __init__(self, m, n, orc, esp):
        self.m, self.n, self.orc, self.esp = m, n, orc, esp
        self.rb = bits_for(m)
        self.keys = np.empty(0, np.int64)
        self.vals = np.empty(0, np.float64)
        self.O = orc.ExtendableSparseMatrix(m, n)

    @property
    def matrix(self):
        return self

    def append(self, kind, I, J, V, op="+", kinds=None):
        I = np.asarray(I, np.int64)
        J = np.asarray(J, np.int64)
        k = np.full(len(I), kind, np.int64) if kinds is None else np.asarray(kinds, np.int64)
        key = ((((J - 1) << self.rb) | (I - 1)) << 2) | k
        self.keys = np.concatenate([self.keys, key])
        self.vals = np.concatenate([self.vals, np.asarray(V, np.float64)])

    def pending(self):
        return len(self.keys)

    def empty(self, count, dtype):
        import torch
        return torch.empty(int(count), dtype=dtype)

    def shard_export(self, P):
        import torch
        col0 = self.keys >> (2 + self.rb)
        owner = (col0 * P) // self.n
        order = np.argsort(owner, kind="stable")
        offsets = np.concatenate([[0], np.cumsum(np.bincount(owner, minlength=P))]).astype(np.int64)
        return torch.from_numpy(self.keys[order].copy()), torch.from_numpy(self.vals[order].copy()), offsets

    def replace_pending(self, pieces):
        self.keys = np.concatenate([k.numpy() for k, _ in pieces]).astype(np.int64)
        self.vals = np.concatenate([v.numpy() for _, v in pieces]).astype(np.float64)

    # -- partitioned exchange, numpy restatement of esp_shard_partition / esp_shard_assemble
    NB = 8            # digits per shard
    fail_partition = False

    def part_partition(self, P, me, entries_per_shard):
        import torch
        if self.fail_partition:
            return None
        nb = self.NB
        bounds = np.array([-(-r * self.n // P) for r in range(P + 1)], np.int64)
        width = -(-int(np.max(np.diff(bounds))) // nb)          # columns per digit
        col0 = self.keys >> (2 + self.rb)
        owner = (col0 * P) // self.n
        g = owner * nb + (col0 - bounds[owner]) // width
        order = np.argsort(g, kind="stable")
        self.keys, self.vals = self.keys[order], self.vals[order]
        self._cnt = np.bincount(g, minlength=P * nb).astype(np.int64)
        self._eoff = np.concatenate([[0], np.cumsum(np.bincount(owner, minlength=P))]).astype(np.int64)
        return (torch.from_numpy(self.keys.copy()), torch.from_numpy(self.vals.copy()), torch.from_numpy(self._cnt.copy()),
                self._eoff, nb)

    def part_wait(self):
        pass

    def part_assemble(self, P, me, rkeys, rvals, rcounts, recv_entries):
        nb = self.NB
        blocks = []
        for q in range(P):
            if q == me:
                k = self.keys[self._eoff[me]:self._eoff[me + 1]]
                v = self.vals[self._eoff[me]:self._eoff[me + 1]]
                c = self._cnt[me * nb:(me + 1) * nb]
            else:
                k, v, c = rkeys[q].numpy(), rvals[q].numpy(), rcounts[q].numpy()
                assert len(k) == recv_entries[q] == int(c.sum())
            blocks.append((k, v, np.concatenate([[0], np.cumsum(c)])))
        ks, vs = [], []
        for d in range(nb):                  # a segment = its pieces in source-rank order
            for (k, v, o) in blocks:
                ks.append(k[o[d]:o[d + 1]])
                vs.append(v[o[d]:o[d + 1]])
        self.keys = np.concatenate(ks).astype(np.int64)
        self.vals = np.concatenate(vs).astype(np.float64)
        return True

    def flush(self):
        kinds = (self.keys & 3).astype(np.uint8)
        I = ((self.keys >> 2) & ((1 << self.rb) - 1)) + 1
        J = (self.keys >> (2 + self.rb)) + 1
        self.O.apply(kinds, I, J, self.vals)
        self.O.flush()
        self.keys = np.empty(0, np.int64)
        self.vals = np.empty(0, np.float64)
        return self.O.nnz()

    def local_csc(self):
        cp, rv, nz = self.O.arrays()
        return self.esp.SparseMatrixCSC(self.m, self.n, cp, rv, nz)


def _worker(rank, world, port, variant, q):
    variant, mode = variant.split("/")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from esparse_loader import load
    from oracle import oracle as orc
    esp = load()
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import sharded_model
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        nx, ny, nz = 7, 6, 5
        N = nx * ny * nz
        I, J, V = orc.fdrand_stream(nx, ny, nz, rand_mode=1, seed=99)
        E = len(I)
        if variant == "slab":      # contiguous halves of the stream (z-slabs): small exchange
            mine = np.arange(E) * world // E == rank
        else:                      # "scrambled": updates dealt round-robin in chunks of 5
            mine = (np.arange(E) // 5) % world == rank
        kinds = np.where(np.arange(E) % 7 == 0, 2, 1).astype(np.uint8)   # mix UPDATE / RAWUPDATE
        be = CpuShardBackend(N, N, orc, esp)
        ctrl = None
        if mode == "partitioned_ctrl":   # the small agreements over a separate control group (bench.py's set-up)
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
            ctrl = dist.new_group(backend="gloo")
            mode = "partitioned"
        A = sharded_model.ShardedExtendableSparseMatrix(N, N, be, ctrl_group=ctrl)
        if mode == "partitioned3":   # counts, keys and values as three collectives (the path of large exchanges)
            sys.modules[type(A).__module__].ONE_MESSAGE_MAX_ELEMS = 0
            mode = "partitioned"
        A.partitioned = mode != "generic"
        be.fail_partition = mode == "rank1_fails" and rank == 1
        A.append(0, I[mine], J[mine], V[mine], kinds=kinds[mine])
        A.flush()
        # every rank takes the same exchange: the choice comes from all-gathered data
        assert A.last_exchange == ("partitioned" if mode == "partitioned" else "generic"), A.last_exchange
        if mode == "partitioned":
            assert A.last_messages == (3 if sys.modules[type(A).__module__].ONE_MESSAGE_MAX_ELEMS == 0 else 1)
        sent, recv = A.exchanged
        assert sent == int(mine.sum())
        # second round on the existing pattern plus new positions, to exercise hit + merge
        I2 = np.concatenate([I[mine], (np.arange(3) * world + rank) % N + 1])
        J2 = np.concatenate([J[mine], (np.arange(3) * 11 + 2 * rank) % N + 1])
        V2 = np.concatenate([V[mine] * 0.5, np.full(3, 1.0 + rank)])
        A.append(1, I2, J2, V2)
        A.flush()
        G = A.gather_sparse(0)
        allI = [None] * world
        dist.all_gather_object(allI, (I[mine], J[mine], V[mine], kinds[mine], I2, J2, V2))
        if rank == 0:
            # reference semantics: ONE buffer fed the streams of rank 0, 1, ... in turn
            O = orc.ExtendableSparseMatrix(N, N)
            for (a, b, c, k, _, _, _) in allI:
                O.apply(k, a, b, c)
            O.flush()
            for (_, _, _, _, a2, b2, c2) in allI:
                O.apply(np.full(len(a2), 1, np.uint8), a2, b2, c2)
            O.flush()
            cp, rv, nzv = O.arrays()
            ok = (np.array_equal(G.colptr, cp) and np.array_equal(G.rowval, rv)
                  and np.array_equal(G.nzval.view(np.uint64), nzv.view(np.uint64)) and A.nnz() == len(rv))
            q.put(("ok" if ok else "mismatch", int(len(rv))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("variant", ["slab/generic", "scrambled/generic", "slab/partitioned", "scrambled/partitioned",
                                     "scrambled/partitioned3", "scrambled/rank1_fails", "scrambled/partitioned_ctrl"])
def test_shard_exchange_world2(variant):
    _run_world(2, variant)


def test_shard_exchange_world3_partitioned():
    """three ranks: every rank exchanges with two others (not only with a neighbour)"""
    _run_world(3, "scrambled/partitioned")


def _run_world(world, variant):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, variant, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0, "rank exited with %r" % p.exitcode
    status, nnz = q.get(timeout=10)
    assert status == "ok" and nnz > 0


def test_owner_ranges(esp):
    for n, P in [(10, 3), (16, 4), (7, 8), (1000003, 8)]:
        r = esp.owner_ranges(n, P)
        assert r[0][0] == 0 and r[-1][1] == n
        for (a, b), (c, d) in zip(r, r[1:]):
            assert b == c
        for p, (a, b) in enumerate(r):
            for col0 in {a, b - 1} if b > a else set():
                assert col0 * P // n == p
