#!/usr/bin/env python3
"""bench.py's sharded step (column shards through the C group API, esp_group_*) with W ranks run as THREADS on one GPU
(transport: the callback table of tests/threaddist.py::ThreadComm): checks the large-size path of the exchange (256^3 per
rank) when only one GPU is at hand.  usage: tools/shard_threads_bench.py [W] [n] [slab|scrambled]
slab: rank r produces the z-slab of nodes it owns (only the cross-slab pairs travel); scrambled: the
z-planes are dealt round-robin to the ranks, so that (W-1)/W of every rank's entries belong to somebody else."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

torch.cuda.init()
from esparse_loader import load  # noqa: E402
from threaddist import ThreadComm, run_ranks  # noqa: E402

esp = load()
W = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
deal = sys.argv[3] if len(sys.argv) > 3 else "slab"
nzg = n * W
N = n * n * nzg
nodes = n ** 3
plane = n * n
E = 12 * n * n * (n - 1) + 6 * n * n
Z_total = N + 2 * ((n - 1) * n * nzg + n * (n - 1) * nzg + n * n * (nzg - 1))
comm = ThreadComm(W, esp._lib)


def produce(A, rank):
    if deal == "slab":
        A.generate_fdrand_range(n, n, nzg, rank * nodes, (rank + 1) * nodes, seed=0x5EED0002, rand_mode=1, kind=esp.ESP_UPDATE)
    else:   # blocks of 8 z-planes, dealt round-robin
        blk = 8 * plane
        for b in range(rank, N // blk, W):
            A.generate_fdrand_range(n, n, nzg, b * blk, (b + 1) * blk, seed=0x5EED0002, rand_mode=1, kind=esp.ESP_UPDATE)


def body(rank, dist):
    holder = {}
    table = comm.table(rank, lambda: holder["A"].local._d.h)
    SA = esp.GroupShardedMatrix(N, N, nranks=W, rank=rank, capacity_hint=E + 8 * n * n, comm=table)
    holder["A"] = SA
    A = SA.local
    out = []
    for it in range(4):
        dist.barrier()
        t0 = time.perf_counter()
        A.reset()
        produce(A, rank)
        SA.flush()
        A.synchronize()
        dist.barrier()
        out.append((time.perf_counter() - t0, SA.last_exchange, A.debug_last_partition(), SA.sent_off_rank))
    # (how the last flush's own range was held: shard_source 2 = the producer partitioned by (owner, digit) itself; key bytes 4 = its
    # own range as 4-byte keys -- above 32 key bits below the plan's prefix through the FINE partition, round 6)
    out.append(("source", A.debug_last_shard_source(), "key_bytes", A.debug_last_key_bytes()))
    return out, SA.nnz()


res = run_ranks(W, body)
if comm.errors:
    raise SystemExit("transport errors: %r" % (comm.errors[:3],))
for r, (o, nnz) in enumerate(res):
    print("rank", r, [(round(t * 1e3, 2), ex, part, sent) for (t, ex, part, sent) in o[:-1]], o[-1], "global nnz", nnz)
assert all(nnz == Z_total for (_, nnz) in res), (Z_total, [nnz for (_, nnz) in res])
print("ok (%s): global nnz" % deal, Z_total, "=", W, "ranks x", n, "^3 nodes")
