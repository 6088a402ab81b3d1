#!/usr/bin/env python3
"""bench.py's sharded step (z-slab producers, column shards) with W ranks run as THREADS on one GPU
(tests/threaddist.py): checks the large-size path of the partitioned exchange (256^3 per rank) when only
one GPU is at hand.  usage: tools/shard_threads_bench.py [W] [n]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

torch.cuda.init()
from esparse_loader import load  # noqa: E402
from threaddist import run_ranks  # noqa: E402

esp = load()
W = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
nzg = n * W
N = n * n * nzg
nodes = n ** 3
E = 12 * n * n * (n - 1) + 6 * n * n
Z_total = N + 2 * ((n - 1) * n * nzg + n * (n - 1) * nzg + n * n * (nzg - 1))


def body(rank, dist):
    be = esp.HipShardBackend(N, N, device=0, capacity_hint=E + 4 * n * n)
    SA = esp.ShardedExtendableSparseMatrix(N, N, be, dist=dist)
    A = be.matrix
    out = []
    for it in range(4):
        dist.barrier()
        t0 = time.perf_counter()
        A.reset()
        A.generate_fdrand_range(n, n, nzg, rank * nodes, (rank + 1) * nodes, seed=0x5EED0002, rand_mode=1, kind=esp.ESP_UPDATE)
        SA.flush()
        A.synchronize()
        dist.barrier()
        out.append((time.perf_counter() - t0, SA.last_exchange, A.debug_last_partition(), SA.sent_off_rank))
    return out, SA.nnz()


res = run_ranks(W, body)
for r, (o, nnz) in enumerate(res):
    print("rank", r, [(round(t * 1e3, 2), ex, part, sent) for (t, ex, part, sent) in o], "global nnz", nnz)
assert all(nnz == Z_total for (_, nnz) in res), (Z_total, [nnz for (_, nnz) in res])
print("ok: global nnz", Z_total, "=", W, "ranks x", n, "^3 nodes")
