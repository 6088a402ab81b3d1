#!/usr/bin/env python3
"""Paths that only problems with more than 32 key bits below the partition prefix take (a 256^3 stencil has exactly 32;
an 8-GPU weak-scaling run has 36): the producer-side partition with packed keys (fdrand_part_k<false,false>), alone and
with multi-window digits (column shards), against the plain producer + the flush's own partition -- three device
results that must agree bit for bit (no CPU oracle at this size).  tools/check_large_keys.py [n]   (default 322)"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

torch.cuda.init()
from esparse_loader import load  # noqa: E402

esp = load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 322
N = n ** 3
E = 12 * n * n * (n - 1) + 6 * n * n


def digest(A):
    cp, rv, nz = A.sparse().arrays()
    h = hashlib.sha256()
    for a in (cp, rv, nz):
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest(), len(rv)


out = {}
for name, force in (("flush_partition", 16), ("producer_partition", 0)):
    A = esp.ExtendableSparseMatrix(N, N, capacity_hint=E)
    A.debug_force_path(force)
    A.generate_fdrand(n, n, n, seed=7, rand_mode=1)
    A.flush()
    out[name] = digest(A) + (A.debug_last_partition(), A.debug_last_key_bytes())
    del A
SA = esp.GroupShardedMatrix(N, N, nranks=1, rank=0, capacity_hint=E)
A = SA.local
for it in range(2):                 # (the second assembly finds the plan of the first flush: the producer partitions)
    A.reset()
    A.generate_fdrand(n, n, n, seed=7, rand_mode=1)
    SA.flush()
out["shard_producer"] = digest(A) + (A.debug_last_partition(), A.debug_last_key_bytes(), A.debug_last_shard_source())
for k, v in out.items():
    print(k, v)
ok = len({v[0] for v in out.values()}) == 1 and out["producer_partition"][2] == 4 and out["shard_producer"][4] == 2
print("large keys:", "ok" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
