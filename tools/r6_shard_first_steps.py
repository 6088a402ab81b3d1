#!/usr/bin/env python3
"""Round 6: the first steps of the column-shard path one by one (one rank): step 1 partitions in the flush (no plan yet), step 2 is the
first producer-partitioned one (tables built: with the FINE partition 2^fb times as many digits), later steps reuse them.
usage: python tools/r6_shard_first_steps.py [n]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

torch.cuda.init()
import bench  # noqa: E402
from esparse_loader import load  # noqa: E402

bench.bind_near_gpu(torch, 0)
esp = load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = n ** 3
E, Z = bench.fd_counts(n)
for rep in range(2):
    SA = esp.GroupShardedMatrix(N, N, nranks=1, rank=0, device=0, capacity_hint=E + 8 * n * n, unique_id=esp.GroupShardedMatrix.unique_id())
    A = SA.local
    ts = []
    for it in range(6):
        A.synchronize()
        t0 = time.perf_counter()
        A.reset()
        A.generate_fdrand_range(n, n, n, 0, N, seed=0x5EED0002, rand_mode=1, kind=esp.ESP_UPDATE)
        SA.flush()
        A.synchronize()
        ts.append(time.perf_counter() - t0)
    print("n %d handle %d: steps (ms) %s  key bytes %d" % (n, rep, [round(x * 1e3, 2) for x in ts], A.debug_last_key_bytes()))
    del SA, A
