#!/usr/bin/env python3
"""Round 6: how long the 16 concurrent fills of cfg_mt_sum stay slow in the FIRST GPU process on a fresh box (they were 6.8 - 7.9 ms
there against 1.5 in every later process): the fill phase alone, 400 rounds, a line every 20.
usage: python tools/r6_mt_cold.py [rounds]"""
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
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 400
dim, npd, p = 2, 3163, 16
nn, nloc = npd ** dim, dim + 1
q = npd - 1
nc = 2 * q * q
xs = [esp.SparseMatrixHIPCOO(nn, nn, device=0) for _ in range(p)]
cn = torch.empty((nc, nloc), dtype=torch.int64, device="cuda")
em = torch.empty((nc, nloc, nloc), dtype=torch.float64, device="cuda")
dg = torch.empty((nc, nloc), dtype=torch.float64, device="cuda")
A0 = esp.ExtendableSparseMatrix(nn, nn, device=0)
A0.generate_fem_mesh(dim, npd, cn, em, dg, seed=0x5EED0004, order_mode=0)
A0.synchronize()
cuts = [nc * t // p for t in range(p + 1)]
from concurrent.futures import ThreadPoolExecutor  # noqa: E402

pool = ThreadPoolExecutor(p)


def fill(t):
    xs[t]._d.ck(xs[t]._d.lib.esp_reset(xs[t]._d.h))
    xs[t].append_elements(cn[cuts[t]:cuts[t + 1]], em[cuts[t]:cuts[t + 1]], dg[cuts[t]:cuts[t + 1]])
    xs[t]._d.ck(xs[t]._d.lib.esp_synchronize(xs[t]._d.h))


t_start = time.perf_counter()
acc = []
for it in range(rounds):
    t0 = time.perf_counter()
    list(pool.map(fill, range(p)))
    acc.append(time.perf_counter() - t0)
    if (it + 1) % 20 == 0:
        print("round %4d  at %6.2f s: fills %.2f ms (min %.2f max %.2f of the last 20)" % (
            it + 1, time.perf_counter() - t_start, sum(acc) / len(acc) * 1e3, min(acc) * 1e3, max(acc) * 1e3), flush=True)
        acc = []
