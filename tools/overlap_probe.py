#!/usr/bin/env python3
"""How much of the PART launch (store-bound) hides behind the bucket kernel (latency- / issue-bound) when both are in
flight: two handles, each driven by its own host thread on its own stream, run the 256^3 step back to back; the pair's
rate against one handle alone.  (A probe for DESIGN.md section 9 -- not a product path.)"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

torch.cuda.init()
from esparse_loader import load  # noqa: E402

esp = load()
n = int(os.environ.get("ESP_PROBE_N", "256"))
N = n ** 3
E = 12 * n * n * (n - 1) + 6 * n * n
steps = 20


def make():
    A = esp.ExtendableSparseMatrix(N, N, capacity_hint=E)
    for _ in range(3):
        A.reset()
        A.generate_fdrand(n, n, n, rand_mode=1)
        A.flush()
    return A


def loop(A, k):
    for _ in range(k):
        A.reset()
        A.generate_fdrand(n, n, n, rand_mode=1)
        A.flush()


A, B = make(), make()
torch.cuda.synchronize()
t0 = time.perf_counter()
loop(A, steps)
torch.cuda.synchronize()
one = (time.perf_counter() - t0) / steps
ths = [threading.Thread(target=loop, args=(X, steps)) for X in (A, B)]
t0 = time.perf_counter()
for th in ths:
    th.start()
for th in ths:
    th.join()
torch.cuda.synchronize()
pair = (time.perf_counter() - t0) / (2 * steps)
print("one handle: %.3f ms per step; two handles side by side: %.3f ms per step each way (%.1f %% of one)" % (one * 1e3, pair * 1e3, 100 * pair / one))
