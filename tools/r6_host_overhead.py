#!/usr/bin/env python3
"""Round 6: what the Python mirror classes add to the headline step -- the same 256^3 step through ExtendableSparseMatrix's methods
(bench.py's step) and through the three C-ABI calls it ends in (esp_reset, esp_generate_fdrand, esp_flush), same handle.
usage: python tools/r6_host_overhead.py [steps]"""
import ctypes as C
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
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n = 256
N = n ** 3
E, Z = bench.fd_counts(n)
A = esp.ExtendableSparseMatrix(N, N, device=0, capacity_hint=E)
d = A._d
lib, h = d.lib, d.h


def step_py():
    A.reset()
    A.generate_fdrand(n, n, n, seed=0x5EED0002, rand_mode=1, kind=esp.ESP_UPDATE)
    A.flush()


z, ch = C.c_int64(), C.c_int32()
zr, cr = C.byref(z), C.byref(ch)


def step_c():
    lib.esp_reset(h)
    lib.esp_generate_fdrand(h, n, n, n, 0x5EED0002, 1, esp.ESP_UPDATE)
    lib.esp_flush(h, esp.ESP_FLUSH_ROUTED, zr, cr)


for name, fn in (("classes", step_py), ("c-abi", step_c), ("classes", step_py), ("c-abi", step_c)):
    for _ in range(10):
        fn()
    A.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    A.synchronize()
    print("%-8s %.4f ms per step" % (name, (time.perf_counter() - t0) / steps * 1e3))


# where the host's time goes inside one step (the GPU idles from the end of a flush until the next PART launch starts)
import statistics  # noqa: E402

tr, tg, tf = [], [], []
for _ in range(steps):
    a0 = time.perf_counter()
    lib.esp_reset(h)
    a1 = time.perf_counter()
    lib.esp_generate_fdrand(h, n, n, n, 0x5EED0002, 1, esp.ESP_UPDATE)
    a2 = time.perf_counter()
    lib.esp_flush(h, esp.ESP_FLUSH_ROUTED, zr, cr)
    a3 = time.perf_counter()
    tr.append(a1 - a0), tg.append(a2 - a1), tf.append(a3 - a2)
print("host time per call (median us): esp_reset %.1f  esp_generate_fdrand %.1f  esp_flush %.1f" % (
    statistics.median(tr) * 1e6, statistics.median(tg) * 1e6, statistics.median(tf) * 1e6))
