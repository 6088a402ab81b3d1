#!/usr/bin/env python3
"""PCIe-inclusive rates of the host-buffer boundary (what a Julia caller sees): bulk esp_append_host of the
256^3 stencil stream from NumPy arrays + flush, and esp_get_csc back to host arrays."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

torch.cuda.init()
from esparse_loader import load  # noqa: E402

esp = load()
n = int(os.environ.get("ESP_HOST_N", "160"))
N = n ** 3
E = 12 * n * n * (n - 1) + 6 * n * n
rng = np.random.default_rng(1)
# any stream of the right size will do for the transfer rate: columns ascending with jitter, like an assembly loop
J = np.minimum(N, np.maximum(1, (np.arange(E) * N // E) + rng.integers(-n * n, n * n + 1, E))).astype(np.int64)
I = np.minimum(N, np.maximum(1, J + rng.integers(-2, 3, E))).astype(np.int64)
V = rng.standard_normal(E)
A = esp.ExtendableSparseMatrix(N, N, capacity_hint=E)
for rep in range(3):
    A.reset()
    A.synchronize()
    t0 = time.perf_counter()
    A.append(esp.ESP_UPDATE, I, J, V)
    A.synchronize()
    t1 = time.perf_counter()
    A.flush()
    A.synchronize()
    t2 = time.perf_counter()
    csc = A.sparse()
    t3 = time.perf_counter()
import ctypes as C  # noqa: E402
d = A._d
Z = csc.nnz()
cp = np.zeros(N + 1, np.int64)
rv = np.zeros(Z, np.int64)
nz = np.zeros(Z, np.float64)          # (touched: no first-touch page faults inside the timed copy)
vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
t4 = time.perf_counter()
d.ck(d.lib.esp_get_csc(d.h, vp(cp), vp(rv), vp(nz)))
t5 = time.perf_counter()
print({"get_csc_into_touched_arrays_s": t5 - t4, "GBs": (16.0 * Z + 8.0 * N) / (t5 - t4) / 1e9})
print({"entries": E, "append_host_s": t1 - t0, "append_entries_per_s": E / (t1 - t0), "append_GBs": 24.0 * E / (t1 - t0) / 1e9,
       "flush_s": t2 - t1, "get_csc_s": t3 - t2, "nnz": csc.nnz(), "get_csc_GBs": (16.0 * csc.nnz() + 8.0 * N) / (t3 - t2) / 1e9})
