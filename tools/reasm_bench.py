#!/usr/bin/env python3
"""Pure re-assembly (the steady state of a time-stepping code): build the 256^3 stencil once, then time
append + flush of the same pattern with new values (every update hits the stored CSC)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

torch.cuda.init()
from esparse_loader import load  # noqa: E402

esp = load()
n = int(os.environ.get("ESP_REASM_N", "256"))
N = n ** 3
A = esp.ExtendableSparseMatrix(N, N, capacity_hint=12 * n * n * (n - 1) + 6 * n * n)
A.timing_enable(1)
A.generate_fdrand(n, n, n, seed=1, rand_mode=1)
A.flush()
Z = A.nnz()
for rep in range(3):
    A.zero_values()
    A.generate_fdrand(n, n, n, seed=2 + rep, rand_mode=1)
    A.flush()
A.synchronize()
A.timing(clear=True)
t0 = time.perf_counter()
R = 5
for rep in range(R):
    A.zero_values()                      # fdrand!'s zero!(A)
    A.generate_fdrand(n, n, n, seed=10 + rep, rand_mode=1)
    A.flush()
A.synchronize()
dt = (time.perf_counter() - t0) / R
tm = A.timing(clear=True)
assert A.nnz() == Z
print({"re-assembly ms": round(dt * 1e3, 3), "stages": {k: round(v[0] / R, 3) for k, v in tm.items() if isinstance(v, tuple) and v[0] > 0}})
