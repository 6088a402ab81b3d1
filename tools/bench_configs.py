#!/usr/bin/env python3
"""Secondary measurements: BASELINE.json configs 3 (re-assembly + 28 % new entries, merge join hot)
and 4 (unstructured-order P1 FEM, ~10 M DoF).  Prints one JSON line per config (not the driver's
bench line; see bench.py for config 2)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402  (initialise torch's HIP runtime first)

torch.cuda.init()
from esparse_loader import load  # noqa: E402

esp = load()


def timed(A, fn, reps=3):
    fn()
    A.synchronize()
    A.timing(clear=True)
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    A.synchronize()
    dt = (time.perf_counter() - t0) / reps
    tm = A.timing(clear=True)
    return dt, {k: round(v[0] / reps, 3) for k, v in tm.items() if isinstance(v, tuple) and v[0] > 0}


def config3(n=256):
    N = n ** 3
    E = 12 * n * n * (n - 1) + 6 * n * n
    A = esp.ExtendableSparseMatrix(N, N, capacity_hint=E + 2 * n * n * (n - 2))
    A.timing_enable(2)
    # new positions: x second-neighbour pairs (l,l+2),(l+2,l) -- 28.4 % of Z0
    g = torch.arange(N, device="cuda", dtype=torch.int64)
    l = g[(g % n) < n - 2] + 1
    rows = torch.cat([l, l + 2])
    cols = torch.cat([l + 2, l])
    gen = torch.Generator(device="cuda")
    gen.manual_seed(0x5EED0003)
    v = torch.rand(l.numel(), device="cuda", dtype=torch.float64, generator=gen)
    vals = torch.cat([v, v])
    torch.cuda.synchronize()
    import ctypes as C
    d = A._d

    def step():
        A.reset()
        A.generate_fdrand(n, n, n, seed=0x5EED0002, rand_mode=1)
        A.flush()                                           # existing CSC = config-2 result
        A.generate_fdrand(n, n, n, seed=0x5EED0012, rand_mode=1)   # all hits
        d.ck(d.lib.esp_append_device(d.h, C.c_void_p(rows.data_ptr()), C.c_void_p(cols.data_ptr()),
                                     C.c_void_p(vals.data_ptr()), None, esp.ESP_UPDATE, 0, rows.numel()))
        A._touch()
        A.flush()

    dt, st = timed(A, step)
    Z1 = A.nnz()
    assert Z1 == N + 6 * n * n * (n - 1) + rows.numel(), Z1
    return {"config": "3: re-assembly %d^3 + %.1f%% new entries (merge-path join)" % (n, 100.0 * rows.numel() / (Z1 - rows.numel())),
            "ms_per_step_incl_first_build": dt * 1e3, "final_nnz": Z1, "stage_ms": st}


def config4(dim, npd):
    nn = npd ** dim
    A = esp.ExtendableSparseMatrix(nn, nn)
    A.timing_enable(2)
    if os.environ.get("ESP_BENCH_FORCE_PATH"):      # experiments only (see esp_debug_force_path)
        A.debug_force_path(int(os.environ["ESP_BENCH_FORCE_PATH"]))

    def step():
        A.reset()
        A.generate_fem(dim, npd, seed=0x5EED0004, order_mode=1)
        A.flush()

    dt, st = timed(A, step, reps=2)
    q = npd - 1
    E = (2 * q * q if dim == 2 else 6 * q ** 3) * (dim + 1) * (dim + 2)
    return {"config": "4: P1 FEM %dD, %d^%d nodes, random cell order" % (dim, npd, dim), "dof": nn, "appended": E,
            "final_nnz": A.nnz(), "ms_per_step": dt * 1e3, "nnz_per_s": A.nnz() / dt, "appended_per_s": E / dt,
            "path": A.debug_last_path(), "stage_ms": st}


def mul_bench(n=256):
    """mul!(r, A, x) on the device-resident CSC (first call builds the row-wise index)."""
    N = n ** 3
    A = esp.ExtendableSparseMatrix(N, N, capacity_hint=12 * n * n * (n - 1) + 6 * n * n)
    A.generate_fdrand(n, n, n, seed=0x5EED0002, rand_mode=1)
    A.flush()
    x = torch.rand(N, device="cuda", dtype=torch.float64)
    r = torch.empty(N, device="cuda", dtype=torch.float64)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    A.mul(x, out=r)
    first = time.perf_counter() - t0
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        A.mul(x, out=r)
    dt = (time.perf_counter() - t0) / reps
    Z = A.nnz()
    algo = 16.0 * Z + 8.0 * (N + 1) + 16.0 * N      # CSC entries once + colptr + x + r
    return {"config": "mul!(r,A,x) %d^3 stencil on the device CSC" % n, "nnz": Z, "first_call_ms_incl_index_build": first * 1e3,
            "ms_per_mul": dt * 1e3, "algorithmic_GBs": algo / dt / 1e9, "index_bytes_per_nnz": 8}


if __name__ == "__main__":
    which = sys.argv[1:] or ["3", "4a", "4b", "mul"]
    if "mul" in which:
        print(json.dumps(mul_bench(int(os.environ.get("ESP_MUL_N", "256")))), flush=True)
    if "3" in which:
        print(json.dumps(config3(int(os.environ.get("ESP_CFG3_N", "256")))), flush=True)
    if "4a" in which:
        print(json.dumps(config4(2, int(os.environ.get("ESP_CFG4_2D", "3163")))), flush=True)
    if "4b" in which:
        print(json.dumps(config4(3, int(os.environ.get("ESP_CFG4_3D", "216")))), flush=True)
