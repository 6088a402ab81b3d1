#!/usr/bin/env python3
"""Config 3 (re-assembly + new couplings) under the variants of its flush: tools/cfg3_ab.py [force ...]
force 0 = batch + tail, 19 = packed keys + ordinary partition; ESP_TAIL_PLAN=0: no extra prefix bit for an expected tail."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

torch.cuda.init()
from esparse_loader import load  # noqa: E402

esp = load()
n = int(os.environ.get("N", "256"))
N = n ** 3
E = 12 * n * n * (n - 1) + 6 * n * n
g = torch.arange(N, device="cuda", dtype=torch.int64)
l = g[(g % n) < n - 2] + 1
rows = torch.cat([l, l + 2])
cols = torch.cat([l + 2, l])
vals = torch.rand(rows.numel(), device="cuda", dtype=torch.float64)
torch.cuda.synchronize()
for force in [int(a) for a in sys.argv[1:]] or [0, 19]:
    A = esp.ExtendableSparseMatrix(N, N, capacity_hint=E + rows.numel())
    d = A._d
    A.debug_force_path(force)
    dts, tm = [], None
    for it in range(5):
        A.timing_enable(0)
        A.reset()
        A.generate_fdrand(n, n, n, seed=2, rand_mode=1)
        A.flush()
        A.synchronize()
        A.timing_enable(1 if it == 4 else 0)
        A.timing(clear=True)
        t0 = time.perf_counter()
        A.generate_fdrand(n, n, n, seed=3, rand_mode=1)
        d.ck(d.lib.esp_append_device(d.h, C.c_void_p(rows.data_ptr()), C.c_void_p(cols.data_ptr()), C.c_void_p(vals.data_ptr()),
                                     None, esp.ESP_UPDATE, 0, rows.numel()))
        A._touch()
        A.flush()
        A.synchronize()
        if it == 4:
            tm = A.timing(clear=True)
        elif it > 0:
            dts.append(time.perf_counter() - t0)
    print("force", force, "ms %.3f" % (1e3 * sum(dts) / len(dts)), "partition", A.debug_last_partition(), "small", A.debug_last_local_small(),
          "key bytes", A.debug_last_key_bytes(), {k: round(v[0], 3) for k, v in tm.items() if isinstance(v, tuple) and v[0] > 0})
    del A
