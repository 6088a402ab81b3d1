#!/usr/bin/env python3
"""Flush time against the column run length (which bucket-kernel tier a matrix lands in):
tools/tier_bench.py [per_col ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

torch.cuda.init()
from esparse_loader import load  # noqa: E402

esp = load()
rng = np.random.default_rng(1)
m, n = 100000, 1 << 17
for per in [int(x) for x in (sys.argv[1:] or ["10", "20", "30", "40", "60", "120"])]:
    cnt = per * n
    J = np.repeat(np.arange(1, n + 1), per)            # column-sorted stream, `per` entries per column
    I = rng.integers(1, m + 1, cnt)
    dup = rng.random(cnt) < 0.5
    I[dup] = (J[dup] % (m - 4)) + 1 + rng.integers(0, 4, dup.sum())   # half of them on 4 rows: duplicates to fold
    V = rng.standard_normal(cnt)
    A = esp.ExtendableSparseMatrix(m, n, capacity_hint=cnt)
    A.timing_enable(1)
    ts = []
    for rep in range(4):
        A.reset()
        A.append(esp.ESP_UPDATE, I, J, V)
        A.synchronize()
        A.timing(clear=True)
        t0 = time.perf_counter()
        A.flush()
        A.synchronize()
        ts.append(time.perf_counter() - t0)
        tm = A.timing(clear=True)
    print("per_col %4d entries %9d flush %.3f ms local %.3f ms (%.1f ps/entry) path %d" % (
        per, cnt, min(ts[1:]) * 1e3, tm["local"][0], tm["local"][0] * 1e9 / cnt, A.debug_last_path()))
