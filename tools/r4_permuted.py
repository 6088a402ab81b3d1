#!/usr/bin/env python3
"""Round 4: esp_append_elements on a mesh whose nodes carry a PERMUTED numbering (nothing leans on grid arithmetic; rows of a
column lie anywhere in the matrix) against the natural numbering, config 4's sizes.  usage: python tools/r4_permuted.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

torch.cuda.init()
from esparse_loader import load  # noqa: E402

esp = load()
for dim, npd in ((2, 3163), (3, 216)):
    nn, nloc = npd ** dim, dim + 1
    q = npd - 1
    nc = 2 * q * q if dim == 2 else 6 * q ** 3
    for node_mode in (0, 1):
        A = esp.ExtendableSparseMatrix(nn, nn)
        cn = torch.empty((nc, nloc), dtype=torch.int64, device="cuda")
        em = torch.empty((nc, nloc, nloc), dtype=torch.float64, device="cuda")
        dg = torch.empty((nc, nloc), dtype=torch.float64, device="cuda")
        A.generate_fem_mesh(dim, npd, cn, em, dg, seed=0x5EED0004, order_mode=1, node_mode=node_mode)
        A.synchronize()
        ts = []
        for it in range(4):
            A.timing_enable(1 if it == 3 else 0)
            A.timing(clear=True)
            A.synchronize()
            t0 = time.perf_counter()
            A.reset()
            A.append_elements(cn, em, dg)
            A.flush()
            A.synchronize()
            ts.append(time.perf_counter() - t0)
        tm = A.timing(clear=True)
        print("%d-D %d^%d node_mode %d: %.2f ms  stages %s  small %d" % (dim, npd, dim, node_mode, min(ts[1:]) * 1e3,
              {k: round(v[0], 2) for k, v in tm.items() if isinstance(v, tuple) and v[0] > 0}, A.debug_last_local_small()), flush=True)
        del A, cn, em, dg
