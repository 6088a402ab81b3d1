#!/usr/bin/env python3
"""Re-assembly of a P1 FEM matrix over its stored pattern (every update hits): time per generate + flush, shuffled cell order."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

torch.cuda.init()
from esparse_loader import load  # noqa: E402

esp = load()
for dim, npd in ((2, 3163), (3, 216)):
    nn = npd ** dim
    A = esp.ExtendableSparseMatrix(nn, nn)
    A.timing_enable(2)
    A.generate_fem(dim, npd, seed=4, order_mode=1)
    A.flush()
    z0 = A.nnz()
    for _ in range(2):
        A.generate_fem(dim, npd, seed=5, order_mode=1)
        A.flush()
    A.synchronize()
    A.timing(clear=True)
    reps = 3
    t0 = time.perf_counter()
    for r in range(reps):
        A.generate_fem(dim, npd, seed=6 + r, order_mode=1)
        A.flush()
    A.synchronize()
    dt = (time.perf_counter() - t0) / reps
    tm = A.timing(clear=True)
    assert A.nnz() == z0
    print("FEM %d-D %d^%d re-assembly over the stored pattern: %.2f ms per step" % (dim, npd, dim, dt * 1e3),
          {k: round(v[0] / reps, 3) for k, v in tm.items() if isinstance(v, tuple) and v[0] > 0})
