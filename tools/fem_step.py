#!/usr/bin/env python3
"""Three assemblies of config 4 (2-D by default: ESP_FEM_DIM / ESP_FEM_NPD) for tools/fem_trace.sh."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

torch.cuda.init()
from esparse_loader import load  # noqa: E402

esp = load()
dim = int(os.environ.get("ESP_FEM_DIM", "2"))
npd = int(os.environ.get("ESP_FEM_NPD", "3163" if dim == 2 else "216"))
nn = npd ** dim
A = esp.ExtendableSparseMatrix(nn, nn)
for it in range(4):
    A.reset()
    A.generate_fem(dim, npd, seed=0x5EED0004, order_mode=1)
    A.flush()
A.synchronize()
print("nnz", A.nnz())
