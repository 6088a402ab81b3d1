#!/usr/bin/env python3
"""The flaky fuzz case (seed 503, case 6): nloc = 10 element batches dealt to 3 partition buffers, esp_flush_sum; repeated, device
against device (digest of the result), in several variants.  usage: r5_case6.py [reps] [variant ...]"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

torch.cuda.init()
from esparse_loader import load  # noqa: E402

esp = load()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
variants = sys.argv[2:] or ["sum3", "single", "each"]
rng = np.random.default_rng(6)
nloc, n, nc, span = 10, 2000000, 60000, 10
start = rng.integers(0, n - span + 1, nc)
local = np.argsort(rng.random((nc, span)), axis=1)[:, :nloc]
perm = rng.permutation(n) + 1
cn = np.asfortranarray(perm[(start[:, None] + local)].T.astype(np.int64))
em = np.asfortranarray(rng.standard_normal((nloc, nloc, nc)))
dg = np.asfortranarray(rng.standard_normal((nloc, nc)))
cuts = [0, 17000, 41000, nc]


def dig(arrs):
    h = hashlib.sha256()
    for a in arrs:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:16]


def dirty(r):
    """leave garbage in the memory the next hipMallocs will get (an uninitialised read shows as a result that changes)"""
    g = torch.Generator(device="cuda")
    g.manual_seed(r)
    t = torch.randint(-2 ** 62, 2 ** 62, (int(os.environ.get("ESP_DIRTY_GB", "12")) * (1 << 27),), dtype=torch.int64, device="cuda", generator=g)
    if r % 3 == 1:
        t.fill_(-1)
    torch.cuda.synchronize()
    del t
    torch.cuda.empty_cache()


for var in variants:
    seen = {}
    for r in range(reps):
        dirty(r)
        try:
            if var == "sum3":
                xs = [esp.SparseMatrixHIPCOO(n, n) for _ in range(3)]
                home = esp.SparseMatrixHIPCOO(n, n)
                for t in range(3):
                    a, b = cuts[t], cuts[t + 1]
                    xs[t].append_elements(cn[:, a:b], em[:, :, a:b], dg[:, a:b])
                csc = esp.SparseMatrixHIPCOO.sum(xs, esp.SparseMatrixCSC(n, n), home=home)
                d = dig(csc.arrays())
            elif var == "single":
                A = esp.ExtendableSparseMatrix(n, n)
                A.append_elements(cn, em, dg)
                A.flush()
                d = dig(A.sparse().arrays())
            else:  # each buffer's batch flushed by itself
                ds = []
                for t in range(3):
                    a, b = cuts[t], cuts[t + 1]
                    A = esp.ExtendableSparseMatrix(n, n)
                    A.append_elements(cn[:, a:b], em[:, :, a:b], dg[:, a:b])
                    A.flush()
                    ds.append(dig(A.sparse().arrays()))
                d = "|".join(ds)
        except Exception as ex:
            d = "EXC " + repr(ex)[:120]
        seen[d] = seen.get(d, 0) + 1
    print(var, "results:", seen, flush=True)
