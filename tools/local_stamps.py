#!/usr/bin/env python3
"""Phase stamps of the bucket kernel (diagnostic build: ESP_EXTRA_FLAGS=-DESP_LOCAL_STAMPS).
Prints the median time every segment spends between the 8 stamps of esplocal::local_k."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

torch.cuda.init()
from esparse_loader import load  # noqa: E402

esp = load()
fem = int(os.environ.get("ESP_STAMP_FEM", "0"))   # e.g. 120: P1 FEM 3-D on 120^3 nodes instead of the stencil
n = 256
fdim = int(os.environ.get("ESP_STAMP_FEM_DIM", "3"))
N = fem ** fdim if fem else n ** 3
A = esp.ExtendableSparseMatrix(N, N, capacity_hint=0 if fem else 12 * n * n * (n - 1) + 6 * n * n)
reasm = int(os.environ.get("ESP_STAMP_REASM", "0"))   # 1: stamps of the SECOND flush (re-assembly over the existing CSC)
cfg3 = int(os.environ.get("ESP_STAMP_CFG3", "0"))     # 1: config 3 -- stamps of the LAST bucket kernel of its flush (the tail's)
if cfg3:
    reasm = 1
    g = torch.arange(N, device="cuda", dtype=torch.int64)
    l = g[(g % n) < n - 2] + 1
    rows3, cols3 = torch.cat([l, l + 2]), torch.cat([l + 2, l])
    vals3 = torch.rand(rows3.numel(), device="cuda", dtype=torch.float64)
    torch.cuda.synchronize()
for it in range(3):
    A.reset()
    if reasm and fem:
        A.generate_fem(fdim, fem, seed=3, order_mode=1)
        A.flush()
    elif reasm:
        A.generate_fdrand(n, n, n, rand_mode=1)
        A.flush()
    if fem:
        A.generate_fem(fdim, fem, seed=4, order_mode=1)
    else:
        A.generate_fdrand(n, n, n, rand_mode=1)
    if cfg3:
        import ctypes as C
        d = A._d
        d.ck(d.lib.esp_append_device(d.h, C.c_void_p(rows3.data_ptr()), C.c_void_p(cols3.data_ptr()), C.c_void_p(vals3.data_ptr()), None,
                                     esp.ESP_UPDATE, 0, rows3.numel()))
        A._touch()
    if it == 2:
        os.environ["ESP_LOCAL_STAMPS"] = "gpurun_out/stamps.bin"
    A.flush()
st16 = np.fromfile("gpurun_out/stamps.bin", dtype=np.uint64).reshape(-1, 16).astype(np.int64)
st = st16[:, :8]
d = np.diff(st, axis=1) * 0.01  # 100 MHz ticks -> us
names = ["segment known -> loads arrived", "column count + scan", "scatter to LDS", "sort (+early look-back) + fold",
         "compaction", "look-back (if not early)", "LDS compact + stores"]
print("segments", len(st), "lifetime us: median %.2f" % np.median((st[:, 7] - st[:, 0]) * 0.01))
for i, nm in enumerate(names):
    print("%-34s median %6.2f  p90 %6.2f" % (nm, np.median(d[:, i]), np.percentile(d[:, i], 90)))
span = (st[:, 7].max() - st[:, 0].min()) * 0.01
print("kernel span %.1f us, %.0f segments resident on average" % (span, (st[:, 7] - st[:, 0]).sum() * 0.01 / span))
# inside "sort + fold" (register tier with early publication): 3 = run start, 8 = sorted + counted,
# 9 = barrier passed, 10 = fold done (wave 0), 11 = look-back done (last wave), 4 = phase end
if st16[:, 8].any():
    for nm, a, b in (("sort + count (wave 0)", 3, 8), ("barrier wait", 8, 9), ("fold (wave 0)", 9, 10),
                     ("look-back (last wave)", 9, 11), ("phase end after fold", 10, 4), ("phase end after look-back", 11, 4)):
        x = (st16[:, b] - st16[:, a]) * 0.01
        print("  %-36s median %6.2f  p90 %6.2f" % (nm, np.median(x), np.percentile(x, 90)))
if st16[:, 14].any():   # group tier: 2 = counts scanned, 12 = keys scattered + row range known, 13 = sorted (wave 0), 14 = folded (wave 0)
    dense = np.median(st16[:, 3] - st16[:, 14]) > 0 and np.median(st16[:, 4] - st16[:, 5]) >= 0   # dense form: 3 = every wave folded, 5 = counts scanned
    for nm, a, b in ((("scatter + row range", 2, 12), ("group sort (wave 0)", 12, 13), ("gather + fold (wave 0)", 13, 14)) +
                     ((("wait for the other waves' folds", 14, 3), ("publish + scan of the columns' counts + barrier", 3, 5),
                       ("dense records + barrier", 5, 4), ("barrier -> look-back wait starts", 4, 6), ("look-back wait + stores", 6, 7)) if dense else
                      (("records + barrier", 14, 4),))):
        x = (st16[:, b] - st16[:, a]) * 0.01
        print("  %-36s median %6.2f  p90 %6.2f" % (nm, np.median(x), np.percentile(x, 90)))
elif st16[:, 12].any():   # radix tier: 2 = counts scanned, 12 = radix sort done, 13 = fold walks done, 4 = records written
    for nm, a, b in (("radix sort", 2, 12), ("fold walks", 12, 13), ("records", 13, 4)):
        x = (st16[:, b] - st16[:, a]) * 0.01
        print("  %-36s median %6.2f  p90 %6.2f" % (nm, np.median(x), np.percentile(x, 90)))
