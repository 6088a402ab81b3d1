"""round 6 probe: esp_get_nzval into a fresh NumPy vector -- with / without torch in the process, with the previous vector kept
alive or not (what the plug-in form's download pays; tools/r5_plugin_split.py measured 13.7 ms, the plug-in trace 36)."""
import os, sys, time
sys.path.insert(0, ".")
if os.environ.get("WITH_TORCH"):
    import torch
    torch.cuda.init()
import numpy as np
from esparse_loader import load
esp = load()
npd = 3163
nn = npd * npd
A = esp.ExtendableSparseMatrix(nn, nn)
A.generate_fem(2, npd, seed=0x5EED0004, order_mode=1)
A.flush()
d = A._d
Z = d.nnz()
vp = lambda a: a.ctypes.data_as(__import__("ctypes").c_void_p)   # noqa: E731
keep = []
for rep in range(5):
    fresh = np.empty(Z, np.float64)
    t2 = time.perf_counter(); d.ck(d.lib.esp_get_nzval(d.h, vp(fresh))); t3 = time.perf_counter()
    d.ck(d.lib.esp_get_nzval(d.h, vp(fresh))); t4 = time.perf_counter()
    print("torch %s keep %s: get_nzval into np.empty %.1f ms (address %% 2MiB = %d), again %.1f ms" %
          (bool(os.environ.get("WITH_TORCH")), bool(os.environ.get("KEEP")), (t3 - t2) * 1e3, fresh.ctypes.data % (2 << 20), (t4 - t3) * 1e3), flush=True)
    if os.environ.get("KEEP"):
        keep = [fresh]
    del fresh
