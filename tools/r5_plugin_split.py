"""Where the plug-in form's host round trip goes (cfg_mt_sum's plugin_same_pattern_ms): values of a 7.0 10^7-entry matrix up
(esp_set_nzval from a NumPy array), down into a FRESH array (np.empty: first touch while the copy runs) and down into an array
that has been written before."""
import sys, time
sys.path.insert(0, ".")
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
nz = np.random.default_rng(0).standard_normal(Z)
vp = lambda a: a.ctypes.data_as(__import__("ctypes").c_void_p)   # noqa: E731
for rep in range(3):
    t0 = time.perf_counter(); d.ck(d.lib.esp_set_nzval(d.h, vp(nz))); t1 = time.perf_counter()
    fresh = np.empty(Z, np.float64)
    t2 = time.perf_counter(); d.ck(d.lib.esp_get_nzval(d.h, vp(fresh))); t3 = time.perf_counter()
    d.ck(d.lib.esp_get_nzval(d.h, vp(fresh))); t4 = time.perf_counter()
    print("nnz %d (%.0f MB): set_nzval %.1f ms, get_nzval into np.empty %.1f ms, into the same array again %.1f ms" %
          (Z, Z * 8 / 1e6, (t1 - t0) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3), flush=True)
    assert np.array_equal(fresh, nz)
    del fresh
