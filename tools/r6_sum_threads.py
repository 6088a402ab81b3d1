"""round 6: does esp_flush_sum's general path (buffers that are NOT element batches: per-entry calls committed chunk by chunk, the
reference's own MT pattern) gain from folding its buffers on one host thread each, now that distinct handles are independent?
(Measured with the experiments switch ESP_SUM_THREADS=1 of the tree before; the product then folded on its host pool, and since the
second session of round 6 folds all buffers in one flush: this prints the product figure; ESP_SUM_HOOK=31 puts a test hook on the
buffers = the one-by-one folds with plans for the whole key window.)  usage: [ESP_SUM_HOOK=31] r6_sum_threads.py [p] [entries per buffer]"""
import ctypes as C, os, sys, time
sys.path.insert(0, ".")
import numpy as np
from esparse_loader import load
esp = load()
p = int(sys.argv[1]) if len(sys.argv) > 1 else 16
cnt = int(sys.argv[2]) if len(sys.argv) > 2 else 2000000
n = 4000000
rng = np.random.default_rng(1)
xs = [esp.SparseMatrixHIPCOO(n, n) for _ in range(p)]
home = esp.SparseMatrixHIPCOO(n, n)
if os.environ.get("ESP_SUM_HOOK"):   # (a test hook on the buffers: no batched folds, no occupied-range plan -- the forms before them)
    for x in xs:
        x._d.ck(x._d.lib.esp_debug_force_path(x._d.h, int(os.environ["ESP_SUM_HOOK"])))
data = []
for t in range(p):
    J = np.sort(rng.integers(t * n // p + 1, (t + 1) * n // p + 1, cnt))
    I = np.clip(J + rng.integers(-30, 31, cnt), 1, n)
    data.append((I, J, rng.standard_normal(cnt), rng.integers(0, 3, cnt).astype(np.uint8)))
hd = home._d
arr = (C.c_void_p * p)(*[x._d.h for x in xs])
ts = []
for it in range(6):
    hd.ck(hd.lib.esp_reset(hd.h))
    for t in range(int(os.environ.get("ESP_SUM_FILL", p))):     # (ESP_SUM_FILL=1: only the first buffer -- ONE band of the matrix)
        I, J, V, K = data[t]
        xs[t].append(0, I, J, V, kinds=K)
    for x in xs:
        x._d.ck(x._d.lib.esp_synchronize(x._d.h))
    z, ch = C.c_int64(), C.c_int32()
    t0 = time.perf_counter()
    hd.ck(hd.lib.esp_flush_sum(hd.h, arr, p, C.byref(z), C.byref(ch)))
    hd.ck(hd.lib.esp_synchronize(hd.h))
    ts.append(time.perf_counter() - t0)
f_ms, c_ms = C.c_double(), C.c_double()
hd.ck(hd.lib.esp_debug_last_sum_ms(hd.h, C.byref(f_ms), C.byref(c_ms)))
print("last call: folds %.2f ms, gather + combine flush %.2f ms" % (f_ms.value, c_ms.value))
print("hook %s" % os.environ["ESP_SUM_HOOK"] if os.environ.get("ESP_SUM_HOOK") else "product", "p %d x %d entries: esp_flush_sum %.2f ms (min of %s)" % (p, cnt, min(ts[1:]) * 1e3, [round(x * 1e3, 2) for x in ts]), "nnz", z.value)
