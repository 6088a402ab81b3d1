"""cfg_mt_sum's plug-in form step by step (bench.py: plugin_fresh_ms / plugin_same_pattern_ms): fills, upload, esp_flush_sum,
download -- twice over the same pattern."""
import ctypes as C, sys, time
sys.path.insert(0, ".")
import torch
from esparse_loader import load
esp = load()
import extendablesparse_jl_amd.matrix as M
dim, npd, p = 2, 3163, 16
nn, nloc = npd ** dim, dim + 1
nc = 2 * (npd - 1) ** 2
home = esp.SparseMatrixHIPCOO(nn, nn)
xs = [esp.SparseMatrixHIPCOO(nn, nn) for _ in range(p)]
cn = torch.empty((nc, nloc), dtype=torch.int64, device="cuda")
em = torch.empty((nc, nloc, nloc), dtype=torch.float64, device="cuda")
dg = torch.empty((nc, nloc), dtype=torch.float64, device="cuda")
A0 = esp.ExtendableSparseMatrix(nn, nn)
A0.generate_fem_mesh(dim, npd, cn, em, dg, seed=0x5EED0004, order_mode=0)
A0.synchronize()
cuts = [nc * t // p for t in range(p + 1)]
fill = lambda t: xs[t].append_elements(cn[cuts[t]:cuts[t + 1]], em[cuts[t]:cuts[t + 1]], dg[cuts[t]:cuts[t + 1]])   # noqa: E731
csc = esp.SparseMatrixCSC(nn, nn)
hd = home._d
for it in range(4):
    for t in range(p):
        fill(t)
    for x in xs:
        x._d.commit()
    hd.ck(hd.lib.esp_synchronize(hd.h))
    t0 = time.perf_counter()
    M._upload_csc(hd, home._mirror, csc)
    t1 = time.perf_counter()
    arr = (C.c_void_p * p)(*[x._d.h for x in xs])
    z, ch = C.c_int64(), C.c_int32()
    hd.ck(hd.lib.esp_flush_sum(hd.h, arr, p, C.byref(z), C.byref(ch)))
    hd.ck(hd.lib.esp_synchronize(hd.h))
    t2 = time.perf_counter()
    csc, home._mirror = M._download_csc(hd, csc, bool(ch.value))
    t3 = time.perf_counter()
    print("round %d: upload %.1f ms, esp_flush_sum %.1f ms (pattern changed %d, nnz %d), download %.1f ms; sum path %d" %
          (it, (t1 - t0) * 1e3, (t2 - t1) * 1e3, ch.value, z.value, (t3 - t2) * 1e3, home.debug_last_lazy_items() if hasattr(home, "debug_last_lazy_items") else -1), flush=True)
