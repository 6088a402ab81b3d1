"""round 6 probe: the plug-in form's download step by step inside the plug-in trace's process (16 buffers, torch arrays)"""
import ctypes as C, sys, time, os
sys.path.insert(0, ".")
import torch
import numpy as np
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
vp = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731


def huge_kb():
    tot = 0
    with open("/proc/self/smaps") as f:
        for ln in f:
            if ln.startswith("AnonHugePages:"):
                tot += int(ln.split()[1])
    return tot


for it in range(4):
    for t in range(p):
        fill(t)
    hd.ck(hd.lib.esp_synchronize(hd.h))
    M._upload_csc(hd, home._mirror, csc)
    arr = (C.c_void_p * p)(*[x._d.h for x in xs])
    z, ch = C.c_int64(), C.c_int32()
    hd.ck(hd.lib.esp_flush_sum(hd.h, arr, p, C.byref(z), C.byref(ch)))
    hd.ck(hd.lib.esp_synchronize(hd.h))
    if ch.value:
        csc, home._mirror = M._download_csc(hd, csc, True)
        continue
    h0 = huge_kb()
    t0 = time.perf_counter()
    nz = np.empty(csc.nnz(), np.float64)
    t1 = time.perf_counter()
    hd.ck(hd.lib.esp_get_nzval(hd.h, vp(nz)))
    t2 = time.perf_counter()
    hd.ck(hd.lib.esp_get_nzval(hd.h, vp(nz)))
    t3 = time.perf_counter()
    print("round %d: np.empty %.2f ms, get_nzval fresh %.1f ms, again %.1f ms, address %% 2MiB %d, AnonHugePages %d -> %d kB" %
          (it, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, nz.ctypes.data % (2 << 20), h0, huge_kb()), flush=True)
    t4 = time.perf_counter()
    csc = esp.SparseMatrixCSC(nn, nn, csc.colptr, csc.rowval, nz)   # (the previous values vector dies here)
    t5 = time.perf_counter()
    print("         rebinding csc (frees the previous 560 MB vector): %.1f ms" % ((t5 - t4) * 1e3), flush=True)
    home._mirror = (csc.colptr, csc.rowval)
