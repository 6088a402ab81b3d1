"""322^3 / 400^3 fresh stencil assemblies with the default plan (more than 32 key bits below the prefix: packed keys) and with a
plan that asks for one / two more prefix bits (esp_debug_plan_cap: 4-byte keys, smaller segments)."""
import sys, time
sys.path.insert(0, ".")
from esparse_loader import load
esp = load()
for n in (256, 322, 400):
    N = n ** 3
    for cap in (0, 2048, 1024):
        A = esp.ExtendableSparseMatrix(N, N)
        if cap:
            A.debug_plan_cap(cap)
        dts = []
        for it in range(8):
            A.synchronize()
            t0 = time.perf_counter()
            A.reset()
            A.generate_fdrand(n, n, n, seed=0x5EED0002, rand_mode=1)
            A.flush()
            A.synchronize()
            if it >= 2:
                dts.append(time.perf_counter() - t0)
        print("n=%d cap=%d ms %.3f key_bytes %d partition %d small %s nnz %d" % (n, cap, 1e3 * sum(dts) / len(dts), A.debug_last_key_bytes(),
              A.debug_last_partition(), A.debug_last_local_small(), A.nnz()), flush=True)
        del A
