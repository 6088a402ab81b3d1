"""Distinct handles driven side by side -- the contract of the reference's multi-threaded assembly: one buffer per task, the tasks
run concurrently (src/matrix/genericmtextendablesparsematrixcsc.jl:87-99; test/femtools.jl:88-107 `@tasks for part`), and of
include/esparse_hip.h ("distinct handles are independent and may be driven from different host threads").

Until round 6 this did not hold: two handles' flushes running beside one another corrupted one of them (NOTES/round6.md
section 1: the radix tier of the bucket kernel kept two words per wave in the LDS array that holds the prefetched segment bounds,
without a workgroup barrier in between).  Two checks:
  * tests/stress_handles.hip, a native host (hipcc): fixed workloads, each handle's result alone against its results while the
    others run -- the shapes of the fuzz case that found the race among them (three uneven parts of a ten-node mesh);
  * Python host threads (ctypes drops the GIL inside every call) filling AND flushing their own handles, all flushes starting
    together, every result bit for bit the CPU oracle's."""
import os
import subprocess
import threading

import numpy as np
import pytest

from refmodel import assert_csc_equal

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "extendablesparse.jl_amd")
EXE = os.path.join(ROOT, "tests", "stress_handles.bin")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def build_stress():
    import importlib.util
    spec = importlib.util.spec_from_file_location("esparse_build", os.path.join(PKG, "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.build()
    src = os.path.join(ROOT, "tests", "stress_handles.hip")
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < max(os.path.getmtime(src), os.path.getmtime(os.path.join(ROOT, "include", "esparse_hip.h"))):
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O2", "-std=c++17", "-o", EXE, src, "-L", PKG, "-l:libesparse_hip.so",
                               "-Wl,-rpath," + PKG, "-lpthread"])
    return EXE


def test_stress_host_compiles():
    assert os.path.exists(build_stress())


STRESS_RUNS = [
    # the shapes that exposed the race: a 4-byte-key flush with direct colptr beside a packed-key flush through the radix tier
    ["--work", "parts", "--kind", "2", "--mode", "spawn", "--handles", "3", "--threads", "3", "--iters", "60"],
    ["--work", "parts", "--kind", "2", "--mode", "lockstep", "--handles", "6", "--threads", "6", "--iters", "40"],
    # every workload kind at once, fills and flushes from five threads, fresh handles every iteration (allocation churn)
    ["--work", "mix", "--mode", "threads", "--handles", "5", "--threads", "5", "--iters", "40", "--fresh", "1"],
    ["--work", "mix", "--mode", "lockstep", "--handles", "5", "--threads", "5", "--iters", "40"],
    ["--work", "fem4", "--mode", "lockstep", "--handles", "4", "--threads", "4", "--iters", "30"],
    ["--work", "trip", "--mode", "spawn", "--handles", "4", "--threads", "4", "--iters", "30"],
]


@pytest.mark.gpu
@pytest.mark.parametrize("args", STRESS_RUNS, ids=lambda a: "-".join(a[1:6:2]))
def test_handles_side_by_side_native(args):
    exe = build_stress()
    r = subprocess.run([exe, "--quiet"] + args, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "stress_handles: ok" in r.stdout, (r.stdout[-3000:], r.stderr[-3000:])


def _ten_node_mesh(rng, n, nc, nloc=10, span=10):
    start = rng.integers(0, n - span + 1, nc)
    local = np.argsort(rng.random((nc, span)), axis=1)[:, :nloc]
    perm = rng.permutation(n) + 1
    cn = np.asfortranarray(perm[(start[:, None] + local)].T.astype(np.int64))
    em = np.asfortranarray(rng.standard_normal((nloc, nloc, nc)))
    dg = np.asfortranarray(rng.standard_normal((nloc, nc)))
    return cn, em, dg


@pytest.mark.gpu
def test_threads_fill_and_flush_against_the_oracle(esp, orc):
    """Six host threads, one handle each, 40 rounds of reset! -> fill -> (barrier) -> flush! -> read: 240 flushes that start together,
    each compared with the oracle's CSC of the same stream.  Workloads: the three uneven parts of ONE ten-node mesh (the middle one
    takes packed keys, the radix tier and the colptr scan; the others 4-byte keys and the direct colptr), the stencil generator,
    shuffled triplets with duplicates and mixed kinds."""
    rng = np.random.default_rng(503)
    n = 2000000
    cn, em, dg = _ten_node_mesh(rng, n, 60000)
    cuts = [0, 24513, 31150, 60000]
    works = []
    for t in range(3):
        a, b = cuts[t], cuts[t + 1]
        c, e, d = np.asfortranarray(cn[:, a:b]), np.asfortranarray(em[:, :, a:b]), np.asfortranarray(dg[:, a:b])
        I, J, V = orc.elements_stream(c, e, d)
        O = orc.ExtendableSparseMatrix(n, n)
        O.apply(np.full(len(I), 2, np.uint8), I, J, V)
        O.flush()
        works.append((n, lambda A, c=c, e=e, d=d: A.append_elements(c, e, d, kind=esp.ESP_RAWUPDATE), O.arrays()))
    nx = 44
    O = orc.fdrand(nx, nx, nx, rand_mode=1, seed=77, style=orc.KIND_UPDATE)
    works.append((nx ** 3, lambda A: A.generate_fdrand(nx, nx, nx, seed=77, rand_mode=1, kind=esp.ESP_UPDATE), O.arrays()))
    for seed in (5, 6):
        r2 = np.random.default_rng(seed)
        nn, cnt = 300000, 1500000
        J = r2.integers(1, nn + 1, cnt)
        I = np.clip(J + r2.integers(-20, 21, cnt), 1, nn)
        V = r2.standard_normal(cnt)
        K = r2.integers(0, 3, cnt).astype(np.uint8)
        O = orc.ExtendableSparseMatrix(nn, nn)
        O.apply(K, I, J, V)
        O.flush()
        works.append((nn, lambda A, I=I, J=J, V=V, K=K: A.append(0, I, J, V, kinds=K), O.arrays()))
    T, rounds = len(works), 40
    bar = threading.Barrier(T)
    errors = []

    def run(t):
        try:
            nn, fill, want = works[t]
            A = esp.ExtendableSparseMatrix(nn, nn)
            for rnd in range(rounds):
                A.reset()
                fill(A)
                bar.wait(timeout=300)
                A.flush()
                assert_csc_equal(A.arrays(), want, "thread %d round %d" % (t, rnd))
        except BaseException as ex:  # noqa: BLE001 (reported by the main thread)
            errors.append((t, repr(ex)[:400]))
            bar.abort()

    th = [threading.Thread(target=run, args=(t,)) for t in range(T)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errors, errors
