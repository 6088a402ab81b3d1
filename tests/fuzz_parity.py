#!/usr/bin/env python3
"""Randomised parity fuzz (GPU): random shapes (incl. very tall matrices: many row bits), kinds, duplicate
patterns, stream orders (sorted / clustered / shuffled), multi-flush sequences, both flush modes --
every result compared bit for bit with the CPU oracle.  usage: tests/fuzz_parity.py [seconds] [seed]   (test infrastructure: it runs the CPU oracle)
ESP_FUZZ_FOCUS=k32: shapes and batches that reach the 4-byte keys / UPDATE-only fold of the bucket kernel;
ESP_FUZZ_FOCUS=elements: only the element-level append (esp_append_elements) and Base.sum (esp_flush_sum) cases;
ESP_FUZZ_FOCUS=sum: only Base.sum over buffers of per-entry calls (esp_flush_sum's general path)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

torch.cuda.init()
from esparse_loader import load  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from refmodel import assert_csc_equal  # noqa: E402

esp = load()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
t_end = time.time() + budget
cases = 0
paths = {}
def generator_case():
    """Device-side producers: the producer-side partition (append = partition), its fall-backs and what follows it --
    a second producer call, a host append, a flush over the stored pattern -- on grids whose lines and planes do not line
    up with the 256-node chunks."""
    shape = rng.choice(["cube", "slab", "line", "plane"])
    if shape == "cube":
        nx, ny, nz = (int(rng.integers(20, 90)) for _ in range(3))
    elif shape == "slab":
        nx, ny, nz = int(rng.integers(2, 9)), int(rng.integers(100, 600)), int(rng.integers(100, 600))
    elif shape == "line":
        nx, ny, nz = int(rng.integers(300000, 2000000)), 1, 1
    else:
        nx, ny, nz = int(rng.integers(300, 1500)), int(rng.integers(300, 1500)), 1
    N = nx * ny * nz
    if N > 3000000:
        return None
    force = int(rng.choice([0, 0, 0, 0, 14, 15, 16, 18, 13, 4, 19, 31, 40]))
    kind = int(rng.choice([1, 1, 2]))
    mode = int(rng.choice([0, 1, 2]))
    A = esp.ExtendableSparseMatrix(N, N)
    A.debug_force_path(force)
    O = orc.ExtendableSparseMatrix(N, N)
    nrounds = int(rng.integers(1, 4))
    for rnd in range(nrounds):
        seed = int(rng.integers(1, 1 << 30))
        I, J, V = orc.fdrand_stream(nx, ny, nz, rand_mode=mode, seed=seed)
        kk = np.full(len(I), kind, np.uint8)
        how = rng.choice(["whole", "two_calls", "host_after", "host_before", "dev_after"])
        if how == "whole":
            A.generate_fdrand(nx, ny, nz, seed=seed, rand_mode=mode, kind=kind)
            O.apply(kk, I, J, V)
        elif how == "two_calls":
            cut_node = int(rng.integers(1, N)) if N > 1 else 0
            A.generate_fdrand_range(nx, ny, nz, 0, cut_node, seed=seed, rand_mode=mode, kind=kind)
            A.generate_fdrand_range(nx, ny, nz, cut_node, N, seed=seed, rand_mode=mode, kind=kind)
            O.apply(kk, I, J, V)
        elif how == "dev_after":
            # triplets resident on the device behind the producer's batch, one kind: over a stored pattern a pre-sorted tail
            # is partitioned as it is appended (partition.hip, append_tail_partitioned)
            cnt = int(rng.choice([5000, 200000, 1500000]))
            Ih, Jh, Vh = rng.integers(1, N + 1, cnt), rng.integers(1, N + 1, cnt), rng.standard_normal(cnt)
            if rng.random() < 0.8:
                Jh = np.sort(Jh)
            kd = int(rng.choice([0, 1, 2]))
            A.generate_fdrand(nx, ny, nz, seed=seed, rand_mode=mode, kind=kind)
            O.apply(kk, I, J, V)
            dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()   # noqa: E731
            sub = rng.random() < 0.3
            A.append_device(kd, dev(Ih), dev(Jh), dev(Vh), op="-" if sub else "+")
            O.apply(np.full(cnt, kd, np.uint8), Ih, Jh, -Vh if (sub and kd != 0) else Vh)
        else:
            cnt = int(rng.choice([1, 7, 5000, 200000]))
            Ih, Jh, Vh = rng.integers(1, N + 1, cnt), rng.integers(1, N + 1, cnt), rng.standard_normal(cnt)
            if rng.random() < 0.5:                       # (a tail behind the producer's batch: pre-sorted or not)
                Jh = np.sort(Jh)
            kh = rng.integers(0, 3, cnt).astype(np.uint8)
            if how == "host_before":
                A.append(0, Ih, Jh, Vh, kinds=kh)
                O.apply(kh, Ih, Jh, Vh)
            A.generate_fdrand(nx, ny, nz, seed=seed, rand_mode=mode, kind=kind)
            O.apply(kk, I, J, V)
            if how == "host_after":
                A.append(0, Ih, Jh, Vh, kinds=kh)
                O.apply(kh, Ih, Jh, Vh)
        t_case = time.time()
        try:
            A.flush()
            if os.environ.get("ESP_FUZZ_VERBOSE"):
                print("gen", dict(nx=nx, ny=ny, nz=nz, force=force, kind=kind, mode=mode, rnd=rnd, how=str(how)), "flush %.2f s" % (time.time() - t_case),
                      "partition", A.debug_last_partition(), flush=True)
        except Exception:
            print("FLUSH FAILED generator case", dict(nx=nx, ny=ny, nz=nz, force=force, kind=kind, mode=mode, rnd=rnd, how=str(how), seed=seed))
            raise
        O.flush()
        key = ("gen", A.debug_last_partition(), A.debug_last_key_bytes(), A.debug_last_local_small())
        paths[key] = paths.get(key, 0) + 1
        try:
            assert_csc_equal(A.sparse().arrays(), O.arrays())
        except AssertionError:
            print("MISMATCH seed", seed, "generator case", dict(nx=nx, ny=ny, nz=nz, force=force, kind=kind, mode=mode, rnd=rnd, how=str(how)))
            raise
    return True


def fem_case():
    """The P1 FEM producer (test/femtools.jl:45-72) on random meshes, natural or shuffled cell order: item partition +
    expansion, the group tier of the bucket kernel in all its kernels (group3_k with three workgroups per CU on a fresh
    matrix; local_k's group tier over a stored pattern, with force_path 30, or when a run is too long), further appends
    behind the batch, re-assembly."""
    dim = int(rng.choice([2, 2, 3]))
    npd = int(rng.integers(6, 400)) if dim == 2 else int(rng.integers(4, 42))
    nn = npd ** dim
    order = int(rng.choice([0, 1, 1]))
    force = int(rng.choice([0, 0, 0, 0, 0, 30, 24, 25, 28, 14, 26, 32, 39, 36]))
    A = esp.ExtendableSparseMatrix(nn, nn)
    A.debug_force_path(force)
    O = orc.ExtendableSparseMatrix(nn, nn)
    for rnd in range(int(rng.integers(1, 4))):
        seed = int(rng.integers(1, 1 << 30))
        I, J, V = orc.fem_stream(dim, npd, seed=seed, order_mode=order)
        A.generate_fem(dim, npd, seed=seed, order_mode=order)
        O.apply(np.full(len(I), 2, np.uint8), I, J, V)
        if rng.random() < 0.4:                               # entries behind the batch: a few, or many in a few columns
            cnt = int(rng.choice([3, 500, 40000]))
            cols = rng.integers(1, nn + 1, cnt) if rng.random() < 0.6 else rng.choice(rng.integers(1, nn + 1, 3), cnt)
            Ih, Jh, Vh = rng.integers(1, nn + 1, cnt), np.sort(cols), rng.standard_normal(cnt)
            kh = rng.integers(0, 3, cnt).astype(np.uint8)
            A.append(0, Ih, Jh, Vh, kinds=kh)
            O.apply(kh, Ih, Jh, Vh)
        try:
            A.flush()
        except Exception:
            print("FLUSH FAILED fem case", dict(dim=dim, npd=npd, order=order, force=force, rnd=rnd, seed=seed))
            raise
        O.flush()
        key = ("fem", A.debug_last_partition(), A.debug_last_key_bytes(), A.debug_last_local_small())
        paths[key] = paths.get(key, 0) + 1
        try:
            assert_csc_equal(A.sparse().arrays(), O.arrays())
        except AssertionError:
            print("MISMATCH fem case", dict(dim=dim, npd=npd, order=order, force=force, rnd=rnd, seed=seed))
            raise
    return True


def elements_case():
    """esp_append_elements[_host] on random meshes: cells of 1 .. 16 nodes drawn from a neighbourhood (so that columns get
    runs of all lengths) under a random node renumbering, random element matrices with zeros, kinds UPDATE / RAWUPDATE / SET,
    with and without the diagonal term, `-`, now and then a cell that names a node twice (stream-order fall-back), host or
    device arrays, forced paths, further appends around the batch, re-assembly over the stored pattern; and the same cells
    dealt to several partition buffers that meet in ONE esp_flush_sum (Base.sum: sparsematrixdilnkc.jl:397-435)."""
    nloc = int(rng.choice([1, 2, 3, 3, 4, 4, 6, 10, 16]))
    n = int(rng.choice([500, 20000, 300000, 2000000]))
    m = n if rng.random() < 0.8 else n + int(rng.integers(1, 1000))
    lim = min(m, n)
    nc = int(rng.choice([1, 50, 3000, 60000, 400000]))
    if nc * nloc * (nloc + 1) > 12000000:
        nc = 12000000 // (nloc * (nloc + 1))
    span = int(rng.choice([nloc, 2 * nloc + 3, 64, 500]))
    span = max(nloc, min(span, lim))
    start = rng.integers(0, lim - span + 1, nc)
    local = np.argsort(rng.random((nc, span)), axis=1)[:, :nloc]          # nloc distinct offsets per cell
    perm = rng.permutation(lim) + 1 if rng.random() < 0.5 else np.arange(1, lim + 1)
    cn = np.asfortranarray(perm[(start[:, None] + local)].T.astype(np.int64))
    if rng.random() < 0.15 and nloc > 1 and nc > 0:
        c = int(rng.integers(0, nc))
        cn[1, c] = cn[0, c]                                                # a cell that names a node twice
    em = rng.standard_normal((nloc, nloc, nc))
    em[rng.random(em.shape) < 0.15] = 0.0
    em = np.asfortranarray(em)
    dg = np.asfortranarray(rng.standard_normal((nloc, nc))) if rng.random() < 0.6 else None
    kind = int(rng.choice([1, 2, 2, 0]))
    sub = rng.random() < 0.2
    force = int(rng.choice([0, 0, 0, 0, 0, 14, 25, 30, 24, 2, 32, 39, 37]))
    I, J, V = orc.elements_stream(cn, em, dg)
    Vs = -V if (sub and kind != 0) else V
    if rng.random() < 0.3 and nc >= 4:
        # ---- Base.sum over p partition buffers
        p = int(rng.choice([2, 3, 7]))
        cuts = np.sort(rng.integers(0, nc + 1, p - 1)).tolist()
        cuts = [0] + cuts + [nc]
        xs = [esp.SparseMatrixHIPCOO(m, n) for _ in range(p)]
        home = esp.SparseMatrixHIPCOO(m, n)
        csc = esp.SparseMatrixCSC(m, n)
        Oc = orc.CSC(m, n)
        for rnd in range(2):
            for t in range(p):
                a, b = cuts[t], cuts[t + 1]
                if b > a:
                    xs[t].append_elements(cn[:, a:b], em[:, :, a:b], None if dg is None else dg[:, a:b], kind=kind, op="-" if sub else "+")
                L = orc.SparseMatrixLNK(m, n)
                It, Jt, Vt = orc.elements_stream(cn[:, a:b], em[:, :, a:b], None if dg is None else dg[:, a:b]) if b > a else ([], [], [])
                for i, j, v in zip(np.asarray(It).tolist(), np.asarray(Jt).tolist(), (np.asarray(Vt) * (-1.0 if (sub and kind != 0) else 1.0)).tolist()):
                    if kind == 0:
                        L[i, j] = v
                    elif kind == 1:
                        L.updateindex(orc.OP_ADD, v, i, j)
                    else:
                        L.rawupdateindex(orc.OP_ADD, v, i, j)
                if L.nnz() > 0:
                    Oc = L + Oc
            csc = esp.SparseMatrixHIPCOO.sum(xs, csc, home=home)
            try:
                assert_csc_equal(csc.arrays(), Oc.arrays())
            except AssertionError:
                lz = __import__("ctypes").c_int32()
                home._d.lib.esp_debug_last_lazy_items(home._d.h, __import__("ctypes").byref(lz))
                print("MISMATCH elements/sum case", dict(nloc=nloc, m=m, n=n, nc=nc, span=span, kind=kind, p=p, rnd=rnd, sub=sub, diag=dg is not None,
                                                         lazy=lz.value, case=cases))
                raise
        paths[("elem_sum", p)] = paths.get(("elem_sum", p), 0) + 1
        return True
    if len(I) > 3000000 and nloc * nloc > 40:
        return None                                                       # (the oracle's list walks: keep the case short)
    A = esp.ExtendableSparseMatrix(m, n)
    A.debug_force_path(force)
    O = orc.ExtendableSparseMatrix(m, n)
    for rnd in range(int(rng.integers(1, 3))):
        how = rng.choice(["alone", "alone", "host_before", "host_after", "twice"])
        cnt = int(rng.choice([5, 3000]))
        Ih, Jh, Vh = rng.integers(1, m + 1, cnt), rng.integers(1, n + 1, cnt), rng.standard_normal(cnt)
        kh = rng.integers(0, 3, cnt).astype(np.uint8)
        if how == "host_before":
            A.append(0, Ih, Jh, Vh, kinds=kh)
            O.apply(kh, Ih, Jh, Vh)
        reps = 2 if how == "twice" else 1
        for _ in range(reps):
            if rng.random() < 0.5 or nc == 0:
                A.append_elements(cn, em, dg, kind=kind, op="-" if sub else "+")
            else:
                tn = torch.from_numpy(np.ascontiguousarray(cn.T)).cuda()
                te = torch.from_numpy(np.ascontiguousarray(em.transpose(2, 1, 0))).cuda()
                td = None if dg is None else torch.from_numpy(np.ascontiguousarray(dg.T)).cuda()
                A.append_elements(tn, te, td, kind=kind, op="-" if sub else "+")
            O.apply(np.full(len(I), kind, np.uint8), I, J, Vs)
        if how == "host_after":
            A.append(0, Ih, Jh, Vh, kinds=kh)
            O.apply(kh, Ih, Jh, Vh)
        try:
            A.flush()
        except Exception:
            print("FLUSH FAILED elements case", dict(nloc=nloc, m=m, n=n, nc=nc, span=span, kind=kind, force=force, rnd=rnd, how=str(how)))
            raise
        O.flush()
        key = ("elem", nloc, A.debug_last_partition(), A.debug_last_key_bytes(), A.debug_last_local_small())
        paths[key] = paths.get(key, 0) + 1
        try:
            assert_csc_equal(A.sparse().arrays(), O.arrays())
        except AssertionError:
            print("MISMATCH elements case", dict(nloc=nloc, m=m, n=n, nc=nc, span=span, kind=kind, force=force, rnd=rnd, how=str(how), diag=dg is not None,
                                                 sub=sub, lazy=A.debug_last_lazy_items(), small=A.debug_last_local_small(), part=A.debug_last_partition(),
                                                 rebuild=A.debug_last_rebuild(), case=cases))
            raise
    return True


def sum_case():
    """Base.sum over p buffers of per-entry calls (sparsematrixdilnkc.jl:397-435; esp_flush_sum's general path: the folds as ONE
    flush of a scratch matrix, round 6): buffers in bands of columns that overlap / spread over the whole matrix / one of them
    empty / tall matrices, SET / UPDATE / RAWUPDATE, repeated positions, zeros; two rounds onto the same stored matrix, the
    second with hits; now and then a test hook on the destination (the one-by-one form)."""
    import ctypes as C
    p = int(rng.choice([2, 3, 5, 9]))
    n = int(rng.choice([40, 3000, 250000, 3000000]))
    m = int(rng.choice([7, 900, 10 ** 6, 2 ** 33]))
    home = esp.SparseMatrixHIPCOO(m, n)
    one_by_one = rng.random() < 0.2
    if one_by_one:
        home._d.ck(home._d.lib.esp_debug_force_path(home._d.h, 31))
    csc = esp.SparseMatrixCSC(m, n)
    Oc = orc.CSC(m, n)
    shape = str(rng.choice(["bands", "spread", "same"]))
    for rnd in range(2):
        xs = [esp.SparseMatrixHIPCOO(m, n) for _ in range(p)]
        nonempty = 0
        for t in range(p):
            cnt = int(rng.choice([0, 3, 400, 20000, 120000]))
            L = orc.SparseMatrixLNK(m, n)
            if cnt:
                nonempty += 1
                if shape == "bands":
                    w = max(1, n // p)
                    lo = max(1, 1 + t * w - int(rng.integers(0, w // 2 + 1)))
                    hi = min(n, (t + 1) * w + int(rng.integers(0, w // 2 + 1)))
                elif shape == "same":
                    lo, hi = max(1, n // 3), max(1, n // 3) + min(n - max(1, n // 3), 50)
                else:
                    lo, hi = 1, n
                J = rng.integers(lo, hi + 1, cnt)
                if rng.random() < 0.6:
                    J = np.sort(J)
                I = np.minimum(m, 1 + (J * 7 + rng.integers(0, 6, cnt)) % min(m, 10 ** 6))
                V = np.where(rng.random(cnt) < 0.05, 0.0, rng.standard_normal(cnt))
                K = rng.choice(np.array([0, 1, 2], np.uint8), cnt, p=[0.1, 0.6, 0.3])
                xs[t].append(0, I, J, V, kinds=K)
                for k, i, j, v in zip(K.tolist(), I.tolist(), J.tolist(), V.tolist()):
                    if k == 0:
                        L[i, j] = v
                    elif k == 1:
                        L.updateindex(orc.OP_ADD, v, i, j)
                    else:
                        L.rawupdateindex(orc.OP_ADD, v, i, j)
            if L.nnz() > 0:
                Oc = L + Oc
        csc = esp.SparseMatrixHIPCOO.sum(xs, csc, home=home)
        flag = C.c_int32(-1)
        home._d.lib.esp_debug_last_sum_batched(home._d.h, C.byref(flag))
        try:
            assert_csc_equal(csc.arrays(), Oc.arrays())
        except AssertionError:
            print("MISMATCH sum case", dict(p=p, m=m, n=n, shape=shape, rnd=rnd, one_by_one=one_by_one, batched=flag.value, case=cases))
            raise
        key = ("sum", shape, flag.value if nonempty else -1)
        paths[key] = paths.get(key, 0) + 1
    return True


max_cases = int(os.environ.get("ESP_FUZZ_MAXCASES", "0"))
while time.time() < t_end and (max_cases == 0 or cases < max_cases):
    if (rng.random() < 0.08 or os.environ.get("ESP_FUZZ_FOCUS") == "sum") and os.environ.get("ESP_FUZZ_FOCUS") not in ("k32", "elements"):
        if sum_case():
            cases += 1
        continue
    if (rng.random() < 0.12 or os.environ.get("ESP_FUZZ_FOCUS") == "elements") and os.environ.get("ESP_FUZZ_FOCUS") != "k32":
        if elements_case():
            cases += 1
        continue
    if rng.random() < 0.12 and os.environ.get("ESP_FUZZ_FOCUS") != "k32":
        if fem_case():
            cases += 1
        continue
    if rng.random() < 0.35 and os.environ.get("ESP_FUZZ_FOCUS") != "k32":
        if generator_case():
            cases += 1
        continue
    # (plan hook: many prefix bits at small sizes -- the 9-bit partition passes, three-pass plans)
    plan_cap = int(rng.choice([12, 40, 300])) if rng.random() < 0.3 else 0
    n = int(rng.choice([1, 3, 64, 257, 5000, 70000, 400000]))
    m = int(rng.choice([1, 2, 100, 4097, 10 ** 6, 2 ** 31 - 1, 2 ** 33, 2 ** 40]))
    focus = os.environ.get("ESP_FUZZ_FOCUS") == "k32"   # few row bits, many columns: <= 32 key bits below the prefix
    if focus:
        n = int(rng.choice([70000, 400000, 3000000]))
        m = int(rng.choice([3, 100, 4097, 60000]))
    A = esp.ExtendableSparseMatrix(m, n)
    A.debug_plan_cap(plan_cap)
    O = orc.ExtendableSparseMatrix(m, n)
    force = int(rng.choice([0, 0, 0, 2, 3, 4, 5, 12, 13, 14, 15, 17, 18, 23]))
    if focus:
        force = int(rng.choice([0, 0, 13, 15, 4, 18]))
    A.debug_force_path(force)
    nflush = int(rng.integers(1, 4))
    for f in range(nflush):
        big = rng.random() < (0.8 if focus else 0.3)     # a batch large enough for the run-based partition, often pre-sorted
        cnt = 1500000 if big else int(rng.choice([0, 1, 50, 5000, 200000, 1500000]))
        per_col = float(rng.choice([0.5, 5, 14, 22, 40, 300]))
        ncols_used = max(1, min(n, int(cnt / per_col) + 1))
        if cnt / ncols_used > 20000:                    # (the oracle's column lists are walked per insert: keep them short)
            cnt = 20000 * ncols_used
        cols = rng.integers(1, n + 1, ncols_used)
        J = cols[rng.integers(0, ncols_used, cnt)]
        nrows_used = max(1, int(rng.choice([1, 3, 50, 10 ** 4, 10 ** 9])))
        rowpool = rng.integers(1, m + 1, min(nrows_used, 10 ** 6))
        I = rowpool[rng.integers(0, len(rowpool), cnt)]
        order = rng.choice(["sorted", "sorted", "clustered"]) if big else rng.choice(["asis", "sorted", "clustered"])
        if order == "sorted":
            o = np.argsort(J, kind="stable")
            I, J = I[o], J[o]
        elif order == "clustered":
            o = np.argsort(J // max(1, n // 64), kind="stable")
            I, J = I[o], J[o]
        kinds = rng.choice(np.array([0, 1, 1, 2], np.uint8), cnt)
        V = np.where(rng.random(cnt) < 0.2, 0.0, rng.standard_normal(cnt))
        V[rng.random(cnt) < 0.02] = -0.0
        # per-entry kinds, or one / two appends with a single kind each (the batch bookkeeping behind the 4-byte keys
        # and the UPDATE-only fold of the bucket kernel)
        style = rng.choice(["one_kind", "one_kind", "two_appends"]) if focus else rng.choice(["kinds", "one_kind", "two_appends"])
        if style == "kinds" or cnt < 2:
            A.append(0, I, J, V, kinds=kinds)
            O.apply(kinds, I, J, V)
        else:
            cut = cnt if style == "one_kind" else cnt // 2
            k1 = int(rng.choice([0, 1, 1, 1, 2]))
            k2 = k1 if rng.random() < 0.5 else int(rng.choice([0, 1, 2]))
            for lo_, hi_, kd in ((0, cut, k1), (cut, cnt, k2)):
                if hi_ > lo_:
                    A.append(kd, I[lo_:hi_], J[lo_:hi_], V[lo_:hi_])
                    O.apply(np.full(hi_ - lo_, kd, np.uint8), I[lo_:hi_], J[lo_:hi_], V[lo_:hi_])
        try:
            A.flush()
        except Exception:
            print("FLUSH FAILED seed", seed, "case", cases, dict(m=m, n=n, force=force, plan_cap=plan_cap, flush=f, cnt=cnt, per_col=per_col, order=str(order),
                                                                style=str(style), nnz=A._d.nnz()))
            raise
        O.flush()
        key = (A.debug_last_path(), A.debug_last_partition(), A.debug_last_key_bytes(), int(A.debug_last_fold_update()),
               A.debug_last_local_small())
        paths[key] = paths.get(key, 0) + 1
        try:
            assert_csc_equal(A.sparse().arrays(), O.arrays())
        except AssertionError:
            print("MISMATCH seed", seed, "case", cases, dict(m=m, n=n, force=force, flush=f, cnt=cnt, per_col=per_col, order=str(order)))
            raise
    cases += 1
print("fuzz ok: cases", cases, "paths (pipeline, partition, key bytes, update-only fold, small bucket kernel | gen, partition, key bytes, small):", paths)
