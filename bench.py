#!/usr/bin/env python3
"""bench.py -- assembled+flushed nnz/s on the 256^3 7-point stencil (BASELINE.json config 2).

A "step" is one fresh assembly of fdrand(Float64,n,n,n; matrixtype=ExtendableSparseMatrix)
(src/matrix/sprand.jl:226-256): the update stream is produced on the device straight into the COO
append buffer (inputs resident in HBM, no PCIe in the timed region), then flush! builds the CSC.
value = nnz of the resulting CSC x steps x ranks / wall time (max over ranks).

`--gpus N` (N > 1): one process per GPU.  Under `python -m torch.distributed.run` the ranks exist already
(RANK / LOCAL_RANK / WORLD_SIZE in the environment); started plainly, this process only LAUNCHES N rank
processes (it never touches the GPU itself), relays rank 0's JSON line and fails loudly when the node has
fewer than N GPUs.  Weak scaling: the global grid is n x n x (n*N), rank r assembles its z-slab, columns are
range-sharded, the entries of the cross-slab pairs travel through an RCCL all-to-all-v.

Prints ONE JSON line (rank 0).  Extra objects: "roofline" (dominant kernel, hipEvent-timed on the
library's stream), "pipeline" (whole step against the compulsory bytes of SURVEY.md section 8d),
"cpu_baseline" (the C oracle timed on this box's host, bounded sample), "extra.configs" (BASELINE.json
configs 3 and 4, three steps each, measured after the headline loop).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def pmc_traffic(kernel_stage):
    """HBM bytes per launch of the stage's kernel from the committed rocprofv3 PMC passes
    (profiles/pmc_traffic.json, produced by tools/pmc_traffic.py from separate FETCH_SIZE and
    WRITE_SIZE runs of this same command, gfx950 corrections applied) and the name of the profile round
    they come from; (None, None) if not collected."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            d = json.load(f)
        return d.get(kernel_stage, {}).get("hbm_bytes_per_launch"), d.get("_meta", {}).get("round")
    except Exception:
        return None, None


def golden_digests():
    """tests/golden/digests_large.txt: sha256 of the ORACLE's CSC for the bench configurations (data only; made by
    tests/golden/make_digests_large.py on the build host)."""
    out = {}
    try:
        with open(os.path.join(ROOT, "tests", "golden", "digests_large.txt")) as f:
            for line in f:
                parts = line.split()
                if parts:
                    out[parts[0]] = dict(p.split("=") for p in parts[1:])
    except Exception:
        pass
    return out


def csc_digest_ok(A, tag, pins):
    """Untimed self-check: sha256 over the device CSC (colptr, rowval, nzval as Julia would see them) against the
    oracle's pin for this configuration.  True / False, or None when there is no pin for `tag`."""
    import hashlib
    if tag not in pins:
        return None
    cp, rv, nz = A.sparse().arrays()
    if len(rv) != int(pins[tag]["nnz"]):
        return False
    h = hashlib.sha256()
    for a in (cp, rv, nz):
        h.update(memoryview(a).cast("B"))
    return h.hexdigest() == pins[tag]["csc"]


def cfg3_new_positions(n, seed=0x5EED0003):
    """Config 3's new entries (same generator as tests/golden_util.cfg3_new_positions, which made the pin): the x
    second-neighbour pairs (l,l+2),(l+2,l), values U[0,1) from numpy's default_rng(seed)."""
    import numpy as np
    g = np.arange(n ** 3, dtype=np.int64)
    l = g[(g % n) < n - 2] + 1
    v = np.random.default_rng(seed).random(len(l))
    return np.concatenate([l, l + 2]), np.concatenate([l + 2, l]), np.concatenate([v, v])


def mt_per_entry_streams(n=4000000, p=16, cnt=2000000, seed=1):
    """The per-entry form of the reference's multi-threaded assembly (same generator as tests/golden_util.mt_per_entry_streams, which
    made the pin mtgen_4M_p16): task t sends cnt updateindex! / rawupdateindex! calls to the columns of its band."""
    import numpy as np
    rng = np.random.default_rng(seed)
    out = []
    for t in range(p):
        J = np.sort(rng.integers(t * n // p + 1, (t + 1) * n // p + 1, cnt))
        I = np.clip(J + rng.integers(-30, 31, cnt), 1, n)
        out.append((I, J, rng.standard_normal(cnt), rng.integers(1, 3, cnt).astype(np.uint8)))
    return out


def fd_counts(n):
    E = 12 * n * n * (n - 1) + 6 * n * n
    Z = n ** 3 + 6 * n * n * (n - 1)
    return E, Z


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(sample_n, mt_n):
    """Reference algorithm (LNK insert via updateindex! + lnk+csc flush) on the host, 1 thread, and the
    P-thread shape of MTExtendableSparseMatrixCSC (one buffer per thread + COO merge) as `mt`."""
    from oracle import oracle as orc
    z, ti, tf = orc.bench_fdrand(sample_n, sample_n, sample_n, orc.KIND_UPDATE)
    out = {"value": z / (ti + tf), "unit": "nnz/s", "cores": 1, "kind": "port",
           "sample": "one fdrand %d^3 fresh assemble+flush! (%d update calls, %d nnz), updateindex! style, "
                     "C restatement of SparseMatrixLNK + lnk+csc (oracle/, %s), insert %.2fs + flush %.2fs; "
                     "host: %s, %d cores, 1 used (the reference path is single-threaded)"
                     % (sample_n, 12 * sample_n * sample_n * (sample_n - 1) + 6 * sample_n * sample_n, z,
                        orc.build_flags(), ti, tf, cpu_model(), os.cpu_count())}
    if mt_n > 0 and hasattr(orc, "bench_fdrand_mt"):
        try:
            p = max(1, min(os.cpu_count() or 1, 16))
            zm, tmi, tmf = orc.bench_fdrand_mt(mt_n, mt_n, mt_n, p)
            out["mt"] = {"value": zm / (tmi + tmf), "unit": "nnz/s", "cores": p, "kind": "port",
                         "sample": "fdrand %d^3, %d threads, one column-slab buffer per thread (SparseMatrixDILNKC "
                                   "shape) + COO merge (sparsematrixdilnkc.jl:397-435): insert %.2fs + merge %.2fs"
                                   % (mt_n, p, tmi, tmf)}
        except Exception as ex:   # the secondary baseline must never cost the headline line
            out["mt"] = {"error": str(ex)}
    return out


def baseline_metric():
    """BASELINE.json's metric string (the line reports its nnz/s part as `value`; the HBM GB/s and %-of-peak
    part is in `roofline` and `pipeline`)."""
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as f:
            return json.load(f)["metric"]
    except Exception:
        return "assembled+flushed nnz/sec and HBM GB/s %peak, 256^3 7-pt stencil, 1/2/4/8 GPU"


# ---------------------------------------------------------------------------------------- launcher
def gpu_count_without_hip():
    """GPUs of this node as the rank processes will see them: the KFD topology in sysfs (nodes with SIMDs), clamped by
    the *_VISIBLE_DEVICES lists a child process inherits (HIP_VISIBLE_DEVICES, ROCR_VISIBLE_DEVICES, CUDA_VISIBLE_DEVICES:
    comma-separated indices or UUIDs -- their length is what counts) -- so that the launcher never brings up a HIP runtime
    of its own.  None when /sys/class/kfd is not there (then torch counts, which may initialise HIP -- harmless here: the
    launcher only spawns fresh children and never re-executes itself)."""
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        n = 0
        for node in os.listdir(base):
            with open(os.path.join(base, node, "properties")) as f:
                props = dict(line.split() for line in f if len(line.split()) == 2)
            n += int(props.get("simd_count", "0")) > 0
    except Exception:
        return None
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip()]))
    return n


def launch(args, argv):
    """Parent of a plain `python bench.py --gpus N`: spawns the N ranks and supervises ALL of them -- the first rank that
    exits with an error (or the wall-clock limit) ends the others, so a rank that dies at start-up cannot leave rank 0
    waiting in the rendezvous for ever; every rank's stderr is kept and shown on failure."""
    have = gpu_count_without_hip()
    if have is None:
        import torch
        have = torch.cuda.device_count()
    if have < args.gpus:
        print("bench.py: --gpus %d but this node shows %d GPU(s); refusing to run a smaller job under that name"
              % (args.gpus, have), file=sys.stderr)
        return 2
    import tempfile
    port = int(os.environ.get("MASTER_PORT", "29541"))
    limit = float(os.environ.get("ESP_BENCH_LAUNCH_TIMEOUT", "1800"))
    procs, errs = [], []
    out0 = tempfile.TemporaryFile(mode="w+")
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(args.gpus), "LOCAL_WORLD_SIZE": str(args.gpus),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        errs.append(tempfile.TemporaryFile(mode="w+"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL, stderr=errs[-1], text=True))
    t0 = time.time()
    rc = 0
    alive = set(range(args.gpus))
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is not None:
                alive.discard(r)
                if code != 0 and rc == 0:
                    rc = code
                    print("bench.py: rank %d exited with code %d" % (r, code), file=sys.stderr)
        if rc != 0 or time.time() - t0 > limit:
            if rc == 0:
                rc = 124
                print("bench.py: ranks still running after %.0f s" % limit, file=sys.stderr)
            for r in alive:
                procs[r].terminate()
            for r in alive:
                try:
                    procs[r].wait(timeout=10)
                except Exception:
                    procs[r].kill()
            alive.clear()
        elif alive:
            time.sleep(0.05)
    if rc != 0:
        for r, e in enumerate(errs):
            e.seek(0)
            tail = e.read()[-2000:]
            if tail.strip():
                print("---- stderr of rank %d ----\n%s" % (r, tail), file=sys.stderr)
    out0.seek(0)
    lines = [ln for ln in out0.read().splitlines() if ln.strip()]
    if lines:
        print(lines[-1], flush=True)
    return rc


def launch_probe(args):
    """Test hook of the launcher (tests/test_bench_launcher.py, no GPU): the ranks it made meet over gloo and rank 0
    reports what it saw instead of running the bench."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        return 3
    if os.environ.get("ESP_BENCH_PROBE_FAIL_RANK") == str(rank):   # (a rank that dies before the rendezvous)
        print("probe: rank %d fails on purpose" % rank, file=sys.stderr)
        return 7
    if world > 1:
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        t = torch.tensor([float(rank)], dtype=torch.float64)
        dist.all_reduce(t)
        ranks = float(t.item())
        dist.destroy_process_group()
    else:
        ranks = 0.0
    if rank == 0:
        print(json.dumps({"probe": True, "n_gpus": world, "rank_sum": ranks, "local_rank": int(os.environ.get("LOCAL_RANK", "-1")),
                          "master": os.environ.get("MASTER_ADDR")}), flush=True)
    return 0


# ---------------------------------------------------------------------------------------- extra configs
def extra_configs(esp, torch, local, n_cfg3, fem2d, fem3d, steps=3):
    """BASELINE.json configs 3 and 4 in front of the driver (bounded: `steps` steps each, after the headline
    loop).  Algorithmic bytes as in SURVEY.md 8d; frac = bytes / time / 8 TB/s."""
    import ctypes as C
    out = {}
    pins = golden_digests()
    only = os.environ.get("ESP_EXTRA_ONLY", "")       # (tools: e.g. "cfg4" -- the driver's run measures everything)

    class _Skip(Exception):
        pass

    # Python's cyclic collector stays off inside the timed loops (what `timeit` does): a full collection of a process that has
    # torch loaded pauses EVERY thread for 50 - 80 ms, and the one the interpreter schedules a few hundred allocations into a run
    # fell into the 10 - 20 timed rounds of cfg_mt_sum in about every other process -- 4.75 ms rounds with one of 55 - 85 ms,
    # i.e. a mean of 7 - 11 ms (NOTES/round6.md section 10; the driver's round-5 record of this line: 23.6 ms against 6.7).
    # Every config starts with an explicit collection instead; reference counting frees everything the loops drop.
    import gc
    gc.collect()
    gc.disable()

    def want(name):
        if only and name not in only.split(","):
            raise _Skip()
        gc.collect()

    def stages(tm, reps):
        return {k: round(v[0] / reps, 3) for k, v in tm.items() if isinstance(v, tuple) and v[0] > 0}

    # ---- config 2 through the path every caller other than the built-in generators takes: the 256^3 stream as resident
    # Int64/Int64/Float64 device arrays -> esp_append_device -> flush!  (cfg2_generic_append), and from pageable HOST arrays
    # -> esp_append_host -> flush! -> esp_get_csc into host arrays (cfg2_host: what a Julia caller of extendable.jl:159-218 +
    # 258-261 sees; PCIe-inclusive, never the headline value)
    try:
        want("cfg2")
        import numpy as np
        n = n_cfg3
        N = n ** 3
        E, Z = fd_counts(n)
        G = esp.ExtendableSparseMatrix(N, N, device=local, capacity_hint=E)
        G.debug_force_path(16)                      # (the plain producer: packed keys in STREAM order, as a caller's loop emits them)
        G.generate_fdrand(n, n, n, seed=0x5EED0002, rand_mode=1, kind=esp.ESP_UPDATE)
        keys = torch.empty(E, dtype=torch.int64, device="cuda")
        vals = torch.empty(E, dtype=torch.float64, device="cuda")
        offs = (C.c_int64 * 2)()
        rbits, cbits = C.c_int32(), C.c_int32()
        gd = G._d
        gd.ck(gd.lib.esp_key_layout(gd.h, C.byref(rbits), C.byref(cbits)))
        gd.ck(gd.lib.esp_shard_export(gd.h, 1, C.c_void_p(keys.data_ptr()), C.c_void_p(vals.data_ptr()), offs))   # (one owner: a copy)
        assert offs[1] == E
        rb = rbits.value
        rows = ((keys >> 2) & ((1 << rb) - 1)) + 1
        cols = (keys >> (2 + rb)) + 1
        del keys, G, gd
        A = esp.ExtendableSparseMatrix(N, N, device=local, capacity_hint=E)
        d = A._d
        torch.cuda.synchronize()
        dts, tm = [], None
        for it in range(steps + 2):
            A.timing_enable(1 if it == steps + 1 else 0)
            A.timing(clear=True)
            A.synchronize()
            t0 = time.perf_counter()
            A.reset()
            d.ck(d.lib.esp_append_device(d.h, C.c_void_p(rows.data_ptr()), C.c_void_p(cols.data_ptr()),
                                         C.c_void_p(vals.data_ptr()), None, esp.ESP_UPDATE, 0, E))
            A._touch()
            A.flush()
            A.synchronize()
            if it == steps + 1:
                tm = A.timing(clear=True)
            elif it > 0:
                dts.append(time.perf_counter() - t0)
        assert A.nnz() == Z
        okg = csc_digest_ok(A, "fd_%d_m1" % n, pins)
        dt = sum(dts) / len(dts)
        algo = 24.0 * E + 2 * 16.0 * E + 16.0 * Z + 8.0 * (N + 1)     # + the caller's triplets, read once
        out["cfg2_generic_append"] = {
            "workload": "fdrand %d^3 stream from resident Int64/Int64/Float64 device arrays: esp_append_device + flush!" % n,
            "ms": dt * 1e3, "nnz_per_s": Z / dt, "algorithmic_bytes": algo, "frac_of_hbm_peak": algo / dt / 1e9 / HBM_PEAK_GBS,
            "partition": A.debug_last_partition(), "stage_ms": stages(tm, 1), "steps": len(dts), "digest_ok": okg}
        # host-visible: pageable host arrays in, host arrays out
        hr, hc, hv = rows.cpu().numpy(), cols.cpu().numpy(), vals.cpu().numpy()
        del rows, cols, vals
        cp = np.empty(N + 1, np.int64)
        rv = np.empty(Z, np.int64)
        nz = np.empty(Z, np.float64)
        rv[:] = 0
        nz[:] = 0                                   # (touched: the first D2H into fresh pages pays the page faults)
        dts = []
        for it in range(3):
            t0 = time.perf_counter()
            A.reset()
            A.append(esp.ESP_UPDATE, hr, hc, hv)
            A.flush()
            A.synchronize()
            t1 = time.perf_counter()
            d.ck(d.lib.esp_get_csc(d.h, C.c_void_p(cp.ctypes.data), C.c_void_p(rv.ctypes.data), C.c_void_p(nz.ctypes.data)))
            t2 = time.perf_counter()
            if it > 0:
                dts.append((t1 - t0, t2 - t1))
        ta = sum(x[0] for x in dts) / len(dts)
        tg = sum(x[1] for x in dts) / len(dts)
        out["cfg2_host"] = {
            "workload": "the same stream from pageable host arrays (esp_append_host), flush!, then esp_get_csc into host arrays: "
                        "the host-visible end point of SURVEY.md 8d (PCIe-inclusive)",
            "append_flush_ms": ta * 1e3, "get_csc_ms": tg * 1e3, "ms": (ta + tg) * 1e3, "nnz_per_s": Z / (ta + tg),
            "h2d_GBs": 24.0 * E / ta / 1e9, "d2h_GBs": (16.0 * Z + 8.0 * (N + 1)) / tg / 1e9, "steps": len(dts)}
        del A, hr, hc, hv, cp, rv, nz
    except _Skip:
        pass
    except Exception as ex:
        out["cfg2_generic_append"] = out.get("cfg2_generic_append", {"error": repr(ex)})
        out["cfg2_host"] = out.get("cfg2_host", {"error": repr(ex)})

    # ---- config 3: existing CSC = the config-2 result; the full stencil stream again (all hits) plus the x
    # second-neighbour pairs (l,l+2),(l+2,l) as new positions (28.4 % of Z0), one flush (merge-path join hot)
    try:
        want("cfg3")
        n = n_cfg3
        N = n ** 3
        E, Z0 = fd_counts(n)
        I2, J2, V2 = cfg3_new_positions(n)           # (host-made: the very values the oracle's pin was made from)
        rows = torch.from_numpy(I2).cuda()
        cols = torch.from_numpy(J2).cuda()
        vals = torch.from_numpy(V2).cuda()
        Zn = rows.numel()
        del I2, J2, V2
        A = esp.ExtendableSparseMatrix(N, N, device=local, capacity_hint=E + Zn)
        d = A._d
        torch.cuda.synchronize()
        dts = []
        tm = None
        for it in range(steps + 2):          # one warm-up, `steps` timed, one with stage events
            A.timing_enable(0)
            A.reset()
            A.generate_fdrand(n, n, n, seed=0x5EED0002, rand_mode=1)
            A.flush()                          # the stored CSC (untimed)
            A.synchronize()
            A.timing_enable(1 if it == steps + 1 else 0)
            A.timing(clear=True)
            t0 = time.perf_counter()
            # (the stencil's stream, then the new couplings behind it: the producer's bucket-ordered batch is flushed as
            # it is, the new couplings as a flush of their own -- esp_debug_last_partition 8: the couplings partitioned as they are appended)
            A.generate_fdrand(n, n, n, seed=0x5EED0012, rand_mode=1)
            d.ck(d.lib.esp_append_device(d.h, C.c_void_p(rows.data_ptr()), C.c_void_p(cols.data_ptr()),
                                         C.c_void_p(vals.data_ptr()), None, esp.ESP_UPDATE, 0, Zn))
            A._touch()
            A.flush()
            A.synchronize()
            if it == steps + 1:
                tm = A.timing(clear=True)
            elif it > 0:
                dts.append(time.perf_counter() - t0)
        Z1 = A.nnz()
        assert Z1 == Z0 + Zn, (Z1, Z0, Zn)
        ok3 = csc_digest_ok(A, "cfg3_%d" % n, pins)
        dt = sum(dts) / len(dts)
        Ea = E + Zn
        algo = 2 * 16.0 * Ea + (16.0 * Z0 + 8.0 * (N + 1)) + (16.0 * Z1 + 8.0 * (N + 1))
        out["cfg3_reassembly"] = {
            "workload": "existing %d^3 stencil CSC (%d nnz) + the full update stream again + %d new positions "
                        "(%.1f %% of the stored nnz): esp_append_device behind the producer's batch + flush! (batch by itself over the stored pattern, then the new couplings; column-tiled join)" % (n, Z0, Zn, 100.0 * Zn / Z0),
            "ms": dt * 1e3, "nnz_per_s": Z1 / dt, "appended_per_s": Ea / dt, "algorithmic_bytes": algo,
            "frac_of_hbm_peak": algo / dt / 1e9 / HBM_PEAK_GBS, "partition": A.debug_last_partition(),
            "stage_ms": stages(tm, 1), "steps": len(dts), "digest_ok": ok3}
        del A, rows, cols, vals
    except _Skip:
        pass
    except Exception as ex:
        out["cfg3_reassembly"] = {"error": repr(ex)}

    # ---- config 4: P1 FEM in random cell order (test/femtools.jl:45-72), ~10 M DoF, 2-D and 3-D
    for tag, dim, npd in (("cfg4_fem2d", 2, fem2d), ("cfg4_fem3d", 3, fem3d)):
        if npd <= 0 or (only and "cfg4" not in only.split(",")):
            continue
        try:
            nn = npd ** dim
            q = npd - 1
            E = (2 * q * q if dim == 2 else 6 * q ** 3) * (dim + 1) * (dim + 2)
            A = esp.ExtendableSparseMatrix(nn, nn, device=local, capacity_hint=E)
            dts = []
            tm = None
            for it in range(steps + 2):
                A.timing_enable(1 if it == steps + 1 else 0)
                A.timing(clear=True)
                A.synchronize()
                t0 = time.perf_counter()
                A.reset()
                A.generate_fem(dim, npd, seed=0x5EED0004, order_mode=1)
                A.flush()
                A.synchronize()
                if it == steps + 1:
                    tm = A.timing(clear=True)
                elif it > 0:
                    dts.append(time.perf_counter() - t0)
            Z = A.nnz()
            ok4 = csc_digest_ok(A, "fem%dd_%d_o1" % (dim, npd), pins)
            dt = sum(dts) / len(dts)
            algo = 2 * 16.0 * E + 16.0 * Z + 8.0 * (nn + 1)
            out[tag] = {"workload": "P1 FEM %d-D, %d^%d nodes (%d DoF), Kuhn grid, random cell order: %d rawupdateindex! "
                                    "calls -> fresh CSC" % (dim, npd, dim, nn, E),
                        "ms": dt * 1e3, "nnz_per_s": Z / dt, "appended_per_s": E / dt, "final_nnz": Z,
                        "algorithmic_bytes": algo, "frac_of_hbm_peak": algo / dt / 1e9 / HBM_PEAK_GBS,
                        "partition": A.debug_last_partition(), "stage_ms": stages(tm, 1), "steps": len(dts), "digest_ok": ok4}
            # ---- a time step's re-assembly over the stored pattern: zero!(A) (sprand.jl:82 style), the stream again, flush! --
            # every update meets a stored position (the group kernel's re-assembly form); the result is the fresh build's, bit
            # for bit (0.0 + v1 + v2 ... in call order either way): the same pin
            rtag = tag + "_reassembly"
            try:
                dts = []
                for it in range(steps + 1):
                    A.synchronize()
                    t0 = time.perf_counter()
                    A.zero_values()
                    A.generate_fem(dim, npd, seed=0x5EED0004, order_mode=1)
                    A.flush()
                    A.synchronize()
                    if it > 0:
                        dts.append(time.perf_counter() - t0)
                okr = csc_digest_ok(A, "fem%dd_%d_o1" % (dim, npd), pins)
                dt = sum(dts) / len(dts)
                algo = 2 * 16.0 * E + 2 * 8.0 * Z + 8.0 * Z + 8.0 * (nn + 1)   # the stream written and read, nzval read and written, rowval + colptr read
                out[rtag] = {"workload": "re-assembly of the same %d-D mesh over its stored pattern: zero!, %d rawupdateindex! calls, flush! "
                                         "(every update hits)" % (dim, E),
                             "ms": dt * 1e3, "appended_per_s": E / dt, "algorithmic_bytes": algo, "frac_of_hbm_peak": algo / dt / 1e9 / HBM_PEAK_GBS,
                             "bucket_kernel": A.debug_last_local_small(), "steps": len(dts), "digest_ok": okr}
            except Exception as ex:
                out[rtag] = {"error": repr(ex)}
            del A
        except Exception as ex:
            out[tag] = {"error": repr(ex)}
        # ---- the same assembly as a CALLER with a mesh in memory runs it (test/femtools.jl:45-72): connectivity and
        # element matrices resident in HBM -> esp_append_elements -> flush!  (the library reads cellnodes / elmat / diag; the
        # built-in generator above derives them from the cell number).  Natural node numbering: the stream is the
        # generator's, the digest the same pin.
        etag = "cfg4_elements_%dd" % dim
        try:
            nloc, W = dim + 1, dim + 2
            nc = E // (nloc * W)
            A = esp.ExtendableSparseMatrix(nn, nn, device=local, capacity_hint=E)
            cn = torch.empty((nc, nloc), dtype=torch.int64, device="cuda")
            em = torch.empty((nc, nloc, nloc), dtype=torch.float64, device="cuda")
            dg = torch.empty((nc, nloc), dtype=torch.float64, device="cuda")
            A.generate_fem_mesh(dim, npd, cn, em, dg, seed=0x5EED0004, order_mode=1)
            A.synchronize()
            dts, tm = [], None
            for it in range(steps + 2):
                A.timing_enable(1 if it == steps + 1 else 0)
                A.timing(clear=True)
                A.synchronize()
                t0 = time.perf_counter()
                A.reset()
                A.append_elements(cn, em, dg, kind=esp.ESP_RAWUPDATE)
                A.flush()
                A.synchronize()
                if it == steps + 1:
                    tm = A.timing(clear=True)
                elif it > 0:
                    dts.append(time.perf_counter() - t0)
            Z = A.nnz()
            oke = csc_digest_ok(A, "fem%dd_%d_o1" % (dim, npd), pins)
            dt = sum(dts) / len(dts)
            inp = 8.0 * nc * nloc * (nloc + 2)          # cellnodes + elmat + diag, read once
            algo = inp + 2 * 16.0 * E + 16.0 * Z + 8.0 * (nn + 1)
            out[etag] = {"workload": "the same P1 FEM assembly from a mesh held in HBM: cellnodes (Int64 %d x %d), elmat "
                                     "(Float64 %d x %d x %d), diag -> esp_append_elements + flush!" % (nloc, nc, nloc, nloc, nc),
                         "ms": dt * 1e3, "nnz_per_s": Z / dt, "appended_per_s": E / dt, "final_nnz": Z,
                         "algorithmic_bytes": algo, "frac_of_hbm_peak": algo / dt / 1e9 / HBM_PEAK_GBS,
                         "vs_generator": dt * 1e3 / out[tag]["ms"] if "ms" in out.get(tag, {}) else None,
                         "partition": A.debug_last_partition(), "stage_ms": stages(tm, 1), "steps": len(dts), "digest_ok": oke}
            # ---- a time step of the same mesh: zero!, the element loop over the KEPT plan (esp_append_elements_again: no pass
            # over the connectivity, no item partition), flush! over the stored pattern; the result is the fresh build's
            ptag = etag + "_timestep"
            try:
                A.elements_keep_plan()
                A.reset()
                A.append_elements(cn, em, dg, kind=esp.ESP_RAWUPDATE)
                A.flush()
                dts = []
                for it in range(steps + 1):
                    A.synchronize()
                    t0 = time.perf_counter()
                    A.zero_values()
                    A.append_elements_again(em, dg, kind=esp.ESP_RAWUPDATE)
                    A.flush()
                    A.synchronize()
                    if it > 0:
                        dts.append(time.perf_counter() - t0)
                okp = csc_digest_ok(A, "fem%dd_%d_o1" % (dim, npd), pins)
                dt = sum(dts) / len(dts)
                algo = 8.0 * nc * nloc * (nloc + 1) + 2 * 16.0 * E + 2 * 8.0 * Z + 8.0 * Z + 8.0 * (nn + 1)
                out[ptag] = {"workload": "a time step of the same mesh: zero!, esp_append_elements_again (kept item order and cell records, new "
                                         "elmat / diag), flush! over the stored pattern", "ms": dt * 1e3, "appended_per_s": E / dt,
                             "algorithmic_bytes": algo, "frac_of_hbm_peak": algo / dt / 1e9 / HBM_PEAK_GBS,
                             "bucket_kernel": A.debug_last_local_small(), "steps": len(dts), "digest_ok": okp}
                A.elements_keep_plan(False)
            except Exception as ex:
                out[ptag] = {"error": repr(ex)}
            # ... and as resident Int64 / Int64 / Float64 triplets through esp_append_device (what a caller without the
            # element-level call pays: a shuffled stream, the flush's own passes over 16-byte records)
            ttag = "cfg4_generic_triplets_%dd" % dim
            try:
                if os.environ.get("ESP_BENCH_SKIP_TRIPLETS"):
                    raise _Skip()
                I = cn[:, :, None].expand(nc, nloc, W).reshape(-1).contiguous()
                J = torch.cat([cn[:, :, None], cn[:, None, :].expand(nc, nloc, nloc)], dim=2).reshape(-1)
                V = torch.cat([dg[:, :, None], em.transpose(1, 2)], dim=2).reshape(-1)
                torch.cuda.synchronize()      # (torch made the triplets on ITS stream; the library reads them on the handle's own)
                del cn, em, dg
                dts, tm = [], None
                for it in range(steps + 2):
                    A.timing_enable(1 if it == steps + 1 else 0)
                    A.timing(clear=True)
                    A.synchronize()
                    t0 = time.perf_counter()
                    A.reset()
                    A.append_device(esp.ESP_RAWUPDATE, I, J, V)
                    A.flush()
                    A.synchronize()
                    if it == steps + 1:
                        tm = A.timing(clear=True)
                    elif it > 0:
                        dts.append(time.perf_counter() - t0)
                okt = csc_digest_ok(A, "fem%dd_%d_o1" % (dim, npd), pins)
                dt = sum(dts) / len(dts)
                algo = 24.0 * E + 2 * 16.0 * E + 16.0 * Z + 8.0 * (nn + 1)
                out[ttag] = {"workload": "the same stream as resident Int64/Int64/Float64 triplets (%d): esp_append_device + flush!" % E,
                             "ms": dt * 1e3, "nnz_per_s": Z / dt, "appended_per_s": E / dt, "algorithmic_bytes": algo,
                             "frac_of_hbm_peak": algo / dt / 1e9 / HBM_PEAK_GBS, "partition": A.debug_last_partition(),
                             "key_bytes": A.debug_last_key_bytes(), "stage_ms": stages(tm, 1), "steps": len(dts), "digest_ok": okt}
                del I, J, V
            except _Skip:
                pass
            except Exception as ex:
                out[ttag] = {"error": repr(ex)}
            del A
        except Exception as ex:
            out[etag] = {"error": repr(ex)}
    # ---- the path every rank of a multi-GPU run takes (BASELINE config 5's per-rank work), on this one GPU: a single-rank group --
    # column-shard plan, producer partitions by (owner, digit), exchange policy with nothing to send, PIECES bucket kernel.  Not a
    # multi-GPU measurement (there has never been a node to make one): the overhead of the sharded path over the headline's.
    try:
        want("cfg5")
        n = n_cfg3
        N = n ** 3
        E, Z = fd_counts(n)
        SA = esp.GroupShardedMatrix(N, N, nranks=1, rank=0, device=local, capacity_hint=E + 8 * n * n, unique_id=esp.GroupShardedMatrix.unique_id())
        A = SA.local
        dts = []
        for it in range(steps + 3):
            A.synchronize()
            t0 = time.perf_counter()
            A.reset()
            A.generate_fdrand_range(n, n, n, 0, N, seed=0x5EED0002, rand_mode=1, kind=esp.ESP_UPDATE)
            SA.flush()
            A.synchronize()
            if it > 2:
                dts.append(time.perf_counter() - t0)
        ok5 = csc_digest_ok(A, "fd_%d_m1" % n, pins)
        dt = sum(dts) / len(dts)
        algo = 2 * 16.0 * E + 16.0 * Z + 8.0 * (N + 1)
        out["cfg5_one_rank_shard"] = {
            "workload": "fdrand %d^3 through the column-shard path with ONE rank (esp_group_flush: shard plan, producer partition by (owner, digit), "
                        "PIECES bucket kernel; nothing travels): what each rank of config 5 does locally -- NOT a multi-GPU measurement" % n,
            "ms": dt * 1e3, "nnz_per_s": Z / dt, "algorithmic_bytes": algo, "frac_of_hbm_peak": algo / dt / 1e9 / HBM_PEAK_GBS,
            "key_bytes": A.debug_last_key_bytes(), "shard_source": A.debug_last_shard_source(), "steps": len(dts), "digest_ok": ok5}
        del SA, A
    except _Skip:
        pass
    except Exception as ex:
        out["cfg5_one_rank_shard"] = {"error": repr(ex)}
    # ---- above 32 key bits below the planned prefix: 322^3 (33 bits) takes the FINE partition -- 4-byte keys as at 256^3 -- unsharded and
    # as the one-rank shard (round 6); pinned by the oracle's digest at this size
    try:
        want("cfg5")
        n = 322
        N = n ** 3
        E, Z = fd_counts(n)
        algo = 2 * 16.0 * E + 16.0 * Z + 8.0 * (N + 1)
        for tag, sharded in (("cfg_large_322", False), ("cfg_large_322_one_rank_shard", True)):
            if sharded:
                SA = esp.GroupShardedMatrix(N, N, nranks=1, rank=0, device=local, capacity_hint=E + 8 * n * n, unique_id=esp.GroupShardedMatrix.unique_id())
                A = SA.local
            else:
                SA = None
                A = esp.ExtendableSparseMatrix(N, N, device=local, capacity_hint=E + 8 * n * n)
            dts = []
            for it in range(min(steps, 10) + 3):
                A.synchronize()
                t0 = time.perf_counter()
                A.reset()
                A.generate_fdrand_range(n, n, n, 0, N, seed=0x5EED0002, rand_mode=1, kind=esp.ESP_UPDATE)
                (SA if sharded else A).flush()
                A.synchronize()
                if it > 2:
                    dts.append(time.perf_counter() - t0)
            okl = csc_digest_ok(A, "fd_%d_m1" % n, pins)
            dt = sum(dts) / len(dts)
            out[tag] = {"workload": "fdrand %d^3 fresh assemble + flush (%s): 33 key bits below the planned prefix, 4-byte keys through the FINE partition"
                                    % (n, "column-shard path with ONE rank" if sharded else "unsharded"),
                        "ms": dt * 1e3, "nnz_per_s": Z / dt, "algorithmic_bytes": algo, "frac_of_hbm_peak": algo / dt / 1e9 / HBM_PEAK_GBS,
                        "key_bytes": A.debug_last_key_bytes(), "steps": len(dts), "digest_ok": okl}
            del SA, A
    except _Skip:
        pass
    except Exception as ex:
        out["cfg_large_322"] = {"error": repr(ex)}
    # ---- flush! of the MT wrapper (genericmtextendablesparsematrixcsc.jl:45-51 = Base.sum(xmatrices, csc)): the 2-D mesh
    # dealt to 16 partition buffers (contiguous chunks of the shuffled cell order), each filled by esp_append_elements, then
    # ONE esp_flush_sum into the handle that keeps the CSC -- device-resident; `plugin_ms`: the same through
    # SparseMatrixHIPCOO.sum with a host SparseMatrixCSC on both sides (what the shim's Base.sum pays in PCIe)
    try:
        want("cfg_mt_sum")
        dim, npd, p = 2, fem2d, 16
        if npd <= 0:
            raise _Skip()
        nn, nloc, W = npd ** dim, dim + 1, dim + 2
        q = npd - 1
        nc = 2 * q * q
        E = nc * nloc * W
        home = esp.SparseMatrixHIPCOO(nn, nn, device=local)
        xs = [esp.SparseMatrixHIPCOO(nn, nn, device=local) for _ in range(p)]
        cn = torch.empty((nc, nloc), dtype=torch.int64, device="cuda")
        em = torch.empty((nc, nloc, nloc), dtype=torch.float64, device="cuda")
        dg = torch.empty((nc, nloc), dtype=torch.float64, device="cuda")
        A0 = esp.ExtendableSparseMatrix(nn, nn, device=local)
        # (the mesh in its natural cell order: a partition = a contiguous chunk of cells = a band of the grid, like the node
        # partitions of a partitioned grid in test_parallel.jl:57; neighbouring bands share the columns of one grid line)
        A0.generate_fem_mesh(dim, npd, cn, em, dg, seed=0x5EED0004, order_mode=0)
        A0.synchronize()
        cuts = [nc * t // p for t in range(p + 1)]
        arr = (C.c_void_p * p)(*[x._d.h for x in xs])
        hd = home._d
        dts = []
        Z = 0
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(p)          # (one host thread per partition, like the tasks of testassemble_parallel!)

        def fill(t):
            xs[t].append_elements(cn[cuts[t]:cuts[t + 1]], em[cuts[t]:cuts[t + 1]], dg[cuts[t]:cuts[t + 1]])

        # (every fill, the buffers' first ones -- all their device allocations -- included, runs on its partition's own host thread:
        # distinct handles are independent since round 6, NOTES/round6.md section 1)
        fills, folds, combine, lazy, join = [], [], [], [], []
        for it in range(steps + 2):
            hd.ck(hd.lib.esp_synchronize(hd.h))
            t0 = time.perf_counter()
            hd.ck(hd.lib.esp_reset(hd.h))
            list(pool.map(fill, range(p)))
            t1 = time.perf_counter()
            z, ch = C.c_int64(), C.c_int32()
            hd.ck(hd.lib.esp_flush_sum(hd.h, arr, p, C.byref(z), C.byref(ch)))
            hd.ck(hd.lib.esp_synchronize(hd.h))
            if it > 1:                             # (two untimed rounds: first use, then the handles' planning history settles)
                dts.append(time.perf_counter() - t0)
                fills.append(t1 - t0)
                f_ms, c_ms, lz, jn = C.c_double(), C.c_double(), C.c_int32(), C.c_int32()
                hd.ck(hd.lib.esp_debug_last_sum_ms(hd.h, C.byref(f_ms), C.byref(c_ms)))
                hd.ck(hd.lib.esp_debug_last_lazy_items(hd.h, C.byref(lz)))
                hd.ck(hd.lib.esp_debug_last_sum_join(hd.h, C.byref(jn)))
                folds.append(f_ms.value), combine.append(c_ms.value), lazy.append(lz.value), join.append(jn.value)
            Z = z.value
        dt = sum(dts) / len(dts)
        # untimed self-check: the device CSC of the last timed round against the oracle's MT wrapper (tests/golden/make_digests_large.py)
        okm = None
        mtag = "mt2d_%d_p%d" % (npd, p)
        if mtag in pins:
            import hashlib
            cpd, rvd, nzd = hd.get_csc().arrays()
            hh = hashlib.sha256()
            for a_ in (cpd, rvd, nzd):
                hh.update(memoryview(a_).cast("B"))
            okm = len(rvd) == int(pins[mtag]["nnz"]) and hh.hexdigest() == pins[mtag]["csc"]
            del cpd, rvd, nzd
        # the plug-in form: host matrix in, host matrix out (fresh: the whole CSC comes back; a second flush over the
        # same pattern moves values only)
        csc = esp.SparseMatrixCSC(nn, nn)
        tp, tfree = [], []
        for it in range(3):
            list(pool.map(fill, range(p)))
            t0 = time.perf_counter()
            res = esp.SparseMatrixHIPCOO.sum(xs, csc, home=home if it else None)
            tp.append(time.perf_counter() - t0)
            # (the previous result dies HERE, outside the timed call: handing 560 MB back to the kernel costs the host allocator
            # ~21 ms on this pool -- tools/r6_dl_probe2.py -- which Julia's GC pays whenever it collects the old matrix, not flush!)
            t0 = time.perf_counter()
            csc = res
            del res
            tfree.append(time.perf_counter() - t0)
            if it == 0:
                home._mirror = None
                hd.set_csc(csc)
                home._mirror = (csc.colptr, csc.rowval)
        algo = 8.0 * nc * nloc * (nloc + 2) + 2 * 16.0 * E + 16.0 * Z + 8.0 * (nn + 1)
        out["cfg_mt_sum"] = {
            "workload": "P1 FEM 2-D %d^2 dealt to %d partition buffers (tids: bands of the grid, one host thread each) filled by "
                        "esp_append_elements, flush! = ONE esp_flush_sum (Base.sum(xmatrices, csc)): every buffer folds by "
                        "itself, the folds meet in one routed flush" % (npd, p),
            "ms": dt * 1e3, "nnz_per_s": Z / dt, "final_nnz": Z, "algorithmic_bytes": algo,
            "frac_of_hbm_peak": algo / dt / 1e9 / HBM_PEAK_GBS, "steps": len(dts), "nnz_ok": csc.nnz() == Z, "digest_ok": okm,
            # which path served (2 = the folds as ONE launch over the buffers' item records; 0 = every buffer's own flush), how many
            # neighbouring segments the combine flush joined, and where the time went (host wall-clock, ms, mean of the timed rounds)
            "lazy_items": min(lazy), "sum_join": min(join), "fills_ms": 1e3 * sum(fills) / len(fills),
            "fills_ms_steps": [round(1e3 * x, 2) for x in fills], "ms_steps": [round(1e3 * x, 2) for x in dts],
            "folds_ms": sum(folds) / len(folds), "combine_ms": sum(combine) / len(combine),
            "plugin_fresh_ms": tp[0] * 1e3, "plugin_same_pattern_ms": min(tp[1:]) * 1e3, "plugin_free_previous_result_ms": max(tfree) * 1e3}
        pool.shutdown()
        del xs, home, cn, em, dg, A0
    except _Skip:
        pass
    except Exception as ex:
        out["cfg_mt_sum"] = {"error": repr(ex)}
    # ---- ... and its per-entry form (test/femtools.jl:88-107: every task calls updateindex! / rawupdateindex! with its tid): 16 buffers
    # of 2 10^6 calls each, mixed kinds -- not element batches: esp_flush_sum's general path -- the buffers' folds as ONE flush of a
    # scratch matrix with their occupied column ranges side by side (round 6), its entries gathered, one routed flush.  Timed: the flush! alone (the per-entry
    # appends are the host loop's).
    try:
        want("cfg_mt_sum")
        import hashlib
        n5, p5 = 4000000, 16
        streams = mt_per_entry_streams(n5, p5)
        xs = [esp.SparseMatrixHIPCOO(n5, n5, device=local) for _ in range(p5)]
        home = esp.SparseMatrixHIPCOO(n5, n5, device=local)
        hd = home._d
        arr = (C.c_void_p * p5)(*[x._d.h for x in xs])
        dts, Z = [], 0
        for it in range(steps + 2):
            hd.ck(hd.lib.esp_reset(hd.h))
            for t, (I, J, V, K) in enumerate(streams):
                xs[t].append(0, I, J, V, kinds=K)
            for x in xs:
                x._d.ck(x._d.lib.esp_synchronize(x._d.h))
            z, ch = C.c_int64(), C.c_int32()
            t0 = time.perf_counter()
            hd.ck(hd.lib.esp_flush_sum(hd.h, arr, p5, C.byref(z), C.byref(ch)))
            hd.ck(hd.lib.esp_synchronize(hd.h))
            if it > 1:
                dts.append(time.perf_counter() - t0)
            Z = z.value
        okp = None
        if "mtgen_4M_p16" in pins:
            cpd, rvd, nzd = hd.get_csc().arrays()
            hh = hashlib.sha256()
            for a_ in (cpd, rvd, nzd):
                hh.update(memoryview(a_).cast("B"))
            okp = len(rvd) == int(pins["mtgen_4M_p16"]["nnz"]) and hh.hexdigest() == pins["mtgen_4M_p16"]["csc"]
            del cpd, rvd, nzd
        dt = sum(dts) / len(dts)
        Ein = sum(len(sx[0]) for sx in streams)
        algo = 16.0 * Ein + 2 * 16.0 * Z + 16.0 * Z + 8.0 * (n5 + 1)   # (pending entries read, the folds written and read, the CSC written)
        out["cfg_mt_sum_per_entry"] = {
            "workload": "%d partition buffers of %d per-entry updateindex! / rawupdateindex! calls each (mixed kinds, a band of columns per tid), "
                        "flush! = ONE esp_flush_sum: general path (the buffers' folds as one flush of a scratch matrix; gather; one routed "
                        "flush) -- the flush! alone is timed" % (p5, len(streams[0][0])),
            "ms": dt * 1e3, "nnz_per_s": Z / dt, "final_nnz": Z, "algorithmic_bytes": algo, "frac_of_hbm_peak": algo / dt / 1e9 / HBM_PEAK_GBS,
            "steps": len(dts), "digest_ok": okp}
        del xs, home, streams
    except _Skip:
        pass
    except Exception as ex:
        out["cfg_mt_sum_per_entry"] = {"error": repr(ex)}
    gc.enable()
    return out


# ---------------------------------------------------------------------------------------- one rank
def bind_near_gpu(torch, local):
    """Run this process on the CPUs of the NUMA node the GPU hangs off (what `numactl --cpunodebind` does for an HPC job): the
    host arrays of the PCIe-inclusive lines (cfg2_host, cfg_mt_sum's plug-in form) are then first-touched beside the GPU's
    root port -- on the two-socket hosts of this pool a process that lands on the other socket moves them at 60 % of the
    rate.  Returns the node, or None (one node, no sysfs, ESP_BENCH_NO_BIND=1: nothing changed)."""
    if os.environ.get("ESP_BENCH_NO_BIND"):
        return None
    try:
        pr = torch.cuda.get_device_properties(local)
        bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        with open("/sys/bus/pci/devices/%s/numa_node" % bdf) as f:
            node = int(f.read().strip())
        if node < 0:
            return None
        with open("/sys/devices/system/node/node%d/cpulist" % node) as f:
            txt = f.read().strip()
        cpus = set()
        for part in txt.split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        cpus &= os.sched_getaffinity(0)
        if len(cpus) < 2:
            return None
        os.sched_setaffinity(0, cpus)
        return node
    except Exception:
        return None


def consumer_lines(esp, torch, A, N, Z, reps=5):
    """mul! / Dirichlet helpers / Jacobi and ILU0 set-up on the resident CSC (device vectors, nothing crosses PCIe): ms per call and,
    for mul!, the fraction of the HBM roofline on its algorithmic bytes (values + row indices once, x and r once: 16 Z + 16 N)."""
    import ctypes as C
    d = A._d
    x = torch.rand(N, dtype=torch.float64, device="cuda")
    r = torch.empty(N, dtype=torch.float64, device="cuda")
    inv = torch.empty(N, dtype=torch.float64, device="cuda")
    idg = torch.empty(N, dtype=torch.int64, device="cuda")
    mk = torch.zeros(N, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    vp = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731

    def timed(fn, n=reps):
        fn()                                  # (first call: the row-wise index of mul!, scratch allocations)
        d.ck(d.lib.esp_synchronize(d.h))
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        d.ck(d.lib.esp_synchronize(d.h))
        return (time.perf_counter() - t0) / n * 1e3

    out = {}
    ms = timed(lambda: d.ck(d.lib.esp_mul(d.h, vp(x), vp(r), 1)))
    algo = 16.0 * Z + 16.0 * N
    out["mul"] = {"ms": ms, "algorithmic_bytes": algo, "frac_of_hbm_peak": algo / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    out["jacobi_setup"] = {"ms": timed(lambda: d.ck(d.lib.esp_jacobi_setup(d.h, vp(inv), 1)))}
    out["ilu0_setup"] = {"ms": timed(lambda: d.ck(d.lib.esp_ilu0_setup(d.h, vp(inv), vp(idg), 1)))}
    out["mark_dirichlet"] = {"ms": timed(lambda: d.ck(d.lib.esp_mark_dirichlet(d.h, C.c_double(1.0e20), vp(mk), 1)))}
    mk[::97] = 1
    torch.cuda.synchronize()
    out["eliminate_dirichlet"] = {"ms": timed(lambda: d.ck(d.lib.esp_eliminate_dirichlet(d.h, vp(mk), 1)), 2), "marked": int(mk.sum().item())}
    del x, r, inv, idg, mk
    return out


def warm_page_cache():
    """Reads every file this process has mapped (the ROCm runtime, torch, the library): on a FRESH box the first GPU process pays
    its page faults from a cold page cache in the middle of timed steps -- stalls of 5 .. 70 ms in some rounds of cfg_mt_sum's
    16-thread fills, gone in every later process (NOTES/round6.md section 10).  Untimed, like the warm-up steps.  Returns bytes read."""
    seen, total = set(), 0
    try:
        with open("/proc/self/maps") as f:
            for ln in f:
                parts = ln.split(None, 5)
                if len(parts) == 6 and parts[5].startswith("/") and ".so" in parts[5]:
                    seen.add(parts[5].strip())
    except OSError:
        return 0
    for path in sorted(seen):
        try:
            with open(path, "rb", buffering=0) as f:
                while True:
                    b = f.read(1 << 24)
                    if not b:
                        break
                    total += len(b)
        except OSError:
            pass
    return total


def summary_of(out):
    """A compact copy of what a reader of the record needs, as the LAST key of the JSON line (the driver keeps the last 2 000
    characters of it): the headline, its roofline and CPU baseline, and for every extra config [ms, fraction of the HBM
    roofline on SURVEY 8d's bytes, digest_ok] -- cfg2_host (PCIe-inclusive): [ms, nnz/s]."""
    r3 = lambda x: None if x is None else round(float(x), 3)
    s = {"ms_per_step": r3(out.get("ms_per_step")), "value": float("%.4g" % out["value"]) if out.get("value") else None,
         "digest_ok": out.get("digest_ok")}
    rf = out.get("roofline") or {}
    s["roofline"] = [rf.get("kernel"), r3(rf.get("avg_launch_ms")), r3(rf.get("frac")),
                     None if rf.get("traffic") is None else float("%.4g" % rf["traffic"])]
    s["pipeline_frac"] = r3((out.get("pipeline") or {}).get("frac_of_hbm_peak"))
    cb = out.get("cpu_baseline") or {}
    if cb:
        s["cpu"] = [float("%.4g" % cb["value"]) if cb.get("value") else None, cb.get("cores"), cb.get("kind")]
    cfgs = {}
    for name, c in ((out.get("extra") or {}).get("configs") or {}).items():
        if "error" in c:
            cfgs[name] = "error"
        elif "frac_of_hbm_peak" in c:
            cfgs[name] = [r3(c.get("ms")), r3(c.get("frac_of_hbm_peak")), c.get("digest_ok", c.get("nnz_ok"))]
            if name == "cfg_mt_sum":  # (... + which path served and where the time went: lazy_items, sum_join, fills / folds / combine ms)
                cfgs[name] += [c.get("lazy_items"), c.get("sum_join"), r3(c.get("fills_ms")), r3(c.get("folds_ms")), r3(c.get("combine_ms"))]
        else:
            cfgs[name] = [r3(c.get("ms")), float("%.4g" % c["nnz_per_s"]) if c.get("nnz_per_s") else None]
    s["plan_reused"], s["first_call_ms"] = out.get("plan_reused"), r3(out.get("first_call_ms"))
    cons = (out.get("extra") or {}).get("consumers") or {}
    if cons and "error" not in cons:
        s["consumers_ms"] = {k: r3(v.get("ms")) for k, v in cons.items()}
        s["mul_frac"] = r3((cons.get("mul") or {}).get("frac_of_hbm_peak"))
    if cfgs:
        s["cfg"] = cfgs
        s["cfg_cols"] = "ms, frac_of_hbm_peak (SURVEY 8d bytes), digest_ok; cfg2_host: ms, nnz_per_s; cfg_mt_sum: + lazy_items, sum_join, fills_ms, folds_ms, combine_ms"
    return s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--n", type=int, default=int(os.environ.get("ESP_BENCH_N", "256")))
    ap.add_argument("--cpu-sample-n", type=int, default=int(os.environ.get("ESP_BENCH_CPU_N", "256")))
    ap.add_argument("--cpu-mt-n", type=int, default=int(os.environ.get("ESP_BENCH_CPU_MT_N", "192")))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra.configs measurements (configs 3 and 4)")
    ap.add_argument("--sharded", action="store_true",
                    help="use the column-shard exchange path even on one GPU (always used for --gpus > 1)")
    ap.add_argument("--scrambled", action="store_true",
                    help="N > 1: the z-planes of the global grid are dealt round-robin to the ranks in blocks of 8 (SURVEY 8d, config 5: "
                         "(N-1)/N of every rank's entries travel through the all-to-all) instead of one z-slab per rank")
    ap.add_argument("--global-n", type=int, default=int(os.environ.get("ESP_BENCH_GLOBAL_N", "0")),
                    help="STRONG scaling on a fixed global G^3 grid (BASELINE.json configs[4]: --gpus 8 --global-n 512): rank r "
                         "assembles the z-slab of G/N planes it owns; default (0): weak scaling, n x n x (n*N)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch(args, sys.argv[1:]))
    if os.environ.get("ESP_BENCH_LAUNCH_PROBE"):
        raise SystemExit(launch_probe(args))

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if torch.cuda.device_count() < max(1, min(world, local + 1)):
        raise SystemExit("bench.py: rank %d needs GPU %d, the node shows %d" % (rank, local, torch.cuda.device_count()))
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    torch.cuda.set_device(local)
    numa_node = bind_near_gpu(torch, local)

    from esparse_loader import load
    esp = load()
    n = args.n
    E, Z = fd_counts(n)
    sharded = args.sharded or world > 1
    SA = None
    if not sharded:
        N = n ** 3
        A = esp.ExtendableSparseMatrix(N, N, device=local, capacity_hint=E)
        Z_total = Z

        def step():
            A.reset()
            A.generate_fdrand(n, n, n, seed=0x5EED0002, rand_mode=1, kind=esp.ESP_UPDATE)
            A.flush()
    else:
        # weak scaling (default): the global grid is n x n x (n*world); rank r assembles the z-slab of nodes
        # it owns (fixed work per GPU), columns are range-sharded, entries of the cross-slab pairs
        # travel through the all-to-all.  --global-n G: the fixed G^3 grid of BASELINE.json configs[4]
        # (strong scaling), rank r = the z-slab of G / world planes
        if args.global_n:
            gx = gy = nzg = args.global_n
            if nzg % world:
                raise SystemExit("bench.py: --global-n %d is not a multiple of the %d ranks" % (nzg, world))
        else:
            gx = gy = n
            nzg = n * world
        N = gx * gy * nzg
        nodes = N // world
        Eg = 4 * ((gx - 1) * gy * nzg + gx * (gy - 1) * nzg + gx * gy * (nzg - 1)) + 2 * (gy * nzg + gx * nzg + gx * gy)
        E = Eg // world                        # (per rank: the bytes of the roofline lines)
        E_hint = E + 8 * gx * gy               # (... with room for the slab's boundary planes)
        # the C group API: exchange policy and RCCL all-to-all-v (grouped ncclSend/ncclRecv on the library's stream)
        # live inside libesparse_hip.so; torch.distributed only carries the 128-byte id, the barrier and the timing
        uid = [esp.GroupShardedMatrix.unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(uid, src=0)
        SA = esp.GroupShardedMatrix(N, N, nranks=world, rank=rank, device=local, capacity_hint=E_hint, unique_id=uid[0])
        A = SA.local
        Z_total = N + 2 * ((gx - 1) * gy * nzg + gx * (gy - 1) * nzg + gx * gy * (nzg - 1))

        scr_blk = 8 * gx * gy                  # (--scrambled: blocks of 8 z-planes, dealt round-robin)
        if args.scrambled and (nzg % (8 * world)):
            raise SystemExit("bench.py: --scrambled deals blocks of 8 z-planes: %d planes do not divide by 8 x %d ranks" % (nzg, world))

        def step():
            A.reset()
            if args.scrambled:
                for b in range(rank, N // scr_blk, world):
                    A.generate_fdrand_range(gx, gy, nzg, b * scr_blk, (b + 1) * scr_blk, seed=0x5EED0002, rand_mode=1,
                                            kind=esp.ESP_UPDATE)
            else:
                A.generate_fdrand_range(gx, gy, nzg, rank * nodes, (rank + 1) * nodes, seed=0x5EED0002, rand_mode=1,
                                        kind=esp.ESP_UPDATE)
            SA.flush()
    # Timed region: HIP events around the bucket kernel only (level 3; every bracketed kernel costs two event records
    # of about 6 us on the stream).  The per-stage breakdown comes from a few extra, untimed steps at level 1
    # afterwards.  ESP_BENCH_STAGE_TIMING=1/2 brackets all big kernels (2: the small scans too) inside the timed region.
    timing_level = int(os.environ.get("ESP_BENCH_STAGE_TIMING", "3"))
    A.timing_enable(timing_level)
    if os.environ.get("ESP_BENCH_FORCE_PATH"):      # experiments only (see esp_debug_force_path)
        A.debug_force_path(int(os.environ["ESP_BENCH_FORCE_PATH"]))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    A.timing(clear=True)
    # (no cyclic collection inside the timed region -- see extra_configs; every step of the region runs, nothing is skipped)
    import gc
    gc.collect()
    gc.disable()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    sent = SA.sent_off_rank if SA is not None and getattr(SA, "sent_off_rank", None) is not None else 0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        s = torch.tensor([float(sent)], dtype=torch.float64, device="cuda")
        dist.all_reduce(s, op=dist.ReduceOp.MAX)
        sent = int(s.item())
    total_nnz = SA.nnz() if sharded else A.nnz()
    assert total_nnz == Z_total or os.environ.get("ESP_LOCAL_STOP"), (total_nnz, Z_total)   # (ablation runs produce nothing)
    tm = A.timing(clear=True)
    partition_kind = A.debug_last_partition()
    # untimed self-check of what the timed loop produced: the device CSC against the oracle's pin (fd_<n>_m1)
    digest_ok = None
    if not sharded and rank == 0 and not os.environ.get("ESP_LOCAL_STOP") and not os.environ.get("ESP_BENCH_NO_DIGEST"):
        digest_ok = csc_digest_ok(A, "fd_%d_m1" % n, golden_digests())
    breakdown_steps = 0
    tm_all = tm
    if timing_level == 3:   # stage breakdown: separate untimed steps with events around every big kernel
        breakdown_steps = min(3, args.steps)
        A.timing_enable(1)
        for _ in range(breakdown_steps):
            step()
        barrier()
        tm_all = A.timing(clear=True)
    A.timing_enable(0)
    # what the timed steps do NOT pay: they repeat the previous assembly, so the producer reuses its count / ranking tables
    # (esp_handle::GenPlan).  `first_call_ms`: the same step with that reuse switched off (esp_debug_force_path(31): COUNT launch,
    # ranking launches and the host round trip for their flags in every step) -- what the first assembly of a grid costs
    plan_reused = A.debug_last_plan_reused()
    first_call_ms = None
    if not os.environ.get("ESP_BENCH_FORCE_PATH"):
        A.debug_force_path(31)
        step()
        barrier()
        t1 = time.perf_counter()
        for _ in range(5):
            step()
        barrier()
        first_call_ms = (time.perf_counter() - t1) / 5 * 1e3
        A.debug_force_path(0)
    # the consumers of the flushed CSC that stay on the device, on the matrix the timed loop built
    consumers = None
    if rank == 0 and not sharded and not args.no_extra:
        try:
            consumers = consumer_lines(esp, torch, A, N, int(Z_total))
        except Exception as ex:
            consumers = {"error": repr(ex)}
    Z = Z_total / world   # per-rank share of the final nnz (value below multiplies by world)

    out = None
    if rank == 0:
        # dominant kernel = the stage with the largest summed device time
        nsteps_all = breakdown_steps if breakdown_steps else args.steps
        stage_ms = {k: v[0] for k, v in tm_all.items() if isinstance(v, tuple)}
        dom = max(stage_ms, key=stage_ms.get)
        # (its launch duration from the timed region when it was bracketed there, else from the breakdown steps)
        in_timed = isinstance(tm.get(dom), tuple) and tm[dom][1] > 0
        dom_ms, dom_launches = tm[dom] if in_timed else tm_all[dom]
        per_launch_bytes = {
            # algorithmic bytes of ONE launch of the stage's kernel over the E appended entries
            "scatter": 32.0 * E,          # read 16 B (key+value), write 16 B per entry
            "hist": 8.0 * E,              # read the keys
            "append": 16.0 * E,           # write key+value
            "local": 16.0 * E + 16.0 * Z + 8.0 * (N / world + 1),
            "fold": 16.0 * E + 16.0 * Z,
        }.get(dom, 16.0 * E)
        avg_ms = max(dom_ms / max(dom_launches, 1), 1e-9)
        achieved = per_launch_bytes / (avg_ms * 1e-3) / 1e9
        algo_bytes = 2 * 16.0 * E + 16.0 * Z + 8.0 * (N / world + 1)   # 72.08 B per final nnz (per rank)
        ms_step = dt / args.steps * 1e3
        traffic, traffic_round = pmc_traffic(dom) if (n == 256 and not sharded) else (None, None)
        out = {
            "metric": baseline_metric(),
            "value": Z * args.steps * world / dt,
            "unit": "nnz/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": "strong" if (sharded and args.global_n) else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "host_numa_node": numa_node,   # the process runs on the CPUs of the GPU's NUMA node (bind_near_gpu); None: unchanged
            # the timed steps repeat the previous assembly: the producer reused its plan (1) -- first_call_ms is the step without that
            "plan_reused": plan_reused, "first_call_ms": first_call_ms,
            "digest_ok": digest_ok,   # sha256 of the device CSC == the CPU oracle's (tests/golden/digests_large.txt); None: no pin for this size
            "config": {"workload": "fdrand %d^3 Float64/Int64 fresh build: device COO append (the producer writes every "
                                   "update to its radix bucket) -> LDS bucket sort + ordered fold -> CSC "
                                   "(BASELINE.json configs[1])" % n,
                       "n": n, "appended_entries": E, "final_nnz": int(Z),
                       "partition": {1: "run-based single pass in flush!", 2: "8-bit passes in flush!",
                                     4: "producer-side (append = partition)", 7: "shard pieces"}.get(partition_kind, str(partition_kind)),
                       "parallelism": ("column-range shards x%d, all-to-all-v entry routing (RCCL world size %d), %s "
                                       "producers, global grid %dx%dx%d%s, at most %d entries sent off-rank per flush"
                                       % (world, world, "scrambled (blocks of 8 z-planes dealt round-robin)" if args.scrambled else "z-slab",
                                          gx, gy, nzg, " (BASELINE.json configs[4])" if args.global_n == 512 else "", sent))
                                      if sharded else "1 GPU"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_from": ("profiles/pmc_traffic.json (%s)" % traffic_round) if traffic else None,
                         "avg_launch_ms": avg_ms, "launches": dom_launches,
                         "events": "timed region" if in_timed else "breakdown steps after the timed region",
                         "algorithmic_bytes_per_launch": per_launch_bytes},
            "pipeline": {"algorithmic_bytes_per_step": algo_bytes, "bytes_per_final_nnz": algo_bytes / Z,
                         "achieved_GBs": algo_bytes / (ms_step * 1e-3) / 1e9,
                         "frac_of_hbm_peak": algo_bytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "stage_ms_per_step": {k: v / nsteps_all for k, v in stage_ms.items()},
                         "stage_ms_from": ("%d untimed steps after the timed region, events around every big kernel"
                                           % breakdown_steps) if breakdown_steps else "timed region",
                         "flush_ms_per_step": tm["flush_ms"] / max(tm["flushes"], 1)},
        }
    if rank == 0 and world == 1 and not sharded and not args.no_extra:
        del A
        out["extra"] = {"configs": extra_configs(esp, torch, local, n, int(os.environ.get("ESP_CFG4_2D", "3163")),
                                                 int(os.environ.get("ESP_CFG4_3D", "216")))}
    if rank == 0 and consumers is not None:
        out.setdefault("extra", {})["consumers"] = consumers
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(args.cpu_sample_n, args.cpu_mt_n)
    if rank == 0:
        out["summary"] = summary_of(out)   # LAST key: the driver's record keeps the tail of the line
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        # RCCL prints a version banner through C stdio: flush it first so the JSON is the last line
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
