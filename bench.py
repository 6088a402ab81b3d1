#!/usr/bin/env python3
"""bench.py -- assembled+flushed nnz/s on the 256^3 7-point stencil (BASELINE.json config 2).

A "step" is one fresh assembly of fdrand(Float64,n,n,n; matrixtype=ExtendableSparseMatrix)
(src/matrix/sprand.jl:226-256): the update stream is produced on the device straight into the COO
append buffer (inputs resident in HBM, no PCIe in the timed region), then flush! builds the CSC.
value = nnz of the resulting CSC x steps x ranks / wall time (max over ranks).

Prints ONE JSON line (rank 0).  Extra objects: "roofline" (dominant kernel, hipEvent-timed on the
library's stream), "pipeline" (whole step against the compulsory bytes of SURVEY.md section 8d),
"cpu_baseline" (the C oracle timed on this box's host, bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def pmc_traffic(kernel_stage):
    """HBM bytes per launch of the stage's kernel from the committed rocprofv3 PMC passes
    (profiles/pmc_traffic.json, produced by tools/pmc_traffic.py from separate FETCH_SIZE and
    WRITE_SIZE runs of this same command, gfx950 corrections applied); None if not collected."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            return json.load(f).get(kernel_stage, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def fd_counts(n):
    E = 12 * n * n * (n - 1) + 6 * n * n
    Z = n ** 3 + 6 * n * n * (n - 1)
    return E, Z


def cpu_baseline(sample_n):
    """Reference algorithm (LNK insert via updateindex! + lnk+csc flush) on the host, 1 thread."""
    from oracle import oracle as orc
    z, ti, tf = orc.bench_fdrand(sample_n, sample_n, sample_n, orc.KIND_UPDATE)
    return {"value": z / (ti + tf), "unit": "nnz/s", "cores": 1, "kind": "port",
            "sample": "one fdrand %d^3 fresh assemble+flush! (%d update calls, %d nnz), updateindex! style, "
                      "C restatement of SparseMatrixLNK + lnk+csc (oracle/), -O2, insert %.2fs + flush %.2fs; "
                      "host has %d cores, 1 used (the reference path is single-threaded)"
                      % (sample_n, 12 * sample_n * sample_n * (sample_n - 1) + 6 * sample_n * sample_n, z, ti, tf,
                         os.cpu_count())}


def baseline_metric():
    """BASELINE.json's metric string (the line reports its nnz/s part as `value`; the HBM GB/s and %-of-peak
    part is in `roofline` and `pipeline`)."""
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "BASELINE.json")) as f:
            return json.load(f)["metric"]
    except Exception:
        return "assembled+flushed nnz/sec and HBM GB/s %peak, 256^3 7-pt stencil, 1/2/4/8 GPU"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", type=int, default=int(os.environ.get("ESP_BENCH_N", "256")))
    ap.add_argument("--cpu-sample-n", type=int, default=int(os.environ.get("ESP_BENCH_CPU_N", "256")))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sharded", action="store_true",
                    help="use the column-shard exchange path even on one GPU (always used for --gpus > 1)")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    torch.cuda.set_device(local)

    from esparse_loader import load
    esp = load()
    n = args.n
    E, Z = fd_counts(n)
    sharded = args.sharded or world > 1
    if not sharded:
        N = n ** 3
        A = esp.ExtendableSparseMatrix(N, N, device=local, capacity_hint=E)
        Z_total = Z

        def step():
            A.reset()
            A.generate_fdrand(n, n, n, seed=0x5EED0002, rand_mode=1, kind=esp.ESP_UPDATE)
            A.flush()
    else:
        # weak scaling: the global grid is n x n x (n*world); rank r assembles the z-slab of nodes
        # it owns (fixed work per GPU), columns are range-sharded, entries of the cross-slab pairs
        # travel through the all-to-all
        if dist is None:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", local))
        nzg = n * world
        N = n * n * nzg
        be = esp.HipShardBackend(N, N, device=local, capacity_hint=E + 4 * n * n)
        # the small agreements of a flush travel over a gloo group (host memory, loopback: one node): a device
        # collective would sit behind the partition's scatter kernel
        ctrl = None
        if not os.environ.get("ESP_BENCH_NO_CTRL_GROUP"):
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
            try:
                ctrl = dist.new_group(backend="gloo")
            except Exception as ex:   # (same outcome on every rank of the node: fall back to the data group)
                print("bench: no gloo control group (%s), using the RCCL group" % ex, file=sys.stderr)
                ctrl = None
        SA = esp.ShardedExtendableSparseMatrix(N, N, be, ctrl_group=ctrl)
        A = be.matrix
        Z_total = N + 2 * ((n - 1) * n * nzg + n * (n - 1) * nzg + n * n * (nzg - 1))
        nodes = n ** 3

        def step():
            A.reset()
            A.generate_fdrand_range(n, n, nzg, rank * nodes, (rank + 1) * nodes, seed=0x5EED0002, rand_mode=1,
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
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    total_nnz = SA.nnz() if sharded else A.nnz()
    assert total_nnz == Z_total or os.environ.get("ESP_LOCAL_STOP"), (total_nnz, Z_total)   # (ablation runs produce nothing)
    tm = A.timing(clear=True)
    breakdown_steps = 0
    tm_all = tm
    if timing_level == 3:   # stage breakdown: separate untimed steps with events around every big kernel
        breakdown_steps = min(3, args.steps)
        A.timing_enable(1)
        for _ in range(breakdown_steps):
            step()
        barrier()
        tm_all = A.timing(clear=True)
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
            "local": 16.0 * E + 16.0 * Z + 8.0 * (N + 1),
            "fold": 16.0 * E + 16.0 * Z,
        }.get(dom, 16.0 * E)
        avg_ms = max(dom_ms / max(dom_launches, 1), 1e-9)
        achieved = per_launch_bytes / (avg_ms * 1e-3) / 1e9
        algo_bytes = 2 * 16.0 * E + 16.0 * Z + 8.0 * (N + 1)   # 72.08 B per final nnz
        ms_step = dt / args.steps * 1e3
        out = {
            "metric": baseline_metric(),
            "value": Z * args.steps * world / dt,
            "unit": "nnz/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "fdrand %d^3 Float64/Int64 fresh build: device COO append -> stable radix "
                                   "partition -> ordered fold -> CSC (BASELINE.json configs[1])" % n,
                       "n": n, "appended_entries": E, "final_nnz": int(Z),
                       "parallelism": ("column-range shards x%d, all-to-all-v entry routing (RCCL), z-slab "
                                       "producers, global grid %dx%dx%d" % (world, n, n, n * world)) if sharded else "1 GPU"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(dom) if (n == 256 and not sharded) else None,
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
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample_n)
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
