"""The sharded flush over REAL RCCL between two GPUs (esp_group_* with nranks = 2, one process per GPU).

Skipped on a box with fewer than two GPUs -- every box this build has had; the exchange policy runs on the CPU under
sanitizers (test_group_policy.py) and the RCCL transport on one GPU through the loop-back hook (test_gpu_parity.py::
test_group_rccl_loopback*), so this test is what remains to be run on a multi-GPU node: two ranks assemble the z-slabs of
one stencil, exchange the cross-slab entries through ncclSend / ncclRecv, and the stitched CSC must equal -- bit for bit --
what ONE GPU builds from the whole stream (the property the driver's SCALE runs rely on)."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import hashlib, json, os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["ESP_ROOT"])
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(rank)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", rank))
from esparse_loader import load
esp = load()
n, rounds = 96, 3
nzg = n * world
N = n * n * nzg
uid = [esp.GroupShardedMatrix.unique_id() if rank == 0 else None]
dist.broadcast_object_list(uid, src=0)
SA = esp.GroupShardedMatrix(N, N, nranks=world, rank=rank, device=rank, unique_id=uid[0])
nodes = n * n * n
kinds = []
for rnd in range(rounds):                       # first flush: the flush's own partition pass; then the producer's
    SA.local.reset()
    SA.local.generate_fdrand_range(n, n, nzg, rank * nodes, (rank + 1) * nodes, seed=0x5EED0002, rand_mode=1)
    SA.flush()
    kinds.append(SA.last_exchange)
c0, c1, cp, rv, nz = SA.local_slice()
h = hashlib.sha256()
for a in ((cp[:-1] if rank + 1 < world else cp), rv, nz):
    h.update(np.ascontiguousarray(a).tobytes())
parts = [None] * world
dist.all_gather_object(parts, (h.hexdigest(), int(len(rv)), kinds, int(SA.sent_off_rank)))
if rank == 0:
    A = esp.ExtendableSparseMatrix(N, N, device=0)
    A.generate_fdrand(n, n, nzg, seed=0x5EED0002, rand_mode=1)
    A.flush()
    cpa, rva, nza = A.sparse().arrays()
    ok = True
    for r in range(world):
        lo, hi = esp.owner_ranges(N, world)[r]
        a0, a1 = cpa[lo] - 1, cpa[hi] - 1
        hh = hashlib.sha256()
        seg = cpa[lo:hi] if r + 1 < world else cpa[lo:hi + 1]
        for a in (seg, rva[a0:a1], nza[a0:a1]):
            hh.update(np.ascontiguousarray(a).tobytes())
        ok = ok and hh.hexdigest() == parts[r][0] and parts[r][1] == a1 - a0
    print(json.dumps({"ok": bool(ok), "kinds": [p[2] for p in parts], "sent": [p[3] for p in parts]}), flush=True)
dist.destroy_process_group()
'''


def _gpus():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 0


@pytest.mark.gpu
@pytest.mark.skipif(_gpus() < 2, reason="needs two GPUs on one node (RCCL between processes)")
def test_two_ranks_over_rccl_equal_one_gpu(tmp_path):
    from procdist import free_port, run_processes
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = str(free_port())
    cmds, envs = [], []
    for r in range(2):
        envs.append(dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                         HSA_ENABLE_IPC_MODE_LEGACY="0", ESP_ROOT=ROOT))
        cmds.append([sys.executable, str(script)])
    # (supervised: a rank that dies at start-up ends the other one instead of leaving it in the rendezvous; nothing stays
    # behind on the GPUs)
    outs = run_processes(cmds, envs, timeout=900)
    assert all(o[0] == 0 for o in outs), [(o[0], o[2][-2000:]) for o in outs]
    d = json.loads([ln for ln in outs[0][1].splitlines() if ln.startswith("{")][-1])
    assert d["ok"], d
    assert all(k[-1] == "partitioned" for k in d["kinds"]), d      # the slab streams take the partitioned exchange
    assert all(s > 0 for s in d["sent"]), d                       # ... and cross-slab entries really travelled
