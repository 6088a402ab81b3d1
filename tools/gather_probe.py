"""Where does the consensus all-gather of a sharded flush wait?  Times the steps of _gather_ints while the
partition's scatter kernel runs on the handle's stream (world size 1 over RCCL)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
torch.cuda.init()
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29545")
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from esparse_loader import load
esp = load()
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import sharded_model  # noqa: E402
n = 256; N = n**3; E = 12*n*n*(n-1)+6*n*n
be = sharded_model.HipShardBackend(N, N, device=0, capacity_hint=E + 4*n*n)
SA = sharded_model.ShardedExtendableSparseMatrix(N, N, be)
A = be.matrix
dev = be.device
pin = torch.empty(8, dtype=torch.int64).pin_memory()
acc = {}
def lap(name, t0):
    t = time.perf_counter(); acc[name] = acc.get(name, 0) + t - t0; return t
for it in range(8):
    A.reset(); A.generate_fdrand_range(n, n, n, 0, N, seed=1, rand_mode=1, kind=esp.ESP_UPDATE)
    if it < 3:
        SA.flush(); continue
    torch.cuda.synchronize(); A.synchronize()
    t = time.perf_counter()
    part = be.part_partition(1, 0, E); t = lap("part_partition (returns while the scatter runs)", t)
    if it >= 5:
        mine = torch.tensor([1, 2, 3], dtype=torch.int64, device=dev); t = lap("torch.tensor H2D from a list (iterations 5-7)", t)
    else:
        pin[3:6] = torch.tensor([1, 2, 3]); mine = torch.empty(3, dtype=torch.int64, device=dev)
        mine.copy_(pin[3:6], non_blocking=True); t = lap("pinned H2D, non-blocking (iterations 3-4)", t)
    out = [torch.empty_like(mine)]
    dist.all_gather(out, mine); t = lap("all_gather call", t)
    st = torch.stack(out); t = lap("stack", t)
    if it % 2:
        r = st.cpu(); t = lap("cpu() (odd iterations)", t)
    else:
        pin[:3].copy_(st[0], non_blocking=True); torch.cuda.current_stream().synchronize(); t = lap("pinned copy + stream sync (even iterations)", t)
    be.part_wait(); t = lap("part_wait", t)
    A._d.ck(A._d.lib.esp_clear_pending(A._d.h)); A._touch()
print({k: round(v * 1e3, 3) for k, v in acc.items()}, "(ms summed over iterations 3-7)")
dist.destroy_process_group()
