import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
torch.cuda.init()
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from esparse_loader import load
esp = load()
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import sharded_model  # noqa: E402
n = 256; N = n**3; E = 12*n*n*(n-1)+6*n*n
be = sharded_model.HipShardBackend(N, N, device=0, capacity_hint=E + 4*n*n)
SA = sharded_model.ShardedExtendableSparseMatrix(N, N, be)
A = be.matrix
T = {}
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); T[name] = T.get(name, 0) + time.perf_counter() - t0; return r
    setattr(obj, name, g)
for nm in ("part_partition", "part_assemble", "flush"): wrap(be, nm)
wrap(SA, "_gather_ints")
def step():
    A.reset(); A.generate_fdrand_range(n, n, n, 0, N, seed=1, rand_mode=1, kind=esp.ESP_UPDATE); SA.flush()
for _ in range(3): step()
torch.cuda.synchronize(); T.clear()
t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print("ms/step", tot*100, {k: round(v*100, 3) for k, v in T.items()})
dist.destroy_process_group()
