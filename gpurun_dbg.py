import sys, os
import torch
torch.cuda.init()
import torch.distributed as dist
os.environ["MASTER_ADDR"]="127.0.0.1"; os.environ["MASTER_PORT"]="29544"
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda",0))
for cnt in (1<<24, 1<<26, 100_000_000, 1<<27, 150_000_000, 200_933_376, 1<<28):
    a = torch.arange(cnt, dtype=torch.int64, device="cuda")
    b = torch.zeros_like(a)
    dist.all_to_all_single(b, a, [cnt], [cnt])
    torch.cuda.synchronize()
    ok = bool(torch.equal(a, b))
    nbad = int((a != b).sum())
    first = int((a != b).nonzero()[0]) if nbad else -1
    print(cnt, "int64 ok" if ok else "int64 BAD nbad=%d first=%d" % (nbad, first))
    # no split sizes
    b.zero_()
    dist.all_to_all_single(b, a)
    torch.cuda.synchronize()
    print(cnt, "nosplit", bool(torch.equal(a, b)))
dist.destroy_process_group()
