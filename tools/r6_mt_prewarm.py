#!/usr/bin/env python3
"""Round 6: is the FIRST GPU process on a fresh box slow in cfg_mt_sum's 16-thread fills (stalls of 5 .. 67 ms in some rounds) because
the shared libraries it maps are not in the page cache yet?  Reads every file the process has mapped (after torch + the library are
loaded), then runs the config.  usage: python tools/r6_mt_prewarm.py [0|1 = read the mapped files first] [steps]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["ESP_EXTRA_ONLY"] = "cfg_mt_sum"
import torch  # noqa: E402

torch.cuda.init()
import bench  # noqa: E402
from esparse_loader import load  # noqa: E402

bench.bind_near_gpu(torch, 0)
esp = load()
if len(sys.argv) > 1 and sys.argv[1] == "1":
    t0 = time.perf_counter()
    n = bench.warm_page_cache()
    print("read %.0f MB of mapped files in %.2f s" % (n / 1e6, time.perf_counter() - t0))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
out = bench.extra_configs(esp, torch, 0, 256, 3163, 216, steps=steps)
v = out["cfg_mt_sum"]
print(json.dumps({k: v[k] for k in ("ms", "fills_ms", "fills_ms_steps", "ms_steps")}))
