#!/usr/bin/env python3
"""Round 4: bench.py's extra.configs by themselves (ESP_EXTRA_ONLY=cfg2,cfg3,cfg4 picks; default cfg4), one JSON object per line.
usage: python tools/r4_extra.py [steps]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("ESP_EXTRA_ONLY", "cfg4")
import torch  # noqa: E402

torch.cuda.init()
import bench  # noqa: E402
from esparse_loader import load  # noqa: E402

bench.bind_near_gpu(torch, 0)   # (as bench.py's main does: the process on the GPU's NUMA node)
esp = load()
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
out = bench.extra_configs(esp, torch, 0, 256, int(os.environ.get("ESP_CFG4_2D", "3163")), int(os.environ.get("ESP_CFG4_3D", "216")), steps=steps)
for k, v in out.items():
    print(json.dumps({k: v}))
