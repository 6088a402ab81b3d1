#!/bin/bash
# config 4 timings + the FEM / group-tier parity subset
timeout 600 python tools/bench_configs.py 4a 4b 2>&1 | grep config | cut -c100-420
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fem or item or group or bucket or long_runs or fuzz" 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -3
