#!/bin/bash
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fem or item" > gpurun_out/r3_i_tests.log 2>&1; grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r3_i_tests.log | tail -4
timeout 600 python tools/bench_configs.py 4a 4b 2>&1 | grep config | cut -c1-400
