#!/bin/bash
# routed fold experiments: parity subset, config 3 + pure re-assembly timings
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config3 or tail or fuzz or stored or reassembly or routed or plus" > gpurun_out/r3_f_tests.log 2>&1; grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r3_f_tests.log | tail -4
timeout 600 python tools/bench_configs.py 3 2>&1 | tail -1
timeout 600 python tools/reasm_bench.py 2>&1 | tail -2
