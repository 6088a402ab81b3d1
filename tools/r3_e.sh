#!/bin/bash
# round 3, join experiments: parity of the staged join + the tail read where it lies, then config 3 A/B
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config3 or tail or fuzz or merge or join or stored or reassembly or shard" > gpurun_out/r3_e_tests.log 2>&1; grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r3_e_tests.log | tail -15
for ct in 1 0 256; do
  echo "== ESP_CT=$ct"
  ESP_CT=$ct timeout 600 python tools/bench_configs.py 3 2>&1 | tail -1
done
echo "== ESP_CT=0 path 29 (tail copied to the front)"
ESP_DEBUG_FORCE_PATH=29 timeout 600 python tools/bench_configs.py 3 2>&1 | tail -1
