#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4; do
  ESP_FUZZ_FOCUS=elements timeout 300 python3 tests/fuzz_parity.py ${SECS:-100} ${SEED:-503} > gpurun_out/fz_$i.log 2>&1
  grep -v amdgpu.ids gpurun_out/fz_$i.log | grep -E "MISMATCH|FAILED|Error|fuzz ok" | cut -c1-400
  grep -q "fuzz ok" gpurun_out/fz_$i.log || { grep -v amdgpu.ids gpurun_out/fz_$i.log | tail -25 | cut -c1-300; }
done
