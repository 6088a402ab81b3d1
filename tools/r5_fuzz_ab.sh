#!/bin/bash
# seed 503 of the element-focused fuzz: round 4's tree (ab/r4tree) against this one, and this one with the AUTO-gated features off
cd $GRAFT_REPO_ROOT
run() {  # dir, env, tag
  for i in 1 2 3; do
    (cd $1 && env $2 ESP_FUZZ_FOCUS=elements timeout 200 python3 tests/fuzz_parity.py ${SECS:-45} ${SEED:-503} > $GRAFT_REPO_ROOT/gpurun_out/fzab.log 2>&1)
    if grep -q "fuzz ok" gpurun_out/fzab.log; then echo "$3 run $i: ok $(grep -o 'cases [0-9]*' gpurun_out/fzab.log)"; else echo "$3 run $i: FAIL $(grep -v amdgpu.ids gpurun_out/fzab.log | grep -E 'MISMATCH|Error|fault|abort' | head -2 | cut -c1-260)"; fi
  done
}
run ab/r4tree "A=1" r4
run . "A=1" r5
run . "ESP_DEBUG_FORCE_PATH=36" r5_force36
