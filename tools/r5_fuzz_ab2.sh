#!/bin/bash
# seed 503, cases 0..6 of the element-focused fuzz, repeated: as it is / with esp_flush_sum's folds one after the other
cd $GRAFT_REPO_ROOT
run() {  # env, tag, reps
  ok=0; bad=0
  for i in $(seq 1 $3); do
    env $1 ESP_FUZZ_MAXCASES=7 ESP_FUZZ_FOCUS=elements timeout 200 python3 tests/fuzz_parity.py 100 503 > gpurun_out/fzab.log 2>&1
    if grep -q "fuzz ok" gpurun_out/fzab.log; then ok=$((ok+1)); else bad=$((bad+1)); echo "$2: FAIL $(grep -v amdgpu.ids gpurun_out/fzab.log | grep -E 'MISMATCH|Error|fault|abort' | head -1 | cut -c1-200)"; fi
  done
  echo "$2: ok $ok fail $bad"
}
run "A=1" as_is 8

