#!/bin/bash
# round 5 experiment (-DESP_EXPERIMENTS build on the box): seed 503, cases 0..6 of the element-focused fuzz, esp_flush_sum's folds on
# one host thread per buffer (ESP_SUM_THREADS=1: round 4's form) -- as it is, and with released buffers kept instead of freed
# (ESP_POISON_FREE=1: no hipFree while the threads run)
export ESP_EXTRA_FLAGS=-DESP_EXPERIMENTS
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
run() {  # env, tag, reps
  ok=0; bad=0
  for i in $(seq 1 $3); do
    env $1 ESP_FUZZ_MAXCASES=7 ESP_FUZZ_FOCUS=elements timeout 200 python3 tests/fuzz_parity.py 100 503 > gpurun_out/fzab.log 2>&1
    if grep -q "fuzz ok" gpurun_out/fzab.log; then ok=$((ok+1)); else bad=$((bad+1)); echo "$2: FAIL $(grep -v amdgpu.ids gpurun_out/fzab.log | grep -E 'MISMATCH|Error|fault|abort|POISON' | head -2 | cut -c1-200)"; fi
  done
  echo "$2: ok $ok fail $bad"
}
run "ESP_SUM_THREADS=1" threads 8
run "ESP_SUM_THREADS=1 ESP_POISON_FREE=1" threads_no_free 8
run "A=1" serial 4
