#!/bin/bash
# round 6: the original repro of the concurrent-handles corruption (tools/r5_race.sh: seed 503, cases 0..6 of the element fuzz,
# esp_flush_sum's folds on one host thread per buffer) under runtime switches that separate host threading from device co-residency
export ESP_EXTRA_FLAGS=-DESP_EXPERIMENTS
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
run() {  # env, tag, reps
  ok=0; bad=0
  for i in $(seq 1 $3); do
    env $1 ESP_FUZZ_MAXCASES=${CASES:-7} ESP_FUZZ_FOCUS=elements timeout 200 python3 tests/fuzz_parity.py 100 503 > gpurun_out/fzab.log 2>&1
    if grep -q "fuzz ok" gpurun_out/fzab.log; then ok=$((ok+1)); else bad=$((bad+1)); echo "$2: FAIL $(grep -v amdgpu.ids gpurun_out/fzab.log | grep -E 'MISMATCH|Error|fault|abort|POISON' | head -3 | cut -c1-300)"; fi
  done
  echo "$2: ok $ok fail $bad"
}
R=${R:-12}
run "ESP_SUM_THREADS=1" threads $R
run "ESP_SUM_THREADS=1 AMD_SERIALIZE_KERNEL=3" threads_serialize_kernel $R
run "ESP_SUM_THREADS=1 GPU_MAX_HW_QUEUES=1" threads_one_hw_queue $R
run "ESP_SUM_THREADS=1 HIP_LAUNCH_BLOCKING=1" threads_launch_blocking $R
