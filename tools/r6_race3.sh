#!/bin/bash
# round 6: the ORIGINAL repro (seed 503, cases 0..6 of the element fuzz, esp_flush_sum's folds on one host thread per buffer,
# HIP_LAUNCH_BLOCKING=1: 4 of 12 runs failed before the fix) on the fixed tree, then the new tests
export ESP_EXTRA_FLAGS=-DESP_EXPERIMENTS
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
ok=0; bad=0
for i in $(seq 1 ${R:-16}); do
  env ESP_SUM_THREADS=1 HIP_LAUNCH_BLOCKING=1 ESP_FUZZ_MAXCASES=7 ESP_FUZZ_FOCUS=elements timeout 200 python3 tests/fuzz_parity.py 100 503 > gpurun_out/fzab.log 2>&1
  if grep -q "fuzz ok" gpurun_out/fzab.log; then ok=$((ok+1)); else bad=$((bad+1)); echo "FAIL $(grep -v amdgpu.ids gpurun_out/fzab.log | grep -E 'MISMATCH|Error|fault|abort' | head -3 | cut -c1-300)"; fi
done
echo "fuzz seed 503 threads + launch blocking: ok $ok fail $bad"
unset ESP_EXTRA_FLAGS
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
python -m pytest tests/test_concurrent_handles.py -q -m gpu 2>&1 | tail -5
