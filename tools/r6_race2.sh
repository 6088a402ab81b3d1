#!/bin/bash
# round 6: the fuzz repro (seed 503, cases 0..6, folds on threads, HIP_LAUNCH_BLOCKING=1) with checksum traces of every append / flush
export ESP_EXTRA_FLAGS=-DESP_EXPERIMENTS
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
R=${R:-12}
for i in $(seq 1 $R); do
  env ESP_SUM_THREADS=1 HIP_LAUNCH_BLOCKING=1 ${EXTRA_ENV} ESP_FUZZ_MAXCASES=7 ESP_FUZZ_FOCUS=elements timeout 300 python3 tests/fuzz_parity.py 100 503 > gpurun_out/race_$i.log 2>&1
  if grep -q "fuzz ok" gpurun_out/race_$i.log; then echo "run $i ok"; else echo "run $i FAIL $(grep -v amdgpu.ids gpurun_out/race_$i.log | grep -E 'MISMATCH|Error|fault|abort' | head -3 | cut -c1-200)"; fi
done
