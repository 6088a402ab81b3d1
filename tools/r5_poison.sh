#!/bin/bash
# round 5 experiment (-DESP_EXPERIMENTS build on the box): released device buffers are poisoned and kept (ESP_POISON_FREE=1) -- a
# stale READ shows as a parity failure, a stale WRITE as a POISON line at the next esp_destroy.  The parity fuzz, all focuses, short,
# and the parity tests that release and re-allocate the most.
export ESP_EXTRA_FLAGS=-DESP_EXPERIMENTS
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export ESP_POISON_FREE=1
show() { grep -E "POISON|MISMATCH|FAILED|fuzz ok|Error|fault|Killed|Traceback" $1 | sort | uniq -c | cut -c1-300 | head -8; }
for s in 503 971; do echo "elements $s"; ESP_FUZZ_FOCUS=elements timeout 400 python3 tests/fuzz_parity.py 60 $s > gpurun_out/poison_e$s.log 2>&1; echo "rc=$?"; show gpurun_out/poison_e$s.log; done
for s in 972 973 975; do echo "general $s"; timeout 400 python3 tests/fuzz_parity.py 60 $s > gpurun_out/poison_g$s.log 2>&1; echo "rc=$?"; show gpurun_out/poison_g$s.log; done
for s in 974; do echo "k32 $s"; ESP_FUZZ_FOCUS=k32 timeout 400 python3 tests/fuzz_parity.py 60 $s > gpurun_out/poison_k$s.log 2>&1; echo "rc=$?"; show gpurun_out/poison_k$s.log; done
echo pytest; timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "not fuzz" > gpurun_out/poison_pytest.log 2>&1; echo "rc=$?"; grep -E "^FAILED|^ERROR|passed|failed|POISON: [0-9]+ words" gpurun_out/poison_pytest.log | sort | uniq -c | head -30
