#!/bin/bash
# round 5: the short-key passes' tests + the fuzz seed that found the invalid launch + two fresh general seeds
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "short_keys or nine_bit or large_keys or more_than_32" > gpurun_out/r5_k32b_pytest.log 2>&1; echo pytest_rc=$?; grep -E "passed|failed|Error|assert" gpurun_out/r5_k32b_pytest.log | tail -5
for s in 911 931 932; do timeout 300 python3 tests/fuzz_parity.py 110 $s 2>&1 | grep -E "MISMATCH|FAILED|fuzz ok|Error|fault" | cut -c1-300; done
