#!/bin/bash
# round 6: parity fuzz campaign on the final tree: general / k32 / elements focuses, 150 s each; Base.sum over per-entry buffers, 90 s each
cd $GRAFT_REPO_ROOT
for s in 6101 6102 6103; do timeout 400 python3 tests/fuzz_parity.py 150 $s 2>&1 | grep -E "MISMATCH|FAILED|fuzz ok|Error|fault" | cut -c1-200; done
for s in 6201 6202; do ESP_FUZZ_FOCUS=k32 timeout 400 python3 tests/fuzz_parity.py 150 $s 2>&1 | grep -E "MISMATCH|FAILED|fuzz ok|Error|fault" | cut -c1-200; done
for s in 6301 6302 503; do ESP_FUZZ_FOCUS=elements timeout 400 python3 tests/fuzz_parity.py 150 $s 2>&1 | grep -E "MISMATCH|FAILED|fuzz ok|Error|fault" | cut -c1-200; done
for s in 7001 7002; do ESP_FUZZ_FOCUS=sum timeout 400 python3 tests/fuzz_parity.py 90 $s 2>&1 | grep -E "MISMATCH|FAILED|fuzz ok|Error|fault" | cut -c1-200; done
