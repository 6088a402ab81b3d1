#!/bin/bash
# round 5, final: the parity fuzz in all three focuses, fresh seeds
cd $GRAFT_REPO_ROOT
for s in 901 902; do ESP_FUZZ_FOCUS=elements timeout 300 python3 tests/fuzz_parity.py 110 $s 2>&1 | grep -E "MISMATCH|FAILED|fuzz ok|Error|fault" | cut -c1-160; done
for s in 911 912 913; do timeout 300 python3 tests/fuzz_parity.py 110 $s 2>&1 | grep -E "MISMATCH|FAILED|fuzz ok|Error|fault" | cut -c1-160; done
for s in 921 922; do ESP_FUZZ_FOCUS=k32 timeout 300 python3 tests/fuzz_parity.py 110 $s 2>&1 | grep -E "MISMATCH|FAILED|fuzz ok|Error|fault" | cut -c1-160; done
