#!/bin/bash
# round 5, final tree: the parity fuzz, all focuses, fresh seeds
cd $GRAFT_REPO_ROOT
for s in 981 982; do ESP_FUZZ_FOCUS=elements timeout 300 python3 tests/fuzz_parity.py 100 $s 2>&1 | grep -E "MISMATCH|FAILED|fuzz ok|Error|fault" | cut -c1-200; done
for s in 983 984 985; do timeout 300 python3 tests/fuzz_parity.py 100 $s 2>&1 | grep -E "MISMATCH|FAILED|fuzz ok|Error|fault" | cut -c1-200; done
for s in 986; do ESP_FUZZ_FOCUS=k32 timeout 300 python3 tests/fuzz_parity.py 100 $s 2>&1 | grep -E "MISMATCH|FAILED|fuzz ok|Error|fault" | cut -c1-200; done
