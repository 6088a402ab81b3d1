#!/bin/bash
# round 5, after the short-key passes: more seeds of the general and k32 focuses
cd $GRAFT_REPO_ROOT
for s in 941 942 943 944; do timeout 300 python3 tests/fuzz_parity.py 110 $s 2>&1 | grep -E "MISMATCH|FAILED|fuzz ok|Error|fault" | cut -c1-300; done
for s in 951 952 953; do ESP_FUZZ_FOCUS=k32 timeout 300 python3 tests/fuzz_parity.py 110 $s 2>&1 | grep -E "MISMATCH|FAILED|fuzz ok|Error|fault" | cut -c1-300; done
