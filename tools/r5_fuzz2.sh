#!/bin/bash
# round 5: hunt a non-deterministic mismatch: the element-focused fuzz, several seeds, every MISMATCH line kept
cd $GRAFT_REPO_ROOT
for s in ${SEEDS:-503 503 701 702 703 704}; do
  ESP_FUZZ_FOCUS=elements timeout 300 python3 tests/fuzz_parity.py ${SECS:-150} $s 2>&1 | grep -E "MISMATCH|FAILED|fuzz ok|Error" | cut -c1-500
done
