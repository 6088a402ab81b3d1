#!/bin/bash
# round 4: the randomised parity fuzz with other seeds, elements / Base.sum focus first (run through gpurun)
cd $GRAFT_REPO_ROOT
for s in ${SEEDS:-101 102 103}; do
  ESP_FUZZ_FOCUS=elements timeout 400 python3 tests/fuzz_parity.py ${SECS:-100} $s 2>&1 | tail -3 | cut -c1-600
done
for s in ${SEEDS2:-201 202}; do
  timeout 400 python3 tests/fuzz_parity.py ${SECS:-100} $s 2>&1 | tail -3 | cut -c1-600
done
