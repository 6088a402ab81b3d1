#!/bin/bash
# round 5: the randomised parity fuzz with other seeds (fused element kernels, plan reuse, rebuild, first pass at append)
cd $GRAFT_REPO_ROOT
for s in ${SEEDS:-501 502 503}; do
  ESP_FUZZ_FOCUS=elements timeout 400 python3 tests/fuzz_parity.py ${SECS:-90} $s 2>&1 | tail -3 | cut -c1-600
done
for s in ${SEEDS2:-601 602 603}; do
  timeout 400 python3 tests/fuzz_parity.py ${SECS:-90} $s 2>&1 | tail -3 | cut -c1-600
done
