#!/bin/bash
# longer randomised parity fuzz against the oracle (several seeds, both focuses)
for s in 401 402 403; do
  timeout 400 python tests/fuzz_parity.py 90 $s 2>&1 | tail -2
  ESP_FUZZ_FOCUS=k32 timeout 400 python tests/fuzz_parity.py 60 $((s+50)) 2>&1 | tail -2
done
