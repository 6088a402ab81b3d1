#!/bin/bash
# round 3, step A: group tier of the bucket kernel -- parity, then config 4 with and without it
timeout 1200 python -m pytest tests -m gpu -x -q -k "group_tier or fem or tiers or fuzz or 24_input or long_duplicate" > gpurun_out/r3a_pytest.log 2>&1; echo pytest_rc=$?; tail -5 gpurun_out/r3a_pytest.log
for fp in 0 24; do
  echo "force_path $fp"
  ESP_BENCH_FORCE_PATH=$fp timeout 600 python tools/bench_configs.py 4a 4b 2>/dev/null | grep "^{" | python -c "
import sys,json
for ln in sys.stdin:
    d=json.loads(ln); print(d['config'], round(d['ms_per_step'],2), d['stage_ms'])"
done
