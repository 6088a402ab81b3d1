#!/bin/bash
for fp in 0 26; do
echo "force $fp"
ESP_BENCH_FORCE_PATH=$fp timeout 600 python tools/bench_configs.py 4a 2>/dev/null | grep "^{" | python -c "
import sys,json
for ln in sys.stdin:
    d=json.loads(ln); print(d['config'], round(d['ms_per_step'],2), d['stage_ms'])"
done
