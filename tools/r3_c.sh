#!/bin/bash
# round 3, step C: bucket producer for shuffled FEM streams -- parity, then config 4
timeout 1500 python -m pytest tests -m gpu -x -q -k "item_producer or fem or fuzz" > gpurun_out/r3c_pytest.log 2>&1; echo pytest_rc=$?; tail -5 gpurun_out/r3c_pytest.log
for bb in 0; do
echo "bucket bits $bb"
ESP_BUCKET_BITS=$bb timeout 600 python tools/bench_configs.py 4a 4b 2>/dev/null | grep "^{" | python -c "
import sys,json
for ln in sys.stdin:
    d=json.loads(ln); print(d['config'], round(d['ms_per_step'],2), d['stage_ms'])"
done
