#!/bin/bash
# round 6: cfg_mt_sum's line (path taken, split, digest) three times in separate processes, then the sum / mt parity tests
mkdir -p gpurun_out
for i in 1 2 3; do
ESP_EXTRA_ONLY=cfg_mt_sum timeout 900 python tools/r4_extra.py 5 2>gpurun_out/r6_sum_err_$i.log | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    for k,v in d.items():
        if isinstance(v,dict) and ('ms' in v or 'error' in v): print(k, {kk:(round(vv,3) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk not in ('workload',)})
"
done
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "flush_sum or mt_ or sum" > gpurun_out/r6_sum_pytest.log 2>&1; echo pytest_rc=$?; grep -E "passed|failed|Error|assert" gpurun_out/r6_sum_pytest.log | tail -5
