#!/bin/bash
# round 5: 4-byte keys out of the last planned radix pass: the new test, the partition tests, config 4's triplet lines
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "short_keys or nine_bit or shuffled or triplet or mixed or golden or assembly or group or tier or append_device or large_keys or more_than_32 or tail" > gpurun_out/r5_k32_pytest.log 2>&1; echo pytest_rc=$?; tail -5 gpurun_out/r5_k32_pytest.log
ESP_EXTRA_ONLY=cfg4 ESP_CFG4_ONLY_TRIPLETS=1 timeout 900 python tools/r4_extra.py 3 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    for k,v in d.items():
        if 'triplets' in k: print(k, {kk:(round(vv,3) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ('ms','frac_of_hbm_peak','digest_ok','stage_ms','error','partition')})
"
