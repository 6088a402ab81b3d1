#!/bin/bash
# round 5: config 3 with the rebuild (flush_rebuild): parity subset + timing
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config3 or reassembly or tail or batch or routed or csc_plus" > gpurun_out/r5_cfg3_pytest.log 2>&1; echo pytest_rc=$?; tail -6 gpurun_out/r5_cfg3_pytest.log
ESP_EXTRA_ONLY=cfg3 timeout 600 python tools/r4_extra.py 3 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    for k,v in d.items(): print(k, {kk:(round(vv,3) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ('ms','frac_of_hbm_peak','digest_ok','stage_ms','error','partition')})
"
