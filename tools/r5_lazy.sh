#!/bin/bash
# round 5: the fused bucket kernel (group3_items.hpp): FEM / elements parity subset, then config 4's timings
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fem or elements or lazy or golden or group3 or sum or mt_wrapper or generic_wrapper" > gpurun_out/r5_lazy_pytest.log 2>&1; echo pytest_rc=$?; tail -12 gpurun_out/r5_lazy_pytest.log
ESP_EXTRA_ONLY=cfg4 timeout 900 python tools/r4_extra.py 3 2> gpurun_out/r5_lazy_extra.err | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    for k,v in d.items(): print(k, {kk:(round(vv,3) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ('ms','frac_of_hbm_peak','digest_ok','stage_ms','error','vs_generator')})
"
tail -3 gpurun_out/r5_lazy_extra.err
