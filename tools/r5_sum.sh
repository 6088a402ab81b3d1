#!/bin/bash
# round 5: Base.sum (esp_flush_sum): its parity tests, two fuzz seeds of the element focus (elem_sum cases), cfg_mt_sum's line
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "flush_sum or mt_ or sum" > gpurun_out/r5_sum_pytest.log 2>&1; echo pytest_rc=$?; grep -E "passed|failed|Error|assert" gpurun_out/r5_sum_pytest.log | tail -5
for s in 961 962; do ESP_FUZZ_FOCUS=elements timeout 300 python3 tests/fuzz_parity.py 100 $s 2>&1 | grep -E "MISMATCH|FAILED|fuzz ok|Error|fault" | cut -c1-300; done
ESP_EXTRA_ONLY=cfg_mt_sum timeout 900 python tools/r4_extra.py 5 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    for k,v in d.items():
        if isinstance(v,dict) and 'ms' in v: print(k, {kk:(round(vv,3) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ('ms','nnz_ok','plugin_fresh_ms','plugin_same_pattern_ms','error','final_nnz')})
"
