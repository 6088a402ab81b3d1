#!/bin/bash
# round 4: group3_k publishes a RAWUPDATE segment's total right after the sort (default) against after the fold (ESP_LATE_TOTAL=1)
timeout 900 python -m pytest tests -m gpu -x -q -k "fem or group or elements or golden" > gpurun_out/early_pytest.log 2>&1; echo pytest_rc=$?; tail -3 gpurun_out/early_pytest.log
for rep in 1 2; do
 for late in 0 1; do
  if [ $late = 0 ]; then unset ESP_LATE_TOTAL; else export ESP_LATE_TOTAL=1; fi
  ESP_EXTRA_ONLY=cfg4 ESP_BENCH_SKIP_TRIPLETS=1 timeout 600 python tools/r4_extra.py 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if not l.startswith('{'): continue
    d=json.loads(l)
    for k,v in d.items():
        if isinstance(v,dict) and 'ms' in v: print('late=$late', k, round(v['ms'],3), {kk: round(x,3) for kk,x in v.get('stage_ms',{}).items()}, v.get('digest_ok'))
"
 done
done
