#!/bin/bash
# round 5: config 4's lines without the triplets (tools/r4_extra.py), product build
ESP_EXTRA_ONLY=cfg4 ESP_BENCH_SKIP_TRIPLETS=1 timeout 900 python tools/r4_extra.py ${1:-3} 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    for k,v in d.items():
        if isinstance(v,dict) and 'ms' in v: print(k, {kk:(round(vv,3) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ('ms','digest_ok','stage_ms','error')})
"
