#!/bin/bash
# extra.configs of bench.py only (no CPU baseline, no digests): ms + stage breakdown per config
ESP_BENCH_NO_DIGEST=${ESP_BENCH_NO_DIGEST:-1} timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('headline ms/step %.3f' % d['ms_per_step'], {k: round(v,3) for k,v in d['pipeline']['stage_ms_per_step'].items() if v>0})
for k,v in d.get('extra',{}).get('configs',{}).items():
    print(k, round(v.get('ms',0),3), v.get('stage_ms'), v.get('digest_ok'))"
