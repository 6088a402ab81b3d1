#!/bin/bash
# A/B of library builds (ab/lib_X.so) on bench.py's extra configs: [ESP_AB_STEPS=n] ESP_EXTRA_ONLY=cfg2|cfg3|cfg4 tools/ab_cfg.sh X Y
for rep in 1 2; do
  for v in "$@"; do
    cp ab/lib_$v.so extendablesparse.jl_amd/libesparse_hip.so
    ESP_BENCH_SKIP_TRIPLETS=1 ESP_BENCH_SKIP_HOST=1 timeout 900 python tools/r4_extra.py ${ESP_AB_STEPS:-3} 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if not l.startswith('{'): continue
    d=json.loads(l)
    for k,v in d.items():
        if isinstance(v,dict) and 'ms' in v: print('$v', k, round(v['ms'],3), {kk: round(x,3) for kk,x in v.get('stage_ms',{}).items()}, v.get('digest_ok'))
"
  done
done
