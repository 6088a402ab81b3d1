#!/bin/bash
# A/B of library builds (ab/lib_X.so) on cfg_mt_sum's split: tools/ab_mt.sh X Y ...
for rep in 1 2 3; do
  for v in "$@"; do
    cp ab/lib_$v.so extendablesparse.jl_amd/libesparse_hip.so
    ESP_EXTRA_ONLY=cfg_mt_sum python3 tools/r4_extra.py 10 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        for k, v in json.loads(l).items():
            if k == 'cfg_mt_sum': print('$v', k, {kk: round(v[kk], 2) for kk in ('ms', 'fills_ms', 'folds_ms', 'combine_ms')}, v.get('digest_ok'))
"
  done
done
