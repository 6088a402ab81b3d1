#!/bin/bash
# same-box A/B of library builds on config 3 (re-assembly): tools/ab_cfg3.sh X Y  (ab/lib_X.so ...)
for rep in 1 2; do
  for v in "$@"; do
    cp ab/lib_$v.so extendablesparse.jl_amd/libesparse_hip.so
    python tools/bench_configs.py 3 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step_incl_first_build'],2), d['stage_ms'])"
  done
done
