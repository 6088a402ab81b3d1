#!/bin/bash
# same-box A/B of library builds on config 4a (2-D FEM): tools/ab_cfg4.sh X Y
for rep in 1 2; do
  for v in "$@"; do
    cp ab/lib_$v.so extendablesparse.jl_amd/libesparse_hip.so
    python tools/bench_configs.py 4a 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],2), d['stage_ms'])"
  done
done
