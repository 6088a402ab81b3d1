#!/bin/bash
# round 4: wave_k builds against each other (ab/lib_X.so), ESP_WAVE=1
export ESP_WAVE=1
for rep in 1 2; do
  for v in "$@"; do
    cp ab/lib_$v.so extendablesparse.jl_amd/libesparse_hip.so
    python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['pipeline']['stage_ms_per_step']; print('$v', round(d['ms_per_step'],3), {k: round(x,3) for k,x in s.items() if x>0}, d.get('digest_ok'))"
  done
done
