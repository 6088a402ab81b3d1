#!/bin/bash
# A/B of library builds (ab/lib_X.so) on the headline step and cfg2_generic_append / cfg3 when asked: tools/ab_headline.sh X Y [extra]
for rep in 1 2 3; do
  for v in "$1" "$2"; do
    cp ab/lib_$v.so extendablesparse.jl_amd/libesparse_hip.so
    python bench.py --steps ${ESP_AB_STEPS:-10} --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['pipeline']['stage_ms_per_step']; print('$v', round(d['ms_per_step'],3), {k: round(x,3) for k,x in s.items() if x>0}, d.get('digest_ok'))"
  done
done
