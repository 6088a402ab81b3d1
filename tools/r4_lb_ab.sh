#!/bin/bash
# round 4: look-back group size A/B (ab/lib_lb8.so: groups of 256 tickets, ab/lib_lb10.so: 1024), with and without the wave kernel
for rep in 1 2; do
  for v in "$@"; do
    cp ab/lib_$v.so extendablesparse.jl_amd/libesparse_hip.so
    for nw in ${NWLIST:-0 1}; do
      if [ $nw = 1 ]; then export ESP_NO_WAVE=1; else unset ESP_NO_WAVE; fi
      python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | \
        python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['pipeline']['stage_ms_per_step']; print('$v nowave=$nw', round(d['ms_per_step'],3), {k: round(x,3) for k,x in s.items() if x>0}, d.get('digest_ok'))"
    done
  done
done
