#!/bin/bash
# round 4: the ticket counter on a line of its own (ESP_TICKET_FAR=1) against beside the granules, wave kernel and local_k
cp ab/lib_cur.so extendablesparse.jl_amd/libesparse_hip.so
for rep in 1 2; do
 for far in 0 1; do
  if [ $far = 1 ]; then export ESP_TICKET_FAR=1; else unset ESP_TICKET_FAR; fi
  for nw in 0 1; do
    if [ $nw = 1 ]; then export ESP_NO_WAVE=1; else unset ESP_NO_WAVE; fi
    python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['pipeline']['stage_ms_per_step']; print('far=$far nowave=$nw', round(d['ms_per_step'],3), {k: round(x,3) for k,x in s.items() if x>0}, d.get('digest_ok'))"
  done
 done
done
