#!/bin/bash
# round 4: local_k with persistent workgroups (ESP_PERSIST = workgroups per CU) against one workgroup per segment
for rep in 1 2; do
 for pc in 0 3 4; do
  if [ $pc = 0 ]; then unset ESP_PERSIST; else export ESP_PERSIST=$pc; fi
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['pipeline']['stage_ms_per_step']; print('persist=$pc', round(d['ms_per_step'],3), {k: round(x,3) for k,x in s.items() if x>0}, d.get('digest_ok'))"
 done
done
