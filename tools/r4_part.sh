#!/bin/bash
# round 4: the stencil producer's PART launch with persistent workgroups (ESP_PART_PERSIST = workgroups per CU) against one per tile
ESP_PART_PERSIST=4 timeout 900 python -m pytest tests -m gpu -x -q -k "fdrand or golden or stream or shard or group" 2>&1 | tail -2
for rep in 1 2 3; do
 for pc in 0 4 5; do
  if [ $pc = 0 ]; then unset ESP_PART_PERSIST; else export ESP_PART_PERSIST=$pc; fi
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['pipeline']['stage_ms_per_step']; print('part_persist=$pc', round(d['ms_per_step'],3), {k: round(x,3) for k,x in s.items() if x>0}, d.get('digest_ok'))"
 done
done
