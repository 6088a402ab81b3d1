#!/bin/bash
# round 4: the headline through group3_k<2, 6> (39 KiB of LDS: four workgroups per CU) against local_k's small variant
for rep in 1 2; do
 for g in 0 1; do
  if [ $g = 0 ]; then unset ESP_G3_SMALL; else export ESP_G3_SMALL=1; fi
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['pipeline']['stage_ms_per_step']; print('g3small=$g', round(d['ms_per_step'],3), {k: round(x,3) for k,x in s.items() if x>0}, d.get('digest_ok'), d.get('host_numa_node'))"
 done
done
