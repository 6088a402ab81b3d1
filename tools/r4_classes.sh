#!/bin/bash
# round 4: eight ticket counters (ESP_TICKET_CLASSES=1) against one, headline + extra configs 3 / 4
for rep in 1 2 3; do
 for c in 0 1; do
  if [ $c = 0 ]; then unset ESP_TICKET_CLASSES; else export ESP_TICKET_CLASSES=1; fi
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['pipeline']['stage_ms_per_step']; print('classes=$c', round(d['ms_per_step'],3), {k: round(x,3) for k,x in s.items() if x>0}, d.get('digest_ok'))"
 done
done
