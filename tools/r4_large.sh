#!/bin/bash
# round 4: 322^3 (33 key bits below the default plan's prefix: packed keys) with the default plan and with the wave kernel's
# finer plan (ESP_WAVE=1: 31 bits below its prefix -- 4-byte keys), against 256^3
for w in 0 1; do
 if [ $w = 1 ]; then export ESP_WAVE=1; else unset ESP_WAVE; fi
 for n in 256 322 400; do
  ESP_BENCH_NO_DIGEST=1 timeout 900 python bench.py --n $n --steps 8 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
E=d['config']['appended_entries']
print('wave=$w n=$n ms/step %.3f  ns per entry %.4f' % (d['ms_per_step'], d['ms_per_step']*1e6/E), {k: round(v,3) for k,v in d['pipeline']['stage_ms_per_step'].items() if v>0})"
 done
done
