#!/bin/bash
# 322^3 / 400^3 (more than 32 key bits below the default plan's prefix) against 256^3: ns per appended entry
for n in 256 322 400; do
  ESP_BENCH_NO_DIGEST=1 timeout 900 python bench.py --n $n --steps 8 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
E=d['config']['appended_entries']
print('n=$n ms/step %.3f  ns per entry %.4f key_bytes %s' % (d['ms_per_step'], d['ms_per_step']*1e6/E, d.get('pipeline',{}).get('key_bytes')), {k: round(v,3) for k,v in d['pipeline']['stage_ms_per_step'].items() if v>0})"
done
