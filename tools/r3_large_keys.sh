#!/bin/bash
# per-entry cost at sizes whose key bits below the prefix exceed 32
for args in "--n 256" "--n 322" "--sharded --n 256" "--sharded --n 322"; do
  echo "== $args"
  ESP_BENCH_NO_DIGEST=1 timeout 900 python bench.py $args --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('ms/step %.3f value %.4g' % (d['ms_per_step'], d['value']), {k: round(v,3) for k,v in d['pipeline']['stage_ms_per_step'].items() if v>0}, d['config'].get('workload','')[:80], 'key_bytes', d.get('extra',{}).get('key_bytes'))"
done
