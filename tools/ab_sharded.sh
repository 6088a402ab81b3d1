#!/bin/bash
# A/B of library builds (ab/lib_X.so) on the one-rank column-shard step: [ESP_AB_N=322] tools/ab_sharded.sh X Y ...
for rep in 1 2; do
  for v in "$@"; do
    cp ab/lib_$v.so extendablesparse.jl_amd/libesparse_hip.so
    python bench.py --n ${ESP_AB_N:-256} --steps 30 --warmup 4 --no-cpu-baseline --no-extra --sharded 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', 'n', d['config'].get('n', '?'), round(d['ms_per_step'],3), 'key_bytes', d.get('pipeline',{}).get('key_bytes'), d.get('digest_ok'))"
  done
done
