#!/bin/bash
# usage: tools/env_ab.sh "VAR=a" "VAR=b" ...  -- same-box A/B of environment settings (3 interleaved bench runs each)
for rep in 1 2 3; do
  for v in "$@"; do
    env $v python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['pipeline']['stage_ms_per_step']; print('$v', round(d['ms_per_step'],3), {k: round(x,3) for k,x in s.items() if x>0})"
  done
done
