#!/bin/bash
# A/B of library builds on ONE GPU box (boxes differ by several percent, never compare across runs):
#   1. build variant X here, copy extendablesparse.jl_amd/libesparse_hip.so to ab/lib_X.so (repeat per variant)
#   2. gpurun -- tools/ab_bench.sh X Y Z
# Runs bench.py three times per variant, interleaved, and prints ms/step and the bucket-kernel time.
for rep in 1 2 3; do
  for v in "$@"; do
    cp ab/lib_$v.so extendablesparse.jl_amd/libesparse_hip.so
    python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['pipeline']['stage_ms_per_step']; print('$v', round(d['ms_per_step'],3), {k: round(x,3) for k,x in s.items() if x>0})"
  done
done
