#!/bin/bash
# round 4: wave_k with 4 / 8 / 16 segments (= waves) per workgroup and ticket
export ESP_WAVE=1
for rep in 1 2; do
 for sg in 4 8 16; do
  export ESP_WAVE_SEGS=$sg
  timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra 2>gpurun_out/wave_bench_err.log | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['pipeline']['stage_ms_per_step']; print('segs=$sg', round(d['ms_per_step'],3), {k: round(x,3) for k,x in s.items() if x>0}, d.get('digest_ok'))" || tail -3 gpurun_out/wave_bench_err.log
 done
done
