#!/bin/bash
# round 4: persistent wave kernel -- workgroups per CU sweep
export ESP_WAVE_DEBUG=1
for pc in 0 2 3 4 5; do
  if [ $pc = 0 ]; then unset ESP_WAVE_PER_CU; else export ESP_WAVE_PER_CU=$pc; fi
  timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra 2>gpurun_out/wave_bench_err.log | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['pipeline']['stage_ms_per_step']; print('per_cu=$pc', round(d['ms_per_step'],3), {k: round(x,3) for k,x in s.items() if x>0}, d.get('digest_ok'))" || tail -5 gpurun_out/wave_bench_err.log
  grep wave_k gpurun_out/wave_bench_err.log | head -1
done
