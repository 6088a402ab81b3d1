#!/bin/bash
# round 4: the wave-per-segment bucket kernel (wavecols.hpp, opt-in: ESP_WAVE=1) -- fdrand parity subset with it, then the
# headline step with and without it on the same box
ESP_WAVE=1 timeout 900 python -m pytest tests -m gpu -x -q -k "fdrand or golden or stream or assembly or wave_kernel" --deselect "tests/test_gpu_parity.py::test_fdrand_full_size_digest[256]" > gpurun_out/wave_pytest.log 2>&1; echo pytest_rc=$?; tail -3 gpurun_out/wave_pytest.log
for rep in 1 2; do
  for wv in 1 0; do
    if [ $wv = 1 ]; then export ESP_WAVE=1; else unset ESP_WAVE; fi
    timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra 2>gpurun_out/wave_bench_err.log | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['pipeline']['stage_ms_per_step']; print('wave=$wv', round(d['ms_per_step'],3), {k: round(x,3) for k,x in s.items() if x>0}, d['roofline']['kernel'][:40], round(d['roofline']['frac'],3), d.get('digest_ok'))" || tail -5 gpurun_out/wave_bench_err.log
  done
done
