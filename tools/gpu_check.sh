#!/bin/bash
# usage: tools/gpu_check.sh [bench args]  -- GPU parity tests + bench summary (run through gpurun)
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo pytest_rc=$?; tail -4 gpurun_out/pytest_gpu.log
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/bench.log 2>&1; echo bench_rc=$?
tail -1 gpurun_out/bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('nnz/s %.4g  ms/step %.3f' % (d['value'], d['ms_per_step'])); print({k: round(v,3) for k,v in d['pipeline']['stage_ms_per_step'].items()}); print('roofline', d['roofline']['kernel'], round(d['roofline']['frac'],3), 'pipeline frac', round(d['pipeline']['frac_of_hbm_peak'],4))" || tail -20 gpurun_out/bench.log
