#!/bin/bash
# round 5: full GPU check: all gpu tests, then the default bench line (summary key printed)
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r5_pytest_gpu.log 2>&1; echo pytest_rc=$?; tail -15 gpurun_out/r5_pytest_gpu.log
timeout 900 python bench.py > gpurun_out/r5_bench.json 2> gpurun_out/r5_bench.err; echo bench_rc=$?
tail -1 gpurun_out/r5_bench.json | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print(json.dumps(d['summary'], indent=0))
" || tail -5 gpurun_out/r5_bench.err
