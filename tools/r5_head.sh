#!/bin/bash
# round 5: the generator's plan reuse: parity subset, headline step, sharded step on one GPU
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "generator or fdrand or golden or producer or plan or stream" > gpurun_out/r5_head_pytest.log 2>&1; echo pytest_rc=$?; tail -4 gpurun_out/r5_head_pytest.log
for f in "" "--sharded"; do
timeout 600 python bench.py --no-extra --no-cpu-baseline $f 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$f', 'ms/step %.4f' % d['ms_per_step'], 'digest', d.get('digest_ok'), 'roofline', d['roofline']['kernel'], round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['frac'],4), {k: round(v,4) for k,v in d['pipeline']['stage_ms_per_step'].items() if v>0})"
done
