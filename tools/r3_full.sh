#!/bin/bash
# full GPU check: all gpu tests, then the default bench line
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r3_pytest_gpu.log 2>&1; echo pytest_rc=$?; tail -6 gpurun_out/r3_pytest_gpu.log
timeout 900 python bench.py > gpurun_out/r3_bench.json 2> gpurun_out/r3_bench.err; echo bench_rc=$?
tail -1 gpurun_out/r3_bench.json | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('value %.4g ms/step %.3f digest_ok %s' % (d['value'], d['ms_per_step'], d.get('digest_ok')))
print('roofline', d['roofline']['kernel'], round(d['roofline']['frac'],4), 'pipeline', round(d['pipeline']['frac_of_hbm_peak'],4))
for k,v in d.get('extra',{}).get('configs',{}).items():
    print(k, {kk: (round(vv,3) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ('ms','frac_of_hbm_peak','digest_ok','stage_ms','error')})
print('cpu', d.get('cpu_baseline',{}).get('value'))
" || tail -5 gpurun_out/r3_bench.err
