#!/bin/bash
# round 4: esp_append_host with six-byte keys over PCIe (default for batches of one kind) against eight (ESP_HOST_KEYS8=1)
timeout 1200 python -m pytest tests -m gpu -x -q -k "append or host or golden or assembly or updates or mixed or i32 or commit or stage" > gpurun_out/six_pytest.log 2>&1; echo pytest_rc=$?; tail -2 gpurun_out/six_pytest.log
for rep in 1 2 3 4; do
 for k8 in 0 1; do
  if [ $k8 = 0 ]; then unset ESP_HOST_KEYS8; else export ESP_HOST_KEYS8=1; fi
  ESP_EXTRA_ONLY=cfg2 timeout 600 python tools/r4_extra.py 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if not l.startswith('{'): continue
    d=json.loads(l)
    v=d.get('cfg2_host')
    if v: print('keys8=$k8', round(v['ms'],1), 'append+flush', round(v['append_flush_ms'],1), 'get_csc', round(v['get_csc_ms'],1), 'nnz/s %.3g' % v['nnz_per_s'])
"
 done
done
