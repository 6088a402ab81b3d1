#!/bin/bash
# round 4: esp_append_elements at config 4's sizes with and without the cell records (same box) (run through gpurun)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export ESP_BENCH_NO_DIGEST=1 ESP_BENCH_SKIP_TRIPLETS=1
for v in "" 1; do
  echo "== ESP_ELEM_NO_CELLREC=$v"
  ESP_ELEM_NO_CELLREC=$v python3 tools/r4_extra.py 3 2>&1 | grep cfg4 | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    for k, v in d.items():
        print(k, 'ms %.3f' % v['ms'], v['stage_ms'], 'digest', v['digest_ok'])"
done
