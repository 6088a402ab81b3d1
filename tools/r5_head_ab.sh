#!/bin/bash
# round 5: the headline loop of this tree (new) against the tree at the start of the second session (ab/old, built here), on one box, alternating
run() { (cd $1 && ESP_BENCH_NO_DIGEST=1 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$2 ms/step %.4f bucket kernel %.4f ms' % (d['ms_per_step'], d['roofline']['avg_launch_ms']))"); }
for i in 1 2 3; do run ab/old old; run . new; done
