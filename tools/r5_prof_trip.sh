#!/bin/bash
# round 5: rocprofv3 kernel stats of config 4's triplet lines only (tools/r4_extra.py, ESP_EXTRA_ONLY=cfg4 ESP_CFG4_ONLY_TRIPLETS=1)
TAG=${1:-r5e}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof5
export ESP_EXTRA_ONLY=cfg4 ESP_CFG4_ONLY_TRIPLETS=1 ESP_BENCH_NO_DIGEST=1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof5 -- python3 tools/r4_extra.py 3 > gpurun_out/prof5.log 2>&1
find gpurun_out/prof5 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${TAG}_triplets_kernel_stats.csv
rm -rf gpurun_out/prof5
grep -o '"cfg4_generic_triplets_.d": {[^}]*}[^}]*}' gpurun_out/prof5.log | cut -c1-400
grep -v "at::native\|rocclr" gpurun_out/${TAG}_triplets_kernel_stats.csv | head -24 | cut -c1-170
