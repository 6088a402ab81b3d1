#!/bin/bash
# round 5: rocprofv3 kernel stats of config 4's assemblies only (tools/r4_extra.py, ESP_EXTRA_ONLY=cfg4; triplets skipped)
TAG=${1:-r5}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof4
export ESP_EXTRA_ONLY=cfg4 ESP_BENCH_SKIP_TRIPLETS=1 ESP_BENCH_NO_DIGEST=1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof4 -- python3 tools/r4_extra.py 3 > gpurun_out/prof4.log 2>&1
find gpurun_out/prof4 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${TAG}_cfg4_kernel_stats.csv
rm -rf gpurun_out/prof4
head -30 gpurun_out/${TAG}_cfg4_kernel_stats.csv | cut -c1-150
