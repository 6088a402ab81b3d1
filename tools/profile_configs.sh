#!/bin/bash
# kernel statistics of the extra configs (BASELINE.json configs 3 and 4) of bench.py (run through gpurun)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/profc
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/profc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/profc_bench.log 2>&1
find gpurun_out/profc -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/configs_kernel_stats.csv
rm -rf gpurun_out/profc
head -24 gpurun_out/configs_kernel_stats.csv | cut -c1-200
