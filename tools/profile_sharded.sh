#!/bin/bash
# rocprofv3 kernel stats of the column-shard path on one GPU (bench.py --sharded) -> gpurun_out/kernel_stats_sharded.csv
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/profs
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/profs -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --sharded > gpurun_out/profs_bench.log 2>&1
find gpurun_out/profs -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/kernel_stats_sharded.csv
rm -rf gpurun_out/profs
grep -E '^\{' gpurun_out/profs_bench.log | tail -1 | cut -c1-260
head -14 gpurun_out/kernel_stats_sharded.csv | cut -c1-200
