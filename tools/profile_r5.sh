#!/bin/bash
# round 5 evidence (same recipe as profile_r4.sh): kernel statistics + PMC traffic of the headline loop and of the extra configs (run through gpurun).
# Every rocprofv3 call runs under `timeout` (a counter pass that aborts must not sit on the box) and the counter passes
# collect ONE counter each.
TAG=${1:-r5}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof gpurun_out/pmc_f gpurun_out/pmc_w gpurun_out/profc gpurun_out/pmcc_f gpurun_out/pmcc_w
export ESP_BENCH_NO_DIGEST=1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra > gpurun_out/prof_bench.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_f -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/pmc_f.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_w -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/pmc_w.log 2>&1
python3 tools/pmc_traffic.py gpurun_out/pmc_f gpurun_out/pmc_w gpurun_out/pmc_traffic.json $TAG > /dev/null
find gpurun_out/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${TAG}_kernel_stats.csv
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/profc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/profc_bench.log 2>&1
find gpurun_out/profc -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${TAG}_configs_kernel_stats.csv
export ESP_BENCH_SKIP_TRIPLETS=1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmcc_f -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmcc_f.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmcc_w -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmcc_w.log 2>&1
python3 tools/pmc_kernels.py gpurun_out/pmcc_f gpurun_out/pmcc_w gpurun_out/${TAG}_configs_pmc_traffic.json $TAG | head -40
rm -rf gpurun_out/prof gpurun_out/pmc_f gpurun_out/pmc_w gpurun_out/profc gpurun_out/pmcc_f gpurun_out/pmcc_w
tail -1 gpurun_out/prof_bench.log | cut -c1-200
head -12 gpurun_out/${TAG}_kernel_stats.csv | cut -c1-160
head -45 gpurun_out/${TAG}_configs_kernel_stats.csv | cut -c1-160
