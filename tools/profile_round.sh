cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof gpurun_out/pmc_f gpurun_out/pmc_w
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra > gpurun_out/prof_bench.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_f -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_w -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/pmc_w.log 2>&1
python3 tools/pmc_traffic.py gpurun_out/pmc_f gpurun_out/pmc_w gpurun_out/pmc_traffic.json ${1:-r2} > /dev/null
find gpurun_out/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/kernel_stats.csv
rm -rf gpurun_out/prof gpurun_out/pmc_f gpurun_out/pmc_w
tail -1 gpurun_out/prof_bench.log | cut -c1-300
head -12 gpurun_out/kernel_stats.csv
