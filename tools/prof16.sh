cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof16
ESP_BENCH_FORCE_PATH=16 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof16 -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra > gpurun_out/prof16.log 2>&1
find gpurun_out/prof16 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/kernel_stats16.csv
rm -rf gpurun_out/prof16
tail -1 gpurun_out/prof16.log | cut -c1-200
head -14 gpurun_out/kernel_stats16.csv | cut -c1-160
