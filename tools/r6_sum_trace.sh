#!/bin/bash
# usage: tools/r6_sum_trace.sh [p] [entries per buffer]  -- kernel statistics of esp_flush_sum's general path (tools/r6_sum_threads.py)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/st
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/st -- python3 tools/r6_sum_threads.py ${1:-16} ${2:-2000000} > gpurun_out/st.log 2>&1
grep -v amdgpu gpurun_out/st.log | tail -2
f=$(find gpurun_out/st -name "*kernel_stats.csv" | head -1)
head -16 "$f" | cut -c1-170
rm -rf gpurun_out/st
