#!/bin/bash
# usage: tools/r6_shard.sh   -- the column-shard step on one GPU: the shard tests, bench.py's cfg5 line (10 steps), the ordered trace of one step
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "shard or group or piece or abi_host" 2>&1 | tail -3
ESP_EXTRA_ONLY=cfg5 python3 tools/r4_extra.py 10 2>/dev/null | cut -c1-700
bash tools/r6_trace.sh cfg5 fdrand_part_k > /dev/null 2>&1; cat gpurun_out/r6_trace_cfg5.txt
