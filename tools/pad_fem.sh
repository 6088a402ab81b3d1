#!/bin/bash
# occupancy sensitivity of the group-tier bucket kernel on 3-D FEM: LDS padded so that ONE workgroup fits a CU instead of two
cd $GRAFT_REPO_ROOT
cp extendablesparse.jl_amd/libesparse_hip.so /tmp/keep.so
echo "== two workgroups per CU (as shipped)"; timeout 600 python tools/bench_configs.py 4b 2>&1 | grep config | cut -c100-420
ESP_EXTRA_FLAGS=-DESP_LOCAL_PAD=40000 python extendablesparse.jl_amd/build.py --force > /dev/null 2>&1
echo "== one workgroup per CU (40 KB of padding)"; timeout 600 python tools/bench_configs.py 4b 2>&1 | grep config | cut -c100-420
cp /tmp/keep.so extendablesparse.jl_amd/libesparse_hip.so
