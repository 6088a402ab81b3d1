#!/bin/bash
cd $GRAFT_REPO_ROOT
cp extendablesparse.jl_amd/libesparse_hip.so /tmp/keep.so
for fl in "-DESP_NO_TOUCH" ""; do
  echo "== flags: $fl"
  ESP_EXTRA_FLAGS="$fl" python extendablesparse.jl_amd/build.py --force > /dev/null 2>&1
  timeout 600 python tools/reasm_bench.py 2>&1 | tail -1
  timeout 600 python tools/bench_configs.py 3 2>&1 | tail -1
done
for fl in "-DESP_NO_TOUCH -DESP_LOCAL_STAMPS" "-DESP_LOCAL_STAMPS"; do
  echo "== stamps, flags: $fl"
  ESP_EXTRA_FLAGS="$fl" python extendablesparse.jl_amd/build.py --force > /dev/null 2>&1
  ESP_STAMP_REASM=1 python tools/local_stamps.py 2>&1 | tail -10
done
cp /tmp/keep.so extendablesparse.jl_amd/libesparse_hip.so
