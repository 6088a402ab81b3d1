cd $GRAFT_REPO_ROOT
cp extendablesparse.jl_amd/libesparse_hip.so /tmp/keep.so
ESP_EXTRA_FLAGS=-DESP_LOCAL_STAMPS python extendablesparse.jl_amd/build.py --force > /dev/null 2>&1
ESP_STAMP_FEM=${ESP_STAMP_FEM:-120} python tools/local_stamps.py 2>&1 | tail -22
cp /tmp/keep.so extendablesparse.jl_amd/libesparse_hip.so
