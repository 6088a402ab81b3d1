cd $GRAFT_REPO_ROOT
cp extendablesparse.jl_amd/libesparse_hip.so /tmp/keep.so
ESP_EXTRA_FLAGS=-DESP_LOCAL_STAMPS python extendablesparse.jl_amd/build.py --force > /dev/null 2>&1
echo "== config 3: the tail's bucket kernel"
ESP_STAMP_CFG3=1 python tools/local_stamps.py 2>&1 | tail -22
echo "== re-assembly: the batch's bucket kernel (all hits)"
ESP_STAMP_REASM=1 python tools/local_stamps.py 2>&1 | tail -22
cp /tmp/keep.so extendablesparse.jl_amd/libesparse_hip.so
