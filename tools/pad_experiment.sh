cd $GRAFT_REPO_ROOT
cp extendablesparse.jl_amd/libesparse_hip.so /tmp/keep.so
for pad in 0 12000; do
  if [ $pad = 0 ]; then cp /tmp/keep.so extendablesparse.jl_amd/libesparse_hip.so; else ESP_EXTRA_FLAGS=-DESP_LOCAL_PAD=$pad python extendablesparse.jl_amd/build.py --force > /dev/null 2>&1; touch extendablesparse.jl_amd/csrc/local.hpp; ESP_EXTRA_FLAGS=-DESP_LOCAL_PAD=$pad python -c "
import importlib.util,os
spec=importlib.util.spec_from_file_location('b','extendablesparse.jl_amd/build.py'); m=importlib.util.module_from_spec(spec); spec.loader.exec_module(m); m.build(force=True)"; fi
  python bench.py --no-extra --no-cpu-baseline --steps 10 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('pad $pad', d['ms_per_step'], d['pipeline']['stage_ms_per_step']['local'])"
done
cp /tmp/keep.so extendablesparse.jl_amd/libesparse_hip.so
