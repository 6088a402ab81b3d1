#!/bin/bash
# per-kernel times of config 3 (rocprofv3 kernel stats), with and without the short-column fold
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
cp extendablesparse.jl_amd/libesparse_hip.so /tmp/keep.so
for fl in ""; do
  echo "== flags: $fl"
  if [ -n "$fl" ]; then ESP_EXTRA_FLAGS="$fl" python extendablesparse.jl_amd/build.py --force > /dev/null 2>&1; fi
  rm -rf gpurun_out/profh
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/profh -- python3 tools/bench_configs.py 3 > gpurun_out/profh.log 2>&1
  find gpurun_out/profh -name "*kernel_stats.csv" | head -1 | xargs -I{} python3 -c "
import csv,sys
for r in csv.DictReader(open('{}')):
    nm=r['Name']
    if 'local_k' in nm or 'colmerge' in nm:
        print(nm[:90], r['Calls'], 'avg_us', round(float(r['AverageNs'])/1e3,1))
"
done
rm -rf gpurun_out/profh
cp /tmp/keep.so extendablesparse.jl_amd/libesparse_hip.so
