#!/bin/bash
# per-launch durations of the partition kernels of config 4a (2-D FEM, shuffled): tools/trace_cfg4.sh [4a|4b]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/trc4
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trc4 -- python3 tools/bench_configs.py ${1:-4a} > gpurun_out/trc4.log 2>&1
python3 - <<'P'
import csv, glob
rows = []
for f in glob.glob('gpurun_out/trc4/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:60], r.get('Grid_Size', r.get('Grid_Size_X', ''))))
rows.sort()
last = [r for r in rows if any(k in r[2] for k in ('scatter_k', 'tile_hist_k', 'local_k', 'fem_k'))][-12:]
for s, e, n, g in last:
    print('%9.1f us  grid %s  %s' % ((e - s) / 1e3, g, n))
P
rm -rf gpurun_out/trc4
