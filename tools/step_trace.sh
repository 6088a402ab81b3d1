#!/bin/bash
# usage: tools/step_trace.sh [bench args, e.g. --sharded]
# Ordered kernel trace of the LAST bench step (name, start offset us, duration us, gap to previous us) -> gpurun_out/step_trace.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/trc
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/trc -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/trc.log 2>&1
python3 - <<'P'
import csv, glob
rows = []
for f in glob.glob('gpurun_out/trc/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:70]))
for f in glob.glob('gpurun_out/trc/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', '')))
rows.sort()
# last step = between the last two launches of the producer's first kernel (COUNT launch, or the plain producer)
mark = 'fd_count_k' if any('fd_count_k' in r[2] for r in rows) else 'fdrand_k'
hits = [i for i, r in enumerate(rows) if mark in r[2]]
last, prev = hits[-1], hits[-2]
out = open('gpurun_out/step_trace.txt', 'w')
t0 = rows[prev][0]
pe = t0
for s, e, n in rows[prev:last]:
    out.write('%9.1f %8.1f %7.1f  %s\n' % ((s - t0) / 1e3, (e - s) / 1e3, (s - pe) / 1e3, n))
    pe = e
out.close()
P
rm -rf gpurun_out/trc
cat gpurun_out/step_trace.txt
