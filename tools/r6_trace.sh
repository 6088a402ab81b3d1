#!/bin/bash
# usage: tools/r6_trace.sh <extra config name> [marker kernel]   -- ordered kernel + copy trace of the LAST step of one extra config
# (name, start offset us, duration us, gap to the previous end us) -> gpurun_out/r6_trace_<config>.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
CFG=${1:-cfg3}; MARK=${2:-fdrand_part_k}
rm -rf gpurun_out/trc
ESP_EXTRA_ONLY=$CFG rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/trc -- python3 tools/r4_extra.py 2 > gpurun_out/trc.log 2>&1
MARK=$MARK CFG=$CFG python3 - <<'P'
import csv, glob, os
rows = []
for f in glob.glob('gpurun_out/trc/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('void ', '')[:110]))
for f in glob.glob('gpurun_out/trc/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', '')))
rows.sort()
mark = os.environ['MARK']
hits = [i for i, r in enumerate(rows) if mark in r[2]]
# the last but one occurrence of the marker .. the last one (= one whole step, the stage-event step excluded: take the one before)
a, b = hits[-3], hits[-2]
out = open('gpurun_out/r6_trace_%s.txt' % os.environ['CFG'], 'w')
t0 = rows[a][0]
pe = t0
for s, e, n in rows[a:b]:
    out.write('%9.1f %8.1f %7.1f  %s\n' % ((s - t0) / 1e3, (e - s) / 1e3, (s - pe) / 1e3, n))
    pe = max(pe, e)
out.close()
P
rm -rf gpurun_out/trc
cat gpurun_out/r6_trace_$CFG.txt
