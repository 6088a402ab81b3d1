#!/bin/bash
# usage: tools/r6_trace_head.sh  -- ordered kernel + copy trace of one warm step of the HEADLINE (bench.py's timed loop)
# (name, start offset us, duration us, gap to the previous end us) -> gpurun_out/r6_trace_head.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/trc
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/trc -- python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extra > gpurun_out/trc.log 2>&1
python3 - <<'P'
import csv, glob
rows = []
for f in glob.glob('gpurun_out/trc/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('void ', '')[:110]))
for f in glob.glob('gpurun_out/trc/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', '')))
rows.sort()
hits = [i for i, r in enumerate(rows) if 'fdrand_part_k' in r[2]]
a, b = hits[-4], hits[-3]
out = open('gpurun_out/r6_trace_head.txt', 'w')
t0 = rows[a][0]
pe = t0
for s, e, n in rows[a:b]:
    out.write('%9.1f %8.1f %7.1f  %s\n' % ((s - t0) / 1e3, (e - s) / 1e3, (s - pe) / 1e3, n))
    pe = max(pe, e)
out.close()
P
rm -rf gpurun_out/trc
cat gpurun_out/r6_trace_head.txt
