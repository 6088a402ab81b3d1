#!/bin/bash
# ordered kernel / copy trace of the LAST config-4 assembly (start us, duration us, gap to the previous op us, name): where the small
# launches and host round trips of the item partition sit.  ESP_FEM_DIM=3 for the 3-D mesh.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/trcf
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/trcf -- python3 tools/fem_step.py > gpurun_out/trcf.log 2>&1
python3 - <<'P'
import csv, glob
rows = []
for f in glob.glob('gpurun_out/trcf/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('void ', '')[:64]))
for f in glob.glob('gpurun_out/trcf/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', '')))
rows.sort()
hits = [i for i, r in enumerate(rows) if 'fem_items_k' in r[2] or 'fem_count_k' in r[2]]
# the last assembly starts at the last fem_count_k (its PART launch leaves at once on a shuffled stream) or fem_items_k
starts = [i for i, r in enumerate(rows) if 'fem_items_k' in r[2]] or hits
first = starts[-1]
out = open('gpurun_out/fem_trace.txt', 'w')
t0 = rows[first][0]
pe = t0
gap = 0.0
busy = 0.0
for s, e, n in rows[first:]:
    out.write('%9.1f %8.1f %7.1f  %s\n' % ((s - t0) / 1e3, (e - s) / 1e3, (s - pe) / 1e3, n))
    gap += max(0, s - pe) / 1e3
    busy += (e - s) / 1e3
    pe = max(pe, e)
out.write('launches %d  busy %.1f us  gaps %.1f us\n' % (len(rows) - first, busy, gap))
out.close()
P
rm -rf gpurun_out/trcf
cat gpurun_out/fem_trace.txt | cut -c1-110
