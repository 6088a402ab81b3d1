#!/bin/bash
# usage: tools/r6_mt_hiptrace.sh  -- HIP API statistics of cfg_mt_sum (20 steps) in the first GPU process of a box: which runtime calls stall a fill
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ht
ESP_EXTRA_ONLY=cfg_mt_sum rocprofv3 --hip-trace --stats --output-format csv -d gpurun_out/ht -- python3 tools/r4_extra.py 20 > gpurun_out/ht.log 2>&1
f=$(find gpurun_out/ht -name "*hip_api_stats.csv" | head -1)
head -25 "$f" | cut -c1-160
python3 - <<'P'
import csv, glob
rows = []
for f in glob.glob('gpurun_out/ht/**/*hip_api_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['End_Timestamp']) - int(r['Start_Timestamp']), r['Function'], int(r['Start_Timestamp']), r.get('Thread_Id')))
rows.sort(reverse=True)
t0 = min(r[2] for r in rows)
for d, fn, st, tid in rows[:12]:
    print('%9.2f ms  %-40s at %8.1f ms  thread %s' % (d / 1e6, fn, (st - t0) / 1e6, tid))
# ... and behind the first rounds (allocations, first launches): what still stalls
first = min(r[2] for r in rows if r[1] == 'hipMalloc')
late = [r for r in rows if r[2] > first + 150e6 and r[1] not in ('hipStreamSynchronize', 'hipEventSynchronize')]
print('-- later than 150 ms behind the first hipMalloc:')
for d, fn, st, tid in late[:30]:
    print('%9.2f ms  %-40s at %8.1f ms  thread %s' % (d / 1e6, fn, (st - t0) / 1e6, tid))
P
rm -rf gpurun_out/ht
