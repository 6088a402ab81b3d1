#!/bin/bash
# ordered kernel + copy trace of ONE timed step of cfg_mt_sum (from a group3_items_k MULTI launch to the next)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/trc
ESP_EXTRA_ONLY=cfg_mt_sum rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/trc -- python3 tools/r4_extra.py 3 > gpurun_out/trc.log 2>&1
python3 - <<'P'
import csv, glob, collections
rows = []
for f in glob.glob('gpurun_out/trc/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('void ', '')[:60], r.get('Queue_Id', '')))
for f in glob.glob('gpurun_out/trc/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', ''), ''))
rows.sort()
hits = [i for i, r in enumerate(rows) if 'group3_items_k' in r[2]]
a, b = hits[2], hits[3]   # one timed step: everything between two folds launches
seg = rows[a + 1:b + 1]
t0 = seg[0][0]
print('step: %d operations in %.1f us' % (len(seg), (seg[-1][1] - t0) / 1e3))
cnt = collections.Counter(r[2].split('(')[0][:44] for r in seg)
for k, v in cnt.most_common(30):
    print('%5d  %s' % (v, k))
# busy time (union of intervals) of the fills: up to the first pair_flags_k
end_fill = next((r[0] for r in seg if 'pair_flags_k' in r[2]), seg[-1][1])
iv = sorted((r[0], r[1]) for r in seg if r[0] < end_fill)
busy, cur_s, cur_e = 0, None, None
for s, e in iv:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
if cur_e is not None: busy += cur_e - cur_s
print('fills: wall %.1f us, device busy (union) %.1f us, sum of durations %.1f us' % ((end_fill - t0) / 1e3, busy / 1e3, sum(e - s for s, e in iv) / 1e3))
P
rm -rf gpurun_out/trc
