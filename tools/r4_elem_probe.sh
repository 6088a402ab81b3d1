#!/bin/bash
# round 4: what each gather of the expansion kernel costs (probes 2-4 give wrong results on purpose), and the size of its memory-side requests
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export ESP_CFG4_2D=${ESP_CFG4_2D:-0} ESP_BENCH_NO_DIGEST=1 ESP_BENCH_SKIP_TRIPLETS=1
for v in ${PROBES:-0 1 2 3 4}; do
  echo "== probe $v"
  ESP_ELEM_PROBE=$v python3 tools/r4_extra.py 3 2>&1 | grep elements | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    for k, v in d.items():
        print(k, 'ms %.3f' % v['ms'], v['stage_ms'], 'digest', v['digest_ok'])"
done
rm -rf gpurun_out/pe_f
timeout 240 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d gpurun_out/pe_f -- python3 tools/r4_extra.py 1 > gpurun_out/pe_f.log 2>&1
python3 - <<'PY'
import collections, csv, glob, re
def short(n):
    n = n.replace("void ", ""); n = re.sub(r"\(.*\)$", "", n); return n[:90]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pe_f/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    if any(sum(x) / len(x) > 1e5 for x in v.values()):
        print("%-92s %s" % (k, "  ".join("%s avg %.5g (x%d)" % (c, sum(x) / len(x), len(x)) for c, x in sorted(v.items()))))
PY
tail -3 gpurun_out/pe_f.log | cut -c1-200
rm -rf gpurun_out/pe_f
