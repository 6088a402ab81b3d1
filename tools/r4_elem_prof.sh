#!/bin/bash
# round 4: kernel durations + memory-side read requests of the element-level append at config 4's 3-D size (run through gpurun)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pe gpurun_out/pe_f gpurun_out/pe_r
export ESP_CFG4_2D=${ESP_CFG4_2D:-0} ESP_BENCH_NO_DIGEST=1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pe -- python3 tools/r4_extra.py 2 > gpurun_out/pe.log 2>&1
find gpurun_out/pe -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r4_elem_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pe_f -- python3 tools/r4_extra.py 1 > gpurun_out/pe_f.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d gpurun_out/pe_r -- python3 tools/r4_extra.py 1 > gpurun_out/pe_r.log 2>&1
python3 - <<'PY'
import collections, csv, glob, re
def short(n):
    n = n.replace("void ", ""); n = re.sub(r"\(.*\)$", "", n); return n[:90]
for d in ("gpurun_out/pe_f", "gpurun_out/pe_r"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        s = "  ".join("%s avg %.4g (x%d)" % (c, sum(x) / len(x), len(x)) for c, x in sorted(v.items()))
        if any(sum(x) / len(x) > 1e5 for x in v.values()):
            print("%-92s %s" % (k, s))
PY
rm -rf gpurun_out/pe gpurun_out/pe_f gpurun_out/pe_r
head -30 gpurun_out/r4_elem_kernel_stats.csv | cut -c1-220
tail -3 gpurun_out/pe_f.log | cut -c1-300
