#!/bin/bash
# usage: tools/r6_instr.sh <extra config list>  -- dynamic instruction counts per wave of the kernels of the given extra configs
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ic
ESP_EXTRA_ONLY=${1:-cfg3} rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES --kernel-trace --output-format csv -d gpurun_out/ic -- python3 tools/r4_extra.py 1 > gpurun_out/ic.log 2>&1
python3 - <<'PY'
import collections, csv, glob
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/ic/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].replace("void ", "").split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    m = {c: sum(x) / len(x) for c, x in v.items()}
    w = m.get("SQ_WAVES", 0)
    if w < 1000:
        continue
    tot = sum(m[c] for c in m if c.startswith("SQ_INSTS"))
    print("%-72s n %3d waves %8d per wave: valu %6.0f salu %6.0f lds %5.0f smem %4.0f vmem %4.0f total %6.0f cycles/4 %7.0f" % (
        k[:72], len(v["SQ_WAVES"]), w, m["SQ_INSTS_VALU"] / w, m["SQ_INSTS_SALU"] / w, m["SQ_INSTS_LDS"] / w, m["SQ_INSTS_SMEM"] / w,
        (m["SQ_INSTS_VMEM_RD"] + m["SQ_INSTS_VMEM_WR"]) / w, tot / w, m["SQ_WAVE_CYCLES"] / w))
PY
rm -rf gpurun_out/ic
