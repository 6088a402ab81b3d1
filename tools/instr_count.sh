#!/bin/bash
# dynamic instruction counts per wave of the pipeline's kernels (rocprofv3 PMC; run through gpurun)
# usage: instr_count.sh [--steps=1]   (any second bench.py flag instead of --no-extra: with the extra configs' kernels)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ic
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES --kernel-trace --output-format csv -d gpurun_out/ic -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline ${1:---no-extra} > gpurun_out/ic.log 2>&1
python3 - <<'PY'
import collections, csv, glob
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/ic/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].replace("void ", "").split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    if not any(x in k for x in ("local_k", "wave_k", "group3_k", "group3_items_k", "elem_", "run_hist_k", "run_scatter_k", "fdrand_k", "fdrand_part_k", "fd_count_k", "fem_", "merge_k", "tile_hist_k", "scatter_k")):
        continue
    m = {c: sum(x) / len(x) for c, x in v.items()}
    w = m["SQ_WAVES"]
    tot = sum(m[c] for c in m if c.startswith("SQ_INSTS"))
    print("%-60s waves %8d per wave: valu %6.0f salu %6.0f lds %5.0f smem %4.0f vmem %4.0f total %6.0f cycles/4 %7.0f" % (
        k[:60], w, m["SQ_INSTS_VALU"] / w, m["SQ_INSTS_SALU"] / w, m["SQ_INSTS_LDS"] / w, m["SQ_INSTS_SMEM"] / w,
        (m["SQ_INSTS_VMEM_RD"] + m["SQ_INSTS_VMEM_WR"]) / w, tot / w, m["SQ_WAVE_CYCLES"] / w))
PY
rm -rf gpurun_out/ic
