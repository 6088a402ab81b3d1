#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counters: tools/sq_counters.py <run_dir> [<run_dir> ...]"""
import collections
import csv
import glob
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("void ", "").split("(")[0]
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc, key=lambda k: -sum(acc[k].get("SQ_BUSY_CYCLES", acc[k][next(iter(acc[k]))]))):
    v = acc[k]
    n = len(next(iter(v.values())))
    print(k[:70], "launches", n, {c: "%.4g" % (sum(x) / len(x)) for c, x in sorted(v.items())})
