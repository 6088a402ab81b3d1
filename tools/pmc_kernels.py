#!/usr/bin/env python3
"""Per-KERNEL HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; collected separately: the TCC has 4
counter slots) of one command, e.g. bench.py with its extra.configs (BASELINE configs 3 and 4).

Every distinct kernel name (template arguments kept) gets: launches seen, average and total bytes fetched / written.
gfx950 corrections as in tools/pmc_traffic.py (/opt/skills/guides/MI355X_MICROARCH.md, section HBM): both counters are
in KiB; FETCH_SIZE reports half of the bytes of a wide coalesced streaming read and is doubled; WRITE_SIZE is exact.

usage: tools/pmc_kernels.py <fetch_run_dir> <write_run_dir> <out.json> [round tag] [min total MB to list]
"""
import collections
import csv
import glob
import json
import re
import sys


def short(name):
    name = name.replace("void ", "")
    name = re.sub(r"\(.*\)$", "", name)          # drop the argument list
    name = re.sub(r"\s+\[clone.*$", "", name)
    return name[:140]


def collect(run_dir, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(run_dir + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return acc


def main():
    fetch = collect(sys.argv[1], "FETCH_SIZE")
    write = collect(sys.argv[2], "WRITE_SIZE")
    floor_mb = float(sys.argv[5]) if len(sys.argv) > 5 else 50.0
    out = {}
    for k in sorted(set(fetch) | set(write)):
        f = [x * 1024.0 * 2.0 for x in fetch.get(k, [])]   # KiB -> B, x2 gfx950 correction
        w = [x * 1024.0 for x in write.get(k, [])]
        tot = sum(f) + sum(w)
        if tot < floor_mb * 1e6:
            continue
        n = max(len(f), len(w), 1)
        out[k] = {"launches_seen": n, "fetch_bytes_per_launch": sum(f) / n, "write_bytes_per_launch": sum(w) / n,
                  "hbm_bytes_per_launch": tot / n, "hbm_bytes_total": tot}
    out["_meta"] = {"round": sys.argv[4] if len(sys.argv) > 4 else "unnamed",
                    "note": "per kernel name; HBM bytes from two rocprofv3 --pmc passes (FETCH_SIZE x2 gfx950 correction, "
                            "WRITE_SIZE), KiB -> B; kernels below %.0f MB in total are left out" % floor_mb}
    json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1].get("hbm_bytes_total", 0) if kv[0] != "_meta" else 1):
        if k != "_meta":
            print("%-100s x%-4d fetch %8.1f MB  write %8.1f MB per launch" % (k[:100], v["launches_seen"],
                                                                            v["fetch_bytes_per_launch"] / 1e6, v["write_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main()
