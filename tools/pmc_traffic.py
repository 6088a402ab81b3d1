#!/usr/bin/env python3
"""Turns two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; collected separately, TCC has 4 slots)
into profiles/pmc_traffic.json: HBM bytes per launch for the pipeline's kernels.

gfx950 corrections (/opt/skills/guides/MI355X_MICROARCH.md, section HBM): FETCH_SIZE and WRITE_SIZE
are in KiB; FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read and is
doubled; WRITE_SIZE is exact for wide streaming stores.

usage: tools/pmc_traffic.py <fetch_run_dir> <write_run_dir> <out.json> [round tag]
"""
import collections
import csv
import glob
import json
import sys

# kernel (qualified name prefix) -> stage of bench.py's "roofline.kernel"; the small launches that sort
# the run list with the 8-bit pass kernels are kept apart from the big ones
STAGE_OF = {"espgen::fdrand_part_k": "append", "espgen::fd_count_k": "hist", "espgen::fdrand_k": "append_plain",
            "espgen::pack_k": "append_pack", "espgen::fem_k": "append_fem", "espgen::fem_part_k": "append_fem_part",
            "espgen::fem_count_k": "hist_fem",
            "esprun::run_hist_k": "hist", "esprun::run_scatter_k": "scatter",
            "espradix::tile_hist_k": "hist_pass8", "espradix::scatter_k": "scatter_pass8",
            "esplocal::local_k": "local", "espfold::fold_k": "fold", "espmerge::merge_k": "merge"}


def collect(run_dir, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(run_dir + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].replace("void ", "")
            for k, st in STAGE_OF.items():
                if name.startswith(k):
                    acc[st].append(float(r["Counter_Value"]))
    return acc


def main():
    fetch = collect(sys.argv[1], "FETCH_SIZE")
    write = collect(sys.argv[2], "WRITE_SIZE")
    out = {}
    for st in sorted(set(fetch) | set(write)):
        f = sum(fetch[st]) / max(len(fetch[st]), 1) * 1024.0 * 2.0   # KiB -> B, x2 gfx950 correction
        w = sum(write[st]) / max(len(write[st]), 1) * 1024.0
        out[st] = {"fetch_bytes_per_launch": f, "write_bytes_per_launch": w, "hbm_bytes_per_launch": f + w,
                   "launches_seen": max(len(fetch[st]), len(write[st]))}
    out["_meta"] = {"round": sys.argv[4] if len(sys.argv) > 4 else "unnamed",
                    "note": "HBM bytes per launch from two rocprofv3 --pmc passes (FETCH_SIZE x2 gfx950 correction, WRITE_SIZE), KiB -> B"}
    json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
