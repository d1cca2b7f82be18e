#!/usr/bin/env python3
"""Fold two rocprofv3 PMC passes (one with --pmc FETCH_SIZE, one with --pmc WRITE_SIZE, each with --kernel-trace only)
into HBM bytes per launch for the kernels of this library.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d OUT/fetch -- python3 bench.py --steps 3 --warmup 1 ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d OUT/write -- python3 bench.py --steps 3 --warmup 1 ...
    python tools/pmc_summary.py OUT/fetch OUT/write > profiles/rNN_hbm_traffic.json

Units and correction as MI355X_MICROARCH.md prescribes: both counters are in KiB; on gfx950 FETCH_SIZE counts the wide
coalesced reads at half their size, so it is doubled.  Launches are grouped by kernel name and grid size."""
import csv, glob, json, os, re, sys
from collections import defaultdict


def fold(d, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter or "uvad::" not in row["Kernel_Name"]:
                continue
            name = re.sub(r"\(.*", "", row["Kernel_Name"].replace("void ", "").replace("uvad::(anonymous namespace)::", ""))
            key = f"{name} grid={row['Grid_Size']}"
            acc[key][0] += float(row["Counter_Value"])
            acc[key][1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}


fetch, write = fold(sys.argv[1], "FETCH_SIZE"), fold(sys.argv[2], "WRITE_SIZE")
out = {"note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes), averaged per launch; FETCH_SIZE doubled per "
               "MI355X_MICROARCH.md (gfx950 counts wide coalesced reads at 1/2); KiB -> bytes x 1024", "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    f, nf = fetch.get(k, (0.0, 0))
    w, nw = write.get(k, (0.0, 0))
    out["kernels"][k] = {"launches": max(nf, nw), "FETCH_SIZE_KB_raw": f, "fetch_MB_corrected_x2": 2 * f * 1024 / 1e6,
                         "WRITE_SIZE_KB": w, "write_MB": w * 1024 / 1e6, "hbm_MB_per_launch": (2 * f + w) * 1024 / 1e6}
print(json.dumps(out, indent=1))
