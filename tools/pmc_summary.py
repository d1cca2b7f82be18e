#!/usr/bin/env python3
"""Fold rocprofv3 PMC passes (each collected with --kernel-trace only, one small counter set per pass) into per-kernel,
per-launch averages for the kernels of this library.

  HBM traffic (as in round 1):
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d OUT/fetch -- python3 bench.py --steps 3 --warmup 1 ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d OUT/write -- python3 bench.py --steps 3 --warmup 1 ...
    python tools/pmc_summary.py traffic OUT/fetch OUT/write > profiles/rNN_hbm_traffic.json
  Units and correction as MI355X_MICROARCH.md prescribes: both counters are in KiB; on gfx950 FETCH_SIZE counts the wide
  coalesced reads at half their size, so it is doubled.

  Matrix-pipe / issue counters:
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES ... -d OUT/p1 -- python3 bench.py ...
    (further passes: SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY; SQ_INSTS_LDS SQ_INSTS_VALU ...)
    python tools/pmc_summary.py counters OUT/p1 OUT/p2 ... > profiles/rNN_mfma_busy.json
  Derived (MI355X_MICROARCH.md): SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe cycles summed over SIMDs; GRBM_GUI_ACTIVE is
  summed over the 8 XCDs, so kernel cycles = GRBM_GUI_ACTIVE / 8; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles.
      mfma_busy_pct_of_chip      = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 256 CUs x 4 SIMDs)
      mfma_busy_pct_of_used_cus  = the same over the CUs the grid can occupy (min(workgroups, 256))
Launches are grouped by kernel name and grid size."""
import csv, glob, json, os, re, sys
from collections import defaultdict


def short(name):
    return re.sub(r"\(.*", "", name.replace("void ", "").replace("uvad::(anonymous namespace)::", ""))


def fold(dirs):
    """{kernel key: {counter: (sum, launches)}}, plus workgroup counts."""
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    wgs = {}
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if "uvad::" not in row["Kernel_Name"]:
                    continue
                key = f"{short(row['Kernel_Name'])} grid={row['Grid_Size']}"
                a = acc[key][row["Counter_Name"]]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
                try:
                    wgs[key] = int(row["Grid_Size"]) // max(int(row["Workgroup_Size"]), 1)
                except (KeyError, ValueError):
                    pass
    return acc, wgs


def main():
    mode = sys.argv[1]
    if mode == "traffic":
        fa, _ = fold([sys.argv[2]])
        wa, _ = fold([sys.argv[3]])
        out = {"note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes), averaged per launch; FETCH_SIZE doubled per "
                       "MI355X_MICROARCH.md (gfx950 counts wide coalesced reads at 1/2); KiB -> bytes x 1024", "kernels": {}}
        for k in sorted(set(fa) | set(wa)):
            f, nf = fa.get(k, {}).get("FETCH_SIZE", (0.0, 0))
            w, nw = wa.get(k, {}).get("WRITE_SIZE", (0.0, 0))
            f, w = f / max(nf, 1), w / max(nw, 1)
            out["kernels"][k] = {"launches": max(nf, nw), "FETCH_SIZE_KB_raw": f, "fetch_MB_corrected_x2": 2 * f * 1024 / 1e6,
                                 "WRITE_SIZE_KB": w, "write_MB": w * 1024 / 1e6, "hbm_MB_per_launch": (2 * f + w) * 1024 / 1e6}
        print(json.dumps(out, indent=1))
        return
    acc, wgs = fold(sys.argv[2:])
    out = {"note": "rocprofv3 --kernel-trace --pmc <set> passes (one set per pass, no other tracing), averaged per launch; derived values as "
                   "in the header of tools/pmc_summary.py", "kernels": {}}
    for k in sorted(acc):
        c = {name: v[0] / max(v[1], 1) for name, v in acc[k].items()}
        row = {"launches": max(v[1] for v in acc[k].values()), "workgroups": wgs.get(k), "counters_per_launch": c}
        if "GRBM_GUI_ACTIVE" in c and c["GRBM_GUI_ACTIVE"] > 0:
            cyc = c["GRBM_GUI_ACTIVE"] / 8.0
            row["kernel_cycles"] = cyc
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
                cus = min(wgs.get(k) or 256, 256)
                row["mfma_busy_pct_of_chip"] = 100.0 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 256 * 4)
                row["mfma_busy_pct_of_used_cus"] = 100.0 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * cus * 4)
                row["cus_used"] = cus
        if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"] > 0:
            for nm in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS"):
                if nm in c:
                    row[nm.lower() + "_pct_of_wave_cycles"] = 100.0 * c[nm] / c["SQ_WAVE_CYCLES"]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_VALU_MFMA_COEXEC_CYCLES" in c and c["SQ_VALU_MFMA_BUSY_CYCLES"] > 0:
            row["mfma_coexec_pct_of_mfma_busy"] = 100.0 * c["SQ_VALU_MFMA_COEXEC_CYCLES"] / c["SQ_VALU_MFMA_BUSY_CYCLES"]
        if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE", 0) > 0:
            row["lds_bank_conflict_pct_of_lds_active"] = 100.0 * c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
        out["kernels"][k] = row
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
