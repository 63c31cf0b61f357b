#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (tools/gpu_pmc.sh) into per-kernel averages and write
profiles/<tag>_pmc_<workload>.json + profiles/traffic.json (bench.py's roofline.traffic).

Unit / gfx950 corrections (MI355X_MICROARCH.md "HBM"): FETCH_SIZE and WRITE_SIZE are in KiB;
FETCH_SIZE under-reports wide reads, so it is CALIBRATED on this kernel family's own access
pattern: sad_cand_kernel over the mv (0,0) list reads every visible src and ref byte of the
ring exactly once (known byte count), which fixes bytes-per-FETCH_SIZE-unit; the same factor
is applied to sad_x4d_kernel.  WRITE_SIZE is taken at face value (KiB)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def per_kernel(path):
    acc = defaultdict(lambda: defaultdict(list))
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"]
        short = "sad_x4d" if "sad_x4d_kernel" in k else "sad_cand" if "sad_cand_kernel" in k else \
            k.split("(")[0].split("::")[-1]
        acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


def main(tag, workload, known_cand_bytes):
    base = os.path.join("gpurun_out", tag, "pmc_" + workload)
    merged = defaultdict(dict)
    for f in glob.glob(os.path.join(base, "*", "*counter_collection.csv")):
        for k, d in per_kernel(f).items():
            merged[k].update(d)
    out = {"workload": workload, "kernels": merged, "notes": __doc__.split("\n\n")[1]}
    cand, x4d = merged.get("sad_cand", {}), merged.get("sad_x4d", {})
    if "FETCH_SIZE" in cand and "FETCH_SIZE" in x4d:
        factor = known_cand_bytes / (cand["FETCH_SIZE"] * 1024.0)
        out["fetch_calibration"] = {"known_bytes_sad_cand_launch": known_cand_bytes,
                                    "raw_FETCH_SIZE_KiB": cand["FETCH_SIZE"], "bytes_per_KiB_unit_factor": factor}
        rd = x4d["FETCH_SIZE"] * 1024.0 * factor
        wr = x4d.get("WRITE_SIZE", 0.0) * 1024.0
        out["sad_x4d_hbm_bytes_per_launch"] = {"read": rd, "write": wr, "total": rd + wr}
        tj = os.path.join("profiles", "traffic.json")
        t = json.load(open(tj)) if os.path.exists(tj) else {}
        t[workload] = rd + wr
        json.dump(t, open(tj, "w"), indent=1, sort_keys=True)
    os.makedirs("profiles", exist_ok=True)
    json.dump(out, open(os.path.join("profiles", "%s_pmc_%s.json" % (tag, workload)), "w"), indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], float(sys.argv[3]))
