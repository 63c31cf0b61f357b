#!/usr/bin/env python3
"""Summarise tools/gpu_pmc_sb.sh: per-kernel PMC averages -> profiles/<tag>_pmc_sb_<workload>.json and
profiles/traffic.json["<workload>:sb"] (bench.py's roofline.traffic for sad_sb_kernel).

Units / gfx950 corrections (MI355X_MICROARCH.md "HBM"): FETCH_SIZE and WRITE_SIZE count KiB; FETCH_SIZE under-reports
wide reads, so it is calibrated inside the same run on sad_cand_kernel over the mv (0,0) list, which reads every
visible source and reference byte of the ring exactly once (known byte count); the factor is applied to
sad_sb_kernel.  WRITE_SIZE is taken at face value."""
import csv, glob, json, os, sys
from collections import defaultdict

sys.path.insert(0, os.getcwd())
from benchlib import common as bench_common, sad as bench  # WORKLOADS (sizes only; no GPU work at import)


def main(tag, wl):
    base = os.path.join("gpurun_out", tag, "pmcsb_" + wl)
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(base, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            short = "sad_sb" if ("sad_sb_kernel" in k or "sad_strip_kernel" in k) else "sad_x4d" if "sad_x4d_kernel" in k else "sad_cand" if "sad_cand_kernel" in k else None
            if short:
                acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
    cfg = bench.WORKLOADS[wl]
    W, H, es, F = cfg["width"], cfg["height"], 1 if cfg["bit_depth"] == 8 else 2, cfg["frames"]
    n = (W // 16) * (H // 16)
    known = F * (2 * W * H * es + 4 * n) + 8 * n
    out = {"workload": wl, "kernels": m, "notes": __doc__.split("\n\n")[1]}
    if "FETCH_SIZE" in m.get("sad_cand", {}) and "FETCH_SIZE" in m.get("sad_sb", {}):
        factor = known / (m["sad_cand"]["FETCH_SIZE"] * 1024.0)
        rd = m["sad_sb"]["FETCH_SIZE"] * 1024.0 * factor
        wr = m["sad_sb"].get("WRITE_SIZE", 0.0) * 1024.0
        out["fetch_calibration"] = {"known_bytes_sad_cand_launch": known, "raw_FETCH_SIZE_KiB": m["sad_cand"]["FETCH_SIZE"],
                                    "bytes_per_KiB_unit_factor": factor}
        out["sad_sb_hbm_bytes_per_launch"] = {"read": rd, "write": wr, "total": rd + wr}
        tj = os.path.join("profiles", "traffic.json")
        t = json.load(open(tj)) if os.path.exists(tj) else {}
        t[wl + ":sb"] = rd + wr
        t.setdefault("_measured_on", {})[wl + ":sb"] = bench_common.kernel_source_sha("sb")
        json.dump(t, open(tj, "w"), indent=1, sort_keys=True)
    json.dump(out, open(os.path.join("profiles", "%s_pmc_sb_%s.json" % (tag, wl)), "w"), indent=1, sort_keys=True)
    print(json.dumps({k: out[k] for k in out if k != "notes"}, sort_keys=True)[:1500])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
