#!/usr/bin/env python3
"""Summarise tools/gpu_pmc_txq.sh -> profiles/<tag>_pmc_txq.json and profiles/traffic.json["txq_<N>x<N>"].

FETCH_SIZE / WRITE_SIZE count KiB at the L2 <-> fabric interface (MI355X_MICROARCH.md, HBM section).  Corrections, as that
section prescribes: FETCH_SIZE reports exactly 1/2 of the bytes of wide coalesced reads on gfx950 (128-byte requests tallied at
64 B) -> x2 for the transform kernels, whose loads are whole residual rows; cross-checked in the same run against the kernel's own
known read volume (every int16 of the 32-plane residual ring exactly once: `read_over_known` must be ~1).  WRITE_SIZE is
"uncalibrated" in the guide, so it is calibrated in the SAME process on the runtime's 256 MiB fill kernel (known writes); the
factor comes out at 1.000.  (plane_sse_kernel, 2 bytes per lane, is also in the run: its narrow reads are tallied differently --
factor 1.6 -- which is why the read factor is not taken from it.)  Per-launch averages over the launches of each transform size."""
import csv, glob, json, os, re, sys
from collections import defaultdict

sys.path.insert(0, os.getcwd())
from benchlib import common as bench  # kernel_source_sha (no GPU work at import)


def main(tag, workload="txq_1080p_8bit"):
    prefix = "txq_" if workload == "txq_1080p_8bit" else workload + "_"     # profiles/traffic.json keys, as bench.py's load_traffic() reads them
    base = os.path.join("gpurun_out", tag, "pmctxq" if workload == "txq_1080p_8bit" else "pmc" + workload)
    acc = defaultdict(lambda: defaultdict(list))
    names = set()
    for f in glob.glob(os.path.join(base, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            names.add(k[:90])
            if "plane_sse_kernel" in k:
                key = "calib_read"
            elif "fillBuffer" in k or "FillBuffer" in k or "fill_buffer" in k.lower():
                key = "calib_write"
            elif "xform_quant" in k:
                m = re.search(r"xform_quant\w*<\s*(\d+)\w*,\s*(\d+)", k)
                key = prefix + "%sx%s" % (m.group(1), m.group(2)) if m else prefix + "other"
            else:
                continue
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {"notes": __doc__.split("\n\n")[1], "kernel_names_seen": sorted(names)[:40]}
    mean = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
    mx = {k: {c: max(v) for c, v in d.items()} for k, d in acc.items()}
    out["raw_mean_KiB"] = mean
    rd_known, wr_known = 2 * 3840 * 2160 * 2, 256 << 20
    narrow = rd_known / (mean["calib_read"]["FETCH_SIZE"] * 1024.0) if "calib_read" in mean and mean["calib_read"].get("FETCH_SIZE") else None
    fr = 2.0
    # the fill may be split into several kernels: calibrate on the SUM of their writes per fill (3 fills in the run)
    fw = None
    if "calib_write" in acc and acc["calib_write"].get("WRITE_SIZE"):
        fw = 3.0 * wr_known / (sum(acc["calib_write"]["WRITE_SIZE"]) * 1024.0)
    # bench.py TxqGrid: every sample of the residual ring read once per launch (32 planes of 1920x1088 / 12 planes of 3840x2176 int16)
    residual_ring = 32 * 1920 * 1088 * 2 if workload == "txq_1080p_8bit" else 12 * 3840 * 2176 * 2
    out["calibration"] = {"read_factor": fr, "write_factor": fw, "write_known_bytes": wr_known, "narrow_read_factor_plane_sse": narrow,
                          "residual_ring_bytes": residual_ring}
    tj = os.path.join("profiles", "traffic.json")
    t = json.load(open(tj)) if os.path.exists(tj) else {}
    per = {}
    for k in sorted(mean):
        if not k.startswith(prefix) or k.startswith("calib") or fr is None or fw is None:
            continue
        rd = mean[k].get("FETCH_SIZE", 0.0) * 1024.0 * fr
        wr = mean[k].get("WRITE_SIZE", 0.0) * 1024.0 * fw
        per[k] = {"read": rd, "write": wr, "total": rd + wr, "read_over_known": rd / residual_ring}
        t[k] = rd + wr
        t.setdefault("_measured_on", {})[k] = bench.kernel_source_sha("txq")
    out["hbm_bytes_per_launch"] = per
    json.dump(t, open(tj, "w"), indent=1, sort_keys=True)
    json.dump(out, open(os.path.join("profiles", "%s_pmc_%s.json" % (tag, "txq" if workload == "txq_1080p_8bit" else workload)), "w"), indent=1, sort_keys=True)
    print(json.dumps({k: out[k] for k in out if k not in ("notes",)}, sort_keys=True)[:3000])


if __name__ == "__main__":
    main(sys.argv[1], *(sys.argv[2:3]))
