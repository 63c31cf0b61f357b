#!/usr/bin/env python3
"""Summarise the FETCH_SIZE / WRITE_SIZE passes of tools/gpu_pmc_traffic_all.sh for the round-6 workloads (variance rings, the filters' ring):
per key of profiles/traffic.json the HBM-side bytes per launch of the named kernels, with the sha of the kernel sources the figure was measured
on (benchlib reads the figure only while that sha is the tree's).

Units / gfx950 corrections (MI355X_MICROARCH.md "HBM"): FETCH_SIZE and WRITE_SIZE count KiB; on gfx950 FETCH_SIZE reports half of the bytes of
wide coalesced reads (128-byte requests tallied at 64) -- doubled here, as the guide prescribes; WRITE_SIZE is taken at face value (the fill
calibration of tools/pmc_traffic_txq.py measures 1.000).  The doubling is exact for 16-byte-per-lane reads (the variance kernels, the strip walk: the
in-run calibration of tools/pmc_traffic_sb.py gives 1.89 - 1.96) and an UPPER BOUND for the filters' 8-byte loads (2-byte-per-lane reads measure 1.6 in
tools/pmc_traffic_txq.py's run): their `read` is between 0.8 and 1.0 of the figure stored.  The raw counters travel with the figure
(profiles/<tag>_pmc_<workload>.json)."""
import csv, glob, json, os, sys
from collections import defaultdict

sys.path.insert(0, os.getcwd())
from benchlib import common

# workload -> [(traffic.json key, kernel-name substrings whose per-launch means are ADDED, source files of the sha)]
TABLE = {}
for wl in ("variance16x16_modeA_1080p_8bit", "variance16x16_modeA_4k_10bit"):
    TABLE[wl] = [(wl, ("variance_kernel",), ("variance.hip", "variance_device.h")), (wl + ":sb", ("sad_strip_kernel",), ("sad_sb.hip",))]
for wl in ("sub_pixel_variance16x16_modeA_1080p_8bit", "sub_pixel_variance16x16_modeA_4k_10bit"):
    TABLE[wl] = [(wl, ("variance_kernel",), ("variance.hip", "variance_device.h"))]
TABLE["filters_ring_4k_10bit"] = [("filters_ring_4k_10bit:deblock_vert", ("deblock_vert", "deblock_horz"), ("deblock.hip",)),
                                  ("filters_ring_4k_10bit:cdef_luma_kernel", ("cdef_luma_kernel",), ("cdef.hip",))]


def main(tag, wl):
    base = os.path.join("gpurun_out", tag, "pmc_" + wl)
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(base, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    tj = os.path.join("profiles", "traffic.json")
    t = json.load(open(tj)) if os.path.exists(tj) else {}
    out = {"workload": wl, "notes": __doc__.split("\n\n")[1], "keys": {}}
    for key, subs, sources in TABLE[wl]:
        rd = wr = 0.0
        raw = {}
        for sub in subs:
            ks = [k for k in acc if sub in k]
            if not ks:
                continue
            fetch = [v for k in ks for v in acc[k].get("FETCH_SIZE", [])]
            write = [v for k in ks for v in acc[k].get("WRITE_SIZE", [])]
            if not fetch or not write:
                continue
            raw[sub] = {"FETCH_SIZE_KiB_mean": sum(fetch) / len(fetch), "WRITE_SIZE_KiB_mean": sum(write) / len(write), "launches": len(fetch)}
            rd += 2.0 * 1024.0 * sum(fetch) / len(fetch)
            wr += 1024.0 * sum(write) / len(write)
        if len(raw) != len(subs):
            print("missing counters for", key, list(raw)); continue
        t[key] = rd + wr
        t.setdefault("_measured_on", {})[key] = common.source_sha(os.getcwd(), sources)
        out["keys"][key] = {"read": rd, "write": wr, "total": rd + wr, "raw": raw}
    json.dump(t, open(tj, "w"), indent=1, sort_keys=True)
    json.dump(out, open(os.path.join("profiles", "%s_pmc_%s.json" % (tag, wl)), "w"), indent=1, sort_keys=True)
    print(json.dumps(out["keys"], sort_keys=True)[:1200])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
