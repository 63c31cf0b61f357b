"""Per-kernel instruction counts of the inner loop from the --pmc passes of tools/gpu_pmc_stages.sh -> gpurun_out/gpu_pmc_stages/summary.json
(copied to profiles/r04_inner_loop_pmc.json).  Counters are normalised per wavefront (counter / SQ_WAVES of the same pass and kernel) and scaled
by the wavefronts the launch really has (grid / 64): the ratio does not depend on how many shader engines a counter is sampled on."""
import csv, glob, json, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
waves = {}
for f in glob.glob(out + "/g*/*counter_collection.csv"):
    rows = list(csv.DictReader(open(f)))
    by = collections.defaultdict(dict)
    for r in rows:
        key = (r["Dispatch_Id"], r["Kernel_Name"])
        by[key][r["Counter_Name"]] = float(r["Counter_Value"])
        by[key]["_waves"] = int(r["Grid_Size"]) // 64
    for (_, k), c in by.items():
        w = c.get("SQ_WAVES", 0)
        if w <= 0: continue
        for n, v in c.items():
            if n not in ("SQ_WAVES", "_waves"):
                acc[k][n].append(v / w)
        waves[k] = c["_waves"]
res = {}
for k, c in acc.items():
    short = k.replace("void ", "").replace("aomhip::", "").replace("(anonymous namespace)::", "").split("(")[0]
    if not any(s in short for s in ("fullpel_diamond", "subpel_bilinear", "inter_pred", "xform_quant", "inv_txfm", "deblock", "cdef_luma", "subtract",
                                    "full_pixel_search", "fp_row", "tf_apply", "sad_strip", "mesh", "encode_inter_block", "compound", "obmc", "refining", "warp_error", "int_pro", "vbp_")):
        continue
    e = {"wavefronts_per_launch": waves[k]}
    for n, v in c.items():
        e[n + "_per_wavefront"] = sum(v) / len(v)
    res[short] = e
json.dump(res, open(out + "/summary.json", "w"), indent=1, sort_keys=True)
for k, e in sorted(res.items()):
    print(k[:60], {a: round(b, 1) for a, b in e.items()})
