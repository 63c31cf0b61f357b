"""The ONE JSON line of the driver contract (< 6 KB) and the full record beside it."""
import json
import os
import sys
import time

import numpy as np

from . import common
from .common import HBM_PEAK_GBS, ROOT, kernel_avg_ms, ramp
from .sad import WORKLOADS
from .variance import VAR_WORKLOADS


def _sig(x, n=5):
    """Floats to n significant digits, recursively (the printed line is read by a parser with a size limit; the side file keeps full precision)."""
    if isinstance(x, float):
        return float("%.*g" % (n, x))
    if isinstance(x, dict):
        return {k: _sig(v, n) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, n) for v in x]
    return x


ROOFLINE_KEYS = ("bound", "limited_by", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "compulsory_bytes_per_launch",
                 "algorithmic_bytes_per_launch", "ceiling_GBs", "frac_of_ceiling", "traffic_over_compulsory", "traffic_measured_on")


LINE_LIMIT = 6000  # bytes of the final stdout line; the driver's record keeps an 8 KB tail (round 4's 31 KB line was not parsed)


def _other_summary(o):
    """One or two scalars per informational workload for the printed line; the whole entry goes to the side file / stderr."""
    out = {}
    for k in ("value", "ms_per_frame", "ms_per_step", "frames_per_s", "roofline_frac"):   # (units: the full record)
        if isinstance(o.get(k), (int, float, str)):
            out[k] = o[k]
    if isinstance(o.get("yuv420"), dict):   # the inner loop's 4:2:0 leg
        out["yuv420_ms_per_frame"], out["yuv420_roofline_frac"] = o["yuv420"]["ms_per_frame"], o["yuv420"].get("roofline_frac")
    for k, v in o.items():  # nested legs that carry a per-frame time (TF q30 / q12, joint search branches, NSTEP / 8-tap ...)
        if isinstance(v, dict) and isinstance(v.get("ms_per_frame", v.get("ms_per_filtered_frame")), (int, float)):
            out[k + "_ms"] = v.get("ms_per_frame", v.get("ms_per_filtered_frame"))
        elif k.endswith("_ms_per_frame") and isinstance(v, (int, float)):
            out[k] = v
    for k in o:
        if k.startswith("parity") and o[k] is not None:
            out["parity"] = bool(out.get("parity", True)) and bool(o[k])
    return out


def build_lines(args, world, main_res, others, strong):
    """(full record, printed line).  The printed line carries the contract's keys, the roofline as flat scalars (the three north-star sizes
    side by side), the cpu baseline, the transform half of the metric and one summary scalar set per informational workload -- and stays
    under LINE_LIMIT bytes.  Everything else (per-size tables, stage timings, notes, cpu legs) is in the full record."""
    cfg = WORKLOADS[args.workload]
    sad_all = [main_res] + [o for o in others if str(o.get("workload", "")).startswith("sad16x16_modeA")]
    roof_full = dict(main_res["roofline"])
    roof_full["sizes"] = {r_["workload"]: dict(r_["roofline"], candidates_per_s=r_["value"]) for r_ in sad_all}
    roof = {k: main_res["roofline"][k] for k in ROOFLINE_KEYS if main_res["roofline"].get(k) is not None}
    roof.setdefault("traffic", None)
    for r_ in sad_all[1:]:
        tag = r_["workload"].replace("sad16x16_modeA_", "")   # 4k_8bit / 4k_10bit / *_range32
        for k_ in ("frac", "avg_launch_ms") if tag.endswith("range32") else ("frac", "avg_launch_ms", "frac_of_ceiling", "traffic_over_compulsory"):
            if r_["roofline"].get(k_) is not None:
                roof["%s_%s" % (k_, tag)] = r_["roofline"][k_]
        if not tag.endswith("range32"):   # (the +-32 contract's other figures: the full record)
            roof["candidates_per_s_%s" % tag] = r_["value"]
    txqs = [o for o in others if str(o.get("workload", "")).startswith("fwd_txfm2d+quantize_b")]
    vars_ = [o for o in others if o.get("workload") in VAR_WORKLOADS]
    filt = next((o for o in others if o.get("workload") == "filters_ring_4k_10bit"), None)
    rest = [o for o in others if o not in txqs and o not in sad_all and o not in vars_ and o is not filt]
    cpu = main_res.get("cpu_baseline")
    head = {
        "metric": "SAD-candidates/s", "value": main_res["value"], "unit": "candidates/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": main_res["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8" if cfg["bit_depth"] == 8 else "u16", "data": "synthetic",
        "config": {"workload": args.workload, "frame": "%dx%d" % (cfg["width"], cfg["height"]),
                   "bit_depth": cfg["bit_depth"], "block": "16x16",
                   "mode": "A: 1 sad16x16 @mv(0,0) + 1 sad16x16x4d (uniform in [-%d,%d]^2) per block" % (cfg.get("search_range", 64), cfg.get("search_range", 64)),
                   "ring_frame_pairs_per_gpu": common.FRAMES_OVERRIDE or cfg["frames"], "candidates_per_step": main_res["candidates_per_step"],
                   "partition": ("balanced tile columns (encoder.c:247-275)" if common.TILE_COLUMNS == "balanced" else "uniform tile columns (tile_common.c:76-97)") +
                                ", one per GPU; no data-path collective",
                   "clock_ramp_s": float(os.environ.get("AOMHIP_BENCH_RAMP_S", "0.25"))},
    }
    full = dict(head, roofline=roof_full, cpu_baseline=cpu,
                txq={t["workload"]: t for t in txqs} or None, strong_scaling_search=strong,
                parity_frame0_and_last_slot=main_res["parity_frame0"], kernels=main_res["kernels"], others=others)
    line = dict(head, roofline=roof,
                cpu_baseline=None if cpu is None else dict({k: cpu.get(k) for k in ("value", "unit", "cores", "kind", "cpu_model")},
                                                           sample=cpu.get("sample_short", "")),
                # the other half of BASELINE.json's metric: fwd_txfm+quant blocks/s at 1080p (8-bit) and 4K (10-bit)
                txq={t["workload"]: {"value": t["value"], "unit": "blocks/s",
                                     "roofline": {k: t["roofline"].get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms")},
                                     "cpu_baseline": {k: (t.get("cpu_baseline") or {}).get(k) for k in ("value", "cores", "kind")},
                                     "per_size_frac": {k: v["frac"] for k, v in t["per_size"].items()},
                                     "qindex_frac_16x16": t.get("qindex_sweep_16x16_frac"),
                                     "tx_type_frac_16x16": None if not t.get("tx_type_sweep_16x16_frac") else
                                     {k: t["tx_type_sweep_16x16_frac"][k] for k in ("min", "max")}} for t in txqs} or None,
                # SURVEY 8(d) rows A6-A8 / E / F on HBM-resident rings: frac = algorithmic bytes, c = compulsory bytes (every ring byte once),
                # t = counter traffic (null until measured on this kernel source), all / launch time / 8 TB/s
                variance={o["workload"].replace("variance16x16_modeA_", "var_").replace("sub_pixel_var_", "subpel_var_"):
                          dict({"frac": o["roofline"]["frac"], "c": o["roofline"]["frac_compulsory"], "t": o["roofline"]["frac_traffic"],
                                "ms": o["roofline"]["avg_launch_ms"], "parity": o["parity_sample_slot0_and_last"]},
                               # (full-pel lists through the strip walk, aomhip_variance_sb_batch: the same evaluations, bit-identical)
                               **({"sb_ms": o["strip_walk"]["avg_launch_ms"], "sb_c": o["strip_walk"]["frac_compulsory"], "sb_t": o["strip_walk"]["frac_traffic"],
                                   "sb_same": o["strip_walk"]["identical_to_direct_slot0_and_last"]} if o.get("strip_walk") else {})) for o in vars_} or None,
                filters_ring=None if filt is None else dict(
                    {k: {"us": filt[k]["ms_per_plane"] * 1e3, "frac": filt[k]["frac"], "c": filt[k]["frac_compulsory"], "t": filt[k]["frac_traffic"]}
                     for k in ("deblock_vert+horz", "cdef_luma")}, parity=filt["parity_slot0_and_last"], ring_GB=1.32,
                    lpf_trial_us=(filt.get("lpf_search_trial") or {}).get("us_per_trial")),
                strong_scaling_search=strong,
                parity_frame0_and_last_slot=main_res["parity_frame0"],
                parity_all=all(bool(v) for o in [main_res] + others for k, v in o.items() if k.startswith("parity") and v is not None),
                others={str(o.get("workload")): _other_summary(o) for o in rest} or None)
    il = next((o for o in rest if o.get("workload") == "encode_inner_loop_4k_10bit"), None)
    if il and il.get("valu_issue_rates"):
        vr = il["valu_issue_rates"]
        # the measured denominator of every valu_frac (aomhip_valu_issue_probe, this run) and the stage fractions re-based on it
        line["valu_issue"] = {"unit": "G wave-instr/s/SIMD", "fast": vr["fast"] / 1e9, "slow": vr["slow"] / 1e9, "trans": vr["trans"] / 1e9,
                              "clocks_per_wave_inst": vr["clocks_per_wave_inst"], "clock_GHz": vr["clock_hz_median"] / 1e9,
                              "stage_valu_frac": {k: v.get("valu_frac") for k, v in il["stages"].items() if v.get("valu_frac") is not None},
                              "stage_ms": {k: v["ms"] for k, v in il["stages"].items()}}
    if isinstance(line.get("strong_scaling_search"), dict) and isinstance(line["strong_scaling_search"].get("exchange"), dict):
        # per-rank lists -> the busiest rank's bytes (the lists stay in the full record)
        st = dict(line["strong_scaling_search"])
        ex = dict(st["exchange"])
        eb = ex.pop("expected_bytes_per_rank_per_frame", None)
        ex.pop("transport", None)
        if eb:
            ex["expected_bytes_per_frame_max_rank"] = {"%s_%s" % (m, d_): max(eb[m][d_]) for m in ("halo", "allgather") for d_ in ("send", "recv")}
        st["exchange"] = ex
        line["strong_scaling_search"] = st
    line = _sig(line)
    # never let the line outgrow the record that reads it: shed the least important keys first (they stay in the full record)
    for drop in ("valu_issue.stage_ms", "valu_issue.stage_valu_frac", "others.wiener_stats_luma_4k", "others.cdef_search_luma_4k_10bit", "others.mesh_search_4k_10bit",
                 "strong_scaling_search.tile_columns_px_balanced", "strong_scaling_search.tile_columns_px_uniform", "others", "txq"):
        if len(json.dumps(line, separators=(",", ":"))) <= LINE_LIMIT:
            break
        if "." in drop:
            a, b = drop.split(".")
            if isinstance(line.get(a), dict):
                line[a].pop(b, None)
        else:
            line[drop] = None
    return full, line


def emit_lines(full, line):
    """Full record -> bench_full.json (gpurun_out/ when it exists, else beside this script; AOMHIP_BENCH_FULL overrides) and, one JSON object
    per workload, to stderr; then the ONE stdout line."""
    path = os.environ.get("AOMHIP_BENCH_FULL")
    if not path:
        d = os.path.join(ROOT, "gpurun_out")
        path = os.path.join(d if os.path.isdir(d) else ROOT, "bench_full.json")
    try:
        with open(path, "w") as f:
            json.dump(full, f)
        line["full_record"] = os.path.relpath(path, ROOT)
    except OSError as e:
        print("bench.py: could not write %s (%s)" % (path, e), file=sys.stderr)
    for o in full.get("others") or []:
        print(json.dumps(_sig(o, 6)), file=sys.stderr)
    sys.stderr.flush()
    out = json.dumps(line, separators=(",", ":"))
    assert len(out) <= LINE_LIMIT + 200, "bench line grew to %d bytes" % len(out)
    print(out, flush=True)
