"""Issue rate of every VALU opcode class of aomhip_valu_issue_probe at 1 / 2 / 4 / 8 wavefronts per SIMD on this box ->
gpurun_out/<tag>/valu_issue.json + a markdown table on stdout (profiles/r05_valu_issue.md)."""
import json, os, sys
sys.path.insert(0, os.getcwd())
import aom_av1_psy_amd as pkg

tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
ctx = pkg.capi.Context(0)
names = pkg.capi.valu_issue_probe_names()
rows = []
for op, nm in enumerate(names):
    row = {"op": nm}
    for w in (1, 2, 4, 8):
        r = ctx.valu_issue_probe(op, w, 400)            # calibration launch ...
        per_wave_rate = r["wave_insts_per_s_per_simd"] / w
        iters = max(200, int(6e-3 * per_wave_rate / 128))  # ... then ~6 ms
        r = ctx.valu_issue_probe(op, w, iters)
        row["w%d" % w] = r
    rows.append(row)
    print("| `%s` | %s | %.3f | %.2f | %.2f GHz | %.1f ms |" % (
        nm, " / ".join("%.3f" % (row["w%d" % w]["wave_insts_per_s_per_simd"] / 1e9) for w in (1, 2, 4, 8)),
        row["w8"]["wave_insts_per_s_per_simd"] / 1e9, row["w4"]["memtime_ticks_per_wave_inst"], row["w8"]["memtime_hz"] / 1e9, row["w8"]["launch_ms"]), flush=True)
os.makedirs("gpurun_out/" + tag, exist_ok=True)
json.dump(rows, open("gpurun_out/%s/valu_issue.json" % tag, "w"), indent=1)
ctx.close()
