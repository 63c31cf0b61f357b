#!/usr/bin/env python3
"""Golden vectors for the SELECTION GLUE of the inter leg of TPL's mode_estimation (av1/encoder/tpl_model.c), obtained by interpreting the
reference's own statements (build container only; tests/golden/ref_c_eval.py):

  prune    the `if (cpi->sf.tpl_sf.prune_starting_mv)` block (:706-731) -- qsort(center_mvs, .., compare_sad) (:308-315), the cut to
           4 - prune_starting_mv candidates and the 20 % rule -- run on the typedef, the comparator and the block's statements as they are written,
           with the candidates' SADs as inputs (the SAD loop :709-716 is the pinned aom_sadWxH).  qsort itself is libc: it is modelled as a STABLE
           sort calling the reference's comparator (glibc's qsort is a merge sort), which is what decides the order of candidates with equal SADs.
  best_of  the loop over the remaining candidates (:733-743): `thissme < bestsme` with bestsme = UINT32_MAX, best_rfidx_mv = { 0 }; motion_estimation's
           return value and MV per candidate are inputs (the function is pinned as a whole, ref_eval_composites.npz / test_oracle_me.py).
  best_ref the per-reference tail of the loop (:755-765): pred_error = AOMMAX(1, inter_cost), `inter_cost < best_inter_cost`; tpl_get_satd_cost's
           value per reference is an input.

int_mv is a union of an int and an MV; the interpreter has no unions: `.as_int` copies / compares of whole MVs are rewritten to the MV member (the same
rewrite as tests/golden/gen_ref_eval_joint.py).  Output: tests/golden/ref_eval_tpl.npz."""
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_c_eval as R  # noqa: E402
from gen_ref_eval_golden import evaluator, save, REF  # noqa: E402


def main():
    ev = evaluator(["av1/common/mv.h"])
    text = open(REF + "av1/encoder/tpl_model.c").read()
    tdef = re.search(r"typedef struct \{\n  int_mv mv;\n  int sad;\n\} center_mv_t;\n", text).group(0)
    cmp_ = re.search(r"static int compare_sad\(.*?\n}\n", text, re.S).group(0)
    prune = text[text.index("    // Prune starting mvs\n"):text.index("    for (idx = 0; idx < refmv_count; ++idx) {\n      int_mv this_mv;")]
    prune, n = re.subn(r"      // Get each center mv's sad\.\n      for \(idx = 0; idx < refmv_count; \+\+idx\) \{.*?\n      \}\n", "", prune, flags=re.S)
    assert n == 1
    best = text[text.index("    for (idx = 0; idx < refmv_count; ++idx) {\n      int_mv this_mv;"):text.index("    tpl_stats->mv[rf_idx].as_int = best_rfidx_mv.as_int;")]
    best, n = re.subn(r"motion_estimation\(cpi, x, src_mb_buffer, ref_mb,\s*src_stride, ref_stride, bsize,\s*center_mvs\[idx\]\.mv\.as_mv, &this_mv\)",
                      "tpl_hook_me(idx, &this_mv)", best)
    assert n == 1
    tail = text[text.index("    inter_cost =\n        tpl_get_satd_cost("):text.index("  if (best_rf_idx != -1 && best_inter_cost < best_intra_cost) {")]
    tail, n = re.subn(r"tpl_get_satd_cost\(bd_info, src_diff, bw, src_mb_buffer, src_stride,\s*predictor, bw, coeff, bw, bh, tx_size\)", "g_cost[rf_idx]", tail)
    assert n == 1 and tail.rstrip().endswith("}")          # (the closing brace of the rf_idx loop travels with the slice)
    adapt = lambda t: re.sub(r"\.as_int\b", ".as_mv", t)
    src = "typedef struct { MV as_mv; } int_mv;\n" + tdef + cmp_ + """
/* libc's qsort for this element type: a stable sort through the reference's comparator */
static void qsort(center_mv_t *base, int n, int size, int (*cmp)(const void *, const void *)) {
  for (int a = 1; a < n; ++a) {
    center_mv_t t;
    t = base[a];
    int j = a - 1;
    while (j >= 0 && cmp(&base[j], &t) > 0) { base[j + 1] = base[j]; --j; }
    base[j + 1] = t;
  }
}
typedef struct { int prune_starting_mv; } TPL_SF_view;
typedef struct { TPL_SF_view tpl_sf; } SF_view;
typedef struct { SF_view sf; } CPI_view;
int tpl_prune_slice(CPI_view *cpi, center_mv_t *center_mvs, int refmv_count) {
  int idx;
""" + adapt(prune.replace("sizeof(center_mvs[0])", "8")) + """
  return refmv_count;
}
unsigned int g_me_err[4];
int g_me_row[4], g_me_col[4];
static unsigned int tpl_hook_me(int idx, int_mv *this_mv) { this_mv->as_mv.row = g_me_row[idx]; this_mv->as_mv.col = g_me_col[idx]; return g_me_err[idx]; }
void tpl_best_of_slice(int refmv_count, int_mv *out) {
  int idx;
  int_mv best_rfidx_mv = { 0 };
  uint32_t bestsme = UINT32_MAX;
""" + adapt(best) + """
  *out = best_rfidx_mv;
}
int g_cost[7], g_have[7];
typedef struct { int32_t pred_error[7]; } TplDepStats_view;
void tpl_best_ref_slice(TplDepStats_view *tpl_stats, int_mv *single_mv, int *out_rf, int *out_cost, int_mv *out_mv) {
  int best_rf_idx = -1;
  int_mv best_mv[2];
  int32_t inter_cost;
  int32_t best_inter_cost = INT32_MAX;
  int rf_idx;
  for (rf_idx = 0; rf_idx < INTER_REFS_PER_FRAME; ++rf_idx) {
    if (!g_have[rf_idx]) continue;            /* tpl_data->ref_frame[rf_idx] == NULL (:633-637) */
    int_mv best_rfidx_mv;
    best_rfidx_mv = single_mv[rf_idx];
""" + adapt(tail) + """
  *out_rf = best_rf_idx; *out_cost = best_inter_cost; *out_mv = best_mv[0];
}
"""
    if "INTER_REFS_PER_FRAME" not in ev.globs:
        ev.define("INTER_REFS_PER_FRAME", "(7)")
    ev.load_text(src, "tpl_model.c:slices")
    rng = np.random.default_rng(20261004)
    cm_t = ev.typedefs["center_mv_t"]
    cpi = ev.interp.alloc(ev.typedefs["CPI_view"], True)
    arr = ev.interp.alloc(("arr", cm_t, 4), True)
    el = lambda k: R.Ptr(arr.buf, k, arr.t)
    prune_cases = []
    for c in range(160):
        cnt = int(rng.integers(1, 5))
        p = int(rng.integers(1, 4))
        base = int(rng.integers(0, 5000))
        sads = [base + int(rng.integers(0, 40)) * int(rng.integers(0, 2)) * int(rng.integers(1, 60)) for _ in range(cnt)]
        if c % 5 == 0 and cnt > 1:
            sads[1] = sads[0]                                     # equal SADs: the order must be kept
        if c % 7 == 0 and cnt > 2:
            sads[2] = sads[1]
        for k in range(4):
            ev.set(el(k), "sad", sads[k] if k < cnt else 2147483647)
            ev.set(el(k), "mv.as_mv.row", 10 + k); ev.set(el(k), "mv.as_mv.col", -10 - k)     # the row identifies the candidate afterwards
        ev.set(cpi, "sf.tpl_sf.prune_starting_mv", p)
        n_out = ev.call("tpl_prune_slice", cpi, R.Ptr(arr.buf, 0, arr.t), cnt)
        order = [ev.get(el(k), "mv.as_mv.row") - 10 for k in range(n_out)]
        prune_cases.append({"sads": sads, "prune": p, "order": order})
    mv_t = ev.typedefs["int_mv"]
    out_mv = ev.interp.alloc(mv_t, True)
    best_cases = []
    for c in range(80):
        cnt = int(rng.integers(1, 5))
        errs = [int(rng.integers(0, 1 << 32)) if rng.random() < 0.3 else int(rng.integers(100, 200)) for _ in range(cnt)]
        if c % 4 == 0 and cnt > 1:
            errs[-1] = errs[0]                                    # a tie: the first smallest wins
        if c % 9 == 0:
            errs[0] = 0xFFFFFFFF                                  # UINT32_MAX is never below bestsme: the MV stays { 0 }
        rows, cols = [int(v) for v in rng.integers(-200, 201, cnt)], [int(v) for v in rng.integers(-200, 201, cnt)]
        for k in range(cnt):
            ev.globs["g_me_err"].buf[k] = errs[k]; ev.globs["g_me_row"].buf[k] = rows[k]; ev.globs["g_me_col"].buf[k] = cols[k]
        ev.call("tpl_best_of_slice", cnt, out_mv)
        best_cases.append({"errs": errs, "rows": rows, "cols": cols, "best": [ev.get(out_mv, "as_mv.row"), ev.get(out_mv, "as_mv.col")]})
    stats = ev.interp.alloc(ev.typedefs["TplDepStats_view"], True)
    single = ev.interp.alloc(("arr", mv_t, 7), True)
    o_rf, o_cost = ev.array([0], "int"), ev.array([0], "int")
    ref_cases = []
    for c in range(80):
        have = [int(v) for v in (rng.random(7) < 0.7)]
        if c % 11 == 0:
            have = [0] * 7
        costs = [int(rng.integers(0, 3)) if rng.random() < 0.3 else int(rng.integers(0, 100000)) for _ in range(7)]
        if c % 3 == 0:
            costs[int(rng.integers(0, 7))] = min(costs)           # ties: the first smallest wins
        for r in range(7):
            ev.globs["g_cost"].buf[r] = costs[r]; ev.globs["g_have"].buf[r] = have[r]
            ev.set(R.Ptr(single.buf, r, single.t), "as_mv.row", 100 + r); ev.set(R.Ptr(single.buf, r, single.t), "as_mv.col", -100 - r)
            ev.set(stats, "pred_error[%d]" % r, -7)
        ev.call("tpl_best_ref_slice", stats, R.Ptr(single.buf, 0, single.t), o_rf, o_cost, out_mv)
        ref_cases.append({"have": have, "costs": costs, "best_rf": int(o_rf.buf[0]), "best_cost": int(o_cost.buf[0]),
                          "pred_error": [ev.get(stats, "pred_error[%d]" % r) for r in range(7)],
                          "best_mv_row": ev.get(out_mv, "as_mv.row") if o_rf.buf[0] >= 0 else None})
    save("ref_eval_tpl.npz", {}, {"prune": prune_cases, "best_of": best_cases, "best_ref": ref_cases})


if __name__ == "__main__":
    main()
