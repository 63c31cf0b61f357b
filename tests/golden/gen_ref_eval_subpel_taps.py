#!/usr/bin/env python3
"""Golden vectors for av1_find_best_sub_pixel_tree with the up-sampled prediction error of the OTHER two SUBPEL_SEARCH_TYPEs, obtained by
interpreting the reference (build container only; harness of gen_ref_eval_compound_subpel.py): USE_4_TAPS -- av1_get_filter ->
av1_interp_4tap[EIGHTTAP_REGULAR] = av1_sub_pel_filters_4, what speed 1 - 2 of the good-quality presets run (speed_features.c:957) -- and
USE_2_TAPS (av1_bilinear_filters), through upsampled_pref_error -> aom_[highbd_]upsampled_pred_c / comp_avg_upsampled_pred /
comp_mask_upsampled_pred (mcomp.c:2339-2428; reconinter_enc.c:424-700), single-reference and compound.  ref_eval_mcomp.npz /
ref_eval_compound_subpel.npz hold the USE_8_TAPS and USE_2_TAPS_ORIG forms.

Output: tests/golden/ref_eval_subpel_taps.npz, and tests/golden/ref_eval_obmc_subpel_taps.npz: av1_find_best_obmc_sub_pixel_tree_up with the same two
types (upsampled_obmc_pref_error, mcomp.c:3314-3357; harness of gen_ref_eval_obmc_subpel.py).
"""
import gen_ref_eval_compound_subpel as CS
import gen_ref_eval_obmc_subpel as OS

if __name__ == "__main__":
    plan = []
    for bd in (8, 10):
        for (w, h) in ((8, 8), (16, 16), (16, 8)):
            for sst in (2, 1):
                plan.append((bd, w, h, 2, sst, 0, 0))            # single reference
                plan.append((bd, w, h, 2, sst, 0, 1))            # averaged compound
                if (w, h) != (16, 8):
                    plan.append((bd, w, h, 2, sst, 1, 1))        # masked compound
    CS.main(plan, "ref_eval_subpel_taps.npz", seed=20261401)
    OS.main([(bd, w, h, stype, trial) for bd in (8, 10) for (w, h) in ((8, 8), (16, 16), (16, 8)) for stype in ("USE_4_TAPS", "USE_2_TAPS") for trial in range(2)],
            "ref_eval_obmc_subpel_taps.npz", seed=20261402)
