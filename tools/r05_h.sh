#!/bin/bash
# round 5: the RD-path additions of the round's second half on the device
timeout 1500 python -m pytest tests/test_gpu_lr_apply.py tests/test_gpu_lrstats.py tests/test_gpu_qm_adaptive.py tests/test_gpu_qm_fp.py tests/test_gpu_xform_quant.py tests/test_gpu_qm.py tests/test_gpu_scale.py tests/test_gpu_sgr.py tests/test_gpu_proj.py tests/test_gpu_warp.py tests/test_gpu_wedge.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert|^E " | tail -12
