#!/bin/bash
# round 5: the self-guided projection statistics, the warped-motion predictor and the wedge helpers on the device
timeout 1500 python -m pytest tests/test_gpu_scale.py tests/test_gpu_sgr.py tests/test_gpu_proj.py tests/test_gpu_warp.py tests/test_gpu_wedge.py tests/test_gpu_lrstats.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert|^E " | tail -12
