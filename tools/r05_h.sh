#!/bin/bash
# round 5: the warped-motion predictor and the wedge helpers on the device
timeout 1500 python -m pytest tests/test_gpu_warp.py tests/test_gpu_wedge.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert|^E " | tail -12
