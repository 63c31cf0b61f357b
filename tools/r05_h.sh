#!/bin/bash
# round 5: the wedge-mask helpers on the device
timeout 1500 python -m pytest tests/test_gpu_wedge.py tests/test_gpu_rd_helpers.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | tail -8
