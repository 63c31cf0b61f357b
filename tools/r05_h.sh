#!/bin/bash
# round 5: the widened OBMC sub-pel parity cases
timeout 1500 python -m pytest tests/test_gpu_obmc_subpel.py tests/test_gpu_compound_search.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | tail -8
