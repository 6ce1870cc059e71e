#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_handover.py tests/test_gpu_shard_sizes.py -x -q -m gpu 2>&1 | grep "passed\|failed\|Error\|error\|assert" | tail -8
bash scripts/gpu_multi_stats.sh 5 2>&1 | grep "snowtri::\|cfg"
