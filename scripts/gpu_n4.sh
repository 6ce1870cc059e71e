#!/bin/bash
mkdir -p gpurun_out/n4
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "undistort or raw_frame" > gpurun_out/n4/tests.log 2>&1; echo "tests rc=$?"
tail -30 gpurun_out/n4/tests.log
