#!/bin/bash
mkdir -p gpurun_out/n4
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "undistort or raw_frame or track_pipeline or c_abi" > gpurun_out/n4/tests.log 2>&1; echo "tests rc=$?"
tail -5 gpurun_out/n4/tests.log
timeout 600 python scripts/bench_next_rows.py 2>/dev/null | grep "N4\|pipeline"
