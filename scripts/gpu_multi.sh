#!/bin/bash
# GPU box: the multi-person kernel (k_frame_recompute): its parity tests, then the BASELINE multi-person shapes.
mkdir -p gpurun_out/multi
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "multi or random or general or workloads or golden or special or dlt" > gpurun_out/multi/tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/multi/tests.log
python scripts/bench_configs.py --full 2>&1 | grep "^{"
