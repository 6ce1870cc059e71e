#!/bin/bash
mkdir -p gpurun_out/k
timeout 900 python -m pytest tests/test_gpu_parity.py -x -v -m gpu -k "$1" > gpurun_out/k/tests.log 2>&1; echo "tests rc=$?"
grep -E "PASSED|FAILED|ERROR|Fatal|fault|Memory" gpurun_out/k/tests.log | tail -40
