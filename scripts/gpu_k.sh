#!/bin/bash
# usage: bash scripts/gpu_k.sh "<pytest -k expression>"
mkdir -p gpurun_out/k
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$1" > gpurun_out/k/tests.log 2>&1; echo "tests rc=$?"
tail -40 gpurun_out/k/tests.log
