#!/bin/bash
mkdir -p gpurun_out/dlt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dlt or DLT" > gpurun_out/dlt/tests.log 2>&1; echo "tests rc=$?"
tail -30 gpurun_out/dlt/tests.log
timeout 600 python bench.py --method dlt --steps 10 --warmup 3 --no-cpu-baseline --no-extra --repeats 3 --large-frames 500000 2>/dev/null | tail -1 > gpurun_out/dlt/bench_dlt.json; cat gpurun_out/dlt/bench_dlt.json
