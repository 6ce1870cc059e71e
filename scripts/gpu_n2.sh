#!/bin/bash
# Dev aid (GPU box): N2 tests + next-row throughput.
mkdir -p gpurun_out/n2
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "blender or smooth or main_loop" > gpurun_out/n2/tests.log 2>&1; echo "tests rc=$?"
tail -15 gpurun_out/n2/tests.log
timeout 600 python scripts/bench_next_rows.py > gpurun_out/n2/next_rows.jsonl 2> gpurun_out/n2/next_rows.err; echo "bench rc=$?"
cat gpurun_out/n2/next_rows.jsonl; tail -5 gpurun_out/n2/next_rows.err
