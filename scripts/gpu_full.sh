#!/bin/bash
# Full validation on the GPU box: every GPU test, smoke(), the default bench line.
mkdir -p gpurun_out/full
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/full/tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/full/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py 2>gpurun_out/full/bench.err | tail -1 > gpurun_out/full/bench.json; cat gpurun_out/full/bench.json | cut -c1-1500
