#!/bin/bash
# Dev aid (GPU box): multi-person throughput for the default build and every build under csrc/ab/.
unset SNOWTRI_LIB; echo "== default"; timeout 600 python scripts/bench_configs.py 2>/dev/null | grep '"kernel"'
for so in snowmocap_amd/csrc/ab/libsnowtri_*.so; do export SNOWTRI_LIB=$PWD/$so; echo "== $so"; timeout 600 python scripts/bench_configs.py 2>/dev/null | grep '"kernel"'; done
