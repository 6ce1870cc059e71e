#!/bin/bash
# GPU box: build and run the issue-rate microbenchmark.
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -w -o /tmp/fp64_rate scripts/ubench/fp64_rate.hip && /tmp/fp64_rate
