#!/bin/bash
cd scripts/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/fp64_rate fp64_rate.hip && /tmp/fp64_rate
