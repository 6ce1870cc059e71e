#!/bin/bash
# GPU box: multi-person shapes through the current build and the development builds under snowmocap_amd/csrc/ab/.
python scripts/bench_configs.py --full 2>&1 | grep '"kernel"' | cut -c1-140
for so in snowmocap_amd/csrc/ab/libsnowtri_*.so; do
  [ -f "$so" ] || continue
  echo "== $so"
  SNOWTRI_LIB=$PWD/$so python scripts/bench_configs.py --full 2>&1 | grep '"kernel"' | cut -c1-140
done
