#!/bin/bash
for so in snowmocap_amd/csrc/ab/libsnowtri_*.so; do
  echo "== $so"
  SNOWTRI_LIB=$PWD/$so bash scripts/gpu_multi_stats.sh 3 2>&1 | grep "snowtri::k_cand"
done
