#!/bin/bash
# dev: which hand-over mode fails the golden ring8x4 case; per-kernel times of the streaming path
for hm in 0 2 1; do
  echo "== HANDOVER_MODE=$hm"
  SNOWTRI_HANDOVER_MODE=$hm timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden" 2>&1 | tail -3
done
bash scripts/gpu_multi_stats.sh 3 2>&1 | tail -8
bash scripts/gpu_multi_stats.sh 5 2>&1 | tail -8
