#!/bin/bash
# Dev aid (GPU box): A/B of k_frame_recompute builds on the multi-person configs.
mkdir -p gpurun_out/abr
for v in "" build/libsnowtri_u4.so build/libsnowtri_w2.so; do
  if [ -n "$v" ]; then export SNOWTRI_LIB=$PWD/snowmocap_amd/csrc/$v; else unset SNOWTRI_LIB; fi
  echo "== ${v:-default}"
  timeout 600 python scripts/bench_configs.py 2>/dev/null | grep '"kernel"'
done
unset SNOWTRI_LIB
echo "== default, 2 WG/CU"; SNOWTRI_RECOMPUTE_WG_PER_CU=2 timeout 600 python scripts/bench_configs.py 2>/dev/null | grep '"kernel"'
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "multi or recompute or g3 or general or cfg" 2>&1 | tail -3
