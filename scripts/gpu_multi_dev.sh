#!/bin/bash
# GPU box, dev aid: the BASELINE multi-person shapes (8x4: 10 000 frames, 16x8: 12 000 frames) on the shipped library and on
# every development build under snowmocap_amd/csrc/ab/ (phase-split builds etc., bound through SNOWTRI_LIB).
# usage: gpurun --timeout 900 -- bash scripts/gpu_multi_dev.sh [cfgs, default "3 5"]
CFGS=${1:-"3 5"}
show() { grep "^{" | python -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln)
    if 'ms' in d: print('   cfg %d frames %d  %.4f ms  %.4g frames/s' % (d['cfg'], d['frames'], d['ms'], d['frames_per_s']))
"; }
for so in snowmocap_amd/libsnowtri.so snowmocap_amd/csrc/ab/libsnowtri_*.so; do
  [ -f "$so" ] || continue
  echo "== $so"
  for c in $CFGS; do SNOWTRI_LIB=$PWD/$so python scripts/bench_configs.py --full --no-oracle --only=$c 2>&1 | show; done
done
