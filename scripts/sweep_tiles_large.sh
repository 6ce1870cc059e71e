#!/bin/bash
# Dev aid (GPU box): frames-per-tile sweep of the fast kernel on the 2M-frame launch.
for T in 4 8 13 25 37 49; do
  echo -n "T=$T  "
  SNOWTRI_TILE_FRAMES=$T python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --repeats 3 --large-frames 1000000 2>&1 | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); l=d['large_batch']; print('large: %.3e joints/s  %.0f GB/s  frac %.3f  kernel_ms %.3f' % (l['joints_per_s'], l['achieved_GBs'], l['frac'], l['kernel_ms']))"
done
