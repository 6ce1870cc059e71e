#!/bin/bash
# Dev aid (GPU box): kernel time vs frames per launch at a FIXED tile size (20 frames): the increment per
# 10 240 frames (= one more tile per resident workgroup) is the steady-state tile cost, the rest is launch overhead.
for F in 20 2560 5120 10240 20480 40960 81920; do
  echo -n "F=$F  "
  SNOWTRI_TILE_FRAMES=20 python bench.py --frames $F --pool 8 --steps 100 --warmup 10 --streams 1 --no-cpu-baseline --no-extra --repeats 3 --large-frames 0 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('kernel_ms %.4f (min %.4f)' % (r['kernel_ms_mean'], r['kernel_ms_min']))"
done
