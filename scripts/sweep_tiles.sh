#!/bin/bash
# Dev aid (GPU box): sweep frames-per-tile of the fast kernel at the bench batch size.
for T in 5 10 13 20 21 25 39 40; do
  echo -n "T=$T  "
  SNOWTRI_TILE_FRAMES=$T python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extra --repeats 3 --large-frames 0 2>&1 | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('value %.3e  ms/step %.4f  kernel_ms %.4f (min %.4f) frac %.3f' % (d['value'], d['ms_per_step'], r['kernel_ms_mean'], r['kernel_ms_min'], r['frac']))"
done
