#!/bin/bash
# Dev aid (GPU box): frames-per-tile x stream-count sweep of the bench step (10 000 frames per launch).
for T in 20 27 40; do for S in 2 3 4; do
  echo -n "T=$T streams=$S  "
  SNOWTRI_TILE_FRAMES=$T python bench.py --streams $S --steps 400 --warmup 40 --no-cpu-baseline --no-extra --repeats 3 --large-frames 0 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('value %.3e  ms/step %.4f  kernel_ms %.4f' % (d['value'], d['ms_per_step'], r['kernel_ms_mean']))"
done; done
