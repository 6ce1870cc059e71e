#!/bin/bash
# GPU box: kernel time of the fast path vs frames per launch (fixed cost vs per-frame cost).
# usage: gpurun -- bash scripts/gpu_sweep_frames.sh [env assignments...]
for F in 512 1024 2048 4096 6144 8192 10000 12288 16384 20480 40960 81920; do
  env "$@" python bench.py --frames $F --pool 8 --steps 100 --warmup 10 --no-cpu-baseline --no-extra --repeats 3 --large-frames 0 --streams 1 2>&1 | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('F %6d  kernel_ms mean %.4f min %.4f  ms/step %.4f' % ($F, r['kernel_ms_mean'], r['kernel_ms_min'], d['ms_per_step']))"
done
