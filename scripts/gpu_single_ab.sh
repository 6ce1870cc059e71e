#!/bin/bash
# GPU box: A/B of the lean kernels on cluster_item (6-8 cameras): development builds csrc/ab/libsnowtri_<tag>.so, interleaved twice.
# usage: gpurun -- bash scripts/gpu_single_ab.sh <tag> ...      (tag "prod" = snowmocap_amd/libsnowtri.so)
CAMS=${CAMS:-6,8}; FRAMES=${FRAMES:-10000,200000}
for rep in 1 2; do for tag in "$@"; do
  lib=$PWD/snowmocap_amd/csrc/ab/libsnowtri_$tag.so; [ "$tag" = prod ] && lib=$PWD/snowmocap_amd/libsnowtri.so
  SNOWTRI_LIB=$lib python scripts/bench_single_rigs.py --cams=$CAMS --frames=$FRAMES 2>&1 | python -c "
import sys, json
for ln in sys.stdin:
    try: d = json.loads(ln)
    except Exception: print(ln.rstrip()); continue
    print('$tag %-52s %8.1f us  %.3e joints/s  fp64 frac %.3f  fast %d  %s' % (d['workload'][:52], d['ms_per_call']*1e3, d['joints_per_s'], d['roofline']['frac'], d['fast_frames'], d['kernels'][:40]))"
done; done
