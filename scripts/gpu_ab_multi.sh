#!/bin/bash
# GPU box: interleaved A/B of development builds snowmocap_amd/csrc/ab/libsnowtri_<tag>.so on the multi-person shapes (hot loop,
# scripts/bench_multi_hot.py), each tag REPS times.  usage: gpurun -- bash scripts/gpu_ab_multi.sh "<cfg: 3|5|both>" <tag> <tag> ...
CFG=$1; shift
ONLY=""; [ "$CFG" != "both" ] && ONLY="--only=$CFG"
for rep in $(seq 1 ${REPS:-3}); do for tag in "$@"; do
  so=$PWD/snowmocap_amd/csrc/ab/libsnowtri_$tag.so; [ "$tag" == "production" ] && so=""
  SNOWTRI_LIB=$so python scripts/bench_multi_hot.py $ONLY 2>&1 | python scripts/show_hot.py | sed "s/^/$tag: /" | cut -c1-200
done; done
