#!/bin/bash
# usage: ab_env.sh "<tag>[,ENV=V...]" ... : cfg5 hot loop per spec, REPS times interleaved
for rep in 1 2 3; do for spec in "$@"; do
  tag=${spec%%,*}; envs=""; [ "$spec" != "$tag" ] && envs=$(echo ${spec#*,} | tr ',' ' ')
  so=$PWD/snowmocap_amd/csrc/ab/libsnowtri_$tag.so
  env SNOWTRI_LIB=$so $envs python scripts/bench_multi_hot.py --only=${CFG:-5} 2>&1 | python scripts/show_hot.py | sed "s/^/$spec: /" | cut -c1-190
done; done
