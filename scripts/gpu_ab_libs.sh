#!/bin/bash
# GPU box: A/B of development builds of the library (snowmocap_amd/csrc/ab/libsnowtri_<tag>.so, SNOWTRI_LIB override) on the bench
# workload, interleaved twice.  usage: gpurun -- bash scripts/gpu_ab_libs.sh <tag> <tag> ...   (FRAMES=10000 by default)
F=${FRAMES:-10000}
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%s step %.2f us | own %.2f (min %.2f) step1 %.2f bracketed %.2f us | frac %.3f region %.3f' % (r['kernel'], d['ms_per_step']*1e3, r['kernel_ms_mean']*1e3, r['kernel_ms_min']*1e3, r['kernel_ms_step_one_stream']*1e3, r['kernel_ms_mean_bracketed']*1e3, r['frac'], d['roofline_region']['frac']))"; }
for rep in 1 2; do for tag in "$@"; do
  echo -n "$tag: "; SNOWTRI_LIB=$PWD/snowmocap_amd/csrc/ab/libsnowtri_$tag.so python bench.py --frames $F --steps 200 --warmup 20 --no-cpu-baseline --no-extra --no-per-frame --repeats 3 --large-frames 0 2>&1 | tail -1 | show
done; done
