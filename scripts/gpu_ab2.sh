#!/bin/bash
# GPU box: same-box A/B of the development builds under snowmocap_amd/csrc/ab/ on the bench workload, two rounds.
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; l=d.get('large_batch') or {}; print('%s  value %.3e  kernel_ms %.4f (min %.4f)  frac %.3f region %.3f large %.3f' % (r['kernel'], d['value'], r['kernel_ms_mean'], r['kernel_ms_min'], r['frac'], d['roofline_region']['frac'], l.get('frac', 0)))"; }
for rep in 1 2; do for so in snowmocap_amd/csrc/ab/libsnowtri_*.so; do
  echo -n "$(basename $so): "; env "$@" SNOWTRI_LIB=$PWD/$so python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra --no-per-frame --repeats 3 2>&1 | tail -1 | show
done; done
