#!/bin/bash
# Dev aid (GPU box): default build vs the A/B builds, two rounds each (run-to-run noise is ~2 %).
line() { python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra --repeats 3 2>/dev/null | tail -1 | \
  python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; l=d['large_batch']; print('value %.3e  kernel_ms %.4f  frac %.3f | large: %.3e joints/s  frac %.3f' % (d['value'], r['kernel_ms_mean'], r['frac'], l['joints_per_s'], l['frac']))"; }
for round in 1 2; do
  unset SNOWTRI_LIB; echo "== default"; line
  for so in snowmocap_amd/csrc/ab/libsnowtri_*.so; do export SNOWTRI_LIB=$PWD/$so; echo "== $so"; line; done
done
