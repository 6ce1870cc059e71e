#!/bin/bash
# Dev aid: bench A/B builds whose outputs are intentionally wrong (memory-path experiments): skips bench.py's fast-path assert
for so in snowmocap_amd/csrc/ab/libsnowtri_*.so; do
  echo "== $so"
  SNOWTRI_LIB=$PWD/$so python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extra --repeats 3 2>&1 | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; l=d['large_batch']; print('value %.3e  kernel_ms %.4f  frac %.3f | large: %.3e joints/s  %.0f GB/s  frac %.3f' % (d['value'], r['kernel_ms_mean'], r['frac'], l['joints_per_s'], l['achieved_GBs'], l['frac']))"
done
