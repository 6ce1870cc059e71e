#!/bin/bash
# GPU box: A/B of the development builds under snowmocap_amd/csrc/ab/ (two interleaved rounds: run-to-run noise is ~1-2 %).
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; l=d['large_batch']; print('value %.3e  ms/step %.4f  kernel_ms %.4f (min %.4f)  frac %.3f | large: %.3e joints/s  %.0f GB/s  frac %.3f' % (d['value'], d['ms_per_step'], r['kernel_ms_mean'], r['kernel_ms_min'], r['frac'], l['joints_per_s'], l['achieved_GBs'], l['frac']))"; }
run() {
  echo "== $*"
  env "$@" python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra --repeats 3 2>&1 | tail -1 | show
}
for round in 1 2; do
  for so in snowmocap_amd/csrc/ab/libsnowtri_*.so; do
    w=2; case $so in *w4*) w=4;; esac
    run SNOWTRI_LIB=$PWD/$so SNOWTRI_LEAN_WG_PER_CU=$w
  done
done
