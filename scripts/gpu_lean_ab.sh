#!/bin/bash
# GPU box: parity of the current build, then A/B of k_fused_lean against k_fused_single, of its residency and of
# the development builds under snowmocap_amd/csrc/ab/.
# usage: gpurun --timeout 1500 -- bash scripts/gpu_lean_ab.sh [notest]
mkdir -p gpurun_out/lean
if [ "$1" != "notest" ]; then
  timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/lean/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/lean/tests.log
fi
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; l=d['large_batch']; print('value %.3e  ms/step %.4f  kernel_ms %.4f (min %.4f)  frac %.3f | large: %.3e joints/s  %.0f GB/s  frac %.3f' % (d['value'], d['ms_per_step'], r['kernel_ms_mean'], r['kernel_ms_min'], r['frac'], l['joints_per_s'], l['achieved_GBs'], l['frac']))"; }
run() {
  echo "== $*"
  env "$@" python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra --no-per-frame --repeats 3 2>&1 | tail -1 | show
}
run SNOWTRI_LEAN_MODE=0
run SNOWTRI_LEAN_WG_PER_CU=1
run SNOWTRI_LEAN_WG_PER_CU=2
run SNOWTRI_LEAN_WG_PER_CU=3
run SNOWTRI_LEAN_WG_PER_CU=4
for so in snowmocap_amd/csrc/ab/libsnowtri_*.so; do
  [ -f "$so" ] || continue
  run SNOWTRI_LIB=$PWD/$so SNOWTRI_LEAN_WG_PER_CU=2
done
