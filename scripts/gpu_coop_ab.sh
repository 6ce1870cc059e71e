#!/bin/bash
# GPU box: k_fused_lean_coop (small launches) against k_fused_lean on the same frames.  usage: gpurun -- bash scripts/gpu_coop_ab.sh
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%s  value %.3e  ms/step %.4f  kernel_ms %.4f (min %.4f)  frac %.3f region %.3f' % (r['kernel'], d['value'], d['ms_per_step'], r['kernel_ms_mean'], r['kernel_ms_min'], r['frac'], d['roofline_region']['frac']))"; }
for F in 10000 4000 16000; do for coop in 1 0 1 0; do
  echo -n "F=$F coop=$coop: "; SNOWTRI_LEAN_COOP=$coop python bench.py --frames $F --steps 200 --warmup 20 --no-cpu-baseline --no-extra --no-per-frame --repeats 3 --large-frames 0 2>&1 | tail -1 | show
done; done
