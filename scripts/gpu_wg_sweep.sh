#!/bin/bash
# GPU box: resident workgroups per CU of the fast kernels (SNOWTRI_LEAN_WG_PER_CU) on the two-stream bench workload.
for w in 2 1 3 2; do echo -n "wg/cu $w: "; SNOWTRI_LEAN_WG_PER_CU=$w python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra --no-per-frame --repeats 5 --large-frames 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%s step %.2f us value %.3e | own %.2f us' % (r['kernel'], d['ms_per_step']*1e3, d['value'], r['kernel_ms_mean']*1e3))"; done
