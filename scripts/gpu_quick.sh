#!/bin/bash
# Dev aid (GPU box): fast-path tests + the default bench line, condensed.
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "not multi and not cfg" 2>&1 | grep -E "passed|failed|error" | tail -3
for i in 1 2; do
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra --repeats 3 2>/dev/null | tail -1 | \
  python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; l=d['large_batch']; print('value %.3e  ms/step %.4f  kernel_ms %.4f  frac %.3f | large: %.3e joints/s  %.0f GB/s  frac %.3f' % (d['value'], d['ms_per_step'], r['kernel_ms_mean'], r['frac'], l['joints_per_s'], l['achieved_GBs'], l['frac']))"
done
