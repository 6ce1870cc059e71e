#!/bin/bash
# One GPU-box cycle: parity tests, default-build bench, then any A/B builds under csrc/ab/.
python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|error" | tail -3
echo "== default build"; python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra --repeats 3 2>&1 | tail -1 | \
  python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; l=d['large_batch']; print('value %.3e  ms/step %.4f  kernel_ms %.4f  frac %.3f | large: %.3e joints/s  %.0f GB/s  frac %.3f' % (d['value'], d['ms_per_step'], r['kernel_ms_mean'], r['frac'], l['joints_per_s'], l['achieved_GBs'], l['frac']))"
ls snowmocap_amd/csrc/ab/*.so >/dev/null 2>&1 && bash scripts/ab_variants.sh
[ -n "$SWEEP" ] && bash scripts/sweep_tiles.sh
