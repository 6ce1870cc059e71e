#!/bin/bash
for F in 25 200 1000; do
  echo -n "F=$F  "
  python bench.py --frames $F --pool 8 --steps 60 --warmup 10 --streams 1 --no-cpu-baseline --no-extra --repeats 3 --large-frames 0 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('value %.3e  ms/step %.4f  kernel_ms %.4f (min %.4f)' % (d['value'], d['ms_per_step'], r['kernel_ms_mean'], r['kernel_ms_min']))"
done
