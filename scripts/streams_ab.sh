#!/bin/bash
for S in 1 2 3 4; do echo -n "streams=$S  "; python bench.py --streams $S --steps 400 --warmup 40 --no-cpu-baseline --no-extra --repeats 3 --large-frames 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('value %.3e  ms/step %.4f  kernel_ms %.4f' % (d['value'], d['ms_per_step'], r['kernel_ms_mean']))"; done
