#!/bin/bash
# GPU box: rocprofv3 kernel durations of the fast kernel and of its timing-only variants (10 000-frame launches).
export SNOWTRI_BENCH_NOCHECK=1 TMPDIR=/tmp
ROOT=$PWD
cd /tmp
for so in $ROOT/snowmocap_amd/csrc/ab/libsnowtri_*.so; do
  nm=$(basename $so .so)
  SNOWTRI_LIB=$so rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/prof_$nm -o s -- python $ROOT/bench.py --frames 10000 --steps 100 --warmup 10 --no-cpu-baseline --no-extra --repeats 3 --large-frames 0 --streams 1 > /tmp/prof_$nm.log 2>&1
  f=$(find /tmp/prof_$nm -name "*kernel_stats.csv" | head -1)
  echo "== $nm"; grep "k_fused" $f | cut -c1-60,150-400 | head -2
  tail -1 /tmp/prof_$nm.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('   events: kernel_ms mean %.4f min %.4f  ms/step %.4f' % (r['kernel_ms_mean'], r['kernel_ms_min'], d['ms_per_step']))"
done
