#!/bin/bash
# Dev aid (GPU box): per-kernel times of the next-row entries (N1/N2).
ROOT=$PWD; OUT=$ROOT/gpurun_out/n2prof; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/p -o p -- python $ROOT/scripts/bench_next_rows.py > $OUT/run.log 2>&1
cd $ROOT
f=$(find $OUT/p -name "*kernel_stats.csv" | head -1)
cp $f $OUT/kernel_stats_next_rows.csv
head -20 $OUT/kernel_stats_next_rows.csv | cut -c1-200
grep "^{" $OUT/run.log
rm -rf $OUT/p
