#!/bin/bash
# GPU box: per-kernel time of the multi-person path on the cfg3 (8 x 4) shape.   usage: gpurun -- bash scripts/gpu_multi_stats.sh [cfg]
CFG=${1:-3}
ROOT=$PWD; OUT=$ROOT/gpurun_out/multi_stats; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/s -o s -- python $ROOT/scripts/bench_configs.py --full --no-oracle --only=$CFG > $OUT/s.log 2>&1
grep "^{" $OUT/s.log | head -3
python - <<PY
import csv, glob
for path in glob.glob("$OUT/s/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        print(r["Name"][:70], r["Calls"], "avg_us", float(r["AverageNs"]) / 1e3, "pct", r["Percentage"])
PY
rm -rf $OUT/s
