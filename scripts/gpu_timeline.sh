#!/bin/bash
# GPU box, dev aid: kernel timeline (start offset, duration) of the LAST fused call of bench_configs on one config.
CFG=${1:-5}
ROOT=$PWD; OUT=$ROOT/gpurun_out/timeline; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --output-format csv --kernel-trace --memory-copy-trace -d $OUT/t -o t -- python $ROOT/scripts/bench_configs.py --full --no-oracle --only=$CFG > $OUT/t.log 2>&1
grep "^{" $OUT/t.log | head -2
python - <<PY
import csv, glob
rows = []
for path in glob.glob("$OUT/t/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
rows.sort()
# last 20 kernels
last = rows[-22:]
t0 = last[0][0]
prev_end = None
for s, e, n in last:
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print("%10.1f us  dur %9.1f us  gap %8.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, n))
    prev_end = e
PY
rm -rf $OUT/t
