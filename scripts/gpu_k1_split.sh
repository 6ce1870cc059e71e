#!/bin/bash
# GPU box: per-kernel time of the multi-person path for the production build and for development builds under
# snowmocap_amd/csrc/ab/ (timing-only variants: -DSNOWTRI_K1_NOFILL, -DSNOWTRI_K1_NOSOLVE, ...).
# usage: gpurun -- bash scripts/gpu_k1_split.sh [cfgs, default "3 5"]
CFGS=${1:-"3 5"}
ROOT=$PWD; OUT=$ROOT/gpurun_out/k1_split; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for so in "" $ROOT/snowmocap_amd/csrc/ab/libsnowtri_*.so; do
  for CFG in $CFGS; do
    tag=$(basename "${so:-production}" .so)_$CFG
    SNOWTRI_LIB=$so rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/$tag -o s -- python $ROOT/scripts/bench_configs.py --full --no-oracle --only=$CFG > $OUT/$tag.log 2>&1
    python - <<PY
import csv, glob
out = []
for path in glob.glob("$OUT/$tag/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "snowtri::" in r["Name"] and float(r["Percentage"]) > 0.5:
            out.append("%s %.1f us x%s" % (r["Name"].split("snowtri::")[1].split("(")[0][:28], float(r["AverageNs"]) / 1e3, r["Calls"]))
print("$tag:", " | ".join(out))
PY
    rm -rf $OUT/$tag
  done
done
