#!/bin/bash
# GPU box: k_candidate_sums experiments -- per-kernel time (rocprofv3 stats) of bench_configs.py for "<lib>[,ENV=VAL...]:<cfg>" specs.
# usage: gpurun -- bash scripts/gpu_sums_exp.sh "rep1:3 rep2:3 rep1,SNOWTRI_SUMS_LDS_KB=80:3 production:5"
ROOT=$PWD; OUT=$ROOT/gpurun_out/sums_exp; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for spec in $1; do
  cfg=${spec##*:}; rest=${spec%:*}; lib=${rest%%,*}; envs=""
  if [ "$rest" != "$lib" ]; then envs=$(echo ${rest#*,} | tr ',' ' '); fi
  so=""; [ "$lib" != "production" ] && so=$ROOT/snowmocap_amd/csrc/ab/libsnowtri_$lib.so
  tag=$(echo $spec | tr ',=:' '___')
  env SNOWTRI_LIB=$so $envs rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/$tag -o s -- python $ROOT/scripts/bench_configs.py --full --no-oracle --only=$cfg > $OUT/$tag.log 2>&1
  python - <<PY
import csv, glob
out = []
for path in glob.glob("$OUT/$tag/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "snowtri::" in r["Name"] and float(r["Percentage"]) > 0.5:
            out.append("%s %.1f us x%s" % (r["Name"].split("snowtri::")[1].split("(")[0][:28], float(r["AverageNs"]) / 1e3, r["Calls"]))
print("$spec:", " | ".join(out))
PY
  rm -rf $OUT/$tag
done
