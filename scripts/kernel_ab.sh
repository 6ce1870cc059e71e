#!/bin/bash
# GPU box: per-kernel average time (rocprofv3 --kernel-trace --stats, one stream) of the 8 x 4 call for several builds of the library.
# usage: gpurun -- bash scripts/kernel_ab.sh <rounds> <lib tag> <lib tag> ...   (snowmocap_amd/csrc/ab/libsnowtri_<tag>.so; "prod" = the product)
ROOT=$PWD; export TMPDIR=/tmp PYTHONPATH=$ROOT
ROUNDS=$1; shift
OUT=$ROOT/gpurun_out/kab; rm -rf $OUT; mkdir -p $OUT
cd /tmp
for r in $(seq 1 $ROUNDS); do for t in "$@"; do
  if [ "$t" = prod ]; then unset SNOWTRI_LIB; else export SNOWTRI_LIB=$ROOT/snowmocap_amd/csrc/ab/libsnowtri_$t.so; fi
  rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/$t.$r -o s -- python $ROOT/scripts/bench_configs.py --only=3 --one-stream --no-oracle > $OUT/$t.$r.log 2>&1
  python - $OUT/$t.$r $t $r <<'PY'
import csv, glob, sys, re
d, t, r = sys.argv[1:4]
rows = []
for p in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
    for x in csv.DictReader(open(p)):
        if "snowtri::k_" in x["Name"] and float(x["AverageNs"]) > 20000:
            rows.append((re.sub(r"\(.*", "", x["Name"].replace("snowtri::", "").replace("void ", ""))[:44], float(x["AverageNs"]) / 1e3))
print(t, r, " | ".join("%s %.1f" % kv for kv in sorted(rows, key=lambda kv: -kv[1])), "| sum %.1f" % sum(v for _, v in rows))
PY
  rm -rf $OUT/$t.$r
done; done
