export TMPDIR=/tmp; ROOT=$PWD; OUT=$ROOT/gpurun_out/noloop; mkdir -p $OUT; cd /tmp
for n in noloop noepi base; do
SNOWTRI_BENCH_NOCHECK=1 SNOWTRI_LIB=$ROOT/snowmocap_amd/csrc/ab/libsnowtri_$n.so rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/$n -o s -- python $ROOT/bench.py --streams 1 --steps 200 --warmup 20 --repeats 3 --no-cpu-baseline --no-extra --no-per-frame --large-frames 0 > $OUT/$n.log 2>&1
python - <<PY
import csv, glob
for path in glob.glob("$OUT/$n/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "k_fused_lean" in r["Name"]: print("$n", "avg_us", float(r["AverageNs"])/1e3, "min", float(r["MinNs"])/1e3, "calls", r["Calls"])
PY
rm -rf $OUT/$n
done
