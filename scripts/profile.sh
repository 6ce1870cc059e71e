#!/bin/bash
# GPU box: rocprofv3 kernel stats + HBM-traffic counters for the bench workload and the other kernels.
# usage: gpurun --timeout 1500 -- bash scripts/profile.sh <tag>      (outputs under gpurun_out/prof_<tag>/)
TAG=${1:-r03}
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# the bench commands whose JSON lines the numbers below belong to (configs[1]: 10 000 frames per launch)
BENCH="python $ROOT/bench.py --streams 1 --steps 200 --warmup 20 --repeats 3 --no-cpu-baseline --no-extra --no-per-frame --large-frames 0"   # one stream: launches do not overlap, so the trace duration is the kernel duration
LARGE="python $ROOT/bench.py --steps 5 --warmup 2 --repeats 1 --no-cpu-baseline --no-extra --no-per-frame --large-frames 2000000"
cd /tmp
# 1. per-kernel time (no counters)
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats -o stats -- $BENCH > $OUT/stats_bench.log 2>&1
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats_large -o stats -- $LARGE > $OUT/stats_large_bench.log 2>&1
# 2. HBM traffic: separate passes (TCC slots: FETCH_SIZE 3, WRITE_SIZE 2), calibration streams in the same run
rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o fetch -- python $ROOT/scripts/pmc_traffic_run.py > $OUT/pmc_fetch.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o write -- python $ROOT/scripts/pmc_traffic_run.py > $OUT/pmc_write.log 2>&1
# 3. SQ counters of the fast kernel on the 2 000 000-frame launch
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pmc_sq -o sq -- $LARGE > $OUT/pmc_sq.log 2>&1
rocprofv3 --output-format csv --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM --kernel-trace -d $OUT/pmc_sq2 -o sq -- $LARGE > $OUT/pmc_sq2.log 2>&1
# 4. the rows after the hot path (kernel stats only)
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats_next -o stats -- python $ROOT/scripts/bench_next_rows.py > $OUT/stats_next.log 2>&1
cd $ROOT
python scripts/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
grep -h '"metric"' $OUT/stats_bench.log $OUT/stats_large_bench.log > $OUT/bench_lines.jsonl
grep -h "^{" $OUT/stats_next.log > $OUT/next_rows.jsonl
# the individual durations of the large launches (the stats CSV averages them with the small warm-up launches)
python - <<PY
import csv, glob, json
rows = []
for path in glob.glob("$OUT/stats_large/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "k_fused_lean" in r["Kernel_Name"] or "k_fused_single" in r["Kernel_Name"]:
            rows.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
big = [d for d in rows if d > 1000000]
json.dump({"what": "rocprofv3 --kernel-trace durations (ns) of the 2 000 000-frame launches of bench.py --large-frames 2000000",
           "durations_ns": big, "small_launches_in_the_same_trace": len(rows) - len(big)}, open("$OUT/large_launches.json", "w"))
print("large launches:", big)
PY
# the bench lines without the profiler: the default run and the driver's command (with this run's PMC summary in place, so
# that the lines quote the traffic and the VALU count measured on exactly these kernel sources)
cp $OUT/pmc_traffic.json $ROOT/profiles/pmc_traffic.json
python bench.py 2>/dev/null | tail -1 > $OUT/bench_default.json
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/bench_steps20.json
# events against the trace of the same launches (one-stream command under the profiler, then without it)
bash scripts/gpu_event_check.sh > $OUT/event_check.log 2>&1; cp gpurun_out/evcheck/event_check.json $OUT/event_check.json
# keep the merge small: drop raw traces, keep CSV summaries
for d in stats stats_large stats_next; do cp $(find $OUT/$d -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_$d.csv 2>/dev/null; done
find $OUT -name "*.db" -delete 2>/dev/null
find $OUT -name "*kernel_trace.csv" -delete 2>/dev/null
find $OUT -name "*counter_collection.csv" -size +1M -delete 2>/dev/null
du -sh $OUT
