#!/bin/bash
# GPU box: do the kernel-attached event pairs of bench.py (`roofline.kernel_ms_mean`) agree with rocprofv3's kernel trace of the
# SAME launches?  One-stream bench under the profiler; the trace's last 2 x N fused launches are the bracketed and the attached
# timing loops (bench.py runs them last when the other legs are switched off).
ROOT=$PWD; OUT=$ROOT/gpurun_out/evcheck; mkdir -p $OUT; export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --streams 1 --steps 200 --warmup 20 --repeats 3 --no-cpu-baseline --no-extra --no-per-frame --large-frames 0"
cd /tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats -o stats -- $BENCH > $OUT/bench_under_rocprof.log 2>&1
cd $ROOT
$BENCH 2>/dev/null | tail -1 > $OUT/bench_plain.json
python - <<PY
import csv, glob, json
rows = []
for path in glob.glob("$OUT/stats/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "k_fused_lean" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort()
dur = [e - s for s, e in rows]
gap = [rows[i + 1][0] - rows[i][1] for i in range(len(rows) - 1)]
line = json.loads([l for l in open("$OUT/bench_under_rocprof.log") if l.startswith("{")][-1])
plain = json.load(open("$OUT/bench_plain.json"))
mean = lambda x: sum(x) / max(1, len(x))
ti = line["roofline"]["trace_index"]     # [first, last) ordinal of each loop's launches among the process's fused calls = the trace's order
res = {"launches_in_trace": len(dur), "fused_calls_counted_by_bench": ti["total_fused_calls"], "trace_mean_all_us": mean(dur) / 1e3}
for name in ("bracketed", "attached", "step"):
    a, b = ti[name]
    res["trace_mean_%s_loop_us" % name] = mean(dur[a:b]) / 1e3
    res["trace_gap_%s_loop_us" % name] = mean(gap[a:b - 1]) / 1e3
res.update({
       "events_attached_under_rocprof_us": line["roofline"]["kernel_ms_mean"] * 1e3,
       "events_bracketed_under_rocprof_us": line["roofline"]["kernel_ms_mean_bracketed"] * 1e3,
       "events_attached_plain_us": plain["roofline"]["kernel_ms_mean"] * 1e3,
       "events_bracketed_plain_us": plain["roofline"]["kernel_ms_mean_bracketed"] * 1e3,
       "ms_per_step_one_stream_plain_us": plain["ms_per_step"] * 1e3, "ms_per_step_one_stream_under_rocprof_us": line["ms_per_step"] * 1e3})
json.dump(res, open("$OUT/event_check.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete
