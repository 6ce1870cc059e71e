#!/bin/bash
# Copy the summaries of scripts/profile.sh <tag> + scripts/pmc_multi.sh (under gpurun_out/) into profiles/<tag>/.
TAG=${1:-r06}; SRC=gpurun_out/prof_$TAG; DST=profiles/$TAG; mkdir -p $DST
cp $SRC/summary.txt $SRC/bench_lines.jsonl $SRC/next_rows.jsonl $SRC/pmc_traffic.json $DST/
cp $SRC/kernel_stats_stats.csv $DST/kernel_stats_cfg2_10k.csv
cp $SRC/kernel_stats_stats_large.csv $DST/kernel_stats_large_2M.csv
cp $SRC/kernel_stats_stats_next.csv $DST/kernel_stats_next_rows.csv
cp $SRC/pmc_traffic.json profiles/pmc_traffic.json
cp $SRC/event_check.json $DST/ 2>/dev/null
if [ -d gpurun_out/pmc_multi ]; then   # (scripts/pmc_multi.sh was run too: the multi-person kernels)
cp gpurun_out/pmc_multi/kernel_stats_cfg3.csv $DST/kernel_stats_multi_cfg3_8x4_10000.csv
cp gpurun_out/pmc_multi/kernel_stats_cfg5.csv $DST/kernel_stats_multi_cfg5_16x8_12000.csv
[ -f gpurun_out/pmc_multi/kernel_stats_cfg3_f64out.csv ] && cp gpurun_out/pmc_multi/kernel_stats_cfg3_f64out.csv $DST/kernel_stats_multi_cfg3_8x4_10000_f64out.csv
grep "^a\|^b\|^{" gpurun_out/pmc_multi_stdout.txt > $DST/multi_person_counters.txt
[ -f gpurun_out/pmc_multi/single_rigs.txt ] && grep "k_fused\|raw:" gpurun_out/pmc_multi/single_rigs.txt > $DST/single_rigs_counters.txt
[ -f gpurun_out/pmc_multi/kernel_stats_single_rigs.csv ] && cp gpurun_out/pmc_multi/kernel_stats_single_rigs.csv $DST/kernel_stats_single_rigs.csv
[ -f gpurun_out/pmc_multi/dlt_rigs.txt ] && grep "k_fused\|k_dlt\|k_cluster\|k_candidate\|k_associate\|k_person\|raw:" gpurun_out/pmc_multi/dlt_rigs.txt > $DST/dlt_counters.txt   # method = SNOWTRI_DLT: SQ counters per kernel
[ -f gpurun_out/pmc_multi/kernel_stats_dlt_8x4.csv ] && cp gpurun_out/pmc_multi/kernel_stats_dlt_8x4.csv gpurun_out/pmc_multi/kernel_stats_dlt_single.csv $DST/
[ -f profiles/pmc_multi.json ] && cp profiles/pmc_multi.json $DST/pmc_multi.json
fi
cp $SRC/bench_default.json $SRC/bench_steps20.json $SRC/large_launches.json $DST/
[ -f gpurun_out/dist/dist_lines.jsonl ] && cp gpurun_out/dist/dist_lines.jsonl $DST/dist_lines.jsonl   # bench.py's N > 1 path on the one GPU: --force-dist (RCCL, one rank) and --gpus 2 --one-device (gloo)
[ -f gpurun_out/multiproc_full.jsonl ] && cp gpurun_out/multiproc_full.jsonl $DST/multiproc_full.jsonl   # tests/test_gpu_multiproc.py: configs[3] / [4] at full size, 8 ranks on the one GPU
python - <<PY
import csv, os
for t in ("a3", "b3", "a5", "b5"):
    src = f"gpurun_out/pmc_multi/csv/pmc_{t}.csv"
    if not os.path.exists(src):
        continue
    keep = [r for r in csv.DictReader(open(src)) if any(k in r["Kernel_Name"] for k in ("k_frame_recompute", "k_cluster_", "k_candidate_", "k_associate"))]
    cfg = {"3": "cfg3_8x4_10000", "5": "cfg5_16x8_12000"}[t[1]]
    dst = f"$DST/pmc_multi_{cfg}_{'sq' if t[0] == 'a' else 'lds'}.csv"
    with open(dst, "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=["Kernel_Name", "Grid_Size", "LDS_Block_Size", "VGPR_Count", "SGPR_Count", "Counter_Name", "Counter_Value"], extrasaction="ignore")
        w.writeheader()
        for r in keep:
            r = dict(r); r["Kernel_Name"] = r["Kernel_Name"][:60]; w.writerow(r)
PY

cp $SRC/summary.json $DST/counters.json
# the clock / socket-power record of scripts/power_trace.py, and the newest record of the DRIVER's own bench run (an
# independently run figure beside ours in the tables: it belongs to the previous round's sources until the round ends)
[ -f gpurun_out/power/power_trace.json ] && cp gpurun_out/power/power_trace.json $DST/power_trace.json && cp gpurun_out/power/samples.csv $DST/power_samples.csv
NEWEST=$(ls BENCH_r*.json 2>/dev/null | sort | tail -1)
[ -n "$NEWEST" ] && python - <<PY
import json
d = json.load(open("$NEWEST"))
line = d.get("parsed")
run = d.get("run")
if isinstance(run, dict):      # the whole JSON line bench.py printed (the driver keeps the contract keys only in its parsed record)
    full = [l for l in (run.get("stdout_tail") or "").splitlines() if l.startswith('{"metric"')]
    try:
        line = json.loads(full[-1])
    except Exception:
        pass
json.dump({"file": "$NEWEST", "head": d.get("head"), "cmd": d.get("cmd"), "parsed": line}, open("$DST/driver_bench.json", "w"))
PY
python scripts/build_summary.py $TAG
python scripts/make_tables.py --write
ls $DST | wc -l
