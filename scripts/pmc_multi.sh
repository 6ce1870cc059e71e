#!/bin/bash
# Dev aid (GPU box): SQ counters of k_frame_recompute on the cfg3 / cfg5 shapes.
ROOT=$PWD; OUT=$ROOT/gpurun_out/pmc_multi; mkdir -p $OUT
export TMPDIR=/tmp
# one stream per call: with the call's segments on two streams (the default for the small rigs) the kernels of the two
# segments overlap and a per-kernel duration would include the time a kernel waits for the other segment's workgroups
# (bench_configs.py --one-stream / bench_multi_hot.py --sweep-split=1: snowtri_ctx_set_split(1) -- the production library reads no environment)
cd /tmp
for CFG in 3 5; do
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace -d $OUT/a$CFG -o a -- python $ROOT/scripts/bench_configs.py --full --one-stream --only=$CFG > $OUT/a$CFG.log 2>&1
rocprofv3 --output-format csv --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --kernel-trace -d $OUT/b$CFG -o b -- python $ROOT/scripts/bench_configs.py --full --one-stream --only=$CFG > $OUT/b$CFG.log 2>&1
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/s$CFG -o s -- python $ROOT/scripts/bench_configs.py --full --one-stream --only=$CFG > $OUT/s$CFG.log 2>&1
cp $(find $OUT/s$CFG -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_cfg$CFG.csv
done
# the same 8 x 4 batch with float64 outputs (the reference's own output type): per-kernel table, one stream
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/s3f64 -o s -- python $ROOT/scripts/bench_multi_hot.py --only=3 --no-two --out64 --sweep-split=1 > $OUT/s3f64.log 2>&1
cp $(find $OUT/s3f64 -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_cfg3_f64out.csv
rm -rf $OUT/s3f64
cd $ROOT
python - <<PY
import csv, glob, collections
for tag in ("a3", "b3", "a5", "b5"):
    for path in glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(path)):
            if any(k in r["Kernel_Name"] for k in ("k_frame_", "k_cluster_", "k_candidate_", "k_associate")):
                key = r["Kernel_Name"].split("snowtri::")[1].split("(")[0][:40] + " grid=" + str(r.get("Grid_Size"))
                acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for g, c in acc.items():
            print(tag, g, {k: "%.4g" % (sum(v) / len(v)) for k, v in c.items()}, "n=%d" % len(next(iter(c.values()))))
PY
grep -h "^{" $OUT/a3.log $OUT/a5.log | head -8
# ONE detection per camera on 6 and 8 cameras (the lean kernels on the complete-graph item) and the headline rig with float64
# outputs: counters + stats of scripts/bench_single_rigs.py (10 000 and 200 000 frames)
bash scripts/pmc_any.sh single "k_fused_lean|k_fused_single" -- python scripts/bench_single_rigs.py --cams=6,8 --frames=10000,200000 --calls=20 > $OUT/single_rigs.txt 2>&1
grep -v "^    raw" $OUT/single_rigs.txt | grep "k_fused"
cp $ROOT/gpurun_out/pmc_single/kernel_stats.csv $OUT/kernel_stats_single_rigs.csv 2>/dev/null
# method = SNOWTRI_DLT (row N3): one detection per camera on the floor rig and on 8 cameras (k_fused_single<C,1>), and the 8 x 4
# batch behind the association (k_candidate_sums -> k_associate -> k_cluster_dlt -> k_person_scores, one stream)
bash scripts/pmc_any.sh dlt_single "k_dlt_coop|k_fused_single" -- python scripts/bench_single_rigs.py --cams=4 --rig=floor --dlt --calls=20 > $OUT/dlt_rigs.txt 2>&1
bash scripts/pmc_any.sh dlt_single8 "k_dlt_coop|k_fused_single" -- python scripts/bench_single_rigs.py --cams=8 --dlt --calls=20 >> $OUT/dlt_rigs.txt 2>&1
bash scripts/pmc_any.sh dlt_multi "k_cluster_dlt|k_candidate_sums<|k_associate|k_person_scores" -- python scripts/bench_configs.py --dlt --only=3 --one-stream --no-oracle >> $OUT/dlt_rigs.txt 2>&1
grep -v "^    raw" $OUT/dlt_rigs.txt | grep "k_fused\|k_dlt\|k_cluster\|k_candidate\|k_associate\|k_person"
cp $ROOT/gpurun_out/pmc_dlt_multi/kernel_stats.csv $OUT/kernel_stats_dlt_8x4.csv 2>/dev/null
cat $ROOT/gpurun_out/pmc_dlt_single/kernel_stats.csv > $OUT/kernel_stats_dlt_single.csv 2>/dev/null
tail -n +2 $ROOT/gpurun_out/pmc_dlt_single8/kernel_stats.csv >> $OUT/kernel_stats_dlt_single.csv 2>/dev/null
for t in a3 b3 a5 b5; do mkdir -p $OUT/csv; cp $(find $OUT/$t -name "*counter_collection.csv" | head -1) $OUT/csv/pmc_$t.csv 2>/dev/null; done
rm -rf $OUT/a3 $OUT/b3 $OUT/a5 $OUT/b5 $OUT/s3 $OUT/s5
