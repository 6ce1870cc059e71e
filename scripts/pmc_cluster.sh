#!/bin/bash
# Dev aid (GPU box): SQ counters of the multi-person kernels (k_frame_recompute, k_cluster_fuse, k_cluster_members) on cfg3.
ROOT=$PWD; OUT=$ROOT/gpurun_out/pmc_cluster; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace -d $OUT/a -o a -- python $ROOT/scripts/bench_configs.py --full --only=${1:-3} > $OUT/a.log 2>&1
rocprofv3 --output-format csv --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --kernel-trace -d $OUT/b -o b -- python $ROOT/scripts/bench_configs.py --full --only=${1:-3} > $OUT/b.log 2>&1
cd $ROOT
python - <<PY
import csv, glob, collections
for tag in ("a", "b"):
    for path in glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(path)):
            if "snowtri::k_" in r["Kernel_Name"]:
                key = r["Kernel_Name"].split("snowtri::")[1][:28] + " grid=" + str(r.get("Grid_Size")) + " vgpr=" + str(r.get("VGPR_Count"))
                acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for g, c in acc.items():
            print(tag, g, {k: "%.4g" % (sum(v) / len(v)) for k, v in c.items()}, "n=%d" % len(next(iter(c.values()))))
PY
rm -rf $OUT/a $OUT/b
