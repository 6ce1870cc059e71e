#!/bin/bash
# Dev aid (GPU box): SQ counters of the fused kernel for one library build.
# usage: bash scripts/pmc_probe.sh <lib.so> <tag>
LIB=${1:-snowmocap_amd/libsnowtri.so}; TAG=${2:-probe}
ROOT=$PWD; OUT=$ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT
export TMPDIR=/tmp SNOWTRI_LIB=$ROOT/$LIB
BENCH="python $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra --repeats 3 --large-frames 500000"
cd /tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ_[A-Z0-9_]+|GRBM_[A-Z_]+|TCC_[A-Z0-9_]+\b|TCP_[A-Z0-9_]+)" | sort -u > $OUT/counters.txt
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace -d $OUT/a -o a -- $BENCH > $OUT/a.log 2>&1
rocprofv3 --output-format csv --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --kernel-trace -d $OUT/b -o b -- $BENCH > $OUT/b.log 2>&1
rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_ACTIVE_INST_MISC --kernel-trace -d $OUT/c -o c -- $BENCH > $OUT/c.log 2>&1
cd $ROOT
python - <<PY
import csv, glob, collections
for tag in "abc":
    for path in glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(path)):
            if "k_fused_" in r["Kernel_Name"]:
                g = r.get("Grid_Size") or r.get("Grid_Size_X") or "?"
                acc[g][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for g, c in acc.items():
            print(tag, "grid", g, {k: "%.4g" % (sum(v) / len(v)) for k, v in c.items()}, "n=%d" % len(next(iter(c.values()))))
PY
wc -l $OUT/counters.txt; tail -3 $OUT/a.log
rm -rf $OUT/a $OUT/b $OUT/c
