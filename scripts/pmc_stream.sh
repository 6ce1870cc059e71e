#!/bin/bash
# Dev aid (GPU box): SQ counters of the streaming multi-person kernels on one config.  usage: pmc_stream.sh [cfg=3] [tag]
CFG=${1:-3}; TAG=${2:-cur}
ROOT=$PWD; OUT=$ROOT/gpurun_out/pmc_stream_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace -d $OUT/a -o a -- python $ROOT/scripts/bench_configs.py --full --no-oracle --only=$CFG > $OUT/a.log 2>&1
rocprofv3 --output-format csv --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --kernel-trace -d $OUT/b -o b -- python $ROOT/scripts/bench_configs.py --full --no-oracle --only=$CFG > $OUT/b.log 2>&1
cd $ROOT
python - <<PY
import csv, glob, collections
for tag in ("a", "b"):
    for path in glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        meta = {}
        for r in csv.DictReader(open(path)):
            kn = r["Kernel_Name"]
            if "snowtri::" in kn:
                key = kn.split("snowtri::")[1].split("(")[0][:40] + " grid=" + str(r.get("Grid_Size"))
                acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta[key] = "vgpr %s sgpr %s lds %s scratch %s wg %s" % (r.get("VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Scratch_Size"), r.get("Workgroup_Size"))
        for g, c in acc.items():
            print(tag, g, meta[g], {k: "%.4g" % (sum(v) / len(v)) for k, v in c.items()}, "n=%d" % len(next(iter(c.values()))))
PY
mkdir -p $OUT/csv; for t in a b; do cp $(find $OUT/$t -name "*counter_collection.csv" | head -1) $OUT/csv/pmc_$t.csv 2>/dev/null; done
rm -rf $OUT/a $OUT/b
