#!/bin/bash
# GPU box: SQ counters + kernel stats of ANY command, per kernel.  Three rocprofv3 passes (two --pmc groups, one --stats;
# never --pmc together with a trace domain other than --kernel-trace).
# usage: gpurun -- bash scripts/pmc_any.sh <tag> <kernel-name filter, | separated> -- <command ...>
TAG=$1; FILT=$2; shift 3
ROOT=$PWD; export PYTHONPATH=$ROOT:$PYTHONPATH
# (rocprofv3 runs from /tmp: repo-relative script paths of the command become absolute)
args=(); for a in "$@"; do case "$a" in scripts/*|bench.py|tests/*) args+=("$ROOT/$a");; *) args+=("$a");; esac; done; set -- "${args[@]}"
 OUT=$ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace -d $OUT/a -o a -- "$@" > $OUT/a.log 2>&1
rocprofv3 --output-format csv --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --kernel-trace -d $OUT/b -o b -- "$@" > $OUT/b.log 2>&1
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/s -o s -- "$@" > $OUT/s.log 2>&1
cd $ROOT
cp $(find $OUT/s -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null
for t in a b; do cp $(find $OUT/$t -name "*counter_collection.csv" | head -1) $OUT/pmc_$t.csv 2>/dev/null; done
python - "$OUT" "$FILT" <<'PY'
import csv, sys, collections, re
out, filt = sys.argv[1], sys.argv[2].split("|")
stats = {}
try:
    for r in csv.DictReader(open(out + "/kernel_stats.csv")):
        stats[r["Name"]] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
except Exception as e:
    print("no stats:", e)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for t in "ab":
    try:
        for r in csv.DictReader(open(out + "/pmc_%s.csv" % t)):
            if any(k in r["Kernel_Name"] for k in filt):
                acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    except Exception as e:
        print("no pmc", t, e)
for name, c in acc.items():
    m = {k: sum(v) / len(v) for k, v in c.items()}
    short = re.sub(r"\(.*", "", name.replace("snowtri::", "").replace("void ", ""))[:60]
    calls, avg = stats.get(name, (0, 0.0))
    gui = m.get("GRBM_GUI_ACTIVE", 0) / 8.0        # summed over the 8 XCDs
    waves = max(1.0, m.get("SQ_WAVES", 1))
    # VALU busy = SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)   (profiles/README.md, r02)
    busy = m.get("SQ_ACTIVE_INST_VALU", 0) * 4 / max(1.0, gui * 1024)
    line = "%-58s calls %5d avg %9.1f us | waves %.4g VALU/wave %.1f LDS/wave %.1f | VALU busy %.3f | clock %.0f MHz | LDS conflict/active %.3f | WAIT_INST_ANY/WAVE_CYCLES %.3f WAIT_ANY/WC %.3f" % (
        short, calls, avg, waves, m.get("SQ_INSTS_VALU", 0) / waves, m.get("SQ_INSTS_LDS", 0) / waves, busy, gui / max(1e-9, avg) if avg else 0,
        m.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, m.get("SQ_ACTIVE_INST_LDS", 1)),
        m.get("SQ_WAIT_INST_ANY", 0) / max(1.0, m.get("SQ_WAVE_CYCLES", 1)), m.get("SQ_WAIT_ANY", 0) / max(1.0, m.get("SQ_WAVE_CYCLES", 1)))
    print(line)
    print("    raw:", {k: "%.4g" % v for k, v in m.items()})
PY
rm -rf $OUT/a $OUT/b $OUT/s
