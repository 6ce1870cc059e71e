#!/usr/bin/env python3
"""profiles/pmc_multi.json: measured VALU utilisation of the kernels of the multi-person calls, from the stdout of
scripts/pmc_multi.sh (gpurun_out/pmc_multi_stdout.txt), stamped with the hash of the kernel sources it was measured on.
bench.py quotes it as extra_workloads[*].roofline.valu_busy beside the NOMINAL fp64 fraction -- only while the hash matches
(the same rule as profiles/pmc_traffic.json).  Run on the GPU box right after pmc_multi.sh, before the bench lines are taken.

    python scripts/build_pmc_multi.py [gpurun_out/pmc_multi_stdout.txt]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from build_summary import multi_counters      # noqa: E402
from bench import kernel_source_hash          # noqa: E402

src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "pmc_multi_stdout.txt")
mc = multi_counters(src)
out = {"source_sha256": kernel_source_hash(),
       "what": "per kernel of ONE fused call (one stream: snowtri_ctx_set_split(1)): valu_busy = SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs), "
               "lds_conflict_ratio = SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS; rocprofv3 --pmc, scripts/pmc_multi.sh",
       "workloads": {cfg: {k: {"valu_busy": v.get("valu_busy"), "lds_conflict_ratio": v.get("lds_conflict_ratio"), "grid": v.get("grid")}
                           for k, v in ks.items() if v.get("valu_busy") is not None} for cfg, ks in mc.items()}}
json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_multi.json"), "w"), indent=1, sort_keys=True)
print("profiles/pmc_multi.json:", {c: len(k) for c, k in out["workloads"].items()})
