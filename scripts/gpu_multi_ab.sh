#!/bin/bash
# GPU box: multi-person path with and without the hand-over of complete-graph clusters to k_cluster_fuse.
# usage: gpurun --timeout 1500 -- bash scripts/gpu_multi_ab.sh [notest]
mkdir -p gpurun_out/multi
if [ "$1" != "notest" ]; then
  timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_shard_sizes.py tests/test_gpu_handover.py -x -q -m gpu > gpurun_out/multi/tests.log 2>&1; echo "tests rc=$?"; tail -15 gpurun_out/multi/tests.log
fi
for hm in 1 2 0; do
  echo "== SNOWTRI_HANDOVER_MODE=$hm"
  SNOWTRI_HANDOVER_MODE=$hm python scripts/bench_configs.py --full --no-oracle 2>&1 | grep "^{" | python -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln)
    print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in d.items() if k in ('config', 'label', 'frames', 'ms', 'ms_per_call', 'frames_per_s', 'max_err_m', 'count_ok', 'C', 'P')})
"
done
