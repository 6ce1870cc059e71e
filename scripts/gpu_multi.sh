#!/bin/bash
# Dev aid (GPU box): multi-person parity tests + throughput of the BASELINE multi-person shapes.
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "multi or recompute or g3 or general or cfg or scenario or dlt" 2>&1 | tail -3
timeout 600 python scripts/bench_configs.py 2>/dev/null | grep '"kernel"'
timeout 600 python scripts/bench_configs.py 2>/dev/null | grep '"kernel"'
