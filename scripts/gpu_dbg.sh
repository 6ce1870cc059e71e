#!/bin/bash
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep "passed\|failed\|Error\|error" | tail -5
python scripts/bench_configs.py --full --no-oracle 2>&1 | grep "^{" | cut -c1-120
