#!/bin/bash
# One GPU-box cycle: parity tests, then the A/B occupancy builds, then the tile sweep.
python -m pytest tests -m gpu -q 2>&1 | tail -6
bash scripts/ab_variants.sh
bash scripts/sweep_tiles.sh
