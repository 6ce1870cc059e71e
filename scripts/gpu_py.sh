#!/bin/bash
# usage: bash scripts/gpu_py.sh <script.py> [args]   (GPU box; output tail)
timeout 1700 python "$@" 2>&1 | tail -60
