#!/bin/bash
# GPU box: SQ counters of the fast kernel at 2 and 3 workgroups per CU (k_fused_lean) and of k_fused_single.
SNOWTRI_LEAN_WG_PER_CU=3 bash scripts/pmc_probe.sh snowmocap_amd/libsnowtri.so lean3 2>&1 | grep -v "^$" | tail -12
SNOWTRI_LEAN_WG_PER_CU=2 bash scripts/pmc_probe.sh snowmocap_amd/libsnowtri.so lean2 2>&1 | grep -v "^$" | tail -12
