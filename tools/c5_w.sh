#!/bin/bash
# C5 (100k x 500k GN-20, cached covariances) under APDGICP_NN_W = waves per 64 source points of the search block
# usage (inside gpurun): bash tools/c5_w.sh
for w in 0 1 2 4 8; do echo -n "NN_W=$w  "; APDGICP_NN_W=$w timeout 120 python3 tools/c5_run.py 2>/dev/null | tail -1; done
