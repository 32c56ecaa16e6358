#!/bin/bash
# Runs the -m gpu suite under every A/B switch listed at the end of docs/experiments.md (one line per configuration).
# usage (inside gpurun): bash tools/knob_matrix.sh
for cfg in "" "APDGICP_NN_MODE=brute" "APDGICP_KNN_MODE=brute" "APDGICP_NN_W=1" "APDGICP_NN_W=2" "APDGICP_NN_W=4" "APDGICP_NN_W=8" \
           "APDGICP_NN_S=2" "APDGICP_NN_S=4" "APDGICP_NN_GATE_CAP=0" "APDGICP_KNN_QPW=4" "APDGICP_KNN_QPW=8" "APDGICP_KNN_QPW=16" \
           "APDGICP_KNN_QPW=64" "APDGICP_KNN_COOP=0" "APDGICP_SORT_REG=0" "APDGICP_SORT_TILED=0" "APDGICP_FUSE=0" "APDGICP_STREAMS=1" "APDGICP_STREAMS=2" \
           "APDGICP_POLL_TICKS=1" "APDGICP_POLL_TICKS=3" "APDGICP_STATS=1" "APDGICP_NN_SKIN=0" "APDGICP_NN_COMPACT=0" \
           "APDGICP_NN_SKIN_REL=0.0 APDGICP_NN_SKIN_ABS=0.0" "APDGICP_NN_SKIN_REL=1.0 APDGICP_NN_SKIN_ABS=0.1" "APDGICP_NN_W=1 APDGICP_NN_COMPACT=0" \
           "APDGICP_FOLD_INIT=0" "APDGICP_FOLD_POLL=0" "APDGICP_DIRECT_STAGE=0" "APDGICP_SPLIT_REG=0" "APDGICP_KNN_QPW=16 APDGICP_SPLIT_REG=0" "APDGICP_NN_COOP_TAIL=0" "APDGICP_NN_W=1 APDGICP_NN_COOP_TAIL=0"; do
  res=$(env $cfg timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tr "\n" " ")
  echo "[$cfg] $res"
done
