#!/bin/bash
# A/B of the neighbour-keeping skin on ONE box: ms per step of tools/phase_bench.py (full / ticks / cov) for several skins.
# usage (inside gpurun): bash tools/skin_sweep.sh
for cfg in "APDGICP_NN_SKIN=0" "APDGICP_NN_SKIN_REL=0.125 APDGICP_NN_SKIN_ABS=0.01" "APDGICP_NN_SKIN_REL=0.25 APDGICP_NN_SKIN_ABS=0.02" \
           "APDGICP_NN_SKIN_REL=0.4 APDGICP_NN_SKIN_ABS=0.03" "APDGICP_NN_SKIN_REL=0.6 APDGICP_NN_SKIN_ABS=0.05" "APDGICP_NN_SKIN_REL=1.0 APDGICP_NN_SKIN_ABS=0.08" \
           "APDGICP_NN_SKIN_REL=0.25 APDGICP_NN_SKIN_ABS=0.06" "APDGICP_NN_COMPACT=0"; do
  echo "[$cfg] $(env $cfg python tools/phase_bench.py 3 32 40 2>/dev/null | head -2 | tr '\n' ' ')"
done
