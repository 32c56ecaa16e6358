#!/bin/bash
# A/B of the neighbour-keeping skin on ONE box: ms_per_step of the default bench for several (relative, absolute) skins.
# usage (inside gpurun): bash tools/skin_sweep.sh [extra bench args]
for cfg in "APDGICP_NN_SKIN=0" "APDGICP_NN_SKIN_REL=0.0 APDGICP_NN_SKIN_ABS=0.0" "APDGICP_NN_SKIN_REL=0.05 APDGICP_NN_SKIN_ABS=0.005" \
           "APDGICP_NN_SKIN_REL=0.125 APDGICP_NN_SKIN_ABS=0.01" "APDGICP_NN_SKIN_REL=0.25 APDGICP_NN_SKIN_ABS=0.01" "APDGICP_NN_SKIN_REL=0.25 APDGICP_NN_SKIN_ABS=0.03" \
           "APDGICP_NN_SKIN_REL=0.5 APDGICP_NN_SKIN_ABS=0.02" "APDGICP_NN_SKIN_REL=0.125 APDGICP_NN_SKIN_ABS=0.0"; do
  echo -n "[$cfg] "
  env $cfg timeout 300 python bench.py --no-cpu-baseline --no-diagnostics "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['avg_launch_ms'])"
done
