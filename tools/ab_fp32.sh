#!/bin/bash
# the headline workload with and without APDGICP_FLAG_FP32_POINT_MATH, alternated on one box: ms per step, k_linearize's share via the line's own fields
# usage (inside gpurun): bash tools/ab_fp32.sh [rounds=3]
rounds=${1:-3}
for i in $(seq $rounds); do
  for v in "" "--fp32-point-math"; do
    echo -n "${v:-fp64-default} "; timeout 300 python bench.py --no-cpu-baseline --no-diagnostics $v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
  done
done
