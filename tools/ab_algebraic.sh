#!/bin/bash
# the headline workload, the pooled LM loop and the single-registration latencies with and without APDGICP_FLAG_ALGEBRAIC_APD (bench.py --extra-flags 8),
# alternated on one box.   usage (inside gpurun): bash tools/ab_algebraic.sh [rounds=3]
rounds=${1:-3}
for i in $(seq $rounds); do
  for v in 0 8; do
    echo -n "flags=$v gn20-step "; timeout 300 python bench.py --no-cpu-baseline --no-diagnostics --extra-flags $v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], 'single-pair ms', d['single_pair']['ms_per_registration']['median'], 'ms/GN-iter cached', d['single_pair']['ms_per_gn_iteration'])"
    echo -n "flags=$v lm-loop "; timeout 300 python bench.py --kind loop --optimizer lm --no-cpu-baseline --no-diagnostics --extra-flags $v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], 'single-pair ms', d['single_pair']['ms_per_registration']['median'])"
  done
done
