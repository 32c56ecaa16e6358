#!/bin/bash
# the C4 shard (tools/lm_loop_bench.py, 24 batches in flight) over pair lists x cloud streams of the pool, alternated on ONE box
# usage (inside gpurun): bash tools/pool_matrix.sh [rounds=2] ["lists..."] ["cloud streams..."]
rounds=${1:-2}; lists=${2:-"2 3 4"}; cs=${3:-"1 2"}
for i in $(seq $rounds); do
  for l in $lists; do for c in $cs; do
    echo -n "lists=$l cloud_streams=$c  "
    APDGICP_POOL_LISTS=$l APDGICP_POOL_CLOUD_STREAMS=$c F_LIST=${F:-24} NO_POLLED=1 REPS=${REPS:-120} timeout 200 python3 tools/lm_loop_bench.py 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print({k: v for k, v in d.items() if 'ms_per_batch' in k or k == 'records_sha'})"
  done; done
done
