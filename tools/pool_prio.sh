#!/bin/bash
# the C4 shard (tools/lm_loop_bench.py, 24 batches in flight) with the cloud stream at low priority / confined to a share of the CUs / more lanes,
# alternated on ONE box.   usage (inside gpurun): bash tools/pool_prio.sh [rounds=2]
rounds=${1:-2}
for i in $(seq $rounds); do
  for e in "" "APDGICP_POOL_CLOUD_PRIO=1" "APDGICP_POOL_CLOUD_CUS=192" "APDGICP_POOL_CLOUD_CUS=128" "APDGICP_POOL_CLOUD_CUS=96" "APDGICP_POOL_LANES=32 F_LIST=32" "APDGICP_POOL_LISTS=1"; do
    printf "[%s]  " "$e"
    env F_LIST=24 NO_POLLED=1 REPS=${REPS:-120} $e timeout 200 python3 tools/lm_loop_bench.py 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print({k: v for k, v in d.items() if 'ms_per_batch' in k or k == 'records_sha'})"
  done
done
