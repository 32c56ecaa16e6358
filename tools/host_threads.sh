#!/bin/bash
# bench.py --host-clouds (every step hands over 64 HOST clouds) under APDGICP_HOST_THREADS = threads packing host clouds (the caller included),
# alternated with the resident-input bench on ONE box.   usage (inside gpurun): bash tools/host_threads.sh [rounds=2]
rounds=${1:-2}
for i in $(seq $rounds); do
  echo -n "resident  "; timeout 300 python bench.py --no-cpu-baseline --no-diagnostics 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
  for t in ${THREADS:-4 8 16 32}; do
    echo -n "host clouds, APDGICP_HOST_THREADS=$t  "; APDGICP_HOST_THREADS=$t timeout 300 python bench.py --host-clouds --no-cpu-baseline --no-diagnostics 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
    APDGICP_HOST_THREADS=$t timeout 200 python3 tools/host_cost.py 200 --host-clouds 2>/dev/null | tail -1
  done
done
