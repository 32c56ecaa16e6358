#!/bin/bash
# ms per step of the default bench for library builds riv-slam_amd/_<name>.bin at several numbers of steps in flight
# usage (inside gpurun): bash tools/ab_handles.sh "4 6 8" old new
hs=$1; shift
export APDGICP_ALLOW_STALE_LIB=1 APDGICP_ALLOW_VARIANT_LIB=1   # (the loader refuses a library with another source stamp or an experiment define)
cp riv-slam_amd/libapdgicp_hip.so riv-slam_amd/_keep.bin
trap 'cp riv-slam_amd/_keep.bin riv-slam_amd/libapdgicp_hip.so; rm -f riv-slam_amd/_keep.bin' EXIT   # also when interrupted: never leave a variant in the product's place
for v in "$@"; do
  cp riv-slam_amd/_$v.bin riv-slam_amd/libapdgicp_hip.so
  for h in $hs; do
    echo -n "$v handles $h: "; timeout 300 python bench.py --handles $h --no-cpu-baseline --no-diagnostics --repeats 8 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['timing']['step_latency_ms']['median'])"
  done
done
