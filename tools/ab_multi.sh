#!/bin/bash
# like ab_bench.sh for any number of variants: usage (inside gpurun): bash tools/ab_multi.sh rounds name1 name2 ...   (riv-slam_amd/_<name>.bin)
rounds=$1; shift
export APDGICP_ALLOW_STALE_LIB=1 APDGICP_ALLOW_VARIANT_LIB=1
cp riv-slam_amd/libapdgicp_hip.so riv-slam_amd/_keep.bin
trap 'cp riv-slam_amd/_keep.bin riv-slam_amd/libapdgicp_hip.so; rm -f riv-slam_amd/_keep.bin' EXIT   # also when interrupted: never leave a variant in the product's place
for i in $(seq $rounds); do
  for v in "$@"; do
    cp riv-slam_amd/_$v.bin riv-slam_amd/libapdgicp_hip.so
    echo -n "$v "; timeout 300 python bench.py --no-cpu-baseline --no-diagnostics 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['timing'].get('p10'), d['timing'].get('hip_event_ms_per_step'))"
  done
done
