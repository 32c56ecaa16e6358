#!/bin/bash
# C5 (tools/c5_run.py: 100k x 500k GN-20) under library builds riv-slam_amd/_<name>.bin, alternated on ONE box
# usage (inside gpurun): bash tools/ab_c5.sh [rounds] name ...
rounds=$1; shift
export APDGICP_ALLOW_STALE_LIB=1 APDGICP_ALLOW_VARIANT_LIB=1
cp riv-slam_amd/libapdgicp_hip.so riv-slam_amd/_keep.bin
trap 'cp riv-slam_amd/_keep.bin riv-slam_amd/libapdgicp_hip.so; rm -f riv-slam_amd/_keep.bin' EXIT
for i in $(seq $rounds); do
  for v in "$@"; do
    cp riv-slam_amd/_$v.bin riv-slam_amd/libapdgicp_hip.so
    echo -n "$v: "; timeout 200 python3 tools/c5_run.py 2>/dev/null | tail -1
  done
done
