#!/bin/bash
# like ab_knn.sh, and after the rounds one APDGICP_STATS=1 run per variant (groups, steps, tightenings, phase timers per wave)
rounds=$1; shift
export APDGICP_ALLOW_STALE_LIB=1
export APDGICP_ALLOW_STALE_LIB=1 APDGICP_ALLOW_VARIANT_LIB=1
cp riv-slam_amd/libapdgicp_hip.so riv-slam_amd/_keep.bin
trap 'cp riv-slam_amd/_keep.bin riv-slam_amd/libapdgicp_hip.so; rm -f riv-slam_amd/_keep.bin' EXIT
for i in $(seq $rounds); do
  for v in "$@"; do
    cp riv-slam_amd/_$v.bin riv-slam_amd/libapdgicp_hip.so
    echo -n "$v: "; timeout 300 python tools/knn_time.py $KNN_ARGS 2>/dev/null | tail -1 | cut -c1-40
  done
done
for v in "$@"; do
  cp riv-slam_amd/_$v.bin riv-slam_amd/libapdgicp_hip.so
  echo "$v:"; APDGICP_STATS=1 timeout 300 python tools/knn_time.py $KNN_ARGS 2>/dev/null | tail -2
done
