#!/bin/bash
# A/B of library builds on ONE box for the covariance k-NN launch alone: alternates riv-slam_amd/_<name>.bin copies of libapdgicp_hip.so and
# prints tools/knn_time.py's sort + k-NN time (min of 8) for each.   usage (inside gpurun): bash tools/ab_knn.sh [rounds] name ...
rounds=$1; shift
export APDGICP_ALLOW_STALE_LIB=1 APDGICP_ALLOW_VARIANT_LIB=1
cp riv-slam_amd/libapdgicp_hip.so riv-slam_amd/_keep.bin
trap 'cp riv-slam_amd/_keep.bin riv-slam_amd/libapdgicp_hip.so; rm -f riv-slam_amd/_keep.bin' EXIT
for i in $(seq $rounds); do
  for v in "$@"; do
    cp riv-slam_amd/_$v.bin riv-slam_amd/libapdgicp_hip.so
    echo -n "$v: "; timeout 300 python tools/knn_time.py $KNN_ARGS | tail -1 | cut -c1-40
  done
done
