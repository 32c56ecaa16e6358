#!/bin/bash
# A/B of library builds on ONE box for the covariance phase and the whole step (tools/phase_bench.py): usage (inside gpurun): bash tools/ab_phase.sh old new [rounds]
a=$1; b=$2; rounds=${3:-2}
export APDGICP_ALLOW_STALE_LIB=1 APDGICP_ALLOW_VARIANT_LIB=1   # (the loader refuses a library with another source stamp or an experiment define)
cp riv-slam_amd/libapdgicp_hip.so riv-slam_amd/_keep.bin
trap 'cp riv-slam_amd/_keep.bin riv-slam_amd/libapdgicp_hip.so; rm -f riv-slam_amd/_keep.bin' EXIT   # also when interrupted: never leave a variant in the product's place
for i in $(seq $rounds); do
  for v in $a $b; do
    cp riv-slam_amd/_$v.bin riv-slam_amd/libapdgicp_hip.so
    echo "$v $(timeout 300 python tools/phase_bench.py 4 32 60 2>/dev/null | grep -v amdgpu | awk '{printf "%s %s | ", $1, $4}')"
  done
done
