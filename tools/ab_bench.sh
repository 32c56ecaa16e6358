#!/bin/bash
# A/B of library builds on ONE box (boxes differ by ~1.5 %): alternates riv-slam_amd/_<name>.bin copies of libapdgicp_hip.so
# and prints ms_per_step of the default bench for each.   usage (inside gpurun): bash tools/ab_bench.sh old new [rounds]
# build a variant here with:  python tools/build_variant.py <name> [extra hipcc flags]
a=$1; b=$2; rounds=${3:-3}
export APDGICP_ALLOW_STALE_LIB=1   # (the loader refuses a library whose compiled-in source stamp is not that of the tree)
export APDGICP_ALLOW_VARIANT_LIB=1 # (... and one that lists an experiment define in apdgicp_build_flags())
cp riv-slam_amd/libapdgicp_hip.so riv-slam_amd/_keep.bin
trap 'cp riv-slam_amd/_keep.bin riv-slam_amd/libapdgicp_hip.so; rm -f riv-slam_amd/_keep.bin' EXIT   # also when interrupted: never leave a variant in the product's place
for i in $(seq $rounds); do
  for v in $a $b; do
    cp riv-slam_amd/_$v.bin riv-slam_amd/libapdgicp_hip.so
    echo -n "$v "; timeout 300 python bench.py --no-cpu-baseline --no-diagnostics | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
  done
done
