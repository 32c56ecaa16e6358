#!/bin/bash
# C3 (one scan against 8 cached keyframes: tests/measure/bench_configs.py C3) under two library builds riv-slam_amd/_<name>.bin, alternated on ONE box.
# usage (inside gpurun): bash tools/ab_c3.sh [old=ocml] [new=new]      (round 5: the shared atan2f against the device library's)
a=${1:-ocml}; b=${2:-new}
export APDGICP_ALLOW_STALE_LIB=1 APDGICP_ALLOW_VARIANT_LIB=1
cp riv-slam_amd/libapdgicp_hip.so riv-slam_amd/_keep.bin
trap 'cp riv-slam_amd/_keep.bin riv-slam_amd/libapdgicp_hip.so; rm -f riv-slam_amd/_keep.bin' EXIT   # never leave a variant in the product's place
for i in 1 2 3; do for v in $a $b; do cp riv-slam_amd/_$v.bin riv-slam_amd/libapdgicp_hip.so; echo -n "$v "; python tests/measure/bench_configs.py C3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['C3_1x8_lm_launch']['ms_per_batch'], d['C3_1x8_gn20']['ms_per_batch'])"; done; done
