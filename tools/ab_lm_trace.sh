#!/bin/bash
# kernel tables of tools/lm_loop_bench.py (16 batches in flight) for two library builds riv-slam_amd/_<name>.bin on ONE box
# usage (inside gpurun): bash tools/ab_lm_trace.sh old new
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export APDGICP_ALLOW_STALE_LIB=1 APDGICP_ALLOW_VARIANT_LIB=1   # (the loader refuses a library with another source stamp or an experiment define)
cp riv-slam_amd/libapdgicp_hip.so riv-slam_amd/_keep.bin
trap 'cp riv-slam_amd/_keep.bin riv-slam_amd/libapdgicp_hip.so; rm -f riv-slam_amd/_keep.bin' EXIT   # also when interrupted: never leave a variant in the product's place
for v in "$@"; do
  cp riv-slam_amd/_$v.bin riv-slam_amd/libapdgicp_hip.so
  rm -rf gpurun_out/abt_$v
  F_LIST=16 NO_POLLED=1 REPS=64 timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/abt_$v -o k -- python3 tools/lm_loop_bench.py > gpurun_out/abt_$v.log 2>&1
  python3 tools/rocpd_summary.py $(find gpurun_out/abt_$v -name "*.db" | head -1) "$v" > gpurun_out/abt_$v.md
  rm -rf gpurun_out/abt_$v
  echo "== $v"; tail -1 gpurun_out/abt_$v.log; head -16 gpurun_out/abt_$v.md
done
