#!/bin/bash
# A/B of library builds on ONE box for the pooled LM loop line (bench.py --kind loop --optimizer lm): alternates riv-slam_amd/_<name>.bin copies
# of libapdgicp_hip.so.   usage (inside gpurun): bash tools/ab_lm.sh rounds name [name ...]     (variants: python tools/build_variant.py <name> [flags])
rounds=$1; shift
export APDGICP_ALLOW_STALE_LIB=1 APDGICP_ALLOW_VARIANT_LIB=1
cp riv-slam_amd/libapdgicp_hip.so riv-slam_amd/_keep.bin
trap 'cp riv-slam_amd/_keep.bin riv-slam_amd/libapdgicp_hip.so; rm -f riv-slam_amd/_keep.bin' EXIT
for i in $(seq $rounds); do
  for v in "$@"; do
    cp riv-slam_amd/_$v.bin riv-slam_amd/libapdgicp_hip.so
    echo -n "$v "; timeout 300 python bench.py --kind loop --optimizer lm --no-cpu-baseline --no-diagnostics | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
  done
done
