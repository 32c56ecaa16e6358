#!/bin/bash
# counters of the issue microbenchmark's kernels (what one instruction of a class adds to SQ_ACTIVE_INST_VALU, SQ_BUSY_CYCLES ...); run inside gpurun
# usage: ubench_pmc.sh <name filter> COUNTER...
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
f=$1; shift
out=gpurun_out/ubench_pmc; rm -rf $out; mkdir -p $out
timeout 200 rocprofv3 --kernel-trace --pmc "$@" -d $out -o k -- ./riv-slam_amd/_ubench.bin $f > $out/log 2>&1
db=$(find $out -name "*.db" | head -1)
if [ -z "$db" ]; then tail -5 $out/log; exit 0; fi
python3 tools/rocpd_summary.py $db "k_" | grep "^| " 
rm -rf $out
