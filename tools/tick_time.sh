#!/bin/bash
# isolated per-tick durations (one handle, one step at a time: --handles 1 keeps three pair groups; here --pmc serialises instead)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/ticktime_$1; rm -rf $out; mkdir -p $out
APDGICP_NN_SPARSE=$1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES -d $out -o k -- python3 bench.py --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --no-diagnostics > $out/log 2>&1
db=$(find $out -name "*.db" | head -1)
echo "sparse=$1"; python3 tools/per_tick_time.py $db 40 20
rm -rf $out
