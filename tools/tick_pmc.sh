#!/bin/bash
# Per-tick instruction counters of one GN-20 batch (tools/one_batch.py under rocprofv3 --pmc); run inside gpurun.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/tick_pmc; rm -rf $out; mkdir -p $out
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $out -o k -- python3 tools/one_batch.py 32 odometry 1 > $out/log 2>&1
db=$(find $out -name "*.db" | head -1)
python3 tools/per_tick_pmc.py $db
