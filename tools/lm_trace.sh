#!/bin/bash
# kernel trace of the pooled LM loop (tools/lm_loop_bench.py, 16 batches in flight): per-kernel table + stream occupancy
# (lm_loop_bench.py: 48 warm-up batches, drained, then 144 timed ones with 24 in flight: the pool is full from batch 72 to 192 of the
# trace; rocpd_streams.py looks at batches 86 .. 186)
# usage (inside gpurun): bash tools/lm_trace.sh <tag>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
t=${1:-lm}; rm -rf gpurun_out/lt_$t
F_LIST=${F:-24} NO_POLLED=1 REPS=144 timeout 300 rocprofv3 --kernel-trace -d gpurun_out/lt_$t -o k -- python3 tools/lm_loop_bench.py > gpurun_out/lt_$t.log 2>&1
db=$(find gpurun_out/lt_$t -name "*.db" | head -1)
tail -1 gpurun_out/lt_$t.log | cut -c1-300
python3 tools/rocpd_streams.py $db 0.45 0.97 | tee gpurun_out/lt_$t.streams.txt
python3 tools/rocpd_summary.py $db "$t" > gpurun_out/lt_$t.md
rm -rf gpurun_out/lt_$t
