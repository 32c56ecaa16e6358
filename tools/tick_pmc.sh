#!/bin/bash
# per-tick VALU instructions and duration of the search launches of one GN-20 step, for APDGICP_NN_SPARSE = $1 (run inside gpurun)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/tickpmc_$1; rm -rf $out; mkdir -p $out
APDGICP_NN_SPARSE=$1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM -d $out -o k -- python3 bench.py --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --no-diagnostics > $out/log 2>&1
db=$(find $out -name "*.db" | head -1)
python3 tools/per_tick_pmc.py $db > $out/pt.txt; grep -c "" $out/pt.txt; grep "k_nn_" $out/pt.txt | head -66 | tail -22 | cut -c1-220; tail -3 $out/log
rm -rf $out
