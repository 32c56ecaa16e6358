#!/bin/bash
# Instruction counters of the covariance k-NN launch alone (tools/knn_time.py under rocprofv3 --pmc); run inside gpurun.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/knn_pmc; rm -rf $out; mkdir -p $out
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $out -o k -- python3 tools/knn_time.py > $out/log 2>&1
db=$(find $out -name "*.db" | head -1)
python3 tools/rocpd_summary.py $db "knn" | grep "knn_cov_coop\|^| kernel"
