#!/bin/bash
# arbitrary counters of the covariance k-NN launch alone (tools/knn_time.py under rocprofv3 --pmc); run inside gpurun.  usage: knn_pmc2.sh COUNTER...
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/knn_pmc2; rm -rf $out; mkdir -p $out
timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $out -o k -- python3 tools/knn_time.py > $out/log 2>&1
db=$(find $out -name "*.db" | head -1)
if [ -z "$db" ]; then tail -5 $out/log; exit 0; fi
python3 tools/rocpd_summary.py $db "knn" | grep "knn_cov_coop\|^| kernel" | tail -2
rm -rf $out
