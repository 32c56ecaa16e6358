#!/bin/bash
# stream occupancy of the default bench's timed region (GN-20, four steps in flight)   usage (inside gpurun): bash tools/gn_trace.sh [bench args]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/gt
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/gt -o k -- python3 bench.py --no-cpu-baseline --no-diagnostics --repeats 40 "$@" > gpurun_out/gt.log 2>&1
tail -1 gpurun_out/gt.log | cut -c1-200
python3 tools/rocpd_streams.py $(find gpurun_out/gt -name "*.db" | head -1) 0.3
rm -rf gpurun_out/gt
