#!/bin/bash
# Regenerates the judged evidence on the GPU box: kernel stats, the two PMC passes, roofline traffic, and the bench line.
# usage: gpurun -- 'bash tools/refresh_evidence.sh'; then copy gpurun_out/ev/* into profiles/ (named per round)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ev
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/ev/ks -o k -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-diagnostics > gpurun_out/ev/ks.log 2>&1
python3 tools/rocpd_summary.py $(find gpurun_out/ev/ks -name "*.db" | head -1) "round 1 (final): python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-diagnostics under rocprofv3 --kernel-trace --stats" > gpurun_out/ev/kernel_stats.md
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/ev/pf -o k -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-diagnostics > gpurun_out/ev/pf.log 2>&1
python3 tools/rocpd_summary.py $(find gpurun_out/ev/pf -name "*.db" | head -1) "round 1 (final) PMC pass 1: FETCH_SIZE (KB per dispatch; double it on gfx950 for wide coalesced reads)" > gpurun_out/ev/pmc_fetch_size.md
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/ev/pw -o k -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-diagnostics > gpurun_out/ev/pw.log 2>&1
python3 tools/rocpd_summary.py $(find gpurun_out/ev/pw -name "*.db" | head -1) "round 1 (final) PMC pass 2: WRITE_SIZE (KB per dispatch)" > gpurun_out/ev/pmc_write_size.md
python3 tools/pmc_nn_json.py $(find gpurun_out/ev/pf -name "*.db" | head -1) $(find gpurun_out/ev/pw -name "*.db" | head -1) gpurun_out/ev/pmc_nn_latest.json
cp gpurun_out/ev/pmc_nn_latest.json profiles/pmc_nn_latest.json
timeout 600 python3 bench.py > gpurun_out/ev/bench.json 2> gpurun_out/ev/bench.err
head -c 900 gpurun_out/ev/bench.json; echo
head -14 gpurun_out/ev/kernel_stats.md; grep -i "FETCH_SIZE\|WRITE_SIZE" gpurun_out/ev/pmc_fetch_size.md gpurun_out/ev/pmc_write_size.md | grep "nn_pruned\|knn_cov\|linearize"
rm -rf gpurun_out/ev/ks gpurun_out/ev/pf gpurun_out/ev/pw
