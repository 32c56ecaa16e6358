#!/bin/bash
# C5 (100k x 500k, GN-20, one pair) under rocprofv3: kernel trace + stats, per-tick search time, the insts / busy / fetch PMC passes and
# the pruning counters (APDGICP_STATS).  Output: gpurun_out/c5_<tag>_*.  usage (inside gpurun): bash tools/c5_profile.sh <tag>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
t=${1:-c5}; o=gpurun_out/c5_$t; rm -rf ${o}_*; mkdir -p gpurun_out
timeout 300 rocprofv3 --kernel-trace --stats -d ${o}_ks -o k -- python3 tools/c5_run.py > ${o}_run.log 2>&1
db=$(find ${o}_ks -name "*.db" | head -1)
python3 tools/rocpd_summary.py $db "$t: 'python3 tools/c5_run.py' under rocprofv3 --kernel-trace --stats" > ${o}_kernel_stats.md
python3 tools/c5_ticks.py $db > ${o}_ticks.txt
pass() { name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d ${o}_p_$name -o k -- python3 tools/c5_run.py > ${o}_p_$name.log 2>&1
  d=$(find ${o}_p_$name -name "*.db" | head -1)
  [ -n "$d" ] && python3 tools/rocpd_summary.py $d "$t PMC pass '$name': $* ('python3 tools/c5_run.py')" > ${o}_pmc_$name.md
  [ -n "$d" ] && [ "$name" == "insts" ] && python3 tools/c5_ticks.py $d pmc > ${o}_ticks_pmc.txt
  [ -n "$d" ] && cp $d ${o}_$name.db
  rm -rf ${o}_p_$name; }
pass insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass busy SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT
pass fetch FETCH_SIZE
pass write WRITE_SIZE
python3 tools/pmc_c5_json.py ${o}_pmc_c5.json ${o}_insts.db ${o}_busy.db ${o}_fetch.db ${o}_write.db > /dev/null && cp ${o}_pmc_c5.json profiles/pmc_c5.json
rm -f ${o}_insts.db ${o}_busy.db ${o}_fetch.db ${o}_write.db
APDGICP_STATS=1 timeout 200 python3 tools/c5_stats.py > ${o}_prune_stats.txt 2>&1
for e in 1 0 1 0; do echo -n "APDGICP_NN_ORDER=$e  "; APDGICP_NN_ORDER=$e timeout 120 python3 tools/c5_run.py 2>/dev/null | tail -1; done > ${o}_order_ab.txt
rm -rf ${o}_ks
cat ${o}_order_ab.txt; cat ${o}_run.log | tail -3; head -12 ${o}_kernel_stats.md; cat ${o}_ticks.txt; cat ${o}_prune_stats.txt
