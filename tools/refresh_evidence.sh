#!/bin/bash
# Regenerates the judged evidence on the GPU box into gpurun_out/ev/ (about 45 minutes); afterwards, HERE, `bash tools/refresh_evidence.sh --collect rNN`
# copies the summaries into profiles/ (named per round; pmc_nn_latest.json / pmc_lm_loop.json under their fixed names).
#   usage: gpurun --timeout 2400 -- 'bash tools/refresh_evidence.sh r04'
#   kernel_stats.md      rocprofv3 --kernel-trace --stats of the DRIVER's command (python3 bench.py --gpus 1 --steps 20 --warmup 5)
#   bench_profiled.json  the JSON line printed by that same profiled run (its roofline.avg_launch_ms must agree with the table)
#   pmc_*.md             separate --pmc passes (never combined with other trace domains): HBM bytes and the issue-side SQ counters
#   pmc_nn_latest.json   per-launch PMC numbers of the batch's nearest-neighbour launches (bench.py's roofline.traffic / roofline_issue)
#   pmc_lm_loop.json     the same per listed pair slot for the pooled LM ticks (bench.py --kind loop --optimizer lm)
#   fuzz_parity.json     tests/measure/fuzz_parity.py, 300 s of adversarial cases against the oracle; parity_sweep*.json: the seeded sweep
#   bench.json           the un-profiled default run; bench_lm_loop.json: the reference's optimiser on the C4 shard
if [ "$1" == "--collect" ]; then
  tag=$2; ev=gpurun_out/ev
  for f in kernel_stats.md kernel_stats_lm_loop.md pmc_fetch.md pmc_write.md pmc_insts.md pmc_busy.md pmc_occ.md pmc_lm_fetch.md pmc_lm_write.md pmc_lm_insts.md pmc_lm_busy.md \
           bench.json bench_profiled.json bench_lm_loop.json other_configs.json odometry_protocol.json cpp_vs_python.json phase_bench.txt lm_pool_streams.txt bench_2ranks_gloo.json fuzz_parity.json fuzz_batch.json parity_sweep.json parity_sweep_xflin.json bench_host_clouds.json \
           c5_kernel_stats.md c5_ticks.txt c5_ticks_pmc.txt c5_prune_stats.txt c5_pmc_insts.md c5_pmc_busy.md c5_pmc_fetch.md c5_pmc_write.md c5_order_ab.txt algebraic_apd.json host_cost.txt; do
    [ -s $ev/$f ] && head -c 16000 $ev/$f > profiles/${tag}_$f   # (the LM passes have a row per launch size: the first ~60 rows)
  done
  for f in pmc_nn_latest.json pmc_lm_loop.json pmc_c5.json; do [ -s $ev/$f ] && cp $ev/$f profiles/$f; done
  ls -la profiles | grep "${tag}_\|pmc_" ; exit 0
fi
tag=${1:-r04}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
ev=gpurun_out/ev; rm -rf $ev; mkdir -p $ev
DRV="python3 bench.py --gpus 1 --steps 20 --warmup 5"
PMC="python3 bench.py --gpus 1 --steps 20 --warmup 5 --repeats 2 --no-cpu-baseline --no-diagnostics"
LMC="python3 bench.py --kind loop --optimizer lm --steps 20 --warmup 5 --repeats 2 --no-cpu-baseline --no-diagnostics"
# (APDGICP_PROFILE_STRIDE=1: bench.py's own timing brackets EVERY search launch in this run, like the trace does, so the two
# averages cover the same launches; the default run samples one tick in ten)
export APDGICP_PROFILE_STRIDE=1
timeout 600 rocprofv3 --kernel-trace --stats -d $ev/ks -o k -- $DRV --no-cpu-baseline > $ev/bench_profiled.json 2> $ev/ks.err
unset APDGICP_PROFILE_STRIDE
python3 tools/rocpd_summary.py $(find $ev/ks -name "*.db" | head -1) "$tag: '$DRV --no-cpu-baseline' under rocprofv3 --kernel-trace --stats" > $ev/kernel_stats.md
pass() {  # name, command, counters...
  name=$1; cmd=$2; shift; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d $ev/p_$name -o k -- $cmd > $ev/p_$name.log 2>&1
  db=$(find $ev/p_$name -name "*.db" | head -1)
  if [ -n "$db" ]; then python3 tools/rocpd_summary.py $db "$tag PMC pass '$name': $* ('$cmd')" > $ev/pmc_$name.md; else echo "pass $name produced no db"; tail -5 $ev/p_$name.log; fi
}
INSTS="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
BUSY="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT"
pass fetch "$PMC" FETCH_SIZE
pass write "$PMC" WRITE_SIZE
pass insts "$PMC" $INSTS
pass busy "$PMC" $BUSY
pass occ "$PMC" GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64
dbs=$(for n in fetch write insts busy occ; do find $ev/p_$n -name "*.db" | head -1; done)
python3 tools/pmc_nn_json.py $ev/pmc_nn_latest.json 8192 odometry $dbs > /dev/null && cp $ev/pmc_nn_latest.json profiles/pmc_nn_latest.json
# the loop-closure regime (pooled LM ticks from the identity): the same passes
pass lm_fetch "$LMC" FETCH_SIZE
pass lm_write "$LMC" WRITE_SIZE
pass lm_insts "$LMC" $INSTS
pass lm_busy "$LMC" $BUSY
dbs=$(for n in lm_fetch lm_write lm_insts lm_busy; do find $ev/p_$n -name "*.db" | head -1; done)
python3 tools/pmc_lm_json.py $ev/pmc_lm_loop.json 8192 $dbs > /dev/null && cp $ev/pmc_lm_loop.json profiles/pmc_lm_loop.json
# C5 (100k x 500k): kernel table, per-tick times and counters, pruning statistics, block order on / off; profiles/pmc_c5.json (bench.py's c5_dense.valu_busy)
bash tools/c5_profile.sh ev > /dev/null 2>&1
for f in kernel_stats.md ticks.txt ticks_pmc.txt prune_stats.txt pmc_insts.md pmc_busy.md pmc_fetch.md pmc_write.md order_ab.txt; do [ -s gpurun_out/c5_ev_$f ] && cp gpurun_out/c5_ev_$f $ev/c5_$f; done
[ -s gpurun_out/c5_ev_pmc_c5.json ] && cp gpurun_out/c5_ev_pmc_c5.json $ev/pmc_c5.json   # (and profiles/pmc_c5.json, written by c5_profile.sh: bench.py below reads it)
timeout 900 python3 bench.py > $ev/bench.json 2> $ev/bench.err
timeout 300 python3 bench.py --host-clouds --no-cpu-baseline --no-diagnostics > $ev/bench_host_clouds.json 2> /dev/null   # the PCIe-inclusive rate
# the reference's optimiser on the C4 shard (bench.py --kind loop --optimizer lm): the JSON line, its kernel table, its streams
timeout 600 python3 bench.py --kind loop --optimizer lm --no-cpu-baseline > $ev/bench_lm_loop.json 2> $ev/bench_lm.err
timeout 600 rocprofv3 --kernel-trace --stats -d $ev/ks_lm -o k -- python3 bench.py --kind loop --optimizer lm --no-cpu-baseline --repeats 3 > /dev/null 2> $ev/ks_lm.err
python3 tools/rocpd_summary.py $(find $ev/ks_lm -name "*.db" | head -1) "$tag: 'python3 bench.py --kind loop --optimizer lm' under rocprofv3 --kernel-trace --stats" > $ev/kernel_stats_lm_loop.md
# stream occupancy of the pool in steady state: the continuous loop of tools/lm_loop_bench.py (bench.py drains the pool between repetitions)
bash tools/lm_trace.sh ev > /dev/null 2>&1; cp gpurun_out/lt_ev.streams.txt $ev/lm_pool_streams.txt
find $ev/ks_lm -type f ! -name "*.db" -delete
# bench.py's own N > 1 loop with two ranks on the one GPU (gloo): a self-test of that code path, not a scaling point
timeout 600 python3 bench.py --gpus 2 --ranks-share-gpu --dist-backend gloo --no-cpu-baseline --no-diagnostics > $ev/bench_2ranks_gloo.json 2> $ev/bench_2ranks.err
# the other BASELINE configs, the odometry protocol through the C++ adapter, the C++ multi-device path against the Python one
timeout 900 python3 tests/measure/bench_configs.py > $ev/other_configs.json 2> $ev/other_configs.err
timeout 600 python3 tests/measure/odometry_protocol.py > $ev/odometry_protocol.json 2> $ev/odometry.err
timeout 900 python3 tools/cpp_vs_python.py 2> $ev/cpp_vs_python.err | tail -1 > $ev/cpp_vs_python.json
timeout 300 python3 tools/phase_bench.py 4 32 60 > $ev/phase_bench.txt 2>&1
# where the host's time goes, resident and host clouds; the opt-in algebraic sensor model against the default (300 seeded pairs + the bench pairs)
(timeout 200 python3 tools/host_cost.py 200; timeout 200 python3 tools/host_cost.py 200 --host-clouds) 2>/dev/null | grep "steps" > $ev/host_cost.txt
timeout 900 python3 tests/measure/fp32_mode.py 60 algebraic > $ev/algebraic_apd.json 2> $ev/algebraic_apd.err
# parity beyond the test suite: the seeded sweep under both transform orders, the adversarial fuzz (five minutes of cases)
timeout 600 python3 tests/measure/parity_sweep.py 24 > $ev/parity_sweep.json 2> $ev/parity_sweep.err
XF_FLAGS=2 timeout 600 python3 tests/measure/parity_sweep.py 24 > $ev/parity_sweep_xflin.json 2>> $ev/parity_sweep.err
timeout 900 python3 tests/measure/fuzz_parity.py 800 0 4281 > $ev/fuzz_parity.json 2> $ev/fuzz_parity.err   # the committed seed range, whole
timeout 400 python3 tests/measure/fuzz_batch.py 240 0 > $ev/fuzz_batch.json 2> $ev/fuzz_batch.err
for d in $ev/p_fetch $ev/p_write $ev/p_insts $ev/p_busy $ev/p_occ $ev/p_lm_fetch $ev/p_lm_write $ev/p_lm_insts $ev/p_lm_busy $ev/ks_lm; do rm -rf $d; done
find $ev/ks -type f ! -name "*.db" -delete
head -c 1500 $ev/bench.json; echo
head -14 $ev/kernel_stats.md
for n in fetch write insts busy occ; do echo "== $n"; grep "k_nn_\|knn_cov_coop\|k_linearize\|k_regularize\|^| kernel" $ev/pmc_$n.md | head -8; done
