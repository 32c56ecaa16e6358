# Pool tuning sweep for tools/lm_loop_bench.py (run inside gpurun).  Round 3 swept more (pair groups per chunk, head-slice size and tick
# rate, the four-wave search threshold, chunks ahead: docs/experiments.md); those switches were measured and removed, what is left:
export NO_POLLED=1 REPS=32
run() { echo "$@"; env "$@" python tools/lm_loop_bench.py 2>&1 | tail -1 | sed 's/.*"n_compute_error_sum": [0-9]*, //'; }
run F_LIST=4,8 APDGICP_POOL_LANES=8
run F_LIST=4,8 APDGICP_POOL_LANES=8 APDGICP_POOL_TICKS=1
run F_LIST=4,8 APDGICP_POOL_LANES=8 APDGICP_POOL_TICKS=3
run F_LIST=4 APDGICP_POOL_LANES=4
