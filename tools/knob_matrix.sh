#!/bin/bash
# The -m gpu suite under every environment switch the shipped library still reads (run inside gpurun; output -> profiles/rNN_knob_matrix.txt).
# Cross-checks: brute-force search / covariance kernels, neighbour keeping off, every search block shape, the point-serial search off and at its maximum, diagnostics on,
# the host-polled LM loop instead of the pair pool, the dense search in index order, the pool with four pair lists and two cloud streams.  (Tuning knobs without a second code path -- APDGICP_POOL_LANES, APDGICP_POOL_TICKS,
# APDGICP_PROFILE_STRIDE -- are exercised by tests/test_lm_pool.py and bench.py.)
for e in "" "APDGICP_NN_MODE=brute" "APDGICP_KNN_MODE=brute" "APDGICP_NN_SKIN=0" "APDGICP_NN_W=1" "APDGICP_NN_W=2" "APDGICP_NN_W=4" "APDGICP_NN_W=8" \
         "APDGICP_NN_SPARSE=0" "APDGICP_NN_SPARSE=64" "APDGICP_STATS=1" "APDGICP_LM_POOL=0" "APDGICP_NN_ORDER=0" "APDGICP_POOL_LISTS=4 APDGICP_POOL_CLOUD_STREAMS=2"; do
  printf "[%s] " "$e"
  env $e python -m pytest tests -q -m gpu -x 2>&1 | tail -1
done
