#!/bin/bash
# bench.py --depth 2 (two steps in flight per handle) against the default, resident and host clouds, alternated on one box (inside gpurun);
# first the nearest-neighbour parity tests.  Result: profiles/r06_depth_sweep.txt (no gain; the default stays 1).
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_hip_parity.py -q -m gpu -k "nearest_neighbours" 2>&1 | tail -3
run() { echo -n "$* : "; timeout 200 python bench.py --no-cpu-baseline --no-diagnostics "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('parity', {}).get('max_translation_error_m'))"; }
for i in 1 2; do
run --handles 4 --depth 1
run --handles 4 --depth 2
run --handles 3 --depth 2
run --handles 2 --depth 2
run --host-clouds --handles 4 --depth 1
run --host-clouds --handles 4 --depth 2
run --host-clouds --handles 3 --depth 2
done
