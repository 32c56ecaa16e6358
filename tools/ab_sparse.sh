#!/bin/bash
# the default bench under APDGICP_NN_SPARSE = 0 / 24 / 32 / 40 / 48 (blocks with at most that many searching points take the point-serial search), four alternations on one box
for r in 1 2 3 4; do
for sp in 0 24 32 40 48; do
  echo -n "sparse=$sp: "; APDGICP_NN_SPARSE=$sp timeout 300 python bench.py --no-cpu-baseline --no-diagnostics 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
done
done
for sp in 0 32 48; do
  echo -n "LM loop sparse=$sp: "; APDGICP_NN_SPARSE=$sp timeout 300 python bench.py --kind loop --optimizer lm --no-cpu-baseline --no-diagnostics 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
done
