python - <<'PY'
import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_cpp_multi_device as T
print(T.build_bench())
PY
python tools/cpp_vs_python.py 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:(v['cpp_ms_per_step'],v['python_ms_per_step']) for k,v in d.items()})"
export GPU_MAX_HW_QUEUES=8
echo direct lm; BENCH_SHARDED_DIRECT=1 tests/cpp/_build/bench_sharded /tmp/cppbench_lm.bin lm 60 12 /tmp/r.bin 8 1 | tail -1
echo direct lm 7; BENCH_SHARDED_DIRECT=1 tests/cpp/_build/bench_sharded /tmp/cppbench_lm.bin lm 60 12 /tmp/r.bin 7 1 | tail -1
echo aligner lm 7; tests/cpp/_build/bench_sharded /tmp/cppbench_lm.bin lm 60 12 /tmp/r.bin 7 1 | tail -1 | cut -c100-300
echo aligner lm 6; tests/cpp/_build/bench_sharded /tmp/cppbench_lm.bin lm 60 12 /tmp/r.bin 6 1 | tail -1 | cut -c100-300
