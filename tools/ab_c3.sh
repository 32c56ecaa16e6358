export APDGICP_ALLOW_STALE_LIB=1
cp riv-slam_amd/libapdgicp_hip.so riv-slam_amd/_keep.bin
for i in 1 2 3; do for v in ocml new; do cp riv-slam_amd/_$v.bin riv-slam_amd/libapdgicp_hip.so; echo -n "$v "; python tests/measure/bench_configs.py C3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['C3_1x8_lm_launch']['ms_per_batch'], d['C3_1x8_gn20']['ms_per_batch'])"; done; done
cp riv-slam_amd/_keep.bin riv-slam_amd/libapdgicp_hip.so; rm -f riv-slam_amd/_keep.bin
