cp riv-slam_amd/libapdgicp_hip.so riv-slam_amd/_keep.bin
for i in 1 2; do for v in old new; do cp riv-slam_amd/_$v.bin riv-slam_amd/libapdgicp_hip.so; echo -n "$v "; F_LIST=24 NO_POLLED=1 REPS=64 timeout 200 python tools/lm_loop_bench.py | tail -1; done; done
cp riv-slam_amd/_keep.bin riv-slam_amd/libapdgicp_hip.so
