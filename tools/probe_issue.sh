# which issue resource binds k_knn_cov_coop<4>: the same number of extra scalar / vector instructions per (query, group) step
for f in "" "-DAPD_PROBE_SALU=24" "-DAPD_PROBE_VALU=24" "-DAPD_PROBE_SALU=48" "-DAPD_PROBE_VALU=48"; do
  APD_EXTRA_FLAGS="$f" python riv-slam_amd/build.py --force >/dev/null 2>&1
  echo "== flags '$f'"; python tools/knn_time.py 2>&1 | tail -1
done
