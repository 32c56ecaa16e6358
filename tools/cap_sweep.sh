export TMPDIR=/tmp
for f in "-DAPD_KNN_CAP4=48" "-DAPD_KNN_CAP4=56" "-DAPD_KNN_CAP4=60" "-DAPD_KNN_CAP4=60 -DAPD_KNN_WPE=4"; do
  APD_EXTRA_FLAGS="$f" python riv-slam_amd/build.py --force >/dev/null 2>&1
  echo "== flags '$f'"; python tools/knn_time.py 2>&1 | tail -1
  (cd /tmp && rm -rf /tmp/kk && rocprofv3 --kernel-trace --stats -d /tmp/kk -o k -- python3 $GRAFT_REPO_ROOT/tools/knn_time.py >/dev/null 2>&1); python tools/rocpd_summary.py $(ls /tmp/kk/*.db | head -1) knn | grep knn_cov
done
