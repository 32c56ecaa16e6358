for w in 5 6 7 8; do
  APD_EXTRA_FLAGS="-DAPD_KNN_WPE=$w" python riv-slam_amd/build.py --force >/dev/null 2>&1
  echo "== WPE $w"; python tools/knn_time.py 2>&1 | tail -1
  python tools/phase_bench.py 4 32 40 2>&1 | grep -E "^(full|cov)" | tail -2
done
