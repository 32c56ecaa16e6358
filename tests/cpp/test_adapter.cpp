// Drives fast_gicp::FastAPDGICPHip the way RIV-SLAM does: created like the FAST_APDGICP branch of
// select_registration_method() (registrations.cpp:38-50), used through the pcl::Registration base
// pointer like ScanMatchingOdometryNodelet::matching() (scan_matching_odometry_nodelet.cpp:437-482).
// usage: test_adapter <pair.bin> [compile-only check when no file is given]
//   pair.bin: int32 n_src, int32 n_tgt, float guess[16] (column-major), src xyz[n_src*3], tgt xyz[n_tgt*3]
// prints: converged iterations T[16] (column-major) and the first transformed output point
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "fast_apdgicp_hip.hpp"

using PointT = pcl::PointXYZI;

pcl::Registration<PointT, PointT>::Ptr select_registration_method_hip() {
  fast_gicp::FastAPDGICPHip<PointT, PointT>::Ptr apdgicp(new fast_gicp::FastAPDGICPHip<PointT, PointT>());
  apdgicp->setNumThreads(0);
  apdgicp->setTransformationEpsilon(0.1);        // launch:96
  apdgicp->setMaximumIterations(64);
  apdgicp->setMaxCorrespondenceDistance(2.0);    // launch:98
  apdgicp->setCorrespondenceRandomness(20);
  apdgicp->setDistVar(0.86);
  apdgicp->setAzimuthVar(1.0);                   // launch:35
  apdgicp->setElevationVar(1.0);
  return apdgicp;
}

static pcl::PointCloud<PointT>::Ptr make_cloud(const float* xyz, int n) {
  pcl::PointCloud<PointT>::Ptr c(new pcl::PointCloud<PointT>());
  c->resize(n);
  for (int i = 0; i < n; i++) {
    c->at(i).x = xyz[3 * i], c->at(i).y = xyz[3 * i + 1], c->at(i).z = xyz[3 * i + 2];
    c->at(i).intensity = 42.f;
  }
  return c;
}

// ---------------------------------------------------------------------------------------------------------------
// Timing of the real caller path: HOST pcl::PointXYZI clouds through the pcl::Registration interface, wall clock around
// the calls the reference makes.  (1) the three protocols of fast_apdgicp/src/align.cpp:52-104 -- single, 100 times,
// 100 times with one cloud's covariances reused -- and (2) scan-to-keyframe odometry as in
// scan_matching_odometry_nodelet.cpp:449-471: the keyframe stays (pointer-equal target: cached), every frame brings a new
// source cloud object, align(*aligned, guess), getFinalTransformation().  Prints one JSON object.
#include <algorithm>
#include <chrono>
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void stats(std::vector<double> v, const char* name, bool last) {
  std::sort(v.begin(), v.end());
  auto q = [&](double f) { return v[(size_t)std::min<double>(v.size() - 1, f * (v.size() - 1) + 0.5)]; };
  std::printf("\"%s\": {\"median\": %.4f, \"p10\": %.4f, \"p90\": %.4f, \"min\": %.4f, \"max\": %.4f, \"n\": %zu}%s", name, q(0.5), q(0.1), q(0.9), v.front(),
              v.back(), v.size(), last ? "" : ", ");
}

static int run_protocols(const float* s, int ns, const float* t, int nt, const float* guess) {
  using Reg = fast_gicp::FastAPDGICPHip<PointT, PointT>;
  auto registration = select_registration_method_hip();
  Reg& reg = *dynamic_cast<Reg*>(registration.get());
  pcl::PointCloud<PointT>::ConstPtr target = make_cloud(t, nt), source = make_cloud(s, ns);
  pcl::PointCloud<PointT>::Ptr aligned(new pcl::PointCloud<PointT>());
  pcl::Registration<PointT, PointT>::Matrix4 g;
  for (int i = 0; i < 16; i++) g.data()[i] = guess[i];
  for (int i = 0; i < 5; i++) {  // warm-up: allocations, code objects
    reg.clearTarget(), reg.clearSource();
    reg.setInputTarget(target), reg.setInputSource(source), reg.align(*aligned, g);
  }
  std::printf("{\"points\": [%d, %d], \"iterations\": %d, ", ns, nt, reg.lastResult().iterations + 1);
  // ---- align.cpp: single
  std::vector<double> single, multi, reuse, odom, odom_src;
  for (int r = 0; r < 30; r++) {
    const double t1 = now_ms();
    reg.clearTarget(), reg.clearSource();
    reg.setInputTarget(target), reg.setInputSource(source), reg.align(*aligned, g);
    single.push_back(now_ms() - t1);
  }
  // ---- align.cpp: 100 times (per-call times of the loop; the loop total is their sum)
  double t0 = now_ms();
  for (int i = 0; i < 100; i++) {
    const double t1 = now_ms();
    reg.clearTarget(), reg.clearSource();
    reg.setInputTarget(target), reg.setInputSource(source), reg.align(*aligned, g);
    multi.push_back(now_ms() - t1);
  }
  const double multi_total = now_ms() - t0;
  // ---- align.cpp: 100 times, reusing the covariances of one cloud
  pcl::PointCloud<PointT>::ConstPtr target_ = target, source_ = source;
  t0 = now_ms();
  for (int i = 0; i < 100; i++) {
    const double t1 = now_ms();
    reg.swapSourceAndTarget();
    reg.clearSource();
    reg.setInputTarget(target_), reg.setInputSource(source_), reg.align(*aligned);   // (as there: no guess)
    target_.swap(source_);
    reuse.push_back(now_ms() - t1);
  }
  const double reuse_total = now_ms() - t0;
  const int reuse_iterations = reg.lastResult().iterations + 1;  // (from the identity, as align.cpp does: more LM iterations than with the guess)
  // ---- odometry: cached keyframe, a NEW source cloud object every frame (two objects with the scan's points, alternating)
  pcl::PointCloud<PointT>::ConstPtr frames[2] = {make_cloud(s, ns), make_cloud(s, ns)};
  reg.clearTarget(), reg.clearSource();
  reg.setInputTarget(target);
  for (int i = 0; i < 220; i++) {
    const double t1 = now_ms();
    reg.setInputTarget(target);             // matching() sets the keyframe on every frame: pointer-equal, cached
    reg.setInputSource(frames[i & 1]);
    const double t2 = now_ms();
    reg.align(*aligned, g);
    const auto T = reg.getFinalTransformation();
    (void)T;
    if (i >= 20) odom.push_back(now_ms() - t1), odom_src.push_back(t2 - t1);
  }
  stats(single, "align_cpp_single_ms", false);
  stats(multi, "align_cpp_100_times_per_call_ms", false);
  std::printf("\"align_cpp_100_times_total_ms\": %.3f, ", multi_total);
  stats(reuse, "align_cpp_100_times_reuse_per_call_ms", false);
  std::printf("\"align_cpp_100_times_reuse_total_ms\": %.3f, \"align_cpp_100_times_reuse_iterations\": %d, ", reuse_total, reuse_iterations);
  stats(odom, "odometry_frame_ms", false);
  stats(odom_src, "odometry_set_source_ms", false);  // of the frame: handing over the new scan (returns before the device has it)
  std::printf("\"converged\": %d, \"inlier_fraction\": %.6f}\n", reg.hasConverged() ? 1 : 0, reg.inlierFraction(0.5));
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// The base-class calls of the nodelets -- getFitnessScore() and getSearchMethodTarget()->nearestKSearch(aligned[i], 1, ...) --
// through the search object the adapter installs (FastAPDGICPHip::DeviceSearch): no CPU kd-tree build ever (the shim counts them,
// pcl::search::kdtree_builds_total()), answers from one batched device search per pose.  Then PCL's own tree
// (setUseDeviceSearch(false)): what pcl::Registration::initCompute() adds in front of the GPU path -- one build per NEW target --
// and the setSkipBaseSearchTree opt-out of round 4.
static int run_base_tree(const float* s, int ns, const float* t, int nt, const float* guess) {
  using Reg = fast_gicp::FastAPDGICPHip<PointT, PointT>;
  auto registration = select_registration_method_hip();
  Reg& reg = *dynamic_cast<Reg*>(registration.get());
  std::vector<float> shifted(t, t + 3 * (size_t)nt);
  for (int i = 0; i < nt; i++) shifted[3 * i + 1] += 0.7f;
  pcl::PointCloud<PointT>::ConstPtr A = make_cloud(t, nt), B = make_cloud(t, nt), D = make_cloud(shifted.data(), nt), E = make_cloud(t, nt);
  pcl::PointCloud<PointT>::ConstPtr source = make_cloud(s, ns);
  pcl::PointCloud<PointT>::Ptr aligned(new pcl::PointCloud<PointT>());
  pcl::Registration<PointT, PointT>::Matrix4 g;
  for (int i = 0; i < 16; i++) g.data()[i] = guess[i];
  auto builds = [&]() { return pcl::search::kdtree_builds_total(); };
  // ---- default: the device search object
  const int uses_device = reg.usesDeviceSearch() ? 1 : 0;
  registration->setInputSource(source);
  registration->setInputTarget(A), registration->align(*aligned, g);
  registration->setInputTarget(A), registration->align(*aligned, g);   // pointer-equal keyframe
  registration->setInputTarget(B), registration->align(*aligned, g);   // a new target object
  const int d_builds = builds();
  const double d_f_pcl = registration->getFitnessScore(4.0), d_f_dev = reg.fitnessScore(4.0);       // loop_detector.cpp:229
  const double d_f_pcl_all = registration->getFitnessScore(), d_f_dev_all = reg.fitnessScore();
  // scan_matching_odometry_nodelet.cpp:697-712, verbatim
  const double max_correspondence_dist = 0.5;
  int num_inliers = 0;
  std::vector<int> k_indices;
  std::vector<float> k_sq_dists;
  for (int i = 0; i < (int)aligned->size(); i++) {
    const auto& pt = aligned->at(i);
    registration->getSearchMethodTarget()->nearestKSearch(pt, 1, k_indices, k_sq_dists);
    if (k_sq_dists[0] < max_correspondence_dist * max_correspondence_dist) num_inliers++;
  }
  const double d_inl_nodelet = static_cast<float>(num_inliers) / aligned->size(), d_inl_dev = reg.inlierFraction(0.5);
  const long d_passes = reg.deviceSearchStats().batched_passes, d_served = reg.deviceSearchStats().served, d_fb0 = reg.deviceSearchStats().fallbacks;
  // a query that is no transformed source point, and k = 5: the exact host scan; checked against a scan written here
  PointT probe;
  probe.x = t[0] + 0.013f, probe.y = t[1] - 0.021f, probe.z = t[2] + 0.004f;
  std::vector<int> pi;
  std::vector<float> pd;
  const int got1 = registration->getSearchMethodTarget()->nearestKSearch(probe, 1, pi, pd);
  int bi = -1;
  float bd = 1e30f;
  for (int j = 0; j < nt; j++) {
    const float dx = probe.x - t[3 * j], dy = probe.y - t[3 * j + 1], dz = probe.z - t[3 * j + 2];
    float d = dx * dx;
    d = d + dy * dy;
    d = d + dz * dz;
    if (d < bd) bd = d, bi = j;
  }
  const int foreign_ok = got1 == 1 && pi[0] == bi && pd[0] == bd;
  const int got5 = registration->getSearchMethodTarget()->nearestKSearch(probe, 5, pi, pd);
  const int k5_ok = got5 == 5 && pi[0] == bi && pd[0] <= pd[1] && pd[1] <= pd[2] && pd[2] <= pd[3] && pd[3] <= pd[4];
  const long d_fb1 = reg.deviceSearchStats().fallbacks;
  // a device-resident target: the base class answers about the REAL target now
  apdgicp_submap* sm = nullptr;
  double d_f_pcl_devtgt = -1.0, d_f_dev_devtgt = -1.0;
  int devtgt_foreign_ok = 0;
  if (apdgicp_submap_create(0, nullptr, &sm) == 0) {
    const void* xyz[1] = {&E->at(0).x};
    const int64_t cnt[1] = {(int64_t)E->size()};
    int64_t m = 0;
    const float* dev = nullptr;
    if (apdgicp_submap_assemble(sm, 1, xyz, cnt, sizeof(PointT), 16, 0, nullptr, nullptr, &m) == 0 && apdgicp_submap_points(sm, &dev, &m) == 0) {
      reg.setInputTargetDevice(dev, (std::size_t)m, 16);
      registration->align(*aligned, g);
      d_f_pcl_devtgt = registration->getFitnessScore(4.0), d_f_dev_devtgt = reg.fitnessScore(4.0);
      const int gq = registration->getSearchMethodTarget()->nearestKSearch(probe, 1, pi, pd);   // (the target is fetched from the device once)
      devtgt_foreign_ok = gq == 1 && pi[0] == bi && pd[0] == bd;
    }
  }
  const int d_builds_end = builds();
  // an empty source cloud: the device side holds no source, the next align fails loudly instead of registering the previous one
  pcl::PointCloud<PointT>::ConstPtr empty(new pcl::PointCloud<PointT>());
  registration->setInputTarget(A);
  registration->setInputSource(empty);
  registration->align(*aligned, g);
  const int empty_converged = registration->hasConverged() ? 1 : 0;
  registration->setInputSource(source);
  registration->align(*aligned, g);
  const int back_converged = registration->hasConverged() ? 1 : 0;

  // ---- PCL's own tree
  reg.setUseDeviceSearch(false);
  const int c0 = builds();
  registration->setInputTarget(B), registration->align(*aligned, g);
  const int b1 = builds() - c0;
  registration->setInputTarget(B), registration->align(*aligned, g);   // pointer-equal keyframe: cached on both sides
  const int b2 = builds() - c0;
  registration->setInputTarget(A), registration->align(*aligned, g);   // a new target object: PCL rebuilds its tree
  const int b3 = builds() - c0;
  const double f_pcl = registration->getFitnessScore(4.0), f_dev = reg.fitnessScore(4.0);
  reg.setSkipBaseSearchTree(true);
  registration->setInputTarget(D), registration->align(*aligned, g);   // new target, shifted by 0.7 m: no build
  const int b4 = builds() - c0;
  const int conv_skip = registration->hasConverged() ? 1 : 0;
  const double f_pcl_stale = registration->getFitnessScore(4.0), f_dev_skip = reg.fitnessScore(4.0);
  const bool tree_is_stale = registration->getSearchMethodTarget()->getInputCloud() == A;
  reg.setSkipBaseSearchTree(false);
  registration->setInputTarget(E), registration->align(*aligned, g);
  const int b5 = builds() - c0;
  const double f_pcl_back = registration->getFitnessScore(4.0), f_dev_back = reg.fitnessScore(4.0);
  double f_pcl_placeholder = -1.0, f_pcl_placeholder_unbounded = -1.0, f_dev_device_target = -1.0;
  if (sm) {
    const float* dev = nullptr;
    int64_t m = 0;
    if (apdgicp_submap_points(sm, &dev, &m) == 0 && m > 0) {
      reg.setInputTargetDevice(dev, (std::size_t)m, 16);
      registration->align(*aligned, g);
      f_pcl_placeholder = registration->getFitnessScore(4.0), f_pcl_placeholder_unbounded = registration->getFitnessScore();
      f_dev_device_target = reg.fitnessScore(4.0);
    }
    apdgicp_submap_destroy(sm);
  }
  std::printf("{\"uses_device_search\": %d, \"device_builds\": %d, \"device_builds_end\": %d, \"d_f_pcl\": %.15g, \"d_f_dev\": %.15g, \"d_f_pcl_all\": %.15g, "
              "\"d_f_dev_all\": %.15g, \"d_inl_nodelet\": %.9g, \"d_inl_dev\": %.9g, \"d_passes\": %ld, \"d_served\": %ld, \"d_fallbacks_before\": %ld, "
              "\"d_fallbacks_after\": %ld, \"foreign_ok\": %d, \"k5_ok\": %d, \"d_f_pcl_devtgt\": %.15g, \"d_f_dev_devtgt\": %.15g, \"devtgt_foreign_ok\": %d, "
              "\"empty_converged\": %d, \"back_converged\": %d, \"n_src\": %d, ",
              uses_device, d_builds, d_builds_end, d_f_pcl, d_f_dev, d_f_pcl_all, d_f_dev_all, d_inl_nodelet, d_inl_dev, d_passes, d_served, d_fb0, d_fb1,
              foreign_ok, k5_ok, d_f_pcl_devtgt, d_f_dev_devtgt, devtgt_foreign_ok, empty_converged, back_converged, ns);
  std::printf("\"builds\": [%d, %d, %d, %d, %d], \"f_pcl\": %.12g, \"f_dev\": %.12g, \"f_pcl_stale\": %.12g, \"f_dev_skip\": %.12g, "
              "\"tree_is_stale\": %d, \"converged_with_skip\": %d, \"f_pcl_back\": %.12g, \"f_dev_back\": %.12g, \"f_pcl_placeholder\": %.6g, "
              "\"f_pcl_placeholder_unbounded\": %.6g, \"f_dev_device_target\": %.12g}\n",
              b1, b2, b3, b4, b5, f_pcl, f_dev, f_pcl_stale, f_dev_skip, tree_is_stale ? 1 : 0, conv_skip, f_pcl_back, f_dev_back, f_pcl_placeholder,
              f_pcl_placeholder_unbounded, f_dev_device_target);
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Round 6 boundary items: setDebugPrint's LM table (lsq_registration_impl.hpp:148-154), the batch form of the search object's
// nearestKSearch (one device pass for arbitrary queries), the one-line warning of the per-query fall-back, an empty target followed
// by the OLD target again (pointer-equal: must reach the device), and the shape of the outputs when a search cannot be answered.
static int run_boundary(const float* s, int ns, const float* t, int nt, const float* guess) {
  using Reg = fast_gicp::FastAPDGICPHip<PointT, PointT>;
  auto registration = select_registration_method_hip();
  Reg& reg = *dynamic_cast<Reg*>(registration.get());
  pcl::PointCloud<PointT>::ConstPtr A = make_cloud(t, nt), source = make_cloud(s, ns);
  pcl::PointCloud<PointT>::Ptr aligned(new pcl::PointCloud<PointT>());
  pcl::Registration<PointT, PointT>::Matrix4 g;
  for (int i = 0; i < 16; i++) g.data()[i] = guess[i];
  registration->setInputSource(source), registration->setInputTarget(A);
  reg.setDebugPrint(true);
  std::printf("TABLE-BEGIN\n");
  registration->align(*aligned, g);
  std::printf("TABLE-END\n");
  reg.setDebugPrint(false);
  const int n_lin = reg.lastResult().n_linearize, n_err = reg.lastResult().n_compute_error, conv0 = registration->hasConverged() ? 1 : 0;
  // ---- batch queries: 700 points that are no transformed source points
  pcl::PointCloud<PointT> queries;
  queries.resize(700);
  for (int i = 0; i < 700; i++) {
    const int j = (i * 37) % nt;
    queries.at(i).x = t[3 * j] + 0.011f * (float)(i % 7 - 3), queries.at(i).y = t[3 * j + 1] - 0.017f * (float)(i % 5 - 2), queries.at(i).z = t[3 * j + 2] + 0.003f * (float)(i % 3);
  }
  auto brute = [&](const PointT& q, int& bi, float& bd) {
    bi = -1, bd = 1e30f;
    for (int j = 0; j < nt; j++) {
      const float dx = q.x - t[3 * j], dy = q.y - t[3 * j + 1], dz = q.z - t[3 * j + 2];
      float d = dx * dx;
      d = d + dy * dy;
      d = d + dz * dz;
      if (d < bd) bd = d, bi = j;
    }
  };
  std::vector<std::vector<int>> bi_out;
  std::vector<std::vector<float>> bd_out;
  const long fb0 = reg.deviceSearchStats().fallbacks;
  registration->getSearchMethodTarget()->nearestKSearch(queries, std::vector<int>(), 1, bi_out, bd_out);
  int batch_ok = bi_out.size() == 700 ? 1 : 0;
  for (int i = 0; i < 700 && batch_ok; i++) {
    int bi;
    float bd;
    brute(queries.at(i), bi, bd);
    batch_ok = bi_out[i].size() == 1 && bd_out[i].size() == 1 && bi_out[i][0] == bi && bd_out[i][0] == bd;
  }
  std::vector<int> sub = {5, 17, 699};
  registration->getSearchMethodTarget()->nearestKSearch(queries, sub, 1, bi_out, bd_out);
  int sub_ok = bi_out.size() == 3 ? 1 : 0;
  for (int i = 0; i < 3 && sub_ok; i++) {
    int bi;
    float bd;
    brute(queries.at(sub[i]), bi, bd);
    sub_ok = bi_out[i][0] == bi && bd_out[i][0] == bd;
  }
  const long batch_device_queries = reg.deviceSearchStats().device_queries, batch_fallbacks = reg.deviceSearchStats().fallbacks - fb0;
  // the handle still registers after serving foreign queries (its own pair is set up again)
  registration->align(*aligned, g);
  const int conv_after_batch = registration->hasConverged() ? 1 : 0, iters_after_batch = reg.lastResult().n_linearize;
  // ---- two single foreign queries: the warning appears ONCE on stderr (the Python side counts the lines)
  std::vector<int> pi;
  std::vector<float> pd;
  registration->getSearchMethodTarget()->nearestKSearch(queries.at(3), 1, pi, pd);
  registration->getSearchMethodTarget()->nearestKSearch(queries.at(4), 1, pi, pd);
  // ---- an empty target, then the OLD target object again
  pcl::PointCloud<PointT>::ConstPtr empty(new pcl::PointCloud<PointT>());
  registration->setInputTarget(empty);
  registration->align(*aligned, g);
  const int conv_empty_target = registration->hasConverged() ? 1 : 0;
  registration->setInputTarget(A);  // pointer-equal to what PCL still holds in target_
  registration->align(*aligned, g);
  const int conv_old_target_again = registration->hasConverged() ? 1 : 0;
  // ---- a search that cannot be answered: no target at all
  reg.clearTarget();
  pi.clear(), pd.clear();
  const int got = registration->getSearchMethodTarget()->nearestKSearch(queries.at(0), 1, pi, pd);
  const int unanswered_shape_ok = got == 0 && pi.size() == 1 && pd.size() == 1 && pi[0] == -1 && pd[0] > 1e38f;
  std::printf("{\"n_linearize\": %d, \"n_compute_error\": %d, \"converged\": %d, \"batch_ok\": %d, \"sub_ok\": %d, \"batch_device_queries\": %ld, "
              "\"batch_fallbacks\": %ld, \"conv_after_batch\": %d, \"n_linearize_after_batch\": %d, \"conv_empty_target\": %d, \"conv_old_target_again\": %d, "
              "\"unanswered_shape_ok\": %d}\n",
              n_lin, n_err, conv0, batch_ok, sub_ok, batch_device_queries, batch_fallbacks, conv_after_batch, iters_after_batch, conv_empty_target,
              conv_old_target_again, unanswered_shape_ok);
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 2) {
    std::printf("compile-only\n");
    return 0;
  }
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) return 2;
  int n[2];
  float guess[16];
  if (std::fread(n, 4, 2, f) != 2 || std::fread(guess, 4, 16, f) != 16) return 2;
  std::vector<float> s(3 * n[0]), t(3 * n[1]);
  if (std::fread(s.data(), 4, s.size(), f) != s.size() || std::fread(t.data(), 4, t.size(), f) != t.size()) return 2;
  std::fclose(f);
  if (argc > 2 && std::string(argv[2]) == "--protocol") return run_protocols(s.data(), n[0], t.data(), n[1], guess);
  if (argc > 2 && std::string(argv[2]) == "--base-tree") return run_base_tree(s.data(), n[0], t.data(), n[1], guess);
  if (argc > 2 && std::string(argv[2]) == "--boundary") return run_boundary(s.data(), n[0], t.data(), n[1], guess);

  auto registration = select_registration_method_hip();
  auto source = make_cloud(s.data(), n[0]);
  auto target = make_cloud(t.data(), n[1]);
  registration->setInputTarget(target);
  registration->setInputSource(source);
  pcl::PointCloud<PointT>::Ptr aligned(new pcl::PointCloud<PointT>());
  pcl::Registration<PointT, PointT>::Matrix4 g;
  for (int i = 0; i < 16; i++) g.data()[i] = guess[i];
  registration->align(*aligned, g);
  // second frame against the same keyframe: the target pointer is unchanged -> cached covariances
  registration->setInputTarget(target);
  registration->align(*aligned, g);
  const auto T = registration->getFinalTransformation();
  std::printf("%d", registration->hasConverged() ? 1 : 0);
  auto* hip = dynamic_cast<fast_gicp::FastAPDGICPHip<PointT, PointT>*>(registration.get());
  std::printf(" %d", hip->lastResult().iterations);
  for (int i = 0; i < 16; i++) std::printf(" %.9g", T.data()[i]);
  std::printf(" %.9g %.9g %.9g %.9g", aligned->at(0).x, aligned->at(0).y, aligned->at(0).z, aligned->at(0).intensity);
  // scan-to-map mode (scan_matching_odometry_nodelet.cpp:606-618): the target is assembled on the device (one keyframe,
  // identity pose, no voxel filter == the target cloud itself) and never comes back; same registration expected
  apdgicp_submap* sm = nullptr;
  int same = 0;
  if (apdgicp_submap_create(0, nullptr, &sm) == 0) {
    const void* xyz[1] = {&target->at(0).x};
    const int64_t cnt[1] = {(int64_t)target->size()};
    int64_t m = 0;
    const float* dev = nullptr;
    if (apdgicp_submap_assemble(sm, 1, xyz, cnt, sizeof(PointT), 16, 0, nullptr, nullptr, &m) == 0 && apdgicp_submap_points(sm, &dev, &m) == 0) {
      hip->setInputTargetDevice(dev, (std::size_t)m, 16);
      registration->align(*aligned, g);
      const auto T2 = registration->getFinalTransformation();
      same = m == (int64_t)target->size();
      for (int i = 0; i < 16; i++) same = same && T2.data()[i] == T.data()[i];
    }
    apdgicp_submap_destroy(sm);
  }
  std::printf(" %d\n", same);
  return 0;
}
