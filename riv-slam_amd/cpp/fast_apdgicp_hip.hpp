// fast_gicp::FastAPDGICPHip -- drop-in replacement for fast_gicp::FastAPDGICP
// (/root/reference/fast_apdgicp/include/fast_gicp/gicp/fast_apdgicp.hpp:19-110) that runs the whole
// registration on an MI355X through the C ABI of libapdgicp_hip.so (include/apdgicp_hip.h).
//
// It derives from pcl::Registration<PointSource, PointTarget, float> exactly like the reference, so
// radar_graph_slam's select_registration_method() (registrations.cpp:38-50) can return it through the
// same pcl::Registration<PointXYZI,PointXYZI>::Ptr, and the callers' setInputTarget / setInputSource /
// align / hasConverged / getFinalTransformation / getFitnessScore / getSearchMethodTarget keep working
// (the last two are served by the PCL base class on its own tree_).
//
// Header-only; needs only <pcl/registration/registration.h> (and whatever it pulls in) plus
// apdgicp_hip.h.  Error convention of the reference: no exceptions, no return codes -- a failed call
// leaves hasConverged() == false and prints one line on stderr (lsq_registration_impl.hpp:72).
#ifndef FAST_GICP_FAST_APDGICP_HIP_HPP
#define FAST_GICP_FAST_APDGICP_HIP_HPP

#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#include <pcl/registration/registration.h>
#include <pcl/search/kdtree.h>

#include <cstdint>
#include <cstdio>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <utility>
#include <vector>

#include "apdgicp_hip.h"

namespace fast_gicp {

#ifndef FAST_GICP_GICP_SETTINGS_HPP  // same enum as gicp/gicp_settings.hpp:6 when that header is absent
#define FAST_GICP_GICP_SETTINGS_HPP
enum class RegularizationMethod { NONE, MIN_EIG, NORMALIZED_MIN_EIG, PLANE, FROBENIUS };
#endif

template <typename PointSource, typename PointTarget>
class FastAPDGICPHip : public pcl::Registration<PointSource, PointTarget, float> {
 public:
  using Scalar = float;
  using Base = pcl::Registration<PointSource, PointTarget, Scalar>;
  using Matrix4 = typename Base::Matrix4;
  using PointCloudSource = typename Base::PointCloudSource;
  using PointCloudSourceConstPtr = typename PointCloudSource::ConstPtr;
  using PointCloudTarget = typename Base::PointCloudTarget;
  using PointCloudTargetConstPtr = typename PointCloudTarget::ConstPtr;
  using CovVector = std::vector<Eigen::Matrix4d, Eigen::aligned_allocator<Eigen::Matrix4d>>;
#if PCL_VERSION >= PCL_VERSION_CALC(1, 10, 0)
  using Ptr = pcl::shared_ptr<FastAPDGICPHip<PointSource, PointTarget>>;
  using ConstPtr = pcl::shared_ptr<const FastAPDGICPHip<PointSource, PointTarget>>;
#else
  using Ptr = boost::shared_ptr<FastAPDGICPHip<PointSource, PointTarget>>;
  using ConstPtr = boost::shared_ptr<const FastAPDGICPHip<PointSource, PointTarget>>;
#endif

 protected:
  using Base::converged_;
  using Base::corr_dist_threshold_;
  using Base::force_no_recompute_;
  using Base::target_cloud_updated_;
  using Base::tree_;
  using Base::final_transformation_;
  using Base::input_;
  using Base::max_iterations_;
  using Base::nr_iterations_;
  using Base::reg_name_;
  using Base::target_;
  using Base::transformation_epsilon_;

 public:
  /// The search object the class hands to its PCL base (setSearchMethodTarget(tree, force_no_recompute = true)), so that the calls the
  /// nodelets make on the BASE class -- getFitnessScore() (loop_detector.cpp:229, scan_matching_odometry_nodelet.cpp:698) and
  /// getSearchMethodTarget()->nearestKSearch(aligned->at(i), 1, ...) (scan_matching_odometry_nodelet.cpp:701-707) -- work
  /// unmodified WITHOUT pcl::Registration::initCompute() ever building its FLANN kd-tree on the CPU (~1 ms per new 8k target,
  /// tens of ms for a submap, in front of a 0.1 ms align): setInputCloud keeps the pointer and builds nothing; nearestKSearch(pt,
  /// 1, ...) is answered from ONE batched device search of all source points at final_transformation_
  /// (apdgicp_nearest_neighbours), run when the first query after an align arrives.  Both callers ask for the transformed source
  /// points in order, so query number i is matched against point i (checked by position, 1e-4 relative: PCL's own
  /// transformPointCloud may round a coordinate differently); a query that is no transformed source point, or k > 1, takes an
  /// exact brute-force scan of the target on the host (the same fp32 distance; a device-resident target is fetched once).
  class DeviceSearch : public pcl::search::KdTree<PointTarget> {
   public:
    using SearchBase = pcl::search::KdTree<PointTarget>;
    using CloudConstPtr = typename SearchBase::PointCloudConstPtr;
    using IndicesConstPtr = typename SearchBase::IndicesConstPtr;
    explicit DeviceSearch(FastAPDGICPHip* owner) : owner_(owner) {}
#if defined(APDGICP_PCL_SHIM) || PCL_VERSION < PCL_VERSION_CALC(1, 12, 0)
    void setInputCloud(const CloudConstPtr& cloud, const IndicesConstPtr& indices = IndicesConstPtr()) override {
      this->input_ = cloud, this->indices_ = indices;  // no kd-tree: the device holds the target
    }
    int nearestKSearch(const PointTarget& p, int k, std::vector<int>& idx, std::vector<float>& d2) const override { return owner_->device_nn(p, k, idx, d2); }
    /// the batch form (pcl/search/search.h): arbitrary queries, ONE device pass for k = 1 (apdgicp_nearest_neighbours_of)
    void nearestKSearch(const typename SearchBase::PointCloud& cloud, const std::vector<int>& indices, int k, std::vector<std::vector<int>>& k_indices,
                        std::vector<std::vector<float>>& k_sqr_distances) const override {
      owner_->device_nn_batch(cloud, indices, k, k_indices, k_sqr_distances);
    }
#else  // PCL >= 1.12: setInputCloud returns bool, indices are pcl::Indices
    bool setInputCloud(const CloudConstPtr& cloud, const IndicesConstPtr& indices = IndicesConstPtr()) override {
      this->input_ = cloud, this->indices_ = indices;
      return true;
    }
    int nearestKSearch(const PointTarget& p, int k, pcl::Indices& idx, std::vector<float>& d2) const override { return owner_->device_nn(p, k, idx, d2); }
    void nearestKSearch(const typename SearchBase::PointCloud& cloud, const pcl::Indices& indices, int k, std::vector<pcl::Indices>& k_indices,
                        std::vector<std::vector<float>>& k_sqr_distances) const override {
      owner_->device_nn_batch(cloud, indices, k, k_indices, k_sqr_distances);
    }
#endif
   private:
    FastAPDGICPHip* owner_;
  };
  struct DeviceSearchStats {
    long batched_passes = 0;   // device searches over all source points
    long served = 0;           // queries answered from one of them
    long fallbacks = 0;        // queries answered by the host brute-force scan
    long device_queries = 0;   // queries answered by a device pass of their own (batch calls; single misses against a large target)
  };

  explicit FastAPDGICPHip(int device = 0) {
    reg_name_ = "FastAPDGICPHip";
    apdgicp_default_params(&params_);
    // the values the reference constructors put into the pcl::Registration members
    // (fast_apdgicp_impl.hpp:23, lsq_registration_impl.hpp:13-15)
    corr_dist_threshold_ = params_.max_correspondence_distance;
    max_iterations_ = params_.max_iterations;
    transformation_epsilon_ = params_.transformation_epsilon;
    if (apdgicp_create(&params_, device, nullptr, &handle_) != 0) report("apdgicp_create");
    setUseDeviceSearch(true);
  }
  ~FastAPDGICPHip() override {
    if (handle_) apdgicp_destroy(handle_);
  }
  FastAPDGICPHip(const FastAPDGICPHip&) = delete;
  FastAPDGICPHip& operator=(const FastAPDGICPHip&) = delete;

  bool ok() const { return handle_ != nullptr; }  // false when no GPU / library error at construction

  // ---- setters of the reference (fast_apdgicp_impl.hpp:34-65, lsq_registration_impl.hpp:30-42)
  void setNumThreads(int) {}  // OpenMP team size: meaningless on the GPU, kept for source compatibility
  void setCorrespondenceRandomness(int k) { params_.k_correspondences = k; }
  void setRegularizationMethod(RegularizationMethod m) { params_.regularization = static_cast<int>(m); }
  void setAzimuthVar(double v) { params_.azimuth_variance_deg = v; }
  void setElevationVar(double v) { params_.elevation_variance_deg = v; }
  void setDistVar(double v) { params_.distance_variance = v; }
  void setRotationEpsilon(double eps) { params_.rotation_epsilon = eps; }
  void setInitialLambdaFactor(double f) { params_.lm_init_lambda_factor = f; }
  /// lsq_registration.hpp:37 -- the reference prints one row per LM trial inside step_lm (lsq_registration_impl.hpp:148-154); here the
  /// device loop records them (apdgicp_set_trace) and align() prints the same table when it returns
  void setDebugPrint(bool on) {
    debug_print_ = on;
    if (handle_ && apdgicp_set_trace(handle_, on ? 1 : 0) != 0) report("setDebugPrint");
  }
  /// not in the reference (its optimizer enum is protected without a setter, lsq_registration.hpp:78)
  void setOptimizer(apdgicp_optimizer o) { params_.optimizer = o; }
  /// upstream fast_gicp::FastGICP cost (no APD covariance): the FAST_GICP branch of the factory (registrations.cpp:28-37)
  void setPlainGICP(bool on) { params_.flags = on ? (params_.flags | APDGICP_FLAG_PLAIN_GICP) : (params_.flags & ~APDGICP_FLAG_PLAIN_GICP); }

  /// opt-in, NOT the reference's arithmetic (include/apdgicp_hip.h, APDGICP_FLAG_ALGEBRAIC_APD): the sensor model of fast_apdgicp_impl.hpp:167-184
  /// from ratios of the point's coordinates instead of three fp32 atan2 and three fp64 sin / cos; poses move by ~1e-7 m
  void setAlgebraicAPD(bool on) { params_.flags = on ? (params_.flags | APDGICP_FLAG_ALGEBRAIC_APD) : (params_.flags & ~APDGICP_FLAG_ALGEBRAIC_APD); }

  /// fp32 summation order of `trans_f * p.getVector4fMap()` (fast_apdgicp_impl.hpp:149), which belongs to the Eigen the reference
  /// is built against, not to the reference: EIGEN_PAIRWISE (default; Eigen >= 3.3) or EIGEN_LINEAR_CHAIN (Eigen 3.2).
  /// tools/eigen_order_probe.cpp prints which one an installed Eigen produces.
  enum TransformOrder { EIGEN_PAIRWISE = 0, EIGEN_LINEAR_CHAIN = 1 };
  void setTransformOrder(TransformOrder o) {
    params_.flags = o == EIGEN_LINEAR_CHAIN ? (params_.flags | APDGICP_FLAG_XF_LINEAR_CHAIN) : (params_.flags & ~APDGICP_FLAG_XF_LINEAR_CHAIN);
  }

  // ---- cache management (fast_apdgicp_impl.hpp:68-108)
  virtual void swapSourceAndTarget() {
    input_.swap(target_);
    std::swap(source_on_device_, target_on_device_);
    forget_search_state();
    if (handle_ && apdgicp_swap_source_and_target(handle_) != 0) report("swapSourceAndTarget");
  }
  virtual void clearSource() {
    input_.reset();
    source_on_device_ = false;
    forget_search_state();
    if (handle_) apdgicp_clear_source(handle_);
  }
  virtual void clearTarget() {
    target_.reset();
    target_on_device_ = false;
    forget_search_state();
    if (handle_) apdgicp_clear_target(handle_);
  }
  // An empty (or null) cloud: the reference builds a kd-tree over nothing and clears the covariances (:95-97); the device side of
  // this class then holds NO cloud -- the next align fails (hasConverged() == false, one line on stderr) instead of silently
  // registering the cloud that was set before.
  // PCL's own setInputSource / setInputTarget REFUSE an empty cloud (an error message, the old pointer stays in input_ / target_).  The
  // device side is emptied all the same -- and remembered as empty, so that handing the OLD cloud over again, which the pointer-equality
  // shortcut below would swallow, puts it back on the device.
  void setInputSource(const PointCloudSourceConstPtr& cloud) override {
    if (input_ == cloud && source_on_device_) return;  // pointer equality keeps the cached covariances, :91-93
    if (cloud && !cloud->empty()) Base::setInputSource(cloud);
    else if (cloud) Base::setInputSource(cloud);  // (PCL prints its "Invalid or empty point cloud dataset given" and keeps input_)
    nn_epoch_++;
    source_on_device_ = false;
    if (!handle_) return;
    if (!cloud || cloud->empty()) {
      apdgicp_clear_source(handle_);
      return;
    }
    if (apdgicp_set_source(handle_, &cloud->at(0).x, (int64_t)cloud->size(), (int64_t)sizeof(PointSource), 0, token_of(cloud.get())) != 0)
      report("setInputSource");
    else source_on_device_ = true;
  }
  void setInputTarget(const PointCloudTargetConstPtr& cloud) override {
    if (target_ == cloud && target_on_device_) return;  // :102-104
    if (cloud) Base::setInputTarget(cloud);   // (PCL itself refuses an empty target with an error message and keeps the old pointer)
    nn_epoch_++;
    device_target_n_ = 0;
    host_target_.clear();
    target_on_device_ = false;
    if (!handle_) return;
    if (!cloud || cloud->empty()) {
      apdgicp_clear_target(handle_);
      return;
    }
    if (apdgicp_set_target(handle_, &cloud->at(0).x, (int64_t)cloud->size(), (int64_t)sizeof(PointTarget), 0, token_of(cloud.get())) != 0)
      report("setInputTarget");
    else target_on_device_ = true;
  }
  /// on (the default): the base class's search object is this class's DeviceSearch -- no CPU kd-tree is ever built, the base-class
  /// getFitnessScore() / getSearchMethodTarget() answer from the device.  off: PCL's own pcl::search::KdTree again (a FLANN build
  /// per new target inside align(), or none with setSkipBaseSearchTree(true)).
  void setUseDeviceSearch(bool on) {
    typename Base::KdTreePtr tree;
    if (on) tree.reset(new DeviceSearch(this));
    else tree.reset(new pcl::search::KdTree<PointTarget>());
    Base::setSearchMethodTarget(tree, on);
    if (!on) force_no_recompute_ = false, target_cloud_updated_ = true;
    if (on && target_) tree_->setInputCloud(target_);
    device_search_ = on;
    skip_base_tree_ = false;
  }
  bool usesDeviceSearch() const { return device_search_; }
  const DeviceSearchStats& deviceSearchStats() const { return nn_stats_; }
  /// (Only with setUseDeviceSearch(false).)  What PCL itself does inside align(): pcl::Registration::initCompute() rebuilds the BASE class's own search tree
  /// (`tree_->setInputCloud(target_)`, a FLANN kd-tree, single-threaded on the CPU) for every NEW target -- ~1 ms at 8k points,
  /// tens of ms for a submap -- in front of a GPU align that takes 0.1 ms.  Nothing on this class's path reads that tree; it only
  /// serves the base-class getFitnessScore() / getSearchMethodTarget() (scan_matching_odometry_nodelet.cpp:697-707,
  /// loop_detector.cpp:229).  on = true hands the tree back with force_no_recompute (pcl/registration/registration.h
  /// setSearchMethodTarget), so initCompute() leaves it alone: getFitnessScore() / getSearchMethodTarget() then answer about
  /// whatever cloud the tree was last built from (or none) and MUST be replaced by fitnessScore() / inlierFraction() below or
  /// LoopVerifierHip, which run on the device against the real target.  on = false restores PCL's behaviour.
  void setSkipBaseSearchTree(bool on) {
    if (device_search_) {  // nothing to skip: the device search object never builds a tree
      skip_base_tree_ = on;
      return;
    }
    if (on) {
      Base::setSearchMethodTarget(tree_, true);
    } else {
      force_no_recompute_ = false;   // (the setter can only turn the flag on; it is a protected member)
      target_cloud_updated_ = true;
    }
    skip_base_tree_ = on;
  }
  bool skipsBaseSearchTree() const { return skip_base_tree_; }
  /// pcl::Registration::getFitnessScore(max_range) at the last pose, on the device against the REAL target (also a device target)
  double fitnessScore(double max_range = std::numeric_limits<double>::max()) {
    double s = std::numeric_limits<double>::max();
    if (!handle_ || apdgicp_fitness_score(handle_, result_.T, max_range, &s, nullptr) != 0) report("fitnessScore");
    return s;
  }

  /// scan-to-map mode: the target already lives in device memory (apdgicp_submap_points, 16-byte stride), so the
  /// setInputTarget(keyframe_cloud_s2m) of scan_matching_odometry_nodelet.cpp:615 needs no host cloud.  PCL's align()
  /// insists on a non-null target_, which gets a ONE-POINT PLACEHOLDER far outside any scene (1e18 on every axis).  With the
  /// device search object (default) the base-class getFitnessScore() / getSearchMethodTarget() answer about the REAL device
  /// target; with PCL's own tree (setUseDeviceSearch(false)) they answer about the placeholder -- an absurd 3e36, or DBL_MAX
  /// with a max_range, never a plausible number.
  void setInputTargetDevice(const float* device_xyz, std::size_t n, std::size_t stride_bytes) {
    typename PointCloudTarget::Ptr placeholder(new PointCloudTarget());
    placeholder->resize(1);
    placeholder->at(0).x = placeholder->at(0).y = placeholder->at(0).z = 1e18f;
    Base::setInputTarget(placeholder);
    nn_epoch_++;
    host_target_.clear();
    device_target_n_ = n;
    target_on_device_ = false;
    if (handle_ && n && apdgicp_set_target(handle_, device_xyz, (int64_t)n, (int64_t)stride_bytes, 1, ++device_epoch_) != 0)
      report("setInputTargetDevice");
    else target_on_device_ = handle_ && n;
  }
  virtual void setSourceCovariances(const CovVector& covs) {
    if (handle_ && apdgicp_set_covariances(handle_, APDGICP_SOURCE, covs[0].data(), (int64_t)covs.size()) != 0) report("setSourceCovariances");
  }
  virtual void setTargetCovariances(const CovVector& covs) {
    if (handle_ && apdgicp_set_covariances(handle_, APDGICP_TARGET, covs[0].data(), (int64_t)covs.size()) != 0) report("setTargetCovariances");
  }
  /// by value (the reference returns a const& to host-resident storage, fast_apdgicp.hpp:67-73)
  CovVector getSourceCovariances() { return get_covs(APDGICP_SOURCE, input_ ? input_->size() : 0); }
  CovVector getTargetCovariances() { return get_covs(APDGICP_TARGET, target_ ? target_->size() : 0); }

  // ---- LsqRegistration probes (lsq_registration.hpp:55-57, lsq_registration_impl.hpp:45-52)
  /// the reference returns `const Eigen::Matrix<double, 6, 6>&`; here it is fetched from the device into a member first
  const Eigen::Matrix<double, 6, 6>& getFinalHessian() {
    if (handle_ && apdgicp_get_final_hessian(handle_, final_hessian_.data()) != 0) report("getFinalHessian");
    return final_hessian_;
  }
  double evaluateCost(const Matrix4& relative_pose, Eigen::Matrix<double, 6, 6>* H = nullptr, Eigen::Matrix<double, 6, 1>* b = nullptr) {
    push_params();
    double T[16], cost = 0.0;
    for (int i = 0; i < 16; i++) T[i] = (double)relative_pose.data()[i];
    if (!handle_ || apdgicp_linearize(handle_, T, H ? H->data() : nullptr, b ? b->data() : nullptr, &cost) != 0) report("evaluateCost");
    return cost;
  }
  const apdgicp_result& lastResult() const { return result_; }
  /// clouds up to this many points get their aligned copy from a host loop (0: always the device kernel)
  void setHostTransformMax(std::size_t n) { host_transform_max_ = n; }
  /// ScanMatchingStatus::inlier_fraction (scan_matching_odometry_nodelet.cpp:701-712) at the last pose, on the device
  double inlierFraction(double max_correspondence_dist = 0.5) {
    double f = 0.0;
    if (!handle_ || apdgicp_inlier_fraction(handle_, result_.T, max_correspondence_dist, &f, nullptr) != 0) report("inlierFraction");
    return f;
  }

  /// pcl::search::KdTree::nearestKSearch for DeviceSearch (see there).  Idx: std::vector<int> or pcl::Indices.
  template <typename Idx>
  int device_nn(const PointTarget& p, int k, Idx& idx, std::vector<float>& d2) {
    if (k == 1 && ensure_nn_cache()) {
      const std::size_t n = nn_idx_.size();
      for (int attempt = 0; attempt < 2; attempt++) {  // the expected position, else the start of a new pass over the cloud
        const std::size_t i = attempt == 0 ? nn_cursor_ : 0;
        if (i >= n || (attempt == 1 && nn_cursor_ == 0)) continue;
        const float* q = &nn_xyz_[3 * i];
        auto close = [](float a, float b) { return std::fabs(a - b) <= 1e-4f * (1.0f + std::fabs(b)); };
        if (close(p.x, q[0]) && close(p.y, q[1]) && close(p.z, q[2])) {
          idx.assign(1, nn_idx_[i]), d2.assign(1, nn_d2_[i]);
          nn_cursor_ = i + 1;
          nn_stats_.served++;
          return nn_idx_[i] >= 0 ? 1 : 0;
        }
      }
    }
    // Not a transformed source point in order (or k > 1).  Said once: this is the slow side of the search object.
    if (!warned_fallback_) {
      warned_fallback_ = true;
      std::fprintf(stderr, "[FastAPDGICPHip] nearestKSearch: a query that is not the next transformed source point (or k > 1) -- answered one at a time "
                           "(%s); hand a batch of such queries to getSearchMethodTarget()->nearestKSearch(cloud, indices, 1, ...) for ONE device pass\n",
                   target_size() > kDeviceSingleQueryMin ? "a device pass per query" : "an exact host scan of the target per query");
    }
    if (k == 1 && handle_ && target_on_device_ && target_size() > kDeviceSingleQueryMin) {  // a submap: 500 k distances on the host per query
      int32_t j = -1;
      float d = 0.f;
      const float q[3] = {p.x, p.y, p.z};
      if (apdgicp_nearest_neighbours_of(handle_, q, 1, 12, &j, &d) == 0) {
        nn_stats_.device_queries++;
        nn_cache_epoch_ = 0;  // (the handle's pair was set up for the query: the cached pass is still right, the NEXT one re-arms the pair)
        idx.assign(1, j), d2.assign(1, d);
        return j >= 0 ? 1 : 0;
      }
      report("nearest neighbour of a query");
    }
    return host_nn(p, k, idx, d2);
  }
  /// pcl::search::Search::nearestKSearch(cloud, indices, k, ...): every listed point of `cloud` (all of them when `indices` is empty)
  template <typename Cloud, typename IdxIn, typename IdxOut>
  void device_nn_batch(const Cloud& cloud, const IdxIn& indices, int k, std::vector<IdxOut>& k_indices, std::vector<std::vector<float>>& k_sqr_distances) {
    const std::size_t n = indices.empty() ? cloud.size() : indices.size();
    k_indices.assign(n, IdxOut()), k_sqr_distances.assign(n, std::vector<float>());
    if (!n) return;
    if (k == 1 && handle_ && target_on_device_) {
      std::vector<float> q(3 * n), d(n);
      std::vector<int32_t> j(n);
      for (std::size_t i = 0; i < n; i++) {
        const auto& pt = cloud.points[indices.empty() ? i : (std::size_t)indices[i]];
        q[3 * i] = pt.x, q[3 * i + 1] = pt.y, q[3 * i + 2] = pt.z;
      }
      if (apdgicp_nearest_neighbours_of(handle_, q.data(), (int64_t)n, 12, j.data(), d.data()) == 0) {
        nn_stats_.device_queries += (long)n;
        for (std::size_t i = 0; i < n; i++) k_indices[i].assign(1, j[i]), k_sqr_distances[i].assign(1, d[i]);
        return;
      }
      report("nearest neighbours of a batch of queries");
    }
    for (std::size_t i = 0; i < n; i++) host_nn(cloud.points[indices.empty() ? i : (std::size_t)indices[i]], k, k_indices[i], k_sqr_distances[i]);
  }

 protected:
  static constexpr std::size_t kDeviceSingleQueryMin = 32768;  // targets above this size: a single stray query goes to the device (~60 us) instead of a host scan
  std::size_t target_size() const { return device_target_n_ ? device_target_n_ : (target_ ? target_->size() : 0); }
  void forget_search_state() {  // swap / clear: the cached pass, a device target's host copy and its size belong to the old clouds
    nn_epoch_++;
    device_target_n_ = 0;
    host_target_.clear();
  }
  // one batched device search of all source points at final_transformation_, kept until the pose or a cloud changes
  bool ensure_nn_cache() {
    if (!handle_ || !input_ || input_->empty()) return false;
    const float* T = final_transformation_.data();
    if (nn_cache_epoch_ == nn_epoch_ && std::memcmp(nn_T_, T, sizeof(nn_T_)) == 0 && nn_idx_.size() == input_->size()) return true;
    const std::size_t n = input_->size();
    nn_idx_.assign(n, -1), nn_d2_.assign(n, 0.f), nn_xyz_.resize(3 * n);
    push_params();
    if (apdgicp_nearest_neighbours(handle_, T, nn_idx_.data(), nn_d2_.data(), (int64_t)n) != 0) {
      report("nearest neighbours");
      nn_idx_.clear();
      return false;
    }
    for (std::size_t i = 0; i < n; i++) {  // where query i is expected to be (pcl::transformPointCloud's arithmetic, up to rounding)
      const PointSource& a = input_->points[i];
      nn_xyz_[3 * i] = T[0] * a.x + T[4] * a.y + T[8] * a.z + T[12];
      nn_xyz_[3 * i + 1] = T[1] * a.x + T[5] * a.y + T[9] * a.z + T[13];
      nn_xyz_[3 * i + 2] = T[2] * a.x + T[6] * a.y + T[10] * a.z + T[14];
    }
    std::memcpy(nn_T_, T, sizeof(nn_T_));
    nn_cache_epoch_ = nn_epoch_;
    nn_cursor_ = 0;
    nn_stats_.batched_passes++;
    return true;
  }
  // exact k-NN by a scan of the target on the host (FLANN's L2_Simple order in fp32; ties: the lower index first)
  template <typename Idx>
  int host_nn(const PointTarget& p, int k, Idx& idx, std::vector<float>& d2) {
    // (on failure the outputs keep one slot of {-1, FLT_MAX}: the callers this object serves -- scan_matching_odometry_nodelet.cpp:701-703,
    // PCL's getFitnessScore -- read k_sq_dists[0] without looking at the return value)
    const std::size_t slots = (std::size_t)std::max(k, 1);
    idx.assign(slots, -1), d2.assign(slots, std::numeric_limits<float>::max());
    nn_stats_.fallbacks++;
    std::size_t m = 0;
    const float* xyz = nullptr;
    std::size_t stride = 3;
    if (device_target_n_) {  // the target never was on the host: fetched once
      if (host_target_.size() != 3 * device_target_n_) {
        host_target_.resize(3 * device_target_n_);
        if (!handle_ || apdgicp_get_points(handle_, APDGICP_TARGET, host_target_.data(), (int64_t)device_target_n_) != 0) {
          report("get target points");
          host_target_.clear();
          return 0;  // (idx / d2: {-1, FLT_MAX})
        }
      }
      xyz = host_target_.data(), m = device_target_n_;
    } else if (target_ && !target_->empty()) {
      xyz = &target_->points[0].x, m = target_->size(), stride = sizeof(PointTarget) / sizeof(float);
    }
    if (!m || k < 1) return 0;
    idx.clear(), d2.clear();
    std::vector<std::pair<float, int>> best;  // (distance, index), ascending
    for (std::size_t j = 0; j < m; j++) {
      const float* q = xyz + j * stride;
      const float dx = p.x - q[0], dy = p.y - q[1], dz = p.z - q[2];
      float d = dx * dx;
      d = d + dy * dy;
      d = d + dz * dz;
      if ((int)best.size() < k || d < best.back().first) {
        best.insert(std::upper_bound(best.begin(), best.end(), std::make_pair(d, (int)j)), std::make_pair(d, (int)j));
        if ((int)best.size() > k) best.pop_back();
      }
    }
    for (const auto& b : best) idx.push_back(b.second), d2.push_back(b.first);
    return (int)best.size();
  }

  // pcl::Registration::align -> this (fast_apdgicp_impl.hpp:121-130 + lsq_registration_impl.hpp:55-80)
  void computeTransformation(PointCloudSource& output, const Matrix4& guess) override {
    converged_ = false;
    final_transformation_ = guess;
    nn_epoch_++;
    if (!handle_ || !input_ || !target_) {
      report("align (no GPU handle or no input)");
      return;
    }
    push_params();
    if (apdgicp_align(handle_, guess.data(), &result_) != 0) {
      report("align");
      return;
    }
    for (int i = 0; i < 16; i++) final_transformation_.data()[i] = result_.T[i];  // both column-major
    converged_ = result_.converged != 0;
    nr_iterations_ = result_.iterations;
    if (debug_print_) print_lm_table();
    if (result_.lm_failed) std::fprintf(stderr, "lm not converged!!\n");  // lsq_registration_impl.hpp:72
    // pcl::transformPointCloud(*input_, output, final_transformation_), :79.  The reference does this on the host, and for a
    // scan-sized cloud so does this class: the source is already here, and one pass over it (a few microseconds) is cheaper
    // than a kernel, a copy back and a second wait for the device.  Large clouds go through the device.
    if (input_->size() <= host_transform_max_) {  // one pass: copy the point (all its fields) and move its coordinates
      const float* T = result_.T;  // column-major
      const std::size_t n = input_->size();
      output.points.resize(n);
      const PointSource* in = input_->points.data();
      PointSource* out = output.points.data();
      // PCL's xyz point types start with float data[4] = {x, y, z, 1} on a 16-byte boundary: the point is then one 4-float
      // column combination, as in pcl::transformPointCloud's SSE path (8192 points: 18 -> 14 us on the host); any other layout
      // takes the scalar loop.  The fourth float is carried over, whatever it holds.
      const bool xyz_first = sizeof(PointSource) >= 16 && sizeof(PointSource) % 16 == 0 && n > 0 && (const void*)&in[0].x == (const void*)&in[0] &&
                             &in[0].y == &in[0].x + 1 && &in[0].z == &in[0].x + 2 && ((std::uintptr_t)in % 16) == 0 && ((std::uintptr_t)out % 16) == 0;
      if (xyz_first) {
        typedef float v4f __attribute__((vector_size(16)));
        v4f c0, c1, c2, c3;
        std::memcpy(&c0, T, 16), std::memcpy(&c1, T + 4, 16), std::memcpy(&c2, T + 8, 16), std::memcpy(&c3, T + 12, 16);
        for (std::size_t q = 0; q < n; q++) {
          PointSource pt = in[q];
          v4f p;
          std::memcpy(&p, &pt, 16);
          const v4f xs = {p[0], p[0], p[0], p[0]}, ys = {p[1], p[1], p[1], p[1]}, zs = {p[2], p[2], p[2], p[2]};
          v4f r = c0 * xs + c1 * ys + c2 * zs + c3;
          r[3] = p[3];
          std::memcpy(&pt, &r, 16);
          out[q] = pt;
        }
      } else {
        for (std::size_t q = 0; q < n; q++) {
          PointSource pt = in[q];
          const float x = pt.x, y = pt.y, z = pt.z;
          pt.x = T[0] * x + T[4] * y + T[8] * z + T[12];
          pt.y = T[1] * x + T[5] * y + T[9] * z + T[13];
          pt.z = T[2] * x + T[6] * y + T[10] * z + T[14];
          out[q] = pt;
        }
      }
    } else if ((output.points = input_->points, true) && apdgicp_transform_source(handle_, result_.T, &output.points[0].x, (int64_t)output.size(), (int64_t)sizeof(PointSource)) != 0) {
      report("transformPointCloud");
    }
  }

 private:
  static uint64_t token_of(const void* p) { return (uint64_t)(uintptr_t)p; }
  void push_params() {  // the PCL base-class setters write plain members; read them at call time
    params_.max_correspondence_distance = corr_dist_threshold_;
    params_.max_iterations = max_iterations_;
    params_.transformation_epsilon = transformation_epsilon_;
    if (handle_ && apdgicp_set_params(handle_, &params_) != 0) report("set_params");
  }
  CovVector get_covs(int which, std::size_t n) {
    CovVector out(n);
    if (n && handle_) {
      push_params();
      if (apdgicp_get_covariances(handle_, which, out[0].data(), (int64_t)n) != 0) report("getCovariances");
    }
    return out;
  }
  // The table step_lm prints under lm_debug_print_ (lsq_registration_impl.hpp:148-154), from the trace the device loop kept
  // (apdgicp_get_trace, apdgicp_get_trace_step_norms): "--- LM optimization ---" and the header in front of the first trial of every call
  // of step_lm (= every new linearisation: y0 changes), one row per trial -- i, y0, yi, rho, lambda, |delta|, dec ('x' when rho > 0).
  // The reference prints while it iterates; here the rows appear when align() returns.
  void print_lm_table() {
    int64_t nt = 0, np = 0;
    if (apdgicp_get_trace(handle_, 0, nullptr, nullptr, nullptr, nullptr, &nt, 0, nullptr, &np) != 0) return report("setDebugPrint trace");
    if (nt == 0) return;
    std::vector<double> lam(nt), rho(nt), y0(nt), yi(nt), dn(nt, 0.0);
    int64_t nt2 = 0;
    if (apdgicp_get_trace(handle_, nt, lam.data(), rho.data(), y0.data(), yi.data(), &nt, 0, nullptr, &np) != 0 ||
        apdgicp_get_trace_step_norms(handle_, nt, dn.data(), &nt2) != 0)
      return report("setDebugPrint trace");
    nt = std::min<int64_t>(nt, (int64_t)lam.size());
    int64_t i_in_call = 0;
    for (int64_t t = 0; t < nt; t++) {
      // (a trial that follows a REJECTED one belongs to the same call of step_lm; after an accepted one -- rho >= 0 -- the next call starts)
      if (t == 0 || !(rho[t - 1] < 0)) {
        std::printf("--- LM optimization ---\n%5s %15s %15s %15s %15s %15s %5s\n", "i", "y0", "yi", "rho", "lambda", "|delta|", "dec");
        i_in_call = 0;
      }
      std::printf("%5d %15g %15g %15g %15g %15g %5c\n", (int)i_in_call++, y0[t], yi[t], rho[t], lam[t], dn[t], rho[t] > 0.0 ? 'x' : ' ');
    }
    std::fflush(stdout);
  }
  void report(const char* what) const { std::fprintf(stderr, "[FastAPDGICPHip] %s failed: %s\n", what, apdgicp_last_error()); }

  uint64_t device_epoch_ = 0x5375624d61700000ull;  // tokens of device targets (never equal to a host cloud's address)
  apdgicp_handle* handle_ = nullptr;
  apdgicp_params params_;
  apdgicp_result result_{};
  Eigen::Matrix<double, 6, 6> final_hessian_ = Eigen::Matrix<double, 6, 6>::Identity();
  std::size_t host_transform_max_ = 65536;
  bool skip_base_tree_ = false;
  bool device_search_ = false;
  bool debug_print_ = false;
  bool warned_fallback_ = false;
  bool source_on_device_ = false, target_on_device_ = false;  // the device holds the cloud input_ / target_ points at
  // DeviceSearch's cache: the batched search of the source at nn_T_
  unsigned long nn_epoch_ = 1, nn_cache_epoch_ = 0;
  float nn_T_[16] = {};
  std::vector<int> nn_idx_;
  std::vector<float> nn_d2_, nn_xyz_, host_target_;
  std::size_t nn_cursor_ = 0, device_target_n_ = 0;
  DeviceSearchStats nn_stats_;
};

}  // namespace fast_gicp
#endif
