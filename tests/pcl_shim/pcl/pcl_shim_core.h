// TEST-ONLY stand-in for the slice of PCL that fast_gicp::FastAPDGICPHip touches, so the adapter in
// riv-slam_amd/cpp/ can be compiled and exercised in an image without PCL/Eigen.  It is never
// installed or shipped and is NOT used to build anything from /root/reference.  Member names and
// semantics follow pcl::Registration / pcl::PointCloud of PCL 1.8-1.10 (SURVEY.md 8b caveat: re-verify
// against the PCL that RIV-SLAM is actually built with).
#pragma once
#include <cfloat>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <vector>

#define PCL_VERSION_CALC(MAJ, MIN, PATCH) ((MAJ)*100000 + (MIN)*100 + (PATCH))
#define PCL_VERSION PCL_VERSION_CALC(1, 10, 0)
#define APDGICP_PCL_SHIM 1

namespace Eigen {  // the three Eigen types named by the adapter's signatures
template <typename S, int R, int C>
struct ShimMatrix {
  S m[R * C];  // column-major like Eigen
  S* data() { return m; }
  const S* data() const { return m; }
  S& operator()(int r, int c) { return m[r + R * c]; }
  const S& operator()(int r, int c) const { return m[r + R * c]; }
  void setZero() { std::memset(m, 0, sizeof(m)); }
  void setIdentity() {
    setZero();
    for (int i = 0; i < (R < C ? R : C); i++) m[i + R * i] = S(1);
  }
  static ShimMatrix Identity() {
    ShimMatrix x;
    x.setIdentity();
    return x;
  }
  static ShimMatrix Zero() {
    ShimMatrix x;
    x.setZero();
    return x;
  }
};
template <typename S, int R, int C>
using Matrix = ShimMatrix<S, R, C>;
using Matrix4f = ShimMatrix<float, 4, 4>;
using Matrix4d = ShimMatrix<double, 4, 4>;
template <typename T>
using aligned_allocator = std::allocator<T>;
}  // namespace Eigen

namespace pcl {
template <typename T>
using shared_ptr = std::shared_ptr<T>;

struct alignas(16) PointXYZI {  // 32 bytes: data[4] = {x,y,z,1}, intensity, padding
  float x, y, z, data3 = 1.f;
  float intensity = 0.f;
  float pad_[3] = {0, 0, 0};
};
struct alignas(16) PointXYZ {
  float x, y, z, data3 = 1.f;
};

template <typename PointT>
struct PointCloud {
  using Ptr = std::shared_ptr<PointCloud<PointT>>;
  using ConstPtr = std::shared_ptr<const PointCloud<PointT>>;
  std::vector<PointT> points;
  std::size_t size() const { return points.size(); }
  bool empty() const { return points.empty(); }
  const PointT& at(std::size_t i) const { return points.at(i); }
  PointT& at(std::size_t i) { return points.at(i); }
  void resize(std::size_t n) { points.resize(n); }
};

namespace search {
// pcl::search::Search / pcl::search::KdTree as pcl::Registration uses them (tree_): the two virtuals a registration's search object
// is called through -- setInputCloud(cloud, indices) and nearestKSearch(point, k, indices, sqr_distances) const, declared in
// pcl/search/search.h and overridden by pcl/search/kdtree.h -- and the protected input_ behind getInputCloud().  In real PCL
// KdTree::setInputCloud is where a FLANN kd-tree is built on the CPU, O(n log n), single-threaded -- counted here (n_builds,
// process-wide total in kdtree_builds_total()) so that a test can see WHEN the base class does it.  The search itself is a linear
// scan: the shim is test infrastructure for scan-sized clouds.
inline int& kdtree_builds_total() {
  static int n = 0;
  return n;
}
template <typename PointT>
class Search {
 public:
  using PointCloudConstPtr = typename PointCloud<PointT>::ConstPtr;
  using IndicesConstPtr = std::shared_ptr<const std::vector<int>>;
  virtual ~Search() {}
  virtual void setInputCloud(const PointCloudConstPtr& cloud, const IndicesConstPtr& indices = IndicesConstPtr()) {
    input_ = cloud;
    indices_ = indices;
  }
  virtual PointCloudConstPtr getInputCloud() const { return input_; }
  virtual int nearestKSearch(const PointT& p, int k, std::vector<int>& idx, std::vector<float>& d2) const = 0;
  // pcl/search/search.h + impl/search.hpp: the batch form -- every listed point of `cloud` (all of them when `indices` is empty), by
  // default one single-point search after the other
  using PointCloud = pcl::PointCloud<PointT>;
  virtual void nearestKSearch(const PointCloud& cloud, const std::vector<int>& indices, int k, std::vector<std::vector<int>>& k_indices,
                              std::vector<std::vector<float>>& k_sqr_distances) const {
    const std::size_t n = indices.empty() ? cloud.size() : indices.size();
    k_indices.resize(n), k_sqr_distances.resize(n);
    for (std::size_t i = 0; i < n; i++) nearestKSearch(cloud.points[indices.empty() ? i : (std::size_t)indices[i]], k, k_indices[i], k_sqr_distances[i]);
  }

 protected:
  PointCloudConstPtr input_;
  IndicesConstPtr indices_;
};
template <typename PointT>
class KdTree : public Search<PointT> {
 public:
  using Ptr = std::shared_ptr<KdTree<PointT>>;
  using PointCloudConstPtr = typename Search<PointT>::PointCloudConstPtr;
  using IndicesConstPtr = typename Search<PointT>::IndicesConstPtr;
  using Search<PointT>::nearestKSearch;  // (pcl/search/kdtree.h does the same: the batch form stays visible)
  void setInputCloud(const PointCloudConstPtr& cloud, const IndicesConstPtr& indices = IndicesConstPtr()) override {
    this->input_ = cloud;
    this->indices_ = indices;
    n_builds++;
    kdtree_builds_total()++;
  }
  int nearestKSearch(const PointT& p, int k, std::vector<int>& idx, std::vector<float>& d2) const override {
    if (!this->input_ || this->input_->empty() || k != 1) return 0;
    float best = std::numeric_limits<float>::infinity();
    int bi = -1;
    for (std::size_t i = 0; i < this->input_->size(); i++) {
      const PointT& q = this->input_->at(i);
      const float dx = p.x - q.x, dy = p.y - q.y, dz = p.z - q.z;
      float d = dx * dx;
      d = d + dy * dy;
      d = d + dz * dz;
      if (d < best) best = d, bi = (int)i;
    }
    idx.assign(1, bi), d2.assign(1, best);
    return 1;
  }
  int n_builds = 0;
};
}  // namespace search

template <typename PointSource, typename PointTarget, typename Scalar = float>
class Registration {
 public:
  using KdTree = search::KdTree<PointTarget>;
  using KdTreePtr = typename KdTree::Ptr;
  using Matrix4 = Eigen::ShimMatrix<Scalar, 4, 4>;
  using PointCloudSource = PointCloud<PointSource>;
  using PointCloudSourcePtr = typename PointCloudSource::Ptr;
  using PointCloudSourceConstPtr = typename PointCloudSource::ConstPtr;
  using PointCloudTarget = PointCloud<PointTarget>;
  using PointCloudTargetPtr = typename PointCloudTarget::Ptr;
  using PointCloudTargetConstPtr = typename PointCloudTarget::ConstPtr;
  using Ptr = std::shared_ptr<Registration<PointSource, PointTarget, Scalar>>;

  Registration() : tree_(new KdTree()) { final_transformation_.setIdentity(); }
  virtual ~Registration() {}
  virtual void setInputSource(const PointCloudSourceConstPtr& cloud) {  // registration.hpp: "Invalid or empty point cloud dataset given!" -- the old cloud stays
    if (!cloud || cloud->empty()) return;
    input_ = cloud;
  }
  virtual void setInputTarget(const PointCloudTargetConstPtr& cloud) {  // registration.hpp: keeps the pointer, flags the tree as stale
    if (!cloud || cloud->empty()) return;
    target_ = cloud;
    target_cloud_updated_ = true;
  }
  // registration.h: a tree handed over with force_no_recompute is never rebuilt by initCompute (the flag only ever turns on here)
  void setSearchMethodTarget(const KdTreePtr& tree, bool force_no_recompute = false) {
    tree_ = tree;
    if (force_no_recompute) force_no_recompute_ = true;
    target_cloud_updated_ = true;
  }
  KdTreePtr getSearchMethodTarget() const { return tree_; }
  // registration.hpp getFitnessScore(max_range): mean squared 1-NN distance of the transformed input in tree_ over the points
  // with SQUARED distance <= max_range; answers about whatever cloud tree_ was last built from
  double getFitnessScore(double max_range = std::numeric_limits<double>::max()) {
    double sum = 0.0;
    int nr = 0;
    std::vector<int> idx(1);
    std::vector<float> d2(1);
    const Scalar* T = final_transformation_.data();
    for (std::size_t i = 0; input_ && i < input_->size(); i++) {
      PointTarget p;
      const PointSource& a = input_->at(i);
      p.x = T[0] * a.x + T[4] * a.y + T[8] * a.z + T[12];
      p.y = T[1] * a.x + T[5] * a.y + T[9] * a.z + T[13];
      p.z = T[2] * a.x + T[6] * a.y + T[10] * a.z + T[14];
      if (tree_->nearestKSearch(p, 1, idx, d2) == 1 && d2[0] <= max_range) sum += d2[0], nr++;
    }
    return nr > 0 ? sum / nr : std::numeric_limits<double>::max();
  }
  void setMaximumIterations(int n) { max_iterations_ = n; }
  void setTransformationEpsilon(double e) { transformation_epsilon_ = e; }
  void setMaxCorrespondenceDistance(double d) { corr_dist_threshold_ = d; }
  double getMaxCorrespondenceDistance() const { return corr_dist_threshold_; }
  bool hasConverged() const { return converged_; }
  Matrix4 getFinalTransformation() const { return final_transformation_; }
  void align(PointCloudSource& output) { align(output, Matrix4::Identity()); }
  void align(PointCloudSource& output, const Matrix4& guess) {  // pcl::Registration::align: initCompute, reset, then the virtual
    if (!initCompute()) return;
    converged_ = false;
    nr_iterations_ = 0;
    final_transformation_ = guess;
    if (input_) output.points = input_->points;
    computeTransformation(output, guess);
  }

 protected:
  // registration.hpp initCompute: "Only update target kd-tree if a new target cloud was set" -- a CPU kd-tree build per NEW target
  bool initCompute() {
    if (!target_) return false;
    if (target_cloud_updated_ && !force_no_recompute_) {
      tree_->setInputCloud(target_);
      target_cloud_updated_ = false;
    }
    return (bool)input_;
  }
  virtual void computeTransformation(PointCloudSource& output, const Matrix4& guess) = 0;
  KdTreePtr tree_;
  bool target_cloud_updated_ = true;
  bool force_no_recompute_ = false;
  std::string reg_name_;
  PointCloudSourceConstPtr input_;
  PointCloudTargetConstPtr target_;
  int nr_iterations_ = 0;
  int max_iterations_ = 10;
  Matrix4 final_transformation_;
  double transformation_epsilon_ = 0.0;
  double corr_dist_threshold_ = DBL_MAX;
  bool converged_ = false;
};
}  // namespace pcl
