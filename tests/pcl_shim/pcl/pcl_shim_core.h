// TEST-ONLY stand-in for the slice of PCL that fast_gicp::FastAPDGICPHip touches, so the adapter in
// riv-slam_amd/cpp/ can be compiled and exercised in an image without PCL/Eigen.  It is never
// installed or shipped and is NOT used to build anything from /root/reference.  Member names and
// semantics follow pcl::Registration / pcl::PointCloud of PCL 1.8-1.10 (SURVEY.md 8b caveat: re-verify
// against the PCL that RIV-SLAM is actually built with).
#pragma once
#include <cfloat>
#include <cstring>
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

template <typename PointSource, typename PointTarget, typename Scalar = float>
class Registration {
 public:
  using Matrix4 = Eigen::ShimMatrix<Scalar, 4, 4>;
  using PointCloudSource = PointCloud<PointSource>;
  using PointCloudSourcePtr = typename PointCloudSource::Ptr;
  using PointCloudSourceConstPtr = typename PointCloudSource::ConstPtr;
  using PointCloudTarget = PointCloud<PointTarget>;
  using PointCloudTargetPtr = typename PointCloudTarget::Ptr;
  using PointCloudTargetConstPtr = typename PointCloudTarget::ConstPtr;
  using Ptr = std::shared_ptr<Registration<PointSource, PointTarget, Scalar>>;

  Registration() { final_transformation_.setIdentity(); }
  virtual ~Registration() {}
  virtual void setInputSource(const PointCloudSourceConstPtr& cloud) { input_ = cloud; }
  virtual void setInputTarget(const PointCloudTargetConstPtr& cloud) { target_ = cloud; }
  void setMaximumIterations(int n) { max_iterations_ = n; }
  void setTransformationEpsilon(double e) { transformation_epsilon_ = e; }
  void setMaxCorrespondenceDistance(double d) { corr_dist_threshold_ = d; }
  double getMaxCorrespondenceDistance() const { return corr_dist_threshold_; }
  bool hasConverged() const { return converged_; }
  Matrix4 getFinalTransformation() const { return final_transformation_; }
  void align(PointCloudSource& output) { align(output, Matrix4::Identity()); }
  void align(PointCloudSource& output, const Matrix4& guess) {  // pcl::Registration::align: reset, then the virtual
    converged_ = false;
    nr_iterations_ = 0;
    final_transformation_ = guess;
    if (input_) output.points = input_->points;
    computeTransformation(output, guess);
  }

 protected:
  virtual void computeTransformation(PointCloudSource& output, const Matrix4& guess) = 0;
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
