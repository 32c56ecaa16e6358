// radar_graph_slam::InformationMatrixCalculator with its nearest-neighbour pass on the device
// (radar_graph_slam/src/radar_graph_slam/information_matrix_calculator.cpp:29-86, include/.../information_matrix_calculator.hpp:40-43):
// the edge information of an odometry or loop-closure factor is derived from a fitness score -- the mean squared 1-NN
// distance of cloud2, moved by the relative pose, to cloud1 (:55-86; PCL getFitnessScore semantics with cloud1 as the
// target) -- for which the reference builds ANOTHER kd-tree on the CPU after every align (radar_graph_slam_nodelet.cpp:419,704,
// loop_detector.cpp:315).  Here that pass is apdgicp_fitness_score on a registration handle; the scalar mapping from the
// score to the 6x6 information matrix (:39-50 and weight(), hpp:40-43) is restated as is.
//
// Header-only; no PCL / Eigen types in the interface.
#ifndef FAST_GICP_INFORMATION_MATRIX_HIP_HPP
#define FAST_GICP_INFORMATION_MATRIX_HIP_HPP

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <limits>

#include "apdgicp_hip.h"

namespace fast_gicp {

struct InformationMatrixParams {  // defaults of the constructor (information_matrix_calculator.cpp:14-25)
  bool use_const_inf_matrix = false;
  double const_stddev_x = 0.5, const_stddev_q = 0.1;
  double var_gain_a = 20.0;
  double min_stddev_x = 0.1, max_stddev_x = 5.0;
  double min_stddev_q = 0.05, max_stddev_q = 0.2;
  double fitness_score_thresh = 0.5;
};

class InformationMatrixCalculatorHip {
 public:
  explicit InformationMatrixCalculatorHip(const InformationMatrixParams& p = InformationMatrixParams(), int device = 0) : prm_(p) {
    if (apdgicp_create(nullptr, device, nullptr, &handle_) != 0) std::fprintf(stderr, "[InformationMatrixCalculatorHip] apdgicp_create failed: %s\n", apdgicp_last_error());
  }
  ~InformationMatrixCalculatorHip() {
    if (handle_) apdgicp_destroy(handle_);
  }
  InformationMatrixCalculatorHip(const InformationMatrixCalculatorHip&) = delete;
  InformationMatrixCalculatorHip& operator=(const InformationMatrixCalculatorHip&) = delete;
  bool ok() const { return handle_ != nullptr; }

  /// calc_fitness_score (:55-86): relpose is a column-major 4x4 (Eigen::Isometry3d::matrix().data()); token1 / token2 identify
  /// the clouds (e.g. the shared_ptr addresses): a keyframe cloud seen again is not uploaded again.  max() when nothing is in range.
  double calc_fitness_score(const float* cloud1_xyz, int64_t n1, const float* cloud2_xyz, int64_t n2, int64_t stride_bytes, const double relpose[16],
                            double max_range = std::numeric_limits<double>::max(), uint64_t token1 = 0, uint64_t token2 = 0) {
    double score = std::numeric_limits<double>::max();
    if (!handle_) return score;
    float T[16];
    for (int q = 0; q < 16; q++) T[q] = (float)relpose[q];  // relpose.cast<float>(), :63
    if (apdgicp_set_target(handle_, cloud1_xyz, n1, stride_bytes, 0, token1) != 0 || apdgicp_set_source(handle_, cloud2_xyz, n2, stride_bytes, 0, token2) != 0 ||
        apdgicp_fitness_score(handle_, T, max_range, &score, nullptr) != 0)
      std::fprintf(stderr, "[InformationMatrixCalculatorHip] calc_fitness_score failed: %s\n", apdgicp_last_error());
    return score;
  }

  /// the scalar part of calc_information_matrix (:29-52): inf is a column-major 6x6, diagonal
  void information_from_fitness(double fitness_score, double inf[36]) const {
    for (int q = 0; q < 36; q++) inf[q] = 0.0;
    double dx, dq;
    if (prm_.use_const_inf_matrix) {  // :30-35
      dx = 1.0 / prm_.const_stddev_x, dq = 1.0 / prm_.const_stddev_q;
    } else {
      const double min_var_x = std::pow(prm_.min_stddev_x, 2), max_var_x = std::pow(prm_.max_stddev_x, 2);
      const double min_var_q = std::pow(prm_.min_stddev_q, 2), max_var_q = std::pow(prm_.max_stddev_q, 2);
      const float w_x = (float)(1.0e-8 * weight(prm_.var_gain_a, prm_.fitness_score_thresh, min_var_x, max_var_x, fitness_score));  // :44-45 (float there too)
      const float w_q = (float)(1.0e-8 * weight(prm_.var_gain_a, prm_.fitness_score_thresh, min_var_q, max_var_q, fitness_score));
      dx = 1.0 / (double)w_x, dq = 1.0 / (double)w_q;
    }
    for (int d = 0; d < 3; d++) inf[d + 6 * d] = dx, inf[(d + 3) + 6 * (d + 3)] = dq;
  }

  /// calc_information_matrix(cloud1, cloud2, relpose) (:29-52)
  void calc_information_matrix(const float* cloud1_xyz, int64_t n1, const float* cloud2_xyz, int64_t n2, int64_t stride_bytes, const double relpose[16],
                               double inf[36], uint64_t token1 = 0, uint64_t token2 = 0) {
    const double fs = prm_.use_const_inf_matrix ? 0.0 : calc_fitness_score(cloud1_xyz, n1, cloud2_xyz, n2, stride_bytes, relpose, std::numeric_limits<double>::max(), token1, token2);
    information_from_fitness(fs, inf);
  }

  static double weight(double a, double max_x, double min_y, double max_y, double x) {  // hpp:40-43
    const double y = (1.0 - std::exp(-a * x)) / (1.0 - std::exp(-a * max_x));
    return min_y + (max_y - min_y) * y;
  }

 private:
  InformationMatrixParams prm_;
  apdgicp_handle* handle_ = nullptr;
};

}  // namespace fast_gicp
#endif
