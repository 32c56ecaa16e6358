// Candidate-batch loop-closure verification for C++ callers (a nodelet), on top of the C ABI (include/apdgicp_hip.h).
//
// Mirrors LoopDetector::matching (radar_graph_slam/src/radar_graph_slam/loop_detector.cpp:387-441) and the live
// single-candidate check of performScanContextLoopClosure (:222-236): the new keyframe is the TARGET, every candidate
// keyframe a SOURCE.  The reference aligns the candidates one after the other and asks getFitnessScore after each; here all
// candidates are registered by ONE apdgicp_batch_align and scored by ONE apdgicp_batch_fitness, and the selection rule is
// applied to the results in the reference's order: a candidate that did not converge, or whose score is worse than the best
// so far, is skipped (:416-418); the loop is rejected when best_score > fitness_score_thresh (:431).
//
// Header-only; no PCL or Eigen types in the interface (points: address of the first x and a byte stride, like the C ABI).
#ifndef FAST_GICP_LOOP_VERIFIER_HIP_HPP
#define FAST_GICP_LOOP_VERIFIER_HIP_HPP

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <vector>

#include "apdgicp_hip.h"

namespace fast_gicp {

struct LoopCloud {
  const float* xyz;       // first coordinate of the first point (host memory)
  int64_t n;              // points
  int64_t stride_bytes;   // 32 for pcl::PointXYZI, 16 for float4, 12 for packed xyz
};

struct LoopMatch {
  int best = -1;                        // index into the candidate list, -1: no loop (loop_detector.cpp:431-434)
  float relative_pose[16] = {0};        // registration->getFinalTransformation() of the best candidate, column-major (:422)
  double best_score = std::numeric_limits<double>::max();
  std::vector<double> scores;           // getFitnessScore(fitness_score_max_range) of every candidate
  std::vector<apdgicp_result> results;  // align() of every candidate (converged, iterations, T ...)
};

class LoopVerifierHip {
 public:
  explicit LoopVerifierHip(const apdgicp_params* params = nullptr, int device = 0) {
    if (apdgicp_batch_create(params, device, nullptr, &batch_) != 0) report("apdgicp_batch_create");
  }
  ~LoopVerifierHip() {
    if (batch_) apdgicp_batch_destroy(batch_);
  }
  LoopVerifierHip(const LoopVerifierHip&) = delete;
  LoopVerifierHip& operator=(const LoopVerifierHip&) = delete;
  bool ok() const { return batch_ != nullptr; }
  int setParams(const apdgicp_params& p) { return batch_ ? apdgicp_batch_set_params(batch_, &p) : APDGICP_ERR_HIP; }

  /// guesses: n_candidates x 16 floats, column-major ((new_keyframe_estimate^-1 * candidate_estimate) with guess(2,3) = 0,
  /// loop_detector.cpp:405-411), or nullptr for the identity (:225).  Returns 0 or a negative apdgicp_status.
  int matching(const LoopCloud& new_keyframe, const std::vector<LoopCloud>& candidates, const float* guesses, double fitness_score_max_range,
               double fitness_score_thresh, LoopMatch* out) {
    if (!out) return APDGICP_ERR_INVALID_ARG;
    *out = LoopMatch();
    if (!batch_) return APDGICP_ERR_HIP;
    if (candidates.empty()) return 0;  // :388-390
    int rc = apdgicp_batch_clear(batch_);
    if (rc < 0) return rc;
    const int64_t n = (int64_t)candidates.size();
    // cloud 0 = the new keyframe (setInputTarget(new_keyframe->cloud), :392), clouds 1 .. n = the candidates.  With one row stride
    // (the usual case: every cloud a pcl::PointCloud<PointT>) all of them go over in ONE call -- packed by the library's host
    // threads into one pinned region, one copy -- instead of n + 1 clouds read over PCIe one by one.
    bool one_stride = true;
    for (const LoopCloud& c : candidates) one_stride &= c.stride_bytes == new_keyframe.stride_bytes;
    if (one_stride) {
      std::vector<const float*> ptrs((size_t)n + 1);
      std::vector<int64_t> ns((size_t)n + 1);
      ptrs[0] = new_keyframe.xyz, ns[0] = new_keyframe.n;
      for (int64_t i = 0; i < n; i++) ptrs[(size_t)i + 1] = candidates[(size_t)i].xyz, ns[(size_t)i + 1] = candidates[(size_t)i].n;
      if ((rc = apdgicp_batch_set_clouds(batch_, 0, (int32_t)(n + 1), ptrs.data(), ns.data(), new_keyframe.stride_bytes, 0)) < 0) return rc;
    }
    const int tgt = one_stride ? 0 : apdgicp_batch_add_cloud(batch_, new_keyframe.xyz, new_keyframe.n, new_keyframe.stride_bytes, 0);
    if (tgt < 0) return tgt;
    std::vector<apdgicp_pair> pairs((size_t)n);
    for (int64_t i = 0; i < n; i++) {
      const int src = one_stride ? (int)i + 1 : apdgicp_batch_add_cloud(batch_, candidates[(size_t)i].xyz, candidates[(size_t)i].n, candidates[(size_t)i].stride_bytes, 0);
      if (src < 0) return src;
      pairs[(size_t)i].source_cloud = src, pairs[(size_t)i].target_cloud = tgt;
      if (guesses) {
        std::memcpy(pairs[(size_t)i].guess, guesses + 16 * i, 16 * sizeof(float));
      } else {
        std::memset(pairs[(size_t)i].guess, 0, 16 * sizeof(float));
        pairs[(size_t)i].guess[0] = pairs[(size_t)i].guess[5] = pairs[(size_t)i].guess[10] = pairs[(size_t)i].guess[15] = 1.f;
      }
    }
    out->results.resize((size_t)n);
    out->scores.assign((size_t)n, std::numeric_limits<double>::max());
    if ((rc = apdgicp_batch_align(batch_, pairs.data(), n, out->results.data())) < 0) return rc;
    if ((rc = apdgicp_batch_fitness(batch_, pairs.data(), n, nullptr, fitness_score_max_range, out->scores.data(), nullptr)) < 0) return rc;
    for (int64_t i = 0; i < n; i++) {  // :415-423
      if (!out->results[(size_t)i].converged || out->scores[(size_t)i] > out->best_score) continue;
      out->best_score = out->scores[(size_t)i];
      out->best = (int)i;
    }
    if (out->best >= 0 && out->best_score > fitness_score_thresh) out->best = -1;  // "loop not found...", :431-434
    if (out->best >= 0) std::memcpy(out->relative_pose, out->results[(size_t)out->best].T, sizeof(out->relative_pose));
    return 0;
  }

 private:
  void report(const char* what) const { std::fprintf(stderr, "[LoopVerifierHip] %s failed: %s\n", what, apdgicp_last_error()); }
  apdgicp_batch* batch_ = nullptr;
};

}  // namespace fast_gicp
#endif
