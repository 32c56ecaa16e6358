// Loop-closure candidate batches sharded over the GPUs of one node from ONE C++ process (the Python / one
// process per GPU form is riv-slam_amd/sharded.py).
//
// Independent (source, target) pairs -- the candidates of loop_detector.cpp:222-236 / :404-423 -- are block-partitioned over
// the devices exactly like sharded.block_partition: pair p belongs to device p / ceil(P / D).  One host thread per device
// registers its block through its own apdgicp_batch handle; there is no collective on the data path.  The only exchange is
// ONE ncclAllGather (RCCL over xGMI) of the fixed-size result records, 96 bytes per pair, issued by every device thread on the
// stream its batch ran on, so that every device -- in particular the one next to the pose-graph owner -- ends up with all
// results.  The payload is a few KB: latency-bound, hence one call per batch.
//
// Needs <hip/hip_runtime_api.h> and <rccl/rccl.h> (link amdhip64 + rccl).  Header-only.
#ifndef FAST_GICP_SHARDED_BATCH_HIP_HPP
#define FAST_GICP_SHARDED_BATCH_HIP_HPP

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "apdgicp_hip.h"

namespace fast_gicp {

struct ShardCloud {
  const float* xyz;
  int64_t n;
  int64_t stride_bytes;
};

/// [begin, end) per device: contiguous blocks of ceil(P / D) pairs, the last ones short or empty (== sharded.block_partition)
inline std::vector<std::pair<int64_t, int64_t>> block_partition(int64_t n_pairs, int n_devices) {
  const int64_t per = n_pairs > 0 ? (n_pairs + n_devices - 1) / n_devices : 0;
  std::vector<std::pair<int64_t, int64_t>> out;
  for (int r = 0; r < n_devices; r++) out.emplace_back(std::min<int64_t>(r * per, n_pairs), std::min<int64_t>((r + 1) * per, n_pairs));
  return out;
}

class ShardedBatchAlignerHip {
 public:
  /// devices: HIP device indices, one rank each (rank r = devices[r])
  ShardedBatchAlignerHip(const apdgicp_params* params, const std::vector<int>& devices) : devices_(devices), ranks_(devices.size()) {
    const int D = (int)devices.size();
    comms_.assign((size_t)D, nullptr);
    if (D == 0) {
      error_ = "no devices";
      return;
    }
    if (ncclCommInitAll(comms_.data(), D, devices.data()) != ncclSuccess) {
      error_ = "ncclCommInitAll failed";
      comms_.assign((size_t)D, nullptr);
      return;
    }
    for (int r = 0; r < D; r++) {
      Rank& k = ranks_[(size_t)r];
      if (hipSetDevice(devices[(size_t)r]) != hipSuccess || hipStreamCreateWithFlags(&k.stream, hipStreamNonBlocking) != hipSuccess) {
        error_ = "stream creation failed on device " + std::to_string(devices[(size_t)r]);
        return;
      }
      if (apdgicp_batch_create(params, devices[(size_t)r], (void*)k.stream, &k.batch) != 0) {
        error_ = std::string("apdgicp_batch_create: ") + apdgicp_last_error();
        return;
      }
    }
  }
  ~ShardedBatchAlignerHip() {
    for (size_t r = 0; r < ranks_.size(); r++) {
      Rank& k = ranks_[r];
      (void)hipSetDevice(devices_[r]);
      if (k.batch) apdgicp_batch_destroy(k.batch);
      if (k.send) (void)hipFree(k.send);
      if (k.recv) (void)hipFree(k.recv);
      if (k.stream) (void)hipStreamDestroy(k.stream);
    }
    for (ncclComm_t c : comms_)
      if (c) ncclCommDestroy(c);
  }
  ShardedBatchAlignerHip(const ShardedBatchAlignerHip&) = delete;
  ShardedBatchAlignerHip& operator=(const ShardedBatchAlignerHip&) = delete;
  bool ok() const { return error_.empty(); }
  const std::string& error() const { return error_; }
  int world() const { return (int)devices_.size(); }

  /// Registers pairs[i] = (source cloud, target cloud, guess) -- indices into `clouds` -- and returns all P records in pair
  /// order (read back from rank `root`, which like every rank holds the gathered buffer).  0 or a negative apdgicp_status.
  int align(const std::vector<ShardCloud>& clouds, const std::vector<apdgicp_pair>& pairs, std::vector<apdgicp_result>* results, int root = 0) {
    if (!ok() || !results) return APDGICP_ERR_INVALID_ARG;
    const int D = world();
    const int64_t P = (int64_t)pairs.size();
    results->assign((size_t)P, apdgicp_result());
    if (P == 0) return 0;
    const auto parts = block_partition(P, D);
    const int64_t per = parts[0].second - parts[0].first;
    std::vector<int> rc((size_t)D, 0);
    std::vector<std::string> msg((size_t)D);
    std::vector<std::thread> threads;
    for (int r = 0; r < D; r++)
      threads.emplace_back([&, r]() { rc[(size_t)r] = run_rank(r, clouds, pairs, parts[(size_t)r].first, parts[(size_t)r].second, per, &msg[(size_t)r]); });
    for (auto& t : threads) t.join();
    for (int r = 0; r < D; r++)
      if (rc[(size_t)r] < 0) {
        error_text_ = msg[(size_t)r];
        return rc[(size_t)r];
      }
    // every rank's `recv` holds D blocks of `per` records (short blocks zero-padded): trim them back into pair order
    Rank& k = ranks_[(size_t)root];
    std::vector<apdgicp_result> all((size_t)(per * D));
    if (hipSetDevice(devices_[(size_t)root]) != hipSuccess ||
        hipMemcpy(all.data(), k.recv, all.size() * sizeof(apdgicp_result), hipMemcpyDeviceToHost) != hipSuccess)
      return APDGICP_ERR_HIP;
    for (int r = 0; r < D; r++)
      for (int64_t p = parts[(size_t)r].first; p < parts[(size_t)r].second; p++) (*results)[(size_t)p] = all[(size_t)(r * per + (p - parts[(size_t)r].first))];
    return 0;
  }
  const std::string& last_error_text() const { return error_text_; }
  /// device pointer of rank r's gathered buffer (world * per records), valid until the next align
  const void* gathered_on(int r) const { return ranks_[(size_t)r].recv; }

 private:
  struct Rank {
    hipStream_t stream = nullptr;
    apdgicp_batch* batch = nullptr;
    char* send = nullptr;
    char* recv = nullptr;
    size_t send_cap = 0, recv_cap = 0;
  };

  int run_rank(int r, const std::vector<ShardCloud>& clouds, const std::vector<apdgicp_pair>& pairs, int64_t b, int64_t e, int64_t per, std::string* msg) {
    Rank& k = ranks_[(size_t)r];
    const int D = world();
    auto fail = [&](int code, const char* what) {
      *msg = std::string(what) + ": " + apdgicp_last_error();
      return code;
    };
    if (hipSetDevice(devices_[(size_t)r]) != hipSuccess) return fail(APDGICP_ERR_HIP, "hipSetDevice");
    const size_t rec = sizeof(apdgicp_result);
    if ((size_t)per * rec > k.send_cap) {
      if (k.send) (void)hipFree(k.send);
      if (hipMalloc((void**)&k.send, (size_t)per * rec) != hipSuccess) return fail(APDGICP_ERR_HIP, "hipMalloc");
      k.send_cap = (size_t)per * rec;
    }
    if ((size_t)per * rec * D > k.recv_cap) {
      if (k.recv) (void)hipFree(k.recv);
      if (hipMalloc((void**)&k.recv, (size_t)per * rec * D) != hipSuccess) return fail(APDGICP_ERR_HIP, "hipMalloc");
      k.recv_cap = (size_t)per * rec * D;
    }
    if (hipMemsetAsync(k.send, 0, (size_t)per * rec, k.stream) != hipSuccess) return fail(APDGICP_ERR_HIP, "hipMemsetAsync");
    int rc = 0;
    if (e > b) {
      // this rank's clouds: the ones its pairs reference, renumbered in order of first use
      std::vector<int> local(clouds.size(), -1);
      std::vector<apdgicp_pair> mine;
      if ((rc = apdgicp_batch_clear(k.batch)) < 0) return fail(rc, "apdgicp_batch_clear");
      for (int64_t p = b; p < e; p++) {
        apdgicp_pair q = pairs[(size_t)p];
        for (int32_t* idx : {&q.source_cloud, &q.target_cloud}) {
          if (*idx < 0 || (size_t)*idx >= clouds.size()) return fail(APDGICP_ERR_INVALID_ARG, "pair references a missing cloud");
          if (local[(size_t)*idx] < 0) {
            const ShardCloud& c = clouds[(size_t)*idx];
            const int id = apdgicp_batch_add_cloud(k.batch, c.xyz, c.n, c.stride_bytes, 0);
            if (id < 0) return fail(id, "apdgicp_batch_add_cloud");
            local[(size_t)*idx] = id;
          }
          *idx = local[(size_t)*idx];
        }
        mine.push_back(q);
      }
      void* d_res = nullptr;
      if ((rc = apdgicp_batch_align_async(k.batch, mine.data(), (int64_t)mine.size(), &d_res)) < 0) return fail(rc, "apdgicp_batch_align_async");
      if (hipMemcpyAsync(k.send, d_res, mine.size() * rec, hipMemcpyDeviceToDevice, k.stream) != hipSuccess) return fail(APDGICP_ERR_HIP, "hipMemcpyAsync");
    }
    // every rank calls the collective, an empty block contributes zeros
    if (ncclAllGather(k.send, k.recv, (size_t)per * rec, ncclChar, comms_[(size_t)r], k.stream) != ncclSuccess) return fail(APDGICP_ERR_HIP, "ncclAllGather");
    if (hipStreamSynchronize(k.stream) != hipSuccess) return fail(APDGICP_ERR_HIP, "hipStreamSynchronize");
    return 0;
  }

  std::vector<int> devices_;
  std::vector<Rank> ranks_;
  std::vector<ncclComm_t> comms_;
  std::string error_, error_text_;
};

}  // namespace fast_gicp
#endif
