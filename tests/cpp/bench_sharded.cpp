// The C++ caller's form of bench.py's step, through ShardedBatchAlignerHip (riv-slam_amd/cpp/sharded_batch_hip.hpp): every visible
// device registers the SAME block of pairs (weak scaling: P pairs per device), clouds resident in that device's HBM and
// re-registered every batch (packed, sorted, covariances recomputed), `in_flight` batches in flight per device, one
// ncclAllGather of the 96-byte records per batch.  ONE host thread drives it: enqueue(batch s), collect(batch s - in_flight).
// usage: bench_sharded <batch.bin> <gn|lm> [steps=40] [warmup=8] [records.bin] [in_flight=4] [max_devices]   (no argument: compile/link check)
//   batch.bin: the format of test_multi_device.cpp (int32 n_clouds, per cloud int32 n + n*3 floats; int32 n_pairs, per pair int32 src, tgt, float guess[16])
//   gn: BASELINE configs[1] parameters (GN, 20 iterations, no early exit); lm: the launch-file parameters (the reference's optimiser)
// prints one JSON object: ms_per_step (all devices work concurrently), registrations_per_s (whole job), records_stable (every
// collected batch byte-equal to the first).  records.bin receives the records of device 0's block.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "sharded_batch_hip.hpp"

int main(int argc, char** argv) {
  if (argc < 3) {
    std::printf("compile-only\n");
    return 0;
  }
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) return 2;
  int32_t nc = 0;
  if (std::fread(&nc, 4, 1, f) != 1) return 2;
  std::vector<std::vector<float>> store((size_t)nc);
  for (int c = 0; c < nc; c++) {
    int32_t n = 0;
    if (std::fread(&n, 4, 1, f) != 1) return 2;
    store[(size_t)c].resize((size_t)n * 3);
    if (std::fread(store[(size_t)c].data(), 4, (size_t)n * 3, f) != (size_t)n * 3) return 2;
  }
  int32_t np = 0;
  if (std::fread(&np, 4, 1, f) != 1) return 2;
  std::vector<apdgicp_pair> block((size_t)np);
  for (int p = 0; p < np; p++) {
    int32_t st[2];
    if (std::fread(st, 4, 2, f) != 2 || std::fread(block[(size_t)p].guess, 4, 16, f) != 16) return 2;
    block[(size_t)p].source_cloud = st[0], block[(size_t)p].target_cloud = st[1];
  }
  std::fclose(f);
  const bool lm = std::string(argv[2]) == "lm";
  const int steps = argc > 3 ? std::atoi(argv[3]) : 40, warmup = argc > 4 ? std::atoi(argv[4]) : 8;
  const char* rec_path = argc > 5 ? argv[5] : nullptr;
  const int in_flight = argc > 6 ? std::atoi(argv[6]) : 4;

  int count = 0;
  if (apdgicp_device_count(&count) != 0 || count < 1) {
    std::fprintf(stderr, "no GPU: %s\n", apdgicp_last_error());
    return 3;
  }
  int D = count;
  if (argc > 7) D = std::min(D, std::atoi(argv[7]));
  std::vector<int> devices;
  for (int d = 0; d < D; d++) devices.push_back(d);
  apdgicp_params prm;
  apdgicp_default_params(&prm);
  prm.max_correspondence_distance = 2.0, prm.azimuth_variance_deg = 1.0;
  if (lm) {
    prm.transformation_epsilon = 0.1;
  } else {
    prm.optimizer = APDGICP_OPT_GN, prm.max_iterations = 20, prm.transformation_epsilon = 1e-300, prm.rotation_epsilon = 1e-300;
  }

  // the block's clouds, resident on every device; the job = D copies of the block, device d's pairs pointing at device d's clouds
  std::vector<fast_gicp::ShardCloud> clouds;
  std::vector<apdgicp_pair> pairs;
  std::vector<void*> dev_mem;
  for (int d = 0; d < D; d++) {
    if (hipSetDevice(d) != hipSuccess) return 3;
    for (int c = 0; c < nc; c++) {
      void* p = nullptr;
      const size_t bytes = store[(size_t)c].size() * 4;
      if (hipMalloc(&p, bytes) != hipSuccess || hipMemcpy(p, store[(size_t)c].data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return 3;
      dev_mem.push_back(p);
      clouds.push_back({(const float*)p, (int64_t)(store[(size_t)c].size() / 3), 12, 1});
    }
    for (apdgicp_pair q : block) {
      q.source_cloud += d * nc, q.target_cloud += d * nc;
      pairs.push_back(q);
    }
  }
  int rc = 1;
  if (std::getenv("BENCH_SHARDED_DIRECT")) {  // experiment: the same schedule straight on the C ABI, device 0, calling thread (no aligner, no gather)
    apdgicp_batch* b = nullptr;
    if (apdgicp_batch_create(&prm, 0, nullptr, &b) != 0) return 4;
    const int F = in_flight;
    std::vector<const float*> ptrs;
    std::vector<int64_t> ns;
    for (int c = 0; c < nc; c++) ptrs.push_back(clouds[(size_t)c].xyz), ns.push_back(clouds[(size_t)c].n);
    std::vector<apdgicp_result> res((size_t)np);
    auto run = [&](int count_) -> int {
      std::vector<uint64_t> tk((size_t)F, 0);
      for (int s = 0; s < count_ + F; s++) {
        const int f = s % F;
        if (tk[(size_t)f] && apdgicp_batch_align_collect(b, tk[(size_t)f], nullptr, res.data()) != 0) return -1;
        tk[(size_t)f] = 0;
        if (s < count_) {
          std::vector<apdgicp_pair> mine = block;
          for (auto& q : mine) q.source_cloud += f * nc, q.target_cloud += f * nc;
          if (apdgicp_batch_set_clouds(b, f * nc, nc, ptrs.data(), ns.data(), 12, 1) != 0 || apdgicp_batch_align_enqueue(b, mine.data(), np, &tk[(size_t)f]) != 0) return -1;
        }
      }
      return 0;
    };
    if (run(warmup) != 0) return 5;
    const auto t0 = std::chrono::steady_clock::now();
    if (run(steps) != 0) return 5;
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / steps;
    std::printf("{\"what\": \"C ABI directly, one handle, %d in flight\", \"ms_per_step\": %.4f}\n", F, ms);
    apdgicp_batch_destroy(b);
    return 0;
  }
  {
    fast_gicp::ShardedBatchAlignerHip sharded(&prm, devices, in_flight);
    if (!sharded.ok()) {
      std::fprintf(stderr, "sharded: %s\n", sharded.error().c_str());
      return 5;
    }
    if (std::getenv("BENCH_SHARDED_NO_RCCL_WHEN_ALONE")) sharded.set_gather_when_alone(false);
    const int F = sharded.in_flight();
    std::vector<apdgicp_result> first, res;
    int stable = 1;
    auto run = [&](int count_) -> int {
      std::vector<uint64_t> tickets;
      // one batch more than the devices keep in flight is outstanding: when the newest arrives at a worker all its slots are
      // busy, so it collects its oldest batch and starts the new one back to back -- the caller's collect of that oldest batch
      // then finds it done, and no device waits for the caller between two batches
      for (int s = 0; s < count_ + F; s++) {
        if (s < count_) {
          uint64_t t = 0;
          if (sharded.enqueue(clouds, pairs, &t) != 0) return -1;
          tickets.push_back(t);
        }
        if (s >= F) {  // the oldest outstanding batch
          if (sharded.collect(tickets[(size_t)(s - F)], &res) != 0) {
            std::fprintf(stderr, "collect: %s\n", sharded.last_error_text().c_str());
            return -1;
          }
          if (first.empty()) first = res;
          stable = stable && res.size() == first.size() && std::memcmp(res.data(), first.data(), res.size() * sizeof(apdgicp_result)) == 0;
        }
      }
      return 0;
    };
    if (run(warmup) != 0) return 5;
    const auto t0 = std::chrono::steady_clock::now();
    if (run(steps) != 0) return 5;
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / steps;
    for (int d = 1; d < D; d++)  // every device registered the same block
      stable = stable && std::memcmp(first.data(), first.data() + (size_t)d * (size_t)np, (size_t)np * sizeof(apdgicp_result)) == 0;
    if (rec_path) {
      FILE* o = std::fopen(rec_path, "wb");
      if (!o || std::fwrite(first.data(), sizeof(apdgicp_result), (size_t)np, o) != (size_t)np) return 2;
      std::fclose(o);
    }
    std::printf("{\"what\": \"C++ ShardedBatchAlignerHip, one host thread\", \"optimizer\": \"%s\", \"devices\": %d, \"pairs_per_device\": %d, \"in_flight\": %d, "
                "\"steps\": %d, \"ms_per_step\": %.4f, \"registrations_per_s\": %.1f, \"records_stable\": %d}\n",
                lm ? "lm" : "gn", D, np, F, steps, ms, 1e3 * np * D / ms, stable);
    rc = stable ? 0 : 1;
  }
  for (void* p : dev_mem) (void)hipFree(p);
  return rc;
}
