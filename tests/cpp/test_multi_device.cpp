// One C++ process, one host thread per visible GPU: ShardedBatchAlignerHip (riv-slam_amd/cpp/sharded_batch_hip.hpp) shards the pairs
// of a batch file over the devices and gathers the 96-byte records with ncclAllGather (RCCL); the records must equal, byte
// for byte, those of ONE apdgicp_batch registering all pairs on device 0.  Then LoopVerifierHip (loop_verifier_hip.hpp)
// picks the best candidate of the same batch the way LoopDetector::matching does (loop_detector.cpp:387-441).
// usage: test_multi_device <batch.bin> [max_devices]        (no argument: compile/link check only)
//   batch.bin: int32 n_clouds, then per cloud int32 n + n*3 floats; int32 n_pairs, then per pair int32 src, int32 tgt, float guess[16]
// prints: world <D> pairs <P> sharded_equals_single <0|1> gathered_on_all_ranks <0|1> pipelined_equals_single <0|1> errors_ok <0|1> loop_best <i> loop_score <s>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "information_matrix_hip.hpp"
#include "loop_verifier_hip.hpp"
#include "sharded_batch_hip.hpp"

int main(int argc, char** argv) {
  if (argc < 2) {
    std::printf("compile-only\n");
    return 0;
  }
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) return 2;
  int32_t nc = 0;
  if (std::fread(&nc, 4, 1, f) != 1) return 2;
  std::vector<std::vector<float>> store((size_t)nc);
  std::vector<fast_gicp::ShardCloud> clouds;
  for (int c = 0; c < nc; c++) {
    int32_t n = 0;
    if (std::fread(&n, 4, 1, f) != 1) return 2;
    store[(size_t)c].resize((size_t)n * 3);
    if (std::fread(store[(size_t)c].data(), 4, (size_t)n * 3, f) != (size_t)n * 3) return 2;
    clouds.push_back({store[(size_t)c].data(), n, 12});
  }
  int32_t np = 0;
  if (std::fread(&np, 4, 1, f) != 1) return 2;
  std::vector<apdgicp_pair> pairs((size_t)np);
  for (int p = 0; p < np; p++) {
    int32_t st[2];
    if (std::fread(st, 4, 2, f) != 2 || std::fread(pairs[(size_t)p].guess, 4, 16, f) != 16) return 2;
    pairs[(size_t)p].source_cloud = st[0], pairs[(size_t)p].target_cloud = st[1];
  }
  std::fclose(f);

  int count = 0;
  if (apdgicp_device_count(&count) != 0 || count < 1) {
    std::fprintf(stderr, "no GPU: %s\n", apdgicp_last_error());
    return 3;
  }
  int D = count;
  if (argc > 2) D = std::min(D, std::atoi(argv[2]));
  std::vector<int> devices;
  for (int d = 0; d < D; d++) devices.push_back(d);
  apdgicp_params prm;
  apdgicp_default_params(&prm);
  prm.max_correspondence_distance = 2.0, prm.transformation_epsilon = 0.01, prm.azimuth_variance_deg = 1.0;

  // ---- all pairs on one handle (device 0)
  std::vector<apdgicp_result> single((size_t)np);
  {
    apdgicp_batch* b = nullptr;
    if (apdgicp_batch_create(&prm, 0, nullptr, &b) != 0) return 4;
    for (const auto& c : clouds)
      if (apdgicp_batch_add_cloud(b, c.xyz, c.n, c.stride_bytes, 0) < 0) return 4;
    if (apdgicp_batch_align(b, pairs.data(), np, single.data()) != 0) {
      std::fprintf(stderr, "single: %s\n", apdgicp_last_error());
      return 4;
    }
    apdgicp_batch_destroy(b);
  }
  // ---- sharded over D devices (one persistent worker thread each), records gathered with RCCL
  fast_gicp::ShardedBatchAlignerHip sharded(&prm, devices);
  if (!sharded.ok()) {
    std::fprintf(stderr, "sharded: %s\n", sharded.error().c_str());
    return 5;
  }
  std::vector<apdgicp_result> res;
  int same = 1, all_ranks = 1, pipelined = 1, errors_ok = 1;
  uint64_t last_ticket = 0;
  for (int rep = 0; rep < 2; rep++) {  // twice: buffers and communicators are reused
    if (sharded.enqueue(clouds, pairs, &last_ticket) != 0 || sharded.collect(last_ticket, &res, /*root=*/D - 1) != 0) {
      std::fprintf(stderr, "sharded align: %s\n", sharded.last_error_text().c_str());
      return 5;
    }
    same = same && (int)res.size() == np && std::memcmp(res.data(), single.data(), (size_t)np * sizeof(apdgicp_result)) == 0;
  }
  {  // every rank holds the same gathered buffer
    const auto parts = fast_gicp::block_partition(np, D);
    const size_t bytes = (size_t)(parts[0].second - parts[0].first) * D * sizeof(apdgicp_result);
    std::vector<char> ref(bytes), got(bytes);
    // collect(ticket, ..., root) waits for the gather of rank `root` only (and hipMemcpy does not wait for a non-blocking stream): every rank's
    // copy is read behind a collect with THAT rank as root (found by the ThreadSanitizer run of this schedule, tests/test_sanitizers.py)
    for (int r = 0; r < D; r++) all_ranks = all_ranks && sharded.collect(last_ticket, &res, r) == 0;
    (void)hipSetDevice(0);
    (void)hipMemcpy(ref.data(), sharded.gathered_on(0, last_ticket), bytes, hipMemcpyDeviceToHost);
    for (int r = 1; r < D; r++) {
      (void)hipSetDevice(r);
      (void)hipMemcpy(got.data(), sharded.gathered_on(r, last_ticket), bytes, hipMemcpyDeviceToHost);
      all_ranks = all_ranks && std::memcmp(ref.data(), got.data(), bytes) == 0;
    }
  }
  {  // pipelined: as many batches in flight as the aligner keeps, then two more, collected in order; the Gauss-Newton form too
    std::vector<uint64_t> tickets;
    for (int q = 0; q < sharded.in_flight() + 2; q++) {
      uint64_t t = 0;
      if (sharded.enqueue(clouds, pairs, &t) != 0) return 5;
      tickets.push_back(t);
      if ((int)tickets.size() > sharded.in_flight() - 1) {  // keep in_flight - 1 uncollected
        const uint64_t c = tickets[tickets.size() - (size_t)sharded.in_flight()];
        if (sharded.collect(c, &res) != 0) return 5;
        pipelined = pipelined && std::memcmp(res.data(), single.data(), (size_t)np * sizeof(apdgicp_result)) == 0;
      }
    }
    for (size_t q = tickets.size() - (size_t)sharded.in_flight() + 1; q < tickets.size(); q++) {
      if (sharded.collect(tickets[q], &res) != 0) return 5;
      pipelined = pipelined && std::memcmp(res.data(), single.data(), (size_t)np * sizeof(apdgicp_result)) == 0;
    }
    apdgicp_params gn = prm;
    gn.optimizer = APDGICP_OPT_GN, gn.max_iterations = 5, gn.transformation_epsilon = 1e-300, gn.rotation_epsilon = 1e-300;
    std::vector<apdgicp_result> single_gn((size_t)np), r1, r2;
    apdgicp_batch* b = nullptr;
    if (apdgicp_batch_create(&gn, 0, nullptr, &b) != 0) return 4;
    for (const auto& c : clouds)
      if (apdgicp_batch_add_cloud(b, c.xyz, c.n, c.stride_bytes, 0) < 0) return 4;
    if (apdgicp_batch_align(b, pairs.data(), np, single_gn.data()) != 0) return 4;
    apdgicp_batch_destroy(b);
    fast_gicp::ShardedBatchAlignerHip sharded_gn(&gn, devices, 3);
    uint64_t t1 = 0, t2 = 0, t3 = 0;
    if (!sharded_gn.ok() || sharded_gn.enqueue(clouds, pairs, &t1) != 0 || sharded_gn.enqueue(clouds, pairs, &t2) != 0 || sharded_gn.enqueue(clouds, pairs, &t3) != 0) return 5;
    if (sharded_gn.collect(t2, &r2) != 0 || sharded_gn.collect(t1, &r1) != 0 || sharded_gn.collect(t3, &res) != 0) return 5;
    for (const auto* v : {&r1, &r2, &res}) pipelined = pipelined && std::memcmp(v->data(), single_gn.data(), (size_t)np * sizeof(apdgicp_result)) == 0;
  }
  {  // errors: a pair that names a missing cloud is refused on the calling thread; a cloud with a non-finite point fails its own
     // batch at collect -- after every rank has been through the collective -- and the aligner stays usable
    std::vector<apdgicp_pair> bad_pairs = pairs;
    bad_pairs.back().source_cloud = (int32_t)clouds.size() + 3;
    uint64_t t = 0;
    errors_ok = errors_ok && sharded.enqueue(clouds, bad_pairs, &t) < 0;
    std::vector<float> poisoned = store[(size_t)pairs[0].source_cloud];
    poisoned[7] = std::nanf("");
    std::vector<fast_gicp::ShardCloud> bad_clouds = clouds;
    bad_clouds[(size_t)pairs[0].source_cloud].xyz = poisoned.data();
    errors_ok = errors_ok && sharded.enqueue(bad_clouds, pairs, &t) == 0 && sharded.collect(t, &res) < 0;
    errors_ok = errors_ok && sharded.align(clouds, pairs, &res) == 0 && std::memcmp(res.data(), single.data(), (size_t)np * sizeof(apdgicp_result)) == 0;
    // a rank whose record buffers could not be allocated still enters the batch's all-gather (fallback buffers): its batch
    // fails at collect -- which RETURNS -- and the next batch finds the communicator paired up and its buffers allocated anew
    sharded.debug_fail_next_record_allocation();
    errors_ok = errors_ok && sharded.enqueue(clouds, pairs, &t) == 0 && sharded.collect(t, &res) < 0 &&
                sharded.last_error_text().find("forced failure") != std::string::npos;
    errors_ok = errors_ok && sharded.align(clouds, pairs, &res) == 0 && std::memcmp(res.data(), single.data(), (size_t)np * sizeof(apdgicp_result)) == 0;
  }
  // ---- candidate selection: every pair whose target is the target of pair 0 is a candidate of that keyframe
  fast_gicp::LoopVerifierHip verifier(&prm, 0);
  std::vector<fast_gicp::LoopCloud> cand;
  std::vector<float> guesses;
  const int tgt0 = pairs[0].target_cloud;
  for (int p = 0; p < np; p++)
    if (pairs[(size_t)p].target_cloud == tgt0) {
      const auto& c = clouds[(size_t)pairs[(size_t)p].source_cloud];
      cand.push_back({c.xyz, c.n, c.stride_bytes});
      guesses.insert(guesses.end(), pairs[(size_t)p].guess, pairs[(size_t)p].guess + 16);
    }
  fast_gicp::LoopMatch match;
  const auto& t0 = clouds[(size_t)tgt0];
  if (verifier.matching({t0.xyz, t0.n, t0.stride_bytes}, cand, guesses.data(), 4.0, 0.5, &match) != 0) {
    std::fprintf(stderr, "verifier: %s\n", apdgicp_last_error());
    return 6;
  }
  // ---- edge information of pair 0 at its registered pose (InformationMatrixCalculator::calc_information_matrix: cloud1 = target)
  fast_gicp::InformationMatrixCalculatorHip infcalc;
  double relpose[16], inf[36];
  for (int q = 0; q < 16; q++) relpose[q] = (double)single[0].T[q];
  const auto& c1 = clouds[(size_t)pairs[0].target_cloud];
  const auto& c2 = clouds[(size_t)pairs[0].source_cloud];
  const double fs = infcalc.calc_fitness_score(c1.xyz, c1.n, c2.xyz, c2.n, 12, relpose);
  infcalc.calc_information_matrix(c1.xyz, c1.n, c2.xyz, c2.n, 12, relpose, inf);
  std::printf("world %d pairs %d sharded_equals_single %d gathered_on_all_ranks %d pipelined_equals_single %d errors_ok %d loop_best %d loop_score %.17g candidates %zu fitness %.17g inf00 %.17g inf33 %.17g\n", D,
              np, same, all_ranks, pipelined, errors_ok, match.best, match.best_score, cand.size(), fs, inf[0], inf[3 + 6 * 3]);
  return same && all_ranks && pipelined && errors_ok ? 0 : 1;
}
