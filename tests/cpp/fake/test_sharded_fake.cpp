// ShardedBatchAlignerHip's host-side thread logic (riv-slam_amd/cpp/sharded_batch_hip.hpp: enqueue / collect / late gather / abort_all, one
// worker thread per device) on FOUR fake devices, linked against tests/cpp/fake/fake_backend.cpp instead of amdhip64 / rccl / the library:
// the schedule of tests/cpp/test_multi_device.cpp -- batches twice, every rank's gathered copy, more batches in flight than slots, the
// Gauss-Newton form collected out of order, a bad pair, a poisoned cloud that fails ONE rank mid-batch, a failed record allocation (fallback
// buffers), and a block larger than the fallback (the communicators are aborted, every collect returns) -- built by tests/test_sanitizers.py
// with -fsanitize=thread and with -fsanitize=address,undefined.  Exit code 0 and "ok 1" = every check held.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "sharded_batch_hip.hpp"

extern "C" long fake_live_allocations();

int main() {
  const int D = 4, NC = 24, NP = 37;  // 37 pairs over 4 ranks: blocks of 10, 10, 10, 7 (a short last block)
  std::vector<std::vector<float>> store((size_t)NC);
  std::vector<fast_gicp::ShardCloud> clouds;
  for (int c = 0; c < NC; c++) {
    store[(size_t)c].assign((size_t)(300 + 17 * c) * 3, 0.25f * (float)c);
    clouds.push_back({store[(size_t)c].data(), (int64_t)store[(size_t)c].size() / 3, 12});
  }
  std::vector<apdgicp_pair> pairs((size_t)NP);
  for (int p = 0; p < NP; p++) {
    pairs[(size_t)p].source_cloud = (p * 5) % NC, pairs[(size_t)p].target_cloud = (p * 7 + 3) % NC;
    for (int q = 0; q < 16; q++) pairs[(size_t)p].guess[q] = (float)(p + q);
  }
  apdgicp_params prm;
  apdgicp_default_params(&prm);
  std::vector<int> devices = {0, 1, 2, 3};
  int ok = 1;
  std::vector<apdgicp_result> single((size_t)NP), res;
  {
    apdgicp_batch* b = nullptr;
    apdgicp_batch_create(&prm, 0, nullptr, &b);
    for (const auto& c : clouds) apdgicp_batch_add_cloud(b, c.xyz, c.n, c.stride_bytes, 0);
    ok = ok && apdgicp_batch_align(b, pairs.data(), NP, single.data()) == 0;
    apdgicp_batch_destroy(b);
  }
  auto same = [&](const std::vector<apdgicp_result>& r) { return (int)r.size() == NP && std::memcmp(r.data(), single.data(), (size_t)NP * sizeof(apdgicp_result)) == 0; };
  {
    fast_gicp::ShardedBatchAlignerHip sharded(&prm, devices);  // pooled LM: one handle per rank, 4 batches in flight
    ok = ok && sharded.ok();
    uint64_t t = 0;
    for (int rep = 0; rep < 2; rep++) ok = ok && sharded.enqueue(clouds, pairs, &t) == 0 && sharded.collect(t, &res, /*root=*/D - 1) == 0 && same(res);
    {  // every rank holds the same gathered buffer
      const auto parts = fast_gicp::block_partition(NP, D);
      const size_t bytes = (size_t)(parts[0].second - parts[0].first) * D * sizeof(apdgicp_result);
      // (collect(ticket, ..., root) waits for the gather of rank `root` only: a rank's copy is read after a collect with THAT rank as root --
      // ThreadSanitizer found the comparison racing with the other ranks' gathers when only the last rank had been waited for)
      std::vector<apdgicp_result> tmp;
      for (int r = 0; r < D; r++) ok = ok && sharded.collect(t, &tmp, r) == 0 && same(tmp);
      for (int r = 1; r < D; r++) ok = ok && std::memcmp(sharded.gathered_on(0, t), sharded.gathered_on(r, t), bytes) == 0;
    }
    std::vector<uint64_t> tickets;  // more batches than slots, in_flight - 1 kept uncollected
    for (int q = 0; q < 3 * sharded.in_flight() + 2; q++) {
      uint64_t tk = 0;
      ok = ok && sharded.enqueue(clouds, pairs, &tk) == 0;
      tickets.push_back(tk);
      if ((int)tickets.size() > sharded.in_flight() - 1) ok = ok && sharded.collect(tickets[tickets.size() - (size_t)sharded.in_flight()], &res) == 0 && same(res);
    }
    for (size_t q = tickets.size() - (size_t)sharded.in_flight() + 1; q < tickets.size(); q++) ok = ok && sharded.collect(tickets[q], &res) == 0 && same(res);
    // errors: refused on the calling thread; a poisoned cloud fails ITS batch on the rank that owns the pair, every rank still gathers
    std::vector<apdgicp_pair> bad_pairs = pairs;
    bad_pairs.back().source_cloud = NC + 3;
    ok = ok && sharded.enqueue(clouds, bad_pairs, &t) < 0;
    std::vector<float> poisoned = store[(size_t)pairs[12].source_cloud];
    poisoned[0] = std::nanf("");
    std::vector<fast_gicp::ShardCloud> bad_clouds = clouds;
    bad_clouds[(size_t)pairs[12].source_cloud].xyz = poisoned.data();   // pair 12 lives on rank 1
    uint64_t tb = 0, tg = 0;
    ok = ok && sharded.enqueue(bad_clouds, pairs, &tb) == 0 && sharded.enqueue(clouds, pairs, &tg) == 0;   // a good batch right behind it
    ok = ok && sharded.collect(tg, &res) == 0 && same(res);
    ok = ok && sharded.collect(tb, &res) < 0 && sharded.last_error_text().find("rank 1") != std::string::npos;
    // a rank whose record buffers could not be allocated enters the gather on its fallback buffers
    sharded.debug_fail_next_record_allocation();
    ok = ok && sharded.enqueue(clouds, pairs, &t) == 0 && sharded.collect(t, &res) < 0 && sharded.last_error_text().find("forced failure") != std::string::npos;
    ok = ok && sharded.align(clouds, pairs, &res) == 0 && same(res);
    // two caller threads are NOT supported on one aligner; one caller enqueueing while the workers gather is the design: a burst
    for (int q = 0; q < 12; q++) ok = ok && sharded.align(clouds, pairs, &res) == 0 && same(res);
  }
  {  // Gauss-Newton form: one handle per batch in flight, collected out of order
    apdgicp_params gn = prm;
    gn.optimizer = APDGICP_OPT_GN;
    fast_gicp::ShardedBatchAlignerHip sharded_gn(&gn, devices, 3);
    uint64_t t1 = 0, t2 = 0, t3 = 0;
    std::vector<apdgicp_result> r1, r2;
    ok = ok && sharded_gn.ok() && sharded_gn.enqueue(clouds, pairs, &t1) == 0 && sharded_gn.enqueue(clouds, pairs, &t2) == 0 && sharded_gn.enqueue(clouds, pairs, &t3) == 0;
    ok = ok && sharded_gn.collect(t2, &r2) == 0 && sharded_gn.collect(t1, &r1) == 0 && sharded_gn.collect(t3, &res) == 0 && same(r1) && same(r2) && same(res);
  }
  {  // a block larger than the fallback buffers (64 KB = 682 records) on a rank that lost its record buffers: that rank cannot enter the
     // gather -> abort_all: every collect RETURNS with an error, nothing hangs, the destructor joins
    const int BIG = 4 * 700;
    std::vector<apdgicp_pair> many((size_t)BIG);
    for (int p = 0; p < BIG; p++) many[(size_t)p] = pairs[(size_t)(p % NP)];
    fast_gicp::ShardedBatchAlignerHip sharded(&prm, devices, 2);
    uint64_t t = 0, t2 = 0;
    ok = ok && sharded.ok();
    sharded.debug_fail_next_record_allocation();
    ok = ok && sharded.enqueue(clouds, many, &t) == 0;
    ok = ok && sharded.collect(t, &res) < 0;
    ok = ok && (sharded.enqueue(clouds, pairs, &t2) < 0 || sharded.collect(t2, &res) < 0);   // the aligner is dead after an abort: errors, not hangs
  }
  std::printf("ok %d live_allocations %ld\n", ok, fake_live_allocations());
  return ok && fake_live_allocations() == 0 ? 0 : 1;
}
