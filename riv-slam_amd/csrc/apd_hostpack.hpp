// Host-side helpers of the engine that touch neither HIP nor the kernels: packing host clouds into pinned memory and the small
// thread pool that does it for a batch of clouds.  Plain C++ (std::thread), so that tests/test_sanitizers.py can run exactly this
// code under ThreadSanitizer / AddressSanitizer on a box without a GPU (the engine itself includes it).
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace apd {

struct alignas(16) HostF4 {  // the layout of HIP's float4
  float x, y, z, w;
};

// A scan-sized host cloud into its pinned buffer: {x, y, z, 1} per point, the bounding box behind the points.  Touches nothing but
// its arguments (the single-cloud path of set_cloud; a batch of host clouds goes through compact_xyz_host below).
inline void pack_staged_host(HostF4* dst, const char* raw, int64_t n, int64_t stride_bytes) {
  const float inf = std::numeric_limits<float>::infinity();
  // four independent running boxes (a single one is a chain of dependent min/max, twice the time of the packing itself);
  // `v < lo ? v : lo` leaves a NaN coordinate out, like the device's fminf
  float lo[4][3], hi[4][3];
  for (int u = 0; u < 4; u++)
    for (int a = 0; a < 3; a++) lo[u][a] = inf, hi[u][a] = -inf;
  auto put = [&](int64_t q, int u) {
    const float* sp = (const float*)(raw + q * stride_bytes);
    const float v[3] = {sp[0], sp[1], sp[2]};
    dst[q] = HostF4{v[0], v[1], v[2], 1.0f};
    for (int a = 0; a < 3; a++) lo[u][a] = v[a] < lo[u][a] ? v[a] : lo[u][a], hi[u][a] = v[a] > hi[u][a] ? v[a] : hi[u][a];
  };
  int64_t q = 0;
  {  // one 16-byte load, blend, store: the fourth float read with a point is its own padding or the next point's x -- inside the caller's buffer
    typedef float v4f __attribute__((ext_vector_type(4)));
    v4f vlo[2] = {v4f(inf), v4f(inf)}, vhi[2] = {v4f(-inf), v4f(-inf)};
    for (; q + 2 <= n - 1; q += 2)  // (not the last point: its fourth float may lie outside the caller's buffer)
      for (int u = 0; u < 2; u++) {
        v4f v;
        memcpy(&v, raw + (q + u) * stride_bytes, 16);
        vlo[u] = v < vlo[u] ? v : vlo[u], vhi[u] = v > vhi[u] ? v : vhi[u];
        v.w = 1.0f;
        memcpy(&dst[q + u], &v, 16);
      }
    for (int u = 0; u < 2; u++)
      for (int a = 0; a < 3; a++) lo[u][a] = vlo[u][a], hi[u][a] = vhi[u][a];
  }
  for (; q + 4 <= n; q += 4) put(q, 0), put(q + 1, 1), put(q + 2, 2), put(q + 3, 3);
  for (; q < n; q++) put(q, 0);
  for (int u = 1; u < 4; u++)
    for (int a = 0; a < 3; a++) lo[0][a] = std::min(lo[0][a], lo[u][a]), hi[0][a] = std::max(hi[0][a], hi[u][a]);
  dst[n] = HostF4{lo[0][0], lo[0][1], lo[0][2], 0.f}, dst[n + 1] = HostF4{hi[0][0], hi[0][1], hi[0][2], 0.f};
}

// A batch's host clouds on their way into ONE pinned region (set_clouds_host): only {x, y, z}, 12 bytes a point -- a plain copy when the
// caller's points already are 12 bytes apart, a compaction otherwise (PCL's PointXYZI: 32 bytes).  The device's pack launch turns them into
// {x, y, z, 1} like any device-resident input; nothing else is computed here (round 5 wrote float4 and a bounding box nobody read: a third
// more bytes over PCIe and twice the host time).  Reads exactly 12 bytes per point.
inline void compact_xyz_host(float* dst, const char* raw, int64_t n, int64_t stride_bytes) {
  if (stride_bytes == 12) {
    memcpy(dst, raw, (size_t)n * 12);
    return;
  }
  for (int64_t q = 0; q < n; q++) memcpy(dst + 3 * q, raw + q * stride_bytes, 12);
}

// A few persistent host threads for set_clouds_host (64 clouds of 8192 points: 0.25 ms of packing on one core, a third of a
// step).  Created on first use, parked on a condition variable in between.
struct HostPool {
  std::vector<std::thread> th;
  std::mutex m;
  std::condition_variable cv, cv_done;
  const std::function<void(int)>* fn = nullptr;
  int n = 0, busy = 0;
  std::atomic<int> next{0};
  uint64_t gen = 0;
  bool stop = false;
  void start(int k) {
    for (int t = 0; t < k; t++)
      th.emplace_back([this]() {
        uint64_t seen = 0;
        for (;;) {
          const std::function<void(int)>* f;
          int cnt;
          {
            std::unique_lock<std::mutex> lk(m);
            cv.wait(lk, [&]() { return stop || gen != seen; });
            if (stop) return;
            seen = gen, f = fn, cnt = n;
          }
          for (int i; (i = next.fetch_add(1, std::memory_order_relaxed)) < cnt;) (*f)(i);
          std::lock_guard<std::mutex> lk(m);
          if (--busy == 0) cv_done.notify_one();
        }
      });
  }
  std::mutex run_mu;  // one run at a time (the pool is shared by every engine of the process)
  void run(int count, const std::function<void(int)>& f) {  // f(0 .. count - 1), the caller takes part; returns when all are done
    std::lock_guard<std::mutex> run_lk(run_mu);
    {
      std::lock_guard<std::mutex> lk(m);
      fn = &f, n = count, busy = (int)th.size(), next.store(0, std::memory_order_relaxed), gen++;
    }
    cv.notify_all();
    for (int i; (i = next.fetch_add(1, std::memory_order_relaxed)) < count;) f(i);
    std::unique_lock<std::mutex> lk(m);
    cv_done.wait(lk, [&]() { return busy == 0; });
  }
  ~HostPool() {
    {
      std::lock_guard<std::mutex> lk(m);
      stop = true;
    }
    cv.notify_all();
    for (auto& t : th) t.join();
  }
};
// ONE pool per process, created by the first batch of host clouds that wants it (APDGICP_HOST_THREADS is read then: a
// per-process setting) and shared by every engine: four bench handles or eight ShardedBatchAlignerHip handles per device each
// with three parked threads of their own were 12 - 24 idle threads.  run() is serialised by the pool's own mutex.
inline HostPool* shared_host_pool(int* threads_out) {
  static std::mutex mu;
  static std::unique_ptr<HostPool> pool;
  static int want = 0;
  std::lock_guard<std::mutex> lk(mu);
  if (!want) {
    const char* e = getenv("APDGICP_HOST_THREADS");  // threads packing host clouds, the caller included (1: no pool)
    const int hc = (int)std::thread::hardware_concurrency();
    want = std::max(1, e ? atoi(e) : std::min(4, hc > 1 ? hc / 2 : 1));
    if (want > 1) pool.reset(new HostPool()), pool->start(want - 1);
  }
  *threads_out = want;
  return pool.get();
}


}  // namespace apd
