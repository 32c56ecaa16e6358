// riv-slam_amd/csrc/apd_hostpack.hpp -- the engine's host-side packing of host clouds and its thread pool -- on the CPU box under
// ThreadSanitizer and AddressSanitizer (tests/test_sanitizers.py): (1) pack_staged_host against a naive loop for strides of 12, 16 and 32
// bytes, sizes 1 .. 70 and 8192, NaN coordinates, from heap blocks of EXACTLY n * stride bytes (the vector path reads 16 bytes per point:
// the sanitizer sees whether the last point's fourth float is ever touched); (2) the process-wide pool, run() called from three threads at
// once (the engines of a process share it), every task executed exactly once.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "apd_hostpack.hpp"

static int check_pack(int64_t n, int64_t stride, unsigned seed) {
  char* raw = (char*)std::malloc((size_t)(n * stride));  // exactly: nothing behind the last point
  std::vector<float> want((size_t)n * 3);
  unsigned s = seed * 2654435761u + 12345u;
  for (int64_t q = 0; q < n; q++) {
    float* p = (float*)(raw + q * stride);
    for (int64_t a = 0; a < stride / 4; a++) {
      s = s * 1664525u + 1013904223u;
      p[a] = (float)(s >> 8) * (1.0f / 65536.0f) - 100.0f;
    }
    if (n > 3 && q == n / 2) p[1] = std::nanf("");
    for (int a = 0; a < 3; a++) want[(size_t)(3 * q + a)] = p[a];
  }
  std::vector<apd::HostF4> dst((size_t)n + 2);
  apd::pack_staged_host(dst.data(), raw, n, stride);
  int ok = 1;
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int64_t q = 0; q < n; q++) {
    const float* w = &want[(size_t)(3 * q)];
    const apd::HostF4& d = dst[(size_t)q];
    ok = ok && std::memcmp(&d.x, w, 12) == 0 && d.w == 1.0f;
    for (int a = 0; a < 3; a++) lo[a] = w[a] < lo[a] ? w[a] : lo[a], hi[a] = w[a] > hi[a] ? w[a] : hi[a];
  }
  ok = ok && dst[(size_t)n].x == lo[0] && dst[(size_t)n].y == lo[1] && dst[(size_t)n].z == lo[2];
  ok = ok && dst[(size_t)n + 1].x == hi[0] && dst[(size_t)n + 1].y == hi[1] && dst[(size_t)n + 1].z == hi[2];
  // the batch path's compaction: {x, y, z} only, exactly 12 bytes read per point, into a block of exactly 12 n bytes
  float* xyz = (float*)std::malloc((size_t)n * 12);
  apd::compact_xyz_host(xyz, raw, n, stride);
  ok = ok && std::memcmp(xyz, want.data(), (size_t)n * 12) == 0;
  std::free(xyz);
  std::free(raw);
  return ok;
}

int main() {
  int ok = 1;
  for (int64_t stride : {12, 16, 32})
    for (int64_t n = 1; n <= 70; n++) ok = ok && check_pack(n, stride, (unsigned)(n * 7 + stride));
  for (int64_t stride : {12, 16, 32}) ok = ok && check_pack(8192, stride, 99u);
  // the shared pool: three callers at once, 64 tasks each, every task exactly once
  int want_threads = 0;
  apd::HostPool* hp = apd::shared_host_pool(&want_threads);
  apd::HostPool own;
  if (!hp) own.start(3), hp = &own;  // (a one-core box: the shared pool is off; test a private one)
  std::vector<std::vector<int>> hits(3, std::vector<int>(64, 0));
  std::vector<std::thread> callers;
  for (int c = 0; c < 3; c++)
    callers.emplace_back([&, c]() {
      for (int rep = 0; rep < 50; rep++) {
        const std::function<void(int)> f = [&](int i) { hits[(size_t)c][(size_t)i]++; };
        hp->run(64, f);
      }
    });
  for (auto& t : callers) t.join();
  for (const auto& h : hits)
    for (int v : h) ok = ok && v == 50;
  std::printf("ok %d threads %d\n", ok, want_threads);
  return ok ? 0 : 1;
}
