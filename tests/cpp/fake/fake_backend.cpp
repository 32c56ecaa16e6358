// TEST-ONLY fakes of the three call families riv-slam_amd/cpp/sharded_batch_hip.hpp is written against -- hip*, nccl*, apdgicp_batch_* -- linked
// INSTEAD of amdhip64 / rccl / libapdgicp_hip.so by tests/test_sanitizers.py, so that the aligner's own thread logic runs under
// -fsanitize=thread and -fsanitize=address,undefined on a box without a GPU.  Nothing here computes a registration: a "batch" returns
// records that are a fixed function of its pairs (so that sharded == single-handle can still be checked byte for byte), a cloud whose
// first coordinate is NaN fails its batch like the device error flag does, and ranks sleep a little, unevenly, so that they drift.
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "apdgicp_hip.h"

// ------------------------------------------------------------------ HIP
struct fakeStream {  // an in-order queue with an executor thread of its own: like a device queue, it makes progress without anybody waiting for it
  std::mutex mu;
  std::condition_variable cv_work, cv_done;
  std::deque<std::function<bool()>> pending;  // a closure returns false when it failed (aborted collective)
  uint64_t enqueued = 0, done = 0;
  bool failed = false, stop = false;
  std::thread th;
  fakeStream() {
    th = std::thread([this]() {
      std::unique_lock<std::mutex> g(mu);
      for (;;) {
        cv_work.wait(g, [&]() { return stop || !pending.empty(); });
        if (pending.empty()) return;  // (stop, everything done)
        std::function<bool()> f = std::move(pending.front());
        pending.pop_front();
        g.unlock();
        const bool ok = f();  // may block: a collective waits for the other ranks
        g.lock();
        failed = failed || !ok;
        done++;
        cv_done.notify_all();
      }
    });
  }
  ~fakeStream() {
    {
      std::lock_guard<std::mutex> g(mu);
      stop = true;
    }
    cv_work.notify_all();
    th.join();
  }
  bool drain_to(uint64_t upto) {
    std::unique_lock<std::mutex> g(mu);
    cv_done.wait(g, [&]() { return done >= upto; });
    return !failed;
  }
  void push(std::function<bool()> f) {
    {
      std::lock_guard<std::mutex> g(mu);
      pending.push_back(std::move(f));
      enqueued++;
    }
    cv_work.notify_all();
  }
};
struct fakeEvent {
  std::mutex mu;
  fakeStream* s = nullptr;
  uint64_t pos = 0;
};
static thread_local int t_device = 0;
extern "C" {
hipError_t hipSetDevice(int device) {
  t_device = device;
  return device >= 0 && device < 64 ? hipSuccess : hipErrorUnknown;
}
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) {
  *s = new fakeStream;
  return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t s) {
  uint64_t upto;
  {
    std::lock_guard<std::mutex> g(s->mu);
    upto = s->enqueued;
  }
  return s->drain_to(upto) ? hipSuccess : hipErrorUnknown;
}
hipError_t hipStreamDestroy(hipStream_t s) {
  delete s;
  return hipSuccess;
}
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) {
  *e = new fakeEvent;
  return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
  uint64_t pos;
  {
    std::lock_guard<std::mutex> g(s->mu);
    pos = s->enqueued;
  }
  std::lock_guard<std::mutex> g(e->mu);
  e->s = s, e->pos = pos;
  return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e) {
  fakeStream* s;
  uint64_t pos;
  {
    std::lock_guard<std::mutex> g(e->mu);
    s = e->s, pos = e->pos;
  }
  if (!s) return hipSuccess;  // never recorded
  (void)s->drain_to(pos);     // (like the real call: an event behind an aborted collective still "completes")
  return hipSuccess;
}
hipError_t hipEventDestroy(hipEvent_t e) {
  delete e;
  return hipSuccess;
}
static std::atomic<long> g_live_allocs{0};
hipError_t hipMalloc(void** p, size_t bytes) {
  *p = std::malloc(bytes ? bytes : 1);
  g_live_allocs++;
  return *p ? hipSuccess : hipErrorUnknown;
}
hipError_t hipFree(void* p) {
  if (p) g_live_allocs--;
  std::free(p);
  return hipSuccess;
}
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned) { return hipMalloc(p, bytes); }
hipError_t hipHostFree(void* p) { return hipFree(p); }
hipError_t hipMemset(void* p, int v, size_t bytes) {
  std::memset(p, v, bytes);
  return hipSuccess;
}
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind) {
  std::memcpy(dst, src, bytes);
  return hipSuccess;
}
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind, hipStream_t s) {
  // in stream order: behind whatever the stream still holds (a collective that has not completed)
  s->push([=]() {
    std::memcpy(dst, src, bytes);
    return true;
  });
  return hipSuccess;
}
long fake_live_allocations() { return g_live_allocs.load(); }
}

// ------------------------------------------------------------------ RCCL
struct fakeGroup {
  std::mutex mu;
  std::condition_variable cv;
  int n = 0;
  bool aborted = false;
  // op number -> the ranks' contributions (taken when a rank's stream REACHES the collective)
  std::map<uint64_t, std::vector<std::vector<char>>> posted;
  std::map<uint64_t, int> arrived, left;
};
struct fakeComm {
  std::shared_ptr<fakeGroup> g;
  int rank = 0;
  uint64_t next_op = 0;  // (one thread per communicator enqueues, as RCCL requires)
};
extern "C" {
ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int*) {
  auto g = std::make_shared<fakeGroup>();
  g->n = ndev;
  for (int r = 0; r < ndev; r++) {
    comms[r] = new fakeComm;
    comms[r]->g = g, comms[r]->rank = r;
  }
  return ncclSuccess;
}
ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t, ncclComm_t comm, hipStream_t stream) {
  std::shared_ptr<fakeGroup> g = comm->g;
  const int rank = comm->rank;
  const uint64_t op = comm->next_op++;
  stream->push([=]() {
    std::unique_lock<std::mutex> lk(g->mu);
    if (g->aborted) return false;
    auto& slots = g->posted[op];
    if (slots.empty()) slots.resize((size_t)g->n);
    slots[(size_t)rank].assign((const char*)send, (const char*)send + count);
    g->arrived[op]++;
    g->cv.notify_all();
    g->cv.wait(lk, [&]() { return g->aborted || g->arrived[op] == g->n; });
    if (g->aborted) return false;
    for (int r = 0; r < g->n; r++) std::memcpy((char*)recv + (size_t)r * count, g->posted[op][(size_t)r].data(), count);
    if (++g->left[op] == g->n) g->posted.erase(op), g->arrived.erase(op), g->left.erase(op);
    return true;
  });
  return ncclSuccess;
}
ncclResult_t ncclCommAbort(ncclComm_t comm) {
  {
    std::lock_guard<std::mutex> lk(comm->g->mu);
    comm->g->aborted = true;
  }
  comm->g->cv.notify_all();
  // (the real call frees the communicator; a collective already enqueued keeps the group alive through its shared_ptr)
  delete comm;
  return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  delete comm;
  return ncclSuccess;
}
}

// ------------------------------------------------------------------ apdgicp_batch_* (the subset the aligner calls)
static thread_local std::string t_err;
struct FakeCloud {
  int64_t n = 0;
  float first = 0.f;
};
struct FakeJob {
  uint64_t ticket = 0;
  std::vector<apdgicp_result> recs;
  int rc = 0;
};
struct apdgicp_batch {
  apdgicp_params prm;
  int device = 0;
  std::vector<FakeCloud> clouds;
  std::deque<FakeJob> jobs;
  uint64_t next_ticket = 0;
  std::atomic<int> in_call{0};  // the contract of the real library: one thread at a time per handle
  std::atomic<long> pumps{0};
};
struct CallGuard {
  apdgicp_batch* b;
  explicit CallGuard(apdgicp_batch* b_) : b(b_) {
    if (b->in_call.fetch_add(1) != 0) std::abort();  // two threads inside one handle: the aligner broke the contract
  }
  ~CallGuard() { b->in_call.fetch_sub(1); }
};
static apdgicp_result record_of(const apdgicp_pair& q, const std::vector<FakeCloud>& clouds) {  // a fixed function of the pair
  apdgicp_result r;
  std::memset(&r, 0, sizeof(r));
  std::memcpy(r.T, q.guess, sizeof(r.T));
  const FakeCloud& s = clouds[(size_t)q.source_cloud];
  const FakeCloud& t = clouds[(size_t)q.target_cloud];
  r.final_cost = (double)s.n * 0.5 + (double)t.first;
  r.converged = 1, r.iterations = (int32_t)(s.n % 7), r.n_linearize = r.iterations + 1, r.n_compute_error = r.iterations + 2, r.n_matched = (int32_t)std::min(s.n, t.n);
  return r;
}
extern "C" {
const char* apdgicp_last_error(void) { return t_err.c_str(); }
int apdgicp_device_count(int* count) {
  *count = 4;
  return 0;
}
void apdgicp_default_params(apdgicp_params* p) {
  std::memset(p, 0, sizeof(*p));
  p->k_correspondences = 20, p->max_iterations = 64, p->lm_max_iterations = 10, p->optimizer = APDGICP_OPT_LM, p->regularization = APDGICP_REG_PLANE;
  p->max_correspondence_distance = 3.4e38, p->transformation_epsilon = 5e-4, p->rotation_epsilon = 2e-3, p->lm_init_lambda_factor = 1e-9;
  p->distance_variance = 0.86, p->azimuth_variance_deg = 0.5, p->elevation_variance_deg = 1.0;
}
int apdgicp_batch_create(const apdgicp_params* p, int device, void*, apdgicp_batch** out) {
  *out = new apdgicp_batch;
  (*out)->prm = *p, (*out)->device = device;
  return 0;
}
int apdgicp_batch_destroy(apdgicp_batch* b) {
  delete b;
  return 0;
}
int apdgicp_batch_is_pooled(apdgicp_batch* b) { return b->prm.optimizer == APDGICP_OPT_LM ? 24 : 0; }
int apdgicp_batch_set_pair_groups(apdgicp_batch*, int) { return 0; }
int apdgicp_batch_synchronize(apdgicp_batch* b) {
  CallGuard g(b);
  return 0;
}
int apdgicp_batch_pump(apdgicp_batch* b) {
  CallGuard g(b);
  b->pumps++;
  return 0;
}
static int set_one(apdgicp_batch* b, int slot, const float* xyz, int64_t n) {
  if (slot < 0 || !xyz || n <= 0) {
    t_err = "cloud is null or empty";
    return APDGICP_ERR_INVALID_ARG;
  }
  if ((int)b->clouds.size() <= slot) b->clouds.resize((size_t)slot + 1);
  b->clouds[(size_t)slot].n = n, b->clouds[(size_t)slot].first = xyz[0];
  return 0;
}
int apdgicp_batch_set_cloud(apdgicp_batch* b, int index, const float* xyz, int64_t n, int64_t, int) {
  CallGuard g(b);
  return set_one(b, index, xyz, n);
}
int apdgicp_batch_add_cloud(apdgicp_batch* b, const float* xyz, int64_t n, int64_t, int) {
  CallGuard g(b);
  const int slot = (int)b->clouds.size();
  const int rc = set_one(b, slot, xyz, n);
  return rc < 0 ? rc : slot;
}
int apdgicp_batch_set_clouds(apdgicp_batch* b, int first, int count, const float* const* xyz, const int64_t* ns, int64_t, int) {
  CallGuard g(b);
  for (int c = 0; c < count; c++) {
    const int rc = set_one(b, first + c, xyz[c], ns[c]);
    if (rc < 0) return rc;
  }
  return 0;
}
int apdgicp_batch_align_enqueue(apdgicp_batch* b, const apdgicp_pair* pairs, int64_t n, uint64_t* ticket) {
  CallGuard g(b);
  FakeJob j;
  j.ticket = ++b->next_ticket;
  for (int64_t i = 0; i < n; i++) {
    const apdgicp_pair& q = pairs[i];
    if (q.source_cloud < 0 || q.target_cloud < 0 || (size_t)q.source_cloud >= b->clouds.size() || (size_t)q.target_cloud >= b->clouds.size()) {
      t_err = "pair references a cloud that is not set";
      return APDGICP_ERR_NO_INPUT;
    }
    // a non-finite point: the real library finds out on the device and fails the batch at collect
    if (std::isnan(b->clouds[(size_t)q.source_cloud].first) || std::isnan(b->clouds[(size_t)q.target_cloud].first)) j.rc = APDGICP_ERR_INTERNAL;
    j.recs.push_back(record_of(q, b->clouds));
  }
  b->jobs.push_back(std::move(j));
  while (b->jobs.size() > 32) b->jobs.pop_front();
  *ticket = b->next_ticket;
  return 0;
}
int apdgicp_batch_align_collect(apdgicp_batch* b, uint64_t ticket, void** d_out, apdgicp_result* host_out) {
  CallGuard g(b);
  std::this_thread::sleep_for(std::chrono::microseconds(50 + 40 * (b->device % 3)));  // ranks drift
  for (FakeJob& j : b->jobs)
    if (j.ticket == ticket) {
      if (j.rc < 0) {
        t_err = "device error flag 2: fewer than k neighbours at a finite distance (non-finite input points?)";
        return j.rc;
      }
      if (host_out) std::memcpy(host_out, j.recs.data(), j.recs.size() * sizeof(apdgicp_result));
      if (d_out) *d_out = j.recs.data();
      return 0;
    }
  t_err = "ticket is not one of the batches in flight";
  return APDGICP_ERR_INVALID_ARG;
}
int apdgicp_batch_align(apdgicp_batch* b, const apdgicp_pair* pairs, int64_t n, apdgicp_result* out) {
  uint64_t t = 0;
  const int rc = apdgicp_batch_align_enqueue(b, pairs, n, &t);
  return rc < 0 ? rc : apdgicp_batch_align_collect(b, t, nullptr, out);
}
}
