// Loop-closure candidate batches sharded over the GPUs of one node from ONE C++ process (the Python / one
// process per GPU form is riv-slam_amd/sharded.py), pipelined: several batches are in flight on every device.
//
// Independent (source, target) pairs -- the candidates of loop_detector.cpp:222-236 / :404-423 -- are block-partitioned over
// the devices exactly like sharded.block_partition: pair p belongs to device p / ceil(P / D).  There is no collective on the
// data path.  The only exchange is ONE ncclAllGather (RCCL over xGMI) of the fixed-size result records, 96 bytes per pair, per
// batch, so that every device -- in particular the one next to the pose-graph owner -- ends up with all results.  The payload
// is a few KB: latency-bound, hence one call per batch.
//
// Execution: one persistent worker thread per device (created once, with the communicators, streams, handles and record
// buffers: nothing is allocated or spawned per call once the buffers have their size).  enqueue() validates the batch on the
// calling thread, hands every worker its block and returns a ticket; a worker registers its block's clouds, enqueues the
// registrations (apdgicp_batch_align_enqueue: nothing waits) and only then collects the OLDEST batch it still has in flight,
// issues that batch's all-gather on a stream of its own and moves on -- the gather of batch j is waited for when its slot is
// needed again or when the caller collects it, so ranks may drift by a few batches and the GPU always has `in_flight`
// batches queued.  Gauss-Newton batches run on `in_flight` handles with one pair group each (bench.py's schedule);
// Levenberg-Marquardt batches -- the reference's default -- run on ONE handle whose pair pool merges the batches in flight
// (include/apdgicp_hip.h), each in its own range of cloud slots.  A worker that fails -- bad cloud, HIP error, device error
// flag, even a failed allocation of its record buffers (preallocated fallback buffers) -- still takes part in the collective
// with a zeroed block and reports the error at collect(): no rank is left waiting in ncclAllGather (every rank calls the
// collectives in ticket order; when a rank truly cannot, the communicators are aborted and every collect() fails).
//
// Needs <hip/hip_runtime_api.h> and <rccl/rccl.h> (link amdhip64 + rccl + pthread).  Header-only.
#ifndef FAST_GICP_SHARDED_BATCH_HIP_HPP
#define FAST_GICP_SHARDED_BATCH_HIP_HPP

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
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
  int on_device = 0;  // != 0: xyz is device memory ON THE DEVICE OF THE RANK that owns the pairs referencing the cloud
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
  /// devices: HIP device indices, one rank each (rank r = devices[r]); in_flight: batches kept in flight per device (Gauss-Newton: 1 .. 8 handles; pooled Levenberg-Marquardt: up to the pool's lanes, 24)
  ShardedBatchAlignerHip(const apdgicp_params* params, const std::vector<int>& devices, int in_flight = 4)
      : devices_(devices), slots_(std::max(1, std::min(32, in_flight))) {
    const int D = (int)devices.size();
    apdgicp_params dflt;
    apdgicp_default_params(&dflt);
    params_ = params ? *params : dflt;
    pooled_ = params_.optimizer == APDGICP_OPT_LM;
    comm_init_.assign((size_t)D, nullptr);
    comms_ = std::vector<std::atomic<ncclComm_t>>((size_t)D);
    for (auto& c : comms_) c.store(nullptr);
    if (D == 0) {
      error_ = "no devices";
      return;
    }
    // Order matters for speed, not for correctness: the HIP runtime deals the streams of a process onto its hardware queues
    // in creation order, so the handles (whose streams carry the registrations) come first, RCCL's internal streams and the
    // gather stream of every rank after them (measured: docs/experiments.md).
    for (int r = 0; r < D; r++) ranks_.emplace_back(new Rank);
    for (int r = 0; r < D; r++) {
      Rank& k = *ranks_[(size_t)r];
      k.slots.resize((size_t)slots_);
      if (hipSetDevice(devices[(size_t)r]) != hipSuccess) {
        error_ = "hipSetDevice failed on device " + std::to_string(devices[(size_t)r]);
        return;
      }
      for (int h = 0; h < (pooled_ ? 1 : slots_); h++) {
        apdgicp_batch* b = nullptr;
        if (apdgicp_batch_create(&params_, devices[(size_t)r], nullptr, &b) != 0) {
          error_ = std::string("apdgicp_batch_create: ") + apdgicp_last_error();
          return;
        }
        if (r == 0 && h == 0) {  // (the library decides: APDGICP_LM_POOL=0 / brute-force search run LM one batch per handle)
          const int lanes = apdgicp_batch_is_pooled(b);
          pooled_ = pooled_ && lanes > 0;
          if (pooled_) slots_ = std::min(slots_, lanes);  // the pool holds that many batches
          else slots_ = std::min(slots_, 8);              // one handle per batch in flight
        }
        k.handles.push_back(b);
      }
      if (!pooled_)
        for (apdgicp_batch* b : k.handles) apdgicp_batch_set_pair_groups(b, slots_ > 1 ? 1 : 3);  // several handles share the GPU: one stream, larger launches each
    }
    if (ncclCommInitAll(comm_init_.data(), D, devices.data()) != ncclSuccess) {
      error_ = "ncclCommInitAll failed";
      comm_init_.assign((size_t)D, nullptr);
      return;
    }
    for (int r = 0; r < D; r++) comms_[(size_t)r].store(comm_init_[(size_t)r]);
    for (int r = 0; r < D; r++) {
      Rank& k = *ranks_[(size_t)r];
      if (hipSetDevice(devices[(size_t)r]) != hipSuccess || hipStreamCreateWithFlags(&k.gstream, hipStreamNonBlocking) != hipSuccess) {
        error_ = "stream creation failed on device " + std::to_string(devices[(size_t)r]);
        return;
      }
      for (Slot& s : k.slots)
        if (hipEventCreateWithFlags(&s.gathered, hipEventDisableTiming) != hipSuccess) {
          error_ = "event creation failed";
          return;
        }
      if (hipMalloc((void**)&k.fb_send, kFallbackBytes) != hipSuccess || hipMalloc((void**)&k.fb_recv, kFallbackBytes * (size_t)D) != hipSuccess ||
          hipMemset(k.fb_send, 0, kFallbackBytes) != hipSuccess) {
        error_ = "fallback gather buffers: allocation failed on device " + std::to_string(devices[(size_t)r]);
        return;
      }
    }
    for (int r = 0; r < D; r++) ranks_[(size_t)r]->th = std::thread([this, r]() { worker(r); });
  }
  ~ShardedBatchAlignerHip() {
    for (auto& kp : ranks_) {
      {
        std::lock_guard<std::mutex> g(kp->mu);
        kp->stop = true;
      }
      kp->cv.notify_all();
    }
    for (auto& kp : ranks_)
      if (kp->th.joinable()) kp->th.join();
    for (size_t r = 0; r < ranks_.size(); r++) {
      Rank& k = *ranks_[r];
      (void)hipSetDevice(devices_[r]);
      if (k.gstream) (void)hipStreamSynchronize(k.gstream);
      for (apdgicp_batch* b : k.handles) apdgicp_batch_destroy(b);
      for (Slot& s : k.slots) {
        if (s.send) (void)hipFree(s.send);
        if (s.recv) (void)hipFree(s.recv);
        if (s.stage) (void)hipHostFree(s.stage);
        if (s.gathered) (void)hipEventDestroy(s.gathered);
      }
      if (k.fb_send) (void)hipFree(k.fb_send);
      if (k.fb_recv) (void)hipFree(k.fb_recv);
      if (k.gstream) (void)hipStreamDestroy(k.gstream);
    }
    for (auto& a : comms_) {
      const ncclComm_t c = a.exchange(nullptr);
      if (c) ncclCommDestroy(c);
    }
  }
  ShardedBatchAlignerHip(const ShardedBatchAlignerHip&) = delete;
  ShardedBatchAlignerHip& operator=(const ShardedBatchAlignerHip&) = delete;
  bool ok() const { return error_.empty(); }
  const std::string& error() const { return error_; }
  int world() const { return (int)devices_.size(); }
  int in_flight() const { return slots_; }
  /// with ONE device the all-gather is a copy; RCCL is still called by default (the path the multi-GPU job takes)
  void set_gather_when_alone(bool on) { gather_when_alone_ = on; }
  /// test hook: the next batch's first rank to start behaves as if the allocation of its record buffers had failed
  void debug_fail_next_record_allocation() { fail_alloc_.store(true); }

  /// Hands the batch to the workers and returns its ticket (> 0) in *ticket; 0 or a negative apdgicp_status.  pairs[i] =
  /// (source cloud, target cloud, guess) with indices into `clouds`.  Nothing waits unless `in_flight` batches are already
  /// queued.  The cloud memory must stay valid until the batch has been collected (or in_flight later batches enqueued).
  int enqueue(const std::vector<ShardCloud>& clouds, const std::vector<apdgicp_pair>& pairs, uint64_t* ticket) {
    if (!ok() || !ticket) return APDGICP_ERR_INVALID_ARG;
    const int D = world();
    const int64_t P = (int64_t)pairs.size();
    // everything that can be wrong with the batch is found HERE, before a worker has entered a collective
    for (const apdgicp_pair& q : pairs)
      for (int32_t idx : {q.source_cloud, q.target_cloud})
        if (idx < 0 || (size_t)idx >= clouds.size() || !clouds[(size_t)idx].xyz || clouds[(size_t)idx].n <= 0 || clouds[(size_t)idx].stride_bytes < 12) {
          error_text_ = "pair references a missing or empty cloud";
          return APDGICP_ERR_INVALID_ARG;
        }
    const auto parts = block_partition(P, D);
    const int64_t per = P > 0 ? parts[0].second - parts[0].first : 0;
    const uint64_t seq = ++seq_;
    for (int r = 0; r < D; r++) {
      std::unique_ptr<Job> job(new Job);
      job->seq = seq, job->per = per, job->begin = parts[(size_t)r].first, job->end = parts[(size_t)r].second;
      // this rank's clouds: the ones its pairs reference, renumbered in order of first use
      std::vector<int> local(clouds.size(), -1);
      for (int64_t p = job->begin; p < job->end; p++) {
        apdgicp_pair q = pairs[(size_t)p];
        for (int32_t* idx : {&q.source_cloud, &q.target_cloud}) {
          if (local[(size_t)*idx] < 0) {
            local[(size_t)*idx] = (int)job->clouds.size();
            job->clouds.push_back(clouds[(size_t)*idx]);
          }
          *idx = local[(size_t)*idx];
        }
        job->pairs.push_back(q);
      }
      Rank& k = *ranks_[(size_t)r];
      std::unique_lock<std::mutex> g(k.mu);
      k.cv.wait(g, [&]() { return (int)k.queue.size() < slots_; });  // back-pressure
      k.queue.push_back(std::move(job));
      g.unlock();
      k.cv.notify_all();
    }
    *ticket = seq;
    return 0;
  }

  /// Waits for the batch of `ticket` (one of the last in_flight enqueued) on every rank and returns all P records in pair
  /// order, read from rank `root`'s copy of the gathered buffer.  0, or the first rank's negative apdgicp_status
  /// (last_error_text() says which and why).
  int collect(uint64_t ticket, std::vector<apdgicp_result>* results, int root = 0) {
    if (!ok() || !results || ticket == 0 || ticket > seq_ || root < 0 || root >= world()) return APDGICP_ERR_INVALID_ARG;
    const int D = world();
    int rc = 0;
    int64_t per = 0, total = 0;
    std::vector<std::pair<int64_t, int64_t>> ranges((size_t)D);
    for (int r = 0; r < D; r++) {
      Rank& k = *ranks_[(size_t)r];
      Slot& s = k.slots[(size_t)(ticket % (uint64_t)slots_)];
      std::unique_lock<std::mutex> g(k.mu);
      k.want = std::max(k.want, ticket);  // (an idle worker finishes what it has in flight up to here)
      k.cv.notify_all();
      k.cv.wait(g, [&]() { return s.issued_seq >= ticket || k.dead || k.aborted; });
      if (s.issued_seq != ticket) {
        error_text_ = k.aborted ? "rank " + std::to_string(r) + ": the communicators were aborted (a rank could not enter a record gather)"
                      : k.dead  ? "rank " + std::to_string(r) + ": worker stopped"
                                : "ticket is older than the batches in flight";
        return APDGICP_ERR_INVALID_ARG;
      }
      if (s.rc < 0 && rc == 0) rc = s.rc, error_text_ = "rank " + std::to_string(r) + ": " + s.msg;
      per = s.per, ranges[(size_t)r] = {s.begin, s.end}, total = std::max(total, s.end);
    }
    Rank& k = *ranks_[(size_t)root];
    Slot& s = k.slots[(size_t)(ticket % (uint64_t)slots_)];
    results->assign((size_t)total, apdgicp_result());
    if (hipSetDevice(devices_[(size_t)root]) != hipSuccess || hipEventSynchronize(s.gathered) != hipSuccess) return APDGICP_ERR_HIP;
    if (rc < 0 || total == 0) return rc;
    std::vector<apdgicp_result> all((size_t)(per * D));
    if (hipMemcpy(all.data(), s.recv, all.size() * sizeof(apdgicp_result), hipMemcpyDeviceToHost) != hipSuccess) return APDGICP_ERR_HIP;
    // every rank's `recv` holds D blocks of `per` records (short blocks zero-padded): trim them back into pair order
    for (int r = 0; r < D; r++)
      for (int64_t p = ranges[(size_t)r].first; p < ranges[(size_t)r].second; p++) (*results)[(size_t)p] = all[(size_t)(r * per + (p - ranges[(size_t)r].first))];
    return 0;
  }

  /// enqueue + collect
  int align(const std::vector<ShardCloud>& clouds, const std::vector<apdgicp_pair>& pairs, std::vector<apdgicp_result>* results, int root = 0) {
    uint64_t t = 0;
    const int rc = enqueue(clouds, pairs, &t);
    if (rc < 0) return rc;
    return collect(t, results, root);
  }
  const std::string& last_error_text() const { return error_text_; }
  /// device pointer of rank r's gathered buffer of batch `ticket` (world * per records).  Valid to read once collect(ticket, ..., root = r) has
  /// returned: collect waits for the gather of ITS root rank; the other ranks' copies may still be on their way
  const void* gathered_on(int r, uint64_t ticket) const { return ranks_[(size_t)r]->slots[(size_t)(ticket % (uint64_t)slots_)].recv; }

 private:
  static constexpr size_t kFallbackBytes = 64 << 10;  // 682 records per rank: any practical batch
  struct Job {
    uint64_t seq = 0;
    int64_t per = 0, begin = 0, end = 0;
    std::vector<ShardCloud> clouds;
    std::vector<apdgicp_pair> pairs;
  };
  struct Slot {
    char* send = nullptr;
    char* recv = nullptr;
    char* stage = nullptr;  // pinned host copy of the block's records on their way into `send`
    size_t send_cap = 0, recv_cap = 0;
    hipEvent_t gathered = nullptr;  // behind the slot's all-gather
    // worker-private until published under the rank's mutex
    uint64_t started_seq = 0, align_ticket = 0;
    bool busy = false;
    int start_rc = 0;
    std::string start_msg;
    int64_t s_per = 0, s_begin = 0, s_end = 0;
    // published: the batch whose all-gather has been ISSUED on this slot
    uint64_t issued_seq = 0;
    int rc = 0;
    std::string msg;
    int64_t per = 0, begin = 0, end = 0;
  };
  struct Rank {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::unique_ptr<Job>> queue;
    bool stop = false, dead = false;   // dead: the worker thread has ended
    bool aborted = false;              // abort_all ran: no collective will complete any more (the worker may still be running)
    uint64_t want = 0;
    hipStream_t gstream = nullptr;  // the all-gathers
    char* fb_send = nullptr;        // kFallbackBytes of zeros / world x kFallbackBytes: what a rank whose record buffers could not be
    char* fb_recv = nullptr;        // allocated gathers with instead, so that it still ENTERS the collective (finish)
    std::vector<apdgicp_batch*> handles;
    std::vector<Slot> slots;
    int cloud_cap = 64;  // pooled mode: cloud slots per batch in flight (grows; the descriptor table the handle uploads is as long as the highest slot)
    std::vector<const float*> ptrs;
    std::vector<int64_t> ns;
    double t_collect = 0, t_post = 0, t_start = 0, t_idle = 0;  // worker-thread milliseconds (SHARDED_TIMING=1 prints them)
    long n_fin = 0;
  };

  apdgicp_batch* handle_of(Rank& k, uint64_t seq) { return pooled_ ? k.handles[0] : k.handles[(size_t)(seq % (uint64_t)slots_)]; }

  // registers the block's clouds and enqueues its registrations on the slot's handle; errors are kept for the gather
  void start(int r, Rank& k, Slot& s, const Job& job) {
    s.busy = true, s.started_seq = job.seq, s.start_rc = 0, s.start_msg.clear(), s.align_ticket = 0;
    s.s_per = job.per, s.s_begin = job.begin, s.s_end = job.end;
    auto fail = [&](int code, const char* what) { s.start_rc = code, s.start_msg = std::string(what) + ": " + apdgicp_last_error(); };
    if (hipSetDevice(devices_[(size_t)r]) != hipSuccess) return fail(APDGICP_ERR_HIP, "hipSetDevice");
    const size_t rec = sizeof(apdgicp_result), D = (size_t)world();
    if (fail_alloc_.exchange(false)) {  // (debug_fail_next_record_allocation: what a failed hipMalloc below leaves behind)
      (void)hipEventSynchronize(s.gathered);
      if (s.send) (void)hipFree(s.send);
      if (s.recv) (void)hipFree(s.recv);
      if (s.stage) (void)hipHostFree(s.stage);
      s.send = s.recv = s.stage = nullptr, s.send_cap = s.recv_cap = 0;
      s.start_rc = APDGICP_ERR_HIP, s.start_msg = "hipMalloc: forced failure (debug_fail_next_record_allocation)";
      return;
    }
    if ((size_t)job.per * rec > s.send_cap || (size_t)job.per * rec * D > s.recv_cap) {  // (first batches only)
      if (hipEventSynchronize(s.gathered) != hipSuccess) return fail(APDGICP_ERR_HIP, "hipEventSynchronize");
      if (s.send) (void)hipFree(s.send);
      if (s.recv) (void)hipFree(s.recv);
      if (s.stage) (void)hipHostFree(s.stage);
      s.send = s.recv = s.stage = nullptr, s.send_cap = s.recv_cap = 0;
      const size_t cap = std::max<size_t>((size_t)job.per * rec * 2, 4096);
      if (hipMalloc((void**)&s.send, cap) != hipSuccess || hipMalloc((void**)&s.recv, cap * D) != hipSuccess ||
          hipHostMalloc((void**)&s.stage, cap, hipHostMallocDefault) != hipSuccess)
        return fail(APDGICP_ERR_HIP, "hipMalloc");
      s.send_cap = cap, s.recv_cap = cap * D;
    }
    if (job.pairs.empty()) return;
    apdgicp_batch* b = handle_of(k, job.seq);
    int base = 0;
    if (pooled_) {  // one handle, the batches in flight side by side in its cloud slots
      if ((int)job.clouds.size() > k.cloud_cap) {
        if (apdgicp_batch_synchronize(b) < 0) return fail(APDGICP_ERR_HIP, "apdgicp_batch_synchronize");
        while (k.cloud_cap < (int)job.clouds.size()) k.cloud_cap *= 2;
      }
      base = (int)(job.seq % (uint64_t)slots_) * k.cloud_cap;
    }
    const bool uniform = std::all_of(job.clouds.begin(), job.clouds.end(),
                                     [&](const ShardCloud& c) { return c.on_device == job.clouds[0].on_device && c.stride_bytes == job.clouds[0].stride_bytes; });
    int rc = 0;
    if (uniform) {  // the whole block in one call: device clouds one pack launch; host clouds (PCL) packed by the library's host threads, ONE copy
      k.ptrs.resize(job.clouds.size()), k.ns.resize(job.clouds.size());
      for (size_t c = 0; c < job.clouds.size(); c++) k.ptrs[c] = job.clouds[c].xyz, k.ns[c] = job.clouds[c].n;
      if ((rc = apdgicp_batch_set_clouds(b, base, (int)job.clouds.size(), k.ptrs.data(), k.ns.data(), job.clouds[0].stride_bytes, job.clouds[0].on_device ? 1 : 0)) < 0)
        return fail(rc, "apdgicp_batch_set_clouds");
    } else {
      for (size_t c = 0; c < job.clouds.size(); c++)
        if ((rc = apdgicp_batch_set_cloud(b, base + (int)c, job.clouds[c].xyz, job.clouds[c].n, job.clouds[c].stride_bytes, job.clouds[c].on_device)) < 0)
          return fail(rc, "apdgicp_batch_set_cloud");
    }
    std::vector<apdgicp_pair> mine = job.pairs;
    for (apdgicp_pair& q : mine) q.source_cloud += base, q.target_cloud += base;
    if ((rc = apdgicp_batch_align_enqueue(b, mine.data(), (int64_t)mine.size(), &s.align_ticket)) < 0) return fail(rc, "apdgicp_batch_align_enqueue");
  }

  // collects the slot's batch, issues its all-gather (always: a failed block contributes zeros) and publishes the outcome
  void finish(int r, Rank& k, Slot& s) {
    const auto tf0 = std::chrono::steady_clock::now();
    int rc = s.start_rc;
    std::string msg = s.start_msg;
    (void)hipSetDevice(devices_[(size_t)r]);
    const size_t rec = sizeof(apdgicp_result);
    const size_t mine = (size_t)(s.s_end - s.s_begin) * rec, block = (size_t)s.s_per * rec;
    bool gather_ok = s.send && s.recv && s.stage && block <= s.send_cap;
    // this rank's communicator, read ONCE: abort_all (any worker thread) takes it out of the table with an atomic exchange before it
    // aborts it -- a collective already inside RCCL with the old pointer is what ncclCommAbort exists to unblock
    const ncclComm_t comm = comms_[(size_t)r].load();
    if (aborted_.load()) {  // the communicators are gone (abort_all): nothing to enter, the batch fails
      if (rc == 0) rc = APDGICP_ERR_HIP, msg = "the record gather was aborted";
      gather_ok = false;
    }
    // The block's records come home with the batch's last poll (pinned memory): they go from the slot's own pinned staging
    // buffer into its send buffer by an asynchronous copy IN FRONT of the gather on the gather stream.  Nothing here waits
    // for the device -- on a busy GPU even a 3 KB copy kernel on a side stream waits a few hundred microseconds for a free
    // CU, and a host thread that waits for it does not serve the handle's pair pool meanwhile (measured: 0.3 ms per batch).
    if (gather_ok) memset(s.stage, 0, block);
    if (rc == 0 && s.align_ticket && gather_ok) {
      rc = apdgicp_batch_align_collect(handle_of(k, s.started_seq), s.align_ticket, nullptr, (apdgicp_result*)s.stage);
      if (rc < 0) {
        msg = std::string("apdgicp_batch_align_collect: ") + apdgicp_last_error();
        memset(s.stage, 0, block);
      }
    }
    const auto tf1 = std::chrono::steady_clock::now();
    (void)mine;
    if (gather_ok) {
      if (block && hipMemcpyAsync(s.send, s.stage, block, hipMemcpyHostToDevice, k.gstream) != hipSuccess) gather_ok = false;
      // every rank calls the collective, in ticket order; an empty or failed block contributes zeros
      if (block && world() == 1 && !gather_when_alone_) {  // one device: the "gather" is a copy
        if (hipMemcpyAsync(s.recv, s.send, block, hipMemcpyDeviceToDevice, k.gstream) != hipSuccess) gather_ok = false;
      } else if (block && (!comm || ncclAllGather(s.send, s.recv, block, ncclChar, comm, k.gstream) != ncclSuccess)) {
        gather_ok = false;
      }
    }
    if (!gather_ok && block) {
      // This rank's record buffers are missing (start() could not allocate them) or the copy into them failed -- but the other
      // ranks are in, or on their way into, this batch's ncclAllGather and a rank that skipped it would leave them there for
      // good, with every later collective of the communicator paired with the wrong batch.  So the collective IS called, on
      // the rank's preallocated fallback buffers (zeros out, a scratch area in), and the batch fails with this rank's error.
      // Only a block larger than the fallback leaves no way to take part: then the communicators are aborted and every
      // rank is marked dead, so that collect() returns an error everywhere instead of waiting.
      bool entered = (world() == 1 && !gather_when_alone_) || aborted_.load();
      if (!entered && block <= kFallbackBytes && k.fb_send && k.fb_recv)
        entered = comm && ncclAllGather(k.fb_send, k.fb_recv, block, ncclChar, comm, k.gstream) == ncclSuccess;
      if (!entered) abort_all("rank " + std::to_string(r) + ": could not enter the record gather");
    }
    if (!gather_ok && rc == 0) rc = APDGICP_ERR_HIP, msg = "record gather failed";
    (void)hipEventRecord(s.gathered, k.gstream);
    s.busy = false;
    {
      std::lock_guard<std::mutex> g(k.mu);
      s.issued_seq = s.started_seq, s.rc = rc, s.msg = msg, s.per = s.s_per, s.begin = s.s_begin, s.end = s.s_end;
    }
    k.cv.notify_all();
    k.t_collect += std::chrono::duration<double, std::milli>(tf1 - tf0).count();
    k.t_post += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tf1).count();
    k.n_fin++;
  }

  // last resort (see finish): no rank may wait for a collective that one of them cannot enter
  void abort_all(const std::string& why) {
    std::lock_guard<std::mutex> ga(abort_mu_);
    if (aborted_.exchange(true)) return;
    std::fprintf(stderr, "[ShardedBatchAlignerHip] %s: aborting the communicators\n", why.c_str());
    for (auto& a : comms_) {
      const ncclComm_t c = a.exchange(nullptr);
      if (c) (void)ncclCommAbort(c);
    }
    for (auto& kp : ranks_) {  // collect() stops waiting everywhere; the worker threads run on (their batches fail: aborted_) until shutdown
      {
        std::lock_guard<std::mutex> g(kp->mu);
        kp->aborted = true;
      }
      kp->cv.notify_all();
    }
  }

  void worker(int r) {
    Rank& k = *ranks_[(size_t)r];
    std::deque<uint64_t> inflight;  // batches started and not yet finished, oldest first (consecutive tickets: every rank gets every batch)
    for (;;) {
      std::unique_ptr<Job> job;
      {
        std::unique_lock<std::mutex> g(k.mu);
        auto ready = [&]() {
          // a new batch; or somebody waits for a batch that is only enqueued so far; or shutdown
          return !k.queue.empty() || k.stop || (!inflight.empty() && k.want >= inflight.front());
        };
        if (pooled_ && !inflight.empty()) {
          // the pair pool of the handle is served by the calls of this thread: while it has nothing else to do it keeps the
          // chunks of ticks enqueued ahead (apdgicp_batch_pump), instead of sleeping until the caller's next move
          while (!ready()) {
            g.unlock();
            (void)apdgicp_batch_pump(k.handles[0]);
            g.lock();
            if (!ready()) k.cv.wait_for(g, std::chrono::microseconds(50));
          }
        } else {
          k.cv.wait(g, ready);
        }
        if (!k.queue.empty()) {
          job = std::move(k.queue.front());
          k.queue.pop_front();
        } else if (k.stop && inflight.empty()) {
          break;
        }
      }
      k.cv.notify_all();  // (room in the queue)
      if (!job) {  // nothing new: the oldest batch in flight leaves
        finish(r, k, k.slots[(size_t)(inflight.front() % (uint64_t)slots_)]);
        inflight.pop_front();
        continue;
      }
      Slot& s = k.slots[(size_t)(job->seq % (uint64_t)slots_)];
      if ((int)inflight.size() == slots_) {  // the slot's previous batch is the oldest in flight
        finish(r, k, s);
        inflight.pop_front();
      }
      (void)hipSetDevice(devices_[(size_t)r]);
      (void)hipEventSynchronize(s.gathered);  // the gather that last read this slot's buffers
      const auto ts0 = std::chrono::steady_clock::now();
      start(r, k, s, *job);
      k.t_start += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ts0).count();
      inflight.push_back(job->seq);
    }
    if (std::getenv("SHARDED_TIMING") && k.n_fin)
      std::fprintf(stderr, "[sharded] rank %d: %ld batches; per batch: collect %.3f ms, gather issue %.3f ms, start %.3f ms\n", r, k.n_fin, k.t_collect / k.n_fin,
                   k.t_post / k.n_fin, k.t_start / k.n_fin);
    {
      std::lock_guard<std::mutex> g(k.mu);
      k.dead = true;
    }
    k.cv.notify_all();
  }

  std::vector<int> devices_;
  int slots_ = 4;
  bool pooled_ = false;
  bool gather_when_alone_ = true;  // world size 1: still go through RCCL (set_gather_when_alone(false): a plain copy)
  apdgicp_params params_;
  std::vector<std::unique_ptr<Rank>> ranks_;
  std::vector<ncclComm_t> comm_init_;             // as ncclCommInitAll filled it
  std::vector<std::atomic<ncclComm_t>> comms_;    // the live table: abort_all empties it while workers read it
  uint64_t seq_ = 0;
  std::string error_, error_text_;
  std::mutex abort_mu_;
  std::atomic<bool> aborted_{false};
  std::atomic<bool> fail_alloc_{false};
};

}  // namespace fast_gicp
#endif
