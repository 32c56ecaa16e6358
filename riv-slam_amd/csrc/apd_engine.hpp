// Host-side engine behind the C ABI: owns the device buffers of a set of clouds and a batch of
// registrations ("pairs"), launches the kernels of apd_kernels.hpp on one HIP stream and drives the
// per-pair GN/LM state machines until every pair is done.  No CPU compute path exists here: if HIP
// is unavailable every entry point fails with APDGICP_ERR_HIP.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <limits>
#include <memory>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <dlfcn.h>

#include "../../include/apdgicp_hip.h"
#include "apd_kernels.hpp"
#include "apd_hostpack.hpp"

namespace apd {

static_assert(sizeof(ResultRec) == sizeof(apdgicp_result), "ResultRec must mirror apdgicp_result");

inline thread_local std::string g_last_error;
inline int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}

#define APD_HIP(expr)                                                                                         \
  do {                                                                                                        \
    hipError_t e_ = (expr);                                                                                   \
    if (e_ != hipSuccess)                                                                                     \
      return fail(APDGICP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_) + " (" __FILE__ ":" + std::to_string(__LINE__) + ")"); \
  } while (0)
#define APD_TRY(expr)          \
  do {                         \
    int rc_ = (expr);          \
    if (rc_ < 0) return rc_;   \
  } while (0)

// roctx ranges over the host-side phases (pack / sort / k-NN covariances / optimiser ticks / poll), visible in
// `rocprofv3 --marker-trace`.  The marker library is looked up at run time -- the copy already in the process (a profiler
// preloads its own; torch ships one), else ROCm's -- so the library has no link-time dependency on a profiler component and
// the ranges cost two indirect calls when no tool listens, nothing at all when no marker library exists.
struct RoctxApi {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
};
inline const RoctxApi& roctx_api() {
  static const RoctxApi api = [] {
    RoctxApi a;
    void* sym = dlsym(RTLD_DEFAULT, "roctxRangePushA");
    void* lib = nullptr;
    if (!sym)
      for (const char* name : {"librocprofiler-sdk-roctx.so.1", "libroctx64.so.4", "libroctx64.so"})
        if ((lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
    a.push = (int (*)(const char*))(sym ? sym : lib ? dlsym(lib, "roctxRangePushA") : nullptr);
    a.pop = (int (*)())(sym ? dlsym(RTLD_DEFAULT, "roctxRangePop") : lib ? dlsym(lib, "roctxRangePop") : nullptr);
    if (!a.push || !a.pop) a.push = nullptr, a.pop = nullptr;
    return a;
  }();
  return api;
}
struct roctx_range {
  bool on;
  explicit roctx_range(const char* name) : on(roctx_api().push != nullptr) {
    if (on) roctx_api().push(name);
  }
  ~roctx_range() {
    if (on) roctx_api().pop();
  }
  roctx_range(const roctx_range&) = delete;
  roctx_range& operator=(const roctx_range&) = delete;
};

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  int ensure(size_t bytes) {
    if (bytes <= cap) return 0;
    if (p) {
      hipError_t e = hipFree(p);
      (void)e;
      p = nullptr;
      cap = 0;
    }
    const size_t want = (bytes + 255) & ~size_t(255);
    APD_HIP(hipMalloc(&p, want));
    cap = want;
    return 0;
  }
  void release() {
    if (p) {
      hipError_t e = hipFree(p);
      (void)e;
    }
    p = nullptr;
    cap = 0;
  }
  template <typename T>
  T* as() const { return (T*)p; }
};

inline int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}

// A small host->device table that is re-uploaded only when its contents change.  A changed table goes through one of two
// pinned staging buffers and an asynchronous copy on the caller's stream -- stream order already keeps it behind the
// kernels that still read the old contents -- and the host never waits for the GPU: a staging buffer is only rewritten
// after the event recorded behind ITS last copy has fired (normally long ago).
struct CachedTable {
  DevBuf dev;
  std::vector<char> host;  // last uploaded contents (for the comparison)
  char* pin[2] = {nullptr, nullptr};
  size_t pin_cap[2] = {0, 0};
  hipEvent_t ev[2] = {nullptr, nullptr};
  int cur = 0;
  ~CachedTable() {
    for (int i = 0; i < 2; i++) {
      if (ev[i]) (void)hipEventDestroy(ev[i]);
      if (pin[i]) (void)hipHostFree(pin[i]);
    }
  }
  int upload(const void* data, size_t bytes, hipStream_t stream) {
    if (dev.p && host.size() == bytes && (bytes == 0 || memcmp(host.data(), data, bytes) == 0)) return 0;
    host.assign((const char*)data, (const char*)data + bytes);
    if (std::max<size_t>(bytes, 16) > dev.cap) APD_HIP(hipStreamSynchronize(stream));  // the old device buffer is about to be freed
    APD_TRY(dev.ensure(std::max<size_t>(bytes, 16)));
    if (!bytes) return 0;
    cur ^= 1;
    if (!ev[cur]) APD_HIP(hipEventCreateWithFlags(&ev[cur], hipEventDisableTiming));
    else APD_HIP(hipEventSynchronize(ev[cur]));
    if (bytes > pin_cap[cur]) {
      if (pin[cur]) APD_HIP(hipHostFree(pin[cur]));
      pin[cur] = nullptr, pin_cap[cur] = 0;
      const size_t cap = std::max<size_t>(bytes * 2, 256);
      APD_HIP(hipHostMalloc((void**)&pin[cur], cap, hipHostMallocDefault));
      pin_cap[cur] = cap;
    }
    memcpy(pin[cur], data, bytes);
    APD_HIP(hipMemcpyAsync(dev.p, pin[cur], bytes, hipMemcpyHostToDevice, stream));
    APD_HIP(hipEventRecord(ev[cur], stream));
    return 0;
  }
  template <typename T>
  T* as() const { return (T*)dev.p; }
  CachedTable() = default;
  CachedTable(const CachedTable&) = delete;
  CachedTable& operator=(const CachedTable&) = delete;
  void swap(CachedTable& o) {
    std::swap(dev, o.dev), host.swap(o.host), std::swap(cur, o.cur);
    for (int i = 0; i < 2; i++) std::swap(pin[i], o.pin[i]), std::swap(pin_cap[i], o.pin_cap[i]), std::swap(ev[i], o.ev[i]);
  }
};

class Engine {
 public:
  struct Cloud {
    DevBuf opts;                     // caller's order, float4
    DevBuf pts, perm, cbox, gbox;    // Z-curve order + boxes (apd_sort.hpp), valid when `sorted`
    DevBuf cov;                      // sorted order
    int n = 0;
    bool sorted = false;
    bool cov_valid = false;
    uint64_t token = 0;
    int prep_lane = -1;  // the cloud stream (pool: cloud lane) the last operation on this cloud was enqueued on; -1: known to be complete
    // a host cloud of the tiled-sort size class stays in this pinned buffer until the sort has read it (TileJob::staged)
    char* stage_p = nullptr;
    char* stage_dev = nullptr;
    size_t stage_cap = 0;
    bool staged = false;            // set_cloud left the points there, the next sort reads them
    bool stage_pending = false;     // a sort that reads the buffer has been enqueued ...
    hipEvent_t stage_wait = nullptr;  // ... in front of this event (the poll event of the align behind it; not owned)
    volatile int* stage_seq_word = nullptr;  // ... and of the poll that posts stage_seq_val (or a later number) here
    int stage_seq_val = 0;
    void release_all() {
      opts.release(), pts.release(), perm.release(), cbox.release(), gbox.release(), cov.release();
      hipError_t e = hipSuccess;
      if (stage_p) e = hipHostFree(stage_p);
      (void)e;
      stage_p = stage_dev = nullptr, stage_cap = 0, staged = stage_pending = false, stage_wait = nullptr, stage_seq_word = nullptr;
    }
  };

  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  // the stream everything that prepares CLOUDS runs on (pack, sort, covariances, descriptor table): `stream` itself, except
  // while the handle pools Levenberg-Marquardt batches (pool_enter) -- there the preparation of the next batch runs beside the
  // optimiser ticks of the batches in flight, and a batch's first tick waits for the event behind its own preparation
  hipStream_t cstream = nullptr;
  apdgicp_params params;
  std::vector<Cloud> clouds;
  bool desc_dirty = true;

  // batch state
  int npairs = 0, nmax_src = 0, nmax_tgt = 0;
  std::vector<PairDesc> h_pairs;
  std::vector<CloudDesc> h_desc;  // host copy of the descriptor table (valid while !desc_dirty)
  std::vector<float> h_guesses;
  DevBuf d_state, d_results, d_errflag, d_probe, d_stage, d_T;
  DevBuf d_keys, d_box6, d_stats;
  CachedTable d_desc, d_pairs, d_guess, d_ids, d_packjobs, d_sortjobs, d_sortjobs_reg[3], d_tilejobs[3], d_active;
  DevBuf d_tkeys;          // sorted tiles of the clouds being sorted by the tiled path
  int n_stage_pending = 0;   // clouds whose pinned copy a sort has been enqueued for, no align behind it yet
  std::vector<int> h_active;  // pairs still running (rebuilt after every poll of an LM batch)
  DevBuf b_nnpart, b_corr, b_nnpt, b_nnaux, b_sqd, b_maha, b_blkpart, b_errpart;
  DevBuf b_blkcost, b_blkorder;  // launch order of a dense one-pair search's blocks (Work::blk_cost / blk_order)
  bool blk_order_on = env_int("APDGICP_NN_ORDER", 1) != 0;  // (0: launch order = index order, the cross-check)
  Work work{};
  int nn_S = 1;
  int nn_W = 0;  // waves per block of k_nn_pruned<1, W> sharing the same 64 points (APDGICP_NN_W = 1, 2, 4; 0: by load)
  int nn_sparse = 32;  // k_nn_compact: blocks with at most this many searching points take them one at a time (APDGICP_NN_SPARSE; 0: off)
  bool nn_pruned = true;   // exact bounding-box pruning on the Z-curve (APDGICP_NN_MODE=brute disables)
  // Neighbour keeping (nn_warm_start): a full search prunes with the squared radius r^2 (1 + skin_rel)^2 + skin_abs^2 instead of
  // r^2, which buys later iterations the right to keep the neighbour without searching while the point has moved by less than
  // the margin (APDGICP_NN_SKIN=0 disables: the cross-check of tests/test_hip_parity.py; margins measured in docs/experiments.md)
  bool nn_skin = true;
  static constexpr float nn_skin_rel = 0.25f, nn_skin_abs = 0.02f;
  float nn_cap = std::numeric_limits<float>::infinity();
  bool knn_pruned = true;  // same for the covariance k-NN (APDGICP_KNN_MODE=brute disables)
  static constexpr int kStatsBlocks = 8192;
  int stats_blocks = 0;    // APDGICP_STATS=2: Work::stats_blocks
  bool post_tick = false;  // this launch of k_error also writes the poll record (one-pair LM handles)
  CachedTable d_post;
  bool init_tick = false;
  bool fold_init = false, tickets_dirty = true;  // the first tick builds the pair state itself unless the arrival counters may be mid-count
  DevBuf b_ticket;
  // pair groups: the tick kernels of each group run on their own stream so that one group's short serial
  // kernels (k_lm_solve) and launch gaps overlap with the other group's wide ones
  std::vector<hipStream_t> gstreams;
  std::vector<hipEvent_t> gevents;
  hipEvent_t ev_main = nullptr;
  static constexpr int ngroups_cfg = 3;  // pair groups (HIP streams) of a batch handle that runs alone (measured: docs/experiments.md)
  static constexpr int kHostResults = 256;  // batches up to this size get their records with the status poll
  char* h_poll = nullptr;    // pinned: [records][status words + error flag] of the last poll
  char* h_poll_dev = nullptr;  // the same memory as the device addresses it
  int* h_status = nullptr;   // inside h_poll (or, for check_errflag, at its start)
  bool results_on_host = false;
  const ResultRec* host_results() const { return results_on_host ? (const ResultRec*)h_poll : nullptr; }
  double* h_probe = nullptr; // pinned, 48 doubles
  hipEvent_t ev_poll = nullptr;
  // Two aligns in flight (apdgicp_batch_align_enqueue / _collect): everything an align leaves behind for its caller -- the
  // record buffers on both sides, the poll event, the timed-launch events -- exists twice; swap_slots() makes the other set
  // current.  All other state is reused in stream order.
  struct AltSlot {
    DevBuf d_results;
    char* h_poll = nullptr;
    char* h_poll_dev = nullptr;
    int* h_status = nullptr;
    bool results_on_host = false;
    hipEvent_t ev_poll = nullptr;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> nn_events;
    size_t nn_events_used = 0;
    long long nn_pairs_acc = 0;
    bool pending = false;
    int pending_npairs = 0, pending_ticks = 0;
    int poll_seq = 0;
    bool pending_spin = false;
  } alt;
  bool pending = false;          // the current slot's final poll has been enqueued but not waited for
  int pending_npairs = 0, pending_ticks = 0;
  unsigned long long align_seq = 0;  // ticket of the current slot's align (the other slot holds align_seq - 1)

  // profiling of the dominant kernel (k_nn_partial)
  bool profile_nn = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> nn_events;
  size_t nn_events_used = 0;
  double last_nn_ms = 0;
  long long last_nn_launches = 0, last_nn_pairs = 0, nn_pairs_acc = 0;
  // events cost ~5 % of a batched step when every launch is bracketed, so only the launches of every
  // `profile_stride`-th tick are timed, with a phase that rotates from align to align (unbiased over ticks)
  int profile_stride = 10, profile_phase = 0, cur_tick = 0;
  int last_ticks = 0;
  int cur_active = 0;  // pairs still running in the tick being launched (0: all of them, e.g. the probes)
  const char* last_nn_kernel = "";  // the search kernel of the last launch (bench.py names it in its roofline object)

  // The opt-in for more than 64 KB of dynamic LDS is a property of the (function, device) pair, so every engine sets it
  // for its own device when it is created -- no process-wide "done once" flag that a second device or thread could trip over.
  static int set_kernel_attributes() {
    APD_HIP(hipFuncSetAttribute((const void*)k_sort_cloud_lds, hipFuncAttributeMaxDynamicSharedMemorySize, SORT_LDS_MAX_N * 8));
    APD_HIP(hipFuncSetAttribute((const void*)k_knn_cov<64>, hipFuncAttributeMaxDynamicSharedMemorySize, knn_lds_bytes_brute(64)));
    APD_HIP(hipFuncSetAttribute((const void*)k_knn_cov<128>, hipFuncAttributeMaxDynamicSharedMemorySize, knn_lds_bytes_brute(128)));
    APD_HIP(hipFuncSetAttribute((const void*)k_merge_tiles, hipFuncAttributeMaxDynamicSharedMemorySize, SORT_LDS_MAX_N * 8));
    return 0;
  }

  int init(const apdgicp_params* p, int dev, void* strm) {
    int count = 0;
    APD_HIP(hipGetDeviceCount(&count));
    if (dev < 0 || dev >= count) return fail(APDGICP_ERR_INVALID_ARG, "device index out of range");
    device = dev;
    APD_HIP(hipSetDevice(device));
    APD_TRY(set_kernel_attributes());
    if (strm) {
      stream = (hipStream_t)strm;
    } else {
      APD_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
      own_stream = true;
    }
    cstream = stream;
    APD_HIP(hipHostMalloc((void**)&h_poll, 65540 * sizeof(int) + kHostResults * sizeof(ResultRec), hipHostMallocDefault));
    memset(h_poll, 0, 65540 * sizeof(int) + kHostResults * sizeof(ResultRec));  // the sequence word the host spins on starts at 0
    h_status = (int*)h_poll;
    APD_HIP(hipHostGetDevicePointer((void**)&h_poll_dev, h_poll, 0));
    APD_HIP(hipHostMalloc((void**)&h_probe, 64 * sizeof(double), hipHostMallocDefault));
    APD_HIP(hipEventCreateWithFlags(&ev_poll, hipEventDisableTiming));
    APD_TRY(d_errflag.ensure(sizeof(int)));
    APD_HIP(hipMemsetAsync(d_errflag.p, 0, sizeof(int), stream));
    APD_TRY(d_probe.ensure(64 * sizeof(double)));
    APD_TRY(d_T.ensure(16 * sizeof(double)));
    if (env_int("APDGICP_STATS", 0)) {
      stats_blocks = env_int("APDGICP_STATS", 0) >= 2 ? kStatsBlocks : 0;  // 2: + the block timeline of the last k_nn_pruned launch
      APD_TRY(d_stats.ensure((16 + (size_t)3 * stats_blocks) * sizeof(unsigned long long)));
      APD_HIP(hipMemsetAsync(d_stats.p, 0, (16 + (size_t)3 * stats_blocks) * sizeof(unsigned long long), stream));
    }
    const char* m = getenv("APDGICP_NN_MODE");
    nn_pruned = !(m && std::string(m) == "brute");
    m = getenv("APDGICP_KNN_MODE");
    knn_pruned = !(m && std::string(m) == "brute");
    nn_skin = env_int("APDGICP_NN_SKIN", 1) != 0;
    // waves per search block: 0 = by load -- 8 / 4 while the batch is small enough to leave the GPU mostly empty (a single
    // registration: 35 -> 27 us per iteration), 2 otherwise (more lose there: every wave repeats the bounds and candidate tests)
    // blocks of k_nn_compact with at most this many points still searching take the point-serial path (measured on the bench step,
    // four alternations: off 0.7255, 24 / 32 / 40 / 48: 0.7136 / 0.7080 / 0.7084 / 0.7105 ms; one wave gathers the results: <= 64)
    nn_sparse = std::min(64, std::max(0, env_int("APDGICP_NN_SPARSE", 32)));
    nn_W = env_int("APDGICP_NN_W", 0);
    if (nn_W != 1 && nn_W != 2 && nn_W != 4 && nn_W != 8) nn_W = 0;
    APD_HIP(hipEventCreateWithFlags(&ev_main, hipEventDisableTiming));
    return set_params(p);
  }
  // The streams of pair groups 1 .. n-1 (group 0 uses the main stream) are created when a batch first needs them, not with the
  // engine: the runtime deals streams onto its hardware queues in creation order, and the never-used group streams of
  // one-group handles pushed the main streams of four handles onto colliding queues (bench: 1.06 -> 0.93 ms per step with
  // four handles once the main streams sat on queues of their own).
  int ensure_group_streams(int ng) {
    while ((int)gstreams.size() + 1 < ng) {
      hipStream_t st_;
      hipEvent_t ev_;
      APD_HIP(hipStreamCreateWithFlags(&st_, hipStreamNonBlocking));
      APD_HIP(hipEventCreateWithFlags(&ev_, hipEventDisableTiming));
      gstreams.push_back(st_);
      gevents.push_back(ev_);
    }
    return 0;
  }

  ~Engine() {
    hipError_t e;
    e = hipSetDevice(device);
    if (stream) e = hipStreamSynchronize(stream);
    for (auto& c : clouds) c.release_all();
    for (CachedTable* t : {&d_desc, &d_pairs, &d_guess, &d_ids, &d_packjobs, &d_sortjobs, &d_sortjobs_reg[0], &d_sortjobs_reg[1], &d_sortjobs_reg[2], &d_tilejobs[0], &d_tilejobs[1], &d_tilejobs[2], &d_active, &d_post}) t->dev.release();
    d_tkeys.release();
    for (DevBuf* b : {&d_state, &d_results, &d_errflag, &d_probe, &d_stage, &d_T, &d_trace,
                      &d_keys, &d_box6, &d_stats, &b_ticket, &b_blkcost, &b_blkorder, &b_nnpart, &b_corr, &b_nnpt, &b_nnaux, &b_sqd, &b_maha, &b_blkpart, &b_errpart})
      b->release();
    if (bulk_host) e = hipHostFree(bulk_host);
    if (bulk_ev) e = hipEventDestroy(bulk_ev);
    bulk_dev.release();
    if (h_poll) e = hipHostFree(h_poll);
    if (alt.h_poll) e = hipHostFree(alt.h_poll);
    if (alt.ev_poll) e = hipEventDestroy(alt.ev_poll);
    alt.d_results.release();
    for (auto& pr : alt.nn_events) e = hipEventDestroy(pr.first), e = hipEventDestroy(pr.second);
    if (h_probe) e = hipHostFree(h_probe);
    if (ev_poll) e = hipEventDestroy(ev_poll);
    if (ev_main) e = hipEventDestroy(ev_main);
    if (ev_producer) e = hipEventDestroy(ev_producer);
    if (pool.cstream) e = hipStreamSynchronize(pool.cstream);
    for (Pool::List& li : pool.L)
      for (hipEvent_t ev_ : li.ev)
        if (ev_) e = hipEventDestroy(ev_);
    for (Pool::Timed& c : pool.timed) e = hipEventDestroy(c.e0), e = hipEventDestroy(c.e1);
    for (PoolJob& j : pool.jobs) {
      if (j.ev_pro) e = hipEventDestroy(j.ev_pro);
      if (j.pin) e = hipHostFree(j.pin);
      j.d_recs.release();
    }
    for (DevBuf* b : {&pool.state, &pool.pairs, &pool.guess, &pool.results, &pool.ticket, &pool.nnpart, &pool.corr,
                      &pool.nnpt, &pool.nnaux, &pool.sqd, &pool.maha, &pool.blkpart, &pool.errpart})
      b->release();
    for (Pool::List& li : pool.L) li.active.release(), li.nactive.release();
    if (pool.cstream2) e = hipStreamSynchronize(pool.cstream2), e = hipStreamDestroy(pool.cstream2);
    if (pool.ev_cross) e = hipEventDestroy(pool.ev_cross);
    for (CachedTable* t : {&lane2.desc, &lane2.ids, &lane2.packjobs, &lane2.sortjobs, &lane2.tilejobs[0], &lane2.tilejobs[1], &lane2.tilejobs[2]}) t->dev.release();
    lane2.tkeys.release(), lane2.bulk_dev.release();
    if (lane2.bulk_host) e = hipHostFree(lane2.bulk_host);
    if (lane2.bulk_ev) e = hipEventDestroy(lane2.bulk_ev);
    if (pool.host) e = hipHostFree(pool.host);
    if (pool.cstream) e = hipStreamDestroy(pool.cstream);
    for (HostStage& hs : h_stage) {
      if (hs.ev) e = hipEventDestroy(hs.ev);
      if (hs.p) e = hipHostFree(hs.p);
    }
    for (auto st_ : gstreams) e = hipStreamSynchronize(st_), e = hipStreamDestroy(st_);
    for (auto ev_ : gevents) e = hipEventDestroy(ev_);
    for (auto& pr : nn_events) e = hipEventDestroy(pr.first), e = hipEventDestroy(pr.second);
    if (own_stream && stream) e = hipStreamDestroy(stream);
    (void)e;
  }

  int set_params(const apdgicp_params* p) {
    if (!p) return fail(APDGICP_ERR_INVALID_ARG, "params is null");
    if (p->k_correspondences < 1) return fail(APDGICP_ERR_INVALID_ARG, "k_correspondences must be >= 1");
    if (p->regularization < 0 || p->regularization > 4) return fail(APDGICP_ERR_UNSUPPORTED, "unknown regularization method");
    if (p->optimizer != APDGICP_OPT_LM && p->optimizer != APDGICP_OPT_GN) return fail(APDGICP_ERR_INVALID_ARG, "unknown optimizer");
    if (p->flags & ~(APDGICP_FLAG_PLAIN_GICP | APDGICP_FLAG_XF_LINEAR_CHAIN | APDGICP_FLAG_FP32_POINT_MATH | APDGICP_FLAG_ALGEBRAIC_APD)) return fail(APDGICP_ERR_INVALID_ARG, "unknown bit in params.flags");
    if ((p->flags & APDGICP_FLAG_FP32_POINT_MATH) && (p->flags & APDGICP_FLAG_ALGEBRAIC_APD))
      return fail(APDGICP_ERR_INVALID_ARG, "APDGICP_FLAG_FP32_POINT_MATH and APDGICP_FLAG_ALGEBRAIC_APD are exclusive");
    if (pool.on) APD_TRY(pool_drain());  // the batches in flight finish with the parameters they were enqueued with
    const bool cov_change = clouds.size() && (p->k_correspondences != params.k_correspondences || p->regularization != params.regularization);
    params = *p;
    if (cov_change)
      for (auto& c : clouds) c.cov_valid = false;  // covariances depend on k and the regularisation mode
    return 0;
  }

  Consts consts() const {
    Consts c;
    c.k = params.k_correspondences;
    c.max_iterations = params.max_iterations;
    c.lm_max_iterations = params.lm_max_iterations;
    c.optimizer = params.optimizer;
    c.regularization = params.regularization;
    c.plain_gicp = (params.flags & APDGICP_FLAG_PLAIN_GICP) ? 1 : 0;
    c.fp32_point = (params.flags & APDGICP_FLAG_FP32_POINT_MATH) ? 1 : 0;
    c.thr2 = params.max_correspondence_distance * params.max_correspondence_distance;
    c.trans_eps = params.transformation_epsilon;
    c.rot_eps = params.rotation_epsilon;
    c.lm_init_lambda_factor = params.lm_init_lambda_factor;
    c.dist_var_400 = params.distance_variance / 400;
    c.sin_az = std::sin(params.azimuth_variance_deg / 180 * M_PI);   // A:170
    c.sin_el = std::sin(params.elevation_variance_deg / 180 * M_PI); // A:171
    return c;
  }

  // Device-resident inputs are produced on the CALLER's stream; the engine's streams are non-blocking, so nothing orders
  // the two unless asked: everything queued on `producer` so far will have finished before anything this engine enqueues
  // from now on starts (an event on the producer, a wait on the engine's stream -- the host does not block).
  hipEvent_t ev_producer = nullptr;
  int wait_producer(void* producer) {
    APD_HIP(hipSetDevice(device));
    if (!ev_producer) APD_HIP(hipEventCreateWithFlags(&ev_producer, hipEventDisableTiming));
    APD_HIP(hipEventRecord(ev_producer, (hipStream_t)producer));
    APD_HIP(hipStreamWaitEvent(stream, ev_producer, 0));
    if (cstream != stream) APD_HIP(hipStreamWaitEvent(cstream, ev_producer, 0));
    if (pool.on && pool.cstream2) APD_HIP(hipStreamWaitEvent(pool.cstream2, ev_producer, 0));
    return 0;
  }

  struct HostStage {
    char* p = nullptr;
    size_t cap = 0;
    hipEvent_t ev = nullptr;
  } h_stage[2];
  int h_stage_cur = 0;

  // ------------------------------------------------------------------ clouds
  int set_cloud(int slot, const float* xyz, int64_t n, int64_t stride_bytes, int on_device, uint64_t token) {
    if (slot < 0) return fail(APDGICP_ERR_INVALID_ARG, "bad cloud slot");
    if (!xyz || n <= 0) return fail(APDGICP_ERR_INVALID_ARG, "cloud is null or empty");
    if (n > (1 << 30)) return fail(APDGICP_ERR_INVALID_ARG, "cloud too large");
    if (stride_bytes < 12 || (stride_bytes & 3)) return fail(APDGICP_ERR_INVALID_ARG, "stride_bytes must be a multiple of 4 and >= 12");
    APD_HIP(hipSetDevice(device));
    APD_TRY(pool_release_clouds(slot, 1));
    roctx_range rr("apdgicp:pack");
    if ((int)clouds.size() <= slot) clouds.resize(slot + 1);
    Cloud& c = clouds[slot];
    lane_touch(c);
    APD_TRY(lane_order());
    // the previous contents may still be in use by queued kernels on this stream; stream order protects us
    if ((size_t)n * 16 > c.opts.cap) {
      APD_HIP(hipStreamSynchronize(cstream));
      if (pool.on && pool.cstream2) APD_HIP(hipStreamSynchronize(pool.cstream2));
    }
    APD_TRY(c.opts.ensure((size_t)n * 16));
    const char* raw = (const char*)xyz;
    c.staged = false;
    if (!on_device && n > 2048 && n <= SORT_LDS_MAX_N) {
      // scan-sized host clouds: packed into the slot's own pinned buffer, bounding box behind the points, and the sort reads
      // them from there (k_sort_tiles).  Nothing is enqueued here, not even an event: the buffer is rewritten after the poll
      // of the align that followed the sort (an event between the sort's launches costs 6 us of the frame), and a cloud
      // replaced with no align in between waits for the stream
      if (c.stage_pending) {
        // (the host has usually SEEN that poll already -- it spun on the word; asking the event costs 6 us when the kernel
        // that posted the word has not retired yet)
        const bool seen = c.stage_seq_word && (int)(*c.stage_seq_word - c.stage_seq_val) >= 0;
        if (seen) std::atomic_thread_fence(std::memory_order_acquire);
        else if (c.stage_wait) APD_HIP(hipEventSynchronize(c.stage_wait));
        else APD_HIP(hipStreamSynchronize(cstream));
        c.stage_pending = false, c.stage_wait = nullptr, c.stage_seq_word = nullptr;
      }
      const size_t need = ((size_t)n + 2) * 16;
      if (need > c.stage_cap) {
        if (c.stage_p) APD_HIP(hipHostFree(c.stage_p));
        c.stage_p = c.stage_dev = nullptr, c.stage_cap = 0;
        const size_t cap = std::max<size_t>(need * 5 / 4, 1 << 16);
        APD_HIP(hipHostMalloc((void**)&c.stage_p, cap, hipHostMallocDefault));
        APD_HIP(hipHostGetDevicePointer((void**)&c.stage_dev, c.stage_p, 0));
        c.stage_cap = cap;
      }
      pack_staged_host((HostF4*)c.stage_p, raw, n, stride_bytes);
      c.staged = true;
    } else if (!on_device) {
      // Host clouds (the odometry nodelet hands over pcl::PointXYZI, 32 bytes a point): the three coordinates are packed into
      // {x, y, z, 1} on the host, straight into one of two pinned staging buffers, and ONE asynchronous copy puts them where
      // the pack kernel would have.  The caller's buffer is free when this returns, and nothing waits for the GPU: a staging
      // buffer is only rewritten after the event behind its last copy has fired (two calls ago).
      HostStage& hs = h_stage[h_stage_cur ^= 1];
      if (!hs.ev) APD_HIP(hipEventCreateWithFlags(&hs.ev, hipEventDisableTiming));
      else APD_HIP(hipEventSynchronize(hs.ev));
      if ((size_t)n * 16 > hs.cap) {
        if (hs.p) APD_HIP(hipHostFree(hs.p));
        hs.p = nullptr, hs.cap = 0;
        const size_t cap = std::max<size_t>((size_t)n * 16 * 5 / 4, 1 << 16);
        APD_HIP(hipHostMalloc((void**)&hs.p, cap, hipHostMallocDefault));
        hs.cap = cap;
      }
      float4* dst = (float4*)hs.p;
      for (int64_t q = 0; q < n; q++) {
        const float* sp = (const float*)(raw + q * stride_bytes);
        dst[q] = make_float4(sp[0], sp[1], sp[2], 1.0f);
      }
      APD_HIP(hipMemcpyAsync(c.opts.p, hs.p, (size_t)n * 16, hipMemcpyHostToDevice, cstream));
      APD_HIP(hipEventRecord(hs.ev, cstream));
    } else {
      hipLaunchKernelGGL(k_pack_points, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, cstream, raw, (long long)stride_bytes, (int)n, c.opts.as<float4>());
      APD_HIP(hipGetLastError());
    }
    c.n = (int)n;
    c.sorted = false;  // Z-curve sort is deferred so that a batch of clouds is sorted by ONE launch
    c.cov_valid = false;
    c.token = token;
    desc_dirty = true;
    return 0;
  }

  // Many host clouds at once (a batch of loop-closure candidates handed over as PCL clouds): packed to {x, y, z, 1} by the host
  // pool into ONE pinned region, ONE asynchronous copy (the DMA engine, beside the kernels of the other steps in flight) and the
  // pack launch of set_clouds_device.  The single-cloud path lets the sort read a scan-sized cloud straight from pinned memory --
  // right for one frame (no copy in front of the first kernel), wrong for 64 clouds: the sort's blocks then sit on their compute
  // units waiting for PCIe (bench.py --host-clouds: 1.07 ms per step that way, 0.96 with the packing vectorised, see docs/experiments.md).
  char* bulk_host = nullptr;
  size_t bulk_cap = 0;
  hipEvent_t bulk_ev = nullptr;
  DevBuf bulk_dev;
  int set_clouds_host(int first, int count, const float* const* xyz, const int64_t* ns, int64_t stride_bytes) {
    if (first < 0 || count <= 0 || first + (long long)count > (1 << 24)) return fail(APDGICP_ERR_INVALID_ARG, "bad cloud range");
    if (!xyz || !ns) return fail(APDGICP_ERR_INVALID_ARG, "null argument");
    if (stride_bytes < 12 || (stride_bytes & 3)) return fail(APDGICP_ERR_INVALID_ARG, "stride_bytes must be a multiple of 4 and >= 12");
    int64_t pts = 0;
    for (int q = 0; q < count; q++) {
      if (!xyz[q] || ns[q] <= 0 || ns[q] > (1 << 30)) return fail(APDGICP_ERR_INVALID_ARG, "cloud is null, empty or too large");
      pts += ns[q];
    }
    if (count < 4 || pts < 32768) {
      for (int q = 0; q < count; q++) APD_TRY(set_cloud(first + q, xyz[q], ns[q], stride_bytes, 0, 0));
      return 0;
    }
    APD_HIP(hipSetDevice(device));
    // 12 bytes a point, every cloud 16-byte aligned in the region
    std::vector<size_t> offs(count);
    size_t need = 0;
    for (int q = 0; q < count; q++) offs[q] = need, need += ((size_t)ns[q] * 12 + 15) & ~(size_t)15;
    if (!bulk_ev) APD_HIP(hipEventCreateWithFlags(&bulk_ev, hipEventDisableTiming));
    else APD_HIP(hipEventSynchronize(bulk_ev));  // the previous copy out of the pinned region (long done: a step ago on this handle)
    if (need > bulk_cap) {
      if (bulk_host) APD_HIP(hipHostFree(bulk_host));
      bulk_host = nullptr, bulk_cap = 0;
      APD_HIP(hipHostMalloc((void**)&bulk_host, need * 5 / 4, hipHostMallocDefault));
      bulk_cap = need * 5 / 4;
    }
    if (need > bulk_dev.cap) APD_HIP(hipStreamSynchronize(cstream));  // (the pack launch of the previous call may still read it)
    APD_TRY(bulk_dev.ensure(need));
    std::vector<const float*> dptr(count);
    for (int q = 0; q < count; q++) dptr[q] = (const float*)((char*)bulk_dev.p + offs[q]);
    int want = 1;
    HostPool* hp = shared_host_pool(&want);
    const std::function<void(int)> f = [&](int i) { compact_xyz_host((float*)(bulk_host + offs[i]), (const char*)xyz[i], ns[i], stride_bytes); };
    if (hp) hp->run(count, f);
    else
      for (int q = 0; q < count; q++) f(q);
    APD_HIP(hipMemcpyAsync(bulk_dev.p, bulk_host, need, hipMemcpyHostToDevice, cstream));
    APD_HIP(hipEventRecord(bulk_ev, cstream));
    return set_clouds_device(first, count, dptr.data(), ns, 12);
  }

  // many device-resident clouds at once: one pack launch instead of one per cloud
  int set_clouds_device(int first, int count, const float* const* xyz, const int64_t* ns, int64_t stride_bytes) {
    if (first < 0 || count <= 0 || first + (long long)count > (1 << 24)) return fail(APDGICP_ERR_INVALID_ARG, "bad cloud range");
    if (!xyz || !ns) return fail(APDGICP_ERR_INVALID_ARG, "null argument");
    if (stride_bytes < 12 || (stride_bytes & 3)) return fail(APDGICP_ERR_INVALID_ARG, "stride_bytes must be a multiple of 4 and >= 12");
    APD_HIP(hipSetDevice(device));
    APD_TRY(pool_release_clouds(first, count));
    roctx_range rr("apdgicp:pack");
    if ((int)clouds.size() < first + count) clouds.resize(first + count);
    bool grew = false;
    int nmax = 0;
    for (int q = 0; q < count; q++) {
      if (!xyz[q] || ns[q] <= 0 || ns[q] > (1 << 30)) return fail(APDGICP_ERR_INVALID_ARG, "cloud is null, empty or too large");
      grew |= (size_t)ns[q] * 16 > clouds[first + q].opts.cap;
      nmax = std::max<int>(nmax, (int)ns[q]);
      lane_touch(clouds[first + q]);
    }
    APD_TRY(lane_order());
    if (grew) {  // a buffer about to be re-allocated may still be in use
      APD_HIP(hipStreamSynchronize(cstream));
      if (pool.on && pool.cstream2) APD_HIP(hipStreamSynchronize(pool.cstream2));
    }
    std::vector<PackJob> jobs(count);
    for (int q = 0; q < count; q++) {
      Cloud& c = clouds[first + q];
      APD_TRY(c.opts.ensure((size_t)ns[q] * 16));
      jobs[q] = PackJob{(const char*)xyz[q], c.opts.as<float4>(), (long long)stride_bytes, (int)ns[q], 0};
      c.n = (int)ns[q];
      c.sorted = false;
      c.cov_valid = false;
      c.token = 0;
      c.staged = false;  // (a host cloud set before and never sorted: its pinned copy is void now)
    }
    APD_TRY(d_packjobs.upload(jobs.data(), jobs.size() * sizeof(PackJob), cstream));
    hipLaunchKernelGGL(k_pack_points_multi, dim3((unsigned)((nmax + 255) / 256), (unsigned)count), dim3(256), 0, cstream, d_packjobs.as<PackJob>());
    APD_HIP(hipGetLastError());
    // device-resident inputs: the caller's buffers are free again once the stream has passed this point
    // (apdgicp_batch_synchronize), no host-side wait here
    desc_dirty = true;
    return 0;
  }

  void clear_cloud(int slot) {
    if (slot < (int)clouds.size()) {
      clouds[slot].n = 0;
      clouds[slot].staged = false;
      clouds[slot].sorted = false;
      clouds[slot].cov_valid = false;
      clouds[slot].token = 0;
      desc_dirty = true;
    }
  }

  // Z-curve sort + chunk/group boxes of every cloud that was (re)set since the last call
  int sort_clouds() {
    std::vector<SortJob> small;
    std::vector<TileJob> tilejobs[3];
    size_t tkeys_bytes = 0;
    std::vector<int> large;
    int np2max = 1;
    bool grew = false, any = false;
    for (size_t i = 0; i < clouds.size(); i++) {
      Cloud& c = clouds[i];
      if (c.n <= 0 || c.sorted) continue;
      any = true;
      const size_t n = c.n, nch = (n + 15) / 16, ngr = (n + kGroupPts - 1) / kGroupPts, nsup = (ngr + kSuperGroups - 1) / kSuperGroups;
      grew |= n * 16 > c.pts.cap || n * 4 > c.perm.cap || nch * sizeof(Box) > c.cbox.cap || (ngr + nsup) * sizeof(Box) > c.gbox.cap;
      lane_touch(c);
    }
    if (!any) return 0;
    APD_TRY(lane_order());
    if (grew) {  // old buffers may still be read by queued kernels
      APD_HIP(hipStreamSynchronize(cstream));
      if (pool.on && pool.cstream2) APD_HIP(hipStreamSynchronize(pool.cstream2));
    }
    roctx_range rr("apdgicp:sort");
    for (size_t i = 0; i < clouds.size(); i++) {
      Cloud& c = clouds[i];
      if (c.n <= 0 || c.sorted) continue;
      const size_t n = c.n, nch = (n + 15) / 16, ngr = (n + kGroupPts - 1) / kGroupPts, nsup = (ngr + kSuperGroups - 1) / kSuperGroups;
      APD_TRY(c.pts.ensure(n * 16));
      APD_TRY(c.perm.ensure(n * 4));
      APD_TRY(c.cbox.ensure(nch * sizeof(Box)));
      APD_TRY(c.gbox.ensure((ngr + nsup) * sizeof(Box)));  // group boxes, then (large clouds) the super boxes
      if (c.n <= SORT_LDS_MAX_N) {
        SortJob j;
        j.pts = c.opts.as<float4>(), j.spts = c.pts.as<float4>(), j.perm = c.perm.as<int>();
        j.cbox = c.cbox.as<Box>(), j.gbox = c.gbox.as<Box>(), j.n = c.n, j.pad_ = 0;
        if (c.n > 2048) {  // four tiles of 1024 / 2048 / 4096 keys
          const int cls = c.n <= 4096 ? 0 : c.n <= 8192 ? 1 : 2;
          TileJob tjb;
          tjb.job = j, tjb.keys = (unsigned long long*)tkeys_bytes /* offset, fixed up below */, tjb.nt = 1024 << cls, tjb.pad_ = 0;
          tjb.staged = c.staged ? (const float4*)c.stage_dev : nullptr;
          tkeys_bytes += (size_t)4 * tjb.nt * 8;
          tilejobs[cls].push_back(tjb);
        } else {
          small.push_back(j);
          int np2 = 1;
          while (np2 < c.n) np2 <<= 1;
          np2max = std::max(np2max, np2);
        }
      } else {
        large.push_back((int)i);
      }
      c.sorted = true;
      desc_dirty = true;
    }
    if (!small.empty()) {
      APD_TRY(d_sortjobs.upload(small.data(), small.size() * sizeof(SortJob), cstream));
      hipLaunchKernelGGL(k_sort_cloud_lds, dim3((unsigned)small.size()), dim3(SORT_BLK), (size_t)np2max * 8, cstream, d_sortjobs.as<SortJob>());
      APD_HIP(hipGetLastError());
    }
    if (tkeys_bytes) {
      if (tkeys_bytes > d_tkeys.cap) APD_HIP(hipStreamSynchronize(cstream));
      APD_TRY(d_tkeys.ensure(tkeys_bytes));
      for (int cls = 0; cls < 3; cls++) {
        if (tilejobs[cls].empty()) continue;
        int nmax_c = 0;
        for (TileJob& tjb : tilejobs[cls]) {
          tjb.keys = (unsigned long long*)(d_tkeys.as<char>() + (size_t)tjb.keys);
          nmax_c = std::max(nmax_c, tjb.job.n);
        }
        APD_TRY(d_tilejobs[cls].upload(tilejobs[cls].data(), tilejobs[cls].size() * sizeof(TileJob), cstream));
        const TileJob* dj = d_tilejobs[cls].as<TileJob>();
        const unsigned cnt = (unsigned)tilejobs[cls].size();
        const int nt = 1024 << cls;
        const size_t lds = (size_t)nt * 8;
        if (cls == 0) hipLaunchKernelGGL(k_sort_tiles<1>, dim3(4, cnt), dim3(SORT_BLK), lds, cstream, dj);
        else if (cls == 1) hipLaunchKernelGGL(k_sort_tiles<2>, dim3(4, cnt), dim3(SORT_BLK), lds, cstream, dj);
        else hipLaunchKernelGGL(k_sort_tiles<4>, dim3(4, cnt), dim3(SORT_BLK), lds, cstream, dj);
        for (Cloud& c : clouds)  // the pinned copies are free again behind this launch
          if (c.staged && c.sorted && c.n > (nt << 1) && c.n <= (nt << 2))
            c.staged = false, c.stage_pending = true, c.stage_wait = nullptr, c.stage_seq_word = nullptr, n_stage_pending++;
        hipLaunchKernelGGL(k_merge_tiles, dim3((unsigned)(4 * nt / SORT_BLK), cnt), dim3(SORT_BLK), (size_t)4 * nt * 8, cstream, dj);
        hipLaunchKernelGGL(k_boxes_sorted, dim3((unsigned)(((nmax_c + 15) / 16 + 255) / 256), cnt), dim3(256), 0, cstream, dj);
        APD_HIP(hipGetLastError());
      }
    }
    for (int id : large) {  // generic path: keys in global memory, one launch per bitonic stage
      Cloud& c = clouds[id];
      const int n = c.n;
      int np2 = VOX_TILE;
      while (np2 < n) np2 <<= 1;
      APD_HIP(hipStreamSynchronize(cstream));
      if (pool.on && pool.cstream2) APD_HIP(hipStreamSynchronize(pool.cstream2));  // (d_keys / d_box6 are shared by the lanes)
      APD_TRY(d_keys.ensure((size_t)np2 * 8));
      APD_TRY(d_box6.ensure(6 * sizeof(int)));
      const int init[6] = {0x7f800000, 0x7f800000, 0x7f800000, (int)0x807fffff, (int)0x807fffff, (int)0x807fffff};  // +inf x3, -inf x3 (ordered-int)
      APD_HIP(hipMemcpyAsync(d_box6.p, init, sizeof(init), hipMemcpyHostToDevice, cstream));
      APD_HIP(hipStreamSynchronize(cstream));
      hipLaunchKernelGGL(k_bbox_atomic, dim3(std::min((n + 255) / 256, 256)), dim3(256), 0, cstream, c.opts.as<float4>(), n, d_box6.as<int>());
      int idx_bits = 1;
      while ((1 << idx_bits) < np2) idx_bits++;
      const int mbits = std::min(21, (64 - idx_bits) / 3);
      hipLaunchKernelGGL(k_morton_keys, dim3((np2 + 255) / 256), dim3(256), 0, cstream, c.opts.as<float4>(), n, np2, d_box6.as<int>(),
                         d_keys.as<unsigned long long>(), mbits, idx_bits);
      hipLaunchKernelGGL(k_bitonic_tile_sort, dim3(np2 / VOX_TILE), dim3(1024), 0, cstream, d_keys.as<unsigned long long>());
      for (int k = 2 * VOX_TILE; k <= np2; k <<= 1) {
        for (int j = k >> 1; j >= VOX_TILE; j >>= 1)
          hipLaunchKernelGGL(k_bitonic_global, dim3((np2 / 2 + 255) / 256), dim3(256), 0, cstream, d_keys.as<unsigned long long>(), np2, k, j);
        hipLaunchKernelGGL(k_bitonic_tile_merge, dim3(np2 / VOX_TILE), dim3(1024), 0, cstream, d_keys.as<unsigned long long>(), k);
      }
      hipLaunchKernelGGL(k_gather_sorted, dim3((n + 255) / 256), dim3(256), 0, cstream, d_keys.as<unsigned long long>(), c.opts.as<float4>(), n,
                         c.pts.as<float4>(), c.perm.as<int>(), idx_bits);
      const int nch = (n + 15) / 16, ngr = (n + kGroupPts - 1) / kGroupPts;
      hipLaunchKernelGGL(k_boxes, dim3((nch + 255) / 256), dim3(256), 0, cstream, c.pts.as<float4>(), n, 16, c.cbox.as<Box>(), nch);
      hipLaunchKernelGGL(k_boxes, dim3((ngr + 255) / 256), dim3(256), 0, cstream, c.pts.as<float4>(), n, kGroupPts, c.gbox.as<Box>(), ngr);
      const int nsup = (ngr + kSuperGroups - 1) / kSuperGroups;
      hipLaunchKernelGGL(k_super_boxes, dim3((nsup + 63) / 64), dim3(64), 0, cstream, c.gbox.as<Box>(), ngr, nsup);
      APD_HIP(hipGetLastError());
    }
    return 0;
  }

  int upload_desc() {
    APD_TRY(sort_clouds());
    if (!desc_dirty) {
      const int ln = pool.on ? pool.clane : 0;
      if (pool.nclanes > 1 && pool.desc_gen_lane[ln] != desc_gen) {  // this lane's device table is an older generation
        APD_TRY(d_desc.upload(h_desc.data(), h_desc.size() * sizeof(CloudDesc), cstream));
        pool.desc_gen_lane[ln] = desc_gen;
      }
      return 0;
    }
    std::vector<CloudDesc>& h = h_desc;
    h.resize(clouds.size());
    for (size_t i = 0; i < clouds.size(); i++) {
      if (clouds[i].n > 0) APD_TRY(clouds[i].cov.ensure((size_t)clouds[i].n * 6 * sizeof(double)));
      h[i].pts = clouds[i].pts.as<float4>();
      h[i].opts = clouds[i].opts.as<float4>();
      h[i].perm = clouds[i].perm.as<int>();
      h[i].cbox = clouds[i].cbox.as<Box>();
      h[i].gbox = clouds[i].gbox.as<Box>();
      h[i].cov = clouds[i].cov.as<double>();
      h[i].n = clouds[i].n;
      h[i].pad_ = 0;
    }
    APD_TRY(d_desc.upload(h.data(), h.size() * sizeof(CloudDesc), cstream));
    desc_dirty = false;
    desc_gen++;
    pool.desc_gen_lane[pool.on ? pool.clane : 0] = desc_gen;
    return 0;
  }

  static std::string errflag_text(int flag) {
    static const char* const meaning[] = {"", "", "fewer than k neighbours at a finite distance (non-finite input points?)",
                                          "covariance regularisation failed (non-finite covariance)", "k-NN candidate list overflow",
                                          "voxel grid: leaf size too small for the cloud"};
    return "device error flag " + std::to_string(flag) + (flag >= 2 && flag <= 5 ? std::string(": ") + meaning[flag] : std::string());
  }
  int check_errflag(const char* what) {
    int flag = 0;
    int* h_flag = (int*)(h_poll + kHostResults * sizeof(ResultRec)) + 65539;  // last pinned word: never part of a poll
    APD_HIP(hipMemcpyAsync(h_flag, d_errflag.p, sizeof(int), hipMemcpyDeviceToHost, cstream));
    APD_HIP(hipStreamSynchronize(cstream));
    flag = h_flag[0];
    if (flag) {
      APD_HIP(hipMemsetAsync(d_errflag.p, 0, sizeof(int), cstream));
      return fail(APDGICP_ERR_INTERNAL, std::string(what) + ": " + errflag_text(flag));
    }
    return 0;
  }

  // calculate_covariances for every listed cloud that lacks them (A:122-127, A:303-363)
  bool warned_select = false;
  bool defer_errflag = false;  // set by the align paths: the flag is then read together with the final status
  int compute_covariances(const std::vector<int>& ids_in, bool force = false) {
    std::vector<int> ids;
    APD_TRY(filter_cov_ids(ids_in, force, ids));
    if (ids.empty()) return 0;
    APD_HIP(hipSetDevice(device));
    APD_TRY(upload_desc());
    APD_TRY(d_ids.upload(ids.data(), ids.size() * sizeof(int), cstream));
    long long total = 0;
    for (int id : ids) total += clouds[id].n;
    if (fuse_first_search && defer_errflag && !pool.on && knn_pruned && nn_pruned && params.k_correspondences <= KNN_NC && ids.size() <= 2 && total < 40000) {
      pending_knn = ids;  // apdgicp_align: launched together with the cold search of the first tick (k_knn_and_search)
      return 0;
    }
    APD_TRY(launch_knn(ids.data(), d_ids.as<int>(), (int)ids.size(), cstream));
    if (!defer_errflag) APD_TRY(check_errflag("k_knn_cov"));
    return 0;
  }
  // apdgicp_align sets fuse_first_search around its set-up and run: the covariance launch of the clouds set since the last align
  // is held back (pending_knn) and goes out with the first tick's search; anything else that could need the covariances first
  // calls flush_pending_knn
  bool fuse_first_search = false;
  std::vector<int> pending_knn;
  int flush_pending_knn() {
    if (pending_knn.empty()) return 0;
    const std::vector<int> ids = pending_knn;
    pending_knn.clear();
    return launch_knn(ids.data(), d_ids.as<int>(), (int)ids.size(), stream);
  }
  // the listed clouds that still need covariances, validated, without duplicates
  int filter_cov_ids(const std::vector<int>& ids_in, bool force, std::vector<int>& ids) {
    for (int id : ids_in) {
      if (id < 0 || id >= (int)clouds.size() || clouds[id].n <= 0) return fail(APDGICP_ERR_NO_INPUT, "cloud not set");
      if (clouds[id].cov_valid && !force) continue;
      if (clouds[id].n < params.k_correspondences)
        return fail(APDGICP_ERR_TOO_FEW_POINTS, "cloud has fewer points than k_correspondences");
      if (std::find(ids.begin(), ids.end(), id) == ids.end()) ids.push_back(id);
    }
    return 0;
  }
  // one covariance launch for the clouds ids[0..count) (device copy of the list: d_list) on stream `st`
  int launch_knn(const int* ids, const int* d_list, int count, hipStream_t st) {
    if (count <= 0) return 0;
    if (st == cstream) {
      for (int i = 0; i < count; i++) lane_touch(clouds[ids[i]]);
      APD_TRY(lane_order());
    }
    roctx_range rr("apdgicp:knn_cov");
    int nmax = 0;
    long long total = 0;
    for (int i = 0; i < count; i++) nmax = std::max(nmax, clouds[ids[i]].n), total += clouds[ids[i]].n;
    if (params.k_correspondences > 2 * KNN_NC) {  // any k: selection by bisection, no lists (for experiments: ~35 sweeps of the cloud per query block)
      // O(n^2): a block of SEL_Q queries sweeps its cloud ~35 times.  One dispatch over a 500k-point submap would run for seconds and
      // hold the device queue of every handle; the launch is cut into dispatches of at most ~2^34 point visits (~10 ms) each
      const long long blocks = (nmax + SEL_Q - 1) / SEL_Q;
      const long long per = std::max<long long>(1, std::min<long long>(blocks, (1ll << 34) / (36ll * std::max(1, nmax) * std::max(1, count))));
      if (blocks > per && !warned_select) {
        warned_select = true;
        std::fprintf(stderr, "[apdgicp] k_correspondences = %d > 64 on a cloud of %d points: the selection kernel is O(n^2) (%lld dispatches of ~10 ms)\n",
                     params.k_correspondences, nmax, (blocks + per - 1) / per);
      }
      for (long long b0 = 0; b0 < blocks; b0 += per)
        hipLaunchKernelGGL(k_knn_cov_select, dim3((unsigned)std::min(per, blocks - b0), (unsigned)count), dim3(64), 0, st, d_desc.as<CloudDesc>(), d_list,
                           params.k_correspondences, params.regularization, d_errflag.as<int>(), (int)b0);
    } else if (knn_pruned && params.k_correspondences <= KNN_NC) {
      // queries per wave = 64 / lanes per query: fewer queries per wave shorten the per-wave dependency chain and shrink its LDS
      // lists, which wins whenever the GPU is not already full (r01, 2 clouds of 8k: 0.10 / 0.13 / 0.21 ms for 4 / 8 / 16)
      const int qpw = total >= 100000 ? 16 : total >= 40000 ? 8 : 4;
      const dim3 grid((unsigned)((nmax + qpw - 1) / qpw), (unsigned)count);
      // throughput launches regularise in a second launch with every lane busy; a single cloud (latency) keeps it fused
      const int raw = qpw != 4 && params.regularization != APDGICP_REG_NONE ? 1 : 0;
      if (qpw == 16)
        hipLaunchKernelGGL(k_knn_cov_coop<4>, grid, dim3(64), knn_coop_lds_bytes(qpw), st, d_desc.as<CloudDesc>(), d_list, params.k_correspondences,
                           params.regularization, d_errflag.as<int>(), d_stats.as<unsigned long long>(), raw);
      else if (qpw == 4)
        hipLaunchKernelGGL(k_knn_cov_coop<16>, grid, dim3(64), knn_coop_lds_bytes(qpw), st, d_desc.as<CloudDesc>(), d_list, params.k_correspondences,
                           params.regularization, d_errflag.as<int>(), d_stats.as<unsigned long long>(), raw);
      else
        hipLaunchKernelGGL(k_knn_cov_coop<8>, grid, dim3(64), knn_coop_lds_bytes(qpw), st, d_desc.as<CloudDesc>(), d_list, params.k_correspondences,
                           params.regularization, d_errflag.as<int>(), d_stats.as<unsigned long long>(), raw);
      if (raw)
        hipLaunchKernelGGL(k_regularize_covs, dim3((unsigned)((nmax + 255) / 256), (unsigned)count), dim3(256), 0, st, d_desc.as<CloudDesc>(), d_list,
                           params.regularization, d_errflag.as<int>());
    } else {
      const dim3 grid((unsigned)((nmax + KNN_BLK - 1) / KNN_BLK), (unsigned)count);
      // the brute-force kernel: the cross-check (APDGICP_KNN_MODE=brute), and the only path for 32 < k <= 64
      if (params.k_correspondences <= KNN_NC)
        hipLaunchKernelGGL(k_knn_cov<64>, grid, dim3(KNN_BLK), knn_lds_bytes_brute(64), st, d_desc.as<CloudDesc>(), d_list, params.k_correspondences,
                           params.regularization, d_errflag.as<int>());
      else
        hipLaunchKernelGGL(k_knn_cov<128>, grid, dim3(KNN_BLK), knn_lds_bytes_brute(128), st, d_desc.as<CloudDesc>(), d_list, params.k_correspondences,
                           params.regularization, d_errflag.as<int>());
    }
    APD_HIP(hipGetLastError());
    for (int i = 0; i < count; i++) clouds[ids[i]].cov_valid = true;
    return 0;
  }

  // ------------------------------------------------------------------ batch set-up
  // pipeline_cov: the covariances of clouds used by ONE pair group only are not launched here but by run_align on that
  // group's stream, so that they overlap with the (latency-bound) optimiser ticks of the other groups
  std::vector<int> cov_list;        // [shared clouds | group 0 | group 1 | ...]
  std::vector<int> cov_group_off;   // offsets into cov_list: group g owns [off[g + 1], off[g + 2]); the shared part is [0, off[1])
  // pair groups in flight: about 8 pairs per group up to the number of streams (r01, one scan against K keyframes, GN-20:
  // K = 8: 0.90 / 0.92 / 0.92 ms with 1 / 2 / 3 groups, K = 12: 1.12 / 0.98 / 1.02, K = 24: 1.41 / 1.23 / 1.17)
  bool keep_maha = true;  // false for batch handles: nothing reads mahalanobis_ there unless the optimiser is LM (k_error)
  int max_groups = 1 << 30;  // apdgicp_batch_set_pair_groups: a caller that keeps several batches (handles) in flight wants one group each
  int ngroups_for_first_tick() const { return group_count(); }
  int group_count() const { return std::max(1, std::min<int>(std::min<int>(ngroups_cfg, max_groups), (npairs + 4) / 8)); }
  // need_cov = false: the pairs are searched only (apdgicp_nearest_neighbours_of): no covariance launch for clouds that lack them
  int setup_pairs(const apdgicp_pair* pairs, int64_t n, bool with_guess, bool pipeline_cov = false, bool need_cov = true) {
    if (n <= 0 || n > 65536) return fail(APDGICP_ERR_INVALID_ARG, "n_pairs must be in [1, 65536]");
    APD_HIP(hipSetDevice(device));
    APD_TRY(pool_leave());
    h_pairs.resize(n);
    std::vector<float>& guesses = h_guesses;
    guesses.resize((size_t)n * 16);
    std::vector<int> need;
    nmax_src = 0, nmax_tgt = 0;
    for (int64_t i = 0; i < n; i++) {
      const int s = pairs[i].source_cloud, t = pairs[i].target_cloud;
      if (s < 0 || t < 0 || s >= (int)clouds.size() || t >= (int)clouds.size() || clouds[s].n <= 0 || clouds[t].n <= 0)
        return fail(APDGICP_ERR_NO_INPUT, "pair references a cloud that is not set");
      h_pairs[i].src = s, h_pairs[i].tgt = t;
      memcpy(&guesses[(size_t)i * 16], pairs[i].guess, 16 * sizeof(float));
      need.push_back(s), need.push_back(t);
      nmax_src = std::max(nmax_src, clouds[s].n);
      nmax_tgt = std::max(nmax_tgt, clouds[t].n);
    }
    cov_list.clear();
    cov_group_off.clear();
    npairs = (int)n;
    const int ng = group_count();
    if (pipeline_cov && ng > 1 && knn_pruned) {
      std::vector<int> ids;
      APD_TRY(filter_cov_ids(need, false, ids));
      // owner group of every cloud: -2 unseen, -1 shared by several groups
      std::vector<int> owner(clouds.size(), -2);
      for (int g = 0; g < ng; g++) {
        const int p0 = (int)((long long)npairs * g / ng), p1 = (int)((long long)npairs * (g + 1) / ng);
        for (int q = p0; q < p1; q++)
          for (int id : {pairs[q].source_cloud, pairs[q].target_cloud}) owner[id] = owner[id] == -2 || owner[id] == g ? g : -1;
      }
      cov_group_off.assign(ng + 2, 0);
      for (int g = -1; g < ng; g++) {
        for (int id : ids)
          if (owner[id] == g) cov_list.push_back(id);
        cov_group_off[g + 2] = (int)cov_list.size();
      }
      APD_TRY(upload_desc());
      if (!cov_list.empty()) {
        APD_TRY(d_ids.upload(cov_list.data(), cov_list.size() * sizeof(int), stream));
        APD_TRY(launch_knn(cov_list.data(), d_ids.as<int>(), cov_group_off[1], stream));  // clouds shared between groups: now
      }
    } else if (need_cov) {
      defer_errflag = true;
      const int rc_cov = compute_covariances(need);
      defer_errflag = false;
      APD_TRY(rc_cov);
    }
    APD_TRY(upload_desc());
    for (int64_t i = 0; i < n; i++) h_pairs[i].s = h_desc[h_pairs[i].src], h_pairs[i].t = h_desc[h_pairs[i].tgt];
    npairs = (int)n;
    if ((size_t)n * sizeof(PairState) > d_state.cap) APD_HIP(hipStreamSynchronize(stream));
    APD_TRY(d_state.ensure(n * sizeof(PairState)));
    APD_TRY(d_results.ensure(n * sizeof(ResultRec) + (n + 1) * sizeof(int)));  // records, then the status words of the poll
    APD_TRY(d_pairs.upload(h_pairs.data(), n * sizeof(PairDesc), stream));
    (void)with_guess;
    APD_TRY(upload_guesses(guesses.data(), n));

    // launch shape of the search: the pruned kernels take one source point per lane; the brute-force cross-check
    // (APDGICP_NN_MODE=brute) S = 2 or 4 per lane and T target splits, enough blocks to fill the GPU
    int S = 1, T = 1;
    if (!nn_pruned) {
      S = ((long long)npairs * ((nmax_src + 1023) / 1024) >= 256) ? 4 : 2;
      const int src_blocks = (nmax_src + NN_BLK * S - 1) / (NN_BLK * S);
      T = std::max(1, std::min((512 + npairs * src_blocks - 1) / (npairs * src_blocks), 64));
    }
    nn_S = S;
    work.T = T;
    work.nstride = (nmax_src + 255) & ~255;
    work.nblk_max = (nmax_src + LIN_BLK - 1) / LIN_BLK;
    work.cap = std::numeric_limits<float>::infinity();
    work.active = nullptr;
    const size_t ns = work.nstride;
    APD_TRY(b_nnpart.ensure((size_t)npairs * T * ns * 8));
    APD_TRY(b_corr.ensure((size_t)npairs * ns * 4));
    APD_TRY(b_nnpt.ensure((size_t)npairs * ns * 16));
    APD_TRY(b_nnaux.ensure((size_t)npairs * ns * 16));
    APD_TRY(b_sqd.ensure((size_t)npairs * ns * 4));
    APD_TRY(b_maha.ensure((size_t)npairs * 6 * ns * 8));
    APD_TRY(b_blkpart.ensure((size_t)npairs * work.nblk_max * kRed * 8));
    APD_TRY(b_errpart.ensure((size_t)npairs * work.nblk_max * 8));
    work.nnpart = b_nnpart.as<unsigned long long>();
    const bool keep_point_results = keep_maha || params.optimizer == APDGICP_OPT_LM;  // see keep_maha
    work.corr = keep_point_results ? b_corr.as<int>() : nullptr;
    work.nnpt = b_nnpt.as<float4>();
    work.nnaux = nn_skin && nn_pruned ? b_nnaux.as<float4>() : nullptr;
    work.skin_mul = (1.f + nn_skin_rel) * (1.f + nn_skin_rel);
    work.skin_add = nn_skin_abs * nn_skin_abs;
    work.sqd = keep_point_results ? b_sqd.as<float>() : nullptr;
    work.maha = (keep_maha || params.optimizer == APDGICP_OPT_LM) ? b_maha.as<double>() : nullptr;
    work.blkpart = b_blkpart.as<double>();
    work.errpart = b_errpart.as<double>();
    work.stats = stats_blocks ? nullptr : d_stats.as<unsigned long long>();  // (the timeline is taken WITHOUT the counters: their atomics slow the kernels fivefold)
    work.timeline = stats_blocks ? d_stats.as<unsigned long long>() + 16 : nullptr;
    work.stats_blocks = stats_blocks;
    APD_TRY(b_ticket.ensure((size_t)2 * npairs * sizeof(int)));
    if (work.ticket != b_ticket.as<int>() || work.npairs != npairs) tickets_dirty = true;  // fresh memory, or another layout
    work.ticket = b_ticket.as<int>();
    // one dense pair with more search blocks than the GPU holds at once (1280 four-wave blocks): the blocks report their cost, and from the third
    // tick of an align on they are launched costliest first (k_block_order)
    work.blk_cost = nullptr, work.blk_order = nullptr;
    const int nn_blocks = (nmax_src + 63) / 64;
    if (blk_order_on && nn_pruned && npairs == 1 && nn_blocks > 1280 && nn_blocks <= ORDER_MAX && nmax_tgt > SORT_LDS_MAX_N) {
      APD_TRY(b_blkcost.ensure((size_t)nn_blocks * 4));
      APD_TRY(b_blkorder.ensure((size_t)nn_blocks * 4));
      work.blk_cost = b_blkcost.as<unsigned>();
    }
    work.init = nullptr;
    work.coop_search = 1;
    work.sparse_max = nn_sparse;
    work.post = nullptr, work.post_seq = 0;
    work.pair0 = 0;
    work.npairs = npairs;
    return 0;
  }

  // A launch covers the pairs [p0, p0 + np) on stream `st`.
  // L:56 x0 = guess.cast<double>(): widened here, so that the kernels read the pose through scalar loads
  std::vector<Rigid> h_guess;
  int upload_guesses(const float* g /* n x 16, column-major */, int64_t n) {
    h_guess.resize((size_t)n);
    for (int64_t p = 0; p < n; p++)
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 4; j++) h_guess[p].m[4 * i + j] = (double)g[16 * p + i + 4 * j];
    return d_guess.upload(h_guess.data(), h_guess.size() * sizeof(Rigid), stream);
  }

  struct Span {
    int p0, np;
    hipStream_t st;
  };
  Span whole() const { return Span{0, npairs, stream}; }
  // what the tick launches run over: the batch set up by setup_pairs, or (inside pool_enqueue_chunk) the pair pool
  bool in_pool = false;
  // debug trace of the optimiser (apdgicp_set_trace): single-registration handles only
  bool trace_on = false;
  DevBuf d_trace;
  int trace_cap_trial = 0, trace_cap_pose = 0;
  size_t trace_bytes() const { return 16 + ((size_t)5 * trace_cap_trial + (size_t)12 * trace_cap_pose) * sizeof(double); }
  int trace_reset() {  // in front of an align: capacities from the current parameters, counts zero
    if (!trace_on) return 0;
    trace_cap_pose = std::max(1, params.max_iterations);
    trace_cap_trial = trace_cap_pose * std::max(1, params.lm_max_iterations);
    if (trace_bytes() > d_trace.cap) APD_HIP(hipStreamSynchronize(stream));
    APD_TRY(d_trace.ensure(trace_bytes()));
    const int hdr[4] = {0, 0, trace_cap_trial, trace_cap_pose};
    APD_HIP(hipMemcpyAsync(d_trace.p, hdr, sizeof(hdr), hipMemcpyHostToDevice, stream));  // (pageable source: staged before the call returns)
    return 0;
  }
  Work t_work() const {
    Work w = in_pool ? pool.work : work;
    w.trace = trace_on && !in_pool && npairs == 1 ? d_trace.as<double>() : nullptr;
    if (in_pool) w.active = pool.L[pool.cur].active.as<int>();
    w.xf_linear = (params.flags & APDGICP_FLAG_XF_LINEAR_CHAIN) ? 1 : 0;  // (read at launch time: set_params may come between aligns)
    return w;
  }
  const PairDesc* t_pairs() const { return in_pool ? pool.pairs.as<PairDesc>() : d_pairs.as<PairDesc>(); }
  PairState* t_state() const { return in_pool ? pool.state.as<PairState>() : d_state.as<PairState>(); }
  int t_npairs() const { return in_pool ? pool.cap : npairs; }
  int t_nmax_src() const { return in_pool ? pool.nmax_src : nmax_src; }
  int t_nmax_tgt() const { return in_pool ? pool.nmax_tgt : nmax_tgt; }

  int launch_nn(Span sp) {
    const int src_blocks = nn_pruned ? (t_nmax_src() + 64 * nn_S - 1) / (64 * nn_S) : (t_nmax_src() + NN_BLK * nn_S - 1) / (NN_BLK * nn_S);
    dim3 grid((unsigned)src_blocks, nn_pruned ? (unsigned)sp.np : (unsigned)t_work().T, nn_pruned ? 1u : (unsigned)sp.np);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool timed = profile_nn && !in_pool && ((cur_tick + profile_phase) % profile_stride == 0);
    if (in_pool && pool.cur_timed) {
      e0 = pool.cur_timed->e0, e1 = pool.cur_timed->e1;
    } else if (timed) {
      nn_pairs_acc += sp.np;
      if (nn_events_used == nn_events.size()) {
        hipEvent_t a, b;
        APD_HIP(hipEventCreate(&a));
        APD_HIP(hipEventCreate(&b));
        nn_events.emplace_back(a, b);
      }
      e0 = nn_events[nn_events_used].first, e1 = nn_events[nn_events_used].second;
      nn_events_used++;
    }
    const CloudDesc* cd = d_desc.as<CloudDesc>();
    const PairDesc* pd = t_pairs();
    const PairState* st = t_state();
    Work w = t_work();
    w.pair0 = sp.p0;
    w.cap = nn_cap;
    w.init = init_tick ? d_guess.as<Rigid>() : nullptr;
    // a timed launch carries its own start/stop events (hipExtLaunchKernelGGL): the kernel's begin and end timestamps, as
    // a profiler reports them, not the stream's idle gaps around it
#define APD_NN_LAUNCH(KERNEL, BLOCK) (last_nn_kernel = #KERNEL, hipExtLaunchKernelGGL(KERNEL, grid, dim3(BLOCK), 0, sp.st, e0, e1, 0, cd, pd, st, w))
    // all groups tick together; an LM batch shrinks to its slow pairs, and those few get the wave split of a small batch
    const long long tick_blocks = (long long)(cur_active > 0 ? cur_active : t_npairs()) * src_blocks;
    // large (dense) targets: a wave's 64 points touch many more groups (10.8 instead of 1.7 per wave for 100k x 500k), so
    // splitting the scans over 4 waves still pays with a few thousand blocks (r01: 0.170 -> 0.143 ms per iteration)
    const bool big_target = t_nmax_tgt() > SORT_LDS_MAX_N;
    // a handle limited to one pair group shares the GPU with other busy handles: throughput counts there, not the latency of
    // this launch, and one wave per 64 points does no redundant bound work (three handles in flight: 1.32 -> 1.27 ms per step)
    const int w_full = max_groups == 1 || in_pool ? 1 : 2;
    const int W = nn_W ? nn_W : tick_blocks <= 256 ? 8 : (tick_blocks <= 1024 || (big_target && tick_blocks <= 8192)) ? 4 : w_full;
    // throughput regime (one wave per 64 points) with neighbour keeping on: blocks of 256 points that pack the points still
    // searching into as few waves as they fill (k_nn_compact)
    // (not for dense targets beyond 16384 points when the engine chose the regime itself: a wave there walks many batches of
    // group boxes, and 100k x 500k measured 0.232 ms per iteration against 0.170 with one-wave blocks, 0.103 with W = 4)
    if (nn_pruned && W == 1 && t_work().nnaux && (!big_target || nn_W == 1)) {
      // (blocks of 512 points waste half as many tail waves -- 29.5 instead of 36 search waves per 256 points over the 20
      // ticks of the bench -- but hold twice the LDS until their slowest wave is done: measured 34.7 k vs 37.0 k registrations/s)
      grid.x = (unsigned)((t_nmax_src() + 255) / 256);
      APD_NN_LAUNCH(k_nn_compact<4>, 256);
    } else if (nn_pruned) {
      if (W == 8) APD_NN_LAUNCH((k_nn_pruned<1, 8>), 512);
      else if (W == 4) APD_NN_LAUNCH((k_nn_pruned<1, 4>), 256);
      else if (W == 2) APD_NN_LAUNCH((k_nn_pruned<1, 2>), 128);
      else APD_NN_LAUNCH((k_nn_pruned<1, 1>), 64);
    } else if (nn_S == 2) APD_NN_LAUNCH(k_nn_partial<2>, NN_BLK);
    else APD_NN_LAUNCH(k_nn_partial<4>, NN_BLK);
#undef APD_NN_LAUNCH
    // the kernel bench.py names: that of the LARGEST timed launch so far (the few-pair tail of a draining pool gets another block shape,
    // and whether the last timed launch of a run is such a tail is a matter of timing)
    if (in_pool && pool.cur_timed && sp.np >= pool.timed_kernel_np) pool.timed_kernel = last_nn_kernel, pool.timed_kernel_np = sp.np;
    return 0;
  }

  int launch_linearize(Span sp, int mode /* 0 cost only, 1 H/b/cost, 2 + fused GN/LM step */) {
    const dim3 grid((unsigned)((t_nmax_src() + LIN_BLK - 1) / LIN_BLK), (unsigned)sp.np);
    Work w = t_work();
    w.pair0 = sp.p0;
    w.init = init_tick && mode == 2 ? d_guess.as<Rigid>() : nullptr;
    const bool f32 = (params.flags & APDGICP_FLAG_FP32_POINT_MATH) != 0;
    const bool alg = (params.flags & APDGICP_FLAG_ALGEBRAIC_APD) != 0;
    if (alg && mode == 2)
      hipLaunchKernelGGL((k_linearize<true, false, true>), grid, dim3(LIN_BLK), 0, sp.st, d_desc.as<CloudDesc>(), t_pairs(), t_state(), w, consts(), mode);
    else if (alg)
      hipLaunchKernelGGL((k_linearize<false, false, true>), grid, dim3(LIN_BLK), 0, sp.st, d_desc.as<CloudDesc>(), t_pairs(), t_state(), w, consts(), mode);
    else if (mode == 2 && f32)
      hipLaunchKernelGGL((k_linearize<true, true>), grid, dim3(LIN_BLK), 0, sp.st, d_desc.as<CloudDesc>(), t_pairs(), t_state(), w, consts(), mode);
    else if (mode == 2)
      hipLaunchKernelGGL(k_linearize<true>, grid, dim3(LIN_BLK), 0, sp.st, d_desc.as<CloudDesc>(), t_pairs(), t_state(), w,
                         consts(), mode);
    else if (f32)
      hipLaunchKernelGGL((k_linearize<false, true>), grid, dim3(LIN_BLK), 0, sp.st, d_desc.as<CloudDesc>(), t_pairs(), t_state(), w, consts(), mode);
    else
      hipLaunchKernelGGL(k_linearize<false>, grid, dim3(LIN_BLK), 0, sp.st, d_desc.as<CloudDesc>(), t_pairs(), t_state(), w,
                         consts(), mode);
    return 0;
  }

  int launch_error(Span sp, bool fuse) {
    const dim3 grid((unsigned)((t_nmax_src() + LIN_BLK - 1) / LIN_BLK), (unsigned)sp.np);
    Work w = t_work();
    w.pair0 = sp.p0;
    if (fuse && post_tick) w.post = d_post.as<PollPost>(), w.post_seq = poll_seq;
    hipLaunchKernelGGL(k_error, grid, dim3(LIN_BLK), 0, sp.st, d_desc.as<CloudDesc>(), t_pairs(), t_state(), w,
                       consts(), fuse ? 1 : 0);
    return 0;
  }

  float gate_cap() const {  // smallest float >= corr_dist_threshold^2 (A:156 compares the float distance with the double square)
    const double thr2 = consts().thr2;
    float capf = (float)thr2;
    if ((double)capf < thr2) capf = std::nextafterf(capf, std::numeric_limits<float>::infinity());
    return capf;
  }

  // one tick of the state machines of the pairs in `sp`
  int launch_tick(Span sp) {
    nn_cap = gate_cap();
    int rc_nn = 0;
    if (!pending_knn.empty()) {
      const long long tick_blocks = (long long)sp.np * ((t_nmax_src() + 63) / 64);
      if (init_tick && sp.np == 1 && !in_pool && !nn_W && tick_blocks <= 256 && !(profile_nn && (cur_tick + profile_phase) % profile_stride == 0)) {
        // one launch: the covariances of the clouds just set + this tick's cold search (k_knn_and_search)
        int nmax = 0;
        for (int id : pending_knn) nmax = std::max(nmax, clouds[id].n);
        const int tasks = (nmax + 3) / 4, knn_blocks = ((int)pending_knn.size() * tasks + 7) / 8;
        Work w = t_work();
        w.pair0 = sp.p0, w.cap = nn_cap, w.init = d_guess.as<Rigid>();
        last_nn_kernel = "k_knn_and_search";
        hipLaunchKernelGGL(k_knn_and_search, dim3((unsigned)(knn_blocks + (t_nmax_src() + 63) / 64)), dim3(512), (size_t)8 * knn_coop_lds_bytes(4), sp.st,
                           d_desc.as<CloudDesc>(), d_ids.as<int>(), (int)pending_knn.size(), tasks, params.k_correspondences, params.regularization,
                           d_errflag.as<int>(), d_stats.as<unsigned long long>(), t_pairs(), (const PairState*)t_state(), w);
        for (int id : pending_knn) clouds[id].cov_valid = true;
        pending_knn.clear();
      } else {
        rc_nn = flush_pending_knn();
        if (rc_nn == 0) rc_nn = launch_nn(sp);
      }
    } else {
      rc_nn = launch_nn(sp);
    }
    nn_cap = std::numeric_limits<float>::infinity();
    APD_TRY(rc_nn);
    APD_TRY(launch_linearize(sp, 2));
#if !defined(APD_ABL_LM_NO_ERROR) || APD_ABL_LM_NO_ERROR < 3
    if (params.optimizer == APDGICP_OPT_LM) APD_TRY(launch_error(sp, true));
#endif
    return 0;
  }

  // waits for the poll just enqueued on the current slot: spinning on its sequence word when it posts one (bounded: a GPU
  // that takes longer than a few hundred microseconds gets the sleeping wait), else on the event
  int poll_seq = 0;
  bool pending_spin = false;
  int wait_poll() {
    if (pending_spin) {
      volatile int* h_seq = (volatile int*)((int*)(h_poll + kHostResults * sizeof(ResultRec)) + 65538);
      const auto t0 = std::chrono::steady_clock::now();
      for (unsigned it = 0;; it++) {
        if (*h_seq == poll_seq) {
          std::atomic_thread_fence(std::memory_order_acquire);
          return 0;
        }
        if ((it & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(400)) break;
      }
    }
    APD_HIP(hipEventSynchronize(ev_poll));
    return 0;
  }

  int ensure_alt_slot() {
    if (alt.h_poll) return 0;
    APD_HIP(hipHostMalloc((void**)&alt.h_poll, 65540 * sizeof(int) + kHostResults * sizeof(ResultRec), hipHostMallocDefault));
    memset(alt.h_poll, 0, 65540 * sizeof(int) + kHostResults * sizeof(ResultRec));
    APD_HIP(hipHostGetDevicePointer((void**)&alt.h_poll_dev, alt.h_poll, 0));
    alt.h_status = (int*)alt.h_poll;
    APD_HIP(hipEventCreateWithFlags(&alt.ev_poll, hipEventDisableTiming));
    return 0;
  }
  void swap_slots() {
    std::swap(d_results, alt.d_results);
    std::swap(h_poll, alt.h_poll), std::swap(h_poll_dev, alt.h_poll_dev), std::swap(h_status, alt.h_status);
    std::swap(results_on_host, alt.results_on_host);
    std::swap(ev_poll, alt.ev_poll);
    nn_events.swap(alt.nn_events);
    std::swap(nn_events_used, alt.nn_events_used), std::swap(nn_pairs_acc, alt.nn_pairs_acc);
    std::swap(pending, alt.pending), std::swap(pending_npairs, alt.pending_npairs), std::swap(pending_ticks, alt.pending_ticks);
    std::swap(poll_seq, alt.poll_seq), std::swap(pending_spin, alt.pending_spin);
  }
  // waits for the deferred final poll of the CURRENT slot's align and reports its device error flag
  int finish_align() {
    if (!pending) return 0;
    pending = false;
    {
      roctx_range rr("apdgicp:poll");
      APD_TRY(wait_poll());
    }
    last_ticks = pending_ticks;
    if (profile_nn) APD_TRY(collect_nn_profile());
    if (h_status[pending_npairs]) {
      const int flag = h_status[pending_npairs];
      tickets_dirty = true;
      APD_HIP(hipMemsetAsync(d_errflag.p, 0, sizeof(int), stream));
      return fail(APDGICP_ERR_INTERNAL, errflag_text(flag));
    }
    return 0;
  }
  int collect_nn_profile() {
    last_nn_ms = 0;
    last_nn_launches = 0;
    last_nn_pairs = nn_pairs_acc;
    for (size_t i = 0; i < nn_events_used; i++) {
      float ms = 0;
      APD_HIP(hipEventElapsedTime(&ms, nn_events[i].first, nn_events[i].second));
      last_nn_ms += ms;
      last_nn_launches++;
    }
    return 0;
  }

  // L:55-80 for every pair; leaves ResultRec[npairs] in d_results.  Blocks until all pairs are done
  // (the loop length is data dependent), polling the device every few ticks.
  // defer_poll: when the whole run is ONE chunk of ticks (Gauss-Newton), return right after enqueueing the final poll;
  // finish_align() waits for it.  Levenberg-Marquardt runs poll as they go and are complete on return either way.
  int run_align(bool defer_poll = false) {
    APD_HIP(hipSetDevice(device));
    APD_TRY(finish_align());  // (an uncollected deferred align of this slot)
    APD_TRY(trace_reset());
    nn_events_used = 0;
    nn_pairs_acc = 0;
    cur_tick = 0;
    work.blk_order = nullptr;  // (every align starts in index order: the costs belong to a pose and a cloud)
    profile_stride = std::max(1, env_int("APDGICP_PROFILE_STRIDE", 10));
    profile_phase = (profile_phase + 1) % profile_stride;
    // every align starts cold: k_init_state zeroes n_lin, and the search ignores the hint array until the first linearize of
    // this align has rewritten it (hints of an earlier run would still be valid bounds, but nothing observable -- timing
    // included -- may depend on call history)
    const bool lm = params.optimizer == APDGICP_OPT_LM;
    const long long tick_cap = (long long)std::max(0, params.max_iterations) * (lm ? std::max(1, params.lm_max_iterations) : 1);
    // the fused optimiser builds the state in the last block of the first k_linearize (Work::init): one launch less per
    // align.  The arrival counters reset themselves; only a run that ended in an error may have left them mid-count
    fold_init = nn_pruned && tick_cap > 0 && !tickets_dirty;
    if (!fold_init || ngroups_for_first_tick() > 1) APD_TRY(flush_pending_knn());  // (the fused launch belongs to the folded first tick on the main stream)
    if (!fold_init) {
      hipLaunchKernelGGL(k_init_state, dim3((npairs + 63) / 64), dim3(64), 0, stream, d_state.as<PairState>(), d_guess.as<Rigid>(), npairs,
                         params.max_iterations, b_ticket.as<int>());
      tickets_dirty = false;
    }
    // LM: the loop length is data dependent, poll every few ticks.  GN runs max_iterations ticks unless a
    // pair hits an exactly-zero step, so one poll at the end is enough.
    // LM polls after 1, then 2, then every 4 ticks: with the shipped parameters most registrations converge in 1-3
    // iterations, and every tick enqueued past convergence is three empty launches
    auto chunk_at = [&](long long done) {
      if (!lm) return std::min(64, std::max(1, params.max_iterations));
      return done == 0 ? 1 : done < 3 ? 2 : 4;
    };
    long long ticks = 0;
    bool all_done = params.max_iterations <= 0;
    work.active = nullptr;
    int n_active = npairs;
    while (!all_done && ticks < tick_cap) {
      // GN needs exactly max_iterations ticks unless a pair converges early; never enqueue more than that
      const int todo = (int)std::min<long long>(chunk_at(ticks), tick_cap - ticks);
      // pair groups on their own streams: fork after the main stream's set-up work, join before the poll
      const int ng = ticks == 0 ? group_count() : std::max(1, std::min<int>(std::min<int>(ngroups_cfg, max_groups), n_active / 2));
      APD_TRY(ensure_group_streams(ng));
      if (ng > 1) {
        APD_HIP(hipEventRecord(ev_main, stream));
        for (int g = 1; g < ng; g++) APD_HIP(hipStreamWaitEvent(gstreams[g - 1], ev_main, 0));
      }
      if (ticks == 0 && (int)cov_group_off.size() == ng + 2) {  // this group's own clouds: behind the fork, in front of its ticks
        for (int g = 0; g < ng; g++) {
          const int o0 = cov_group_off[g + 1], o1 = cov_group_off[g + 2];
          APD_TRY(launch_knn(cov_list.data() + o0, d_ids.as<int>() + o0, o1 - o0, g == 0 ? stream : gstreams[g - 1]));
        }
        cov_group_off.clear();
      }
      cur_active = n_active;
      // poll = finalize: the result records and, right behind them in the same buffer, the status words and the error
      // flag; small batches bring the records home in the same copy, so align() needs no second round trip
      int* d_stat = (int*)(d_results.as<char>() + (size_t)npairs * sizeof(ResultRec));
      // k_finalize writes the host's copy itself (pinned, device-visible memory): no copy kernel behind it
      results_on_host = npairs <= kHostResults;
      h_status = results_on_host ? (int*)(h_poll + (size_t)npairs * sizeof(ResultRec)) : (int*)h_poll;
      // one block of pairs: the poll also posts a sequence number behind its records, and the host spins on that word
      const bool spin = npairs <= 64 && results_on_host;
      int* h_seq = (int*)(h_poll + kHostResults * sizeof(ResultRec)) + 65538;
      poll_seq = spin ? poll_seq + 1 : poll_seq;
      // a single registration: the last kernel of the chunk's last tick writes the poll itself (post_result), no k_finalize
      // (Levenberg-Marquardt: k_error ends the tick)
      const bool fold_poll = lm && npairs == 1 && spin && ng == 1;
      if (fold_poll) {  // where the record goes: the same for every poll of this handle (uploaded when it changes)
        const PollPost post{d_results.as<ResultRec>(), d_stat, (ResultRec*)h_poll_dev, (int*)(h_poll_dev + ((char*)h_status - h_poll)),
                            (int*)(h_poll_dev + ((char*)h_seq - h_poll)), d_errflag.as<int>()};
        APD_TRY(d_post.upload(&post, sizeof(post), stream));
      }
      {
        roctx_range rr("apdgicp:ticks");
        struct FlagReset {  // also on the error returns inside the loop
          bool &a, &b;
          ~FlagReset() { a = b = false; }
        } flag_reset{post_tick, init_tick};
        for (int t = 0; t < todo; t++, cur_tick++) {
          post_tick = fold_poll && t == todo - 1;
          init_tick = fold_init && cur_tick == 0;  // (only here: the probe entry points launch the same kernels)
          for (int g = 0; g < ng; g++) {
            const int p0 = (int)((long long)n_active * g / ng), p1 = (int)((long long)n_active * (g + 1) / ng);
            APD_TRY(launch_tick(Span{p0, p1 - p0, g == 0 ? stream : gstreams[g - 1]}));
          }
          if (work.blk_cost && !work.blk_order && cur_tick == 1) {  // behind the second tick (the first warm search): its costs order the rest
            const int nn_blocks = (nmax_src + 63) / 64;
            hipLaunchKernelGGL(k_block_order, dim3(1), dim3(1024), 0, stream, work.blk_cost, nn_blocks, b_blkorder.as<unsigned>());
            work.blk_order = b_blkorder.as<unsigned>();
          }
        }
      }
      cur_active = 0;
      for (int g = 1; g < ng; g++) {
        APD_HIP(hipEventRecord(gevents[g - 1], gstreams[g - 1]));
        APD_HIP(hipStreamWaitEvent(stream, gevents[g - 1], 0));
      }
      ticks += todo;
      if (!fold_poll)
      hipLaunchKernelGGL(k_finalize, dim3((npairs + 63) / 64), dim3(64), 0, stream, d_state.as<PairState>(), d_results.as<ResultRec>(), d_stat, npairs,
                         d_errflag.as<int>(), results_on_host ? (ResultRec*)h_poll_dev : (ResultRec*)nullptr,
                         (int*)(h_poll_dev + ((char*)h_status - h_poll)), spin ? (int*)(h_poll_dev + ((char*)h_seq - h_poll)) : (int*)nullptr, poll_seq);
      APD_HIP(hipEventRecord(ev_poll, stream));
      if (n_stage_pending) {  // pinned clouds read by a sort in front of this event
        for (Cloud& c : clouds)
          if (c.stage_pending && !c.stage_wait) {
            c.stage_wait = ev_poll;
            if (spin) c.stage_seq_word = (volatile int*)h_seq, c.stage_seq_val = poll_seq;
          }
        n_stage_pending = 0;
      }
      pending_spin = spin;
      if (defer_poll && ticks == tick_cap && ticks == todo) {  // the only chunk: nothing on the host depends on its outcome
        pending = true, pending_npairs = npairs, pending_ticks = (int)ticks;
        work.active = nullptr;
        APD_HIP(hipGetLastError());
        return 0;
      }
      {
        roctx_range rr("apdgicp:poll");
        APD_TRY(wait_poll());
      }
      h_active.clear();
      for (int p = 0; p < npairs; p++)
        if (h_status[p] != ST_DONE) h_active.push_back(p);
      all_done = h_active.empty();
      if (!all_done && (int)h_active.size() < n_active) {  // from now on launch over the pairs that still run
        APD_TRY(d_active.upload(h_active.data(), h_active.size() * sizeof(int), stream));
        work.active = d_active.as<int>();
        n_active = (int)h_active.size();
      }
      if (h_status[npairs]) {
        const int flag = h_status[npairs];
        tickets_dirty = true;
        APD_HIP(hipMemsetAsync(d_errflag.p, 0, sizeof(int), stream));
        work.active = nullptr;
        return fail(APDGICP_ERR_INTERNAL, errflag_text(flag));
      }
    }
    work.active = nullptr;
    last_ticks = (int)ticks;
    if (ticks == 0) {  // max_iterations <= 0: no tick, no poll -- the records of the initial state
      APD_TRY(check_errflag("k_knn_cov"));
      results_on_host = false;
      hipLaunchKernelGGL(k_finalize, dim3((npairs + 63) / 64), dim3(64), 0, stream, d_state.as<PairState>(), d_results.as<ResultRec>(), (int*)nullptr,
                         npairs, (const int*)nullptr, (ResultRec*)nullptr, (int*)nullptr);
    }
    APD_HIP(hipGetLastError());
    if (profile_nn) {
      APD_HIP(hipStreamSynchronize(stream));
      APD_TRY(collect_nn_profile());
    }
    return 0;
  }

  // ------------------------------------------------------------------ pooled Levenberg-Marquardt batches
  // (see k_pool_poll in apd_kernels.hpp for the device side)  A batch handle whose optimiser is LM keeps the pairs of up to
  // `lanes` batches in one pool of pair slots; enqueue prepares the batch's clouds on the cloud stream, hands its pair
  // descriptors and guesses to the lane and returns; the ticks -- every launch over the device-side list of running pairs --
  // are enqueued in chunks of a few, always `depth` chunks ahead of the header the host has seen, by whichever call of the
  // handle is running (enqueue tops up, collect pumps until its batch is done).  A cloud slot referenced by a batch in flight
  // must not be replaced: set_cloud on such a slot first waits for that batch.
  struct PoolJob {
    enum State { FREE, PENDING, RUNNING, DONE };
    State state = FREE;
    uint64_t ticket = 0, admit_seq = 0;
    int np = 0;
    int list = 0;                     // the pair list (tick stream) the batch runs on
    bool collected = false;
    int err = 0;
    std::string errmsg;
    std::vector<int> cloud_ids;       // (unique) clouds the batch reads
    std::vector<std::pair<int, int>> pair_ids;
    std::vector<ResultRec> recs;      // host copy of the records, taken when the batch completed
    hipEvent_t ev_pro = nullptr;      // behind the preparation of its clouds (cloud stream)
    char* pin = nullptr;              // staging of the lane's pair descriptors and guesses
    size_t pin_cap = 0;
    int layout_gen = 0;               // Pool::layout_gen when the batch was enqueued: where its device records are
    DevBuf d_recs;                    // device copy of `recs` for a collect that comes after the pool was laid out anew
  };
  struct Pool {
    bool on = false;
    bool layout_valid = false;
    int layout_gen = 0;  // counts the layouts: a new one moves (or frees) the record segments of every earlier batch
    int lanes = 0, segcap = 0, cap = 0, nmax_src = 0, nmax_tgt = 0;
    hipStream_t cstream = nullptr;
    DevBuf state, pairs, guess, results, ticket, nnpart, corr, nnpt, nnaux, sqd, maha, blkpart, errpart;
    Work work{};
    char* host = nullptr;  // pinned: ResultRec[cap], then PoolHdr[kLists][kPoolRing]
    char* host_dev = nullptr;
    size_t host_cap = 0;
    PoolJob jobs[kPoolLanes];
    // TWO independent pair lists, each with its own tick stream, poll kernel, header ring and chunk numbering; a batch lives on
    // one of them (the less loaded one when it is enqueued).  Nothing ever waits across the two streams: the poll of a list is
    // a bubble only on its own stream, while the other list ticks.  (Until round 4: ONE list, cut into two slices on two
    // streams that forked behind every poll and joined in front of the next -- 13 % of the time no kernel ran at all.  Two pooled
    // handles on two host threads showed what independence is worth, tools/lm_threads.py: 1.03 -> 0.96 ms per batch of 32 loop
    // pairs with sixteen batches in flight, 0.91 with thirty-two.)
    static constexpr int kMaxLists = 4;
    int nlists = 2;  // APDGICP_POOL_LISTS (1 .. 4), read when the pool is first entered
    struct List {
      DevBuf active, nactive;          // the device-side list of running pairs and its length
      hipStream_t st = nullptr;        // list 0: the engine's stream; list 1: gstreams[0]
      hipEvent_t ev[kPoolRing] = {};   // behind the poll of chunk seq % kPoolRing
      uint64_t seq_enq = 0, seq_seen = 0;
      int adm[kPoolRing] = {};
      int ub = 0;                      // upper bound of the device's list length behind the last ENQUEUED poll
      int kill_mask = 0;
    } L[kMaxLists];
    int cur = 0;                       // the list whose ticks are being launched (t_work)
    // TWO cloud lanes (round 6): the preparation of consecutive batches -- pack, sort, covariances -- alternates between two streams, each
    // with its own job tables (descriptor table, id list, pack / sort / tile jobs, tile keys, bulk staging), so that the chain of batch
    // n + 1 (five launches of which four leave most of the GPU empty) runs beside the covariance launch of batch n.  The engine's
    // d_desc / d_ids / ... members always are those of the CURRENT lane; pool_swap_cloud_lane() exchanges them with the parked set.
    int nclanes = 1;                   // APDGICP_POOL_CLOUD_STREAMS (1 or 2)
    int clane = 0;                     // the current lane
    hipStream_t cstream2 = nullptr;    // the parked lane's stream
    hipEvent_t ev_cross = nullptr;     // orders one lane behind the other when a batch touches a cloud the other lane prepared
    uint64_t desc_gen_lane[2] = {0, 0};
    int ticks_per_chunk = 1;  // APDGICP_POOL_TICKS
    // measured (tools/pool_sweep.sh, docs/experiments.md): chunks two deep, the list cut into two slices from 24 pairs on
    static constexpr int groups = 2, group_min = 24;
    int depth = 1;  // APDGICP_POOL_DEPTH: chunks enqueued ahead of the last header seen, per list
    int last_lane = -1;
    long long n_chunks = 0, n_ticks = 0, n_pair_ticks = 0;  // statistics (apdgicp_batch_last_ticks)
    std::vector<int> cloud_busy;
    // Timed search launches (profile_nn): the first tick of every profile_stride-th chunk carries its own start / stop events.  The
    // pairs such a launch really covered are known when the chunk's header arrives (n_active: the list length its ticks ran over,
    // the rest of the launch's slots held -1), its duration once the NEXT header has arrived (the poll behind the chunk's ticks).
    struct Timed {
      hipEvent_t e0 = nullptr, e1 = nullptr;
      uint64_t seq = 0;
      int list = 0, p0 = 0, p1 = 0, n_active = -1;
      bool busy = false;
    };
    std::vector<Timed> timed;
    Timed* cur_timed = nullptr;      // set around the launch_nn that is to be timed
    const char* timed_kernel = "";   // the search kernel of the largest timed launch
    int timed_kernel_np = 0;
    double nn_ms = 0;                // harvested since the last read (apdgicp_batch_last_nn_profile)
    long long nn_launches = 0, nn_pairs = 0;
  } pool;

  bool pool_eligible() const {
    const bool enabled = env_int("APDGICP_LM_POOL", 1) != 0;  // (0: the host-polled loop of run_align, the cross-check)
    return enabled && params.optimizer == APDGICP_OPT_LM && params.max_iterations > 0 && nn_pruned;
  }
  // batches of one handle that may be in flight at once (round 4, two lists: 8 / 16 / 24 / 32 in flight: 1.08 / 0.91 / 0.87 / 0.86 ms per 32 loop
  // pairs; round 3, one list: 8 / 12 / 16 / 24 / 32: 1.27 / 1.13 / 1.07 / 1.03 / 1.01)
  static int pool_lanes_cfg() { return std::max(1, std::min(kPoolLanes, env_int("APDGICP_POOL_LANES", 24))); }
  // the parked cloud lane's tables (the current lane's are the engine's own members)
  struct ParkedLane {
    CachedTable desc, ids, packjobs, sortjobs, tilejobs[3];
    DevBuf tkeys, bulk_dev;
    char* bulk_host = nullptr;
    size_t bulk_cap = 0;
    hipEvent_t bulk_ev = nullptr;
  } lane2;
  uint64_t desc_gen = 0;  // counts the rebuilds of h_desc: a lane's device table is current when it has uploaded this generation
  void pool_swap_cloud_lane() {
    if (pool.nclanes < 2 || !pool.on) return;
    d_desc.swap(lane2.desc), d_ids.swap(lane2.ids), d_packjobs.swap(lane2.packjobs), d_sortjobs.swap(lane2.sortjobs);
    for (int i = 0; i < 3; i++) d_tilejobs[i].swap(lane2.tilejobs[i]);
    std::swap(d_tkeys, lane2.tkeys), std::swap(bulk_dev, lane2.bulk_dev);
    std::swap(bulk_host, lane2.bulk_host), std::swap(bulk_cap, lane2.bulk_cap), std::swap(bulk_ev, lane2.bulk_ev);
    std::swap(pool.cstream, pool.cstream2);
    cstream = pool.cstream;
    pool.clane ^= 1;
  }
  // In front of operations on clouds on the current cloud stream: `touch` every cloud involved, then `lane_order`.  A cloud whose last
  // operation went to the OTHER lane (and is not known to be complete) puts the current lane behind everything the other lane holds.
  bool lane_cross = false;
  void lane_touch(Cloud& c) {
    if (pool.on && pool.nclanes > 1) {
      if (c.prep_lane >= 0 && c.prep_lane != pool.clane) lane_cross = true;
      c.prep_lane = pool.clane;
    } else {
      c.prep_lane = -1;  // (one cloud stream: stream order)
    }
  }
  int lane_order() {
    if (!lane_cross) return 0;
    lane_cross = false;
    APD_HIP(hipEventRecord(pool.ev_cross, pool.cstream2));
    APD_HIP(hipStreamWaitEvent(pool.cstream, pool.ev_cross, 0));
    return 0;
  }
  PoolHdr* pool_hdr(int l, uint64_t seq) const { return (PoolHdr*)(pool.host + (size_t)pool.cap * sizeof(ResultRec)) + (size_t)l * kPoolRing + seq % kPoolRing; }
  bool pool_busy() const {
    for (const PoolJob& j : pool.jobs)
      if (j.state == PoolJob::PENDING || j.state == PoolJob::RUNNING) return true;
    return false;
  }

  int pool_enter() {
    if (pool.on) return 0;
    APD_HIP(hipStreamSynchronize(stream));  // whatever the clouds went through on the main stream so far
    if (!pool.cstream) {
      // experiments (round 6, docs/experiments.md): the cloud stream at the lowest stream priority and / or confined to a share of the CUs,
      // so that the tick launches -- the chain the pool's throughput hangs on (their streams are 96 % busy, the cloud stream 77 %) -- get
      // the GPU first.  APDGICP_POOL_CLOUD_PRIO=1, APDGICP_POOL_CLOUD_CUS=<CUs of 256>; default: neither
      auto make_cloud_stream = [&](hipStream_t* out) -> int {
        const int cus = std::max(0, std::min(256, env_int("APDGICP_POOL_CLOUD_CUS", 0)));
        if (cus > 0) {
          uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
          for (int b = 0; b < cus; b++) mask[b >> 5] |= 1u << (b & 31);
          APD_HIP(hipExtStreamCreateWithCUMask(out, 8, mask));
          return 0;
        }
        if (env_int("APDGICP_POOL_CLOUD_PRIO", 0)) {
          int least = 0, greatest = 0;
          APD_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
          APD_HIP(hipStreamCreateWithPriority(out, hipStreamNonBlocking, least));
          return 0;
        }
        APD_HIP(hipStreamCreateWithFlags(out, hipStreamNonBlocking));
        return 0;
      };
      APD_TRY(make_cloud_stream(&pool.cstream));
      for (Pool::List& li : pool.L)
        for (int i = 0; i < kPoolRing; i++) APD_HIP(hipEventCreateWithFlags(&li.ev[i], hipEventDisableTiming));
      // measured (round 6, docs/experiments.md): lists 2 / 3 / 4 and cloud streams 1 / 2 on the C4 shard
      pool.nlists = std::max(1, std::min(Pool::kMaxLists, env_int("APDGICP_POOL_LISTS", 2)));
      pool.nclanes = std::max(1, std::min(2, env_int("APDGICP_POOL_CLOUD_STREAMS", 1)));
      if (pool.nclanes > 1) {
        APD_TRY(make_cloud_stream(&pool.cstream2));
        APD_HIP(hipEventCreateWithFlags(&pool.ev_cross, hipEventDisableTiming));
      }
      APD_TRY(ensure_group_streams(pool.nlists + 1));  // gstreams[l - 1]: list l; gstreams[nlists - 1]: the slice stream of a list that ticks alone
      pool.L[0].st = stream;
      for (int l = 1; l < pool.nlists; l++) pool.L[l].st = gstreams[l - 1];
      for (PoolJob& j : pool.jobs) APD_HIP(hipEventCreateWithFlags(&j.ev_pro, hipEventDisableTiming));
      // measured on the two-list pool (docs/experiments.md, round 4): ticks per chunk 1 / 2 / 3 / 4 -> 0.931 / 0.962 / 1.018 / 1.065 ms per batch of
      // 32 loop pairs (16 in flight, two chunks ahead); chunks ahead 1 / 2 / 3 / 4 -> 0.906 / 0.933 / 0.951 / 0.970 (one tick per chunk): a poll
      // is a bubble on its own list's stream only, and every tick enqueued past a pair's last one is three empty launches
      pool.ticks_per_chunk = std::max(1, std::min(16, env_int("APDGICP_POOL_TICKS", 1)));
      pool.depth = std::max(1, std::min(kPoolRing - 2, env_int("APDGICP_POOL_DEPTH", 1)));
      profile_stride = std::max(1, env_int("APDGICP_PROFILE_STRIDE", 10));

    }
    cstream = pool.cstream;
    pool.on = true;
    for (Cloud& c : clouds) c.prep_lane = -1;  // (the main stream was synchronised above)
    return 0;
  }
  // back to one stream: every other entry point (Gauss-Newton batches, fitness, probes) prepares clouds and ticks in stream order
  int pool_leave() {
    if (!pool.on) return 0;
    APD_TRY(pool_drain());
    APD_HIP(hipStreamSynchronize(pool.cstream));
    if (pool.cstream2) APD_HIP(hipStreamSynchronize(pool.cstream2));
    if (pool.clane != 0) pool_swap_cloud_lane();  // (the engine's own tables are lane 0's again)
    for (Cloud& c : clouds) c.prep_lane = -1;
    for (Pool::List& li : pool.L)
      if (li.st) APD_HIP(hipStreamSynchronize(li.st));
    APD_HIP(hipStreamSynchronize(stream));
    for (Cloud& c : clouds) c.stage_pending = false, c.stage_wait = nullptr, c.stage_seq_word = nullptr;  // (every sort has run)
    n_stage_pending = 0;
    cstream = stream;
    pool.on = false;
    return 0;
  }

  int pool_layout(int n, int nsrc, int ntgt) {
    pool.nmax_tgt = std::max(pool.nmax_tgt, ntgt);
    if (pool.layout_valid && n <= pool.segcap && nsrc <= pool.nmax_src) return 0;
    APD_TRY(pool_drain());
    // every stream that may still run a poll or a no-op tick of the old layout (list 1 ticks on gstreams[0], a list that ticks alone
    // is sliced onto gstreams[1]): the pinned headers are wiped and the list buffers possibly re-allocated below
    for (Pool::List& li : pool.L)
      if (li.st) APD_HIP(hipStreamSynchronize(li.st));
    for (hipStream_t gs : gstreams) APD_HIP(hipStreamSynchronize(gs));
    APD_HIP(hipStreamSynchronize(stream));
    const int segcap = std::max(pool.segcap, n);
    const int nmax = std::max(pool.nmax_src, nsrc);
    const size_t ns = ((size_t)nmax + 255) & ~(size_t)255, nblk = (nmax + LIN_BLK - 1) / LIN_BLK;
    const size_t per_pair = ns * 96 + nblk * (kRed + 1) * 8 + sizeof(PairState) + sizeof(PairDesc) + sizeof(Rigid) + 2 * sizeof(ResultRec) + 16;
    int lanes = pool_lanes_cfg();
    while (lanes > 1 && (size_t)lanes * segcap * per_pair > ((size_t)16 << 30)) lanes--;
    const int cap = lanes * segcap;
    APD_TRY(pool.state.ensure((size_t)cap * sizeof(PairState)));
    APD_TRY(pool.pairs.ensure((size_t)cap * sizeof(PairDesc)));
    APD_TRY(pool.guess.ensure((size_t)cap * sizeof(Rigid)));
    for (Pool::List& li : pool.L) {
      APD_TRY(li.active.ensure((size_t)cap * sizeof(int)));
      APD_TRY(li.nactive.ensure(sizeof(int)));
    }
    APD_TRY(pool.results.ensure((size_t)cap * sizeof(ResultRec)));
    APD_TRY(pool.ticket.ensure((size_t)2 * cap * sizeof(int)));
    APD_TRY(pool.nnpart.ensure((size_t)cap * ns * 8));
    APD_TRY(pool.corr.ensure((size_t)cap * ns * 4));
    APD_TRY(pool.nnpt.ensure((size_t)cap * ns * 16));
    APD_TRY(pool.nnaux.ensure((size_t)cap * ns * 16));
    APD_TRY(pool.sqd.ensure((size_t)cap * ns * 4));
    APD_TRY(pool.maha.ensure((size_t)cap * 6 * ns * 8));
    APD_TRY(pool.blkpart.ensure((size_t)cap * nblk * kRed * 8));
    APD_TRY(pool.errpart.ensure((size_t)cap * nblk * 8));
    const size_t host_bytes = (size_t)cap * sizeof(ResultRec) + (size_t)Pool::kMaxLists * kPoolRing * sizeof(PoolHdr);
    if (host_bytes > pool.host_cap) {
      if (pool.host) APD_HIP(hipHostFree(pool.host));
      pool.host = pool.host_dev = nullptr, pool.host_cap = 0;
      APD_HIP(hipHostMalloc((void**)&pool.host, host_bytes, hipHostMallocDefault));
      APD_HIP(hipHostGetDevicePointer((void**)&pool.host_dev, pool.host, 0));
      pool.host_cap = host_bytes;
    }
    memset(pool.host, 0, pool.host_cap);  // (also the sequence words: a header counts once its word equals the expected number, never 0)
    for (Pool::List& li : pool.L) {
      APD_HIP(hipMemsetAsync(li.nactive.p, 0, sizeof(int), stream));
      APD_HIP(hipMemsetAsync(li.active.p, 0xff, (size_t)cap * sizeof(int), stream));
    }
    APD_HIP(hipStreamSynchronize(stream));
    Work& w = pool.work;
    w = Work{};
    w.T = 1;
    w.nstride = (int)ns, w.nblk_max = (int)nblk;
    w.cap = std::numeric_limits<float>::infinity();
    w.nnpart = pool.nnpart.as<unsigned long long>();
    w.corr = pool.corr.as<int>();
    w.nnpt = pool.nnpt.as<float4>();
    w.nnaux = nn_skin ? pool.nnaux.as<float4>() : nullptr;
    w.skin_mul = (1.f + nn_skin_rel) * (1.f + nn_skin_rel);
    w.skin_add = nn_skin_abs * nn_skin_abs;
    w.sqd = pool.sqd.as<float>();
    w.maha = pool.maha.as<double>();
    w.blkpart = pool.blkpart.as<double>();
    w.errpart = pool.errpart.as<double>();
    w.stats = d_stats.as<unsigned long long>();
    w.ticket = pool.ticket.as<int>();
    w.coop_search = 1;
    w.sparse_max = nn_sparse;
    w.pair0 = 0, w.npairs = cap;
    w.active = pool.L[0].active.as<int>();  // (t_work puts in the list that is being launched)
    nn_S = 1;  // (one source point per lane: the launch shape setup_pairs chooses for the pruned search)
    pool.lanes = lanes, pool.segcap = segcap, pool.cap = cap, pool.nmax_src = nmax;
    pool.layout_gen++;  // (the batches that finished under the old layout keep their host records; pool_collect knows by this number)
    for (Pool::Timed& c : pool.timed) c.busy = false;
    for (Pool::List& li : pool.L) {  // (nothing is in flight; headers of older chunks were wiped)
      li.ub = 0, li.seq_seen = li.seq_enq, li.kill_mask = 0;
      for (int& a : li.adm) a = 0;
    }
    pool.layout_valid = true;
    return 0;
  }

  void pool_job_finished(PoolJob& j) {
    j.state = PoolJob::DONE;
    for (int id : j.cloud_ids) {
      if (id < (int)pool.cloud_busy.size() && pool.cloud_busy[id] > 0) pool.cloud_busy[id]--;
      // (its ticks ran behind the event behind its clouds' preparation: whatever lane that was on, it is complete -- unless the cloud has
      // not been given covariances yet by a LATER enqueue, which cannot be: a cloud a batch in flight reads is never replaced)
      if (id < (int)clouds.size() && clouds[id].cov_valid && clouds[id].sorted) clouds[id].prep_lane = -1;
    }
  }

  // one chunk of list l: the poll (completions of everything enqueued before, admissions) and ticks_per_chunk ticks over the list
  int pool_enqueue_chunk(int l) {
    Pool::List& li = pool.L[l];
    PoolAdmit adm{};
    int admitted = 0;
    // A pending batch joins when its clouds are ready (sorted, covariances computed -- on the cloud stream).  While pairs are
    // still ticking the tick stream does not WAIT for that: the batch stays pending and is looked at again in front of the next
    // chunk.  (Every batch used to be admitted by the first chunk after its enqueue, behind a stream wait for its covariance
    // launch: 0.4 ms without a tick per batch, the tick stream 75 % busy.)
    for (int ln = 0; ln < pool.lanes; ln++) {
      PoolJob& j = pool.jobs[ln];
      if (j.state != PoolJob::PENDING || j.list != l) continue;
      const hipError_t ready = hipEventQuery(j.ev_pro);
      if (ready == hipErrorNotReady) {
        if (li.ub > 0 || admitted > 0) continue;
        APD_HIP(hipStreamWaitEvent(li.st, j.ev_pro, 0));  // nothing else to do: the ticks wait for these clouds
      } else {
        APD_HIP(ready);
      }
      adm.seg0[adm.count] = ln * pool.segcap, adm.np[adm.count] = j.np, adm.count++;
      j.state = PoolJob::RUNNING, j.admit_seq = li.seq_enq + 1;
      admitted += j.np;
    }
    adm.kill_mask = li.kill_mask, li.kill_mask = 0;
    const uint64_t seq = ++li.seq_enq;
    li.adm[seq % kPoolRing] = admitted;
    PoolHdr* hdr_dev = (PoolHdr*)(pool.host_dev + (size_t)pool.cap * sizeof(ResultRec)) + (size_t)l * kPoolRing + seq % kPoolRing;
    hipLaunchKernelGGL(k_pool_poll, dim3(1), dim3(256), 0, li.st, pool.state.as<PairState>(), li.active.as<int>(), li.nactive.as<int>(), pool.cap,
                       pool.segcap, adm, pool.guess.as<Rigid>(), params.max_iterations, pool.ticket.as<int>(), pool.results.as<ResultRec>(),
                       (ResultRec*)pool.host_dev, hdr_dev, (int)seq, d_errflag.as<int>());
    APD_HIP(hipEventRecord(li.ev[seq % kPoolRing], li.st));
    li.ub += admitted;
    pool.n_chunks++;
    if (li.ub > 0) {
      roctx_range rr("apdgicp:pool_ticks");
      struct Reset {
        Engine& e;
        ~Reset() { e.in_pool = false, e.cur_active = 0; }
      } reset{*this};
      in_pool = true;
      pool.cur = l;
      // A list that ticks ALONE (a single batch in flight, or its sister list is empty) is cut into two slices on two streams
      // between its polls, like the one list of round 3: its head holds few pairs with many iterations to go, a latency-bound
      // chain of small launches, its tail the young wide ones.  With both lists busy every list is one stream of its own.
      int other = 0;
      for (int o = 0; o < pool.nlists; o++)
        if (o != l) other += pool.L[o].ub;
      const int G = other == 0 && li.ub >= pool.group_min ? std::max(1, std::min(pool.groups, env_int("APDGICP_POOL_GROUPS", pool.groups))) : 1;
      hipStream_t slice = gstreams[pool.nlists - 1];
      if (G > 1) {
        APD_HIP(hipEventRecord(ev_main, li.st));
        APD_HIP(hipStreamWaitEvent(slice, ev_main, 0));
      }
      for (int g = 0; g < G; g++) {
        const int p0 = (int)((long long)li.ub * g / G), p1 = (int)((long long)li.ub * (g + 1) / G);
        if (p1 <= p0) continue;
        const int nt = pool.ticks_per_chunk;
        cur_active = p1 - p0;
        for (int t = 0; t < nt; t++) {
          pool.cur_timed = nullptr;
          // (which tick of the chunk: in turn -- the first one alone would over-represent the cold searches of pairs just admitted)
          if (profile_nn && (int)(seq % (uint64_t)profile_stride) == 0 && t == (int)((seq / (uint64_t)profile_stride) % (uint64_t)nt)) {
            Pool::Timed* slot = nullptr;
            for (Pool::Timed& c : pool.timed)
              if (!c.busy) slot = &c;
            if (!slot && pool.timed.size() < 64) {
              pool.timed.emplace_back();
              slot = &pool.timed.back();
              APD_HIP(hipEventCreate(&slot->e0));
              APD_HIP(hipEventCreate(&slot->e1));
            }
            if (slot) slot->busy = true, slot->list = l, slot->seq = seq, slot->p0 = p0, slot->p1 = p1, slot->n_active = -1, pool.cur_timed = slot;
          }
          const int rc_t = launch_tick(Span{p0, p1 - p0, g == 0 ? li.st : slice});
          pool.cur_timed = nullptr;
          APD_TRY(rc_t);
        }
        pool.n_pair_ticks += (long long)nt * (p1 - p0);
      }
      if (G > 1) {
        APD_HIP(hipEventRecord(gevents[pool.nlists - 1], slice));
        APD_HIP(hipStreamWaitEvent(li.st, gevents[pool.nlists - 1], 0));
      }
      pool.n_ticks += pool.ticks_per_chunk;
    }
    APD_HIP(hipGetLastError());
    return 0;
  }

  int pool_process(int l, uint64_t seq) {  // header `seq` of list l has arrived
    Pool::List& li = pool.L[l];
    std::atomic_thread_fence(std::memory_order_acquire);
    const PoolHdr h = *pool_hdr(l, seq);
    li.seq_seen = seq;
    for (Pool::Timed& c : pool.timed) {
      if (!c.busy || c.list != l) continue;
      if (c.seq == seq) c.n_active = h.n_active;
      if (c.seq < seq) {  // its chunk's ticks ran in front of this poll: the events have fired
        float ms = 0.f;
        if (c.n_active >= 0 && hipEventElapsedTime(&ms, c.e0, c.e1) == hipSuccess) {
          const int covered = std::max(0, std::min(c.p1, c.n_active) - c.p0);
          if (covered > 0) pool.nn_ms += ms, pool.nn_launches++, pool.nn_pairs += covered;
        }
        c.busy = false;
      }
    }
    int ub = h.n_active;
    for (uint64_t c = seq + 1; c <= li.seq_enq; c++) ub += li.adm[c % kPoolRing];
    li.ub = ub;
    if (h.errflag) {  // raised by a covariance launch: whose, the flag does not say -- every batch in flight fails
      for (int ln = 0; ln < pool.lanes; ln++) {
        PoolJob& j = pool.jobs[ln];
        if (j.state != PoolJob::PENDING && j.state != PoolJob::RUNNING) continue;
        j.err = APDGICP_ERR_INTERNAL, j.errmsg = errflag_text(h.errflag);
        if (j.state == PoolJob::PENDING) pool_job_finished(j);
        else pool.L[j.list].kill_mask |= 1 << ln;  // its pairs end with the next poll of its list
      }
    }
    for (int ln = 0; ln < pool.lanes; ln++) {
      PoolJob& j = pool.jobs[ln];
      if (j.state != PoolJob::RUNNING || j.list != l || j.admit_seq > seq || h.lane_left[ln] != 0) continue;
      j.recs.resize(j.np);
      memcpy(j.recs.data(), (const ResultRec*)pool.host + (size_t)ln * pool.segcap, (size_t)j.np * sizeof(ResultRec));
      pool_job_finished(j);
    }
    return 0;
  }

  int pool_topup() {
    for (;;) {
      bool any = false;
      for (int l = 0; l < pool.nlists; l++) {
        Pool::List& li = pool.L[l];
        if ((int)(li.seq_enq - li.seq_seen) >= pool.depth) continue;
        bool pending = li.kill_mask != 0;
        for (const PoolJob& j : pool.jobs) pending |= j.state == PoolJob::PENDING && j.list == l;
        if (li.ub <= 0 && !pending) continue;
        APD_TRY(pool_enqueue_chunk(l));
        any = true;
      }
      if (!any) return 0;
    }
  }

  // serves the pool: reads the headers that have arrived, keeps `depth` chunks enqueued per list; block: waits for one more header
  int pool_pump(bool block) {
    if (!pool.layout_valid) return block ? fail(APDGICP_ERR_INTERNAL, "pool: nothing in flight") : 0;
    APD_HIP(hipSetDevice(device));
    auto arrived = [&]() -> int {  // processes every header that is there; > 0 when there was one
      int n = 0;
      for (int l = 0; l < pool.nlists; l++) {
        Pool::List& li = pool.L[l];
        while (li.seq_seen < li.seq_enq && *(volatile int*)&pool_hdr(l, li.seq_seen + 1)->seq == (int)(li.seq_seen + 1)) {
          const int rc = pool_process(l, li.seq_seen + 1);
          if (rc < 0) return rc;
          n++;
        }
      }
      return n;
    };
    int got = arrived();
    if (got < 0) return got;
    APD_TRY(pool_topup());
    if (!block || got > 0) return 0;  // (a header that had arrived already may be the one the caller waits for: it looks again)
    bool outstanding = false;
    for (const Pool::List& li : pool.L) outstanding |= li.seq_seen < li.seq_enq;
    if (!outstanding) {
      std::string st;
      for (int ln = 0; ln < pool.lanes; ln++)
        st += " [" + std::to_string(ln) + ": state " + std::to_string((int)pool.jobs[ln].state) + " list " + std::to_string(pool.jobs[ln].list) + " ticket " +
              std::to_string(pool.jobs[ln].ticket) + " np " + std::to_string(pool.jobs[ln].np) + " admitted at " + std::to_string(pool.jobs[ln].admit_seq) + "]";
      std::string ls;
      for (int l = 0; l < pool.nlists; l++) ls += " [list " + std::to_string(l) + ": chunks " + std::to_string(pool.L[l].seq_enq) + ", bound " + std::to_string(pool.L[l].ub) + "]";
      return fail(APDGICP_ERR_INTERNAL, "pool: nothing in flight to wait for (" + ls + ";" + st + ")");
    }
    {
      roctx_range rr("apdgicp:pool_wait");
      const auto t0 = std::chrono::steady_clock::now();
      for (unsigned it = 0;; it++) {  // the next header of either list
        got = arrived();
        if (got != 0) break;
        if ((it & 255) == 255 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(600)) break;
      }
      if (got < 0) return got;
      if (got == 0) {  // a GPU that takes this long gets the sleeping wait: the older outstanding poll of the two lists
        int l = 0;
        while (l + 1 < pool.nlists && !(pool.L[l].seq_seen < pool.L[l].seq_enq)) l++;
        const uint64_t want = pool.L[l].seq_seen + 1;
        APD_HIP(hipEventSynchronize(pool.L[l].ev[want % kPoolRing]));
        if (*(volatile int*)&pool_hdr(l, want)->seq != (int)want) return fail(APDGICP_ERR_INTERNAL, "pool: a poll finished without posting its header");
        got = arrived();
        if (got < 0) return got;
      }
    }
    return pool_topup();
  }

  int pool_drain() {
    while (pool_busy()) APD_TRY(pool_pump(true));
    return 0;
  }
  // before a cloud slot is replaced: the batches in flight that read it
  int pool_release_clouds(int first, int count) {
    if (!pool.on) return 0;
    for (int q = first; q < first + count; q++)
      while (q >= 0 && q < (int)pool.cloud_busy.size() && pool.cloud_busy[q] > 0) APD_TRY(pool_pump(true));
    return 0;
  }

  int pool_enqueue(const apdgicp_pair* pairs, int64_t n, uint64_t* ticket) {
    if (n <= 0 || n > 65536) return fail(APDGICP_ERR_INVALID_ARG, "n_pairs must be in [1, 65536]");
    APD_HIP(hipSetDevice(device));
    APD_TRY(pool_enter());
    std::vector<int> need;
    int nsrc = 0, ntgt = 0;
    for (int64_t i = 0; i < n; i++) {
      const int s = pairs[i].source_cloud, t = pairs[i].target_cloud;
      if (s < 0 || t < 0 || s >= (int)clouds.size() || t >= (int)clouds.size() || clouds[s].n <= 0 || clouds[t].n <= 0)
        return fail(APDGICP_ERR_NO_INPUT, "pair references a cloud that is not set");
      need.push_back(s), need.push_back(t);
      nsrc = std::max(nsrc, clouds[s].n), ntgt = std::max(ntgt, clouds[t].n);
    }
    APD_TRY(pool_layout((int)n, nsrc, ntgt));
    APD_TRY(pool_pump(false));
    // a lane: a free one, else that of the oldest collected batch, else that of the oldest batch (waited for if it still runs;
    // its ticket is void from here on)
    int lane = -1, best_rank = 3;
    uint64_t best_ticket = ~0ull;
    for (int l = 0; l < pool.lanes; l++) {
      const PoolJob& c = pool.jobs[l];
      const int rank = c.state == PoolJob::FREE ? 0 : c.state == PoolJob::DONE && c.collected ? 1 : 2;
      if (rank < best_rank || (rank == best_rank && c.ticket < best_ticket)) lane = l, best_rank = rank, best_ticket = c.ticket;
    }
    // the pair list the batch joins: the one with fewer pairs pending or running
    int load[Pool::kMaxLists] = {0, 0, 0, 0};
    for (const PoolJob& c : pool.jobs)
      if (c.state == PoolJob::PENDING || c.state == PoolJob::RUNNING) load[c.list] += c.np;
    int list = 0;
    for (int l = 1; l < pool.nlists; l++)
      if (load[l] < load[list]) list = l;
    while (pool.jobs[lane].state == PoolJob::PENDING || pool.jobs[lane].state == PoolJob::RUNNING) APD_TRY(pool_pump(true));  // (the oldest batch: its lane is next)
    PoolJob& j = pool.jobs[lane];
    // the clouds of this batch, on the cloud stream: sort, covariances of those that lack them, descriptor table
    std::vector<int> ids;
    APD_TRY(filter_cov_ids(need, false, ids));
    // every cloud the batch reads: one the other lane prepared (a keyframe shared with the previous batch) puts this lane behind it, so
    // that the event below covers it
    for (int id : need) lane_touch(clouds[id]);
    APD_TRY(lane_order());
    APD_TRY(upload_desc());
    if (!ids.empty()) {
      APD_TRY(d_ids.upload(ids.data(), ids.size() * sizeof(int), cstream));
      APD_TRY(launch_knn(ids.data(), d_ids.as<int>(), (int)ids.size(), cstream));
    }
    APD_HIP(hipEventRecord(j.ev_pro, cstream));
    if (n_stage_pending) {  // pinned host clouds read by a sort in front of this event
      for (Cloud& c : clouds)
        if (c.stage_pending && !c.stage_wait) c.stage_wait = j.ev_pro, c.stage_seq_word = nullptr;
      n_stage_pending = 0;
    }
    // descriptors and guesses of the lane's pairs: staged in the lane's pinned buffer (its last reader, the copy of the lane's
    // previous batch, is long done), copied in stream order in front of the chunk that admits them
    const size_t bytes = (size_t)n * (sizeof(PairDesc) + sizeof(Rigid));
    if (bytes > j.pin_cap) {
      if (j.pin) APD_HIP(hipHostFree(j.pin));
      j.pin = nullptr, j.pin_cap = 0;
      APD_HIP(hipHostMalloc((void**)&j.pin, bytes * 2, hipHostMallocDefault));
      j.pin_cap = bytes * 2;
    }
    PairDesc* hp = (PairDesc*)j.pin;
    Rigid* hg = (Rigid*)(j.pin + (size_t)n * sizeof(PairDesc));
    j.pair_ids.resize(n);
    for (int64_t i = 0; i < n; i++) {
      const int s = pairs[i].source_cloud, t = pairs[i].target_cloud;
      hp[i].src = s, hp[i].tgt = t, hp[i].s = h_desc[s], hp[i].t = h_desc[t];
      j.pair_ids[i] = {s, t};
      for (int r = 0; r < 3; r++)
        for (int c = 0; c < 4; c++) hg[i].m[4 * r + c] = (double)pairs[i].guess[r + 4 * c];  // L:56 x0 = guess.cast<double>()
    }
    // (on the list's own stream: in front of the poll that admits them)
    APD_HIP(hipMemcpyAsync(pool.pairs.as<PairDesc>() + (size_t)lane * pool.segcap, hp, (size_t)n * sizeof(PairDesc), hipMemcpyHostToDevice, pool.L[list].st));
    APD_HIP(hipMemcpyAsync(pool.guess.as<Rigid>() + (size_t)lane * pool.segcap, hg, (size_t)n * sizeof(Rigid), hipMemcpyHostToDevice, pool.L[list].st));
    std::sort(need.begin(), need.end());
    need.erase(std::unique(need.begin(), need.end()), need.end());
    if (pool.cloud_busy.size() < clouds.size()) pool.cloud_busy.resize(clouds.size(), 0);
    for (int id : need) pool.cloud_busy[id]++;
    j.cloud_ids = need;
    j.state = PoolJob::PENDING, j.ticket = ++align_seq, j.np = (int)n, j.collected = false, j.err = 0, j.errmsg.clear(), j.admit_seq = 0, j.list = list;
    j.layout_gen = pool.layout_gen;
    pool.last_lane = lane;
    *ticket = j.ticket;
    pool_swap_cloud_lane();  // the next batch's clouds are prepared on the other cloud stream
    return pool_pump(false);
  }

  PoolJob* pool_find(uint64_t ticket) {
    for (int l = 0; l < kPoolLanes; l++)
      if (pool.jobs[l].state != PoolJob::FREE && pool.jobs[l].ticket == ticket) return &pool.jobs[l];
    return nullptr;
  }
  // waits for the batch of `ticket`; d_out / host_out as apdgicp_batch_align_collect
  int pool_collect(uint64_t ticket, void** d_out, apdgicp_result* host_out) {
    PoolJob* j = pool_find(ticket);
    if (!j) return fail(APDGICP_ERR_INVALID_ARG, "ticket is not one of the batches in flight (its lane has been reused)");
    while (j->state != PoolJob::DONE) APD_TRY(pool_pump(true));
    j->collected = true;
    last_ticks = (int)std::min<long long>(pool.n_ticks, 1 << 30);
    if (j->err) return fail(j->err, j->errmsg);
    if (d_out) {
      if (j->layout_gen == pool.layout_gen) {
        *d_out = pool.results.as<ResultRec>() + (size_t)(j - pool.jobs) * pool.segcap;
      } else {
        // A later batch had more pairs or larger clouds than the pool was laid out for: the layout (segment size, possibly the
        // number of lanes, the record buffer itself) is another one now and this batch's segment is gone.  Its records survive
        // on the host (taken when the batch completed): the caller gets a device copy of those, the job's own, valid until
        // the lane is reused like any other device record pointer.
        APD_TRY(j->d_recs.ensure((size_t)j->np * sizeof(ResultRec)));
        APD_HIP(hipMemcpy(j->d_recs.p, j->recs.data(), (size_t)j->np * sizeof(ResultRec), hipMemcpyHostToDevice));
        *d_out = j->d_recs.p;
      }
    }
    if (host_out) memcpy(host_out, j->recs.data(), (size_t)j->np * sizeof(ResultRec));
    return 0;
  }

  // ------------------------------------------------------------------ probes on pair 0
  int probe_linearize(const double T[16], double* H, double* b, double* cost, int* matched) {
    APD_HIP(hipSetDevice(device));
    APD_HIP(hipMemcpyAsync(d_T.p, T, 16 * sizeof(double), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(k_set_probe, dim3(1), dim3(1), 0, stream, d_state.as<PairState>(), d_T.as<double>(), (int)ST_NEED_LIN, 0);
    APD_TRY(launch_nn(whole()));
    APD_TRY(launch_linearize(whole(), H && b ? 1 : 0));
    hipLaunchKernelGGL(k_probe_reduce, dim3(1), dim3(64), 0, stream, d_desc.as<CloudDesc>(), d_pairs.as<PairDesc>(), d_state.as<PairState>(), work,
                       d_probe.as<double>(), 0);
    APD_HIP(hipMemcpyAsync(h_probe, d_probe.p, 44 * sizeof(double), hipMemcpyDeviceToHost, stream));
    APD_HIP(hipStreamSynchronize(stream));
    if (H && b) {
      memcpy(H, h_probe, 36 * sizeof(double));
      memcpy(b, h_probe + 36, 6 * sizeof(double));
    }
    if (cost) *cost = h_probe[42];
    if (matched) *matched = (int)h_probe[43];
    return 0;
  }

  int probe_error(const double T[16], double* cost) {
    APD_HIP(hipSetDevice(device));
    APD_HIP(hipMemcpyAsync(d_T.p, T, 16 * sizeof(double), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(k_set_probe, dim3(1), dim3(1), 0, stream, d_state.as<PairState>(), d_T.as<double>(), (int)ST_NEED_ERR, 1);
    APD_TRY(launch_error(whole(), false));
    hipLaunchKernelGGL(k_probe_reduce, dim3(1), dim3(64), 0, stream, d_desc.as<CloudDesc>(), d_pairs.as<PairDesc>(), d_state.as<PairState>(), work,
                       d_probe.as<double>(), 1);
    APD_HIP(hipMemcpyAsync(h_probe, d_probe.p, 44 * sizeof(double), hipMemcpyDeviceToHost, stream));
    APD_HIP(hipStreamSynchronize(stream));
    *cost = h_probe[42];
    return 0;
  }
};

}  // namespace apd
