// libapdgicp_hip.so -- the C ABI declared in include/apdgicp_hip.h, on top of apd::Engine.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -shared -fPIC (see build.py)
#include <cstddef>
#include <limits>
#include <new>

#include "apd_engine.hpp"
#include "apd_voxel.hpp"

using namespace apd;

struct apdgicp_handle {
  Engine eng;
  bool pair_ready = false;   // work buffers / descriptors match the current source+target
  bool have_corr = false;    // correspondences_/mahalanobis_ hold a linearize result
  int n_src_at_corr = 0;
  // apdgicp_set_trace: the trace of the last apdgicp_align_host_loop (apdgicp_align leaves its own on the device, Engine::d_trace)
  bool trace_from_host_loop = false;
  std::vector<double> tr_lambda, tr_rho, tr_y0, tr_yi, tr_dnorm, tr_poses;  // poses: 16 doubles each, column-major
};

struct apdgicp_batch {
  Engine eng;
  int64_t slot_pairs[2] = {0, 0};  // pairs of the last two enqueued batches (by ticket parity)
};

struct apdgicp_submap {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  DevBuf stage, cat, keys, pos, bsum, out, scal;  // scal: box6[6], total, err
  CachedTable jobs;
  int* h_scal = nullptr;  // pinned mirror of {total, err}
  int64_t n_last = 0;
  bool last_is_cat = false;  // no downsampling: the result is the concatenation itself
  ~apdgicp_submap() {
    if (h_scal) (void)hipHostFree(h_scal);
    for (DevBuf* b : {&stage, &cat, &keys, &pos, &bsum, &out, &scal, &jobs.dev}) b->release();
    if (own_stream && stream) (void)hipStreamDestroy(stream);
  }
};

namespace {

constexpr int kSrc = 0, kTgt = 1;

void identity16(float* g) {
  memset(g, 0, 16 * sizeof(float));
  g[0] = g[5] = g[10] = g[15] = 1.f;
}

// guess16: the initial guess the caller is about to align with (column-major; null: identity).  It goes into the pair
// table so that the guess buffer is uploaded once per call, not once with a placeholder and once with the real one.
int ensure_pair(apdgicp_handle* h, const float* guess16 = nullptr) {
  Engine& e = h->eng;
  if (e.clouds.size() < 2 || e.clouds[kSrc].n <= 0) return fail(APDGICP_ERR_NO_INPUT, "source cloud is not set");
  if (e.clouds[kTgt].n <= 0) return fail(APDGICP_ERR_NO_INPUT, "target cloud is not set");
  if (h->pair_ready && e.clouds[kSrc].cov_valid && e.clouds[kTgt].cov_valid) return 0;
  apdgicp_pair p;
  p.source_cloud = kSrc;
  p.target_cloud = kTgt;
  if (guess16) memcpy(p.guess, guess16, sizeof(p.guess));
  else identity16(p.guess);
  APD_TRY(e.setup_pairs(&p, 1, true));
  h->pair_ready = true;
  h->have_corr = false;
  return 0;
}


void rigid_to_colmajor(const Rigid& r, double* T) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 4; j++) T[i + 4 * j] = r.m[4 * i + j];
  T[3] = T[7] = T[11] = 0.0;
  T[15] = 1.0;
}

template <typename F>
int guarded(F&& f) {
  try {
    return f();
  } catch (const std::bad_alloc&) {
    return fail(APDGICP_ERR_INTERNAL, "out of host memory");
  } catch (...) {
    return fail(APDGICP_ERR_INTERNAL, "unexpected C++ exception");
  }
}

}  // namespace

extern "C" {

int apdgicp_abi_version(void) { return APDGICP_ABI_VERSION; }
// What this library was compiled with: the compiler flags build.py passed ("unknown" for a hand build), then " | variant:" and every
// experiment define that CHANGES RESULTS OR KERNELS -- detected here by the preprocessor, so no build script can leave one out.  A
// library that lists a variant is an ablation / A-B build (some are wrong by design): the Python loader and the C++ adapter's self-check
// refuse it unless APDGICP_ALLOW_VARIANT_LIB=1.
#ifndef APD_BUILD_FLAGS
#define APD_BUILD_FLAGS "unknown"
#endif
#define APD_STR2(x) #x
#define APD_STR(x) APD_STR2(x)
const char* apdgicp_build_flags(void) {
  return APD_BUILD_FLAGS " | variant:"
#ifdef APD_ABL_KNN_SKIP_C
      " APD_ABL_KNN_SKIP_C"
#endif
#ifdef APD_ABL_LM_NO_ERROR
      " APD_ABL_LM_NO_ERROR"
#endif
#ifdef APD_ABL_LIN_DOUBLE_ATAN
      " APD_ABL_LIN_DOUBLE_ATAN"
#endif
#ifdef APD_ABL_LIN_DOUBLE_SINCOS
      " APD_ABL_LIN_DOUBLE_SINCOS"
#endif
#ifdef APD_ABL_LIN_NO_ATAN
      " APD_ABL_LIN_NO_ATAN"
#endif
#ifdef APD_ABL_LIN_NO_SINCOS
      " APD_ABL_LIN_NO_SINCOS"
#endif
#ifdef APD_ABL_SEARCH_KEEP_AFTER
      " APD_ABL_SEARCH_KEEP_AFTER=" APD_STR(APD_ABL_SEARCH_KEEP_AFTER)
#endif
#ifdef APD_OCML_ATAN2F
      " APD_OCML_ATAN2F"
#endif
#ifdef APD_BLOCK_TIMELINE
      " APD_BLOCK_TIMELINE"
#endif
#ifdef APD_AB_NO_ASM_NOP
      " APD_AB_NO_ASM_NOP"
#endif
#ifdef APD_SINCOS_NO_TABLE
      " APD_SINCOS_NO_TABLE"
#endif
      ;
}
#ifndef APD_SOURCE_STAMP
#define APD_SOURCE_STAMP "unstamped"
#endif
// (build.py passes "apd-source-stamp:<16 hex digits>": the marker is what build.library_stamp() looks for in the file)
const char* apdgicp_source_stamp(void) {
  static const char stamp[] = APD_SOURCE_STAMP;
  const char* colon = strchr(stamp, ':');
  return colon ? colon + 1 : stamp;
}
const char* apdgicp_last_error(void) { return g_last_error.c_str(); }

int apdgicp_device_count(int* count) {
  if (!count) return fail(APDGICP_ERR_INVALID_ARG, "count is null");
  *count = 0;
  APD_HIP(hipGetDeviceCount(count));
  return 0;
}

void apdgicp_default_params(apdgicp_params* p) {
  if (!p) return;
  p->k_correspondences = 20;                 // A:21
  p->max_iterations = 64;                    // L:13
  p->lm_max_iterations = 10;                 // L:19
  p->optimizer = APDGICP_OPT_LM;             // L:17
  p->regularization = APDGICP_REG_PLANE;     // A:25
  p->flags = 0;
  p->max_correspondence_distance = (double)FLT_MAX;  // A:23
  p->transformation_epsilon = 5e-4;          // L:15
  p->rotation_epsilon = 2e-3;                // L:14
  p->lm_init_lambda_factor = 1e-9;           // L:20
  p->distance_variance = 0.86;               // H:109
  p->azimuth_variance_deg = 0.5;             // H:107
  p->elevation_variance_deg = 1.0;           // H:108
}

// ------------------------------------------------------------------------------------ single
int apdgicp_create(const apdgicp_params* p, int device, void* stream, apdgicp_handle** out) {
  return guarded([&]() -> int {
    if (!out) return fail(APDGICP_ERR_INVALID_ARG, "out is null");
    *out = nullptr;
    apdgicp_params dflt;
    apdgicp_default_params(&dflt);
    apdgicp_handle* h = new apdgicp_handle;
    const int rc = h->eng.init(p ? p : &dflt, device, stream);
    if (rc < 0) {
      delete h;
      return rc;
    }
    h->eng.clouds.resize(2);
    *out = h;
    return 0;
  });
}

int apdgicp_destroy(apdgicp_handle* h) {
  delete h;
  return 0;
}

int apdgicp_set_params(apdgicp_handle* h, const apdgicp_params* p) {
  if (!h) return fail(APDGICP_ERR_INVALID_ARG, "handle is null");
  return h->eng.set_params(p);
}

int apdgicp_get_params(const apdgicp_handle* h, apdgicp_params* p) {
  if (!h || !p) return fail(APDGICP_ERR_INVALID_ARG, "null argument");
  *p = h->eng.params;
  return 0;
}

static int set_cloud_common(apdgicp_handle* h, int slot, const float* xyz, int64_t n, int64_t stride, int on_device, uint64_t token) {
  return guarded([&]() -> int {
    if (!h) return fail(APDGICP_ERR_INVALID_ARG, "handle is null");
    Engine& e = h->eng;
    if (token != 0 && e.clouds[slot].n > 0 && e.clouds[slot].token == token) return 0;  // A:91-93 / A:102-104
    APD_TRY(e.set_cloud(slot, xyz, n, stride, on_device, token));
    h->pair_ready = false;
    h->have_corr = false;
    return 0;
  });
}

int apdgicp_set_source(apdgicp_handle* h, const float* xyz, int64_t n, int64_t stride_bytes, int on_device, uint64_t token) {
  return set_cloud_common(h, kSrc, xyz, n, stride_bytes, on_device, token);
}
int apdgicp_set_target(apdgicp_handle* h, const float* xyz, int64_t n, int64_t stride_bytes, int on_device, uint64_t token) {
  return set_cloud_common(h, kTgt, xyz, n, stride_bytes, on_device, token);
}

int apdgicp_clear_source(apdgicp_handle* h) {
  if (!h) return fail(APDGICP_ERR_INVALID_ARG, "handle is null");
  h->eng.clear_cloud(kSrc);
  h->pair_ready = h->have_corr = false;
  return 0;
}
int apdgicp_clear_target(apdgicp_handle* h) {
  if (!h) return fail(APDGICP_ERR_INVALID_ARG, "handle is null");
  h->eng.clear_cloud(kTgt);
  h->pair_ready = h->have_corr = false;
  return 0;
}
int apdgicp_swap_source_and_target(apdgicp_handle* h) {
  if (!h) return fail(APDGICP_ERR_INVALID_ARG, "handle is null");
  std::swap(h->eng.clouds[kSrc], h->eng.clouds[kTgt]);  // input_.swap(target_), covs swap, A:68-75
  h->eng.desc_dirty = true;
  h->pair_ready = h->have_corr = false;
  return 0;
}

int apdgicp_compute_covariances(apdgicp_handle* h, int which) {
  return guarded([&]() -> int {
    if (!h || (which != kSrc && which != kTgt)) return fail(APDGICP_ERR_INVALID_ARG, "bad argument");
    return h->eng.compute_covariances({which});
  });
}

int apdgicp_get_covariances(apdgicp_handle* h, int which, double* out, int64_t n) {
  return guarded([&]() -> int {
    if (!h || !out || (which != kSrc && which != kTgt)) return fail(APDGICP_ERR_INVALID_ARG, "bad argument");
    Engine& e = h->eng;
    APD_TRY(e.compute_covariances({which}));
    Engine::Cloud& c = e.clouds[which];
    if (n != c.n) return fail(APDGICP_ERR_INVALID_ARG, "n does not match the cloud size");
    APD_HIP(hipStreamSynchronize(e.stream));
    APD_TRY(e.d_stage.ensure((size_t)n * 16 * sizeof(double)));
    hipLaunchKernelGGL(k_unpack_cov, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, e.stream, c.cov.as<double>(), c.perm.as<int>(), (int)n,
                       e.d_stage.as<double>());
    APD_HIP(hipMemcpyAsync(out, e.d_stage.p, (size_t)n * 16 * sizeof(double), hipMemcpyDeviceToHost, e.stream));
    APD_HIP(hipStreamSynchronize(e.stream));
    return 0;
  });
}

int apdgicp_set_covariances(apdgicp_handle* h, int which, const double* in, int64_t n) {
  return guarded([&]() -> int {
    if (!h || !in || (which != kSrc && which != kTgt)) return fail(APDGICP_ERR_INVALID_ARG, "bad argument");
    Engine& e = h->eng;
    Engine::Cloud& c = e.clouds[which];
    if (c.n <= 0) return fail(APDGICP_ERR_NO_INPUT, "cloud not set");
    if (n != c.n) return fail(APDGICP_ERR_INVALID_ARG, "n does not match the cloud size");
    APD_TRY(e.upload_desc());  // allocates c.cov
    APD_HIP(hipStreamSynchronize(e.stream));
    APD_TRY(e.d_stage.ensure((size_t)n * 16 * sizeof(double)));
    APD_HIP(hipMemcpyAsync(e.d_stage.p, in, (size_t)n * 16 * sizeof(double), hipMemcpyHostToDevice, e.stream));
    hipLaunchKernelGGL(k_pack_cov, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, e.stream, e.d_stage.as<double>(), c.perm.as<int>(), (int)n,
                       c.cov.as<double>());
    APD_HIP(hipStreamSynchronize(e.stream));
    c.cov_valid = true;
    return 0;
  });
}

int apdgicp_linearize(apdgicp_handle* h, const double T[16], double H[36], double b[6], double* cost) {
  return guarded([&]() -> int {
    if (!h || !T) return fail(APDGICP_ERR_INVALID_ARG, "null argument");
    APD_TRY(ensure_pair(h));
    APD_TRY(h->eng.probe_linearize(T, H, b, cost, nullptr));
    h->have_corr = true;
    return 0;
  });
}

int apdgicp_compute_error(apdgicp_handle* h, const double T[16], double* cost) {
  return guarded([&]() -> int {
    if (!h || !T || !cost) return fail(APDGICP_ERR_INVALID_ARG, "null argument");
    if (!h->pair_ready || !h->have_corr) return fail(APDGICP_ERR_NO_INPUT, "compute_error needs a previous linearize");
    return h->eng.probe_error(T, cost);
  });
}

int apdgicp_get_correspondences(apdgicp_handle* h, int32_t* corr, float* sq, int64_t n) {
  return guarded([&]() -> int {
    if (!h) return fail(APDGICP_ERR_INVALID_ARG, "handle is null");
    Engine& e = h->eng;
    if (!h->pair_ready || !h->have_corr) return fail(APDGICP_ERR_NO_INPUT, "no correspondences yet");
    if (n != e.clouds[kSrc].n) return fail(APDGICP_ERR_INVALID_ARG, "n does not match the source size");
    APD_HIP(hipStreamSynchronize(e.stream));
    APD_TRY(e.d_stage.ensure((size_t)n * 8));
    int* d_c = e.d_stage.as<int>();
    float* d_s = (float*)(d_c + n);
    hipLaunchKernelGGL(k_export_corr, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, e.stream, e.work.corr, e.work.sqd, e.clouds[kSrc].perm.as<int>(),
                       e.clouds[kTgt].perm.as<int>(), (int)n, d_c, d_s);
    if (corr) APD_HIP(hipMemcpyAsync(corr, d_c, n * sizeof(int), hipMemcpyDeviceToHost, e.stream));
    if (sq) APD_HIP(hipMemcpyAsync(sq, d_s, n * sizeof(float), hipMemcpyDeviceToHost, e.stream));
    APD_HIP(hipStreamSynchronize(e.stream));
    return 0;
  });
}

int apdgicp_get_mahalanobis(apdgicp_handle* h, double* out, int64_t n) {
  return guarded([&]() -> int {
    if (!h || !out) return fail(APDGICP_ERR_INVALID_ARG, "null argument");
    Engine& e = h->eng;
    if (!h->pair_ready || !h->have_corr) return fail(APDGICP_ERR_NO_INPUT, "no correspondences yet");
    if (n != e.clouds[kSrc].n) return fail(APDGICP_ERR_INVALID_ARG, "n does not match the source size");
    const size_t ns = e.work.nstride;
    std::vector<double> m6(6 * ns);
    std::vector<int> corr(n), perm(n);
    APD_HIP(hipMemcpyAsync(m6.data(), e.work.maha, 6 * ns * sizeof(double), hipMemcpyDeviceToHost, e.stream));
    APD_HIP(hipMemcpyAsync(corr.data(), e.work.corr, n * sizeof(int), hipMemcpyDeviceToHost, e.stream));
    APD_HIP(hipMemcpyAsync(perm.data(), e.clouds[kSrc].perm.p, n * sizeof(int), hipMemcpyDeviceToHost, e.stream));
    APD_HIP(hipStreamSynchronize(e.stream));
    for (int64_t i = 0; i < n; i++) {  // i = position on the Z-curve, perm[i] = the caller's index
      double* o = out + 16 * perm[i];
      memset(o, 0, 16 * sizeof(double));
      if (corr[i] < 0) continue;
      const double xx = m6[i], xy = m6[ns + i], xz = m6[2 * ns + i], yy = m6[3 * ns + i], yz = m6[4 * ns + i], zz = m6[5 * ns + i];
      o[0] = xx, o[1] = xy, o[2] = xz, o[4] = xy, o[5] = yy, o[6] = yz, o[8] = xz, o[9] = yz, o[10] = zz;  // (3,3) = 0, A:192
    }
    return 0;
  });
}

int apdgicp_align(apdgicp_handle* h, const float guess[16], apdgicp_result* out) {
  return guarded([&]() -> int {
    if (!h || !out) return fail(APDGICP_ERR_INVALID_ARG, "null argument");
    float g[16];
    if (guess) memcpy(g, guess, sizeof(g));
    else identity16(g);
    Engine& e = h->eng;
    struct Fuse {  // the covariance launch of freshly set clouds goes out with the first tick's search (k_knn_and_search)
      Engine& e;
      explicit Fuse(Engine& e_) : e(e_) { e.fuse_first_search = true; }
      ~Fuse() {
        e.fuse_first_search = false;
        (void)e.flush_pending_knn();  // (only after an error on the way: nothing stays pending)
      }
    } fuse(e);
    APD_TRY(ensure_pair(h, g));
    APD_TRY(e.upload_guesses(g, 1));  // no-op when ensure_pair has just uploaded it
    APD_TRY(e.run_align());
    h->trace_from_host_loop = false;
    if (const ResultRec* r = e.host_results()) {  // came home with the last poll
      memcpy(out, r, sizeof(apdgicp_result));
    } else {
      APD_HIP(hipMemcpyAsync(out, e.d_results.p, sizeof(apdgicp_result), hipMemcpyDeviceToHost, e.stream));
      APD_HIP(hipStreamSynchronize(e.stream));
    }
    h->have_corr = out->n_linearize > 0;
    return 0;
  });
}

// The reference's own control flow (L:55-173) on the host, calling the device through the same two
// virtuals the reference uses (linearize, compute_error).  Used to separate "kernel parity" from
// "state-machine parity" in the tests.
int apdgicp_align_host_loop(apdgicp_handle* h, const float guess[16], apdgicp_result* out) {
  return guarded([&]() -> int {
    if (!h || !out) return fail(APDGICP_ERR_INVALID_ARG, "null argument");
    APD_TRY(ensure_pair(h));
    Engine& e = h->eng;
    const apdgicp_params& p = e.params;
    float g[16];
    if (guess) memcpy(g, guess, sizeof(g));
    else identity16(g);
    Rigid x0;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 4; j++) x0.m[4 * i + j] = (double)g[i + 4 * j];  // L:56
    double lambda = -1.0;                                                   // L:58
    bool converged = false;
    int nr_iterations = 0, n_lin = 0, n_err = 0, failed = 0, matched = 0;
    double final_H[36];
    for (int q = 0; q < 36; q++) final_H[q] = (q % 7 == 0) ? 1.0 : 0.0;
    double y0 = 0.0;
    h->tr_lambda.clear(), h->tr_rho.clear(), h->tr_y0.clear(), h->tr_yi.clear(), h->tr_dnorm.clear(), h->tr_poses.clear();
    h->trace_from_host_loop = true;
    for (int it = 0; it < p.max_iterations && !converged; it++) {  // L:67
      nr_iterations = it;
      double T16[16], H[36], b[6], d[6];
      rigid_to_colmajor(x0, T16);
      APD_TRY(e.probe_linearize(T16, H, b, &y0, &matched));
      n_lin++;
      Rigid delta = rigid_identity();
      bool ok = false;
      if (p.optimizer == APDGICP_OPT_GN) {  // L:107-123
        solve6_spd(H, 0.0, b, d);
        delta = make_delta(d);
        x0 = rigid_mul(delta, x0);
        memcpy(final_H, H, sizeof(H));
        ok = true;
      } else {  // L:127-173
        if (lambda < 0.0) {
          double mx = 0.0;
          for (int q = 0; q < 6; q++) mx = std::max(mx, std::fabs(H[q + 6 * q]));
          lambda = p.lm_init_lambda_factor * mx;
        }
        double nu = 2.0;
        for (int in = 0; in < p.lm_max_iterations; in++) {
          solve6_spd(H, lambda, b, d);
          delta = make_delta(d);
          const Rigid xi = rigid_mul(delta, x0);
          double yi = 0.0;
          rigid_to_colmajor(xi, T16);
          APD_TRY(e.probe_error(T16, &yi));
          n_err++;
          double den = 0.0;
          for (int q = 0; q < 6; q++) den += d[q] * (lambda * d[q] - b[q]);
          const double rho = (y0 - yi) / den;
          if (e.trace_on) {
            double nn = 0.0;
            for (int q = 0; q < 6; q++) nn += d[q] * d[q];
            h->tr_lambda.push_back(lambda), h->tr_rho.push_back(rho), h->tr_y0.push_back(y0), h->tr_yi.push_back(yi), h->tr_dnorm.push_back(std::sqrt(nn));
          }
          if (rho < 0) {
            if (is_converged(delta, p.rotation_epsilon, p.transformation_epsilon)) {
              ok = true;
              break;
            }
            lambda = nu * lambda;
            nu = 2 * nu;
            continue;
          }
          x0 = xi;
          const double t = 2 * rho - 1;
          lambda = lambda * std::max(1.0 / 3.0, 1 - t * t * t);
          memcpy(final_H, H, sizeof(H));
          ok = true;
          break;
        }
      }
      if (!ok) {
        failed = 1;
        break;
      }
      if (e.trace_on) {
        double P[16];
        rigid_to_colmajor(x0, P);
        h->tr_poses.insert(h->tr_poses.end(), P, P + 16);
      }
      converged = is_converged(delta, p.rotation_epsilon, p.transformation_epsilon);
    }
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 4; j++) out->T[i + 4 * j] = (float)x0.m[4 * i + j];
    out->T[3] = out->T[7] = out->T[11] = 0.f;
    out->T[15] = 1.f;
    out->final_cost = y0;
    out->converged = converged;
    out->iterations = nr_iterations;
    out->n_linearize = n_lin;
    out->n_compute_error = n_err;
    out->lm_failed = failed;
    out->n_matched = matched;
    h->have_corr = n_lin > 0;
    // keep getFinalHessian() coherent with this path
    APD_HIP(hipMemcpyAsync((char*)e.d_state.p + offsetof(PairState, final_H), final_H, sizeof(final_H), hipMemcpyHostToDevice, e.stream));
    APD_HIP(hipStreamSynchronize(e.stream));
    return 0;
  });
}

int apdgicp_set_trace(apdgicp_handle* h, int enable) {
  if (!h) return fail(APDGICP_ERR_INVALID_ARG, "handle is null");
  h->eng.trace_on = enable != 0;
  return 0;
}

int apdgicp_get_trace(apdgicp_handle* h, int64_t trial_capacity, double* lambdas, double* rhos, double* y0s, double* yis, int64_t* n_trials,
                      int64_t pose_capacity, double* poses16, int64_t* n_poses) {
  return guarded([&]() -> int {
    if (!h || !n_trials || !n_poses) return fail(APDGICP_ERR_INVALID_ARG, "null argument");
    if (trial_capacity < 0 || pose_capacity < 0 || (trial_capacity > 0 && (!lambdas || !rhos)) || (pose_capacity > 0 && !poses16))
      return fail(APDGICP_ERR_INVALID_ARG, "bad capacity / buffer");
    Engine& e = h->eng;
    if (!e.trace_on) return fail(APDGICP_ERR_NO_INPUT, "tracing is off (apdgicp_set_trace)");
    if (h->trace_from_host_loop) {
      *n_trials = (int64_t)h->tr_lambda.size(), *n_poses = (int64_t)h->tr_poses.size() / 16;
      const int64_t nt = std::min<int64_t>(*n_trials, trial_capacity), np = std::min<int64_t>(*n_poses, pose_capacity);
      if (nt) memcpy(lambdas, h->tr_lambda.data(), nt * sizeof(double)), memcpy(rhos, h->tr_rho.data(), nt * sizeof(double));
      if (nt && y0s) memcpy(y0s, h->tr_y0.data(), nt * sizeof(double));
      if (nt && yis) memcpy(yis, h->tr_yi.data(), nt * sizeof(double));
      if (np) memcpy(poses16, h->tr_poses.data(), np * 16 * sizeof(double));
      return 0;
    }
    if (!e.d_trace.p) return fail(APDGICP_ERR_NO_INPUT, "no align has run since tracing was enabled");
    std::vector<double> buf(e.trace_bytes() / sizeof(double));
    APD_HIP(hipMemcpyAsync(buf.data(), e.d_trace.p, e.trace_bytes(), hipMemcpyDeviceToHost, e.stream));
    APD_HIP(hipStreamSynchronize(e.stream));
    int hdr[4];
    memcpy(hdr, buf.data(), sizeof(hdr));
    *n_trials = hdr[0], *n_poses = hdr[1];
    const int64_t nt = std::min<int64_t>(std::min(hdr[0], hdr[2]), trial_capacity), np = std::min<int64_t>(std::min(hdr[1], hdr[3]), pose_capacity);
    if (nt) memcpy(lambdas, &buf[2], nt * sizeof(double)), memcpy(rhos, &buf[2 + hdr[2]], nt * sizeof(double));
    if (nt && y0s) memcpy(y0s, &buf[2 + 2 * (size_t)hdr[2]], nt * sizeof(double));
    if (nt && yis) memcpy(yis, &buf[2 + 3 * (size_t)hdr[2]], nt * sizeof(double));
    for (int64_t q = 0; q < np; q++) {
      Rigid r;
      memcpy(r.m, &buf[2 + 4 * (size_t)hdr[2] + 12 * (size_t)q], sizeof(r.m));
      rigid_to_colmajor(r, poses16 + 16 * q);
    }
    return 0;
  });
}

int apdgicp_get_trace_step_norms(apdgicp_handle* h, int64_t capacity, double* norms, int64_t* n_trials) {
  return guarded([&]() -> int {
    if (!h || !n_trials || capacity < 0 || (capacity > 0 && !norms)) return fail(APDGICP_ERR_INVALID_ARG, "bad argument");
    Engine& e = h->eng;
    if (!e.trace_on) return fail(APDGICP_ERR_NO_INPUT, "tracing is off (apdgicp_set_trace)");
    if (h->trace_from_host_loop) {
      *n_trials = (int64_t)h->tr_dnorm.size();
      const int64_t nt = std::min<int64_t>(*n_trials, capacity);
      if (nt) memcpy(norms, h->tr_dnorm.data(), nt * sizeof(double));
      return 0;
    }
    if (!e.d_trace.p) return fail(APDGICP_ERR_NO_INPUT, "no align has run since tracing was enabled");
    std::vector<double> buf(e.trace_bytes() / sizeof(double));
    APD_HIP(hipMemcpyAsync(buf.data(), e.d_trace.p, e.trace_bytes(), hipMemcpyDeviceToHost, e.stream));
    APD_HIP(hipStreamSynchronize(e.stream));
    int hdr[4];
    memcpy(hdr, buf.data(), sizeof(hdr));
    *n_trials = hdr[0];
    const int64_t nt = std::min<int64_t>(std::min(hdr[0], hdr[2]), capacity);
    if (nt) memcpy(norms, &buf[2 + 4 * (size_t)hdr[2] + 12 * (size_t)hdr[3]], nt * sizeof(double));
    return 0;
  });
}

int apdgicp_debug_atan2f(int device, const float* y, const float* x, float* out, int64_t n) {
  return guarded([&]() -> int {
    if (!y || !x || !out || n < 0) return fail(APDGICP_ERR_INVALID_ARG, "bad argument");
    if (n == 0) return 0;
    APD_HIP(hipSetDevice(device));
    DevBuf buf;
    APD_TRY(buf.ensure((size_t)n * 12));
    float* dy = buf.as<float>();
    int rc = 0;
    hipError_t he = hipMemcpy(dy, y, (size_t)n * 4, hipMemcpyHostToDevice);
    if (he == hipSuccess) he = hipMemcpy(dy + n, x, (size_t)n * 4, hipMemcpyHostToDevice);
    if (he == hipSuccess) {
      hipLaunchKernelGGL(k_debug_atan2f, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, dy, dy + n, dy + 2 * n, (long long)n);
      he = hipGetLastError();
    }
    if (he == hipSuccess) he = hipMemcpy(out, dy + 2 * n, (size_t)n * 4, hipMemcpyDeviceToHost);
    if (he != hipSuccess) rc = fail(APDGICP_ERR_HIP, hipGetErrorString(he));
    buf.release();
    return rc;
  });
}

int apdgicp_get_final_hessian(apdgicp_handle* h, double H[36]) {
  return guarded([&]() -> int {
    if (!h || !H) return fail(APDGICP_ERR_INVALID_ARG, "null argument");
    Engine& e = h->eng;
    if (!h->pair_ready) {  // L:23: identity until the first accepted step
      for (int q = 0; q < 36; q++) H[q] = (q % 7 == 0) ? 1.0 : 0.0;
      return 0;
    }
    APD_HIP(hipMemcpyAsync(H, (char*)e.d_state.p + offsetof(PairState, final_H), 36 * sizeof(double), hipMemcpyDeviceToHost, e.stream));
    APD_HIP(hipStreamSynchronize(e.stream));
    return 0;
  });
}

int apdgicp_transform_source(apdgicp_handle* h, const float T[16], float* out_xyz, int64_t n, int64_t out_stride_bytes) {
  return guarded([&]() -> int {
    if (!h || !T || !out_xyz) return fail(APDGICP_ERR_INVALID_ARG, "null argument");
    Engine& e = h->eng;
    Engine::Cloud& c = e.clouds[kSrc];
    if (c.n <= 0) return fail(APDGICP_ERR_NO_INPUT, "source cloud is not set");
    if (n != c.n) return fail(APDGICP_ERR_INVALID_ARG, "n does not match the source size");
    if (out_stride_bytes < 12 || (out_stride_bytes & 3)) return fail(APDGICP_ERR_INVALID_ARG, "bad output stride");
    APD_HIP(hipStreamSynchronize(e.stream));
    APD_TRY(e.d_stage.ensure((size_t)n * 12 + 64));
    float* dT = (float*)((char*)e.d_stage.p + (size_t)n * 12);
    APD_HIP(hipMemcpyAsync(dT, T, 16 * sizeof(float), hipMemcpyHostToDevice, e.stream));
    hipLaunchKernelGGL(k_transform_points, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, e.stream, c.opts.as<float4>(), (int)n, dT, e.d_stage.as<float>(),
                       3ll);
    if (out_stride_bytes == 12) {
      APD_HIP(hipMemcpyAsync(out_xyz, e.d_stage.p, (size_t)n * 12, hipMemcpyDeviceToHost, e.stream));
      APD_HIP(hipStreamSynchronize(e.stream));
    } else {
      std::vector<float> tmp((size_t)n * 3);
      APD_HIP(hipMemcpyAsync(tmp.data(), e.d_stage.p, (size_t)n * 12, hipMemcpyDeviceToHost, e.stream));
      APD_HIP(hipStreamSynchronize(e.stream));
      for (int64_t i = 0; i < n; i++) memcpy((char*)out_xyz + i * out_stride_bytes, &tmp[3 * i], 12);
    }
    return 0;
  });
}

// nearest-neighbour statistics of the T-transformed source: sum and count of the squared distances inside the range
static int nn_range_stats(apdgicp_handle* h, const float T[16], double max_range2, int strict, double* sum, double* cnt) {
  APD_TRY(ensure_pair(h));
  Engine& e = h->eng;
  roctx_range rr("apdgicp:fitness");
  double T16[16];
  for (int q = 0; q < 16; q++) T16[q] = (double)T[q];
  APD_HIP(hipMemcpyAsync(e.d_T.p, T16, sizeof(T16), hipMemcpyHostToDevice, e.stream));
  hipLaunchKernelGGL(k_set_probe, dim3(1), dim3(1), 0, e.stream, e.d_state.as<PairState>(), e.d_T.as<double>(), (int)ST_NEED_LIN, 0);
  APD_TRY(e.launch_nn(e.whole()));
  APD_HIP(hipMemsetAsync(e.d_probe.p, 0, 2 * sizeof(double), e.stream));
  hipLaunchKernelGGL(k_fitness, dim3((unsigned)e.work.nblk_max, 1), dim3(LIN_BLK), 0, e.stream, e.d_desc.as<CloudDesc>(), e.d_pairs.as<PairDesc>(), e.work,
                     max_range2, e.d_probe.as<double>(), strict);
  APD_HIP(hipMemcpyAsync(e.h_probe, e.d_probe.p, 2 * sizeof(double), hipMemcpyDeviceToHost, e.stream));
  APD_HIP(hipStreamSynchronize(e.stream));
  h->have_corr = false;  // the nn partials were overwritten at another pose
  *sum = e.h_probe[0], *cnt = e.h_probe[1];
  return 0;
}

int apdgicp_fitness_score(apdgicp_handle* h, const float T[16], double max_range, double* score, int64_t* n_inliers) {
  return guarded([&]() -> int {
    if (!h || !T || !score) return fail(APDGICP_ERR_INVALID_ARG, "null argument");
    double sum = 0, cnt = 0;
    APD_TRY(nn_range_stats(h, T, max_range, 0, &sum, &cnt));
    *score = cnt > 0 ? sum / cnt : std::numeric_limits<double>::max();  // pcl returns max() when nothing is in range
    if (n_inliers) *n_inliers = (int64_t)cnt;
    return 0;
  });
}

int apdgicp_inlier_fraction(apdgicp_handle* h, const float T[16], double max_correspondence_dist, double* fraction, int64_t* n_inliers) {
  return guarded([&]() -> int {
    if (!h || !T || !fraction) return fail(APDGICP_ERR_INVALID_ARG, "null argument");
    double sum = 0, cnt = 0;
    APD_TRY(nn_range_stats(h, T, max_correspondence_dist * max_correspondence_dist, 1, &sum, &cnt));
    *fraction = (double)((float)(int)cnt / (float)h->eng.clouds[kSrc].n);  // static_cast<float>(num_inliers) / aligned->size()
    if (n_inliers) *n_inliers = (int64_t)cnt;
    return 0;
  });
}

int apdgicp_nearest_neighbours(apdgicp_handle* h, const float T[16], int32_t* index, float* sq_dist, int64_t n) {
  return guarded([&]() -> int {
    if (!h || !T || !index || !sq_dist) return fail(APDGICP_ERR_INVALID_ARG, "null argument");
    APD_TRY(ensure_pair(h));
    Engine& e = h->eng;
    if (n != e.clouds[kSrc].n) return fail(APDGICP_ERR_INVALID_ARG, "n does not match the source size");
    double T16[16], cost = 0.0;
    for (int q = 0; q < 16; q++) T16[q] = (double)T[q];
    APD_TRY(e.probe_linearize(T16, nullptr, nullptr, &cost, nullptr));  // search (ungated, cold) + the per-point pass that settles the exact index
    h->have_corr = true;
    APD_TRY(e.d_stage.ensure((size_t)n * 8));
    int* d_i = e.d_stage.as<int>();
    float* d_s = (float*)(d_i + n);
    hipLaunchKernelGGL(k_export_nn, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, e.stream, e.work.nnpt, e.work.sqd, e.clouds[kSrc].perm.as<int>(),
                       e.clouds[kTgt].perm.as<int>(), (int)n, d_i, d_s);
    APD_HIP(hipMemcpyAsync(index, d_i, n * sizeof(int), hipMemcpyDeviceToHost, e.stream));
    APD_HIP(hipMemcpyAsync(sq_dist, d_s, n * sizeof(float), hipMemcpyDeviceToHost, e.stream));
    APD_HIP(hipStreamSynchronize(e.stream));
    return 0;
  });
}

int apdgicp_nearest_neighbours_of(apdgicp_handle* h, const float* queries_xyz, int64_t n, int64_t stride_bytes, int32_t* index, float* sq_dist) {
  return guarded([&]() -> int {
    if (!h || !queries_xyz || !index || !sq_dist || n <= 0) return fail(APDGICP_ERR_INVALID_ARG, "null argument or no queries");
    Engine& e = h->eng;
    constexpr int kQry = 2;  // a scratch cloud slot behind source and target
    if (e.clouds.size() < 2 || e.clouds[kTgt].n <= 0) return fail(APDGICP_ERR_NO_INPUT, "target cloud is not set");
    APD_TRY(e.set_cloud(kQry, queries_xyz, n, stride_bytes, 0, 0));
    apdgicp_pair p;
    p.source_cloud = kQry, p.target_cloud = kTgt;
    identity16(p.guess);
    h->pair_ready = false, h->have_corr = false;  // the handle's own pair is set up again by whoever needs it next
    APD_TRY(e.setup_pairs(&p, 1, true, false, /*need_cov=*/false));
    // the ungated cold search + the per-point pass that settles the exact index (the queries' covariances are whatever the buffer holds:
    // the pass computes a cost nobody reads)
    const double I16[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    double cost = 0.0;
    APD_TRY(e.probe_linearize(I16, nullptr, nullptr, &cost, nullptr));
    APD_TRY(e.d_stage.ensure((size_t)n * 8));
    int* d_i = e.d_stage.as<int>();
    float* d_s = (float*)(d_i + n);
    hipLaunchKernelGGL(k_export_nn, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, e.stream, e.work.nnpt, e.work.sqd, e.clouds[kQry].perm.as<int>(),
                       e.clouds[kTgt].perm.as<int>(), (int)n, d_i, d_s);
    APD_HIP(hipMemcpyAsync(index, d_i, n * sizeof(int), hipMemcpyDeviceToHost, e.stream));
    APD_HIP(hipMemcpyAsync(sq_dist, d_s, n * sizeof(float), hipMemcpyDeviceToHost, e.stream));
    APD_HIP(hipStreamSynchronize(e.stream));
    return 0;
  });
}

int apdgicp_get_points(apdgicp_handle* h, int which, float* out_xyz, int64_t n) {
  return guarded([&]() -> int {
    if (!h || !out_xyz || (which != kSrc && which != kTgt)) return fail(APDGICP_ERR_INVALID_ARG, "bad argument");
    Engine& e = h->eng;
    Engine::Cloud& c = e.clouds[which];
    if (c.n <= 0) return fail(APDGICP_ERR_NO_INPUT, "cloud not set");
    if (n != c.n) return fail(APDGICP_ERR_INVALID_ARG, "n does not match the cloud size");
    APD_HIP(hipSetDevice(e.device));
    if (c.staged) {  // a scan-sized host cloud still waiting in its pinned buffer for the sort: the float4 copy is right there
      const float4* p = (const float4*)c.stage_p;
      for (int64_t i = 0; i < n; i++) out_xyz[3 * i] = p[i].x, out_xyz[3 * i + 1] = p[i].y, out_xyz[3 * i + 2] = p[i].z;
      return 0;
    }
    APD_TRY(e.upload_desc());  // (sorts what is not sorted yet: a staged cloud's opts are written by its sort)
    APD_HIP(hipStreamSynchronize(e.stream));
    APD_TRY(e.d_stage.ensure((size_t)n * 12));
    hipLaunchKernelGGL(k_unpack_points, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, e.stream, c.opts.as<float4>(), (int)n, e.d_stage.as<float>());
    APD_HIP(hipMemcpyAsync(out_xyz, e.d_stage.p, (size_t)n * 12, hipMemcpyDeviceToHost, e.stream));
    APD_HIP(hipStreamSynchronize(e.stream));
    return 0;
  });
}

int apdgicp_wait_producer(apdgicp_handle* h, void* producer_stream) {
  if (!h) return fail(APDGICP_ERR_INVALID_ARG, "handle is null");
  return h->eng.wait_producer(producer_stream);
}

int apdgicp_get_stream(apdgicp_handle* h, void** stream) {
  if (!h || !stream) return fail(APDGICP_ERR_INVALID_ARG, "null argument");
  *stream = (void*)h->eng.stream;
  return 0;
}

int apdgicp_batch_get_stream(apdgicp_batch* b, void** stream) {
  if (!b || !stream) return fail(APDGICP_ERR_INVALID_ARG, "null argument");
  *stream = (void*)b->eng.stream;
  return 0;
}

int apdgicp_batch_wait_producer(apdgicp_batch* b, void* producer_stream) {
  if (!b) return fail(APDGICP_ERR_INVALID_ARG, "batch is null");
  return b->eng.wait_producer(producer_stream);
}

int apdgicp_synchronize(apdgicp_handle* h) {
  if (!h) return fail(APDGICP_ERR_INVALID_ARG, "handle is null");
  APD_HIP(hipStreamSynchronize(h->eng.stream));
  return 0;
}

// ------------------------------------------------------------------------------------ batch
int apdgicp_batch_create(const apdgicp_params* p, int device, void* stream, apdgicp_batch** out) {
  return guarded([&]() -> int {
    if (!out) return fail(APDGICP_ERR_INVALID_ARG, "out is null");
    *out = nullptr;
    apdgicp_params dflt;
    apdgicp_default_params(&dflt);
    apdgicp_batch* b = new apdgicp_batch;
    const int rc = b->eng.init(p ? p : &dflt, device, stream);
    if (rc < 0) {
      delete b;
      return rc;
    }
    b->eng.profile_nn = env_int("APDGICP_PROFILE_NN", 0) != 0;
    b->eng.keep_maha = false;  // no batch entry point reads the Mahalanobis matrices back
    *out = b;
    return 0;
  });
}

int apdgicp_batch_destroy(apdgicp_batch* b) {
  delete b;
  return 0;
}

int apdgicp_batch_set_params(apdgicp_batch* b, const apdgicp_params* p) {
  if (!b) return fail(APDGICP_ERR_INVALID_ARG, "batch is null");
  return b->eng.set_params(p);
}

int apdgicp_batch_clear(apdgicp_batch* b) {
  if (!b) return fail(APDGICP_ERR_INVALID_ARG, "batch is null");
  APD_TRY(b->eng.pool_leave());
  APD_HIP(hipStreamSynchronize(b->eng.stream));
  for (auto& c : b->eng.clouds) c.release_all();
  b->eng.clouds.clear();
  b->eng.desc_dirty = true;
  return 0;
}

int apdgicp_batch_add_cloud(apdgicp_batch* b, const float* xyz, int64_t n, int64_t stride_bytes, int on_device) {
  return guarded([&]() -> int {
    if (!b) return fail(APDGICP_ERR_INVALID_ARG, "batch is null");
    const int slot = (int)b->eng.clouds.size();
    const int rc = b->eng.set_cloud(slot, xyz, n, stride_bytes, on_device, 0);
    if (rc < 0) {
      if ((int)b->eng.clouds.size() > slot) b->eng.clouds.resize(slot);
      return rc;
    }
    return slot;
  });
}

int apdgicp_batch_set_cloud(apdgicp_batch* b, int32_t index, const float* xyz, int64_t n, int64_t stride_bytes, int on_device) {
  return guarded([&]() -> int {
    if (!b) return fail(APDGICP_ERR_INVALID_ARG, "batch is null");
    if (index < 0 || index >= (1 << 24)) return fail(APDGICP_ERR_INVALID_ARG, "cloud index out of range");
    APD_TRY(b->eng.set_cloud(index, xyz, n, stride_bytes, on_device, 0));
    return index;
  });
}

int apdgicp_batch_set_clouds(apdgicp_batch* b, int32_t first_index, int32_t count, const float* const* xyz, const int64_t* n, int64_t stride_bytes,
                             int on_device) {
  return guarded([&]() -> int {
    if (!b) return fail(APDGICP_ERR_INVALID_ARG, "batch is null");
    if (on_device) return b->eng.set_clouds_device(first_index, count, xyz, n, stride_bytes);
    return b->eng.set_clouds_host(first_index, count, xyz, n, stride_bytes);
  });
}

int apdgicp_batch_compute_covariances(apdgicp_batch* b) {
  return guarded([&]() -> int {
    if (!b) return fail(APDGICP_ERR_INVALID_ARG, "batch is null");
    std::vector<int> ids;
    for (int i = 0; i < (int)b->eng.clouds.size(); i++)
      if (b->eng.clouds[i].n > 0) ids.push_back(i);
    return b->eng.compute_covariances(ids);
  });
}

int apdgicp_batch_align_async(apdgicp_batch* b, const apdgicp_pair* pairs, int64_t n_pairs, void** d_results) {
  return guarded([&]() -> int {
    if (!b || !pairs) return fail(APDGICP_ERR_INVALID_ARG, "null argument");
    if (b->eng.pool_eligible()) {  // Levenberg-Marquardt: through the pair pool (complete on return, like every LM run)
      uint64_t ticket = 0;
      APD_TRY(b->eng.pool_enqueue(pairs, n_pairs, &ticket));
      return b->eng.pool_collect(ticket, d_results, nullptr);
    }
    APD_TRY(b->eng.setup_pairs(pairs, n_pairs, true, /*pipeline_cov=*/true));
    APD_TRY(b->eng.run_align());
    if (d_results) *d_results = b->eng.d_results.p;
    return 0;
  });
}

int apdgicp_batch_align_enqueue(apdgicp_batch* b, const apdgicp_pair* pairs, int64_t n_pairs, uint64_t* ticket) {
  return guarded([&]() -> int {
    if (!b || !pairs || !ticket) return fail(APDGICP_ERR_INVALID_ARG, "null argument");
    Engine& e = b->eng;
    if (e.pool_eligible()) return e.pool_enqueue(pairs, n_pairs, ticket);
    APD_TRY(e.ensure_alt_slot());
    e.swap_slots();  // the slot of the batch before the last one becomes current (run_align waits for it if nobody collected it)
    e.align_seq++;
    b->slot_pairs[e.align_seq & 1] = n_pairs;
    APD_TRY(e.setup_pairs(pairs, n_pairs, true, /*pipeline_cov=*/true));
    APD_TRY(e.run_align(/*defer_poll=*/true));
    *ticket = e.align_seq;
    return 0;
  });
}

int apdgicp_batch_pump(apdgicp_batch* b) {
  return guarded([&]() -> int {
    if (!b) return fail(APDGICP_ERR_INVALID_ARG, "batch is null");
    return b->eng.pool.on ? b->eng.pool_pump(false) : 0;
  });
}

int apdgicp_batch_is_pooled(apdgicp_batch* b) {
  if (!b) return fail(APDGICP_ERR_INVALID_ARG, "batch is null");
  return b->eng.pool_eligible() ? Engine::pool_lanes_cfg() : 0;
}

int apdgicp_batch_align_collect(apdgicp_batch* b, uint64_t ticket, void** d_results, apdgicp_result* host_results) {
  return guarded([&]() -> int {
    if (!b) return fail(APDGICP_ERR_INVALID_ARG, "batch is null");
    Engine& e = b->eng;
    if (e.pool_find(ticket)) return e.pool_collect(ticket, d_results, host_results);
    const bool previous = ticket + 1 == e.align_seq;
    if (ticket == 0 || (ticket != e.align_seq && !previous)) return fail(APDGICP_ERR_INVALID_ARG, "ticket is not one of the last two enqueued batches");
    if (previous) e.swap_slots();
    int rc = e.finish_align();
    const int64_t n = b->slot_pairs[ticket & 1];
    if (rc == 0) {
      if (d_results) *d_results = e.d_results.p;
      if (host_results) {
        if (const ResultRec* r = e.host_results()) {
          memcpy(host_results, r, n * sizeof(apdgicp_result));
        } else {
          const hipError_t he = hipMemcpy(host_results, e.d_results.p, n * sizeof(apdgicp_result), hipMemcpyDeviceToHost);
          if (he != hipSuccess) rc = fail(APDGICP_ERR_HIP, hipGetErrorString(he));
        }
      }
    }
    if (previous) e.swap_slots();
    return rc;
  });
}

int apdgicp_batch_align(apdgicp_batch* b, const apdgicp_pair* pairs, int64_t n_pairs, apdgicp_result* results) {
  return guarded([&]() -> int {
    if (!results) return fail(APDGICP_ERR_INVALID_ARG, "results is null");
    if (b && pairs && b->eng.pool_eligible()) {
      uint64_t ticket = 0;
      APD_TRY(b->eng.pool_enqueue(pairs, n_pairs, &ticket));
      return b->eng.pool_collect(ticket, nullptr, results);
    }
    APD_TRY(apdgicp_batch_align_async(b, pairs, n_pairs, nullptr));
    Engine& e = b->eng;
    if (const ResultRec* r = e.host_results()) {
      memcpy(results, r, n_pairs * sizeof(apdgicp_result));
      return 0;
    }
    APD_HIP(hipMemcpyAsync(results, e.d_results.p, n_pairs * sizeof(apdgicp_result), hipMemcpyDeviceToHost, e.stream));
    APD_HIP(hipStreamSynchronize(e.stream));
    return 0;
  });
}

int apdgicp_batch_fitness(apdgicp_batch* b, const apdgicp_pair* pairs, int64_t n_pairs, const float* T, double max_range, double* scores,
                          int64_t* inliers) {
  return guarded([&]() -> int {
    if (!b || !pairs || !scores) return fail(APDGICP_ERR_INVALID_ARG, "null argument");
    Engine& e = b->eng;
    const bool same = e.npairs == n_pairs && (int64_t)e.h_pairs.size() == n_pairs && !e.desc_dirty;  // no cloud replaced since
    bool same_pairs = same;
    for (int64_t i = 0; same_pairs && i < n_pairs; i++)
      same_pairs = e.h_pairs[i].src == pairs[i].source_cloud && e.h_pairs[i].tgt == pairs[i].target_cloud;
    std::vector<float> pool_T;
    if (!T && e.pool.on && e.pool.last_lane >= 0) {  // the last align ran in the pair pool: its poses are in the batch's records
      const Engine::PoolJob& j = e.pool.jobs[e.pool.last_lane];
      bool match = j.state != Engine::PoolJob::FREE && (int64_t)j.pair_ids.size() == n_pairs;
      for (int64_t i = 0; match && i < n_pairs; i++) match = j.pair_ids[i].first == pairs[i].source_cloud && j.pair_ids[i].second == pairs[i].target_cloud;
      if (match) {
        APD_TRY(e.pool_collect(j.ticket, nullptr, nullptr));
        pool_T.resize((size_t)n_pairs * 16);
        for (int64_t i = 0; i < n_pairs; i++) memcpy(&pool_T[(size_t)i * 16], j.recs[i].T, 16 * sizeof(float));
        T = pool_T.data();
        same_pairs = false;
      }
    }
    if (!T && !same_pairs) return fail(APDGICP_ERR_NO_INPUT, "T == NULL needs a previous align of the same pair list");
    if (!same_pairs) APD_TRY(e.setup_pairs(pairs, n_pairs, true));
    APD_HIP(hipStreamSynchronize(e.stream));
    APD_TRY(e.d_stage.ensure((size_t)n_pairs * (16 * sizeof(float) + 2 * sizeof(double))));
    double* d_out = e.d_stage.as<double>();
    float* d_T = (float*)(d_out + 2 * n_pairs);
    if (T) APD_HIP(hipMemcpyAsync(d_T, T, (size_t)n_pairs * 16 * sizeof(float), hipMemcpyHostToDevice, e.stream));
    APD_HIP(hipMemsetAsync(d_out, 0, (size_t)n_pairs * 2 * sizeof(double), e.stream));
    hipLaunchKernelGGL(k_set_poses, dim3((unsigned)((n_pairs + 63) / 64)), dim3(64), 0, e.stream, e.d_state.as<PairState>(), T ? d_T : (const float*)nullptr,
                       (int)n_pairs);
    APD_TRY(e.launch_nn(e.whole()));
    hipLaunchKernelGGL(k_fitness, dim3((unsigned)e.work.nblk_max, (unsigned)n_pairs), dim3(LIN_BLK), 0, e.stream, e.d_desc.as<CloudDesc>(),
                       e.d_pairs.as<PairDesc>(), e.work, max_range, d_out, 0);
    std::vector<double> h((size_t)n_pairs * 2);
    APD_HIP(hipMemcpyAsync(h.data(), d_out, h.size() * sizeof(double), hipMemcpyDeviceToHost, e.stream));
    APD_HIP(hipStreamSynchronize(e.stream));
    for (int64_t i = 0; i < n_pairs; i++) {
      scores[i] = h[2 * i + 1] > 0 ? h[2 * i] / h[2 * i + 1] : std::numeric_limits<double>::max();
      if (inliers) inliers[i] = (int64_t)h[2 * i + 1];
    }
    return 0;
  });
}

int apdgicp_batch_synchronize(apdgicp_batch* b) {
  if (!b) return fail(APDGICP_ERR_INVALID_ARG, "batch is null");
  return guarded([&]() -> int {
    if (b->eng.pool.on) APD_TRY(b->eng.pool_drain());  // (pooled LM batches: every batch in flight runs to its end)
    if (b->eng.cstream != b->eng.stream) APD_HIP(hipStreamSynchronize(b->eng.cstream));
    if (b->eng.pool.on && b->eng.pool.cstream2) APD_HIP(hipStreamSynchronize(b->eng.pool.cstream2));
    APD_HIP(hipStreamSynchronize(b->eng.stream));
    return 0;
  });
}

int apdgicp_batch_copy_results(apdgicp_batch* b, void* dst, int64_t n_pairs, int dst_on_device) {
  if (!b || !dst) return fail(APDGICP_ERR_INVALID_ARG, "null argument");
  Engine& e = b->eng;
  if (e.pool.on && e.pool.last_lane >= 0) {  // the last align ran in the pair pool
    Engine::PoolJob& j = e.pool.jobs[e.pool.last_lane];
    if (n_pairs <= 0 || n_pairs > j.np) return fail(APDGICP_ERR_INVALID_ARG, "n_pairs exceeds the last batch");
    void* d = nullptr;
    APD_TRY(e.pool_collect(j.ticket, &d, nullptr));
    APD_HIP(hipMemcpyAsync(dst, d, n_pairs * sizeof(apdgicp_result), dst_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, e.stream));
    APD_HIP(hipStreamSynchronize(e.stream));
    return 0;
  }
  if (n_pairs <= 0 || n_pairs > e.npairs) return fail(APDGICP_ERR_INVALID_ARG, "n_pairs exceeds the last batch");
  APD_HIP(hipSetDevice(e.device));
  APD_HIP(hipMemcpyAsync(dst, e.d_results.p, n_pairs * sizeof(apdgicp_result), dst_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, e.stream));
  APD_HIP(hipStreamSynchronize(e.stream));
  return 0;
}

int apdgicp_batch_set_pair_groups(apdgicp_batch* b, int max_groups) {
  if (!b || max_groups < 1) return fail(APDGICP_ERR_INVALID_ARG, "batch is null or max_groups < 1");
  b->eng.max_groups = max_groups;
  return 0;
}

int apdgicp_batch_set_profiling(apdgicp_batch* b, int enable) {
  if (!b) return fail(APDGICP_ERR_INVALID_ARG, "batch is null");
  b->eng.profile_nn = enable != 0;
  return 0;
}

int apdgicp_batch_last_nn_time(apdgicp_batch* b, double* total_ms, int64_t* launches) {
  if (!b) return fail(APDGICP_ERR_INVALID_ARG, "batch is null");
  if (total_ms) *total_ms = b->eng.last_nn_ms;
  if (launches) *launches = b->eng.last_nn_launches;
  return 0;
}

int apdgicp_batch_debug_stats(apdgicp_batch* b, unsigned long long out[16]) {
  if (!b || !out) return fail(APDGICP_ERR_INVALID_ARG, "null argument");
  Engine& e = b->eng;
  memset(out, 0, 16 * sizeof(unsigned long long));
  if (!e.d_stats.p) return 0;
  APD_HIP(hipMemcpyAsync(out, e.d_stats.p, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost, e.stream));
  APD_HIP(hipMemsetAsync(e.d_stats.p, 0, 16 * sizeof(unsigned long long), e.stream));
  APD_HIP(hipStreamSynchronize(e.stream));
  return 0;
}

int apdgicp_batch_debug_block_timeline(apdgicp_batch* b, unsigned long long* out, int64_t capacity_blocks, int64_t* n_blocks) {
  if (!b || !out || !n_blocks || capacity_blocks < 0) return fail(APDGICP_ERR_INVALID_ARG, "bad argument");
  Engine& e = b->eng;
  *n_blocks = 0;
  if (!e.d_stats.p || !e.stats_blocks) return 0;
  const int64_t n = std::min<int64_t>(capacity_blocks, e.stats_blocks);
  APD_HIP(hipMemcpyAsync(out, e.d_stats.as<unsigned long long>() + 16, (size_t)n * 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, e.stream));
  APD_HIP(hipMemsetAsync(e.d_stats.as<unsigned long long>() + 16, 0, (size_t)e.stats_blocks * 3 * sizeof(unsigned long long), e.stream));
  APD_HIP(hipStreamSynchronize(e.stream));
  *n_blocks = n;
  return 0;
}

int apdgicp_batch_last_nn_profile(apdgicp_batch* b, double* total_ms, int64_t* launches, int64_t* pairs_covered) {
  if (!b) return fail(APDGICP_ERR_INVALID_ARG, "batch is null");
  if (b->eng.pool.on) {  // pooled LM batches: the timed launches harvested since the last call (they belong to no single batch)
    if (total_ms) *total_ms = b->eng.pool.nn_ms;
    if (launches) *launches = b->eng.pool.nn_launches;
    if (pairs_covered) *pairs_covered = b->eng.pool.nn_pairs;
    b->eng.pool.nn_ms = 0, b->eng.pool.nn_launches = 0, b->eng.pool.nn_pairs = 0;
    return 0;
  }
  if (total_ms) *total_ms = b->eng.last_nn_ms;
  if (launches) *launches = b->eng.last_nn_launches;
  if (pairs_covered) *pairs_covered = b->eng.last_nn_pairs;
  return 0;
}

int apdgicp_batch_last_nn_kernel(apdgicp_batch* b, char* name, int capacity) {
  if (!b || !name || capacity < 1) return fail(APDGICP_ERR_INVALID_ARG, "bad argument");
  // (pooled LM batches: the kernel of the launches that were timed -- the last launch of all is a few-pair tail of another shape)
  snprintf(name, (size_t)capacity, "%s", b->eng.pool.on && b->eng.pool.timed_kernel[0] ? b->eng.pool.timed_kernel : b->eng.last_nn_kernel);
  return 0;
}

int apdgicp_batch_last_ticks(apdgicp_batch* b, int* ticks, int* nn_sources_per_lane, int* nn_target_splits) {
  if (!b) return fail(APDGICP_ERR_INVALID_ARG, "batch is null");
  if (ticks) *ticks = b->eng.last_ticks;
  if (nn_sources_per_lane) *nn_sources_per_lane = b->eng.nn_S;
  if (nn_target_splits) *nn_target_splits = b->eng.work.T;
  return 0;
}


int apdgicp_batch_pool_counters(apdgicp_batch* b, int64_t* chunks, int64_t* ticks, int64_t* slot_ticks) {
  if (!b) return fail(APDGICP_ERR_INVALID_ARG, "batch is null");
  const Engine::Pool& p = b->eng.pool;
  if (chunks) *chunks = p.n_chunks;
  if (ticks) *ticks = p.n_ticks;
  if (slot_ticks) *slot_ticks = p.n_pair_ticks;
  return 0;
}

// ------------------------------------------------------------------ scan-to-submap target assembly (apd_voxel.hpp)
int apdgicp_submap_create(int device, void* stream, apdgicp_submap** out) {
  return guarded([&]() -> int {
    if (!out) return fail(APDGICP_ERR_INVALID_ARG, "out is null");
    *out = nullptr;
    int count = 0;
    APD_HIP(hipGetDeviceCount(&count));
    if (device < 0 || device >= count) return fail(APDGICP_ERR_INVALID_ARG, "device index out of range");
    APD_HIP(hipSetDevice(device));
    apdgicp_submap* s = new apdgicp_submap;
    s->device = device;
    if (stream) {
      s->stream = (hipStream_t)stream;
    } else {
      const hipError_t e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
      if (e != hipSuccess) {
        delete s;
        return fail(APDGICP_ERR_HIP, hipGetErrorString(e));
      }
      s->own_stream = true;
    }
    if (hipHostMalloc((void**)&s->h_scal, 2 * sizeof(int), hipHostMallocDefault) != hipSuccess || s->scal.ensure(8 * sizeof(int)) < 0) {
      delete s;
      return fail(APDGICP_ERR_HIP, "allocation failed");
    }
    *out = s;
    return 0;
  });
}

int apdgicp_submap_destroy(apdgicp_submap* s) {
  delete s;
  return 0;
}

int apdgicp_submap_assemble(apdgicp_submap* s, int n_clouds, const void* const* xyz, const int64_t* n_points, int64_t stride_bytes,
                            int64_t intensity_offset_bytes, int on_device, const double* rel_poses, const float* leaf, int64_t* n_out) {
  return guarded([&]() -> int {
    if (!s || !xyz || !n_points || !n_out) return fail(APDGICP_ERR_INVALID_ARG, "null argument");
    if (n_clouds < 1 || n_clouds > 4096) return fail(APDGICP_ERR_INVALID_ARG, "n_clouds must be in [1, 4096]");
    if (stride_bytes < 12 || stride_bytes % 4) return fail(APDGICP_ERR_INVALID_ARG, "stride must be a multiple of 4 bytes and >= 12");
    if (intensity_offset_bytes >= 0 && (intensity_offset_bytes % 4 || intensity_offset_bytes + 4 > stride_bytes))
      return fail(APDGICP_ERR_INVALID_ARG, "intensity offset outside the point");
    APD_HIP(hipSetDevice(s->device));
    *n_out = 0;
    s->n_last = 0;
    int64_t total = 0, nmax = 0;
    for (int c = 0; c < n_clouds; c++) {
      if (n_points[c] < 0 || (n_points[c] > 0 && !xyz[c])) return fail(APDGICP_ERR_INVALID_ARG, "bad cloud");
      total += n_points[c], nmax = std::max<int64_t>(nmax, n_points[c]);
    }
    if (total > (1ll << 27)) return fail(APDGICP_ERR_UNSUPPORTED, "more than 2^27 points");
    if (total == 0) return 0;
    // inputs: device pointers are read in place, host clouds go through one staging buffer
    std::vector<SubmapJob> jobs(n_clouds);
    const int stride_f = (int)(stride_bytes / 4);
    if (!on_device) {
      APD_HIP(hipStreamSynchronize(s->stream));  // the staging buffer may still be read by the previous call
      APD_TRY(s->stage.ensure((size_t)total * stride_bytes));
    }
    int64_t off = 0;
    for (int c = 0; c < n_clouds; c++) {
      SubmapJob& j = jobs[c];
      memset(&j, 0, sizeof(j));
      j.n = n_points[c], j.stride = stride_f, j.out_off = (int)off;
      j.intensity_off = intensity_offset_bytes >= 0 ? (int)(intensity_offset_bytes / 4) : -1;
      if (on_device) {
        j.xyz = (const float*)xyz[c];
      } else {
        float* dst = s->stage.as<float>() + off * stride_f;
        // the last point may be shorter than the stride in the caller's buffer: copy up to its last used float only
        const size_t used = std::max<int64_t>(12, intensity_offset_bytes >= 0 ? intensity_offset_bytes + 4 : 12);
        if (j.n > 0) APD_HIP(hipMemcpyAsync(dst, xyz[c], (size_t)(j.n - 1) * stride_bytes + used, hipMemcpyHostToDevice, s->stream));
        j.xyz = dst;
      }
      for (int r = 0; r < 3; r++)
        for (int q = 0; q < 4; q++) j.T[4 * r + q] = rel_poses ? rel_poses[(size_t)c * 16 + r + 4 * q] : (r == q ? 1.0 : 0.0);
      off += j.n;
    }
    APD_TRY(s->jobs.upload(jobs.data(), jobs.size() * sizeof(SubmapJob), s->stream));
    const int n = (int)total;
    APD_TRY(s->cat.ensure((size_t)n * 16));
    hipLaunchKernelGGL(k_submap_transform, dim3((unsigned)((nmax + 255) / 256), (unsigned)n_clouds), dim3(256), 0, s->stream,
                       s->jobs.as<SubmapJob>(), s->cat.as<float4>());
    APD_HIP(hipGetLastError());
    if (!leaf || !(leaf[0] > 0.f)) {  // downsample_method NONE: downsample() returns the cloud itself (:413-415)
      APD_HIP(hipStreamSynchronize(s->stream));
      s->n_last = n, s->last_is_cat = true;
      *n_out = n;
      return 0;
    }
    if (!(leaf[1] > 0.f) || !(leaf[2] > 0.f)) return fail(APDGICP_ERR_INVALID_ARG, "leaf sizes must be positive");
    const float il[3] = {1.f / leaf[0], 1.f / leaf[1], 1.f / leaf[2]};  // inverse_leaf_size_ = Array4f::Ones() / leaf_size_
    int np2 = VOX_TILE;
    while (np2 < n) np2 <<= 1;
    const int nsb = (np2 + SCAN_BLK * SCAN_ITEMS - 1) / (SCAN_BLK * SCAN_ITEMS);
    APD_TRY(s->keys.ensure((size_t)np2 * 8));
    APD_TRY(s->pos.ensure((size_t)np2 * 4));
    APD_TRY(s->bsum.ensure((size_t)nsb * 4));
    APD_TRY(s->out.ensure((size_t)n * 16));
    int* box6 = s->scal.as<int>();
    int* d_total = box6 + 6;
    int* d_err = box6 + 7;
    const int init[8] = {0x7f800000, 0x7f800000, 0x7f800000, (int)0x807fffff, (int)0x807fffff, (int)0x807fffff, 0, 0};
    APD_HIP(hipMemcpyAsync(box6, init, sizeof(init), hipMemcpyHostToDevice, s->stream));  // pageable source: staged before the call returns
    hipLaunchKernelGGL(k_vox_bbox, dim3(std::min((n + 255) / 256, 256)), dim3(256), 0, s->stream, s->cat.as<float4>(), n, box6);
    unsigned long long* keys = s->keys.as<unsigned long long>();
    hipLaunchKernelGGL(k_vox_keys, dim3((np2 + 255) / 256), dim3(256), 0, s->stream, s->cat.as<float4>(), n, np2, box6, il[0], il[1], il[2], keys, d_err);
    hipLaunchKernelGGL(k_bitonic_tile_sort, dim3(np2 / VOX_TILE), dim3(1024), 0, s->stream, keys);
    for (int k = 2 * VOX_TILE; k <= np2; k <<= 1) {
      for (int j = k >> 1; j >= VOX_TILE; j >>= 1)
        hipLaunchKernelGGL(k_bitonic_global, dim3((np2 / 2 + 255) / 256), dim3(256), 0, s->stream, keys, np2, k, j);
      hipLaunchKernelGGL(k_bitonic_tile_merge, dim3(np2 / VOX_TILE), dim3(1024), 0, s->stream, keys, k);
    }
    hipLaunchKernelGGL(k_vox_heads, dim3(nsb), dim3(SCAN_BLK), 0, s->stream, keys, np2, s->pos.as<int>(), s->bsum.as<int>());
    hipLaunchKernelGGL(k_scan_bsum, dim3(1), dim3(SCAN_BLK), 0, s->stream, s->bsum.as<int>(), nsb, d_total);
    hipLaunchKernelGGL(k_vox_centroids, dim3((np2 + 255) / 256), dim3(256), 0, s->stream, keys, s->cat.as<float4>(), s->pos.as<int>(),
                       s->bsum.as<int>(), np2, s->out.as<float4>(), n);
    APD_HIP(hipGetLastError());
    APD_HIP(hipMemcpyAsync(s->h_scal, d_total, 2 * sizeof(int), hipMemcpyDeviceToHost, s->stream));
    APD_HIP(hipStreamSynchronize(s->stream));
    if (s->h_scal[1] == 5) {
      // pcl::VoxelGrid::applyFilter (filters/impl/voxel_grid.hpp): "Leaf size is too small for the input dataset. Integer indices would
      // overflow." is a WARNING there and the output is the input cloud, unfiltered -- so is it here: the assembled (transformed,
      // concatenated) cloud itself, the same line on stderr, success; apdgicp_last_error() holds the message.
      std::fprintf(stderr, "[apdgicp_submap_assemble] Leaf size is too small for the input dataset. Integer indices would overflow.\n");
      g_last_error = "leaf size is too small for the extent of the submap (voxel index overflows int32): the unfiltered cloud is returned, like pcl::VoxelGrid";
      APD_HIP(hipMemsetAsync(d_err, 0, sizeof(int), s->stream));
      s->n_last = n, s->last_is_cat = true;
      *n_out = n;
      return 0;
    }
    if (s->h_scal[1]) return fail(APDGICP_ERR_INTERNAL, "voxel filter failed");
    s->n_last = s->h_scal[0], s->last_is_cat = false;
    *n_out = s->n_last;
    return 0;
  });
}

int apdgicp_submap_points(apdgicp_submap* s, const float** device_xyzi, int64_t* n) {
  if (!s || !device_xyzi || !n) return fail(APDGICP_ERR_INVALID_ARG, "null argument");
  *device_xyzi = s->n_last ? (s->last_is_cat ? s->cat.as<float>() : s->out.as<float>()) : nullptr;
  *n = s->n_last;
  return 0;
}

int apdgicp_submap_copy(apdgicp_submap* s, float* dst_xyzi, int64_t capacity_points, int dst_on_device) {
  return guarded([&]() -> int {
    if (!s || !dst_xyzi) return fail(APDGICP_ERR_INVALID_ARG, "null argument");
    if (capacity_points < s->n_last) return fail(APDGICP_ERR_INVALID_ARG, "destination holds fewer points than the assembled cloud");
    if (!s->n_last) return 0;
    APD_HIP(hipSetDevice(s->device));
    const float* src = s->last_is_cat ? s->cat.as<float>() : s->out.as<float>();
    APD_HIP(hipMemcpyAsync(dst_xyzi, src, (size_t)s->n_last * 16, dst_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s->stream));
    APD_HIP(hipStreamSynchronize(s->stream));
    return 0;
  });
}

}  // extern "C"
