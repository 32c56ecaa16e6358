// Hand-written HIP kernels for the APD-GICP hot path on gfx950 (MI355X, wave64).
//
// Reference path (citations relative to /root/reference/fast_apdgicp/include/fast_gicp/):
//   A = gicp/impl/fast_apdgicp_impl.hpp, L = gicp/impl/lsq_registration_impl.hpp
//
//   k_pack_points      PointXYZI/xyz strided -> float4 {x,y,z,1} (getVector4fMap, A:149)
//   k_knn_cov          calculate_covariances, A:303-363  (exact k-NN + 3x3 eigen + regularisation)
//   k_nn_partial       the 1-NN search of update_correspondences, A:149-153 (brute force, LDS tiles)
//   k_linearize        rest of update_correspondences A:156-192 + linearize A:221-260
//   k_error            compute_error, A:275-298
//   (last block of k_linearize / k_error)  LsqRegistration::step_gn / step_lm, L:107-173, one lane per registration
//   k_finalize         final_transformation_ = x0.cast<float>(), L:78
//
// All kernels are batched: the registration ("pair") index is a grid dimension and every pair
// carries its own state machine, so a whole batch of GN/LM loops advances without host round trips.
// The whole file is compiled with -ffp-contract=off: the fp32 nearest-neighbour arithmetic must not
// be fused (the reference is built without FMA and FLANN's L2_Simple is mul+add).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "apd_math.hpp"
#include "apd_sort.hpp"

namespace apd {

typedef float float2v __attribute__((ext_vector_type(2)));

// All per-point device arrays of a cloud are kept in Z-curve (Morton) order, see apd_sort.hpp;
// `perm` maps a sorted position back to the caller's index and is applied by the getters.
struct CloudDesc {
  const float4* pts;   // n x {x,y,z,1}, SORTED order
  const float4* opts;  // same points in the caller's original order
  const int* perm;     // sorted position -> original index
  const Box* cbox;     // ceil(n/16) chunk boxes
  const Box* gbox;     // ceil(n/128) group boxes
  double* cov;         // SoA [6][n] (sorted order): xx,xy,xz,yy,yz,zz (top-left 3x3 of the reference's Matrix4d)
  int n;
  int pad_;
};
// A pointer a kernel finds in a table in memory (CloudDesc) is generic to the compiler, and every access through it a
// flat_load -- which counts on the LDS counter as well: each LDS access behind it waits for all of them
// (`s_waitcnt vmcnt(0) lgkmcnt(0)`), so no load overlaps with a scan out of LDS.  G(p) says what the pointer is.
#if defined(__HIP_DEVICE_COMPILE__)
#define APD_AS1 __attribute__((address_space(1)))
#else
#define APD_AS1  // (the host pass only parses the kernels)
#endif
template <typename T>
__device__ __forceinline__ const APD_AS1 T* G(const T* p) {
  return (const APD_AS1 T*)p;
}
template <typename T>
__device__ __forceinline__ APD_AS1 T* GW(T* p) {
  return (APD_AS1 T*)p;
}

struct PairDesc {
  int src, tgt;
  CloudDesc s, t;  // copies of clouds[src] / clouds[tgt]: one hop from the pair index to every pointer a kernel needs
};

struct Consts {
  int k, max_iterations, lm_max_iterations, optimizer, regularization;
  int plain_gicp;    // APDGICP_FLAG_PLAIN_GICP: no APD covariance (upstream FastGICP cost)
  int fp32_point;    // APDGICP_FLAG_FP32_POINT_MATH (informational: the engine selects k_linearize<.., true> itself; k_error always evaluates in fp64 over the widened M)
  double thr2;       // corr_dist_threshold_^2 in double (A:156)
  double trans_eps, rot_eps, lm_init_lambda_factor;
  double dist_var_400, sin_az, sin_el;  // A:169-171: distance_variance / 400, sin(azimuth variance), sin(elevation variance)
};

enum { ST_NEED_LIN = 0, ST_NEED_ERR = 1, ST_DONE = 2 };

struct PairState {
  Rigid x0, xi, delta;
  double H[36], b[6], d[6], final_H[36];
  double y0, yi, lambda, nu;
  int status, converged, iter, inner, n_lin, n_err, failed, n_matched;
};

// result record, identical to apdgicp_result in include/apdgicp_hip.h (static_assert'ed in the engine)
struct ResultRec {
  float T[16];
  double final_cost;
  int converged, iterations, n_linearize, n_compute_error, lm_failed, n_matched;
};

constexpr int kRed = 32;        // doubles per block partial: 21 H + 6 b + cost + matched + pad
constexpr int kChunk = 16;      // NN index granularity: the winner is located inside a 16-target chunk
constexpr unsigned kNoChunk = 0xFFFFFFFFu;
constexpr unsigned kTieBit = 0x80000000u;  // set in a chunk id when the minimum was reached in more than one chunk
constexpr unsigned kKeptBit = 0x40000000u; // set in a chunk id when the search PROVED that the previous neighbour (Work::nnpt) is still the
                                           // unique nearest neighbour: k_linearize takes index and point from there, no re-scan of the chunk
constexpr unsigned kChunkMask = ~(kTieBit | kKeptBit);

// Second moments of the k neighbours about the query point, summed in ONE order by every covariance kernel: four partial
// sums -- partial j takes the ranks j, j + 4, j + 8, ... in increasing order -- combined as (p0 + p1) + (p2 + p3).  The
// 4-lanes-per-query kernel gives each lane one partial (a quarter of the fp64 work the shared rank-order sum cost every lane);
// the one-lane-per-query kernels keep four accumulators.  A:323-324 sums in Eigen's order; parity with it is by tolerance.
struct Mom9 {
  double s1x = 0, s1y = 0, s1z = 0, sxx = 0, sxy = 0, sxz = 0, syy = 0, syz = 0, szz = 0;
  __device__ __forceinline__ void add(double x, double y, double z) {
    s1x += x, s1y += y, s1z += z;
    sxx += x * x, sxy += x * y, sxz += x * z, syy += y * y, syz += y * z, szz += z * z;
  }
  __device__ __forceinline__ void merge(const Mom9& o) {
    s1x += o.s1x, s1y += o.s1y, s1z += o.s1z;
    sxx += o.sxx, sxy += o.sxy, sxz += o.sxz, syy += o.syy, syz += o.syz, szz += o.szz;
  }
};
struct Mom9x4 {  // one lane per query
  Mom9 p0, p1, p2, p3;
  __device__ __forceinline__ void add(int rank, double x, double y, double z) {
    switch (rank & 3) {
      case 0: p0.add(x, y, z); break;
      case 1: p1.add(x, y, z); break;
      case 2: p2.add(x, y, z); break;
      default: p3.add(x, y, z); break;
    }
  }
  __device__ __forceinline__ Mom9 total() {
    p0.merge(p1), p2.merge(p3), p0.merge(p2);
    return p0;
  }
};

// A one-pair poll without its own launch: the kernel that ends the tick writes what k_finalize would (see there).
struct PollPost {
  ResultRec* out;        // device record
  int* status_out;       // device status word, error flag behind it
  ResultRec* host_out;   // the same in pinned host memory
  int* host_status;
  int* host_seq;         // receives `seq` when everything above has been written through
  const int* err_flag;
};

struct Work {
  unsigned long long* nnpart;  // [pair][split][nstride]  (fp32 bits of min sqdist << 32 | chunk id)
  int* corr;                   // [pair][nstride]   correspondences_ (A:156)
  float4* nnpt;                // [pair][nstride]   nearest neighbour found by the previous linearize, ungated: its coordinates and, in
                               //                   .w, the bits of its sorted index (-1: none) -- one coalesced load instead of index + gather:
                               //                   a warm start for the pruned search, never an input of the result
  float4* nnaux;               // [pair][nstride]   {transformed point, s} at the point's last FULL search: every target other than the
                               //                   neighbour found then was at a computed squared distance >= s.  While the pose moves the
                               //                   point by less than the gap between s and the neighbour, the neighbour cannot change and
                               //                   the search is skipped for this point (see nn_search); s = 0: no such knowledge
  float skin_mul, skin_add;    // a full search prunes with the radius best * skin_mul + skin_add instead of best, so that s exceeds the
                               // neighbour distance by a margin (skin_mul = 1, skin_add = 0: off)
  float* sqd;                  // [pair][nstride]   sq_distances_ (A:153)
  double* maha;                // [pair][6][nstride] mahalanobis_ upper triangle (A:191)
  double* blkpart;             // [pair][nblk_max][kRed]
  double* errpart;             // [pair][nblk_max]
  int nstride, nblk_max, T;
  float cap;                   // pruned searches: squared search radius (+inf: unbounded).  The optimiser ticks pass the smallest float
                               // >= corr_dist_threshold^2: a point without any target inside it has no correspondence whatever its
                               // true neighbour is (A:156), so the search may stop there (sqd then holds the cap, the hint index -1)
  int pair0, npairs;           // this launch covers pairs [pair0, pair0 + gridDim pairs) of npairs (one stream per pair group)
  const int* active;           // optional: the launch covers active[pair0 ...] instead -- the pairs that are not DONE yet (LM batches whose
                               // pairs converge after very different numbers of iterations would otherwise launch mostly idle blocks)
  int* ticket;                 // [2][pairs] arrival counters of k_linearize / k_error blocks (last block runs the LM step)
  unsigned long long* stats;   // optional diagnostics (null): [0] groups scanned, [1] chunks tested, [2] chunks scanned, [3] waves
  unsigned* blk_cost;          // optional (one dense pair, more blocks than the GPU holds at once): what the search of every block of 64 source points cost this tick
  const unsigned* blk_order;   // optional: launch position -> block, the costliest first (k_block_order, round 6)
  unsigned long long* timeline; // APDGICP_STATS=2 (diagnostics builds, APD_BLOCK_TIMELINE): the block timeline behind the 16 counters; the counters themselves are off then
  int stats_blocks;            // APDGICP_STATS=2: behind the 16 counters, room for the timeline of this many blocks of the LAST k_nn_pruned launch:
                               // {start, end} in 100 MHz wall-clock ticks and the block's index, three words per block (tools/c5_blocks.py)
  int coop_search;             // k_nn_compact: a block with at most 64 points left searches them with all of its waves
  int sparse_max;              // k_nn_compact: ... and with at most this many, one point at a time with a whole wave (0: never)
  int xf_linear;               // fp32 summation order of T * p (A:149), see xf_row: 0 = pairwise (Eigen >= 3.3), 1 = linear chain (Eigen 3.2)
  const PollPost* post;        // non-null (k_error only): this launch is the last of a one-pair poll and writes the result record itself
                               // (post_result).  Not in k_linearize: with the record code in its last block the register allocator
                               // moved that kernel's spills into the per-point pass (32.6 -> 45.0 us per batch launch)
  int post_seq;
  double* trace;               // debug (apdgicp_set_trace; null otherwise): the optimiser's per-iteration trace of pair 0, see trace_trial
  const Rigid* init;           // non-null in the FIRST tick of an align (fused optimiser): the guesses, one pose per pair.  The
                               // search and the per-point pass take their pose from there and run cold, and the last block of
                               // k_linearize builds the pair's state from scratch (L:56-59) instead of loading it: no k_init_state launch
};

// L:56-59: x0 = guess.cast<double>() (the host widens it: a pose read through a uniform pointer stays in scalar registers),
// lm_lambda_ = -1, converged_ = false
__device__ __forceinline__ void init_pair_state(PairState& s, const Rigid* g /* or null: identity */, int max_iterations) {
  s.x0 = g ? *g : rigid_identity();
  s.xi = s.x0;
  s.delta = rigid_identity();
  for (int q = 0; q < 36; q++) s.H[q] = 0.0, s.final_H[q] = (q % 7 == 0) ? 1.0 : 0.0;
  for (int q = 0; q < 6; q++) s.b[q] = 0.0, s.d[q] = 0.0;
  s.y0 = s.yi = 0.0;
  s.lambda = -1.0;
  s.nu = 2.0;
  s.status = max_iterations > 0 ? ST_NEED_LIN : ST_DONE;
  s.converged = 0, s.iter = 0, s.inner = 0, s.n_lin = 0, s.n_err = 0, s.failed = 0, s.n_matched = 0;
}

__device__ __forceinline__ ResultRec result_record(const PairState& s) {
  ResultRec r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 4; j++) r.T[i + 4 * j] = (float)s.x0.m[4 * i + j];  // x0.cast<float>(), L:78
  r.T[3] = 0.f, r.T[7] = 0.f, r.T[11] = 0.f, r.T[15] = 1.f;
  r.final_cost = s.y0;
  r.converged = s.converged;
  r.iterations = s.iter;
  r.n_linearize = s.n_lin;
  r.n_compute_error = s.n_err;
  r.lm_failed = s.failed;
  r.n_matched = s.n_matched;
  return r;
}
// one thread, pair 0 of a one-pair handle; every block of this align that could have raised the error flag has passed the
// arrival counter (or finished with an earlier launch) before this runs
__device__ __forceinline__ void post_result(const PollPost& pp, int seq, const PairState& s) {
  const ResultRec r = result_record(s);
  const int flag = __hip_atomic_load(pp.err_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  pp.out[0] = r, pp.status_out[0] = s.status, pp.status_out[1] = flag;
  pp.host_out[0] = r, pp.host_status[0] = s.status, pp.host_status[1] = flag;
  __threadfence_system();
  __hip_atomic_store(pp.host_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ __forceinline__ int pair_of(const Work& w, unsigned block) { return w.active ? w.active[w.pair0 + block] : w.pair0 + (int)block; }

// XCD-aware block -> work mapping for 2-D grids (x: blocks of one pair / cloud, y: pairs / clouds).  Workgroups are
// dispatched round-robin over the 8 XCDs (block b runs on XCD b % 8, observed; a speed assumption only), each with its
// own L2: the linear work index is permuted so that every XCD gets one contiguous eighth of the work, i.e. the blocks
// that read the same target cloud share an L2 instead of pulling it into all eight.
__device__ __forceinline__ void xcd_remap(unsigned& bx, unsigned& by) {
  const unsigned nx = gridDim.x, U = nx * gridDim.y, lid = blockIdx.y * nx + blockIdx.x;
  const unsigned u = (U & 7u) == 0 ? (lid & 7u) * (U >> 3) + (lid >> 3) : lid;
  bx = u % nx, by = u / nx;
}

// ----------------------------------------------------------------------------------------------
// fp32 helpers: the exact operation order of the reference (see oracle/apdgicp_ref.cpp)
__device__ __forceinline__ void load_Tf(const Rigid& T, float Tf[12]) {
#pragma unroll
  for (int i = 0; i < 12; i++) Tf[i] = (float)T.m[i];  // trans.cast<float>(), A:137
}
// One row of `trans_f * p.getVector4fMap()` (A:149; the fourth coefficient of the point is 1, so the last product is t itself).
// Two fp32 summation orders exist in the Eigen releases the reference can be built against:
//   pairwise (default)  (r0 x + r1 y) + (r2 z + t)    Eigen >= 3.3: coefficient-based lazy product, the 4 products summed by
//                                                      redux_novec_unroller, which halves the range
//   linear chain        ((r0 x + r1 y) + r2 z) + t    Eigen 3.2: product_coeff_impl accumulates from the left
// `linear` is Work::xf_linear (APDGICP_FLAG_XF_LINEAR_CHAIN), uniform over the launch.  No FMA either way (-ffp-contract=off).
__device__ __forceinline__ float xf_row(const float* r, float x, float y, float z, int linear) {
  const float a = r[0] * x + r[1] * y, c = r[2] * z;
  return linear ? (a + c) + r[3] : a + (c + r[3]);
}

typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float sqdist1(float tx, float ty, float tz, float px, float py, float pz) {
  const float dx = tx - px, dy = ty - py, dz = tz - pz;
  float r = dx * dx;
  r = r + dy * dy;
  r = r + dz * dz;
  return r;
}

// ----------------------------------------------------------------------------------------------
__global__ void k_pack_points(const char* raw, long long stride_bytes, int n, float4* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* p = (const float*)(raw + (long long)i * stride_bytes);
  out[i] = make_float4(p[0], p[1], p[2], 1.0f);
}

// sorted SoA covariances -> n x Matrix4d (column-major) in the caller's point order
struct PackJob {
  const char* raw;
  float4* out;
  long long stride_bytes;
  int n;
  int pad_;
};
__global__ void k_pack_points_multi(const PackJob* jobs) {
  const PackJob j = jobs[blockIdx.y];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= j.n) return;
  const float* p = (const float*)(j.raw + (long long)i * j.stride_bytes);
  j.out[i] = make_float4(p[0], p[1], p[2], 1.0f);
}

__global__ void k_unpack_cov(const double* cov6, const int* perm, int n, double* out16) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double xx = cov6[i], xy = cov6[n + i], xz = cov6[2 * n + i], yy = cov6[3 * n + i], yz = cov6[4 * n + i], zz = cov6[5 * n + i];
  double* o = out16 + 16ll * perm[i];
  o[0] = xx, o[1] = xy, o[2] = xz, o[3] = 0;
  o[4] = xy, o[5] = yy, o[6] = yz, o[7] = 0;
  o[8] = xz, o[9] = yz, o[10] = zz, o[11] = 0;
  o[12] = 0, o[13] = 0, o[14] = 0, o[15] = 0;
}

__global__ void k_pack_cov(const double* in16, const int* perm, int n, double* cov6) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double* o = in16 + 16ll * perm[i];
  cov6[i] = o[0], cov6[n + i] = o[4], cov6[2 * n + i] = o[8], cov6[3 * n + i] = o[5], cov6[4 * n + i] = o[9], cov6[5 * n + i] = o[10];
}

// correspondences_ / sq_distances_ from the internal (sorted source, sorted target index) form to the
// caller's indexing
__global__ void k_export_corr(const int* corr, const float* sqd, const int* perm_src, const int* perm_tgt, int n, int* out_corr, float* out_sqd) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  const int o = perm_src[s], j = corr[s];
  if (out_corr) out_corr[o] = j >= 0 ? perm_tgt[j] : -1;
  if (out_sqd) out_sqd[o] = sqd[s];
}

// the ungated nearest neighbour of every source point as the last search + per-point pass left it (Work::nnpt: sorted target
// index in .w, -1: none; Work::sqd), in the caller's indexing -- apdgicp_nearest_neighbours
__global__ void k_export_nn(const float4* nnpt, const float* sqd, const int* perm_src, const int* perm_tgt, int n, int* out_idx, float* out_sqd) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  const int o = perm_src[s], j = __float_as_int(nnpt[s].w);
  out_idx[o] = j >= 0 ? perm_tgt[j] : -1;
  out_sqd[o] = sqd[s];
}

// caller's order: n x {x, y, z} floats out of a cloud's float4 copy (apdgicp_get_points)
__global__ void k_unpack_points(const float4* opts, int n, float* out3) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 p = opts[i];
  out3[3 * i] = p.x, out3[3 * i + 1] = p.y, out3[3 * i + 2] = p.z;
}

// ----------------------------------------------------------------------------------------------
// k_knn_cov: exact k nearest neighbours of every point inside its own cloud (self included), then
// population covariance and regularisation.  One thread per query, candidates streamed through
// LDS tiles (wave-uniform address -> broadcast ds_read_b128).
//   sweep 1: minima of 32 index-strided candidate classes.  k distinct points lie within the k-th
//            smallest of those minima, so it bounds the k-th neighbour distance (tau).
//   sweep 2: collect every candidate with (d, idx) <= tau into a per-thread LDS list (expected
//            ~1.5 k entries); on overflow tau is tightened from the list and the sweep repeats.
//   select : k rounds of lexicographic (d, idx) min-extraction; accumulate sums in fp64.
//   KNN_CAP: list entries per query -- 64 for k <= 32; 128 for 32 < k <= 64 (setCorrespondenceRandomness takes any k, A:45-47;
//   the launch file ships 20), where sweep 1 runs twice for 64 classes: the slow path of a setting nobody ships.
constexpr int KNN_BLK = 128, KNN_NC = 32, KNN_TILE = 1024;
__host__ __device__ constexpr int knn_lds_bytes_brute(int cap) { return KNN_TILE * 16 + cap * KNN_BLK * 8; }

__device__ __forceinline__ void knn_load_tile(float4* tile, const float4* pts, int t0, int n, int tid) {
  const float inf = __builtin_inff();
  for (int e = tid; e < KNN_TILE; e += KNN_BLK) {
    const int j = t0 + e;
    tile[e] = j < n ? pts[j] : make_float4(inf, inf, inf, 0.f);
  }
}

template <int KNN_CAP>
__global__ __launch_bounds__(KNN_BLK) void k_knn_cov(const CloudDesc* clouds, const int* cloud_ids, int k, int reg, int* err_flag) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float4* tile = (float4*)smem;
  int* lst_i = (int*)(smem + KNN_TILE * 16);
  float* lst_d = (float*)(lst_i + KNN_CAP * KNN_BLK);

  const CloudDesc c = clouds[cloud_ids[blockIdx.y]];
  double* cov = c.cov;
  const int n = c.n, tid = threadIdx.x;
  if ((int)(blockIdx.x * KNN_BLK) >= n) return;  // block-uniform
  const int i = blockIdx.x * KNN_BLK + tid;
  const bool valid = i < n;
  const float4 q = c.pts[valid ? i : n - 1];
  const float inf = __builtin_inff();

  // ---- sweep 1
  float cm[KNN_NC];
  float tau_d = inf;
  if constexpr (KNN_CAP <= 64) {
#pragma unroll
  for (int s = 0; s < KNN_NC; s++) cm[s] = inf;
  for (int t0 = 0; t0 < n; t0 += KNN_TILE) {
    __syncthreads();
    knn_load_tile(tile, c.pts, t0, n, tid);
    __syncthreads();
    const int cnt = min(KNN_TILE, (n - t0 + KNN_NC - 1) / KNN_NC * KNN_NC);
    for (int jj = 0; jj < cnt; jj += KNN_NC) {
#pragma unroll
      for (int s = 0; s < KNN_NC; s++) {
        const float4 t = tile[jj + s];
        cm[s] = fminf(cm[s], sqdist1(t.x, t.y, t.z, q.x, q.y, q.z));
      }
    }
  }
  // bitonic sort of the 32 minima (static indices only)
#pragma unroll
  for (int kk = 2; kk <= KNN_NC; kk <<= 1) {
#pragma unroll
    for (int j = kk >> 1; j > 0; j >>= 1) {
#pragma unroll
      for (int a = 0; a < KNN_NC; a++) {
        const int l = a ^ j;
        if (l > a) {
          const bool up = (a & kk) == 0;
          const float x = cm[a], y = cm[l];
          const float lo = fminf(x, y), hi = fmaxf(x, y);
          cm[a] = up ? lo : hi;
          cm[l] = up ? hi : lo;
        }
      }
    }
  }
#pragma unroll
  for (int s = 0; s < KNN_NC; s++)
    if (s == k - 1) tau_d = cm[s];
  } else {
    // 32 < k <= 64: SIXTY-FOUR index-strided classes (two passes of 32 register minima, parked in the still empty list rows),
    // tau = the k-th smallest class minimum.  (Until round 4 this path started from tau = +inf and on overflow kept the first
    // 128 candidates in curve order -- not a sample: the candidate set shrank so slowly that clouds beyond ~16k points ran
    // into the restart guard, error flag 1 and garbage covariances on a supported setting.)
    constexpr int NCL2 = 2 * KNN_NC;
    static_assert(KNN_TILE % NCL2 == 0 && KNN_CAP >= NCL2, "class layout");
    for (int pass = 0; pass < 2; pass++) {
#pragma unroll
      for (int s = 0; s < KNN_NC; s++) cm[s] = inf;
      for (int t0 = 0; t0 < n; t0 += KNN_TILE) {
        __syncthreads();
        knn_load_tile(tile, c.pts, t0, n, tid);   // (padded with +inf up to the tile size, a multiple of 64)
        __syncthreads();
        const int cnt = min(KNN_TILE, (n - t0 + NCL2 - 1) / NCL2 * NCL2);
        for (int jj = 0; jj < cnt; jj += NCL2) {
#pragma unroll
          for (int s = 0; s < KNN_NC; s++) {
            const float4 t = tile[jj + KNN_NC * pass + s];
            cm[s] = fminf(cm[s], sqdist1(t.x, t.y, t.z, q.x, q.y, q.z));
          }
        }
      }
#pragma unroll
      for (int s = 0; s < KNN_NC; s++) lst_d[(KNN_NC * pass + s) * KNN_BLK + tid] = cm[s];   // (this thread's own column: no barrier needed)
    }
    float last_d = -1.f;
    int last_a = -1;
    for (int r = 0; r < k; r++) {  // k-th smallest of the 64 minima: (value, class) in lexicographic order, one per round
      float bd = inf;
      int ba = 0x7fffffff;
      for (int a = 0; a < NCL2; a++) {
        const float d = lst_d[a * KNN_BLK + tid];
        const bool gt = d > last_d || (d == last_d && a > last_a);
        const bool lt = d < bd || (d == bd && a < ba);
        if (gt && lt) bd = d, ba = a;
      }
      last_d = bd, last_a = ba;
    }
    tau_d = last_d;
  }
  int tau_i = 0x7fffffff;

  // ---- sweep 2 (+ tightening restarts)
  bool need = valid;
  int cnt_final = 0;
  for (int round = 0;; round++) {
    int cnt = 0;
    for (int t0 = 0; t0 < n; t0 += KNN_TILE) {
      __syncthreads();
      knn_load_tile(tile, c.pts, t0, n, tid);
      __syncthreads();
      const int m = min(KNN_TILE, n - t0);
      if (need) {
#pragma unroll 4
        for (int jj = 0; jj < m; jj++) {
          const float4 t = tile[jj];
          const float d = sqdist1(t.x, t.y, t.z, q.x, q.y, q.z);
          const int j = t0 + jj;
          if (d <= tau_d) {
            const int o = c.perm[j];  // ties are ordered by the caller's ORIGINAL index, like the reference restatement
            if (d < tau_d || o <= tau_i) {
              if (cnt < KNN_CAP) {
                lst_i[cnt * KNN_BLK + tid] = o;
                lst_d[cnt * KNN_BLK + tid] = d;
              }
              cnt++;
            }
          }
        }
      }
    }
    bool ovf = false;
    if (need) {
      ovf = cnt > KNN_CAP;
      cnt_final = min(cnt, KNN_CAP);
      if (ovf) {  // tighten tau to the k-th smallest (d, idx) of the stored entries
        float last_d = -1.f;
        int last_i = -1;
        for (int r = 0; r < k; r++) {
          float bd = inf;
          int bi = 0x7fffffff;
          for (int a = 0; a < KNN_CAP; a++) {
            const float d = lst_d[a * KNN_BLK + tid];
            const int ii = lst_i[a * KNN_BLK + tid];
            const bool gt = d > last_d || (d == last_d && ii > last_i);
            const bool lt = d < bd || (d == bd && ii < bi);
            if (gt && lt) bd = d, bi = ii;
          }
          last_d = bd, last_i = bi;
        }
        tau_d = last_d, tau_i = last_i;
      }
    }
    need = ovf;
    if (!__syncthreads_or(ovf ? 1 : 0)) break;
    if (round >= 4096) {  // cannot happen: of the KNN_CAP stored candidates only k survive a restart, so each one removes at least KNN_CAP - k
      if (tid == 0) atomicExch(err_flag, 1);
      break;
    }
  }
  if (!valid) return;

  // ---- select the k nearest, accumulate in fp64 relative to the query point (differences of
  // fp32 numbers are exact in fp64); cov = S2/k - mean*mean^T  ==  A:323-324
  Mom9x4 acc;
  {
    float last_d = -1.f;
    int last_i = -1;
    for (int r = 0; r < k; r++) {
      float bd = inf;
      int bi = 0x7fffffff;
      for (int a = 0; a < cnt_final; a++) {
        const float d = lst_d[a * KNN_BLK + tid];
        const int ii = lst_i[a * KNN_BLK + tid];
        const bool gt = d > last_d || (d == last_d && ii > last_i);
        const bool lt = d < bd || (d == bd && ii < bi);
        if (gt && lt) bd = d, bi = ii;
      }
      last_d = bd, last_i = bi;
      if (bi == 0x7fffffff) {  // fewer than k candidates: impossible when n >= k
        atomicExch(err_flag, 2);
        break;
      }
      const float4 p = c.opts[bi];
      acc.add(r, (double)p.x - (double)q.x, (double)p.y - (double)q.y, (double)p.z - (double)q.z);
    }
  }
  const Mom9 mt = acc.total();
  const double s1x = mt.s1x, s1y = mt.s1y, s1z = mt.s1z, sxx = mt.sxx, sxy = mt.sxy, sxz = mt.sxz, syy = mt.syy, syz = mt.syz, szz = mt.szz;
  const double ik = 1.0 / (double)k;
  const double mx = s1x * ik, my = s1y * ik, mz = s1z * ik;
  Sym3 pc;
  pc.xx = sxx * ik - mx * mx, pc.xy = sxy * ik - mx * my, pc.xz = sxz * ik - mx * mz;
  pc.yy = syy * ik - my * my, pc.yz = syz * ik - my * mz, pc.zz = szz * ik - mz * mz;
  Sym3 out;
  if (!regularize_cov(reg, pc, out)) atomicExch(err_flag, 3);
  cov[i] = out.xx, cov[n + i] = out.xy, cov[2 * n + i] = out.xz, cov[3 * n + i] = out.yy, cov[4 * n + i] = out.yz, cov[5 * n + i] = out.zz;
}

// ----------------------------------------------------------------------------------------------
// k_knn_cov_select: calculate_covariances for ANY k (the reference takes any k, A:45-47; it ships 20).  One wave owns SEL_Q
// consecutive sorted queries and keeps no list: the k-th smallest key (distance bits, original index) is found by bisection -- on the
// bits of the fp32 distance, then, only when several points share the k-th distance and not all of them fit, on the original
// index -- every probe ONE sweep of the cloud counting the keys at or below it; a last sweep adds up the moments of the keys at or
// below the k-th.  About 35 sweeps of the cloud per SEL_Q queries: for experiments with k = 65 ... n, not for 10 Hz.  Exact, and
// in the (distance, original index) order of every other covariance kernel; the sums run in curve order (compared by tolerance).
constexpr int SEL_Q = 4;
__global__ __launch_bounds__(64) void k_knn_cov_select(const CloudDesc* clouds, const int* cloud_ids, int k, int reg, int* err_flag, int block0) {
  const CloudDesc c = clouds[cloud_ids[blockIdx.y]];
  const int n = c.n, lane = threadIdx.x;
  const int i0 = (block0 + (int)blockIdx.x) * SEL_Q;  // (block0: the launch is cut into dispatches of bounded duration, Engine::launch_knn)
  if (i0 >= n) return;
  const float4* pts = G(c.pts);
  float qx[SEL_Q], qy[SEL_Q], qz[SEL_Q];
#pragma unroll
  for (int u = 0; u < SEL_Q; u++) {
    const float4 q = pts[min(i0 + u, n - 1)];
    qx[u] = q.x, qy[u] = q.y, qz[u] = q.z;
  }
  auto wave_sum_i = [](int v) {
#pragma unroll
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
  };
  // D[u]: the smallest distance with at least k points at or below it (distances are finite and >= 0: their bits order like the values)
  unsigned lo[SEL_Q], hi[SEL_Q];
#pragma unroll
  for (int u = 0; u < SEL_Q; u++) lo[u] = 0u, hi[u] = 0x7F800000u;
  for (int it = 0; it < 32; it++) {
    unsigned mid[SEL_Q];
    bool open = false;
#pragma unroll
    for (int u = 0; u < SEL_Q; u++) mid[u] = lo[u] + ((hi[u] - lo[u]) >> 1), open |= lo[u] < hi[u];
    if (!open) break;
    int cnt[SEL_Q];
#pragma unroll
    for (int u = 0; u < SEL_Q; u++) cnt[u] = 0;
    for (int j = lane; j < n; j += 64) {
      const float4 t = pts[j];
#pragma unroll
      for (int u = 0; u < SEL_Q; u++) cnt[u] += __float_as_uint(sqdist1(t.x, t.y, t.z, qx[u], qy[u], qz[u])) <= mid[u] ? 1 : 0;
    }
#pragma unroll
    for (int u = 0; u < SEL_Q; u++) {
      const int tot = wave_sum_i(cnt[u]);
      if (lo[u] < hi[u]) {
        if (tot >= k) hi[u] = mid[u];
        else lo[u] = mid[u] + 1u;
      }
    }
  }
  // how many lie strictly below D, how many at D: all of the latter belong to the answer unless more than k - below share D
  int need[SEL_Q];
  unsigned ilo[SEL_Q], ihi[SEL_Q];
  bool surplus = false;
  {
    int clt[SEL_Q], ceq[SEL_Q];
#pragma unroll
    for (int u = 0; u < SEL_Q; u++) clt[u] = 0, ceq[u] = 0;
    for (int j = lane; j < n; j += 64) {
      const float4 t = pts[j];
#pragma unroll
      for (int u = 0; u < SEL_Q; u++) {
        const unsigned d = __float_as_uint(sqdist1(t.x, t.y, t.z, qx[u], qy[u], qz[u]));
        clt[u] += d < hi[u] ? 1 : 0, ceq[u] += d == hi[u] ? 1 : 0;
      }
    }
#pragma unroll
    for (int u = 0; u < SEL_Q; u++) {
      const int lt = wave_sum_i(clt[u]), eq = wave_sum_i(ceq[u]);
      need[u] = k - lt;
      ilo[u] = 0u, ihi[u] = 0x7FFFFFFFu;
      if (eq > need[u]) surplus = true;
      else ilo[u] = ihi[u];  // every point at D is taken
    }
  }
  if (surplus) {  // (uniform) I[u]: the smallest original index with `need` points at D at or below it
    for (int it = 0; it < 32; it++) {
      unsigned mid[SEL_Q];
      bool open = false;
#pragma unroll
      for (int u = 0; u < SEL_Q; u++) mid[u] = ilo[u] + ((ihi[u] - ilo[u]) >> 1), open |= ilo[u] < ihi[u];
      if (!open) break;
      int cnt[SEL_Q];
#pragma unroll
      for (int u = 0; u < SEL_Q; u++) cnt[u] = 0;
      for (int j = lane; j < n; j += 64) {
        const float4 t = pts[j];
#pragma unroll
        for (int u = 0; u < SEL_Q; u++)
          cnt[u] += (__float_as_uint(sqdist1(t.x, t.y, t.z, qx[u], qy[u], qz[u])) == hi[u] && __float_as_uint(t.w) <= mid[u]) ? 1 : 0;
      }
#pragma unroll
      for (int u = 0; u < SEL_Q; u++) {
        const int tot = wave_sum_i(cnt[u]);
        if (ilo[u] < ihi[u]) {
          if (tot >= need[u]) ihi[u] = mid[u];
          else ilo[u] = mid[u] + 1u;
        }
      }
    }
  }
  // moments of the k points, relative to the query (exact differences)
  Mom9 mp[SEL_Q];
  int taken[SEL_Q];
#pragma unroll
  for (int u = 0; u < SEL_Q; u++) taken[u] = 0;
  for (int j = lane; j < n; j += 64) {
    const float4 t = pts[j];
#pragma unroll
    for (int u = 0; u < SEL_Q; u++) {
      const unsigned d = __float_as_uint(sqdist1(t.x, t.y, t.z, qx[u], qy[u], qz[u]));
      if (d < hi[u] || (d == hi[u] && __float_as_uint(t.w) <= ihi[u])) {
        mp[u].add((double)t.x - (double)qx[u], (double)t.y - (double)qy[u], (double)t.z - (double)qz[u]);
        taken[u]++;
      }
    }
  }
  auto wave_sum_d = [](double v) {
#pragma unroll
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
  };
#pragma unroll
  for (int u = 0; u < SEL_Q; u++) {
    const int i = i0 + u;
    const int tot = wave_sum_i(taken[u]);
    const double s1x = wave_sum_d(mp[u].s1x), s1y = wave_sum_d(mp[u].s1y), s1z = wave_sum_d(mp[u].s1z);
    const double sxx = wave_sum_d(mp[u].sxx), sxy = wave_sum_d(mp[u].sxy), sxz = wave_sum_d(mp[u].sxz);
    const double syy = wave_sum_d(mp[u].syy), syz = wave_sum_d(mp[u].syz), szz = wave_sum_d(mp[u].szz);
    if (i >= n || lane != 0) continue;
    if (tot != k) atomicExch(err_flag, 2);  // (cannot happen for k <= n)
    const double ik = 1.0 / (double)k;
    const double mx = s1x * ik, my = s1y * ik, mz = s1z * ik;
    Sym3 pc;
    pc.xx = sxx * ik - mx * mx, pc.xy = sxy * ik - mx * my, pc.xz = sxz * ik - mx * mz;
    pc.yy = syy * ik - my * my, pc.yz = syz * ik - my * mz, pc.zz = szz * ik - mz * mz;
    Sym3 out;
    if (!regularize_cov(reg, pc, out)) atomicExch(err_flag, 3);
    double* cov = c.cov;
    cov[i] = out.xx, cov[n + i] = out.xy, cov[2 * n + i] = out.xz, cov[3 * n + i] = out.yy, cov[4 * n + i] = out.yz, cov[5 * n + i] = out.zz;
  }
}

// ----------------------------------------------------------------------------------------------
// k_nn_partial: brute-force nearest neighbour of T*p_i in the target (A:149-153), exact fp32
// difference form.  grid = (source blocks, target splits, pairs); block = 256 lanes, S sources per
// lane held as packed float2 registers so the distance math issues as v_pk_{add,mul}_f32.
// A block stages its target split through LDS in 1024-point tiles; all lanes of a wave read the
// same target (broadcast ds_read_b128).  The running minimum is kept per 16-target chunk
// (v_min3_f32) and the winning chunk id, not the index, is tracked: k_linearize re-scans that one
// chunk to recover the exact index (lowest index on ties).
constexpr int NN_BLK = 256, NN_TILE = 1024;

template <int S>
__global__ __launch_bounds__(NN_BLK) void k_nn_partial(const CloudDesc* clouds, const PairDesc* pairs, const PairState* st, Work w) {
  static_assert(S % 2 == 0, "S must be even");
  // LDS tile, two targets per entry: {x0,x1,y0,y1} (ds_read_b128) + {z0,z1} (ds_read_b64): every byte read is
  // used and each coordinate pair sits in one 64-bit register pair for the v_pk op_sel splats.
  __shared__ float4 txy[NN_TILE / 2];
  __shared__ float2 tz[NN_TILE / 2];
  const int pair = pair_of(w, blockIdx.z);
  if (st[pair].status != ST_NEED_LIN) return;
  const PairDesc pd = pairs[pair];
  const CloudDesc src = pd.s, tgt = pd.t;
  const int N = src.n, M = tgt.n, tid = threadIdx.x;
  const int base = blockIdx.x * (NN_BLK * S);
  if (base >= N) return;
  float Tf[12];
  load_Tf(st[pair].x0, Tf);

  float2v px[S / 2], py[S / 2], pz[S / 2];
#pragma unroll
  for (int s = 0; s < S; s++) {
    const int i = base + s * NN_BLK + tid;
    const float4 p = src.pts[i < N ? i : N - 1];
    const float x = xf_row(Tf + 0, p.x, p.y, p.z, w.xf_linear), y = xf_row(Tf + 4, p.x, p.y, p.z, w.xf_linear), z = xf_row(Tf + 8, p.x, p.y, p.z, w.xf_linear);
    if (s & 1) px[s / 2].y = x, py[s / 2].y = y, pz[s / 2].y = z;
    else px[s / 2].x = x, py[s / 2].x = y, pz[s / 2].x = z;
  }
  const float inf = __builtin_inff();
  float best[S];
  unsigned bestc[S];  // winning chunk | kTieBit when a second chunk reached exactly the same minimum
#pragma unroll
  for (int s = 0; s < S; s++) best[s] = inf, bestc[s] = kNoChunk;

  const int nchunks = (M + kChunk - 1) / kChunk;
  const int per = (nchunks + w.T - 1) / w.T;
  const int c0 = blockIdx.y * per, c1 = min(c0 + per, nchunks);
  for (int tc0 = c0; tc0 < c1; tc0 += NN_TILE / kChunk) {
    __syncthreads();
    const int jend = min(c1 * kChunk, M);
    for (int e = tid; e < NN_TILE / 2; e += NN_BLK) {
      const int j = tc0 * kChunk + 2 * e;
      const float4 a = j < jend ? tgt.pts[j] : make_float4(inf, inf, inf, 0.f);
      const float4 b = j + 1 < jend ? tgt.pts[j + 1] : make_float4(inf, inf, inf, 0.f);
      txy[e] = make_float4(a.x, b.x, a.y, b.y);
      tz[e] = make_float2(a.z, b.z);
    }
    __syncthreads();
    const int nch = min(NN_TILE / kChunk, c1 - tc0);
    for (int ch = 0; ch < nch; ch++) {
      float m[S];
#pragma unroll
      for (int s = 0; s < S; s++) m[s] = inf;
#pragma unroll
      for (int jj = 0; jj < kChunk / 2; jj++) {
        const float4 A = txy[ch * (kChunk / 2) + jj];
        const float2 Z = tz[ch * (kChunk / 2) + jj];
#pragma unroll
        for (int sp = 0; sp < S / 2; sp++) {
          float2v dx = A.x - px[sp], dy = A.z - py[sp], dz = Z.x - pz[sp];
          float2v d0 = dx * dx;
          d0 = d0 + dy * dy;
          d0 = d0 + dz * dz;
          dx = A.y - px[sp], dy = A.w - py[sp], dz = Z.y - pz[sp];
          float2v d1 = dx * dx;
          d1 = d1 + dy * dy;
          d1 = d1 + dz * dz;
          m[2 * sp] = fminf(fminf(m[2 * sp], d0.x), d1.x);
          m[2 * sp + 1] = fminf(fminf(m[2 * sp + 1], d0.y), d1.y);
        }
      }
#pragma unroll
      for (int s = 0; s < S; s++) {
        if (m[s] < best[s]) best[s] = m[s], bestc[s] = (unsigned)(tc0 + ch);
        else if (m[s] == best[s] && m[s] < inf) bestc[s] |= kTieBit;
      }
    }
  }
  unsigned long long* out = w.nnpart + ((size_t)pair * w.T + blockIdx.y) * w.nstride;
#pragma unroll
  for (int s = 0; s < S; s++) {
    const int i = base + s * NN_BLK + tid;
    if (i < N) out[i] = ((unsigned long long)__float_as_uint(best[s]) << 32) | bestc[s];
  }
}

// ----------------------------------------------------------------------------------------------
// fp32 lower bounds used for pruning.  They are evaluated with the same operation order as the
// distance itself; rounding is monotone, so for EVERY target t inside the box
//   lb(p, box) <= sqdist1(t, p)   holds for the computed fp32 values, not just the real numbers.
// A chunk is skipped only when lb > best (strict), so ties survive and the result is unchanged.
__device__ __forceinline__ float lb_point_box(const Box& b, float px, float py, float pz) {
  const float gx = fmaxf(fmaxf(b.lx - px, px - b.hx), 0.f);
  const float gy = fmaxf(fmaxf(b.ly - py, py - b.hy), 0.f);
  const float gz = fmaxf(fmaxf(b.lz - pz, pz - b.hz), 0.f);
  float r = gx * gx;
  r = r + gy * gy;
  r = r + gz * gz;
  return r;
}
__device__ __forceinline__ float lb_box_box(const Box& a, const Box& b) {
  const float gx = fmaxf(fmaxf(b.lx - a.hx, a.lx - b.hx), 0.f);
  const float gy = fmaxf(fmaxf(b.ly - a.hy, a.ly - b.hy), 0.f);
  const float gz = fmaxf(fmaxf(b.lz - a.hz, a.lz - b.hz), 0.f);
  float r = gx * gx;
  r = r + gy * gy;
  r = r + gz * gz;
  return r;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

__device__ __forceinline__ float readlane_f63(float v) { return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v), 63)); }
__device__ __forceinline__ float readlane_f(float v, int l) { return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), l)); }
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned lo = __shfl_xor((unsigned)v, off, 64), hi = __shfl_xor((unsigned)(v >> 32), off, 64);
    const unsigned long long o = ((unsigned long long)hi << 32) | lo;
    v = o < v ? o : v;
  }
  return v;
}
// wave-wide min / max through DPP (row_shr 1,2,4,8, row_bcast 15/31; lanes without a source keep their own value),
// result read from lane 63 into an SGPR: uniform, no LDS traffic
template <bool MAX>
__device__ __forceinline__ float wave_minmax_uniform(float v) {
  auto step = [](float x, float o) { return MAX ? fmaxf(x, o) : fminf(x, o); };
#define APD_DPP_F(ctrl, rmask) __uint_as_float((unsigned)__builtin_amdgcn_update_dpp((int)__float_as_uint(v), (int)__float_as_uint(v), ctrl, rmask, 0xf, false))
  v = step(v, APD_DPP_F(0x111, 0xf));
  v = step(v, APD_DPP_F(0x112, 0xf));
  v = step(v, APD_DPP_F(0x114, 0xf));
  v = step(v, APD_DPP_F(0x118, 0xf));
  v = step(v, APD_DPP_F(0x142, 0xa));
  v = step(v, APD_DPP_F(0x143, 0xc));
#undef APD_DPP_F
  return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v), 63));
}

// The same reductions with the DPP modifier on the min / max itself (one instruction per step instead of copy + v_mov_dpp +
// canonicalise + op) -- the compiler does not fold them -- for the bounding box of a wave's points: three minima and three
// maxima, interleaved so that no step reads a register written less than two instructions earlier (the DPP read-after-
// VALU-write hazard needs two wait states; the leading s_nop covers the producers of the inputs).
__device__ __forceinline__ void wave_box_uniform(float& lx, float& ly, float& lz, float& hx, float& hy, float& hz) {
#define APD_BOX_STEP(ctrl)                                                \
  "v_min_f32_dpp %0, %0, %0 " ctrl "\n v_min_f32_dpp %1, %1, %1 " ctrl "\n" \
  "v_min_f32_dpp %2, %2, %2 " ctrl "\n v_max_f32_dpp %3, %3, %3 " ctrl "\n" \
  "v_max_f32_dpp %4, %4, %4 " ctrl "\n v_max_f32_dpp %5, %5, %5 " ctrl "\n"
  asm volatile("s_nop 1\n" APD_BOX_STEP("row_shr:1 row_mask:0xf bank_mask:0xf") APD_BOX_STEP("row_shr:2 row_mask:0xf bank_mask:0xf")
                   APD_BOX_STEP("row_shr:4 row_mask:0xf bank_mask:0xf") APD_BOX_STEP("row_shr:8 row_mask:0xf bank_mask:0xf")
                       APD_BOX_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf") APD_BOX_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
               : "+v"(lx), "+v"(ly), "+v"(lz), "+v"(hx), "+v"(hy), "+v"(hz));
#undef APD_BOX_STEP
  lx = readlane_f63(lx), ly = readlane_f63(ly), lz = readlane_f63(lz), hx = readlane_f63(hx), hy = readlane_f63(hy), hz = readlane_f63(hz);
}
// wave-wide maximum of one value (lane 63 -> SGPR); the wait states sit between the dependent steps
__device__ __forceinline__ float wave_max_uniform(float v) {
  asm volatile(
      "s_nop 1\n v_max_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n"
      "s_nop 1\n v_max_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n"
      "s_nop 1\n v_max_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n"
      "s_nop 1\n v_max_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n"
      "s_nop 1\n v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
      "s_nop 1\n v_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
      : "+v"(v));
  return readlane_f63(v);
}

// one wave stages one 128-target group into its LDS tile: lane l loads targets 2l, 2l+1, and lanes
// 0..47 the group's 8 chunk boxes (48 consecutive floats)
template <bool WITH_PERM = false>
__device__ __forceinline__ void stage_group(float4* txy, float2* tz, float* cbl, const float4* pts, const Box* cbox, int g, int M, int nchunks,
                                            int lane, int* perml = nullptr, const int* perm = nullptr) {
  const float inf = __builtin_inff();
  const int j = g * kGroupPts + 2 * lane;
  const float4 a = j < M ? pts[j] : make_float4(inf, inf, inf, 0.f);
  const float4 b = j + 1 < M ? pts[j + 1] : make_float4(inf, inf, inf, 0.f);
  txy[lane] = make_float4(a.x, b.x, a.y, b.y);
  tz[lane] = make_float2(a.z, b.z);
  if (lane < 6 * kGroupChunks) {
    const int c = g * kGroupChunks + lane / 6;
    cbl[lane] = c < nchunks ? ((const float*)cbox)[(size_t)g * kGroupChunks * 6 + lane] : inf;
  }
  if (WITH_PERM) {
    perml[2 * lane] = j < M ? perm[j] : 0x7fffffff;
    perml[2 * lane + 1] = j + 1 < M ? perm[j + 1] : 0x7fffffff;
  }
}
__device__ __forceinline__ Box lds_box(const float* cbl, int ch) {
  return Box{cbl[6 * ch], cbl[6 * ch + 1], cbl[6 * ch + 2], cbl[6 * ch + 3], cbl[6 * ch + 4], cbl[6 * ch + 5]};
}

// ----------------------------------------------------------------------------------------------
// k_nn_pruned: the same exact 1-NN as k_nn_partial, on Z-curve-sorted clouds.  One wave (= one block)
// owns 64*S consecutive sorted source points, i.e. a spatially compact set.  For every 128-point
// target group each lane evaluates the lower bound of ITS points to the group box; the wave stages
// and scans a group only if some lane can still improve, and inside a group the 16-point chunks are
// tested the same way before their distances are evaluated.  The group that contains most of the
// wave's points is scanned first so that `best` is tight before the sweep.
constexpr int GB_BATCH = 64;   // group boxes staged per LDS batch (1.5 KB): small on purpose, three pair groups of search blocks share the CUs

// The search of k_nn_pruned for the 64*S points [base, base + 64*S) of one pair.  LDS: txy[64] float4, tz[64] float2, cbl[48], gbl[6*GB_BATCH] floats.
// Returns the transformed points, the minimum squared distance and (chunk | kTieBit) per point.
//
// W waves per block share the SAME 64*S points: every wave computes the bounds and the candidate groups (cheap), then
// scans only every W-th candidate group out of its own LDS tile, and the partial minima are merged through LDS at the end.
// The per-block latency of the scan phase drops by about W at the price of W times the waves; with the GPU mostly idle
// during an optimiser tick that is the better trade.  txy/tz/cbl: this wave's tile; gbl: shared by the block.
__device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }  // same-wave LDS write -> read

// Warm start of ONE point of a search, shared by k_nn_pruned and k_nn_compact.
//
// The neighbour found by the previous iteration is a real target point, so its distance at the new pose is a valid upper
// bound of the minimum and makes the pruning effective from the first group on.  It only changes which chunks are visited,
// never the result.
//
// Keeping a neighbour without searching.  A full search leaves behind, per point, the transformed point p0 it ran for and a
// bound s: every target other than the neighbour q it found had a computed squared distance >= s at p0 (the second smallest
// distance among the scanned targets, and everything it never scanned lies beyond its pruning radius, which was inflated
// by the skin: r^2 -> r^2 * skin_mul + skin_add).  At a later pose the point sits at p1.  With eps bounding the relative
// rounding error of the fp32 distance (5 roundings of non-negative terms: < 3.1e-7) and |p1 - p0| its displacement,
//     D(t, p1) >= (sqrt(s) (1 - eps) - |p1 - p0|)^2 (1 - eps)      for every target t != q,
// so when that exceeds D(q, p1) -- evaluated directly -- q is still the unique nearest neighbour, bit for bit what the
// full search would return, and the point takes no part in the search.  (A point without any target inside the cap keeps
// that status the same way, with the cap in place of D(q, p1).)  The test evaluates the inequality with 4e-6 of slack on
// every factor.
struct NNStart {
  float px, py, pz, best;
  unsigned bestc;
  bool kept, hinted;
};
__device__ __forceinline__ NNStart nn_warm_start(const CloudDesc& src, int M, const float* Tf, const Work& w, int pair, int ii, bool cold, bool skin_on) {
  const float inf = __builtin_inff();
  NNStart o;
  const float4 p = G(src.pts)[ii];
  o.px = xf_row(Tf + 0, p.x, p.y, p.z, w.xf_linear), o.py = xf_row(Tf + 4, p.x, p.y, p.z, w.xf_linear), o.pz = xf_row(Tf + 8, p.x, p.y, p.z, w.xf_linear);
  o.best = w.cap, o.bestc = kNoChunk, o.kept = false;
  const float4 t = w.nnpt[(size_t)pair * w.nstride + ii];
  const int hint = cold ? -1 : __float_as_int(t.w);
  float dq = w.cap;  // distance to the previous neighbour at this pose (the cap when there was none inside it)
  o.hinted = hint >= 0 && hint < M;
  if (o.hinted) {
    dq = sqdist1(t.x, t.y, t.z, o.px, o.py, o.pz);
    if (dq < o.best) o.best = dq, o.bestc = (unsigned)(hint / kChunk);
  }
  if (skin_on) {
    const float4 a = w.nnaux[(size_t)pair * w.nstride + ii];
    // (v_sqrt_f32 itself, 1 ulp: the library's correctly rounded sqrtf is that plus a dozen instructions of fix-up per call, and every
    // factor of the test carries 4e-6 of slack; a denormal argument comes back as 0 -- a displacement below 1e-19 m, or no bound)
    const float mv = __builtin_amdgcn_sqrtf(sqdist1(a.x, a.y, a.z, o.px, o.py, o.pz));
    const float lo = __builtin_amdgcn_sqrtf(a.w) * (1.f - 4e-6f) - mv * (1.f + 4e-6f);
    // The record describes the targets OTHER than the neighbour on record, so it is only usable while that neighbour is
    // the one k_linearize last wrote: a neighbour that has left the cap (k_linearize then records "none") takes the full
    // search.  hint == -1 with a.w > 0: the last full search found no target inside the cap, and s covers every target.
    const bool ok = a.w > 0.f && lo > 0.f && lo * lo * (1.f - 4e-6f) > fminf(dq, w.cap) * (1.f + 4e-6f) && (o.hinted ? dq < w.cap : hint == -1) &&
                    w.cap < inf;
    if (ok) {
      o.kept = true;
      if (o.hinted) o.best = dq, o.bestc = (unsigned)(hint / kChunk) | kKeptBit;
      else o.best = w.cap, o.bestc = kNoChunk;
    }
  }
  return o;
}

// What every search ends with, for the wave that holds the final (merged) minima: the exact index of each point's neighbour, ties
// settled, and the record the next search keeps its neighbour by.  Lanes without a point come in with pidx = -1 (or kept).
template <int S>
__device__ __forceinline__ void nn_finish(const CloudDesc& tgt, const Work& w, int pair, int lane, bool skin_on, float k_mul, float k_add, float (&px)[S],
                                          float (&py)[S], float (&pz)[S], float (&best)[S], unsigned (&bestc)[S], int (&pidx)[S], bool (&kept)[S],
                                          float (&g1)[S], float (&g2)[S]) {
  const int M = tgt.n;
  // The exact index: among the targets of the winning chunk at distance `best`, the one with the lowest ORIGINAL index (ties
  // resolve like a linear scan in the caller's point order).  Resolved here, by the waves that searched, and handed to k_linearize through nnpt + kKeptBit: the
  // re-scan used to run in every wave of k_linearize that held a single point without the bit, i.e. in nearly all of them
  // even when nine points in ten had kept their neighbour.  (Equal minima in several chunks -- kTieBit -- stay with
  // k_linearize's scan of the whole target.)
  // Equal minima in several chunks (kTieBit: two different targets at exactly the same fp32 distance, or duplicates -- about one
  // point search in four million on radar scans).  The wave settles it here, together: every lane looks at a 64th of the target
  // for the lowest original index at that distance, about 10 us.  (Until round 4 the bit travelled to k_linearize, where the ONE
  // lane that held the point walked the whole target: 1.1 ms for 8192 targets, during which its launch -- and with it the tick
  // of every pair of a pooled batch -- stood still: one tie per batch of 32 loop-closure pairs was 1.03 -> 1.9 ms per batch.)
  {
#pragma unroll
    for (int s = 0; s < S; s++) {
      unsigned long long tmask = __ballot(pidx[s] >= 0 && !kept[s] && bestc[s] != kNoChunk && (bestc[s] & kTieBit) != 0);
      while (tmask) {
        const int l = __builtin_ctzll(tmask);
        tmask &= tmask - 1;
        const float qx = readlane_f(px[s], l), qy = readlane_f(py[s], l), qz = readlane_f(pz[s], l), qb = readlane_f(best[s], l);
        unsigned long long key = ~0ull;  // (original index << 32 | sorted index) of the best candidate this lane has seen
#pragma unroll 4
        for (int g = lane; g < M; g += 64) {
          const float4 t = G(tgt.pts)[g];
          const unsigned long long k_ = ((unsigned long long)__float_as_uint(t.w) << 32) | (unsigned)g;  // (.w: the original index, >= 0)
          if (sqdist1(t.x, t.y, t.z, qx, qy, qz) == qb && k_ < key) key = k_;
        }
        key = wave_min_u64(key);
        if (lane == l && key != ~0ull) {
          const int j = (int)(unsigned)key;
          const float4 tq = G(tgt.pts)[j];
          w.nnpt[(size_t)pair * w.nstride + pidx[s]] = make_float4(tq.x, tq.y, tq.z, __int_as_float(j));
          bestc[s] = (unsigned)(j / kChunk) | kKeptBit;
        }
      }
    }
#pragma unroll
    for (int s = 0; s < S; s++) {
      const bool resolve = pidx[s] >= 0 && !kept[s] && bestc[s] != kNoChunk && !(bestc[s] & (kTieBit | kKeptBit));
      if (resolve) {
        const int c0 = (int)(bestc[s] & kChunkMask) * kChunk;
        int j = -1, jorig = 0x7fffffff;
        float4 tq = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
        for (int h = 0; h < kChunk; h += 8) {  // eight loads in flight at a time: 32 registers, not 64
          float4 t[8];
#pragma unroll
          for (int jj = 0; jj < 8; jj++) t[jj] = G(tgt.pts)[min(c0 + h + jj, M - 1)];
#pragma unroll
          for (int jj = 0; jj < 8; jj++) {
            const float d = sqdist1(t[jj].x, t[jj].y, t[jj].z, px[s], py[s], pz[s]);
            const int po = __float_as_int(t[jj].w);  // (the sorted points carry their original index in .w)
            if (c0 + h + jj < M && d == best[s] && po < jorig) jorig = po, j = c0 + h + jj, tq = t[jj];
          }
        }
        if (j >= 0) {
          w.nnpt[(size_t)pair * w.nstride + pidx[s]] = make_float4(tq.x, tq.y, tq.z, __int_as_float(j));
          bestc[s] |= kKeptBit;
        }
      }
    }
  }
  // what this search leaves behind for the next one (points that kept their neighbour keep their old record)
  if (w.nnaux) {
#pragma unroll
    for (int s = 0; s < S; s++) {
      const int i = pidx[s];
      if (i < 0 || kept[s]) continue;
      float sb = 0.f;
      if (skin_on && !(bestc[s] & kTieBit) && bestc[s] != kNoChunk) {
        // everything never scanned lies beyond the final pruning radius (radii only shrink); among the scanned targets the
        // neighbour itself is g1 (if it is not, e.g. it was only ever seen as the hint, nothing is claimed)
        if (g1[s] == best[s]) sb = fminf(g2[s], fmaf(best[s], k_mul, k_add));
      } else if (skin_on && bestc[s] == kNoChunk) {
        sb = fminf(g1[s], fmaf(best[s], k_mul, k_add));  // no target inside the cap: the nearest one seen, or the radius
      }
      w.nnaux[(size_t)pair * w.nstride + i] = make_float4(px[s], py[s], pz[s], sb);
    }
  }
}

// GIVEN = false: the classic form -- the W waves of the block share the points [base, base + 64 S) and warm-start them here.
// GIVEN = true (k_nn_compact): the caller hands the points over -- px/py/pz/best/bestc/kept/hinted_in are inputs, pidx[s] is
// the index of the point a lane works for (-1: none; such a lane must come in with kept = true).  With W == 1 the wave owns
// ALL of its LDS (gbl included) and nothing here synchronises with the other waves of the block; with W > 1 the W waves of the
// block came in with the SAME points and split the groups between them like the classic form (block barriers, shared gbl).
template <int S, int W, bool GIVEN = false>
__device__ __forceinline__ void nn_search(const CloudDesc& src, const CloudDesc& tgt, const Rigid& T0, const Work& w, int pair, int base, int lane,
                                          int wid, bool cold, float4* txy, float2* tz, float* cbl, float* gbl, unsigned long long* mrg /* [W][64*S] */,
                                          float2* mrg2 /* [W][64*S] */, float (&px)[S], float (&py)[S], float (&pz)[S], float (&best)[S],
                                          unsigned (&bestc)[S], int (&pidx)[S], bool (&kept)[S], bool hinted_in = true) {
  constexpr bool OWN = GIVEN && W == 1;
  const int N = src.n, M = tgt.n;
  const int tid = wid * 64 + lane;
  auto block_sync = [&]() {
    if (OWN) wave_lds_fence();
    else __syncthreads();
  };
  const bool tstat = w.stats && wid == 0 && (blockIdx.x & 15) == 0;  // phase timing: sampled waves only
  long long tcy[4] = {0, 0, 0, 0}, tm = tstat ? clock64() : 0;
  float Tf[12];
  load_Tf(T0, Tf);
  const int ngroups = (M + kGroupPts - 1) / kGroupPts;
  const int nchunks = (M + kChunk - 1) / kChunk;
  // boxes of the first 64 target groups: requested now, parked in registers, written to LDS after the warm start
  constexpr int NPRE = (6 * 64 + 64 * W - 1) / (64 * W);
  float gpre[NPRE];
#pragma unroll
  for (int u = 0; u < NPRE; u++) {
    const int e = u * 64 * W + tid;
    gpre[u] = G((const float*)tgt.gbox)[min(e, 6 * ngroups - 1)];
  }
  const float inf = __builtin_inff();

  bool all_hinted = hinted_in;
  float bestR[S], sk_mul[S], sk_add[S];   // pruning radius of the point and how it follows `best` (kept points: always -1)
  float g1[S], g2[S];                     // smallest and second smallest computed distance among the SCANNED targets
  const bool skin_on = !cold && w.nnaux != nullptr && w.skin_mul > 0.f;
  const float k_mul = skin_on ? w.skin_mul : 1.f, k_add = skin_on ? w.skin_add : 0.f;
#pragma unroll
  for (int s = 0; s < S; s++) {
    if (!GIVEN) {
      const int i = base + s * 64 + lane;
      pidx[s] = i < N ? i : -1;
      const NNStart st = nn_warm_start(src, M, Tf, w, pair, i < N ? i : N - 1, cold, skin_on);
      px[s] = st.px, py[s] = st.py, pz[s] = st.pz, best[s] = st.best, bestc[s] = st.bestc, kept[s] = st.kept;
      all_hinted &= st.hinted;
    }
    g1[s] = inf, g2[s] = inf;
    sk_mul[s] = kept[s] ? 0.f : k_mul, sk_add[s] = kept[s] ? -1.f : k_add;
    bestR[s] = kept[s] ? -1.f : fmaf(best[s], sk_mul[s], sk_add[s]);
  }
  if (!GIVEN) {
    bool allk = true;
#pragma unroll
    for (int s = 0; s < S; s++) allk &= kept[s];
    if (__all(allk)) {  // (the W waves of a block hold the same points: they all leave here, or none does)
      if (w.stats && tid == 0) atomicAdd(w.stats + 3, 1ull), atomicAdd(w.stats + 6, 64ull * S);
      return;
    }
    if (w.stats && wid == 0) {
      unsigned long long nk = 0;
#pragma unroll
      for (int s = 0; s < S; s++) nk += __popcll(__ballot(kept[s]));
      if (lane == 0) atomicAdd(w.stats + 6, nk);
    }
  }
  // seeded cold start only when (almost) nobody has a hint; a few lanes without one (no target inside the cap last
  // time) start from the cap and need no seed
  all_hinted = w.cap < inf ? __popcll(__ballot(all_hinted)) >= 32 : __all(all_hinted);
  // bounding box of this wave's (transformed) points
  Box wbox;
  {
    float lx = inf, ly = inf, lz = inf, hx = -inf, hy = -inf, hz = -inf;  // (points that kept their neighbour need nothing)
#pragma unroll
    for (int s = 0; s < S; s++) {
      if (kept[s]) continue;
      lx = fminf(lx, px[s]), ly = fminf(ly, py[s]), lz = fminf(lz, pz[s]);
      hx = fmaxf(hx, px[s]), hy = fmaxf(hy, py[s]), hz = fmaxf(hz, pz[s]);
    }
    wave_box_uniform(lx, ly, lz, hx, hy, hz);
    wbox = Box{lx, ly, lz, hx, hy, hz};
  }
  if (tstat) { const long long t = clock64(); tcy[0] += t - tm, tm = t; }

  unsigned n_groups = 0, n_ctest = 0, n_cscan = 0, n_batches = 0;
  // registers of the group in flight: loaded from L2 while the previous group is scanned out of LDS
  float4 pa = make_float4(inf, inf, inf, 0.f), pb = pa;
  float pc = inf;
  // The three loads of a group are UNCONDITIONAL (clamped addresses; what lies beyond the cloud is replaced when the group
  // is committed): written as `j < M ? load : inf` each one sat in a branch of its own with an `s_waitcnt vmcnt(0)` right
  // behind it -- three dependent trips to memory per group, none of them under the scan of the previous group.
  auto fetch_group = [&](int g) {
    const int j = g * kGroupPts + 2 * lane;
    pa = G(tgt.pts)[min(j, M - 1)];
    pb = G(tgt.pts)[min(j + 1, M - 1)];
    pc = G((const float*)tgt.cbox)[min(g * kGroupChunks * 6 + min(lane, 6 * kGroupChunks - 1), nchunks * 6 - 1)];
  };
  auto commit_group = [&](int g) {  // registers -> the wave's LDS tile ({x0,x1,y0,y1} + {z0,z1} pairs, 8 chunk boxes)
    const int j = g * kGroupPts + 2 * lane;
    const bool va = j < M, vb = j + 1 < M, vc = g * kGroupChunks + lane / 6 < nchunks;
    wave_lds_fence();          // the tile belongs to this wave alone: its previous scan is done with it
    txy[lane] = make_float4(va ? pa.x : inf, vb ? pb.x : inf, va ? pa.y : inf, vb ? pb.y : inf);
    tz[lane] = make_float2(va ? pa.z : inf, vb ? pb.z : inf);
    if (lane < 6 * kGroupChunks) cbl[lane] = vc ? pc : inf;
    wave_lds_fence();
  };
  auto lane_needs = [&](const Box& bx) {
    bool need = false;
#pragma unroll
    for (int s = 0; s < S; s++) need |= lb_point_box(bx, px[s], py[s], pz[s]) <= bestR[s];
    return need;
  };
  auto scan_tile = [&](int g) {
    n_groups++;
    const int cend = min(kGroupChunks, nchunks - g * kGroupChunks);
    // all 8 chunk tests first (independent LDS reads, no branches), then only the scans that are needed
    unsigned cmask = 0;
#pragma unroll
    for (int ch = 0; ch < kGroupChunks; ch++)
      if (__any(lane_needs(lds_box(cbl, ch)))) cmask |= 1u << ch;
    cmask &= (1u << cend) - 1u;
    n_ctest += cend;
    while (cmask) {
      const int ch = __builtin_ctz(cmask);
      cmask &= cmask - 1;
      n_cscan++;
      float m[S], m2[S];  // smallest / second smallest distance of this chunk (v_min3 + v_med3 + v_min per two targets)
#pragma unroll
      for (int s = 0; s < S; s++) m[s] = inf, m2[s] = inf;
#pragma unroll
      for (int jj = 0; jj < kChunk / 2; jj++) {
        // (throughput kernel: the compiler may not pull all sixteen targets' LDS reads to the top of the scan -- four at a time keep
        // the kernel at 92 registers without scratch: 0.7385 -> 0.7305 ms per step)
        if (GIVEN && jj % 2 == 0 && jj) __builtin_amdgcn_sched_barrier(0);
        const float4 A = txy[ch * (kChunk / 2) + jj];
        const float2 Z = tz[ch * (kChunk / 2) + jj];
        // both targets at once, as 2-vectors (sqdist1's operations in sqdist1's order, per component): written as two scalar
        // distances the compiler packed some of the operations and paid for it with register moves -- 17.5 vector
        // instructions per pair of targets against 11
        // (in the throughput kernel only -- GIVEN: packed operations take two passes through the vector unit, and the chain of a
        // lone search wave got 1 us longer per odometry frame with them)
        const v2f X = {A.x, A.y}, Y = {A.z, A.w}, ZZ = {Z.x, Z.y};
#pragma unroll
        for (int s = 0; s < S; s++) {
          float d0, d1;
          if constexpr (GIVEN) {
            const v2f dx = X - px[s], dy = Y - py[s], dz = ZZ - pz[s];
            v2f r = dx * dx;
            r = r + dy * dy;
            r = r + dz * dz;
            d0 = r.x, d1 = r.y;
          } else {
            d0 = sqdist1(A.x, A.z, Z.x, px[s], py[s], pz[s]);
            d1 = sqdist1(A.y, A.w, Z.y, px[s], py[s], pz[s]);
          }
          m2[s] = fminf(m2[s], __builtin_amdgcn_fmed3f(m[s], d0, d1));
          m[s] = fminf(fminf(m[s], d0), d1);
        }
      }
      const unsigned c = (unsigned)(g * kGroupChunks + ch);
#pragma unroll
      for (int s = 0; s < S; s++) {
        g2[s] = fminf(fmaxf(g1[s], m[s]), fminf(g2[s], m2[s]));
        g1[s] = fminf(g1[s], m[s]);
        if (m[s] < best[s]) best[s] = m[s], bestc[s] = c, bestR[s] = fmaf(m[s], sk_mul[s], sk_add[s]);
        else if (m[s] == best[s] && m[s] < inf && (bestc[s] & kChunkMask) != c) bestc[s] |= kTieBit;
      }
    }
  };

  // Large targets carry one more level (apd_sort.hpp: a box per 64 groups, behind the group boxes): lane s tests super box
  // s against the box of this wave's points with the radius known NOW -- before any scan, so the W waves of the block, which
  // hold the same points and hints, agree on the batches they visit (the barriers below need that); later radii are smaller.
  static_assert(GB_BATCH == kSuperGroups, "one batch of group boxes per super box");
  const bool use_super = M > SORT_LDS_MAX_N;
  const int nsuper = (ngroups + GB_BATCH - 1) / GB_BATCH;
  float rad0 = bestR[0];
#pragma unroll
  for (int s = 1; s < S; s++) rad0 = fmaxf(rad0, bestR[s]);
  rad0 = wave_max_uniform(rad0);
  for (int sb0 = 0; sb0 < nsuper; sb0 += 64) {
  unsigned long long smask = nsuper - sb0 >= 64 ? ~0ull : (1ull << (nsuper - sb0)) - 1ull;
  if (use_super && rad0 < inf) {
    const Box sbx = G(tgt.gbox)[ngroups + min(sb0 + lane, nsuper - 1)];  // lane b: super box sb0 + b
    smask &= __ballot(lb_box_box(wbox, sbx) <= rad0);
    // A SCATTERED wave (round 6): 64 consecutive points of the curve that lie far apart -- a jump of the curve, outliers at the edge of a
    // sparse scan -- have a box that nearly every super box of a dense map touches, and one large radius among them inflates all of it:
    // in the 100k x 500k registration ONE such block tested 2 100 group boxes where its points need nine groups, three times the
    // duration of the median block and as long as the whole launch.  When more than eight super boxes pass the wave's box, every point
    // is looked at on its own (lane = super box, the points broadcast in turn, each with ITS radius): what no point needs is left out.
    // Uniform over the block's waves like the test above: points and start radii are the same in all of them.
    if (!GIVEN && __popcll(smask) > 8) {  // (not in the throughput kernel of scan-sized clouds: its registers are spoken for)
      unsigned long long fine = 0;
#pragma unroll 1
      for (int l = 0; l < 64; l++) {
#pragma unroll
        for (int s = 0; s < S; s++) fine |= __ballot(lb_point_box(sbx, readlane_f(px[s], l), readlane_f(py[s], l), readlane_f(pz[s], l)) <= readlane_f(bestR[s], l));
      }
      smask &= fine;
    }
  }
  while (smask) {
    const int sbi = __builtin_ctzll(smask);
    const int gb0 = (sb0 + sbi) * GB_BATCH;
    smask &= smask - 1;
    const int nbb = min(GB_BATCH, ngroups - gb0);
    n_batches++;
    block_sync();  // uniform over the block's waves (the batch loop is): nobody still reads the previous batch
    if (gb0 == 0) {
#pragma unroll
      for (int u = 0; u < NPRE; u++)
        if (u * 64 * W + tid < 6 * 64) gbl[u * 64 * W + tid] = gpre[u];
    }
    for (int e = (gb0 == 0 ? 6 * 64 : 0) + tid; e < 6 * nbb; e += 64 * W) gbl[e] = G((const float*)tgt.gbox)[(size_t)gb0 * 6 + e];
    block_sync();
    for (int sb = 0; sb < nbb; sb += 64) {  // 64 groups at a time: their need bits fit one mask
      const int nb = min(64, nbb - sb);
      const float* boxes = gbl + 6 * sb;
      const int g0 = gb0 + sb;
      // cold start only: scan first the group whose box contains the most points of this wave
      int seed = -1;
      if (g0 == 0 && !all_hinted) {
        int seed_cnt = 0;
        for (int g = 0; g < nb; g++) {
          const Box gb = lds_box(boxes, g);
          int inside = 0;
#pragma unroll
          for (int s = 0; s < S; s++) inside += lb_point_box(gb, px[s], py[s], pz[s]) == 0.f ? 1 : 0;
          const int cntg = __popcll(__ballot(inside > 0));
          if (cntg > seed_cnt) seed_cnt = cntg, seed = g;
        }
        if (seed >= 0) {
          fetch_group(g0 + seed);
          commit_group(g0 + seed);
          scan_tile(g0 + seed);
        }
      }
      // candidate groups with the bounds known now; a candidate is re-tested against the then-current
      // `best` just before its scan.  First one test per LANE: group `lane` against the box of all points of
      // this wave with the wave's largest radius (a lower bound of every per-point bound, so it only removes
      // groups no lane needs); the per-point tests then run on the few survivors, 4 per trip.
      float rad = bestR[0];
#pragma unroll
      for (int s = 1; s < S; s++) rad = fmaxf(rad, bestR[s]);
      rad = wave_max_uniform(rad);
      unsigned long long pre = nb < 64 ? (1ull << nb) - 1ull : ~0ull;
      if (rad < inf) pre &= __ballot(lb_box_box(wbox, lds_box(boxes, min(lane, nb - 1))) <= rad);
      if (!GIVEN && use_super && S == 1 && __popcll(pre) > 16) {
        // ... and the same one level down: most of a batch's groups pass the wave's box although only a few of the wave's points reach into the
        // batch's super box at all -- those points are tested against the 64 group boxes (lane = group) instead of every group against all points
        // (the batch's super box through a uniform load: kept in a register per lane from the test above it cost the kernel a wave per SIMD)
        const Box sb_u = G(tgt.gbox)[ngroups + sb0 + sbi];
        unsigned long long pm = __ballot(lb_point_box(sb_u, px[0], py[0], pz[0]) <= bestR[0]);
        if (__popcll(pm) <= 16) {
          const Box mine = lds_box(boxes, min(lane, nb - 1));
          unsigned long long need = 0;
#pragma unroll 1
          while (pm) {
            const int l = __builtin_ctzll(pm);
            pm &= pm - 1;
            need |= __ballot(lb_point_box(mine, readlane_f(px[0], l), readlane_f(py[0], l), readlane_f(pz[0], l)) <= readlane_f(bestR[0], l));
          }
          pre &= need;
        }
      }
      // This wave scans the groups with index = wid (mod W), so those are the only ones it has to test.  The split must not
      // depend on the candidate set: after the first 64 groups the waves hold different partial minima, hence different
      // candidate sets (a group missing from one wave's set cannot beat that wave's minimum, so it cannot beat the merged
      // minimum either).
      if (W > 1) pre &= (W == 2 ? 0x5555555555555555ull : W == 4 ? 0x1111111111111111ull : 0x0101010101010101ull) << wid;
      unsigned long long cand = 0;
      while (pre) {
        int gq[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          gq[u] = pre ? __builtin_ctzll(pre) : -1;
          if (pre) pre &= pre - 1;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int gg = gq[u] >= 0 ? gq[u] : gq[0];
          if (__any(lane_needs(lds_box(boxes, gg)))) cand |= 1ull << gg;
        }
      }
      if (seed >= 0) cand &= ~(1ull << seed);
      if (tstat) { const long long t = clock64(); tcy[1] += t - tm, tm = t; }
      // software pipeline: group k+1 travels L2 -> registers while group k is scanned out of LDS
      int cur = cand ? __builtin_ctzll(cand) : -1;
      if (cur >= 0) {
        cand &= cand - 1;
        fetch_group(g0 + cur);
      }
      while (cur >= 0) {
        commit_group(g0 + cur);
        const int nxt = cand ? __builtin_ctzll(cand) : -1;
        if (nxt >= 0) {
          cand &= cand - 1;
          fetch_group(g0 + nxt);
        }
        if (__any(lane_needs(lds_box(boxes, cur)))) scan_tile(g0 + cur);
        cur = nxt;
      }
    }
  }
  }
  if (W > 1) {  // merge the waves' partial minima: smallest distance; the same minimum in two different chunks is a tie
#pragma unroll
    for (int s = 0; s < S; s++) {
      mrg[(wid * S + s) * 64 + lane] = ((unsigned long long)__float_as_uint(best[s]) << 32) | bestc[s];
      mrg2[(wid * S + s) * 64 + lane] = make_float2(g1[s], g2[s]);
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < S; s++) {
      float bb = __uint_as_float((unsigned)(mrg[s * 64 + lane] >> 32));
      unsigned cc = (unsigned)mrg[s * 64 + lane];
      float a1 = mrg2[s * 64 + lane].x, a2 = mrg2[s * 64 + lane].y;
#pragma unroll
      for (int o = 1; o < W; o++) {
        const unsigned long long v = mrg[(o * S + s) * 64 + lane];
        const float bo = __uint_as_float((unsigned)(v >> 32));
        const unsigned co = (unsigned)v;
        if (bo < bb) {
          bb = bo, cc = co;
        } else if (bo == bb && bo < inf) {
          if ((co & kChunkMask) != (cc & kChunkMask)) cc |= kTieBit;
          cc |= co & kTieBit;
        }
        const float2 gg = mrg2[(o * S + s) * 64 + lane];  // the waves scanned disjoint sets of targets
        a2 = fminf(fmaxf(a1, gg.x), fminf(a2, gg.y));
        a1 = fminf(a1, gg.x);
      }
      best[s] = bb, bestc[s] = cc;
      g1[s] = a1, g2[s] = a2;
    }
  }
  if (wid == 0) nn_finish<S>(tgt, w, pair, lane, skin_on, k_mul, k_add, px, py, pz, best, bestc, pidx, kept, g1, g2);
  if (tstat) { const long long t = clock64(); tcy[2] += t - tm, tm = t; }
  // what this block's search cost (wave 0's share stands for the block: the W waves split the same groups): the order of the NEXT launches
  if (w.blk_cost && wid == 0 && lane == 0) w.blk_cost[base / (64 * S)] = 16u * n_cscan + 2u * n_ctest + 8u * n_groups + 4u * n_batches + 32u;
  if (w.stats && lane == 0) {  // (W > 1: every wave of the block scanned its own share of the groups, reports it and counts as a wave)
    atomicAdd(w.stats + 0, (unsigned long long)n_groups), atomicAdd(w.stats + 1, (unsigned long long)n_ctest);
    atomicAdd(w.stats + 2, (unsigned long long)n_cscan), atomicAdd(w.stats + 3, 1ull), atomicAdd(w.stats + 5, (unsigned long long)n_batches);
    if (tstat) atomicAdd(w.stats + 10, (unsigned long long)tcy[0]), atomicAdd(w.stats + 11, (unsigned long long)tcy[1]), atomicAdd(w.stats + 12, (unsigned long long)tcy[2]), atomicAdd(w.stats + 14, 1ull);
  }
}

template <int S, int W>
__device__ __forceinline__ void nn_pruned_block(const PairDesc* pairs, const PairState* st, const Work& w, unsigned bx, unsigned by) {
  __shared__ float4 txy[W][kGroupPts / 2];
  __shared__ float2 tz[W][kGroupPts / 2];
  __shared__ float cbl[W][6 * kGroupChunks];
  __shared__ float gbl[6 * GB_BATCH];
  __shared__ unsigned long long mrg[W > 1 ? W * 64 * S : 1];
  __shared__ float2 mrg2[W > 1 ? W * 64 * S : 1];
  const int pair = pair_of(w, by);
  if (pair < 0) return;  // (pooled batches: the launch was sized for more pairs than are still running)
  const int status = w.init ? (int)ST_NEED_LIN : st[pair].status;  // status, descriptors and pose: one round of scalar loads, not three
  const bool cold = w.init || st[pair].n_lin == 0;  // no linearize yet in this align: the hint array holds leftovers, ignore it
  const PairDesc pd = pairs[pair];
  const Rigid T0 = *(w.init ? w.init + pair : &st[pair].x0);
  if (status != ST_NEED_LIN) return;
  const CloudDesc src = pd.s, tgt = pd.t;
  const int N = src.n, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int base = (int)bx * (64 * S);
  if (base >= N) return;
  float px[S], py[S], pz[S], best[S];
  unsigned bestc[S];
  int pidx[S];
  bool kept[S];
  nn_search<S, W>(src, tgt, T0, w, pair, base, lane, wid, cold, txy[wid], tz[wid], cbl[wid], gbl, mrg, mrg2, px, py, pz, best, bestc, pidx, kept);
  if (wid != 0) return;
  unsigned long long* out = w.nnpart + (size_t)pair * w.T * w.nstride;  // T == 1 in this mode
#pragma unroll
  for (int s = 0; s < S; s++) {
    const int i = base + s * 64 + lane;
    if (i < N) out[i] = ((unsigned long long)__float_as_uint(best[s]) << 32) | bestc[s];
  }
}

#ifndef APD_NN_PRUNED_WPE
#define APD_NN_PRUNED_WPE 1  // (experiments: -DAPD_NN_PRUNED_WPE=6 asks for six waves per SIMD = 80 registers + 72 B scratch: C5 0.091 against 0.087 ms per iteration, profiles/r06_ab_c5_wpe6.txt)
#endif
template <int S, int W>
__global__ __launch_bounds__(64 * W) __attribute__((amdgpu_waves_per_eu(APD_NN_PRUNED_WPE, 8))) void k_nn_pruned(const CloudDesc* clouds, const PairDesc* pairs, const PairState* st, Work w) {
  unsigned bx, by;
  xcd_remap(bx, by);
  // A dense one-pair search has more blocks than the GPU holds at once (100k source points: 1563 blocks of four waves on 1280 places), blocks are
  // dealt in launch order and the curve's costliest stretch may come last: 100k x 500k -- the second round started at 30 us with blocks of
  // 35 - 44 us against a median of 26, the launch took 70 us where longest-first takes 49 (list scheduling of the measured durations,
  // tools/c5_blocks.py).  From the third tick of a registration on the blocks are launched by the cost they reported the tick before.
  if (w.blk_order) bx = w.blk_order[bx];
#ifdef APD_BLOCK_TIMELINE  // diagnostics build only (tools/c5_blocks.py; listed by apdgicp_build_flags): the start time carried through the kernel cost it 6 - 8 registers, a wave per SIMD
  const unsigned long long t0 = w.stats_blocks ? wall_clock64() : 0ull;
#endif
  nn_pruned_block<S, W>(pairs, st, w, bx, by);
#ifdef APD_BLOCK_TIMELINE
  if (w.stats_blocks) {
    const unsigned lid = blockIdx.y * gridDim.x + blockIdx.x;
    __syncthreads();
    if (threadIdx.x == 0 && (int)lid < w.stats_blocks) w.timeline[3 * lid] = t0, w.timeline[3 * lid + 1] = wall_clock64(), w.timeline[3 * lid + 2] = ((unsigned long long)by << 32) | bx;
  }
#endif
}

// k_nn_compact: the same search for optimiser ticks in which many points keep their neighbour (nn_warm_start).  A wave pays
// for a chunk scan whether one of its lanes needs it or all 64 do, so kept points scattered over the waves save little.
// Here a block of W waves warm-starts 64 W consecutive points, writes the results of the kept ones, packs the others into
// as few waves as they fill (still consecutive on the curve, i.e. compact) and only those waves search -- each on its own,
// with its own LDS, exactly like a one-wave block of k_nn_pruned.  Which wave searches for a point never changes a result.
template <int W>
__global__ __launch_bounds__(64 * W) __attribute__((amdgpu_waves_per_eu(5, 8))) void k_nn_compact(const CloudDesc* clouds, const PairDesc* pairs, const PairState* st, Work w) {
  __shared__ float4 txy[W][kGroupPts / 2];
  __shared__ float2 tz[W][kGroupPts / 2];
  __shared__ float cbl[W][6 * kGroupChunks];
  __shared__ float gbl[W][6 * GB_BATCH];
  __shared__ float4 cpt[64 * W];  // packed points: transformed position, best
  __shared__ int2 cci[64 * W];    // (chunk of the hint | kNoChunk, point index | bit 30: no hint)
  __shared__ int wcnt[W];
  unsigned bx, by;
  xcd_remap(bx, by);
  const int pair = pair_of(w, by);
  if (pair < 0) return;
  const int status = w.init ? (int)ST_NEED_LIN : st[pair].status;
  const bool cold = w.init || st[pair].n_lin == 0;
  const PairDesc pd = pairs[pair];
  const Rigid T0 = *(w.init ? w.init + pair : &st[pair].x0);
  if (status != ST_NEED_LIN) return;
  const CloudDesc src = pd.s, tgt = pd.t;
  const int N = src.n, M = tgt.n, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int base = (int)bx * (64 * W);
  if (base >= N) return;
  const int i = base + tid;
  const bool valid = i < N;
  const bool skin_on = !cold && w.nnaux != nullptr && w.skin_mul > 0.f;
  float Tf[12];
  load_Tf(T0, Tf);
  NNStart s0 = nn_warm_start(src, M, Tf, w, pair, valid ? i : N - 1, cold, skin_on);
#ifdef APD_ABL_SEARCH_KEEP_AFTER  // ABLATION builds only (wrong results by design): from linearize n on every hinted point counts as kept --
                                  // what a step would cost if the late-tick searches were free (docs/experiments.md, round 5)
  if (!w.init && st[pair].n_lin >= APD_ABL_SEARCH_KEEP_AFTER && s0.hinted && !s0.kept) s0.kept = true, s0.bestc |= kKeptBit;
#endif
  unsigned long long* out = w.nnpart + (size_t)pair * w.T * w.nstride;  // T == 1 in this mode
  if (valid && s0.kept) out[i] = ((unsigned long long)__float_as_uint(s0.best) << 32) | s0.bestc;
  const bool active = valid && !s0.kept;
  const unsigned long long am = __ballot(active);
  if (lane == 0) wcnt[wid] = __popcll(am);
  __syncthreads();
  int off = 0, total = 0;
#pragma unroll
  for (int o = 0; o < W; o++) {
    const int c = wcnt[o];
    off += o < wid ? c : 0;
    total += c;
  }
  if (active) {
    const int pos = off + __popcll(am & ((1ull << lane) - 1ull));
    cpt[pos] = make_float4(s0.px, s0.py, s0.pz, s0.best);
    cci[pos] = make_int2((int)s0.bestc, i | (s0.hinted ? 0 : 1 << 30));
  }
  __syncthreads();
  if (w.stats) {  // (uniform)
    const unsigned long long nkept = (unsigned long long)__popcll(__ballot(valid && s0.kept));  // (the ballot needs every lane)
    if (lane == 0) atomicAdd(w.stats + 6, nkept);
  }
  // (not on a cold tick: without hints every radius is infinite, every lane of every group a hit taken one at a time -- a tail block of a few
  // dozen points then walked the whole target serially)
  if (W > 1 && total > 0 && total <= w.sparse_max && M <= SORT_LDS_MAX_N && !cold) {
    // A FEW points left in this block (nine of 256 at tick 20 of a Gauss-Newton run, two or three dozen from tick 10 on): the
    // cooperative search below then runs its whole machinery -- group masks, eight chunk tests and a staged LDS tile per group, the
    // merge of four waves -- for lanes that are mostly empty: 700 wave-instructions per searching point at tick 20, what a brute-force
    // scan of the target would cost.  Here instead the waves of the block take the points in turn (wave w: points w, w + W, ...; lane
    // k of a wave keeps its k-th point's record) and work GROUP-major on registers: lane g holds the box of target group g (64 groups at a
    // time, at most 128: targets of up to 16384 points) and collects, one point per trip, the bit mask of the wave's points that need it (radius: what the hint gave the
    // point); every group some point needs is read ONCE straight from L2, two targets per lane, the next one in flight; each of its
    // points (broadcast by readlane) takes its two distances per lane, and a lane whose distances are beyond the point's radius --
    // nearly all of them -- drops out at one compare; the rest (the hint's target, a second candidate inside the skin) update the
    // point's record one lane at a time.  No chunk tests, no LDS tile, no barrier before the end.  Any exact pruning gives the
    // records of the brute-force search bit for bit (the radius only shrinks; what is never scanned lies beyond the final radius,
    // which is also all the keep-bound of the next tick claims -- nn_finish).
    const int ngroups = (M + kGroupPts - 1) / kGroupPts;
    const float inf = __builtin_inff();
    const float k_mul = skin_on ? w.skin_mul : 1.f, k_add = skin_on ? w.skin_add : 0.f;
    float4* resv = (float4*)txy;  // [point] {best, g1, g2, bits of bestc} (the tiles are not used on this path)
    const Box nobox{inf, inf, inf, inf, inf, inf};
    Box mybox = G(tgt.gbox)[min(lane, ngroups - 1)], mybox1 = G(tgt.gbox)[min(64 + lane, ngroups - 1)];  // (both requested now)
    if (lane >= ngroups) mybox = nobox;
    if (64 + lane >= ngroups) mybox1 = nobox;
    const int np = total > wid ? (total - wid + W - 1) / W : 0;  // points of this wave (<= 16 W / W ... at most 64 / W * ... <= 64)
    const bool hask = lane < np;
    const int ek = wid + W * (hask ? lane : 0);
    unsigned n_groups = 0;
    {
      const float4 cp = cpt[ek];
      const int2 ci = cci[ek];
      const float px = cp.x, py = cp.y, pz = cp.z;
      float best = cp.w, g1 = inf, g2 = inf;
      unsigned bestc = (unsigned)ci.x;
      float bestR = fmaf(best, k_mul, k_add);
      for (int gb0 = 0; gb0 < ngroups; gb0 += 64) {  // (the second 64 groups with the radii the first left behind)
        if (gb0) mybox = mybox1;
        unsigned long long pmask = 0;  // bit k: this lane's group is needed by point k of the wave
        for (int k = 0; k < np; k++) {
          const float qx = readlane_f(px, k), qy = readlane_f(py, k), qz = readlane_f(pz, k), qR = readlane_f(bestR, k);
          if (lb_point_box(mybox, qx, qy, qz) <= qR) pmask |= 1ull << k;
        }
        unsigned long long un = __ballot(pmask != 0);
        float4 a = make_float4(inf, inf, inf, 0.f), b = a;
        int gcur = -1;
        auto fetch = [&](int g) {  // (unconditional, clamped)
          const int j = (gb0 + g) * kGroupPts + 2 * lane;
          a = G(tgt.pts)[min(j, M - 1)];
          b = G(tgt.pts)[min(j + 1, M - 1)];
          gcur = g;
        };
        int g = un ? __builtin_ctzll(un) : -1;
        if (g >= 0) un &= un - 1, fetch(g);
        while (g >= 0) {
          float4 ca = a, cb = b;
          const int gl = gcur, gg = gb0 + gcur;
          if ((gg + 1) * kGroupPts > M) {  // (uniform) only the last group of a cloud can reach beyond it
            const int j = gg * kGroupPts + 2 * lane;
            if (j >= M) ca = make_float4(inf, inf, inf, 0.f);
            if (j + 1 >= M) cb = make_float4(inf, inf, inf, 0.f);
          }
          unsigned long long pm = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(pmask >> 32), gl) << 32) |
                                  (unsigned)__builtin_amdgcn_readlane((int)(unsigned)pmask, gl);
          g = un ? __builtin_ctzll(un) : -1;
          if (g >= 0) un &= un - 1, fetch(g);  // the next group travels while this one is looked at
          n_groups++;
          const v2f X = {ca.x, cb.x}, Y = {ca.y, cb.y}, Z = {ca.z, cb.z};
          while (pm) {
            const int k = __builtin_ctzll(pm);
            pm &= pm - 1;
            const float qx = readlane_f(px, k), qy = readlane_f(py, k), qz = readlane_f(pz, k), qR = readlane_f(bestR, k);
            const v2f dx = X - qx, dy = Y - qy, dz = Z - qz;  // (sqdist1's operations in sqdist1's order, per target)
            v2f r = dx * dx;
            r = r + dy * dy;
            r = r + dz * dz;
            const float m = fminf(r.x, r.y), m2 = fmaxf(r.x, r.y);
            unsigned long long hits = __ballot(m <= qR);
            while (hits) {
              const int l = __builtin_ctzll(hits);
              hits &= hits - 1;
              const float ml = readlane_f(m, l), m2l = readlane_f(m2, l);
              const unsigned c = (unsigned)(gg * kGroupChunks + (l >> 3));  // lane l: targets 2 l and 2 l + 1 of the group
              if (lane == k) {
                g2 = fminf(fmaxf(g1, ml), fminf(g2, m2l));
                g1 = fminf(g1, ml);
                if (ml < best) best = ml, bestc = c, bestR = fmaf(ml, k_mul, k_add);
                else if (ml == best && ml < inf && (bestc & kChunkMask) != c) bestc |= kTieBit;
              }
            }
          }
        }
      }
      if (hask) resv[ek] = make_float4(best, g1, g2, __uint_as_float(bestc));
    }
    __syncthreads();
    if (wid != 0) return;
    if (w.stats && lane == 0) atomicAdd(w.stats + 0, (unsigned long long)n_groups), atomicAdd(w.stats + 3, 1ull);
    const bool has = lane < total;
    const float4 cp = cpt[has ? lane : total - 1];
    const int2 ci = cci[has ? lane : total - 1];
    const float4 rv = resv[has ? lane : total - 1];
    float px[1] = {cp.x}, py[1] = {cp.y}, pz[1] = {cp.z}, best[1] = {rv.x}, g1[1] = {rv.y}, g2[1] = {rv.z};
    unsigned bestc[1] = {__float_as_uint(rv.w)};
    int pidx[1] = {has ? (ci.y & ~(1 << 30)) : -1};
    bool kept[1] = {!has};
    nn_finish<1>(tgt, w, pair, lane, skin_on, k_mul, k_add, px, py, pz, best, bestc, pidx, kept, g1, g2);
    if (has) out[pidx[0]] = ((unsigned long long)__float_as_uint(best[0]) << 32) | bestc[0];
    return;
  }
  if (W > 1 && w.coop_search && total > 0 && total <= 64) {
    // One wave's worth of points left in this block (the usual case from the middle of a Gauss-Newton run on: 63 of 256
    // points search at tick 10, 9 at tick 20): the waves that would leave now split the target groups of that ONE wave between
    // them instead -- a lone search wave is a serial chain of about 19 us, 12 of them in fetching and scanning its groups.
    const bool has = lane < total;
    const float4 cp = cpt[has ? lane : total - 1];
    const int2 ci = cci[has ? lane : total - 1];
    __syncthreads();  // (cpt / cci become the scratch of the merge)
    float px[1] = {cp.x}, py[1] = {cp.y}, pz[1] = {cp.z}, best[1] = {cp.w};
    unsigned bestc[1] = {(unsigned)ci.x};
    int pidx[1] = {has ? (ci.y & ~(1 << 30)) : -1};
    bool kept[1] = {!has};
    nn_search<1, W, true>(src, tgt, T0, w, pair, 0, lane, wid, cold, txy[wid], tz[wid], cbl[wid], gbl[0], (unsigned long long*)cpt, (float2*)cci, px, py, pz,
                          best, bestc, pidx, kept, !has || (ci.y & (1 << 30)) == 0);
    if (wid == 0 && has) out[pidx[0]] = ((unsigned long long)__float_as_uint(best[0]) << 32) | bestc[0];
    return;
  }
  if (wid * 64 >= total) {  // nothing left for this wave
    if (w.stats && lane == 0) atomicAdd(w.stats + 3, 1ull);
    return;
  }
  const int e = wid * 64 + lane;
  const bool has = e < total;
  const float4 cp = cpt[has ? e : total - 1];
  const int2 ci = cci[has ? e : total - 1];
  float px[1] = {cp.x}, py[1] = {cp.y}, pz[1] = {cp.z}, best[1] = {cp.w};
  unsigned bestc[1] = {(unsigned)ci.x};
  int pidx[1] = {has ? (ci.y & ~(1 << 30)) : -1};
  bool kept[1] = {!has};
  nn_search<1, 1, true>(src, tgt, T0, w, pair, 0, lane, 0, cold, txy[wid], tz[wid], cbl[wid], gbl[wid], nullptr, nullptr, px, py, pz, best, bestc, pidx,
                        kept, !has || (ci.y & (1 << 30)) == 0);
  if (has) out[pidx[0]] = ((unsigned long long)__float_as_uint(best[0]) << 32) | bestc[0];
}

// ----------------------------------------------------------------------------------------------

// wave-wide bitonic sort (ascending over lane id) of one 64-bit key per lane
__device__ __forceinline__ unsigned long long wave_sort_u64(unsigned long long v, int lane) {
#pragma unroll
  for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      const unsigned lo = __shfl_xor((unsigned)v, j, 64), hi = __shfl_xor((unsigned)(v >> 32), j, 64);
      const unsigned long long o = ((unsigned long long)hi << 32) | lo;
      const bool asc = (lane & k) == 0, lower = (lane & j) == 0;
      const bool take_min = lower == asc;
      v = take_min ? (o < v ? o : v) : (o > v ? o : v);
    }
  }
  return v;
}
// A threshold tk with k <= #{keys <= tk} <= keep_max among the up-to-3 keys per lane (~0ull = absent; the keys are unique and more
// than keep_max of them are present): bisection by rank with the keys themselves as probes -- the key in the lowest lane still
// between the bounds, its rank counted with two or three ballots; a probe outside [k, keep_max] becomes the new lower or upper
// bound.  A key of every rank exists, so some probe ends inside the window: about log2(n / (keep_max - k)) + 1 probes of ~20
// instructions each.  Enough for tightening a full list: tau stays an upper bound of the k-th neighbour distance, the list shrinks
// to at most keep_max entries.  (Until round 3 a most-significant-bit-first radix selection on ballot masks returned the EXACT k-th
// smallest key: up to 32 + 13 bit steps of ~10 scalar instructions, 3 to 4 times per wave -- a fifth of the kernel's time:
// 64 clouds 0.494 -> 0.404 ms with the window, whose lists keep up to twice as many entries and overflow 3.7 instead of 2.9 times.)
__device__ __forceinline__ unsigned long long wave_kth_window3(unsigned long long a, unsigned long long b, unsigned long long c, int k, int keep_max) {
  unsigned long long ea = __ballot(a != ~0ull), eb = __ballot(b != ~0ull), ec = __ballot(c != ~0ull);  // still between the bounds
  for (;;) {
    unsigned long long x;
    if (ea) {
      const int l = __builtin_ctzll(ea);
      x = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(a >> 32), l) << 32) | (unsigned)__builtin_amdgcn_readlane((int)(unsigned)a, l);
    } else if (eb) {
      const int l = __builtin_ctzll(eb);
      x = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), l) << 32) | (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, l);
    } else if (ec) {
      const int l = __builtin_ctzll(ec);
      x = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(c >> 32), l) << 32) | (unsigned)__builtin_amdgcn_readlane((int)(unsigned)c, l);
    } else {
      return ~0ull;  // no key left between the bounds: cannot happen with unique keys; the caller's capacity check reports it, nothing spins
    }
    const int cnt = (int)(__popcll(__ballot(a <= x)) + __popcll(__ballot(b <= x)) + __popcll(__ballot(c <= x)));
    if (cnt < k) ea &= __ballot(a > x), eb &= __ballot(b > x), ec &= __ballot(c > x);
    else if (cnt > keep_max) ea &= __ballot(a < x), eb &= __ballot(b < x), ec &= __ballot(c < x);
    else return x;
  }
}
// base + number of set bits of `mask` below this lane
__device__ __forceinline__ int mbcnt_add(unsigned long long mask, int base) {
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, (unsigned)base));
}
// mask bit of this lane ? a : b, with the mask where a ballot left it (written as `(m >> lane) & 1 ? a : b` the compiler rebuilds
// the lane predicate with a compare of its own)
// (gfx940 / gfx950: a VALU instruction that reads an SGPR needs TWO wait states behind the VALU instruction that wrote it -- v_cmp, v_readlane --
// and the compiler, which inserts them for its own instructions, does not look into inline asm: the s_nop belongs to the asm.  Round 6 found
// this the hard way: the sixteenth v_writelane of a mask loop directly behind the v_cmp that produced its operand wrote a stale mask.)
__device__ __forceinline__ int lane_select(unsigned long long mask, int a, int b) {
  int r;
#ifdef APD_AB_NO_ASM_NOP  // (A/B builds only: what the two wait states cost; such a build may read a stale mask)
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(mask));
#else
  asm("s_nop 1\n v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(mask));
#endif
  return r;
}
// v_writelane_b32 with a constant lane (this toolchain has no builtin for it; one scalar source at most besides the lane)
template <int LANE>
__device__ __forceinline__ void writelane_const(int& v, int val) {
#ifdef APD_AB_NO_ASM_NOP
  asm("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(val), "n"(LANE));
#else
  asm("s_nop 1\n v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(val), "n"(LANE));  // (see lane_select: the operand may come straight from a v_cmp)
#endif
}
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

__device__ __forceinline__ unsigned long long dist_key(float d, int orig) { return ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)orig; }

// ----------------------------------------------------------------------------------------------
// k_knn_cov_coop<L>: calculate_covariances (A:303-363) on curve-sorted clouds; exact k-NN.  One wave owns 64 / L consecutive
// sorted queries (spatial neighbours, so they need the same few target groups), L = 4, 8 or 16 lanes per query, and
// alternates between two lane mappings:
//   A  lane = (query, 1/L of the work): tau = k-th smallest of 32 strided class minima over the KQ_WIN sorted neighbours
//      around the wave, read straight from L1 (k distinct points lie within tau, so it bounds the k-th neighbour distance);
//      the L lanes of a query split the 32 classes (L = 4: sorted in place by a network over (lane, register)).  Group
//      masks: lane g keeps the box of group g in registers and tests it against one query per trip, broadcast through
//      SGPRs -- the ballot is that query's mask;
//   B  lane = candidate: every group some query needs is loaded ONCE (2 candidates per lane, coalesced); for each query
//      that needs it (uniform loop, query broadcast by readlane) the 128 keys (distance bits << 32 | original index) are
//      compared with the query's tau and the hits are compacted into its LDS list with ballot / popcount -- no divergence.
//      A full list is tightened in place to its k smallest keys (tau only decreases, nothing of the final answer is
//      dropped); the query state (tau, count) is kept identical in the L lanes;
//   C  lane = (query, 1/L): each lane sorts its share of the list in REGISTERS, the k neighbours come out of an L-way merge
//      (a DPP butterfly over the heads per round, no LDS reads inside the rounds; u64 order == the reference's (distance,
//      index) order); gather of the k points, moments in fp64 relative to the query (exact differences) as four partial
//      sums in rank order (Mom9), cov = S2/k - m m^T (A:323-324), 3x3 Jacobi eigen-decomposition and the regularisation
//      (A:326-357) -- in the epilogue for a single cloud, in k_regularize_covs for batches.
// (The one-lane-per-query predecessor of this kernel, k_knn_cov_pruned, is in the history of round 2 and in docs/experiments.md.)
#ifndef APD_KNN_WIN
#define APD_KNN_WIN 128
#endif
constexpr int KQ_WIN = APD_KNN_WIN;  // window: measured 64 / 96 / 128 / 192 / 256 -> 0.683 / 0.674 / 0.675 / 0.695 / 0.735 ms (64 clouds)
// list entries per query -- 60 with 4 lanes per query (16 queries per wave: 7.8 KB of lists, what five resident waves per SIMD
// leave of the LDS; 48 overflowed 3.8 times per wave, 60 does 2.9 times, and every overflow is a radix select of about 300
// scalar instructions on the ONE scalar unit the four SIMDs share), 48 otherwise
#ifndef APD_KNN_CAP4
#define APD_KNN_CAP4 60
#endif
__host__ __device__ constexpr int knn_coop_cap(int lanes_per_query) { return lanes_per_query == 4 ? APD_KNN_CAP4 : 48; }
#ifndef APD_KNN_KEEP
#define APD_KNN_KEEP 40
#endif
constexpr int KQ_KEEP = APD_KNN_KEEP;  // a full list is tightened to between k and this many entries (32 / 40 / 48: 0.409 / 0.404 / 0.406 ms)
__host__ __device__ constexpr int knn_coop_lds_bytes(int qpw) { return qpw * (knn_coop_cap(64 / qpw) + 1) * 8; }  // lists only (boxes in registers, window from L1)

template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_u64(unsigned long long v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)v, CTRL, 0xf, 0xf, false);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), CTRL, 0xf, 0xf, false);
  return ((unsigned long long)hi << 32) | lo;
}
// butterflies over aligned groups of L lanes: quad_perm [1,0,3,2], quad_perm [2,3,0,1], then (L >= 8) row_half_mirror and
// (L == 16) row_mirror, which pair every lane with one of the other half once the halves are uniform
// Keys as doubles.  A key is (bits of a finite non-negative float) << 32 | index: as an IEEE double that is a finite non-negative
// number (the high word stays below 0x7F800000 < 0x7FF00000), and non-negative doubles order like their bit patterns -- so the
// smaller / larger of two keys is ONE v_min_f64 / v_max_f64 instead of a 64-bit compare and two selects per word.  fp64
// denormals are never flushed on this target (keys of tiny distances are denormal doubles), nothing here is a NaN: the
// "absent" key of phase C is +infinity (kKeyInf), not ~0, which as a double would be a NaN that min / max drop.
constexpr unsigned long long kKeyInf = 0x7FF0000000000000ull;
// (the instructions themselves: through fmin() / fmax() the compiler first canonicalises every operand it has not produced itself)
__device__ __forceinline__ unsigned long long key_min(unsigned long long a, unsigned long long b) {
  unsigned long long r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ unsigned long long key_max(unsigned long long a, unsigned long long b) {
  unsigned long long r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
template <int L>
__device__ __forceinline__ unsigned long long group_min_u64(unsigned long long v) {
  v = key_min(v, dpp_u64<0xB1>(v));
  v = key_min(v, dpp_u64<0x4E>(v));
  if (L >= 8) v = key_min(v, dpp_u64<0x141>(v));   // row_half_mirror
  if (L == 16) v = key_min(v, dpp_u64<0x140>(v));  // row_mirror: pairs the two (by now uniform) halves of the 16-lane row
  return v;
}
template <int CTRL>
__device__ __forceinline__ Mom9 mom_dpp(const Mom9& m) {  // the partner lane's sums
  auto f = [](double v) { return __longlong_as_double((long long)dpp_u64<CTRL>((unsigned long long)__double_as_longlong(v))); };
  Mom9 o;
  o.s1x = f(m.s1x), o.s1y = f(m.s1y), o.s1z = f(m.s1z), o.sxx = f(m.sxx), o.sxy = f(m.sxy), o.sxz = f(m.sxz), o.syy = f(m.syy), o.syz = f(m.syz), o.szz = f(m.szz);
  return o;
}
template <int L>
__device__ __forceinline__ unsigned long long group_or_u64(unsigned long long v) {
  v |= dpp_u64<0xB1>(v);
  v |= dpp_u64<0x4E>(v);
  if (L >= 8) v |= dpp_u64<0x141>(v);
  if (L == 16) v |= dpp_u64<0x140>(v);
  return v;
}

// (5 waves per SIMD asked for: the allocation sits at 95 .. 97 registers, and 97 would cost the fifth wave)
#ifndef APD_KNN_WPE
#define APD_KNN_WPE 5
#endif
// The work of ONE wave: the 64 / L queries starting at bx * (64 / L) of cloud c; knn_smem: the wave's own knn_coop_lds_bytes(64 / L)
// bytes of LDS.  (k_knn_cov_coop: one wave per block; k_knn_and_search: eight.)
template <int L>
__device__ __forceinline__ void knn_cov_coop_wave(const CloudDesc& c, unsigned bx, int lane, unsigned long long* knn_smem, int k, int reg, int* err_flag,
                                                  unsigned long long* stats, int raw /* 1: store the population covariance, k_regularize_covs follows */) {
  constexpr int QPW = 64 / L, NCL = KNN_NC / L, KQ_CAP = knn_coop_cap(L), KQ_STRIDE = KQ_CAP + 1, EPL = KQ_CAP / L;  // queries per wave, classes and list entries per lane
  static_assert(L == 4 || L == 8 || L == 16, "L lanes per query");
  static_assert(KQ_CAP % L == 0 && KQ_CAP <= 64 && KQ_WIN % KNN_NC == 0, "layout");
  unsigned long long* lst = knn_smem;                            // [query][slot], padded row
  float* cml = (float*)knn_smem;                                 // phase A only: [query][33] class minima (the lists are still empty)
  const int n = c.n;
  const int base = (int)bx * QPW;
  if (base >= n) return;
  const float inf = __builtin_inff();
  const int slot = lane / L, sub = lane % L, owner = lane - sub;
  const int i = base + slot;
  const bool valid = i < n;
  const float4 q = G(c.pts)[valid ? i : n - 1];
  const int ngroups = (n + kGroupPts - 1) / kGroupPts;
  unsigned n_groups = 0, n_pairs = 0, n_compact = 0;
  if ((bx & 63) != 0) stats = nullptr;  // diagnostics sample every 64th wave (the atomics would dominate otherwise)

  long long tA = 0, tG = 0, tB = 0, tC = 0, tm = stats ? clock64() : 0;
  // ---- A: bound from the sorted neighbourhood
  const int w0 = min(max(base - (KQ_WIN - QPW) / 2, 0), max(n - KQ_WIN, 0));
  const Box nobox{inf, inf, inf, inf, inf, inf};
  Box mybox = G(c.gbox)[min(lane, ngroups - 1)];
  if (lane >= ngroups) mybox = nobox;  // lane g keeps the box of group gb0 + g in registers: no LDS copy
  int cnt = 0;  // entries in this query's list (same value in its L lanes)
  wave_lds_fence();  // (the wave's own LDS region: no block barrier -- eight independent waves share a block in k_knn_and_search)
  float tau_q = inf;  // (L == 4) the k-th smallest class minimum
  {
    float cmp[NCL];  // minima of the classes sub + L*m
#pragma unroll
    for (int m = 0; m < NCL; m++) cmp[m] = inf;
    // straight from global memory: the L lanes of a query read L consecutive points, the queries of a wave the same
    // window, so the lines stay in L1; no LDS copy of the window (3 KB per wave more for resident waves).
    // The whole window lies inside the cloud unless the cloud is smaller than the window: a per-element bound check made
    // the compiler branch around every load and wait for each one on its own (32 dependent round trips to memory per wave)
    if (n >= KQ_WIN) {
      if constexpr (L == 16) {  // eight points per lane: every load in flight together (one round trip instead of four)
        float4 t[KQ_WIN / L];
#pragma unroll
        for (int u = 0; u < KQ_WIN / L; u++) t[u] = G(c.pts)[w0 + sub + L * u];
#pragma unroll
        for (int u = 0; u < KQ_WIN / L; u++) cmp[u % NCL] = fminf(cmp[u % NCL], sqdist1(t[u].x, t[u].y, t[u].z, q.x, q.y, q.z));
      } else {
      for (int t0 = 0; t0 < KQ_WIN / L; t0 += NCL) {
        float4 t[NCL];
#pragma unroll
        for (int m = 0; m < NCL; m++) t[m] = G(c.pts)[w0 + sub + L * (t0 + m)];
#pragma unroll
        for (int m = 0; m < NCL; m++) cmp[m] = fminf(cmp[m], sqdist1(t[m].x, t[m].y, t[m].z, q.x, q.y, q.z));
      }
      }
    } else {
      for (int t0 = 0; t0 < KQ_WIN / L; t0 += NCL) {
#pragma unroll
        for (int m = 0; m < NCL; m++) {
          const int j = w0 + sub + L * (t0 + m);
          const float4 t = G(c.pts)[min(j, n - 1)];
          cmp[m] = fminf(cmp[m], j < n ? sqdist1(t.x, t.y, t.z, q.x, q.y, q.z) : inf);
        }
      }
    }
    if constexpr (L == 4) {
      // Four lanes per query: the 32 class minima stay where they are, 8 per lane, and are sorted in place by a bitonic
      // network over (lane, register) -- element sub * 8 + r; every merge starts with the mirror step i <-> i ^ (kk - 1), so
      // all compare-exchanges are ascending -- 12 in-lane stages and 3 stages whose partner sits in another lane of the quad
      // (quad_perm): 190 instructions instead of the 32 LDS reads and the 480 of a full network run by every lane.  The
      // keys are non-negative floats (or +inf): their bit patterns order like unsigned integers, no canonicalisation needed.
      unsigned v[8];
#pragma unroll
      for (int m = 0; m < 8; m++) v[m] = __float_as_uint(cmp[m]);
      auto ce = [&](int a, int b) {
        const unsigned lo = min(v[a], v[b]), hi = max(v[a], v[b]);
        v[a] = lo, v[b] = hi;
      };
      auto in_lane = [&](int j) {  // partners r <-> r ^ j
#pragma unroll
        for (int r = 0; r < 8; r++)
          if ((r & j) == 0) ce(r, r ^ j);
      };
      auto mirror_in_lane = [&](int kk) {  // partners r <-> r ^ (kk - 1)
#pragma unroll
        for (int r = 0; r < 8; r++)
          if (r < (r ^ (kk - 1))) ce(r, r ^ (kk - 1));
      };
      mirror_in_lane(2);
      mirror_in_lane(4), in_lane(1);
      mirror_in_lane(8), in_lane(2), in_lane(1);
      {  // kk = 16: element i <-> i ^ 15: lane sub ^ 1, register r ^ 7
        const bool lower = (sub & 1) == 0;
        unsigned y[8];
#pragma unroll
        for (int r = 0; r < 8; r++) y[r] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v[7 - r], 0xB1, 0xf, 0xf, false);
#pragma unroll
        for (int r = 0; r < 8; r++) v[r] = lower ? min(v[r], y[r]) : max(v[r], y[r]);
      }
      in_lane(4), in_lane(2), in_lane(1);
      {  // kk = 32: i <-> i ^ 31: lane sub ^ 3, register r ^ 7; then i <-> i ^ 8: lane sub ^ 1, same register
        const bool lower3 = sub < 2, lower1 = (sub & 1) == 0;
        unsigned y[8];
#pragma unroll
        for (int r = 0; r < 8; r++) y[r] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v[7 - r], 0x1B, 0xf, 0xf, false);
#pragma unroll
        for (int r = 0; r < 8; r++) v[r] = lower3 ? min(v[r], y[r]) : max(v[r], y[r]);
#pragma unroll
        for (int r = 0; r < 8; r++) y[r] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v[r], 0xB1, 0xf, 0xf, false);
#pragma unroll
        for (int r = 0; r < 8; r++) v[r] = lower1 ? min(v[r], y[r]) : max(v[r], y[r]);
      }
      in_lane(4), in_lane(2), in_lane(1);
      // the k-th smallest: element k - 1 = register (k - 1) & 7 of lane (k - 1) >> 3 of the quad
      unsigned pick = v[0];
#pragma unroll
      for (int r = 1; r < 8; r++) pick = ((k - 1) & 7) == r ? v[r] : pick;
      cmp[0] = __uint_as_float((unsigned)__shfl((int)pick, owner + ((k - 1) >> 3), 64));
    } else if constexpr (L == 16) {
      // Sixteen lanes per query: the 32 class minima sit two per lane (classes sub and 16 + sub) and are sorted where they are,
      // position p = 16 r + sub, by the same all-ascending bitonic network -- partners in another lane of the 16-lane row come by
      // DPP (quad_perm, row_half_mirror, row_mirror, row_ror:8, row_shl/shr:4), partners 16 positions away are the lane's other
      // register: about 130 instructions.  (Until round 3 every one of the 16 lanes read all 32 minima back from LDS and ran
      // the full 32-key network on its own: 720 instructions, a quarter of this variant's wave.)
      unsigned v0 = __float_as_uint(cmp[0]), v1 = __float_as_uint(cmp[1]);
      auto dpp = [](unsigned x, auto ctrl_) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, decltype(ctrl_)::value, 0xf, 0xf, false); };
      using IC = std::integral_constant<int, 0>;
      (void)sizeof(IC);
      auto lane_stage = [&](auto ctrl_, int low_bit) {  // partner (r, sub ^ J) through DPP control `ctrl_`; the smaller key stays where sub & low_bit == 0
        const bool lower = (sub & low_bit) == 0;
        const unsigned o0 = dpp(v0, ctrl_), o1 = dpp(v1, ctrl_);
        v0 = lower ? min(v0, o0) : max(v0, o0);
        v1 = lower ? min(v1, o1) : max(v1, o1);
      };
      auto xor4_stage = [&]() {  // lane ^ 4: row_shl:4 (lane + 4) / row_shr:4 (lane - 4), chosen by lane bit 2
        const bool lower = (sub & 4) == 0;
        // (both moves by EVERY lane, then the choice: written as `lower ? shl : shr` the compiler branches, and a DPP move executed
        // by half of the lanes reads its sources from the inactive other half)
        const unsigned up0 = dpp(v0, std::integral_constant<int, 0x104>{}), dn0 = dpp(v0, std::integral_constant<int, 0x114>{});
        const unsigned up1 = dpp(v1, std::integral_constant<int, 0x104>{}), dn1 = dpp(v1, std::integral_constant<int, 0x114>{});
        const unsigned o0 = lower ? up0 : dn0, o1 = lower ? up1 : dn1;
        v0 = lower ? min(v0, o0) : max(v0, o0);
        v1 = lower ? min(v1, o1) : max(v1, o1);
      };
      const std::integral_constant<int, 0xB1> X1{};    // quad_perm [1,0,3,2]: lane ^ 1
      const std::integral_constant<int, 0x4E> X2{};    // quad_perm [2,3,0,1]: lane ^ 2
      const std::integral_constant<int, 0x1B> M3{};    // quad_perm [3,2,1,0]: lane ^ 3
      const std::integral_constant<int, 0x141> M7{};   // row_half_mirror: lane ^ 7
      const std::integral_constant<int, 0x140> M15{};  // row_mirror: lane ^ 15
      const std::integral_constant<int, 0x128> X8{};   // row_ror:8: lane ^ 8
      lane_stage(X1, 1);                                                       // kk = 2
      lane_stage(M3, 2), lane_stage(X1, 1);                                    // kk = 4
      lane_stage(M7, 4), lane_stage(X2, 2), lane_stage(X1, 1);                 // kk = 8
      lane_stage(M15, 8), xor4_stage(), lane_stage(X2, 2), lane_stage(X1, 1);  // kk = 16
      {  // kk = 32: p <-> p ^ 31 = (r ^ 1, sub ^ 15): register 0 keeps the minimum with the mirrored register 1
        const unsigned m0 = dpp(v0, M15), m1 = dpp(v1, M15);
        v0 = min(v0, m1), v1 = max(v1, m0);
      }
      lane_stage(X8, 8), xor4_stage(), lane_stage(X2, 2), lane_stage(X1, 1);
      // the k-th smallest: position k - 1 = register (k - 1) >> 4 of lane (k - 1) & 15 of the query's row
      const unsigned pick = ((k - 1) >> 4) ? v1 : v0;
      cmp[0] = __uint_as_float((unsigned)__shfl((int)pick, owner + ((k - 1) & 15), 64));
    } else {
#pragma unroll
      for (int m = 0; m < NCL; m++) cml[slot * (KNN_NC + 1) + sub + L * m] = cmp[m];
    }
    tau_q = cmp[0];
  }
  float tau_d = inf;
  if constexpr (L == 4 || L == 16) {
    tau_d = tau_q;
  } else {
  wave_lds_fence();
  float cm[KNN_NC];
#pragma unroll
  for (int s = 0; s < KNN_NC; s++) cm[s] = cml[slot * (KNN_NC + 1) + s];
  wave_lds_fence();  // cml aliases the lists
#pragma unroll
  for (int kk = 2; kk <= KNN_NC; kk <<= 1) {
#pragma unroll
    for (int j = kk >> 1; j > 0; j >>= 1) {
#pragma unroll
      for (int a = 0; a < KNN_NC; a++) {
        const int l = a ^ j;
        if (l > a) {
          const bool up = (a & kk) == 0;
          const float x = cm[a], y = cm[l];
          const float lo = fminf(x, y), hi = fmaxf(x, y);
          cm[a] = up ? lo : hi;
          cm[l] = up ? hi : lo;
        }
      }
    }
  }
#pragma unroll
  for (int s = 0; s < KNN_NC; s++)
    if (s == k - 1) tau_d = cm[s];
  }
  if (!valid) tau_d = -1.f;
  unsigned tau_hi = __float_as_uint(tau_d), tau_lo = 0xFFFFFFFFu;  // tau key = (tau_hi << 32) | tau_lo

  if (stats) { const long long t = clock64(); tA += t - tm, tm = t; }
  // ---- B
  // large clouds: super boxes first (apd_sort.hpp), with the bounds of phase A -- tau only decreases afterwards
  const bool use_super = n > SORT_LDS_MAX_N;
  const int nsuper = (ngroups + kSuperGroups - 1) / kSuperGroups;
  for (int sb0 = 0; sb0 < nsuper; sb0 += 64) {
  unsigned long long smask = nsuper - sb0 >= 64 ? ~0ull : (1ull << (nsuper - sb0)) - 1ull;
  if (use_super) {
    Box sbx = G(c.gbox)[ngroups + min(sb0 + lane, nsuper - 1)];
    if (sb0 + lane >= nsuper) sbx = nobox;
    unsigned long long sany = 0;
#pragma unroll
    for (int qi = 0; qi < QPW; qi++) {
      const float qx = readlane_f(q.x, qi * L), qy = readlane_f(q.y, qi * L), qz = readlane_f(q.z, qi * L);
      const float td = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)tau_hi, qi * L));
      sany |= __ballot(lb_point_box(sbx, qx, qy, qz) <= td);
    }
    smask &= sany;
  }
  while (smask) {
    const int gb0 = (sb0 + __builtin_ctzll(smask)) * kSuperGroups;
    smask &= smask - 1;
    if (gb0 > 0) {
      mybox = G(c.gbox)[min(gb0 + lane, ngroups - 1)];
      if (gb0 + lane >= ngroups) mybox = nobox;
    }
    // group masks, one query per trip: lane g tests ITS box against the query broadcast through SGPRs, the ballot IS the
    // query's mask (a finished or padding query has tau = -1 and gets none); gany = groups some query of this wave needs
    unsigned long long gany = 0;
    int gneed_lo = 0, gneed_hi = 0;
    static_for<0, QPW>([&](auto qi_) {
      constexpr int qi = decltype(qi_)::value;
      const float qx = readlane_f(q.x, qi * L), qy = readlane_f(q.y, qi * L), qz = readlane_f(q.z, qi * L);
      const float td = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)tau_hi, qi * L));
      // (a lane without a group holds the all-infinite box: its bound is +inf, no `lane < nb` needed -- as a condition it became
      // a branch around the bound in every one of the QPW trips)
      const unsigned long long mq = __ballot(lb_point_box(mybox, qx, qy, qz) <= td);
      gany |= mq;
      // only the query's owner lane reads its mask (qm below): two v_writelane instead of a select per lane and trip, whose QPW
      // lane masks the compiler kept in 2 QPW scalar registers for the whole kernel
      writelane_const<qi * L>(gneed_lo, (int)(unsigned)mq);
      writelane_const<qi * L>(gneed_hi, (int)(unsigned)(mq >> 32));
    });
    const unsigned long long gneed = ((unsigned long long)(unsigned)gneed_hi << 32) | (unsigned)gneed_lo;
    if (stats) { const long long t = clock64(); tG += t - tm, tm = t; }
    float4 c0, c1, d0, d1;
    unsigned o0, o1;
    int gcur = -1;  // the group whose points are in d0 / d1
    // The two loads of a group are UNCONDITIONAL (clamped addresses) and nothing looks at them here: what lies beyond the
    // cloud is replaced when the group becomes current (`take`).  A load that sits in a branch of its own -- `if (j < n)
    // a = pts[j]`, also when written as a select -- gets its `s_waitcnt vmcnt(0)` right behind it: the group requested "one
    // ahead" then arrives before the scan of the current one starts, two dependent trips to memory per group.
    auto fetch = [&](int g) {
      const int j0 = (gb0 + g) * kGroupPts + lane, j1 = j0 + 64;
      d0 = G(c.pts)[min(j0, n - 1)];
      d1 = G(c.pts)[min(j1, n - 1)];
      gcur = g;
    };
    v2f cx, cy, cz;  // the two candidates of this lane, component-wise: every step's distances are packed operations on these
    auto take = [&]() {  // d -> c (the sorted points carry their original index in .w)
      c0 = d0, c1 = d1, o0 = __float_as_uint(d0.w), o1 = __float_as_uint(d1.w);
      if ((gb0 + gcur + 1) * kGroupPts > n) {  // (uniform) only the last group of a cloud can reach beyond it
        const int j0 = (gb0 + gcur) * kGroupPts + lane, j1 = j0 + 64;
        if (j0 >= n) c0 = make_float4(inf, inf, inf, 0.f), o0 = 0xFFFFFFFFu;
        if (j1 >= n) c1 = make_float4(inf, inf, inf, 0.f), o1 = 0xFFFFFFFFu;
      }
      cx = v2f{c0.x, c1.x}, cy = v2f{c0.y, c1.y}, cz = v2f{c0.z, c1.z};
    };
        // L >= 8 (one or two clouds per launch, every wave resident: the launch lasts as long as its slowest wave): nearest groups
    // first, by the lower bound to the wave's middle query, so that tau is tight before the far groups are reached and the
    // lists overflow less often.  In index order a few waves tightened 36 times (17 along the Z-curve): two clouds of 8192
    // points 0.123 -> 0.07 ms.
    const float lbg = lb_point_box(mybox, readlane_f(q.x, 32), readlane_f(q.y, 32), readlane_f(q.z, 32));
    auto next_group = [&]() {
      if (!gany) return -1;
      if constexpr (L < 8) {  // many clouds per launch: throughput counts, index order is 2 % cheaper
        const int g0_ = __builtin_ctzll(gany);
        gany &= gany - 1;
        return g0_;
      }
      const bool candl = ((gany >> lane) & 1ull) != 0;
      const float mlb = wave_minmax_uniform<false>(candl ? lbg : inf);
      const unsigned long long eq = __ballot(candl && lbg == mlb);
      const int gsel = __builtin_ctzll(eq ? eq : gany);
      gany &= ~(1ull << gsel);
      return gsel;
    };
    int g = next_group();
    if (g >= 0) fetch(g);
    while (g >= 0) {
      take();
      unsigned long long qm = __ballot(((gneed >> g) & 1ull) != 0 && sub == 0);
      g = next_group();
      if (g >= 0) fetch(g);
      n_groups++;
      n_pairs += (unsigned)__popcll(qm);
      // (The scalar unit is what this loop waits for -- see the probe in docs/experiments.md -- so its bookkeeping is written for
      // it: one s_bitset0 instead of the add / addc / and of `qm &= qm - 1`, the step counter outside, stores without exec-mask
      // regions, the can't-happen check only behind a tightening.)
      while (qm) {
        const int qq = __builtin_ctzll(qm);  // owner lane of the query
        asm("s_bitset0_b64 %0, %1" : "+s"(qm) : "s"(qq));
        const float qx = readlane_f(q.x, qq), qy = readlane_f(q.y, qq), qz = readlane_f(q.z, qq);
#if defined(APD_PROBE_SALU) || defined(APD_PROBE_VALU)
        {  // issue probe (tools/probe_issue.sh, docs/experiments.md): n extra independent scalar / vector instructions per step
          int t_ = 0;
#if defined(APD_PROBE_SALU)
#pragma unroll
          for (int u_ = 0; u_ < APD_PROBE_SALU; u_++) asm volatile("s_add_u32 %0, %0, 1" : "+s"(t_));
#else
#pragma unroll
          for (int u_ = 0; u_ < APD_PROBE_VALU; u_++) asm volatile("v_add_u32 %0, %0, 1" : "+v"(t_));
#endif
        }
#endif
        unsigned long long tk = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(tau_hi, qq) << 32) |
                                (unsigned long long)(unsigned)__builtin_amdgcn_readlane(tau_lo, qq);  // readlane returns int: no sign extension
        const v2f ddx = cx - qx, ddy = cy - qy, ddz = cz - qz;  // (sqdist1's operations in sqdist1's order, per candidate)
        v2f dd = ddx * ddx;
        dd = dd + ddy * ddy;
        dd = dd + ddz * ddz;
        const unsigned long long k0 = ((unsigned long long)__float_as_uint(dd.x) << 32) | o0;
        const unsigned long long k1 = ((unsigned long long)__float_as_uint(dd.y) << 32) | o1;
        unsigned long long* row = lst + (qq / L) * KQ_STRIDE;
        unsigned long long m0 = __ballot(k0 <= tk), m1 = __ballot(k1 <= tk);
        if ((m0 | m1) == 0) continue;  // the box passed the test, none of its 128 points does
        int cntq = __builtin_amdgcn_readlane(cnt, qq);
        if (cntq + __popcll(m0) + __popcll(m1) > KQ_CAP) {
          // List full: tau becomes a key of rank k .. KQ_KEEP among (stored keys U this group's hits); only keys <= tau survive.
          // tau only decreases and stays >= the k-th smallest key seen, so nothing of the final answer is ever dropped, and at
          // most KQ_KEEP keys remain afterwards, which always fits.
          const unsigned long long mine = lane < cntq ? row[lane] : ~0ull;  // KQ_CAP <= 64: one entry per lane
          const unsigned long long h0 = k0 <= tk ? k0 : ~0ull, h1 = k1 <= tk ? k1 : ~0ull;
          tk = wave_kth_window3(mine, h0, h1, k, max(k, KQ_KEEP));
          const unsigned long long keep = __ballot(mine <= tk);
          if (mine <= tk) row[__popcll(keep & ((1ull << lane) - 1ull))] = mine;
          cntq = __popcll(keep);
          if (owner == qq) tau_hi = (unsigned)(tk >> 32), tau_lo = (unsigned)tk;
          m0 = __ballot(k0 <= tk), m1 = __ballot(k1 <= tk);
          n_compact++;
          if (cntq + (int)__popcll(m0) + (int)__popcll(m1) > KQ_CAP) {  // impossible by construction (<= KQ_KEEP now); never silently wrong
            if (lane == 0) atomicExch(err_flag, 4);
            m0 = m1 = 0;
          }
        }
        const int n0 = cntq + (int)__popcll(m0), total = n0 + (int)__popcll(m1);
        // Every lane stores both keys, the ones that are no hits into the row's spare slot (KQ_CAP: never read) -- as `if (hit)
        // row[..] = key` each store sat in an exec-mask region of its own: s_and_saveexec, s_cbranch_execz, s_or exec per key.
        // Slot = entries before + hits in lower lanes: v_mbcnt adds its count to an operand (as `popc(m & below)` the compiler
        // builds two ANDs and two bit counts per key).
        row[lane_select(m0, mbcnt_add(m0, cntq), KQ_CAP)] = k0;
        row[lane_select(m1, mbcnt_add(m1, n0), KQ_CAP)] = k1;
        if (owner == qq) cnt = total;
      }
    }
  }
  }
  wave_lds_fence();
  if (stats) { const long long t = clock64(); tB += t - tm, tm = t; }
  if (stats && lane == 0) {
    atomicAdd(stats + 4, (unsigned long long)n_groups), atomicAdd(stats + 7, 1ull);
    atomicAdd(stats + 8, (unsigned long long)n_compact), atomicAdd(stats + 9, (unsigned long long)n_pairs);
  }
  if (!valid) return;  // whole groups leave together: the butterflies below never see a missing partner

  // ---- C
  // Every lane sorts its EPL entries with a fixed network (optimal ones for 3 / 6 / 12 keys: 3 / 12 / 39 compare-exchanges),
  // the sorted sub-lists go back into the row, and the k neighbours come out of an L-way merge: per round one butterfly
  // over the heads and, in the lane that held the winner, the next entry of its list (requested one round ahead).  A
  // round used to scan all EPL register entries of every lane: 86 instructions against 20.
  unsigned long long e[EPL];
  unsigned long long* row = lst + slot * KQ_STRIDE;
#pragma unroll
  for (int t = 0; t < EPL; t++) {
    const int a = sub + L * t;
    e[t] = a < cnt ? row[a] : kKeyInf;
  }
#ifdef APD_ABL_KNN_SKIP_C  // ABLATION builds only (wrong results by design): the first k list entries stand in for the k nearest -- what the
                           // launch would cost without phase C's sorting network and merge rounds (docs/experiments.md, round 5)
  int mysel[KNN_NC / L];
#pragma unroll
  for (int t = 0; t < KNN_NC / L; t++) mysel[t] = (sub + L * t < k && e[t] != kKeyInf) ? (int)(unsigned)e[t] : i;
#else
  {
    auto cx = [&](int a, int b) {
      const unsigned long long x = e[a], y = e[b];
      e[a] = key_min(x, y), e[b] = key_max(x, y);
    };
    if constexpr (EPL == 3) {
      cx(0, 2), cx(0, 1), cx(1, 2);
    } else if constexpr (EPL == 6) {
      cx(0, 5), cx(1, 3), cx(2, 4), cx(1, 2), cx(3, 4), cx(0, 3), cx(2, 5), cx(0, 1), cx(2, 3), cx(4, 5), cx(1, 2), cx(3, 4);
    } else if constexpr (EPL != 12) {
      // any other length: Batcher's merge-exchange network for the next power of two, the comparators that would touch an
      // element >= EPL left out (those elements are +infinity and stay where they are); 15 keys: 59 compare-exchanges
      constexpr int P = EPL <= 4 ? 4 : EPL <= 8 ? 8 : EPL <= 16 ? 16 : 32;
#pragma unroll
      for (int p_ = 1; p_ < P; p_ <<= 1)
#pragma unroll
        for (int k_ = p_; k_ >= 1; k_ >>= 1)
#pragma unroll
          for (int j_ = k_ % p_; j_ + k_ < P; j_ += 2 * k_)
#pragma unroll
            for (int i_ = 0; i_ < k_; i_++)
              if ((i_ + j_) / (2 * p_) == (i_ + j_ + k_) / (2 * p_) && i_ + j_ + k_ < EPL) cx(i_ + j_, i_ + j_ + k_);
    } else {
      cx(0, 8), cx(1, 7), cx(2, 6), cx(3, 11), cx(4, 10), cx(5, 9), cx(0, 1), cx(2, 5), cx(3, 4), cx(6, 9), cx(7, 8), cx(10, 11), cx(0, 2);
      cx(1, 6), cx(5, 10), cx(9, 11), cx(0, 3), cx(1, 2), cx(4, 6), cx(5, 7), cx(8, 11), cx(9, 10), cx(1, 4), cx(3, 5), cx(6, 8), cx(7, 10);
      cx(1, 3), cx(2, 5), cx(6, 9), cx(8, 10), cx(2, 3), cx(4, 5), cx(6, 7), cx(8, 9), cx(4, 6), cx(5, 7), cx(3, 4), cx(5, 6), cx(7, 8);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (every lane of the wave has read its entries)
  unsigned long long* mine = row + sub * EPL;
#pragma unroll
  for (int t = 2; t < EPL; t++) mine[t] = e[t];  // (the first two stay in registers: head and next)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  int mysel[KNN_NC / L];  // the neighbours of rank sub, sub + L, ...: the ones this lane gathers
#pragma unroll
  for (int t = 0; t < KNN_NC / L; t++) mysel[t] = i;
  {
    unsigned long long head = e[0], next = e[1];
    int cpos = 1;  // position of `next` in this lane's list
#pragma unroll
    for (int t = 0; t < KNN_NC / L; t++) {
#pragma unroll
      for (int u = 0; u < L; u++) {
        if (t * L + u < k) {  // (uniform)
          unsigned long long bk = group_min_u64<L>(head);
          if (bk == kKeyInf) {
            atomicExch(err_flag, 2);
            bk = (unsigned long long)(unsigned)i;  // keeps the gathers in range
          }
          if (sub == u) mysel[t] = (int)(unsigned)bk;
          if (head == bk) {  // keys are unique: one lane of the query
            head = next;
            cpos++;
            next = cpos < EPL ? mine[cpos] : kKeyInf;
          }
        }
      }
    }
  }
#endif  // APD_ABL_KNN_SKIP_C
  // gathers: lane sub fetches the neighbours of rank sub, sub+L, ... -- all loads in flight together
  float4 nbv[KNN_NC / L];
#pragma unroll
  for (int t = 0; t < KNN_NC / L; t++) {
    const int r = sub + L * t;
    nbv[t] = q;
    if (r < k) nbv[t] = G(c.opts)[mysel[t]];
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  float* nbl = (float*)row;  // [rank][xyz], overwrites the sorted lists (already consumed)
#pragma unroll
  for (int t = 0; t < KNN_NC / L; t++) {
    const int r = sub + L * t;
    if (r < k) nbl[3 * r] = nbv[t].x, nbl[3 * r + 1] = nbv[t].y, nbl[3 * r + 2] = nbv[t].z;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  // moments (Mom9): lanes 0..3 of the query take one partial sum each, two quad butterflies add them up in the common order
  // (the operands of an addition swap between partner lanes: the same bits)
  Mom9 mp;
  if (sub < 4)
    for (int r = sub; r < k; r += 4) mp.add((double)nbl[3 * r] - (double)q.x, (double)nbl[3 * r + 1] - (double)q.y, (double)nbl[3 * r + 2] - (double)q.z);
  mp.merge(mom_dpp<0xB1>(mp));  // quad_perm [1,0,3,2]: p0 + p1 | p2 + p3
  mp.merge(mom_dpp<0x4E>(mp));  // quad_perm [2,3,0,1]: (p0 + p1) + (p2 + p3)
  const double s1x = mp.s1x, s1y = mp.s1y, s1z = mp.s1z, sxx = mp.sxx, sxy = mp.sxy, sxz = mp.sxz, syy = mp.syy, syz = mp.syz, szz = mp.szz;
  if (stats && lane == 0) { tC = clock64() - tm; atomicAdd(stats + 10, (unsigned long long)tA), atomicAdd(stats + 11, (unsigned long long)tG), atomicAdd(stats + 12, (unsigned long long)tB), atomicAdd(stats + 13, (unsigned long long)tC); }
  if (sub != 0) return;
  const double ik = 1.0 / (double)k;
  const double mx = s1x * ik, my = s1y * ik, mz = s1z * ik;
  Sym3 pc;
  pc.xx = sxx * ik - mx * mx, pc.xy = sxy * ik - mx * my, pc.xz = sxz * ik - mx * mz;
  pc.yy = syy * ik - my * my, pc.yz = syz * ik - my * mz, pc.zz = szz * ik - mz * mz;
  // Throughput launches (L lanes per query, so only one lane in L would work here -- and the Jacobi sweeps are a quarter of
  // this kernel's fp64-weighted instructions) leave the regularisation to k_regularize_covs, one lane per point.
  Sym3 out = pc;
  if (!raw && !regularize_cov(reg, pc, out)) atomicExch(err_flag, 3);
  if (stats && lane == 0) atomicAdd(stats + 15, (unsigned long long)(clock64() - tm));  // rounds + regularisation
  auto cov = GW(c.cov);
  cov[i] = out.xx, cov[n + i] = out.xy, cov[2 * n + i] = out.xz, cov[3 * n + i] = out.yy, cov[4 * n + i] = out.yz, cov[5 * n + i] = out.zz;
}

template <int L>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(APD_KNN_WPE, 8))) void k_knn_cov_coop(const CloudDesc* clouds, const int* cloud_ids, int k, int reg, int* err_flag,
                                                     unsigned long long* stats, int raw) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long knn_smem[];
  unsigned bx, by;
  xcd_remap(bx, by);
  const CloudDesc c = clouds[cloud_ids[by]];
  knn_cov_coop_wave<L>(c, bx, (int)threadIdx.x, knn_smem, k, reg, err_flag, stats, raw);
}

// k_knn_and_search: the covariance k-NN of one or two freshly set clouds AND the first, cold search of a single registration in ONE
// launch.  The search needs the sorted clouds, not their covariances (those are first read by k_linearize behind it), so the two
// are independent -- but a HIP stream runs them one after the other (hipExtAnyOrderLaunch is not supported on gfx9 boards, and a
// second stream costs more in event hand-overs than it hides): 42 + 21.5 us of a 131 us odometry frame.  Blocks of 512 threads:
// the first knn_blocks run eight k_knn_cov_coop<16> waves each (wave-task t = block * 8 + wave: cloud t / tasks_per_cloud,
// queries 4 (t % tasks_per_cloud) ...; the wave index must be wave-uniform -- readfirstlane -- or the cloud descriptor becomes per-lane
// data and the compiler's divergent code around the cross-lane operations goes wrong), the others are the blocks of k_nn_pruned<1, 8>.
// Same device functions, same results: 42.3 + 21.5 us -> 44.3 us, the odometry frame 131 -> 118 us.
__global__ __launch_bounds__(512) void k_knn_and_search(const CloudDesc* clouds, const int* cloud_ids, int knn_count, int tasks_per_cloud, int k, int reg,
                                                        int* err_flag, unsigned long long* stats, const PairDesc* pairs, const PairState* st, Work w) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long knn_smem[];
  constexpr unsigned wpb = 8;
  const unsigned knn_blocks = ((unsigned)(knn_count * tasks_per_cloud) + wpb - 1u) / wpb;
  if (blockIdx.x < knn_blocks) {
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = (int)(threadIdx.x & 63);  // (wave-uniform: the descriptor loads stay scalar)
    const unsigned task = blockIdx.x * wpb + (unsigned)wid;
    const unsigned by = task / (unsigned)tasks_per_cloud, bx = task % (unsigned)tasks_per_cloud;
    if ((int)by >= knn_count) return;  // (a wave that has ended does not count at the block's barriers)
    knn_cov_coop_wave<16>(clouds[cloud_ids[by]], bx, lane, knn_smem + wid * (knn_coop_lds_bytes(4) / 8), k, reg, err_flag, stats, 0);
  } else {
    nn_pruned_block<1, 8>(pairs, st, w, blockIdx.x - knn_blocks, 0u);
  }
}

// fast_apdgicp_impl.hpp:326-357 for covariances stored raw by k_knn_cov_coop: one lane per point, in place (the same function
// on the same numbers as the fused path: bitwise the same result).  grid (ceil(nmax / 256), clouds)
__global__ __launch_bounds__(256) void k_regularize_covs(const CloudDesc* clouds, const int* cloud_ids, int reg, int* err_flag) {
  const CloudDesc c = clouds[cloud_ids[blockIdx.y]];
  const int n = c.n, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  auto cov = GW(c.cov);
  const Sym3 pc{cov[i], cov[n + i], cov[2 * n + i], cov[3 * n + i], cov[4 * n + i], cov[5 * n + i]};
  Sym3 out;
  if (!regularize_cov(reg, pc, out)) atomicExch(err_flag, 3);
  cov[i] = out.xx, cov[n + i] = out.xy, cov[2 * n + i] = out.xz, cov[3 * n + i] = out.yy, cov[4 * n + i] = out.yz, cov[5 * n + i] = out.zz;
}

// ----------------------------------------------------------------------------------------------
// block reduction of R doubles per lane: DPP wave reduction (no LDS traffic), then LDS across the
// block's waves.  Fixed order -> bitwise reproducible (the reference's per-thread slots are not, A:262-269).
//
// wave_sum_to_lane63: inclusive row scans (row_shr 1,2,4,8 inside each 16-lane row, out-of-row sources
// read as 0), then row_bcast:15 into rows 1,3 and row_bcast:31 into rows 2,3 -- the classic gfx9
// cross-lane reduction; lane 63 ends up with the sum of all 64 lanes.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_fetch(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);  // lanes without a source (or masked rows) receive +0.0
}
__device__ __forceinline__ double wave_sum_to_lane63(double v) {
  v += dpp_fetch<0x111, 0xf>(v);  // row_shr:1
  v += dpp_fetch<0x112, 0xf>(v);  // row_shr:2
  v += dpp_fetch<0x114, 0xf>(v);  // row_shr:4
  v += dpp_fetch<0x118, 0xf>(v);  // row_shr:8   -> lane 15 of every row holds the row sum
  v += dpp_fetch<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
  v += dpp_fetch<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
  return v;
}

template <int R, int BLK>
__device__ __forceinline__ void block_reduce(double* v, double* lds /* [BLK/64][R] */, int tid) {
#pragma unroll
  for (int r = 0; r < R; r++) v[r] = wave_sum_to_lane63(v[r]);
  const int wave = tid >> 6, lane = tid & 63;
  if (lane == 63) {
#pragma unroll
    for (int r = 0; r < R; r++) lds[wave * R + r] = v[r];
  }
  __syncthreads();
}

// The same sums for many values per lane (R = 29 in k_linearize) through an LDS transpose instead of six DPP steps per value
// (3 instructions each: 29 x 6 x 3 = 522 per wave, a fifth of that kernel).  16 values at a time: every lane stores its
// 16 values ([value][lane], padded rows), lane (q, c) adds the 16 lanes' values of quarter q of column c in lane order, the
// four quarter sums of a column are added in quarter order.  Fixed order -> bitwise reproducible, like the DPP tree.
// scratch: BLK/64 x RED_LDS_WAVE doubles.
constexpr int RED_LDS_COLS = 8, RED_LDS_STRIDE = 65, RED_LDS_WAVE = RED_LDS_COLS * RED_LDS_STRIDE + 64;  // 8 values per pass: 4.7 KB per wave
template <int R, int BLK, typename Gen>
__device__ __forceinline__ void block_reduce_lds(Gen&& gen /* gen(r): this lane's value of sum r */, double* lds /* [BLK/64][R] */, double* scratch,
                                                 int tid) {
  const int wave = tid >> 6, lane = tid & 63;
  double* buf = scratch + wave * RED_LDS_WAVE;
  double* part = buf + RED_LDS_COLS * RED_LDS_STRIDE;
  const int col = lane & 15, rb = (lane / 16) * 16;  // quarter q = lane / 16 adds lanes 16q .. 16q + 15 of column lane & 15 (< nr)
#pragma unroll
  for (int r0 = 0; r0 < R; r0 += RED_LDS_COLS) {
    const int nr = R - r0 < RED_LDS_COLS ? R - r0 : RED_LDS_COLS;
#pragma unroll
    for (int u = 0; u < RED_LDS_COLS; u++)
      if (u < nr) buf[u * RED_LDS_STRIDE + lane] = gen(r0 + u);
    wave_lds_fence();
    double p = 0.0;
    if (col < nr) {
#pragma unroll
      for (int j = 0; j < 16; j++) p += buf[col * RED_LDS_STRIDE + rb + j];
    }
    part[lane] = p;
    wave_lds_fence();
    if (lane < nr) lds[wave * R + r0 + lane] = ((part[lane] + part[lane + 16]) + part[lane + 32]) + part[lane + 48];
    wave_lds_fence();
  }
  __syncthreads();
}

// ---- fp64 contraction region.  The whole library is compiled -ffp-contract=off because the fp32 nearest-neighbour arithmetic must
// not be fused (A:149-153: the reference's build has no FMA).  The fp64 algebra behind the search -- RCR, its inverse, e, J^T M J,
// compute_error -- is compared with the reference by tolerance (5e-6, set by the fp32 atan2f), and there a*b + c as ONE rounding
// instead of two moves results by ~1e-16 relative while a third of k_linearize's fp64 instructions disappear (v_mul_f64 + v_add_f64
// -> v_fma_f64: 961 -> ~800 vector instructions per wave).  Everything fp32 inside stays in explicit contract(off) blocks; the
// covariance kernels (bitwise equal across their four variants) and apd_math.hpp stay unfused.
#pragma clang fp contract(fast)
__device__ __forceinline__ Sym3 sym3_rotate_c(const Rigid& T, const Sym3& c) {  // sym3_rotate (apd_math.hpp), contracted
  double rc[9];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const double a = T.m[4 * i], b = T.m[4 * i + 1], d = T.m[4 * i + 2];
    rc[3 * i + 0] = a * c.xx + b * c.xy + d * c.xz;
    rc[3 * i + 1] = a * c.xy + b * c.yy + d * c.yz;
    rc[3 * i + 2] = a * c.xz + b * c.yz + d * c.zz;
  }
  Sym3 o;
  o.xx = rc[0] * T.m[0] + rc[1] * T.m[1] + rc[2] * T.m[2];
  o.xy = rc[0] * T.m[4] + rc[1] * T.m[5] + rc[2] * T.m[6];
  o.xz = rc[0] * T.m[8] + rc[1] * T.m[9] + rc[2] * T.m[10];
  o.yy = rc[3] * T.m[4] + rc[4] * T.m[5] + rc[5] * T.m[6];
  o.yz = rc[3] * T.m[8] + rc[4] * T.m[9] + rc[5] * T.m[10];
  o.zz = rc[6] * T.m[8] + rc[7] * T.m[9] + rc[8] * T.m[10];
  return o;
}
__device__ __forceinline__ Sym3 sym3_inverse_c(const Sym3& a) {  // sym3_inverse (apd_math.hpp), contracted
  const double c00 = a.yy * a.zz - a.yz * a.yz;
  const double c01 = a.yz * a.xz - a.xy * a.zz;
  const double c02 = a.xy * a.yz - a.yy * a.xz;
  const double det = a.xx * c00 + a.xy * c01 + a.xz * c02;
  const double id = 1.0 / det;
  Sym3 r;
  r.xx = c00 * id;
  r.xy = c01 * id;
  r.xz = c02 * id;
  r.yy = (a.xx * a.zz - a.xz * a.xz) * id;
  r.yz = (a.xy * a.xz - a.xx * a.yz) * id;
  r.zz = (a.xx * a.yy - a.xy * a.xy) * id;
  return r;
}

// the interval table of apd_atan2f (include/apd_atan2f.h) into the block's LDS: the first 40 threads, before a barrier
__device__ __forceinline__ void atan_tab_to_lds(float* lds, int tid) {
  static constexpr float init[APD_ATAN_TAB_ROWS * APD_ATAN_TAB_STRIDE] = APD_ATAN_TAB_INIT;
  if (tid < APD_ATAN_TAB_ROWS * APD_ATAN_TAB_STRIDE) lds[tid] = init[tid];
}

// what one source point contributes to linearize (A:229-258); all zero without a correspondence
struct LinPoint {
  Sym3 Mi;                    // RCR^-1, A:191
  double vx, vy, vz;          // transed_mean_A
  double mex, mey, mez, cost; // M e, e^T M e
  double matched;
};

// The per-point part of update_correspondences + linearize (A:137-258) once the 1-NN search has produced the
// minimum distance m and the chunk it was found in: exact index, gate, APD covariance, RCR^-1, e, J, H, b.
// lp receives this point's quantities (left all zero when it has no correspondence); lin_term turns them into the 29 sums.
// reciprocal square root of a positive double: the hardware's estimate (v_rsq_f64) and two Newton steps (~1e-16 relative); +inf for 0
__device__ __forceinline__ double rsqrt_nr(double x) {
  const double h = 0.5 * x;
  double y = __builtin_amdgcn_rsq(x);
  y = y * (1.5 - h * y * y);
  y = y * (1.5 - h * y * y);
  return x > 0.0 ? y : __builtin_inf();
}

// ALG: APDGICP_FLAG_ALGEBRAIC_APD -- the sensor model's ratios straight from the point's coordinates (see include/apdgicp_hip.h)
template <bool ALG = false>
__device__ __forceinline__ void linearize_point(const CloudDesc& src, const CloudDesc& tgt, const Rigid& T, const Work& w, const Consts& cst,
                                                int pair, int i, const float4 p, float ptx, float pty, float ptz, float m, unsigned chunk,
                                                bool tie, bool kept, const float4 tq_rec /* nnpt[i], read by the caller */,
                                                const Sym3& cov_A /* the point's covariance, read by the caller */, LinPoint& lp,
                                                const float* atan_tab /* the block's LDS copy of apd_atan2f's interval table */) {
  const int M = tgt.n;
  int j = -1;
  float4 tq = make_float4(0.f, 0.f, 0.f, 0.f);  // the neighbour itself
  if (kept) {  // the search proved the previous neighbour still is the nearest one (kKeptBit): index and point are on record
    tq = tq_rec;
    j = __float_as_int(tq.w);
  } else if (chunk != kNoChunk) {
    // exact index: among the targets at distance m, the one with the lowest ORIGINAL index
    int jorig = 0x7fffffff;
    if (!tie) {  // (only results of the brute-force search get here: the pruned search resolves its own, see nn_search)
#pragma unroll 1
      for (int h = 0; h < kChunk; h += 4) {
        float4 t[4];
#pragma unroll
        for (int jj = 0; jj < 4; jj++) t[jj] = tgt.pts[min((int)chunk * kChunk + h + jj, M - 1)];
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
          const int g = (int)chunk * kChunk + h + jj;
          const float d = sqdist1(t[jj].x, t[jj].y, t[jj].z, ptx, pty, ptz);
          const int po = __float_as_int(t[jj].w);  // (the sorted points carry their original index in .w)
          if (g < M && d == m && po < jorig) jorig = po, j = g, tq = t[jj];
        }
      }
    } else {  // rare: the same fp32 minimum in several chunks (duplicates / exact ties): look at every target
      for (int g = 0; g < M; g++) {
        const float4 t = tgt.pts[g];
        if (sqdist1(t.x, t.y, t.z, ptx, pty, ptz) == m) {
          const int o = __float_as_int(t.w);
          if (o < jorig) jorig = o, j = g, tq = t;
        }
      }
    }
  }
  if (w.sqd) w.sqd[(size_t)pair * w.nstride + i] = m;  // (sqd, corr: null in a Gauss-Newton batch, like maha)
  if (!kept) w.nnpt[(size_t)pair * w.nstride + i] = make_float4(tq.x, tq.y, tq.z, __int_as_float(j));
  const int corr = (j >= 0 && (double)m < cst.thr2) ? j : -1;  // A:156
  if (w.corr) w.corr[(size_t)pair * w.nstride + i] = corr;
  if (corr >= 0) {
    const auto cb = G((const double*)tgt.cov);
    const Sym3 cov_B{cb[corr], cb[M + corr], cb[2 * M + corr], cb[3 * M + corr], cb[4 * M + corr], cb[5 * M + corr]};
    // APD sensor-noise covariance from the transformed point (A:167-184)
    double s_x, s_y, s_z, ce, se, caz, saz;
    if constexpr (ALG) {
      // no angles: every quantity the model takes from them is a ratio of the point's coordinates (fp64, from the fp32 point)
      const double x = (double)ptx, y = (double)pty, z = (double)ptz;
      const double xx = x * x, yy = y * y, zz = z * z;
      const double rho2 = xx + yy, yz2 = yy + zz, r2 = rho2 + zz;
      const double ir = r2 > 0.0 ? rsqrt_nr(r2) : 0.0;            // the origin: dist = 0, the model vanishes
      const double irho = rho2 > 0.0 ? rsqrt_nr(rho2) : 0.0;      // on the z axis: azimuth = atan2(0, 0) = 0
      const double dist = r2 * ir;
      // 1 / cos(AoA) = r / sqrt(y^2 + z^2), at most 1 / |cos((double)(float)(pi/2))|: what the reference's fp32 angle can give at most
      const double inv_cos = fmin(dist * rsqrt_nr(yz2), 1.0 / 4.371138828673793e-08);
      const double dist_c = dist * inv_cos;
      s_x = dist * cst.dist_var_400;
      s_y = dist_c * cst.sin_az;
      s_z = dist_c * cst.sin_el;
      se = rho2 * irho * ir, ce = z * ir;
      caz = rho2 > 0.0 ? x * irho : 1.0, saz = y * irho;
    } else {
    const double dist = sqrt((double)ptx * (double)ptx + (double)pty * (double)pty + (double)ptz * (double)ptz);
    double aoa, elevation, azimuth;
    {  // fp32, as the reference evaluates it (float overloads, no FMA): not contracted.  apd_atan2f is glibc's generic atan2f
       // (fdlibm, include/apd_atan2f.h) restated: the device library's own atan2f is another ~1 ulp implementation
#pragma clang fp contract(off)
#if defined(APD_ABL_LIN_NO_ATAN)  // ABLATION builds only (wrong results by design): what the three atan2f cost
#define APD_ATAN2F(y, x) ((y) * 0.01f + (x) * 0.001f)
#elif defined(APD_OCML_ATAN2F)  // (A/B builds only, tools/ab_bench.sh: what the restated atan2f costs against the device library's)
#define APD_ATAN2F(y, x) atan2f(y, x)
#else
#define APD_ATAN2F(y, x) apd_atan2f_tab(y, x, atan_tab)
#endif
      aoa = (double)APD_ATAN2F(ptx, sqrtf(pty * pty + ptz * ptz));
      elevation = (double)APD_ATAN2F(sqrtf(ptx * ptx + pty * pty), ptz);
      azimuth = (double)APD_ATAN2F(pty, ptx);
#ifdef APD_ABL_LIN_DOUBLE_ATAN  // ABLATION builds only: the three atan2f evaluated a second time on other arguments, results never used (the
                                // compiler cannot know): what they cost without changing a single result
      const float a2 = APD_ATAN2F(ptx + 1.f, sqrtf(pty * pty + ptz * ptz)), e2 = APD_ATAN2F(sqrtf(ptx * ptx + pty * pty), ptz + 1.f), z2 = APD_ATAN2F(pty, ptx + 1.f);
      if (a2 == 123.f) aoa = a2;
      if (e2 == 123.f) elevation = e2;
      if (z2 == 123.f) azimuth = z2;
#endif
    }
    double sin_aoa, cos_aoa;
#ifdef APD_ABL_LIN_NO_SINCOS  // ABLATION builds only (wrong results by design): what the three fp64 sin / cos pairs cost
#define APD_SINCOS(x, s, c) (*(s) = (x), *(c) = 1.0 - 0.5 * (x) * (x))
#else
#define APD_SINCOS(x, s, c) sincos_pi(x, s, c)
#endif
    APD_SINCOS(aoa, &sin_aoa, &cos_aoa);
    (void)sin_aoa;
    // (A:169-171 divide three times -- dist * dist_var / 400, dist * sin(az) / cos(aoa), dist * sin(el) / cos(aoa); here one
    // division, dist / cos(aoa), and dist_var / 400 from the host: a rounding apart, like the contraction, 22 fp64 instructions less)
    s_x = dist * cst.dist_var_400;
    const double dist_c = dist / cos_aoa;
    s_y = dist_c * cst.sin_az;
    s_z = dist_c * cst.sin_el;
#if defined(APD_ABL_LIN_NO_SINCOS) || defined(APD_SINCOS_NO_TABLE)  // (APD_SINCOS_NO_TABLE: A/B builds -- sincos_pi for all three angles, as until round 5)
    APD_SINCOS(elevation, &se, &ce);
    APD_SINCOS(azimuth, &saz, &caz);
#else
    sincos_tab(elevation, &se, &ce);  // rotation-matrix entries: from the table (apd_math.hpp); cos(AoA) above is a divisor and is not
    sincos_tab(azimuth, &saz, &caz);
#endif
#ifdef APD_ABL_LIN_DOUBLE_SINCOS  // ABLATION builds only: three more fp64 sin / cos pairs whose results are never used
    {
      double s2, c2, s3, c3, s4, c4;
      sincos_pi(elevation * 0.5, &s2, &c2), sincos_pi(azimuth * 0.5, &s3, &c3), sincos_pi(aoa * 0.5, &s4, &c4);
      if (s2 + c2 + s3 + c3 + s4 + c4 == 123.0) se = s2;
    }
#endif
    }  // (!ALG)
    // A = (Rz(azimuth) * Ry(elevation)) * diag(s)
    const double a00 = caz * ce * s_x, a01 = -saz * s_y, a02 = caz * se * s_z;
    const double a10 = saz * ce * s_x, a11 = caz * s_y, a12 = saz * se * s_z;
    const double a20 = -se * s_x, a22 = ce * s_z;  // a21 = 0
    Sym3 cd;
    cd.xx = a00 * a00 + a01 * a01 + a02 * a02;
    cd.xy = a00 * a10 + a01 * a11 + a02 * a12;
    cd.xz = a00 * a20 + a02 * a22;
    cd.yy = a10 * a10 + a11 * a11 + a12 * a12;
    cd.yz = a10 * a20 + a12 * a22;
    cd.zz = a20 * a20 + a22 * a22;
    if (cst.plain_gicp) cd = Sym3{0.0, 0.0, 0.0, 0.0, 0.0, 0.0};  // fast_gicp_impl.hpp: RCR = cov_B + T cov_A T^T
    const Sym3 RCR = sym3_add(sym3_add(cov_B, cd), sym3_rotate_c(T, sym3_add(cov_A, cd)));  // A:188
    const Sym3 Mi = sym3_inverse_c(RCR);                                                      // A:191
    if (w.maha) {  // (null in a Gauss-Newton batch: only compute_error and the Mahalanobis getter of a single handle read it)
      double* mo = w.maha + (size_t)pair * 6 * w.nstride + i;
      mo[0] = Mi.xx, mo[w.nstride] = Mi.xy, mo[2 * (size_t)w.nstride] = Mi.xz;
      mo[3 * (size_t)w.nstride] = Mi.yy, mo[4 * (size_t)w.nstride] = Mi.yz, mo[5 * (size_t)w.nstride] = Mi.zz;
    }

    const float4 q = tq;  // corr == j here
    const double ax = (double)p.x, ay = (double)p.y, az = (double)p.z;
    const double vx = T.m[0] * ax + T.m[1] * ay + T.m[2] * az + T.m[3];   // transed_mean_A, A:236
    const double vy = T.m[4] * ax + T.m[5] * ay + T.m[6] * az + T.m[7];
    const double vz = T.m[8] * ax + T.m[9] * ay + T.m[10] * az + T.m[11];
    const double ex = (double)q.x - vx, ey = (double)q.y - vy, ez = (double)q.z - vz;  // A:237
    lp.Mi = Mi;
    lp.vx = vx, lp.vy = vy, lp.vz = vz;
    lp.mex = Mi.xx * ex + Mi.xy * ey + Mi.xz * ez;
    lp.mey = Mi.xy * ex + Mi.yy * ey + Mi.yz * ez;
    lp.mez = Mi.xz * ex + Mi.yz * ey + Mi.zz * ez;
    lp.cost = ex * lp.mex + ey * lp.mey + ez * lp.mez;  // A:240
    lp.matched = 1.0;
  }
}

// This point's contribution to sum r of the 29 (21 H upper triangle, row-major; 6 b; cost; matched), from the per-point
// quantities above.  Written as a function of r so that the block reduction can produce the sums a few at a time: 29 live
// fp64 accumulators cost the kernel a wave per SIMD.  J = [skew(v) | -I] (A:248-250), MA = M * skew(v).
__device__ __forceinline__ double lin_term(const LinPoint& lp, int want_Hb, int r) {
  const Sym3& Mi = lp.Mi;
  const double vx = lp.vx, vy = lp.vy, vz = lp.vz, mex = lp.mex, mey = lp.mey, mez = lp.mez;
  if (r == 27) return lp.cost;
  if (r == 28) return lp.matched;
  if (!want_Hb) return 0.0;
  const double m0x = Mi.xy * vz - Mi.xz * vy, m0y = Mi.yy * vz - Mi.yz * vy, m0z = Mi.yz * vz - Mi.zz * vy;     // MA[:,0]
  const double m1x = -Mi.xx * vz + Mi.xz * vx, m1y = -Mi.xy * vz + Mi.yz * vx, m1z = -Mi.xz * vz + Mi.zz * vx;  // MA[:,1]
  const double m2x = Mi.xx * vy - Mi.xy * vx, m2y = Mi.xy * vy - Mi.yy * vx, m2z = Mi.xz * vy - Mi.yz * vx;     // MA[:,2]
  switch (r) {
    // rotation block skew^T M skew: row p = skew[:,p] . MA[:,q]; top-right block = -MA^T
    case 0: return vz * m0y - vy * m0z;    // (0,0)
    case 1: return vz * m1y - vy * m1z;    // (0,1)
    case 2: return vz * m2y - vy * m2z;    // (0,2)
    case 3: return -m0x;                   // (0,3)
    case 4: return -m0y;                   // (0,4)
    case 5: return -m0z;                   // (0,5)
    case 6: return -vz * m1x + vx * m1z;   // (1,1)
    case 7: return -vz * m2x + vx * m2z;   // (1,2)
    case 8: return -m1x;                   // (1,3)
    case 9: return -m1y;                   // (1,4)
    case 10: return -m1z;                  // (1,5)
    case 11: return vy * m2x - vx * m2y;   // (2,2)
    case 12: return -m2x;                  // (2,3)
    case 13: return -m2y;                  // (2,4)
    case 14: return -m2z;                  // (2,5)
    case 15: return Mi.xx;                 // (3,3)
    case 16: return Mi.xy;
    case 17: return Mi.xz;
    case 18: return Mi.yy;                 // (4,4)
    case 19: return Mi.yz;
    case 20: return Mi.zz;                 // (5,5)
    // b = J^T M e : rotation part skew^T (Me), translation part -(Me)   (A:254)
    case 21: return vz * mey - vy * mez;
    case 22: return -vz * mex + vx * mez;
    case 23: return vy * mex - vx * mey;
    case 24: return -mex;
    case 25: return -mey;
    default: return -mez;
  }
}

// ---- APDGICP_FLAG_FP32_POINT_MATH (opt-in): the same per-point algebra in fp32.  SURVEY.md 7 (hard part 5) measured what the
// reference's fp64 buys behind the fp32 search: nothing the tolerance can see (poses move by ~1e-6 m) -- except in 1 / cos(AoA),
// which cancels near the +-x axis, in the sums over the points and in the 6x6 solve.  So: the three fp32 angles as always; cos(AoA)
// and its reciprocal in fp64; sin / cos of elevation and azimuth, A A^T, RCR, its inverse, e, M e and every J^T M J term in fp32
// (contracted); each point's 29 terms are widened to fp64 before the block reduction, which, like the optimiser step, is unchanged.
__device__ __forceinline__ void sincos_pi_f32(float x, float* so, float* co) {  // |x| <= pi (an atan2f result): Cody-Waite in two steps, minimax kernels on [-pi/4, pi/4]
  const float fn = rintf(x * 6.36619747e-01f);
  float r = fmaf(-fn, 1.57079637e+00f, x);   // pi/2 = 0x1.921fb6p+0 - 4.37113900e-8 (fn in -2 .. 2: both products are exact)
  r = fmaf(-fn, -4.37113900e-08f, r);
  const float z = r * r;                      // (cephes sinf / cosf kernels: 7e-8 absolute over [-pi, pi])
  const float s = fmaf(r * z, fmaf(z, fmaf(z, -1.95152959e-04f, 8.33216087e-03f), -1.66666546e-01f), r);
  const float c = fmaf(z, fmaf(z, fmaf(z, fmaf(z, 2.44331571e-05f, -1.38873163e-03f), 4.16666457e-02f), -0.5f), 1.0f);
  const int n = (int)fn;
  const float sv = (n & 1) ? c : s, cv = (n & 1) ? s : c;
  *so = (n & 2) ? -sv : sv;
  *co = ((n + 1) & 2) ? -cv : cv;
}
struct Sym3f {
  float xx, xy, xz, yy, yz, zz;
};
struct LinPointF {
  Sym3f Mi;
  float vx, vy, vz, mex, mey, mez, cost, matched;
};
__device__ __forceinline__ void linearize_point_f32(const CloudDesc& src, const CloudDesc& tgt, const Rigid& T, const Work& w, const Consts& cst, int pair, int i,
                                                    const float4 p, float ptx, float pty, float ptz, float m, unsigned chunk, bool tie, bool kept,
                                                    const float4 tq_rec, const Sym3& cov_A, LinPointF& lp, const float* atan_tab) {
  const int M = tgt.n;
  int j = -1;
  float4 tq = make_float4(0.f, 0.f, 0.f, 0.f);
  if (kept) {
    tq = tq_rec;
    j = __float_as_int(tq.w);
  } else if (chunk != kNoChunk) {  // (as linearize_point: only brute-force search results reach the re-scan)
    int jorig = 0x7fffffff;
    const int g0 = tie ? 0 : (int)chunk * kChunk, g1 = tie ? M : min(g0 + kChunk, M);
    for (int g = g0; g < g1; g++) {
      const float4 t = tgt.pts[g];
      if (sqdist1(t.x, t.y, t.z, ptx, pty, ptz) == m) {
        const int o = __float_as_int(t.w);
        if (o < jorig) jorig = o, j = g, tq = t;
      }
    }
  }
  if (w.sqd) w.sqd[(size_t)pair * w.nstride + i] = m;
  if (!kept) w.nnpt[(size_t)pair * w.nstride + i] = make_float4(tq.x, tq.y, tq.z, __int_as_float(j));
  const int corr = (j >= 0 && (double)m < cst.thr2) ? j : -1;  // A:156
  if (w.corr) w.corr[(size_t)pair * w.nstride + i] = corr;
  if (corr < 0) return;
  const auto cb = G((const double*)tgt.cov);
  const Sym3f cB{(float)cb[corr], (float)cb[M + corr], (float)cb[2 * M + corr], (float)cb[3 * M + corr], (float)cb[4 * M + corr], (float)cb[5 * M + corr]};
  const Sym3f cA{(float)cov_A.xx, (float)cov_A.xy, (float)cov_A.xz, (float)cov_A.yy, (float)cov_A.yz, (float)cov_A.zz};
  float aoa_f, el_f, az_f;
  {
#pragma clang fp contract(off)
    aoa_f = apd_atan2f_tab(ptx, sqrtf(pty * pty + ptz * ptz), atan_tab);
    el_f = apd_atan2f_tab(sqrtf(ptx * ptx + pty * pty), ptz, atan_tab);
    az_f = apd_atan2f_tab(pty, ptx, atan_tab);
  }
  double sin_aoa, cos_aoa;
  sincos_pi((double)aoa_f, &sin_aoa, &cos_aoa);  // fp64: cos(AoA) cancels near the +-x axis, and its reciprocal scales two of the three sigmas
  (void)sin_aoa;
  const float dist = sqrtf(ptx * ptx + pty * pty + ptz * ptz);
  const float dist_c = (float)((double)dist / cos_aoa);
  const float s_x = dist * (float)cst.dist_var_400, s_y = dist_c * (float)cst.sin_az, s_z = dist_c * (float)cst.sin_el;
  float ce, se, caz, saz;
  sincos_pi_f32(el_f, &se, &ce);
  sincos_pi_f32(az_f, &saz, &caz);
  const float a00 = caz * ce * s_x, a01 = -saz * s_y, a02 = caz * se * s_z;
  const float a10 = saz * ce * s_x, a11 = caz * s_y, a12 = saz * se * s_z;
  const float a20 = -se * s_x, a22 = ce * s_z;
  Sym3f cd{a00 * a00 + a01 * a01 + a02 * a02, a00 * a10 + a01 * a11 + a02 * a12, a00 * a20 + a02 * a22, a10 * a10 + a11 * a11 + a12 * a12, a10 * a20 + a12 * a22,
           a20 * a20 + a22 * a22};
  if (cst.plain_gicp) cd = Sym3f{0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float Rf[9];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c2 = 0; c2 < 3; c2++) Rf[3 * r + c2] = (float)T.m[4 * r + c2];
  const Sym3f sA{cA.xx + cd.xx, cA.xy + cd.xy, cA.xz + cd.xz, cA.yy + cd.yy, cA.yz + cd.yz, cA.zz + cd.zz};
  float rc[9];
#pragma unroll
  for (int r = 0; r < 3; r++) {
    const float a = Rf[3 * r], b = Rf[3 * r + 1], d = Rf[3 * r + 2];
    rc[3 * r + 0] = a * sA.xx + b * sA.xy + d * sA.xz;
    rc[3 * r + 1] = a * sA.xy + b * sA.yy + d * sA.yz;
    rc[3 * r + 2] = a * sA.xz + b * sA.yz + d * sA.zz;
  }
  Sym3f RCR;  // (cov_B + cd) + R (cov_A + cd) R^T, A:188
  RCR.xx = (cB.xx + cd.xx) + (rc[0] * Rf[0] + rc[1] * Rf[1] + rc[2] * Rf[2]);
  RCR.xy = (cB.xy + cd.xy) + (rc[0] * Rf[3] + rc[1] * Rf[4] + rc[2] * Rf[5]);
  RCR.xz = (cB.xz + cd.xz) + (rc[0] * Rf[6] + rc[1] * Rf[7] + rc[2] * Rf[8]);
  RCR.yy = (cB.yy + cd.yy) + (rc[3] * Rf[3] + rc[4] * Rf[4] + rc[5] * Rf[5]);
  RCR.yz = (cB.yz + cd.yz) + (rc[3] * Rf[6] + rc[4] * Rf[7] + rc[5] * Rf[8]);
  RCR.zz = (cB.zz + cd.zz) + (rc[6] * Rf[6] + rc[7] * Rf[7] + rc[8] * Rf[8]);
  const float c00 = RCR.yy * RCR.zz - RCR.yz * RCR.yz, c01 = RCR.yz * RCR.xz - RCR.xy * RCR.zz, c02 = RCR.xy * RCR.yz - RCR.yy * RCR.xz;
  const float id = 1.0f / (RCR.xx * c00 + RCR.xy * c01 + RCR.xz * c02);
  Sym3f Mi{c00 * id, c01 * id, c02 * id, (RCR.xx * RCR.zz - RCR.xz * RCR.xz) * id, (RCR.xy * RCR.xz - RCR.xx * RCR.yz) * id, (RCR.xx * RCR.yy - RCR.xy * RCR.xy) * id};
  if (w.maha) {  // (the Mahalanobis getter and compute_error read fp64: widened here)
    double* mo = w.maha + (size_t)pair * 6 * w.nstride + i;
    mo[0] = Mi.xx, mo[w.nstride] = Mi.xy, mo[2 * (size_t)w.nstride] = Mi.xz;
    mo[3 * (size_t)w.nstride] = Mi.yy, mo[4 * (size_t)w.nstride] = Mi.yz, mo[5 * (size_t)w.nstride] = Mi.zz;
  }
  // transed_mean_A and e (A:236-237) in fp64, then narrowed: the residual is the difference of two points 2 .. 100 m from the sensor
  // that lie centimetres apart -- formed in fp32 it carries ~4e-6 m of rounding per point (a first version did: poses of small
  // ill-conditioned pairs moved by up to 1.6e-3 m); formed in fp64 and THEN rounded it keeps 24 bits of its own magnitude.  Twelve
  // fp64 operations per point.
  const double ax = (double)p.x, ay = (double)p.y, az = (double)p.z;
  const double vxd = T.m[0] * ax + T.m[1] * ay + T.m[2] * az + T.m[3];
  const double vyd = T.m[4] * ax + T.m[5] * ay + T.m[6] * az + T.m[7];
  const double vzd = T.m[8] * ax + T.m[9] * ay + T.m[10] * az + T.m[11];
  const float vx = (float)vxd, vy = (float)vyd, vz = (float)vzd;
  const float ex = (float)((double)tq.x - vxd), ey = (float)((double)tq.y - vyd), ez = (float)((double)tq.z - vzd);
  lp.Mi = Mi;
  lp.vx = vx, lp.vy = vy, lp.vz = vz;
  lp.mex = Mi.xx * ex + Mi.xy * ey + Mi.xz * ez;
  lp.mey = Mi.xy * ex + Mi.yy * ey + Mi.yz * ez;
  lp.mez = Mi.xz * ex + Mi.yz * ey + Mi.zz * ez;
  lp.cost = ex * lp.mex + ey * lp.mey + ez * lp.mez;
  lp.matched = 1.0f;
}
__device__ __forceinline__ double lin_term_f32(const LinPointF& lp, int want_Hb, int r) {
  const Sym3f& Mi = lp.Mi;
  const float vx = lp.vx, vy = lp.vy, vz = lp.vz, mex = lp.mex, mey = lp.mey, mez = lp.mez;
  if (r == 27) return (double)lp.cost;
  if (r == 28) return (double)lp.matched;
  if (!want_Hb) return 0.0;
  const float m0x = Mi.xy * vz - Mi.xz * vy, m0y = Mi.yy * vz - Mi.yz * vy, m0z = Mi.yz * vz - Mi.zz * vy;
  const float m1x = -Mi.xx * vz + Mi.xz * vx, m1y = -Mi.xy * vz + Mi.yz * vx, m1z = -Mi.xz * vz + Mi.zz * vx;
  const float m2x = Mi.xx * vy - Mi.xy * vx, m2y = Mi.xy * vy - Mi.yy * vx, m2z = Mi.xz * vy - Mi.yz * vx;
  float v;
  switch (r) {
    case 0: v = vz * m0y - vy * m0z; break;
    case 1: v = vz * m1y - vy * m1z; break;
    case 2: v = vz * m2y - vy * m2z; break;
    case 3: v = -m0x; break;
    case 4: v = -m0y; break;
    case 5: v = -m0z; break;
    case 6: v = -vz * m1x + vx * m1z; break;
    case 7: v = -vz * m2x + vx * m2z; break;
    case 8: v = -m1x; break;
    case 9: v = -m1y; break;
    case 10: v = -m1z; break;
    case 11: v = vy * m2x - vx * m2y; break;
    case 12: v = -m2x; break;
    case 13: v = -m2y; break;
    case 14: v = -m2z; break;
    case 15: v = Mi.xx; break;
    case 16: v = Mi.xy; break;
    case 17: v = Mi.xz; break;
    case 18: v = Mi.yy; break;
    case 19: v = Mi.yz; break;
    case 20: v = Mi.zz; break;
    case 21: v = vz * mey - vy * mez; break;
    case 22: v = -vz * mex + vx * mez; break;
    case 23: v = vy * mex - vx * mey; break;
    case 24: v = -mex; break;
    case 25: v = -mey; break;
    default: v = -mez; break;
  }
  return (double)v;
}

#pragma clang fp contract(off)
// ---- end of the fp64 contraction region

// ----------------------------------------------------------------------------------------------
// k_linearize: one lane per source point.  Merges the split partials, re-scans the winning chunk
// for the exact correspondence, applies the gate (A:156), builds the APD covariance (A:167-184),
// RCR and its inverse (A:188-192, stored for k_error), then e, J, H, b (A:229-258) and reduces
// 21+6+1 sums per block.  fp64 throughout after the NN, as in the reference.
constexpr int LIN_BLK = 256;
constexpr int kSerialRows = 32;  // up to this many block rows (8192 points) the last block adds them one after the other

__device__ __forceinline__ void lm_decide_after_sum(PairState& s, double yi, const Consts& c, double* ws, double* tr);
__device__ __forceinline__ void lm_after_gather(PairState& s, const Consts& c, double* ws, double* tr);
__device__ __forceinline__ void fill_from_sums(PairState& s, const double* v);

// Cross-block traffic inside one launch (the rows of partial sums, read by the last block of a pair) goes through
// agent-scope atomic loads and stores: they are coherent at device level by themselves (sc1), so no release/acquire
// fence is needed.  On this multi-XCD part an agent-scope fence writes back / invalidates a whole L2, which is what made
// the first fused version (fence pair per block) slower than a separate one-block-per-pair solve launch.
__device__ __forceinline__ double ld_coh(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_coh(double* p, double v) {
  __hip_atomic_store((unsigned long long*)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// p[0] + p[stride] + ... (n terms, in index order), the coherent loads issued 16 at a time
__device__ __forceinline__ double sum_rows_coh(const double* p, int n, int stride) {
  double v = 0.0;
  for (int b0 = 0; b0 < n; b0 += 16) {
    double t[16];
#pragma unroll
    for (int u = 0; u < 16; u++) t[u] = ld_coh(p + (size_t)min(b0 + u, n - 1) * stride);
#pragma unroll
    for (int u = 0; u < 16; u++)
      if (b0 + u < n) v += t[u];
  }
  return v;
}
// Arrival ticket: returns true (block-uniformly) in the LAST block of this pair to get here.  The caller has written its
// row with st_coh; every wave waits for its own stores, then one lane draws the ticket.
__device__ __forceinline__ bool last_block_of_pair(int* ticket, int nblk, int tid) {
  __shared__ int s_last;
  __builtin_amdgcn_s_waitcnt(0);
  asm volatile("" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const int t = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = t == nblk - 1;
    if (last) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next tick
    s_last = last;
  }
  __syncthreads();
  return s_last != 0;
}

// want_Hb: 0 = cost only, 1 = H, b, cost, 2 = also run the GN/LM step (L:107-144) in the last block
// FUSED = false: the per-point pass alone (want_Hb 0 or 1): without the optimiser step's register footprint (a 6x6 LDL^T, so3_exp
// and the pose products, all in registers on one lane) the kernel fits 6 waves per SIMD instead of 4.
#ifndef APD_LIN_WPE
#define APD_LIN_WPE 7  // fused kernel: ask for 7 waves per SIMD (71 registers, no scratch; what the LDS of 7 blocks per CU allows): 0.7435 -> 0.7402 ms per step against 6
#endif
#ifndef APD_LIN_WPE_F32
#define APD_LIN_WPE_F32 6  // the fp32 per-point variant: at 7 waves per SIMD (73 registers) it spilled 44 B to scratch
#endif
template <bool FUSED, bool F32 = false, bool ALG = false>
__global__ __launch_bounds__(LIN_BLK) __attribute__((amdgpu_waves_per_eu(FUSED ? (F32 ? APD_LIN_WPE_F32 : APD_LIN_WPE) : 1, 8))) void k_linearize(const CloudDesc* clouds, const PairDesc* pairs, PairState* st, Work w, Consts cst,
                                                       int want_Hb) {
  __shared__ double red[(LIN_BLK / 64) * 29];
  unsigned bx, by;
  xcd_remap(bx, by);
  const int pair = pair_of(w, by);
  if (pair < 0) return;
  const int status = w.init ? (int)ST_NEED_LIN : st[pair].status;  // status, descriptors and pose: one round of scalar loads, not three
  const PairDesc pd = pairs[pair];
  const Rigid T = *(w.init ? w.init + pair : &st[pair].x0);
  if (status != ST_NEED_LIN) return;
  const CloudDesc src = pd.s, tgt = pd.t;
  const int N = src.n, tid = threadIdx.x;
  if ((int)(bx * LIN_BLK) >= N) return;
  const int i = (int)bx * LIN_BLK + tid;

  using LP = std::conditional_t<F32, LinPointF, LinPoint>;
  LP lp;
  if constexpr (F32) {
    lp.Mi = Sym3f{0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    lp.vx = lp.vy = lp.vz = lp.mex = lp.mey = lp.mez = lp.cost = lp.matched = 0.f;
  } else {
    lp.Mi = Sym3{0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    lp.vx = lp.vy = lp.vz = lp.mex = lp.mey = lp.mez = lp.cost = lp.matched = 0.0;
  }
  __shared__ float s_atan[APD_ATAN_TAB_ROWS * APD_ATAN_TAB_STRIDE];
  if constexpr (!ALG) {
    atan_tab_to_lds(s_atan, tid);
    __syncthreads();  // (every exit above is taken by the whole block)
  }

  if (i < N) {
    // everything that depends on the point index alone is requested FIRST -- the point, its covariance, the neighbour on
    // record -- so that it travels together with the search result: read where they are used these were four dependent trips
    // to memory (result -> point -> record -> covariances) in a kernel that lasts 21 us
    const float4 p = G(src.pts)[i];
    const float4 tq_rec = w.nnpt[(size_t)pair * w.nstride + i];
    const auto ca = G((const double*)src.cov);
    const Sym3 cov_A{ca[i], ca[N + i], ca[2 * N + i], ca[3 * N + i], ca[4 * N + i], ca[5 * N + i]};
    unsigned long long bestp = ~0ull;
    bool tie = false;
    const unsigned long long* part = w.nnpart + (size_t)pair * w.T * w.nstride + i;
    for (int sp = 0; sp < w.T; sp++) {
      const unsigned long long v = part[(size_t)sp * w.nstride];
      if ((unsigned)(v >> 32) == (unsigned)(bestp >> 32) && (unsigned)v != kNoChunk) tie = true;  // same minimum in two splits
      if (v < bestp) {
        if ((unsigned)(v >> 32) != (unsigned)(bestp >> 32)) tie = false;
        bestp = v;
      }
    }
    const float m = __uint_as_float((unsigned)(bestp >> 32));
    unsigned chunk = (unsigned)bestp;
    bool kept = false;
    if (chunk != kNoChunk) {
      tie |= (chunk & kTieBit) != 0;
      kept = (chunk & kKeptBit) != 0 && !tie;
      chunk &= kChunkMask;
    }
    float Tf[12];
    load_Tf(T, Tf);
    const float ptx = xf_row(Tf + 0, p.x, p.y, p.z, w.xf_linear), pty = xf_row(Tf + 4, p.x, p.y, p.z, w.xf_linear), ptz = xf_row(Tf + 8, p.x, p.y, p.z, w.xf_linear);
    if constexpr (F32) linearize_point_f32(src, tgt, T, w, cst, pair, i, p, ptx, pty, ptz, m, chunk, tie, kept, tq_rec, cov_A, lp, s_atan);
    else linearize_point<ALG>(src, tgt, T, w, cst, pair, i, p, ptx, pty, ptz, m, chunk, tie, kept, tq_rec, cov_A, lp, s_atan);
  }
  __shared__ double red_scratch[(LIN_BLK / 64) * RED_LDS_WAVE];
  block_reduce_lds<29, LIN_BLK>([&](int r) {
    if constexpr (F32) return lin_term_f32(lp, want_Hb, r);
    else return lin_term(lp, want_Hb, r);
  }, red, red_scratch, tid);
  if (tid < 29) {
    double s = 0.0;
#pragma unroll
    for (int wv = 0; wv < LIN_BLK / 64; wv++) s += red[wv * 29 + tid];
    double* row = w.blkpart + ((size_t)pair * w.nblk_max + bx) * kRed + tid;
    if (FUSED && want_Hb == 2) st_coh(row, s);
    else *row = s;
  }
  if constexpr (FUSED)
  if (want_Hb == 2) {  // the last block of the pair to arrive takes the GN/LM step: no launch of its own
    __shared__ PairState ls;
    const int nblk = (N + LIN_BLK - 1) / LIN_BLK;
    if (last_block_of_pair(w.ticket + pair, nblk, tid)) {
      static_assert(sizeof(PairState) / 8 <= LIN_BLK, "one state word per thread");
      const double word = (!w.init && tid < (int)(sizeof(PairState) / 8)) ? ((const double*)&st[pair])[tid] : 0.0;  // in flight with the rows
      const double* rows = w.blkpart + (size_t)pair * w.nblk_max * kRed;
      if (nblk <= kSerialRows) {
        if (tid < 29) red[tid] = sum_rows_coh(rows + tid, nblk, kRed);
      } else {
        // large clouds (hundreds of rows): 8 segments of rows summed side by side, then added in segment order -- 29 lanes
        // walking 391 rows 16 at a time took longer (37 us for 100k points) than the whole per-point pass
        __shared__ double seg[(LIN_BLK / 32) * 32];
        const int col = tid & 31, sg = tid >> 5, per = (nblk + LIN_BLK / 32 - 1) / (LIN_BLK / 32);
        const int r0 = min(sg * per, nblk), r1 = min(r0 + per, nblk);
        seg[tid] = (col < 29 && r0 < r1) ? sum_rows_coh(rows + (size_t)r0 * kRed + col, r1 - r0, kRed) : 0.0;
        __syncthreads();
        if (tid < 29) {
          double v = 0.0;
#pragma unroll
          for (int g = 0; g < LIN_BLK / 32; g++) v += seg[g * 32 + tid];
          red[tid] = v;
        }
      }
      if (!w.init && tid < (int)(sizeof(PairState) / 8)) ((double*)&ls)[tid] = word;
      __syncthreads();
      if (tid == 0) {
        if (w.init) init_pair_state(ls, w.init + pair, cst.max_iterations);  // first tick: the state starts here
        fill_from_sums(ls, red);
        lm_after_gather(ls, cst, red_scratch, pair == 0 ? w.trace : nullptr);  // (the reduction scratch is free by now)
      }
      __syncthreads();
      for (int q = tid; q < (int)(sizeof(PairState) / 8); q += LIN_BLK) ((double*)&st[pair])[q] = ((const double*)&ls)[q];
    }
  }
}

// ----------------------------------------------------------------------------------------------
// k_error: compute_error (A:275-298) at the trial pose xi with the FROZEN correspondences and
// Mahalanobis matrices of the last linearize.
__global__ __launch_bounds__(LIN_BLK) void k_error(const CloudDesc* clouds, const PairDesc* pairs, PairState* st, Work w, Consts cst, int fuse) {
  __shared__ double red[LIN_BLK / 64];
  unsigned bx, by;
  xcd_remap(bx, by);
  const int pair = pair_of(w, by);
  if (pair < 0) return;
  if (st[pair].status != ST_NEED_ERR) {
    if (w.post && bx == 0 && threadIdx.x == 0) post_result(*w.post, w.post_seq, st[pair]);  // (the step ended in k_linearize)
    return;
  }
  const PairDesc pd = pairs[pair];
  const CloudDesc src = pd.s, tgt = pd.t;
  const int N = src.n, tid = threadIdx.x;
  if ((int)(bx * LIN_BLK) >= N) return;
  const int i = (int)bx * LIN_BLK + tid;
  const Rigid T = st[pair].xi;
  double acc[1] = {0.0};
#if defined(APD_ABL_LM_NO_ERROR)  // ABLATION builds only (wrong results by design): what compute_error's per-point pass costs an LM tick (1), its blocks (2), its launch (3)
  if (false)
#else
  if (i < N)
#endif
  {
    const int corr = w.corr[(size_t)pair * w.nstride + i];
    if (corr >= 0) {
      const float4 p = G(src.pts)[i], q = G(tgt.pts)[corr];
      const double* mo = w.maha + (size_t)pair * 6 * w.nstride + i;
      const size_t ns = w.nstride;
      const double mxx = mo[0], mxy = mo[ns], mxz = mo[2 * ns], myy = mo[3 * ns], myz = mo[4 * ns], mzz = mo[5 * ns];
      const double ax = (double)p.x, ay = (double)p.y, az = (double)p.z;
      {  // fp64 algebra, contracted like linearize_point's (the two costs an LM step compares are evaluated alike)
#pragma clang fp contract(fast)
        const double ex = (double)q.x - (T.m[0] * ax + T.m[1] * ay + T.m[2] * az + T.m[3]);
        const double ey = (double)q.y - (T.m[4] * ax + T.m[5] * ay + T.m[6] * az + T.m[7]);
        const double ez = (double)q.z - (T.m[8] * ax + T.m[9] * ay + T.m[10] * az + T.m[11]);
        const double mex = mxx * ex + mxy * ey + mxz * ez, mey = mxy * ex + myy * ey + myz * ez, mez = mxz * ex + myz * ey + mzz * ez;
        acc[0] = ex * mex + ey * mey + ez * mez;
      }
    }
  }
  block_reduce<1, LIN_BLK>(acc, red, tid);
  if (tid == 0) {
    double s = 0.0;
#pragma unroll
    for (int wv = 0; wv < LIN_BLK / 64; wv++) s += red[wv];
    double* row = w.errpart + (size_t)pair * w.nblk_max + bx;
    if (fuse) st_coh(row, s);
    else *row = s;
  }
  if (fuse) {  // the last block of the pair to arrive decides (L:145-172): no launch of its own
    __shared__ PairState ls;
    __shared__ double s_yi, s_ws[48];
    const int nblk = (N + LIN_BLK - 1) / LIN_BLK;
    if (last_block_of_pair(w.ticket + w.npairs + pair, nblk, tid)) {
      const double* rows = w.errpart + (size_t)pair * w.nblk_max;
      if (nblk <= kSerialRows) {
        if (tid == 0) s_yi = sum_rows_coh(rows, nblk, 1);
      } else {  // large clouds: every thread adds rows tid, tid + 256, ...; the 256 partial sums are then added in thread order
        __shared__ double part[LIN_BLK];
        double v = 0.0;
        for (int r = tid; r < nblk; r += LIN_BLK) v += ld_coh(rows + r);
        part[tid] = v;
        __syncthreads();
        if (tid == 0) {
          double t = 0.0;
          for (int q = 0; q < LIN_BLK; q++) t += part[q];
          s_yi = t;
        }
      }
      for (int q = tid; q < (int)(sizeof(PairState) / 8); q += LIN_BLK) ((double*)&ls)[q] = ((const double*)&st[pair])[q];
      __syncthreads();
      if (tid == 0) {
        lm_decide_after_sum(ls, s_yi, cst, s_ws, pair == 0 ? w.trace : nullptr);
        if (w.post) post_result(*w.post, w.post_seq, ls);
      }
      __syncthreads();
      for (int q = tid; q < (int)(sizeof(PairState) / 8); q += LIN_BLK) ((double*)&st[pair])[q] = ((const double*)&ls)[q];
    }
  }
}

// ----------------------------------------------------------------------------------------------
// Debug trace of the state machine (apdgicp_set_trace / apdgicp_get_trace; Work::trace, null in every product run): what a
// debugger stepping through L:64-76 / L:127-173 would write down -- per LM trial the lambda it was solved with and its rho
// and the two costs it compares (L:137-146: y0 from linearize, yi from compute_error at the trial pose), per completed outer
// iteration the pose x0 behind it (L:119 / L:166).  Layout: {int n_trial, n_pose, cap_trial, cap_pose} (16 bytes),
// lambda[cap_trial], rho[cap_trial], y0[cap_trial], yi[cap_trial], pose[cap_pose][12] (row-major 3x4), then (round 6, behind the rest so
// that nothing moved) dnorm[cap_trial]: |d| of the trial's step, the "|delta|" column of the reference's debug table (L:148-154).
// Counts keep counting past the capacity; only what fits is stored.
__device__ __forceinline__ void trace_trial(double* tr, double lambda, double rho, double y0, double yi, const double* d6) {
  int* h = (int*)tr;
  const int n = h[0], cap = h[2];
  if (n < cap) {
    tr[2 + n] = lambda, tr[2 + cap + n] = rho, tr[2 + 2 * cap + n] = y0, tr[2 + 3 * cap + n] = yi;
    double nn = 0.0;
    for (int q = 0; q < 6; q++) nn += d6[q] * d6[q];
    tr[2 + 4 * cap + 12 * h[3] + n] = sqrt(nn);
  }
  h[0] = n + 1;
}
__device__ __forceinline__ void trace_pose(double* tr, const Rigid& x) {
  int* h = (int*)tr;
  const int n = h[1];
  if (n < h[3]) {
    double* o = tr + 2 + 4 * h[2] + 12 * n;
    for (int q = 0; q < 12; q++) o[q] = x.m[q];
  }
  h[1] = n + 1;
}

// ----------------------------------------------------------------------------------------------
// The Gauss-Newton / Levenberg-Marquardt bookkeeping (L:55-173) as a per-pair state machine.
// s lives in LDS, ws is 48 doubles of LDS: H, b, the factor and the step stay there (see solve6_spd_ws)
__device__ __forceinline__ void lm_trial(PairState& s, double* ws) {  // L:137-144
  // phase by phase through LDS (the compiler barriers keep one phase's values from staying in registers through the next:
  // the step runs in ONE lane of a kernel whose register budget belongs to the per-point pass)
  solve6_spd_ws(s.H, s.lambda, s.b, s.d, ws);
  asm volatile("" ::: "memory");
  s.delta = make_delta(s.d);
  asm volatile("" ::: "memory");
#pragma unroll
  for (int i = 0; i < 3; i++) {  // xi = delta * x0 (rigid_mul's operations, one row at a time)
    const double a0 = s.delta.m[4 * i], a1 = s.delta.m[4 * i + 1], a2 = s.delta.m[4 * i + 2], a3 = s.delta.m[4 * i + 3];
#pragma unroll
    for (int j = 0; j < 3; j++) s.xi.m[4 * i + j] = a0 * s.x0.m[j] + a1 * s.x0.m[4 + j] + a2 * s.x0.m[8 + j];
    s.xi.m[4 * i + 3] = a0 * s.x0.m[3] + a1 * s.x0.m[7] + a2 * s.x0.m[11] + a3;
  }
  asm volatile("" ::: "memory");
}

// after step_optimize returned `ok` (L:71-75): convergence test and loop bookkeeping
__device__ __forceinline__ void step_done(PairState& s, const Consts& c, bool ok, double* tr) {
  if (!ok) {  // "lm not converged!!" -> break
    s.failed = 1;
    s.status = ST_DONE;
    return;
  }
  if (tr) trace_pose(tr, s.x0);
  s.converged = is_converged(s.delta, c.rot_eps, c.trans_eps) ? 1 : 0;
  if (s.converged || s.iter + 1 >= c.max_iterations) {
    s.status = ST_DONE;
    return;
  }
  s.iter += 1;  // nr_iterations_ = i (L:68)
  s.status = ST_NEED_LIN;
}

// sums the block partials of k_linearize in block order (deterministic) into s.H / s.b / s.y0
// the 29 sums of a linearize -> H (both triangles), b, cost, number of matched points
__device__ __forceinline__ void fill_from_sums(PairState& s, const double* v) {
  int q = 0;
  for (int r = 0; r < 6; r++)
    for (int c2 = r; c2 < 6; c2++, q++) s.H[r + 6 * c2] = v[q], s.H[c2 + 6 * r] = v[q];
  for (int r = 0; r < 6; r++) s.b[r] = v[21 + r];
  s.y0 = v[27];
  s.n_matched = (int)v[28];
}
// the probes' reduction (k_probe_reduce): the block partials of k_linearize summed in block order, one thread per sum
__device__ __forceinline__ void gather_linearize(PairState& s, const Work& w, int pair, int nblk, double* lds, int tid) {
  const double* p = w.blkpart + (size_t)pair * w.nblk_max * kRed;
  if (tid < 29) {
    double v = 0.0;
    for (int b = 0; b < nblk; b++) v += p[(size_t)b * kRed + tid];
    lds[tid] = v;
  }
  __syncthreads();
  if (tid == 0) fill_from_sums(s, lds);
}

// one lane: the optimiser step once H, b and the cost are in the state
__device__ __forceinline__ void lm_after_gather(PairState& s, const Consts& c, double* ws, double* tr) {
  s.n_lin += 1;
  const bool gn = c.optimizer == 1;
  if (gn) {  // step_gn
    s.lambda = 0.0;
  } else {
    if (s.lambda < 0.0) {  // L:131-133
      double mx = 0.0;
      for (int q = 0; q < 6; q++) mx = fmax(mx, fabs(s.H[q + 6 * q]));
      s.lambda = c.lm_init_lambda_factor * mx;
    }
    s.nu = 2.0;
    s.inner = 0;
    if (c.lm_max_iterations <= 0) {  // the for loop at L:136 never runs -> return false
      step_done(s, c, false, tr);
      return;
    }
  }
  lm_trial(s, ws);  // (ONE call site: two inlined copies of the solve and the trigonometry doubled the one-lane region's register demand)
  if (gn) {
    s.x0 = s.xi;
    for (int q = 0; q < 36; q++) s.final_H[q] = s.H[q];
    step_done(s, c, true, tr);
  } else {
#if defined(APD_ABL_LM_NO_ERROR) && APD_ABL_LM_NO_ERROR >= 2
    lm_decide_after_sum(s, 0.5 * s.y0, c, ws, tr);  // (every first trial accepted, as on the loop pairs of the bench: the ticks per pair stay what they are)
#else
    s.status = ST_NEED_ERR;
#endif
  }
}

// after k_error: L:145-172 (one lane)
__device__ __forceinline__ void lm_decide_after_sum(PairState& s, double yi, const Consts& c, double* ws, double* tr) {
  s.yi = yi;
  s.n_err += 1;
  double den = 0.0;
  for (int q = 0; q < 6; q++) den += s.d[q] * (s.lambda * s.d[q] - s.b[q]);
  const double rho = (s.y0 - yi) / den;  // L:146
  if (tr) trace_trial(tr, s.lambda, rho, s.y0, yi, s.d);
  if (rho < 0) {                         // L:156-164
    if (is_converged(s.delta, c.rot_eps, c.trans_eps)) {
      step_done(s, c, true, tr);  // returns true WITHOUT applying delta
      return;
    }
    s.lambda = s.nu * s.lambda;
    s.nu = 2 * s.nu;
    s.inner += 1;
    if (s.inner >= c.lm_max_iterations) {  // L:172
      step_done(s, c, false, tr);
      return;
    }
    lm_trial(s, ws);  // next inner iteration, status stays ST_NEED_ERR
    return;
  }
  s.x0 = s.xi;  // L:166-169
  const double t = 2 * rho - 1;
  s.lambda = s.lambda * fmax(1.0 / 3.0, 1 - t * t * t);
  for (int q = 0; q < 36; q++) s.final_H[q] = s.H[q];
  step_done(s, c, true, tr);
}

// Launch order of a dense search's blocks: positions 0 .. n - 1 -> block indices by descending cost (ties: ascending index).  One block,
// keys (0xFFFFF - cost) << 12 | index in LDS, bitonic network over the next power of two; n <= 4096.
constexpr int ORDER_MAX = 4096;
__global__ __launch_bounds__(1024) void k_block_order(const unsigned* cost, int n, unsigned* order) {
  __shared__ unsigned key[ORDER_MAX];
  const int tid = threadIdx.x;
  int np2 = 1024;
  while (np2 < n) np2 <<= 1;
  for (int e = tid; e < np2; e += 1024) key[e] = e < n ? ((0xFFFFFu - min(cost[e], 0xFFFFFu)) << 12) | (unsigned)e : 0xFFFFFFFFu;
  __syncthreads();
  for (int kk = 2; kk <= np2; kk <<= 1)
    for (int j = kk >> 1; j > 0; j >>= 1) {
      for (int e = tid; e < np2; e += 1024) {
        const int l = e ^ j;
        if (l > e) {
          const unsigned a = key[e], b = key[l];
          const bool up = (e & kk) == 0;
          if ((a > b) == up) key[e] = b, key[l] = a;
        }
      }
      __syncthreads();
    }
  for (int e = tid; e < n; e += 1024) order[e] = key[e] & 0xFFFu;
}

// L:56-59: x0 = guess.cast<double>(), lm_lambda_ = -1, converged_ = false
__global__ void k_init_state(PairState* st, const Rigid* guesses /* one per pair, or null */, int npairs, int max_iterations,
                             int* ticket /* [2][npairs] arrival counters of the fused optimiser step */) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npairs) return;
  ticket[p] = 0, ticket[npairs + p] = 0;  // (a run that ended in an error may have left them mid-count)
  init_pair_state(st[p], guesses ? guesses + p : nullptr, max_iterations);
}

// batch probe support: x0 of every pair := T[pair] (column-major float 4x4), status := NEED_LIN
__global__ void k_set_poses(PairState* st, const float* T16, int npairs) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npairs) return;
  PairState& s = st[p];
  if (T16) {
    const float* g = T16 + 16 * p;
    for (int r = 0; r < 3; r++)
      for (int c2 = 0; c2 < 4; c2++) s.x0.m[4 * r + c2] = (double)g[r + 4 * c2];
  }
  s.status = ST_NEED_LIN;
  s.n_lin = 0;  // cold search (see k_set_probe)
}

// probe support: put pair 0 into a given state at pose T (column-major double 4x4)
__global__ void k_set_probe(PairState* st, const double* T16, int status, int use_xi) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  PairState& s = st[0];
  Rigid r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 4; j++) r.m[4 * i + j] = T16[i + 4 * j];
  if (use_xi) s.xi = r;
  else s.x0 = r;
  s.status = status;
  s.n_lin = 0;  // probes always search cold: the hint array may describe another pair or pose
}

// probe support: reduce the linearize / error partials of pair 0 into out[0..43]:
// H (36, column-major), b (6), cost, matched
__global__ __launch_bounds__(64) void k_probe_reduce(const CloudDesc* clouds, const PairDesc* pairs, PairState* st, Work w, double* out, int which) {
  __shared__ double lds[32];
  const int tid = threadIdx.x;
  const int N = pairs[0].s.n;
  const int nblk = (N + LIN_BLK - 1) / LIN_BLK;
  PairState& s = st[0];
  if (which == 0) {
    gather_linearize(s, w, 0, nblk, lds, tid);
    if (tid == 0) {
      for (int q = 0; q < 36; q++) out[q] = s.H[q];
      for (int q = 0; q < 6; q++) out[36 + q] = s.b[q];
      out[42] = s.y0;
      out[43] = (double)s.n_matched;
      s.n_lin += 1;
    }
  } else if (tid == 0) {
    double yi = 0.0;
    for (int b = 0; b < nblk; b++) yi += w.errpart[b];
    out[42] = yi;
    s.n_err += 1;
  }
}

// also the poll: status_out (optional) receives every pair's status and, behind them, the device error flag
// host_out / host_status (optional): the same records / status words + error flag written straight into pinned host memory,
// so the poll needs no copy behind this kernel
__device__ __forceinline__ void finalize_pair(const PairState* st, ResultRec* out, int* status_out, ResultRec* host_out, int* host_status, int p);
// host_seq (optional; one-block launches only): after every record and status word of this poll has been written through to
// host memory the word receives `seq` -- the host spins on it instead of sleeping on an event (a few microseconds per poll,
// which is most of what a one-iteration LM registration has left to give)
__global__ void k_finalize(const PairState* st, ResultRec* out, int* status_out, int npairs, const int* err_flag, ResultRec* host_out,
                           int* host_status, int* host_seq = nullptr, int seq = 0) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p == 0) {
    const int flag = err_flag ? *err_flag : 0;
    if (status_out) status_out[npairs] = flag;
    if (host_status) host_status[npairs] = flag;
  }
  if (host_seq) {  // (gridDim.x == 1)
    if (p < npairs) finalize_pair(st, out, status_out, host_out, host_status, p);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(host_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    return;
  }
  if (p >= npairs) return;
  finalize_pair(st, out, status_out, host_out, host_status, p);
}
__device__ __forceinline__ void finalize_pair(const PairState* st, ResultRec* out, int* status_out, ResultRec* host_out, int* host_status, int p) {
  const PairState& s = st[p];
  const ResultRec r = result_record(s);
  out[p] = r;
  if (host_out) host_out[p] = r;
  if (status_out) status_out[p] = s.status;
  if (host_status) host_status[p] = s.status;
}

// ----------------------------------------------------------------------------------------------
// Pooled Levenberg-Marquardt batches (Engine::pool_*).  The run length of an LM registration is data dependent (L:64-76:
// 2 ... 51 outer iterations over the loop-closure pairs of one batch), so the host cannot enqueue "the" ticks of a batch;
// waiting for a status poll every few ticks left the GPU idle during every round trip and held a whole batch by its slowest
// pair.  Instead the pairs of ALL batches in flight on a handle sit in one pool of pair slots (lane = the segment of one
// batch) and every tick launch covers the device-side list of the pairs that still run.  This kernel is the only
// bookkeeping between chunks of ticks, one block:
//   * drops the pairs that reached ST_DONE from the list (stable, in place) and writes their result records -- device copy
//     and pinned host copy (k_finalize's work, per pair, once);
//   * admits the pairs of newly enqueued batches: state from the guess (L:56-59, k_init_state's work), arrival counters
//     zeroed, appended to the list;
//   * pads the list with -1 (tick launches are sized by the host's upper bound of its length) and posts a header -- length,
//     pairs left per lane, the device error flag (taken and reset) -- behind everything else in pinned memory.
// The host never has to answer for the GPU to keep working: it keeps two chunks enqueued and reads the headers as they arrive.
constexpr int kPoolLanes = 32, kPoolRing = 8;
struct PoolHdr {
  int n_active, errflag;
  int lane_left[kPoolLanes];
  int pad_[29];
  int seq;  // written last, system-scope release
};
struct PoolAdmit {
  int count, kill_mask;  // kill_mask: lanes whose pairs end now (their batch failed: device error flag)
  int seg0[kPoolLanes], np[kPoolLanes];
};
__global__ __launch_bounds__(256) void k_pool_poll(PairState* st, int* active, int* n_active, int cap, int segcap, PoolAdmit adm, const Rigid* guesses,
                                                   int max_iterations, int* ticket, ResultRec* out, ResultRec* host_out, PoolHdr* hdr, int seq,
                                                   int* err_flag) {
  __shared__ int s_left[kPoolLanes], s_wcnt[4], s_base;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (tid < kPoolLanes) s_left[tid] = 0;
  if (tid == 0) s_base = 0;
  const int n = *n_active;
  __syncthreads();
  for (int r0 = 0; r0 < n; r0 += 256) {
    const int i = r0 + tid;
    const int pair = i < n ? active[i] : -1;
    bool keep = false;
    if (pair >= 0) {
      const int ln = pair / segcap;
      if ((adm.kill_mask >> ln) & 1) st[pair].status = ST_DONE, st[pair].failed = 1;
      keep = st[pair].status != ST_DONE;
      if (keep) {
        atomicAdd(&s_left[ln], 1);
      } else {
        const ResultRec r = result_record(st[pair]);
        out[pair] = r;
        host_out[pair] = r;
      }
    }
    const unsigned long long m = __ballot(keep);
    if (lane == 0) s_wcnt[wid] = __popcll(m);
    __syncthreads();  // (every read of this round is done: the in-place writes below never pass position i)
    int off = s_base;
    for (int o = 0; o < wid; o++) off += s_wcnt[o];
    if (keep) active[off + __popcll(m & ((1ull << lane) - 1ull))] = pair;
    __syncthreads();
    if (tid == 0) s_base += s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
    __syncthreads();
  }
  int base = s_base;
  for (int a = 0; a < adm.count; a++) {
    for (int j = tid; j < adm.np[a]; j += 256) {
      const int p = adm.seg0[a] + j;
      init_pair_state(st[p], guesses + p, max_iterations);
      ticket[p] = 0, ticket[cap + p] = 0;
      active[base + j] = p;
    }
    if (tid == 0) s_left[adm.seg0[a] / segcap] += adm.np[a];
    base += adm.np[a];
  }
  for (int i = base + tid; i < cap; i += 256) active[i] = -1;
  __threadfence_system();
  __syncthreads();
  if (tid == 0) {
    *n_active = base;
    hdr->n_active = base;
    hdr->errflag = __hip_atomic_exchange(err_flag, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int q = 0; q < kPoolLanes; q++) hdr->lane_left[q] = s_left[q];
    __threadfence_system();
    __hip_atomic_store(&hdr->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// apdgicp_debug_atan2f: the device's evaluation of include/apd_atan2f.h, for the bit-for-bit comparison with the host's
__global__ void k_debug_atan2f(const float* y, const float* x, float* out, long long n) {
  __shared__ float s_atan[APD_ATAN_TAB_ROWS * APD_ATAN_TAB_STRIDE];
  atan_tab_to_lds(s_atan, (int)threadIdx.x);
  __syncthreads();
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = apd_atan2f_tab(y[i], x[i], s_atan);
}

// pcl::transformPointCloud (L:79): float 4x4 times {x,y,z,1}
__global__ void k_transform_points(const float4* pts /* original order */, int n, const float* T16 /* column-major, device */, float* out, long long out_stride_floats) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 p = pts[i];
  float* o = out + (long long)i * out_stride_floats;
  o[0] = T16[0] * p.x + T16[4] * p.y + T16[8] * p.z + T16[12];
  o[1] = T16[1] * p.x + T16[5] * p.y + T16[9] * p.z + T16[13];
  o[2] = T16[2] * p.x + T16[6] * p.y + T16[10] * p.z + T16[14];
}

// fitness (pcl::Registration::getFitnessScore): per pair, sum and count of the nearest-neighbour squared
// distances <= max_range2 taken from the merged nn partials of a search at the pose in st[pair].x0.
// out[2*pair] += d2 (double), out[2*pair+1] += 1
__global__ __launch_bounds__(LIN_BLK) void k_fitness(const CloudDesc* clouds, const PairDesc* pairs, Work w, double max_range2, double* out, int strict) {
  __shared__ double red[(LIN_BLK / 64) * 2];
  const int pair = pair_of(w, blockIdx.y);
  const int N = pairs[pair].s.n, tid = threadIdx.x;
  if ((int)(blockIdx.x * LIN_BLK) >= N) return;
  const int i = blockIdx.x * LIN_BLK + tid;
  double acc[2] = {0.0, 0.0};
  if (i < N) {
    unsigned long long bestp = ~0ull;
    const unsigned long long* part = w.nnpart + (size_t)pair * w.T * w.nstride + i;
    for (int sp = 0; sp < w.T; sp++) {
      const unsigned long long v = part[(size_t)sp * w.nstride];
      bestp = v < bestp ? v : bestp;
    }
    const float m = __uint_as_float((unsigned)(bestp >> 32));
    // pcl getFitnessScore keeps d <= max_range; the odometry status counts inliers with d < max^2 (scan_matching_odometry_nodelet.cpp:707)
    if ((unsigned)bestp != kNoChunk && (strict ? (double)m < max_range2 : (double)m <= max_range2)) acc[0] = (double)m, acc[1] = 1.0;
  }
  block_reduce<2, LIN_BLK>(acc, red, tid);
  if (tid == 0) {
    double a = 0, b = 0;
    for (int wv = 0; wv < LIN_BLK / 64; wv++) a += red[wv * 2], b += red[wv * 2 + 1];
    atomicAdd(out + 2 * pair, a);
    atomicAdd(out + 2 * pair + 1, b);
  }
}

}  // namespace apd
