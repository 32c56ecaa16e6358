/*
 * apdgicp_hip.h -- C ABI of libapdgicp_hip.so: RIV-SLAM's APD-GICP scan matcher on MI355X (gfx950).
 *
 * This is the drop-in boundary for ONE hot path of Wayne-DWA/RIV-SLAM: fast_gicp::FastAPDGICP
 * behind pcl::Registration, as selected by select_registration_method()
 * (radar_graph_slam/src/radar_graph_slam/registrations.cpp:38-50).  Each entry point names the
 * reference member it replaces; "A:" = fast_apdgicp/include/fast_gicp/gicp/impl/fast_apdgicp_impl.hpp,
 * "L:" = .../gicp/impl/lsq_registration_impl.hpp, "H:" = .../gicp/fast_apdgicp.hpp.
 *
 * Conventions
 *   - Plain C: pointers, sizes, PODs.  No C++/torch types.  All functions return 0 on success and a
 *     negative apdgicp_status on failure; apdgicp_last_error() gives the message (thread-local).
 *     Nothing aborts or throws.  Registration failure is DATA (result.converged == 0), like
 *     hasConverged() in the reference (L:71-75), not an error code.
 *   - 4x4 matrices are COLUMN-MAJOR (Eigen::Matrix4f / Matrix4d memory layout): m[row + 4*col].
 *     6x6 H is column-major too (symmetric anyway); b is [rot(3), trans(3)] as in A:248-250.
 *   - Points: `xyz` is the address of the first x; consecutive points are `stride_bytes` apart
 *     (12 for packed xyz, 16 for float4, 32 for pcl::PointXYZI).  Only x,y,z are read
 *     (intensity is never touched on this path).  Points must be finite.
 *   - `on_device` != 0 means `xyz` is a device (HIP) pointer on the handle's device; the data is
 *     copied into the handle's own buffers either way, so the caller may free/reuse its buffer when
 *     the call returns (host) / when the handle's stream has passed the call (device).
 *   - A handle is single-caller (one thread at a time), owns one HIP stream and no global state;
 *     any number of handles may live in one process and be used from different threads.
 */
#ifndef APDGICP_HIP_H
#define APDGICP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define APDGICP_ABI_VERSION 6   /* 2: + inlier_fraction, wait_producer, get_stream; 3: pooled LM batches (enqueue never blocks), + batch_pump, sparse cloud slots;
                                   4: T*p is summed pairwise by default (Eigen >= 3.3), APDGICP_FLAG_XF_LINEAR_CHAIN selects the former order;
                                   5: the three fp32 angles of the sensor model (A:168,172-173) through glibc's atan2f algorithm (apd_atan2f.h) instead of the
                                      device library's; + source_stamp, set_trace / get_trace, debug_atan2f;
                                   6: + APDGICP_FLAG_ALGEBRAIC_APD, build_flags, nearest_neighbours_of, get_trace_step_norms */

typedef enum {
  APDGICP_OK = 0,
  APDGICP_ERR_INVALID_ARG = -1,   /* null pointer, bad size, k out of range ... */
  APDGICP_ERR_HIP = -2,           /* a HIP runtime call failed (no device, OOM, launch failure) */
  APDGICP_ERR_NO_INPUT = -3,      /* source/target (or correspondences) not set yet */
  APDGICP_ERR_TOO_FEW_POINTS = -4,/* cloud has fewer than k_correspondences points (reference: UB, A:318-321) */
  APDGICP_ERR_UNSUPPORTED = -5,   /* e.g. unknown regularization (reference aborts, A:341-343) */
  APDGICP_ERR_INTERNAL = -6
} apdgicp_status;

/* fast_gicp::RegularizationMethod, gicp/gicp_settings.hpp:6 (same numeric values) */
typedef enum {
  APDGICP_REG_NONE = 0,
  APDGICP_REG_MIN_EIG = 1,
  APDGICP_REG_NORMALIZED_MIN_EIG = 2,
  APDGICP_REG_PLANE = 3,
  APDGICP_REG_FROBENIUS = 4
} apdgicp_regularization;

/* apdgicp_params.flags.  PLAIN_GICP: drop the range-dependent polar noise covariance (cov_dist, A:167-184), which turns the
 * cost into upstream fast_gicp::FastGICP (gicp/impl/fast_gicp_impl.hpp: RCR = cov_B + T cov_A T^T) -- the FAST_GICP branch of
 * select_registration_method() (registrations.cpp:28-37). */
/* XF_LINEAR_CHAIN: the fp32 summation order of `pt = trans_f * p.getVector4fMap()` (A:137,149), the one fp32 operation of the path
 * whose order belongs to Eigen, not to the reference.  Default (flag clear): (r0 x + r1 y) + (r2 z + t) -- Eigen >= 3.3, whose
 * coefficient-based product sums the four terms of a row with redux_novec_unroller (halving), i.e. the Eigen 3.3.4 / 3.3.7 of the
 * platforms the reference names (README.md:5-7).  Flag set: ((r0 x + r1 y) + r2 z) + t -- Eigen 3.2's product_coeff_impl, and
 * what ABI versions <= 3 evaluated.  The two differ in the last ulp of a transformed coordinate: enough to resolve a
 * nearest-neighbour near-tie the other way (poses move by <= 1e-5 m, an LM run may stop one iteration earlier or later).
 * INTEGRATION.md section 7 holds a 30-line probe that tells which order an installed Eigen produces. */
/* FP32_POINT_MATH (opt-in, not the reference's arithmetic): the per-point algebra behind the nearest-neighbour search -- sin / cos of
 * elevation and azimuth, the APD covariance A A^T, RCR = (C_B + cov_d) + R (C_A + cov_d) R^T, its inverse, the residual, M e and
 * every J^T M J term (A:174-192, A:229-258) -- in fp32 instead of fp64.  Stays fp64: cos(AoA) and its reciprocal (it cancels near
 * the +-x axis, A:170-171), the sums over the points, the 6x6 solve and the pose update; the search, the gate and the three fp32
 * angles are unchanged, so correspondences are identical at a given pose.  Measured: final poses move by
 * ~1e-6 m / 1e-7 rad against the default, a Levenberg-Marquardt run may stop an iteration earlier or later.  The default (flag
 * clear) is the reference's precision; bench.py's headline runs with the flag clear. */
/* ALGEBRAIC_APD (opt-in, not the reference's arithmetic; ABI 6): the sensor model of A:167-184 without its three fp32 atan2 calls and
 * three fp64 sin / cos pairs.  The reference rounds the angle of arrival, the elevation (polar from +z) and the azimuth of the
 * transformed point to fp32 and takes fp64 sines and cosines of those; everything the model USES of them are ratios of the point's own
 * coordinates, which this mode forms directly in fp64 from the fp32 transformed point (x, y, z):
 *     cos(az) = x / rho, sin(az) = y / rho (rho = sqrt(x^2 + y^2)),   sin(el) = rho / r, cos(el) = z / r (r = sqrt(x^2 + y^2 + z^2)),
 *     1 / cos(AoA) = r / sqrt(y^2 + z^2), clamped to 1 / |cos(float(pi/2))| = 2.2877e7 -- the largest value the reference's fp32 angle can give,
 * three reciprocal square roots instead of ~255 fp32 and ~120 fp64 instructions per matched point.  Conventions of atan2(0, 0) = 0 on the
 * axes are kept (rho = 0: cos(az) = 1, sin(az) = 0).  The search, the gate and hence the correspondences at a given pose are unchanged;
 * the APD covariance differs from the reference's by the fp32 rounding of its angles (~6e-8 relative in the rotation), final poses by
 * ~1e-7 m (measured: DESIGN.md section 6; tests/test_hip_parity.py::test_algebraic_apd_*), a Levenberg-Marquardt run may stop an
 * iteration earlier or later.  Exclusive with FP32_POINT_MATH.  The default (flag clear) is the reference's arithmetic and bench.py's
 * headline runs with the flag clear. */
enum { APDGICP_FLAG_PLAIN_GICP = 1, APDGICP_FLAG_XF_LINEAR_CHAIN = 2, APDGICP_FLAG_FP32_POINT_MATH = 4, APDGICP_FLAG_ALGEBRAIC_APD = 8 };

/* fast_gicp::LSQ_OPTIMIZER_TYPE, gicp/lsq_registration.hpp:13 (reference default: LM, L:17) */
typedef enum { APDGICP_OPT_LM = 0, APDGICP_OPT_GN = 1 } apdgicp_optimizer;

/* Every tunable the reference object has.  Defaults (apdgicp_default_params) are the class
 * defaults: A:14-28, H:107-109, L:11-24.  The ROS factory overrides some of them
 * (registrations.cpp:41-48). */
typedef struct {
  int32_t k_correspondences;            /* setCorrespondenceRandomness, A:45 ; default 20 ; any k >= 1, exact: 1..32 the pruned kernel, 33..64 the brute-force one, above a selection kernel (~35 sweeps of the cloud per query block: experiments) */
  int32_t max_iterations;               /* pcl setMaximumIterations ; default 64, L:13 */
  int32_t lm_max_iterations;            /* L:19 ; default 10 */
  int32_t optimizer;                    /* apdgicp_optimizer ; default LM */
  int32_t regularization;               /* apdgicp_regularization ; default PLANE, A:25 */
  int32_t flags;                        /* APDGICP_FLAG_*; default 0 */
  double max_correspondence_distance;   /* pcl setMaxCorrespondenceDistance ; default FLT_MAX, A:23 */
  double transformation_epsilon;        /* pcl setTransformationEpsilon ; default 5e-4, L:15 */
  double rotation_epsilon;              /* setRotationEpsilon, L:30 ; default 2e-3 */
  double lm_init_lambda_factor;         /* setInitialLambdaFactor, L:35 ; default 1e-9 */
  double distance_variance;             /* setDistVar, A:63 ; default 0.86 */
  double azimuth_variance_deg;          /* setAzimuthVar, A:55 ; default 0.5 */
  double elevation_variance_deg;        /* setElevationVar, A:59 ; default 1.0 */
} apdgicp_params;

/* What align() reports.  converged/iterations mirror pcl::Registration::converged_ /
 * nr_iterations_ as written by L:59,68,75 ; T is final_transformation_ (L:78). */
typedef struct {
  float T[16];                 /* column-major source->target */
  double final_cost;           /* last linearize() cost y0 (sum of e^T M e), for logging */
  int32_t converged;
  int32_t iterations;          /* nr_iterations_: zero-based index of the last outer iteration */
  int32_t n_linearize;         /* number of linearize() evaluations (A:198) */
  int32_t n_compute_error;     /* number of compute_error() evaluations (A:275) */
  int32_t lm_failed;           /* 1 when step_lm exhausted lm_max_iterations ("lm not converged!!", L:71-74) */
  int32_t n_matched;           /* correspondences inside the gate at the last linearize (A:156) */
} apdgicp_result;

typedef struct apdgicp_handle apdgicp_handle;   /* one registration object == one FastAPDGICP */
typedef struct apdgicp_batch apdgicp_batch;     /* many independent registrations on one GPU */

enum { APDGICP_SOURCE = 0, APDGICP_TARGET = 1 };

/* ------------------------------------------------------------------ library */
int apdgicp_abi_version(void);
/* Fingerprint of the kernel sources this library was compiled from (riv-slam_amd/build.py:source_stamp(), the first 16 hex digits
 * of a SHA-256 over csrc/ and include/, passed in at compile time): what bench.py and the test suite compare with the sources on
 * disk before they trust a prebuilt library, and what a committed counter profile names.  "unstamped" for a build that did not
 * go through build.py. */
const char* apdgicp_source_stamp(void);
/* The compiler flags this library was built with (as build.py passed them; "unknown" for a hand build), then " | variant:" followed by every
 * experiment define compiled in that changes kernels or results (APD_ABL_*: ablations, wrong by design; APD_OCML_ATAN2F, APD_SINCOS_NO_TABLE:
 * A/B builds) -- detected by the preprocessor inside the library.  Empty after "variant:" = the product.  The Python loader refuses a variant
 * library unless APDGICP_ALLOW_VARIANT_LIB=1; bench.py prints the string in its line. */
const char* apdgicp_build_flags(void);
const char* apdgicp_last_error(void);
int apdgicp_device_count(int* count);
void apdgicp_default_params(apdgicp_params* p);                                  /* A:14-28, L:11-24 */

/* ------------------------------------------------------------------ single registration object */
/* FastAPDGICP::FastAPDGICP() (A:14).  `stream` may be NULL (the handle creates its own) or a
 * hipStream_t the caller owns. */
int apdgicp_create(const apdgicp_params* p, int device, void* stream, apdgicp_handle** out);
int apdgicp_destroy(apdgicp_handle* h);                                           /* ~FastAPDGICP */
int apdgicp_set_params(apdgicp_handle* h, const apdgicp_params* p);               /* the setters A:34-65, L:30-37 */
int apdgicp_get_params(const apdgicp_handle* h, apdgicp_params* p);

/* setInputSource / setInputTarget (A:90-108).  `token` is the caller's identity of the cloud
 * (the adapter passes the shared_ptr's raw address): a call with the token already held is the
 * reference's pointer-equality early return and keeps the cached covariances; token 0 never matches. */
int apdgicp_set_source(apdgicp_handle* h, const float* xyz, int64_t n, int64_t stride_bytes, int on_device, uint64_t token);
int apdgicp_set_target(apdgicp_handle* h, const float* xyz, int64_t n, int64_t stride_bytes, int on_device, uint64_t token);
int apdgicp_clear_source(apdgicp_handle* h);                                      /* A:78-81 */
int apdgicp_clear_target(apdgicp_handle* h);                                      /* A:84-87 */
int apdgicp_swap_source_and_target(apdgicp_handle* h);                            /* A:68-75 */

/* calculate_covariances (A:303-363), normally run lazily by align (A:122-127).  which = APDGICP_SOURCE/TARGET */
int apdgicp_compute_covariances(apdgicp_handle* h, int which);
/* getSourceCovariances / getTargetCovariances (H:67-73): n x 16 doubles, each a column-major 4x4
 * with zero 4th row/column, exactly the reference's std::vector<Matrix4d> memory. */
int apdgicp_get_covariances(apdgicp_handle* h, int which, double* out_n16, int64_t n);
/* setSourceCovariances / setTargetCovariances (A:111-118) */
int apdgicp_set_covariances(apdgicp_handle* h, int which, const double* in_n16, int64_t n);

/* linearize (A:198-272) at pose T (the public probe is LsqRegistration::evaluateCost, L:50-52).
 * H and b may both be NULL (cost only, A:242-244).  Updates the correspondences and Mahalanobis
 * matrices held by the handle (update_correspondences, A:133-194). */
int apdgicp_linearize(apdgicp_handle* h, const double T[16], double H[36], double b[6], double* cost);
/* compute_error (A:275-298): frozen correspondences / Mahalanobis of the last linearize */
int apdgicp_compute_error(apdgicp_handle* h, const double T[16], double* cost);
/* correspondences_ / sq_distances_ (H:104-105) and mahalanobis_ (H:102; n x 16 doubles, zeros for
 * unmatched points) after the last linearize; any pointer may be NULL.  After apdgicp_linearize every sq_dist is the
 * exact nearest-neighbour distance.  Inside apdgicp_align the search stops at the correspondence gate: a point with
 * corr == -1 then reports the smallest float >= max_correspondence_distance^2 (or the distance to its previous
 * neighbour) instead of the distance to a neighbour the reference would reject anyway (A:156); correspondences are exact. */
int apdgicp_get_correspondences(apdgicp_handle* h, int32_t* corr, float* sq_dist, int64_t n);
int apdgicp_get_mahalanobis(apdgicp_handle* h, double* out_n16, int64_t n);

/* pcl::Registration::align(output, guess) -> FastAPDGICP::computeTransformation (A:121-130) ->
 * LsqRegistration::computeTransformation (L:55-80): the whole GN/LM loop runs on the device.
 * guess may be NULL (identity, as align(output)). */
int apdgicp_align(apdgicp_handle* h, const float guess[16], apdgicp_result* out);
/* same loop, but driven from the host through apdgicp_linearize / apdgicp_compute_error exactly
 * like the reference's virtual calls (L:127-173): the bit-faithful debug path */
int apdgicp_align_host_loop(apdgicp_handle* h, const float guess[16], apdgicp_result* out);
int apdgicp_get_final_hessian(apdgicp_handle* h, double H[36]);                   /* getFinalHessian, L:45 */
/* Debug: the optimiser's per-iteration trace, as a debugger stepping through L:64-76 / L:127-173 would write it down.  With
 * tracing enabled every apdgicp_align (the device state machine writes the trace itself) and apdgicp_align_host_loop of this
 * handle records, per Levenberg-Marquardt trial, the lambda the step was solved with, its gain ratio rho and the two costs rho
 * compares (L:137-146: y0 of linearize, yi of compute_error at the trial pose; y0s / yis may be NULL) and, per completed outer
 * iteration, the pose x0 behind it (L:119 / L:166; column-major 4x4 doubles).  get_trace returns the counts of the last align
 * (they may exceed the capacities given: only what fits is copied).  Gauss-Newton runs record poses only. */
int apdgicp_set_trace(apdgicp_handle* h, int enable);
int apdgicp_get_trace(apdgicp_handle* h, int64_t trial_capacity, double* lambdas, double* rhos, double* y0s, double* yis, int64_t* n_trials,
                      int64_t pose_capacity, double* poses16, int64_t* n_poses);
/* ... and, per trial, the norm of the step d it solved for: the "|delta|" column of the table the reference prints under setDebugPrint
 * (L:148-154).  Same counts and order as the trials of apdgicp_get_trace. */
int apdgicp_get_trace_step_norms(apdgicp_handle* h, int64_t capacity, double* norms, int64_t* n_trials);
/* Debug: out[i] = atan2f(y[i], x[i]) as the kernels evaluate it on `device` (include/apd_atan2f.h: glibc's generic atan2f restated;
 * A:168,172-173 call the C library's float overload); host arrays.  For the bit-for-bit comparison with the host's evaluation. */
int apdgicp_debug_atan2f(int device, const float* y, const float* x, float* out, int64_t n);
/* pcl::transformPointCloud(*input_, output, final_transformation_) (L:79): writes n xyz triples
 * `out_stride_bytes` apart into host memory */
int apdgicp_transform_source(apdgicp_handle* h, const float T[16], float* out_xyz, int64_t n, int64_t out_stride_bytes);
/* pcl::Registration::getFitnessScore(max_range): mean squared 1-NN distance of the T-transformed
 * source to the target over the points whose SQUARED distance is <= max_range (PCL compares the
 * squared distance with max_range as is); DBL_MAX when no point qualifies.  Callers:
 * loop_detector.cpp:229, scan_matching_odometry_nodelet.cpp:698.  n_inliers may be NULL */
int apdgicp_fitness_score(apdgicp_handle* h, const float T[16], double max_range, double* score, int64_t* n_inliers);
/* ScanMatchingStatus::inlier_fraction (scan_matching_odometry_nodelet.cpp:701-712): the number of T-transformed source
 * points whose nearest target point is STRICTLY closer than max_correspondence_dist (squared float distance < dist*dist
 * in double, as there), divided by the source size in float.  n_inliers may be NULL */
int apdgicp_inlier_fraction(apdgicp_handle* h, const float T[16], double max_correspondence_dist, double* fraction, int64_t* n_inliers);
/* The nearest target point of every T-transformed source point: what pcl::search::KdTree::nearestKSearch(T * source[i], 1, ...)
 * returns, for all i in ONE batched search on the device -- index into the target cloud as set (a tie: the lowest index) and the
 * fp32 squared distance (FLANN L2_Simple order on the fp32-transformed point, like A:149-153); no correspondence gate.  This is
 * what serves the base-class calls of the nodelets -- getFitnessScore() (loop_detector.cpp:229) and
 * getSearchMethodTarget()->nearestKSearch(aligned[i], 1, ...) (scan_matching_odometry_nodelet.cpp:697-707) -- through the search
 * object FastAPDGICPHip installs, so that PCL never builds its CPU kd-tree.  Leaves the handle as apdgicp_linearize at T would. */
int apdgicp_nearest_neighbours(apdgicp_handle* h, const float T[16], int32_t* index, float* sq_dist, int64_t n);
/* The nearest target point of ARBITRARY query points (host memory, n x {x, y, z, ...} floats, stride_bytes >= 12) in one batched device
 * pass: what pcl::search::Search::nearestKSearch(cloud, indices, 1, ...) returns (search.h: the batch form of the call above) --
 * index into the target as set (a tie: the lowest index) and the fp32 squared distance (FLANN L2_Simple order), no gate.  The queries
 * are sorted along the curve like a source cloud and searched with the same exact pruned kernel; no covariances are computed for them.
 * This serves the search object's queries that are NOT the transformed source points (the per-query host scan of round 5 cost 500 k
 * distance evaluations per query on a submap).  The next align / linearize of the handle sets its own pair up again. */
int apdgicp_nearest_neighbours_of(apdgicp_handle* h, const float* queries_xyz, int64_t n, int64_t stride_bytes, int32_t* index, float* sq_dist);
/* the points of the source / target cloud as set, n x {x, y, z} floats in the caller's order, into host memory (the fall-back of
 * that search object for a query that is not one of the transformed source points needs the target of a device-resident submap) */
int apdgicp_get_points(apdgicp_handle* h, int which, float* out_xyz, int64_t n);
int apdgicp_synchronize(apdgicp_handle* h);
/* Ordering against the stream that PRODUCED device-resident inputs.  The handle's streams are non-blocking: without this
 * call nothing orders a kernel that still writes the cloud (on the caller's stream) against the handle's pack kernel.
 * Everything queued on `producer_stream` (a hipStream_t; NULL = the legacy default stream) before this call completes
 * before anything the handle enqueues afterwards starts; the host does not wait.  The other direction is the caller's:
 * a device buffer handed to set_source/set_target/batch_set_cloud(s) may be freed or overwritten only after the handle
 * has passed the call (apdgicp_synchronize / the align that follows / batch_align_collect of the batch that used it). */
int apdgicp_wait_producer(apdgicp_handle* h, void* producer_stream);
/* the hipStream_t the handle enqueues on (its own, or the one given to apdgicp_create): for callers that order their own
 * work -- or their timing events -- against the handle without a host-side wait.  Every call of a handle ends with all
 * of its work joined back onto this stream. */
int apdgicp_get_stream(apdgicp_handle* h, void** stream);

/* ------------------------------------------------------------------ batched registrations
 * Independent (source, target) pairs -- loop-closure candidates (loop_detector.cpp:222-236,404-423)
 * or one scan against several keyframes -- solved concurrently on one GPU.  Clouds are registered
 * once and referenced by index, so a cloud shared by many pairs has its covariances computed once. */
typedef struct {
  int32_t source_cloud;
  int32_t target_cloud;
  float guess[16];            /* column-major */
} apdgicp_pair;

int apdgicp_batch_create(const apdgicp_params* p, int device, void* stream, apdgicp_batch** out);
int apdgicp_batch_destroy(apdgicp_batch* b);
int apdgicp_batch_set_params(apdgicp_batch* b, const apdgicp_params* p);
/* drops all clouds (and their covariances) */
int apdgicp_batch_clear(apdgicp_batch* b);
/* returns the cloud's index (>= 0) or a negative status */
int apdgicp_batch_add_cloud(apdgicp_batch* b, const float* xyz, int64_t n, int64_t stride_bytes, int on_device);
/* sets cloud slot `index` (>= 0; slots need not be contiguous: a caller that keeps several batches in flight gives each its own
 * range), reusing the slot's device buffers; its covariances are recomputed by the next align */
int apdgicp_batch_set_cloud(apdgicp_batch* b, int32_t index, const float* xyz, int64_t n, int64_t stride_bytes, int on_device);
/* sets clouds first_index .. first_index+count-1 in one call: xyz[i] / n[i] describe cloud first_index+i, all with the same stride.
 * on_device: one pack launch.  Host clouds (four or more, 32 k points or more in all): packed by a few host threads
 * (APDGICP_HOST_THREADS, default 4, the caller included; ONE pool per process, sized when first used) into one pinned region, ONE asynchronous copy, one pack launch; the
 * caller's buffers are free on return */
int apdgicp_batch_set_clouds(apdgicp_batch* b, int32_t first_index, int32_t count, const float* const* xyz, const int64_t* n,
                             int64_t stride_bytes, int on_device);
/* covariances of every cloud that does not have them yet (align does this lazily as well) */
int apdgicp_batch_compute_covariances(apdgicp_batch* b);
/* aligns all pairs; results[i] belongs to pairs[i].  `results` is host memory. */
int apdgicp_batch_align(apdgicp_batch* b, const apdgicp_pair* pairs, int64_t n_pairs, apdgicp_result* results);
/* same, but results stay on the device (n_pairs x sizeof(apdgicp_result) bytes at *d_results,
 * owned by the batch, valid until the next align) and the call does not wait: for callers that
 * gather results with RCCL.  apdgicp_batch_synchronize() waits for the stream. */
int apdgicp_batch_align_async(apdgicp_batch* b, const apdgicp_pair* pairs, int64_t n_pairs, void** d_results);
int apdgicp_batch_synchronize(apdgicp_batch* b);
int apdgicp_batch_wait_producer(apdgicp_batch* b, void* producer_stream);   /* see apdgicp_wait_producer */
int apdgicp_batch_get_stream(apdgicp_batch* b, void** stream);              /* see apdgicp_get_stream */
/* A stream of batches (one per keyframe, loop_detector.cpp:222-236) with several of them in flight on ONE handle and ONE host
 * thread: enqueue prepares the clouds set since the last call (packing, sorting, covariances), hands the batch to the device and
 * returns without waiting; collect(ticket) waits for that batch, reports its error if it had one, and hands out its records:
 * *d_results (device, n_pairs x sizeof(apdgicp_result)) and/or host_results (either may be NULL).  Collecting is optional.
 *   Gauss-Newton (the run length is known): every tick and the final poll are enqueued at once; TWO batches may be in flight, the
 *   next batch may reuse the same cloud slots (stream order), and a ticket stays collectable until the SECOND enqueue after its own.
 *   Levenberg-Marquardt (the reference's default, L:17; the run length is data dependent, L:64-76): the pairs of up to TWENTY-FOUR
 *   batches (APDGICP_POOL_LANES, at most 32; apdgicp_batch_is_pooled reports the number) share one pool of pair slots on the device; every optimiser tick is one launch over the pairs of all batches that
 *   still run, pairs leave as they converge and the pairs of the next batch join between two ticks, so a batch is never held
 *   by the slowest pair of another one and the GPU never waits for the host (ticks are enqueued two chunks ahead by whichever
 *   call of the handle is running; collect pumps until its batch is done).  A ticket stays collectable until its lane is
 *   needed again: the twenty-fourth enqueue after its own at the latest (a batch with more pairs or larger clouds than any before makes
 *   the pool lay itself out anew; the device record pointer of an EARLIER ticket collected after that is a copy of its
 *   host records, not a view of the pool).  A cloud slot referenced by a batch in flight must not be
 *   replaced -- set_cloud(s) on such a slot first waits for that batch -- so callers that want overlap give consecutive batches
 *   disjoint slot ranges (keyframe clouds that stay registered are shared freely).  If a covariance launch raises the device
 *   error flag, every batch in flight at that moment fails at its collect; the handle stays usable.
 *   APDGICP_LM_POOL=0 in the environment selects the round-2 host-polled loop instead (the cross-check of tests/test_lm_pool.py). */
int apdgicp_batch_align_enqueue(apdgicp_batch* b, const apdgicp_pair* pairs, int64_t n_pairs, uint64_t* ticket);
/* Serves the Levenberg-Marquardt pair pool without waiting: reads the polls that have arrived and keeps the chunks of ticks
 * enqueued ahead.  Every call of the handle does this anyway; a caller that goes away for more than a few hundred microseconds
 * between calls while batches are in flight (a worker thread waiting for its next job) calls this in between so that the GPU
 * does not run out of enqueued ticks.  No-op for Gauss-Newton handles and when nothing is in flight. */
int apdgicp_batch_pump(apdgicp_batch* b);
/* > 0 when apdgicp_batch_align_enqueue runs batches through the pair pool with the handle's current parameters (Levenberg-Marquardt,
 * pruned search, APDGICP_LM_POOL != 0): the number of batches that may be in flight on this one handle (24 unless
 * APDGICP_POOL_LANES says otherwise); 0: two record buffers, one batch at a time per handle for LM.  For callers that choose their
 * pipelining accordingly (ShardedBatchAlignerHip). */
int apdgicp_batch_is_pooled(apdgicp_batch* b);
/* A batch runs as up to three pair groups on three HIP streams (about 8 pairs per group), which is the best a single handle
 * can do.  A caller that keeps several HANDLES busy at once -- batch s on handle s % 3, each enqueued before the previous
 * ones are collected -- does better with one group (= one stream, larger launches) per handle: one handle's covariance
 * phase then fills the latency gaps of the others' optimiser ticks (bench.py: 1.70 -> 1.30 ms per batch of 32). */
int apdgicp_batch_set_pair_groups(apdgicp_batch* b, int max_groups);
int apdgicp_batch_align_collect(apdgicp_batch* b, uint64_t ticket, void** d_results, apdgicp_result* host_results);
/* pcl getFitnessScore(max_range) of every pair at the given poses (T: n_pairs x 16 floats, column-major;
 * NULL = the poses found by the last align of the same pair list): mean squared nearest-neighbour distance of the
 * transformed source over the points with squared distance <= max_range, DBL_MAX when none qualifies.  One NN launch
 * for the whole batch -- the per-candidate getFitnessScore of LoopDetector::matching (loop_detector.cpp:415).
 * inliers may be NULL. */
int apdgicp_batch_fitness(apdgicp_batch* b, const apdgicp_pair* pairs, int64_t n_pairs, const float* T, double max_range,
                          double* scores, int64_t* inliers);
/* copies the n_pairs result records of the last align into caller memory (device pointer when
 * dst_on_device != 0, e.g. a tensor that RCCL will all-gather) and waits for the copy */
int apdgicp_batch_copy_results(apdgicp_batch* b, void* dst, int64_t n_pairs, int dst_on_device);
/* Measurement hooks (bench.py's roofline leg).  With profiling enabled, launches of the dominant kernel (the
 * nearest-neighbour search) carry their own start/stop events (hipExtLaunchKernelGGL: the kernel's begin and end
 * timestamps on the stream it runs on); last_nn_time returns their summed milliseconds and the launch count for
 * the last align; last_ticks returns the number of state-machine ticks and the launch shape that was used. */
int apdgicp_batch_set_profiling(apdgicp_batch* b, int enable);
int apdgicp_batch_last_nn_time(apdgicp_batch* b, double* total_ms, int64_t* launches);
/* same, plus the number of pairs the timed launches covered (a launch covers one pair group; only every 10th tick is
 * timed, with a phase rotating from align to align, because timing every launch costs ~5 % of a step).  Pooled LM batches:
 * the timed launches (the first tick of every 10th chunk, per list slice) belong to no single batch -- the call returns what
 * has been harvested since the previous call and resets it; pairs_covered counts the pairs really on the list, not the slots */
int apdgicp_batch_last_nn_profile(apdgicp_batch* b, double* total_ms, int64_t* launches, int64_t* pairs_covered);
int apdgicp_batch_last_ticks(apdgicp_batch* b, int* ticks, int* nn_sources_per_lane, int* nn_target_splits);
/* name of the nearest-neighbour kernel the last launch used (e.g. "k_nn_compact<4>", "(k_nn_pruned<1, 8>)", "k_nn_partial<4>") */
int apdgicp_batch_last_nn_kernel(apdgicp_batch* b, char* name, int capacity);
/* pruning diagnostics, collected only when the environment has APDGICP_STATS=1 (else zeros); reading
 * resets them.  [0..3] nearest neighbour: groups scanned, chunks tested, chunks scanned, waves;
 * [4..9] covariance k-NN: groups loaded, (NN: batches of 64 group boxes visited), (NN: points that kept their neighbour without a search), waves sampled, list tightenings, (query, group) steps;
 * [10..15] sampled phase timers (s_memtime ticks) of whichever of the two kernels ran last (tools/prune_stats.py, tools/knn_time.py) */
/* Measurement: what the pair pool of a Levenberg-Marquardt batch handle has enqueued since it was created -- chunks (polls), ticks, and
 * slot-ticks = the sum over the tick launches of the pair slots they covered (grid y; slots behind the end of a list execute nothing).
 * bench.py turns per-slot counter profiles into a per-batch figure with it (the LM line's valu_busy).  Zeros for a handle without a pool. */
int apdgicp_batch_pool_counters(apdgicp_batch* b, int64_t* chunks, int64_t* ticks, int64_t* slot_ticks);
int apdgicp_batch_debug_stats(apdgicp_batch* b, unsigned long long out[16]);
/* APDGICP_STATS=2 and a library built with -DAPD_BLOCK_TIMELINE (a diagnostics variant: tools/build_variant.py) only: {start, end} (100 MHz wall-clock ticks) and the index of every block of the LAST one-pair dense search launch
 * (k_nn_pruned), three words per block, up to 8192 blocks; reading resets.  For tools/c5_blocks.py: where the time of a 100k x 500k
 * iteration goes -- the blocks' own durations or the order they are dealt in. */
int apdgicp_batch_debug_block_timeline(apdgicp_batch* b, unsigned long long* out, int64_t capacity_blocks, int64_t* n_blocks);

/* ------------------------------------------------------------------ scan-to-submap target assembly
 * The step in front of registration_s2m->setInputTarget in scan-to-map mode
 * (scan_matching_odometry_nodelet.cpp:606-618): the clouds of the last <= max_submap_frames keyframes are
 * transformed by their poses relative to the newest keyframe (pcl::transformPointCloud with a Matrix4d),
 * concatenated, and downsampled by downsample() (:412-422) -- pcl::VoxelGrid with the configured leaf
 * (preprocessing_nodelet.cpp:137-144, "VOXELGRID").  Everything stays on the device; the result can be handed to
 * apdgicp_set_target / apdgicp_batch_set_cloud as a device pointer (16-byte stride). */
typedef struct apdgicp_submap apdgicp_submap;
int apdgicp_submap_create(int device, void* stream, apdgicp_submap** out);
int apdgicp_submap_destroy(apdgicp_submap* s);
/* xyz[c]: first coordinate of cloud c (n_points[c] points, stride_bytes apart, host or device memory as on_device says);
 * intensity_offset_bytes: distance from a point's x to its intensity field (16 for pcl::PointXYZI), < 0: none (0 is kept);
 * rel_poses: n_clouds x 16 doubles, column-major 4x4 (keyframes[i].odom^-1 * keyframes.back().odom, :609), NULL = identity;
 * leaf: voxel size per axis (downsample_resolution), NULL or leaf[0] <= 0: no downsampling (downsample_method NONE);
 * n_out: number of points of the assembled cloud.  Non-finite points are skipped by the voxel filter, as PCL does for
 * non-dense clouds.  A leaf too small for the extent (the voxel index would overflow int32): pcl::VoxelGrid warns ("Leaf size is
 * too small for the input dataset") and returns its input unfiltered -- so does this call: the transformed, concatenated cloud,
 * the warning on stderr and in apdgicp_last_error(), status 0. */
int apdgicp_submap_assemble(apdgicp_submap* s, int n_clouds, const void* const* xyz, const int64_t* n_points, int64_t stride_bytes,
                            int64_t intensity_offset_bytes, int on_device, const double* rel_poses, const float* leaf, int64_t* n_out);
/* device pointer to the last assembled cloud: n points of {x, y, z, intensity} floats, valid until the next assemble */
int apdgicp_submap_points(apdgicp_submap* s, const float** device_xyzi, int64_t* n);
/* copies the last assembled cloud ({x, y, z, intensity} per point) into caller memory */
int apdgicp_submap_copy(apdgicp_submap* s, float* dst_xyzi, int64_t capacity_points, int dst_on_device);

#ifdef __cplusplus
}
#endif
#endif /* APDGICP_HIP_H */
