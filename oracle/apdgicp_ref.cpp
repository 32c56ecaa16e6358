// ORACLE (test infrastructure, NOT product code) -- C++17/OpenMP CPU restatement of RIV-SLAM's
// APD-GICP scan matcher (fast_gicp::FastAPDGICP + fast_gicp::LsqRegistration).
//
// PARITY UNPINNED: the reference itself cannot be compiled here (it needs PCL, Eigen, FLANN,
// Boost -- none are in the image) and its own tests never touch FastAPDGICP
// (fast_apdgicp/src/test/gicp_test.cpp:103-124).  This file follows the reference statement by
// statement and is cross-checked against an independent numpy restatement (oracle/apdgicp_np.py);
// the two must agree before anything on the GPU is compared with either.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
// It is also the reported CPU baseline ("kind": "port"): same loop structure as the reference
// (#pragma omp parallel for schedule(guided, 8), per-thread H/b slots, exact kd-tree NN).
//
// Citations are relative to /root/reference/fast_apdgicp/include/fast_gicp/ :
//   A = gicp/impl/fast_apdgicp_impl.hpp, L = gicp/impl/lsq_registration_impl.hpp, S = so3/so3.hpp
//
// Third-party arithmetic restated (un-vendored, un-pinned in the reference): FLANN L2_Simple<float>
// (fp32 (a-b)^2 accumulation over x,y,z, exact k-NN; ties -> lower index here), Eigen
// Isometry3f*Vector4f ((m0*x+m1*y)+(m2*z+m3) as Eigen >= 3.3 sums it, or the linear chain of Eigen 3.2: see xf_row; fp32, no
// FMA: reference is built -msse4.2 only),
// Eigen JacobiSVD of a symmetric PSD 3x3 (== symmetric eigen-decomposition, descending),
// Matrix4d::inverse of blkdiag(C,1), LDLT<6x6> (diagonal-pivoted), Quaterniond::toRotationMatrix.
//
// Build: g++ -O2 -fopenmp -ffp-contract=off -shared -fPIC (see oracle/Makefile).

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

// glibc's generic atan2f (fdlibm) restated once for kernels and checker: with the C library's own atan2f the checker's angles
// (A:168,172-173) would be those of whatever libm the TEST box has.  tests/test_atan2f.py pins the header bit for bit to glibc.
#include "../include/apd_atan2f.h"

namespace {

struct RefParams {  // independent mirror of include/apdgicp_hip.h:apdgicp_params
  int32_t k_correspondences;
  int32_t max_iterations;
  int32_t lm_max_iterations;
  int32_t optimizer;       // 0 = LevenbergMarquardt, 1 = GaussNewton (L:17, lsq_registration.hpp:13)
  int32_t regularization;  // gicp_settings.hpp:6 NONE, MIN_EIG, NORMALIZED_MIN_EIG, PLANE, FROBENIUS
  int32_t flags;           // (bit 3: the product's opt-in algebraic sensor model, see update_correspondences) bit 0: plain GICP (no cov_dist), gicp/impl/fast_gicp_impl.hpp update_correspondences;
                           // bit 1: T*p summed as a linear chain (Eigen 3.2) instead of pairwise (Eigen >= 3.3), see xf_row
  double max_correspondence_distance;
  double transformation_epsilon;
  double rotation_epsilon;
  double lm_init_lambda_factor;
  double distance_variance;
  double azimuth_variance_deg;
  double elevation_variance_deg;
};

// ------------------------------------------------------------------ small fixed-size algebra
struct M3 { double m[3][3]; };
struct M4 { double m[4][4]; };  // row-major storage, used as a rigid transform

inline M3 mul(const M3& a, const M3& b) {
  M3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
  return r;
}
inline M3 transpose(const M3& a) {
  M3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r.m[i][j] = a.m[j][i];
  return r;
}
inline M3 add(const M3& a, const M3& b) {
  M3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r.m[i][j] = a.m[i][j] + b.m[i][j];
  return r;
}
inline M3 inverse(const M3& a) {
  const double c00 = a.m[1][1] * a.m[2][2] - a.m[1][2] * a.m[2][1];
  const double c01 = a.m[1][2] * a.m[2][0] - a.m[1][0] * a.m[2][2];
  const double c02 = a.m[1][0] * a.m[2][1] - a.m[1][1] * a.m[2][0];
  const double det = a.m[0][0] * c00 + a.m[0][1] * c01 + a.m[0][2] * c02;
  const double id = 1.0 / det;
  M3 r;
  r.m[0][0] = c00 * id;
  r.m[1][0] = c01 * id;
  r.m[2][0] = c02 * id;
  r.m[0][1] = (a.m[0][2] * a.m[2][1] - a.m[0][1] * a.m[2][2]) * id;
  r.m[1][1] = (a.m[0][0] * a.m[2][2] - a.m[0][2] * a.m[2][0]) * id;
  r.m[2][1] = (a.m[0][1] * a.m[2][0] - a.m[0][0] * a.m[2][1]) * id;
  r.m[0][2] = (a.m[0][1] * a.m[1][2] - a.m[0][2] * a.m[1][1]) * id;
  r.m[1][2] = (a.m[0][2] * a.m[1][0] - a.m[0][0] * a.m[1][2]) * id;
  r.m[2][2] = (a.m[0][0] * a.m[1][1] - a.m[0][1] * a.m[1][0]) * id;
  return r;
}
inline double frob(const M3& a) {
  double s = 0;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) s += a.m[i][j] * a.m[i][j];
  return std::sqrt(s);
}

// Symmetric 3x3 eigen-decomposition by cyclic Jacobi; eigenvalues descending, columns of U.
// Stands in for Eigen::JacobiSVD<Matrix3d>(cov, FullU|FullV) on a symmetric PSD input (A:337).
void sym_eig3(const M3& a_in, double w[3], M3& U) {
  double a[3][3];
  std::memcpy(a, a_in.m, sizeof(a));
  double v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int sweep = 0; sweep < 64; sweep++) {
    const double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
    const double diag = a[0][0] * a[0][0] + a[1][1] * a[1][1] + a[2][2] * a[2][2];
    if (off <= 1e-40 * diag || off == 0.0) break;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        if (a[p][q] == 0.0) continue;
        const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; k++) {  // A <- A * G
          const double akp = a[k][p], akq = a[k][q];
          a[k][p] = c * akp - s * akq;
          a[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; k++) {  // A <- G^T * A
          const double apk = a[p][k], aqk = a[q][k];
          a[p][k] = c * apk - s * aqk;
          a[q][k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; k++) {
          const double vkp = v[k][p], vkq = v[k][q];
          v[k][p] = c * vkp - s * vkq;
          v[k][q] = s * vkp + c * vkq;
        }
      }
  }
  int order[3] = {0, 1, 2};
  std::sort(order, order + 3, [&](int x, int y) { return a[x][x] > a[y][y]; });
  for (int j = 0; j < 3; j++) {
    w[j] = a[order[j]][order[j]];
    for (int i = 0; i < 3; i++) U.m[i][j] = v[i][order[j]];
  }
}

// 6x6 LDL^T with diagonal pivoting (the algorithm behind Eigen::LDLT, L:112,137), then solve.
void ldlt6_solve(const double Hin[36] /*col-major, symmetric*/, const double rhs[6], double x[6]) {
  double A[6][6];
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) A[i][j] = Hin[i + 6 * j];
  int perm[6];
  for (int k = 0; k < 6; k++) perm[k] = k;
  for (int k = 0; k < 6; k++) {
    int piv = k;
    double best = std::fabs(A[k][k]);
    for (int i = k + 1; i < 6; i++)
      if (std::fabs(A[i][i]) > best) best = std::fabs(A[i][i]), piv = i;
    if (piv != k) {  // symmetric row+column swap
      for (int j = 0; j < 6; j++) std::swap(A[k][j], A[piv][j]);
      for (int i = 0; i < 6; i++) std::swap(A[i][k], A[i][piv]);
      std::swap(perm[k], perm[piv]);
    }
    const double d = A[k][k];
    if (d == 0.0) continue;
    for (int i = k + 1; i < 6; i++) A[i][k] /= d;
    for (int i = k + 1; i < 6; i++)
      for (int j = k + 1; j <= i; j++) {
        A[i][j] -= A[i][k] * d * A[j][k];
        A[j][i] = A[i][j];
      }
  }
  double y[6];
  for (int i = 0; i < 6; i++) y[i] = rhs[perm[i]];
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < i; j++) y[i] -= A[i][j] * y[j];
  const double tol = 1.0 / DBL_MAX;  // Eigen::LDLT::_solve_impl pseudo-inverse threshold
  for (int i = 0; i < 6; i++) y[i] = std::fabs(A[i][i]) > tol ? y[i] / A[i][i] : 0.0;
  for (int i = 5; i >= 0; i--)
    for (int j = i + 1; j < 6; j++) y[i] -= A[j][i] * y[j];
  for (int i = 0; i < 6; i++) x[perm[i]] = y[i];
}

// S:59-78 followed by Eigen::Quaterniond::toRotationMatrix()
void so3_exp_matrix(const double om[3], double R[3][3]) {
  const double theta_sq = om[0] * om[0] + om[1] * om[1] + om[2] * om[2];
  double imag, real;
  if (theta_sq < 1e-10) {
    const double theta_quad = theta_sq * theta_sq;
    imag = 0.5 - 1.0 / 48.0 * theta_sq + 1.0 / 3840.0 * theta_quad;
    real = 1.0 - 1.0 / 8.0 * theta_sq + 1.0 / 384.0 * theta_quad;
  } else {
    const double theta = std::sqrt(theta_sq), half = 0.5 * theta;
    imag = std::sin(half) / theta;
    real = std::cos(half);
  }
  const double w = real, x = imag * om[0], y = imag * om[1], z = imag * om[2];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0][0] = 1 - (tyy + tzz); R[0][1] = txy - twz;       R[0][2] = txz + twy;
  R[1][0] = txy + twz;       R[1][1] = 1 - (txx + tzz); R[1][2] = tyz - twx;
  R[2][0] = txz - twy;       R[2][1] = tyz + twx;       R[2][2] = 1 - (txx + tyy);
}

M4 rigid_mul(const M4& a, const M4& b) {  // Isometry3d * Isometry3d
  M4 r;
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
    r.m[i][3] = a.m[i][0] * b.m[0][3] + a.m[i][1] * b.m[1][3] + a.m[i][2] * b.m[2][3] + a.m[i][3];
  }
  r.m[3][0] = r.m[3][1] = r.m[3][2] = 0;
  r.m[3][3] = 1;
  return r;
}

// ------------------------------------------------------------------ exact kd-tree (FLANN stand-in)
struct F3 { float x, y, z; };

inline float sqdist(const F3& a, const F3& b) {  // FLANN L2_Simple<float>, dims 0..2
  float d = a.x - b.x;
  float r = d * d;
  d = a.y - b.y;
  r += d * d;
  d = a.z - b.z;
  r += d * d;
  return r;
}

struct Cand {
  float d;
  int idx;
};
inline bool cand_less(const Cand& a, const Cand& b) { return a.d < b.d || (a.d == b.d && a.idx < b.idx); }

struct KdTree {
  struct Node {
    int left, right;  // children (internal) ; for leaves: [begin, end) into order
    int dim;          // -1 = leaf
    float split;
  };
  const F3* pts = nullptr;
  int n = 0;
  std::vector<int> order;
  std::vector<Node> nodes;
  static constexpr int kLeaf = 12;

  void build(const F3* p, int count) {
    pts = p;
    n = count;
    order.resize(n);
    std::iota(order.begin(), order.end(), 0);
    nodes.clear();
    nodes.reserve(2 * (n / kLeaf + 2));
    if (n > 0) build_rec(0, n);
  }
  int build_rec(int b, int e) {
    const int id = (int)nodes.size();
    nodes.push_back(Node{b, e, -1, 0.f});
    if (e - b <= kLeaf) return id;
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = b; i < e; i++) {
      const float c[3] = {pts[order[i]].x, pts[order[i]].y, pts[order[i]].z};
      for (int d = 0; d < 3; d++) lo[d] = std::min(lo[d], c[d]), hi[d] = std::max(hi[d], c[d]);
    }
    int dim = 0;
    for (int d = 1; d < 3; d++)
      if (hi[d] - lo[d] > hi[dim] - lo[dim]) dim = d;
    if (!(hi[dim] > lo[dim])) return id;  // all identical: keep as a (large) leaf
    const int mid = (b + e) / 2;
    auto coord = [&](int i) { return dim == 0 ? pts[i].x : dim == 1 ? pts[i].y : pts[i].z; };
    std::nth_element(order.begin() + b, order.begin() + mid, order.begin() + e,
                     [&](int x, int y) { return coord(x) < coord(y); });
    const float split = coord(order[mid]);
    const int l = build_rec(b, mid);
    const int r = build_rec(mid, e);
    nodes[id].left = l;
    nodes[id].right = r;
    nodes[id].dim = dim;
    nodes[id].split = split;
    return id;
  }
  // heap = max-heap on (d, idx) of size <= k
  void search(const F3& q, int k, std::vector<Cand>& heap) const {
    heap.clear();
    if (n > 0) search_rec(0, q, k, heap);
    std::sort_heap(heap.begin(), heap.end(), cand_less);
  }
  void search_rec(int id, const F3& q, int k, std::vector<Cand>& heap) const {
    const Node& nd = nodes[id];
    if (nd.dim < 0) {
      for (int i = nd.left; i < nd.right; i++) {
        const Cand c{sqdist(q, pts[order[i]]), order[i]};
        if ((int)heap.size() < k) {
          heap.push_back(c);
          std::push_heap(heap.begin(), heap.end(), cand_less);
        } else if (cand_less(c, heap.front())) {
          std::pop_heap(heap.begin(), heap.end(), cand_less);
          heap.back() = c;
          std::push_heap(heap.begin(), heap.end(), cand_less);
        }
      }
      return;
    }
    const float qc = nd.dim == 0 ? q.x : nd.dim == 1 ? q.y : q.z;
    const float diff = qc - nd.split;
    const int nearc = diff < 0 ? nd.left : nd.right, farc = diff < 0 ? nd.right : nd.left;
    search_rec(nearc, q, k, heap);
    // fp32 plane distance is a lower bound of every fp32 sqdist behind the plane (rounding is
    // monotone); on equality a lower-index tie may hide there, so only prune on strict >.
    const float pd = diff * diff;
    if ((int)heap.size() < k || !(pd > heap.front().d)) search_rec(farc, q, k, heap);
  }
};

// ------------------------------------------------------------------ the registration object
struct Cloud {
  std::vector<F3> pts;
  std::vector<M3> covs;  // top-left 3x3 of the reference's Matrix4d (rest is zero)
  KdTree tree;
  bool tree_valid = false;
};

struct Ref {
  RefParams p;
  int num_threads = 1;
  Cloud src, tgt;
  std::vector<int> corr;
  std::vector<float> sqd;
  std::vector<M3> maha;
  double lm_lambda = -1.0;
  double final_hessian[36];
  int n_linearize = 0, n_compute_error = 0;
  // trace of the last align: per LM trial (L:136-164) the lambda it was solved with and its rho; per completed outer iteration
  // (L:67-76) the pose x0 behind it (column-major 4x4)
  std::vector<double> tr_lambda, tr_rho, tr_y0, tr_yi, tr_poses;
};

void ensure_tree(Cloud& c) {
  if (!c.tree_valid) {
    c.tree.build(c.pts.data(), (int)c.pts.size());
    c.tree_valid = true;
  }
}

// A:303-363
int calculate_covariances(Ref& r, Cloud& c) {
  const int n = (int)c.pts.size(), k = r.p.k_correspondences;
  if (n < k || k < 1) return -1;  // reference reads uninitialised neighbour columns (A:318-321)
  ensure_tree(c);
  c.covs.resize(n);
  const int reg = r.p.regularization;
  int bad = 0;
#pragma omp parallel for num_threads(r.num_threads) schedule(guided, 8)
  for (int i = 0; i < n; i++) {
    static thread_local std::vector<Cand> heap;  // one candidate list per thread, not one allocation per query
    heap.clear();
    heap.reserve(k + 1);
    c.tree.search(c.pts[i], k, heap);  // A:316 (query point itself is among the k)
    double mean[3] = {0, 0, 0};
    for (int j = 0; j < k; j++) {
      const F3& q = c.pts[heap[j].idx];
      mean[0] += (double)q.x, mean[1] += (double)q.y, mean[2] += (double)q.z;
    }
    for (int d = 0; d < 3; d++) mean[d] /= k;  // A:323
    M3 cov{};
    for (int j = 0; j < k; j++) {
      const F3& q = c.pts[heap[j].idx];
      const double v[3] = {(double)q.x - mean[0], (double)q.y - mean[1], (double)q.z - mean[2]};
      for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) cov.m[a][b] += v[a] * v[b];
    }
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) cov.m[a][b] /= k;  // A:324
    if (reg == 0) {                                   // NONE A:326-328
      c.covs[i] = cov;
    } else if (reg == 4) {                            // FROBENIUS A:329-335
      M3 C = cov;
      for (int d = 0; d < 3; d++) C.m[d][d] += 1e-3;
      M3 Ci = inverse(C);
      const double nf = frob(Ci);
      for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) Ci.m[a][b] /= nf;
      c.covs[i] = inverse(Ci);
    } else {                                          // A:337-357
      double w[3];
      M3 U;
      sym_eig3(cov, w, U);
      double vals[3];
      if (reg == 3) {                                 // PLANE
        vals[0] = 1, vals[1] = 1, vals[2] = 1e-3;
      } else if (reg == 1) {                          // MIN_EIG
        for (int d = 0; d < 3; d++) vals[d] = std::max(w[d], 1e-3);
      } else if (reg == 2) {                          // NORMALIZED_MIN_EIG
        const double mx = std::max(w[0], std::max(w[1], w[2]));
        for (int d = 0; d < 3; d++) vals[d] = std::max(w[d] / mx, 1e-3);
      } else {
#pragma omp atomic write
        bad = 1;                                      // reference abort()s, A:341-343
        vals[0] = vals[1] = vals[2] = 0;
      }
      M3 out{};
      for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++)
          out.m[a][b] = U.m[a][0] * vals[0] * U.m[b][0] + U.m[a][1] * vals[1] * U.m[b][1] + U.m[a][2] * vals[2] * U.m[b][2];
      c.covs[i] = out;
    }
  }
  return bad ? -2 : 0;
}

void load_T(const double Tcm[16], M4& T) {
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) T.m[i][j] = Tcm[i + 4 * j];
}

// A:149: `pt = trans.cast<float>() * input_->at(i).getVector4fMap()` -- an Isometry3f times a Vector4f whose 4th coefficient is 1.
// The fp32 operation order of this product belongs to Eigen (un-vendored, un-pinned), not to the reference:
//   * Eigen >= 3.3 (the 3.3.4 / 3.3.7 of the platforms README.md:5-7 names): Transform::operator* -> transform_right_product_impl
//     -> `T.affine() * other`, a 3x4 * 4x1 product -> CoeffBasedProductMode (3 rows are not a multiple of a Packet4f, a row of a
//     column-major block has inner stride 4: no packet path) -> coeff(i) = (lhs.row(i).transpose().cwiseProduct(rhs)).sum() ->
//     redux_novec_unroller<.., 0, 4>, which HALVES the range: (p0 + p1) + (p2 + p3) = (r0 x + r1 y) + (r2 z + t * 1).   [default]
//   * Eigen 3.2: product_coeff_impl<DefaultTraversal, 3> accumulates from the left: ((r0 x + r1 y) + r2 z) + t.    [flags bit 1]
// No FMA in either (the reference is built with -msse4.2 only, fast_apdgicp/CMakeLists.txt:11-13).  Read from the Eigen sources
// as remembered, not compiled here: tools/eigen_order_probe.cpp lets an integrator check it against THEIR Eigen.
// g_xf_order != 0 exists ONLY for tests/measure/transform_order_sensitivity.py, which quantifies how far the registered pose
// moves under still other orders (packet code with FMA, other associations).
int g_xf_order = 0;
inline float xf_row(const float* r, const F3& a, bool linear_chain) {
  switch (g_xf_order) {
    case 1: return std::fmaf(r[2], a.z, std::fmaf(r[1], a.y, r[0] * a.x)) + r[3];              // packet pmadd chain, translation added last
    case 2: return r[0] * a.x + (r[1] * a.y + (r[2] * a.z + r[3]));                            // accumulated from the last column
    case 3: return (r[0] * a.x + r[1] * a.y) + (r[2] * a.z + r[3]);                            // pairwise (forced)
    case 4: return std::fmaf(r[0], a.x, std::fmaf(r[1], a.y, std::fmaf(r[2], a.z, r[3])));     // fully fused, from the last column
    case 5: return std::fmaf(r[2], a.z, std::fmaf(r[1], a.y, std::fmaf(r[0], a.x, r[3])));     // fully fused, translation first
    case 6: return ((r[0] * a.x + r[1] * a.y) + r[2] * a.z) + r[3];                            // linear chain (forced)
    default: break;
  }
  if (linear_chain) return ((r[0] * a.x + r[1] * a.y) + r[2] * a.z) + r[3];
  return (r[0] * a.x + r[1] * a.y) + (r[2] * a.z + r[3]);
}

// A:133-194
void update_correspondences(Ref& r, const M4& T) {
  const int n = (int)r.src.pts.size();
  ensure_tree(r.tgt);
  r.corr.resize(n);
  r.sqd.resize(n);
  r.maha.resize(n);
  float Tf[3][4];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 4; j++) Tf[i][j] = (float)T.m[i][j];  // A:137
  M3 R;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) R.m[i][j] = T.m[i][j];
  const M3 Rt = transpose(R);
  const double thr2 = r.p.max_correspondence_distance * r.p.max_correspondence_distance;
  const double sin_az = std::sin(r.p.azimuth_variance_deg / 180 * M_PI);
  const double sin_el = std::sin(r.p.elevation_variance_deg / 180 * M_PI);
#pragma omp parallel for num_threads(r.num_threads) schedule(guided, 8)
  for (int i = 0; i < n; i++) {
    const F3& a = r.src.pts[i];
    F3 pt;  // A:149
    const bool lin = (r.p.flags & 2) != 0;
    pt.x = xf_row(Tf[0], a, lin), pt.y = xf_row(Tf[1], a, lin), pt.z = xf_row(Tf[2], a, lin);
    static thread_local std::vector<Cand> heap;
    heap.clear();
    heap.reserve(2);
    r.tgt.tree.search(pt, 1, heap);  // A:151
    r.sqd[i] = heap[0].d;            // A:153
    r.corr[i] = (double)heap[0].d < thr2 ? heap[0].idx : -1;  // A:156
    if (r.corr[i] < 0) continue;
    const M3& cov_A = r.src.covs[i];
    const M3& cov_B = r.tgt.covs[r.corr[i]];
    const double dist = std::sqrt((double)pt.x * pt.x + (double)pt.y * pt.y + (double)pt.z * pt.z);  // A:167
    const double aoa = (double)apd::apd_atan2f(pt.x, sqrtf(pt.y * pt.y + pt.z * pt.z));   // A:168 float overloads
    const double s_x = dist * r.p.distance_variance / 400;                          // A:169
    const double s_y = dist * sin_az / std::cos(aoa);                               // A:170
    const double s_z = dist * sin_el / std::cos(aoa);                               // A:171
    const double elevation = (double)apd::apd_atan2f(sqrtf(pt.x * pt.x + pt.y * pt.y), pt.z);  // A:172
    const double azimuth = (double)apd::apd_atan2f(pt.y, pt.x);                               // A:173
    double ce = std::cos(elevation), se = std::sin(elevation);
    double ca = std::cos(azimuth), sa = std::sin(azimuth);
    double s_y_ = s_y, s_z_ = s_z;
    if (r.p.flags & 8) {
      // NOT the reference: the checker of the product's opt-in APDGICP_FLAG_ALGEBRAIC_APD (include/apdgicp_hip.h) -- the same quantities as
      // ratios of the point's coordinates, in plain libm double arithmetic: cos(az) = x / rho, sin(az) = y / rho, sin(el) = rho / r,
      // cos(el) = z / r, 1 / cos(AoA) = r / sqrt(y^2 + z^2) (at most 1 / |cos((double)(float)(pi/2))|); atan2(0, 0) = 0 on the axes
      const double x = pt.x, y = pt.y, z = pt.z;
      const double rho = std::sqrt(x * x + y * y), yz = std::sqrt(y * y + z * z);
      const double inv_cos = std::fmin(yz > 0 ? dist / yz : INFINITY, 1.0 / 4.371138828673793e-08);
      s_y_ = dist * inv_cos * sin_az, s_z_ = dist * inv_cos * sin_el;
      se = dist > 0 ? rho / dist : 0.0, ce = dist > 0 ? z / dist : 0.0;
      ca = rho > 0 ? x / rho : 1.0, sa = rho > 0 ? y / rho : 0.0;
    }
    M3 Ry{{{ce, 0, se}, {0, 1, 0}, {-se, 0, ce}}};
    M3 Rz{{{ca, -sa, 0}, {sa, ca, 0}, {0, 0, 1}}};
    M3 Rot = mul(Rz, Ry);  // A:174-177
    M3 Am;                 // A = R * S, A:181
    const double s[3] = {s_x, s_y_, s_z_};
    for (int p = 0; p < 3; p++)
      for (int q = 0; q < 3; q++) Am.m[p][q] = Rot.m[p][q] * s[q];
    M3 cov_r = mul(Am, transpose(Am));  // A:182
    if (r.p.flags & 1) cov_r = M3{};      // upstream FastGICP: RCR = cov_B + T cov_A T^T
    const M3 RCR = add(add(cov_B, cov_r), mul(mul(R, add(cov_A, cov_r)), Rt));  // A:188
    r.maha[i] = inverse(RCR);                                                    // A:191-192
  }
}

// A:198-272 (H, b nullable) ; returns sum of errors
double linearize(Ref& r, const M4& T, double* H, double* b) {
  r.n_linearize++;
  update_correspondences(r, T);
  const int n = (int)r.src.pts.size();
  const int nt = std::max(1, r.num_threads);
  std::vector<double> Hs((size_t)nt * 36, 0.0), bs((size_t)nt * 6, 0.0);
  double sum_errors = 0.0;
  const bool want = H && b;
#pragma omp parallel for num_threads(r.num_threads) reduction(+ : sum_errors) schedule(guided, 8)
  for (int i = 0; i < n; i++) {
    const int j = r.corr[i];
    if (j < 0) continue;
    const F3& pa = r.src.pts[i];
    const F3& pb = r.tgt.pts[j];
    const double a[3] = {(double)pa.x, (double)pa.y, (double)pa.z};
    double Ta[3], e[3];
    for (int d = 0; d < 3; d++) Ta[d] = T.m[d][0] * a[0] + T.m[d][1] * a[1] + T.m[d][2] * a[2] + T.m[d][3];
    e[0] = (double)pb.x - Ta[0], e[1] = (double)pb.y - Ta[1], e[2] = (double)pb.z - Ta[2];  // A:236-237
    const M3& M = r.maha[i];
    double Me[3];
    for (int d = 0; d < 3; d++) Me[d] = M.m[d][0] * e[0] + M.m[d][1] * e[1] + M.m[d][2] * e[2];
    sum_errors += e[0] * Me[0] + e[1] * Me[1] + e[2] * Me[2];  // A:240
    if (!want) continue;
    double J[3][6] = {{0, -Ta[2], Ta[1], -1, 0, 0}, {Ta[2], 0, -Ta[0], 0, -1, 0}, {-Ta[1], Ta[0], 0, 0, 0, -1}};  // A:248-250, S:21-31
    double MJ[3][6];
    for (int p = 0; p < 3; p++)
      for (int q = 0; q < 6; q++) MJ[p][q] = M.m[p][0] * J[0][q] + M.m[p][1] * J[1][q] + M.m[p][2] * J[2][q];
#ifdef _OPENMP
    const int tid = omp_get_thread_num();
#else
    const int tid = 0;
#endif
    double* Ht = &Hs[(size_t)tid * 36];
    double* bt = &bs[(size_t)tid * 6];
    for (int p = 0; p < 6; p++) {
      for (int q = 0; q < 6; q++) Ht[p + 6 * q] += J[0][p] * MJ[0][q] + J[1][p] * MJ[1][q] + J[2][p] * MJ[2][q];  // A:253,257
      bt[p] += J[0][p] * Me[0] + J[1][p] * Me[1] + J[2][p] * Me[2];                                                 // A:254,258
    }
  }
  if (want) {  // A:262-269
    std::fill(H, H + 36, 0.0);
    std::fill(b, b + 6, 0.0);
    for (int t = 0; t < nt; t++) {
      for (int q = 0; q < 36; q++) H[q] += Hs[(size_t)t * 36 + q];
      for (int q = 0; q < 6; q++) b[q] += bs[(size_t)t * 6 + q];
    }
  }
  return sum_errors;
}

// A:275-298
double compute_error(Ref& r, const M4& T) {
  r.n_compute_error++;
  const int n = (int)r.src.pts.size();
  double sum_errors = 0.0;
#pragma omp parallel for num_threads(r.num_threads) reduction(+ : sum_errors) schedule(guided, 8)
  for (int i = 0; i < n; i++) {
    const int j = r.corr[i];
    if (j < 0) continue;
    const F3& pa = r.src.pts[i];
    const F3& pb = r.tgt.pts[j];
    const double a[3] = {(double)pa.x, (double)pa.y, (double)pa.z};
    double e[3];
    for (int d = 0; d < 3; d++) e[d] = T.m[d][0] * a[0] + T.m[d][1] * a[1] + T.m[d][2] * a[2] + T.m[d][3];
    e[0] = (double)pb.x - e[0], e[1] = (double)pb.y - e[1], e[2] = (double)pb.z - e[2];
    const M3& M = r.maha[i];
    double Me[3];
    for (int d = 0; d < 3; d++) Me[d] = M.m[d][0] * e[0] + M.m[d][1] * e[1] + M.m[d][2] * e[2];
    sum_errors += e[0] * Me[0] + e[1] * Me[1] + e[2] * Me[2];
  }
  return sum_errors;
}

// L:83-92
bool is_converged(const Ref& r, const M4& delta) {
  double rmax = 0, tmax = 0;
  bool nan = false;
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) {
      const double v = 1.0 / r.p.rotation_epsilon * std::fabs(delta.m[i][j] - (i == j ? 1.0 : 0.0));
      if (v != v) nan = true;
      rmax = std::max(rmax, v);
    }
    const double v = 1.0 / r.p.transformation_epsilon * std::fabs(delta.m[i][3]);
    if (v != v) nan = true;
    tmax = std::max(tmax, v);
  }
  if (nan) return false;
  return std::max(rmax, tmax) < 1;
}

void make_delta(const double d[6], M4& delta) {  // L:115-117 / L:140-142
  double R[3][3];
  so3_exp_matrix(d, R);
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) delta.m[i][j] = R[i][j];
    delta.m[i][3] = d[3 + i];
  }
  delta.m[3][0] = delta.m[3][1] = delta.m[3][2] = 0;
  delta.m[3][3] = 1;
}

// L:107-123
bool step_gn(Ref& r, M4& x0, M4& delta) {
  double H[36], b[6], nb[6], d[6];
  linearize(r, x0, H, b);
  for (int i = 0; i < 6; i++) nb[i] = -b[i];
  ldlt6_solve(H, nb, d);
  make_delta(d, delta);
  x0 = rigid_mul(delta, x0);
  std::memcpy(r.final_hessian, H, sizeof(H));
  return true;
}

// L:127-173
bool step_lm(Ref& r, M4& x0, M4& delta) {
  double H[36], b[6];
  const double y0 = linearize(r, x0, H, b);
  if (r.lm_lambda < 0.0) {
    double mx = 0;
    for (int i = 0; i < 6; i++) mx = std::max(mx, std::fabs(H[i + 6 * i]));
    r.lm_lambda = r.p.lm_init_lambda_factor * mx;
  }
  double nu = 2.0;
  for (int it = 0; it < r.p.lm_max_iterations; it++) {
    double Hl[36], nb[6], d[6];
    std::memcpy(Hl, H, sizeof(H));
    for (int i = 0; i < 6; i++) Hl[i + 6 * i] += r.lm_lambda, nb[i] = -b[i];
    ldlt6_solve(Hl, nb, d);
    make_delta(d, delta);
    const M4 xi = rigid_mul(delta, x0);
    const double yi = compute_error(r, xi);
    double den = 0;
    for (int i = 0; i < 6; i++) den += d[i] * (r.lm_lambda * d[i] - b[i]);
    const double rho = (y0 - yi) / den;
    r.tr_lambda.push_back(r.lm_lambda), r.tr_rho.push_back(rho), r.tr_y0.push_back(y0), r.tr_yi.push_back(yi);
    if (rho < 0) {
      if (is_converged(r, delta)) return true;
      r.lm_lambda = nu * r.lm_lambda;
      nu = 2 * nu;
      continue;
    }
    x0 = xi;
    r.lm_lambda = r.lm_lambda * std::max(1.0 / 3.0, 1 - std::pow(2 * rho - 1, 3));
    std::memcpy(r.final_hessian, H, sizeof(H));
    return true;
  }
  return false;
}

}  // namespace

extern "C" {

void* ref_create(const RefParams* p) {
  Ref* r = new Ref;
  r->p = *p;
#ifdef _OPENMP
  r->num_threads = omp_get_max_threads();  // A:15-19, setNumThreads(0) A:34-42
#endif
  for (int i = 0; i < 36; i++) r->final_hessian[i] = (i % 7 == 0) ? 1.0 : 0.0;  // L:23
  return r;
}
void ref_destroy(void* h) { delete (Ref*)h; }
// measurement only (tests/measure/transform_order_sensitivity.py): 0 = the reference order, see xf_row
void ref_set_transform_order(int order) { g_xf_order = order; }
void ref_set_params(void* h, const RefParams* p) { ((Ref*)h)->p = *p; }
int ref_set_num_threads(void* h, int n) {
  Ref* r = (Ref*)h;
#ifdef _OPENMP
  r->num_threads = n > 0 ? n : omp_get_max_threads();
#else
  r->num_threads = 1;
#endif
  return r->num_threads;
}
static void set_cloud(Cloud& c, const float* xyz, int n, int stride_floats) {
  c.pts.resize(n);
  for (int i = 0; i < n; i++) c.pts[i] = F3{xyz[(size_t)i * stride_floats], xyz[(size_t)i * stride_floats + 1], xyz[(size_t)i * stride_floats + 2]};
  c.covs.clear();
  c.tree_valid = false;
}
// A:90-98 / A:101-108 (kd-tree build included, as in the reference)
void ref_set_source(void* h, const float* xyz, int n, int stride_floats) {
  Ref* r = (Ref*)h;
  set_cloud(r->src, xyz, n, stride_floats);
  ensure_tree(r->src);
}
void ref_set_target(void* h, const float* xyz, int n, int stride_floats) {
  Ref* r = (Ref*)h;
  set_cloud(r->tgt, xyz, n, stride_floats);
  ensure_tree(r->tgt);
}
int ref_compute_covariances(void* h, int which) {
  Ref* r = (Ref*)h;
  return calculate_covariances(*r, which == 0 ? r->src : r->tgt);
}
int ref_get_covariances(void* h, int which, double* out9n) {  // row-major 3x3 per point
  Ref* r = (Ref*)h;
  const Cloud& c = which == 0 ? r->src : r->tgt;
  if (c.covs.size() != c.pts.size()) return -1;
  std::memcpy(out9n, c.covs.data(), c.covs.size() * sizeof(M3));
  return 0;
}
static int ensure_covs(Ref* r) {  // A:122-127
  int rc = 0;
  if (r->src.covs.size() != r->src.pts.size()) rc |= calculate_covariances(*r, r->src);
  if (r->tgt.covs.size() != r->tgt.pts.size()) rc |= calculate_covariances(*r, r->tgt);
  return rc;
}
// lsq_registration_impl.hpp:50-52 evaluateCost -> linearize.  T column-major 4x4 (Eigen layout).
int ref_linearize(void* h, const double* T16, double* H36, double* b6, double* cost) {
  Ref* r = (Ref*)h;
  if (ensure_covs(r)) return -1;
  M4 T;
  load_T(T16, T);
  *cost = linearize(*r, T, H36, b6);
  return 0;
}
int ref_compute_error(void* h, const double* T16, double* cost) {
  Ref* r = (Ref*)h;
  if (r->corr.size() != r->src.pts.size()) return -1;
  M4 T;
  load_T(T16, T);
  *cost = compute_error(*r, T);
  return 0;
}
int ref_get_correspondences(void* h, int* corr, float* sqd) {
  Ref* r = (Ref*)h;
  if (corr) std::memcpy(corr, r->corr.data(), r->corr.size() * sizeof(int));
  if (sqd) std::memcpy(sqd, r->sqd.data(), r->sqd.size() * sizeof(float));
  return (int)r->corr.size();
}
int ref_get_mahalanobis(void* h, double* out9n) {
  Ref* r = (Ref*)h;
  for (size_t i = 0; i < r->maha.size(); i++) {
    if (r->corr[i] < 0) std::memset(out9n + 9 * i, 0, 9 * sizeof(double));
    else std::memcpy(out9n + 9 * i, r->maha[i].m, 9 * sizeof(double));
  }
  return (int)r->maha.size();
}
// A:121-130 + L:55-80.  guess / out_T column-major 4x4 float.
// out_info[0..3] = converged, nr_iterations, n_linearize, n_compute_error
int ref_align(void* h, const float* guess16, float* out_T16, int* out_info) {
  Ref* r = (Ref*)h;
  if (ensure_covs(r)) return -1;
  M4 x0;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) x0.m[i][j] = (double)guess16[i + 4 * j];  // L:56
  r->lm_lambda = -1.0;  // L:58
  bool converged = false;
  int nr_iterations = 0;
  r->n_linearize = r->n_compute_error = 0;
  r->tr_lambda.clear(), r->tr_rho.clear(), r->tr_y0.clear(), r->tr_yi.clear(), r->tr_poses.clear();
  for (int i = 0; i < r->p.max_iterations && !converged; i++) {  // L:67-76
    nr_iterations = i;
    M4 delta;
    const bool ok = r->p.optimizer == 1 ? step_gn(*r, x0, delta) : step_lm(*r, x0, delta);
    if (!ok) break;  // "lm not converged!!"
    converged = is_converged(*r, delta);
    for (int c = 0; c < 4; c++)
      for (int q = 0; q < 4; q++) r->tr_poses.push_back(x0.m[q][c]);
  }
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) out_T16[i + 4 * j] = (float)x0.m[i][j];  // L:78
  out_info[0] = converged ? 1 : 0;
  out_info[1] = nr_iterations;
  out_info[2] = r->n_linearize;
  out_info[3] = r->n_compute_error;
  return 0;
}
// trace of the last ref_align: returns the number of LM trials; *n_poses the number of completed outer iterations.
// lambdas / rhos / y0s / yis: room for max_iterations * lm_max_iterations doubles, poses16: max_iterations x 16 (any may be null)
int ref_get_trace(void* h, double* lambdas, double* rhos, double* y0s, double* yis, double* poses16, int* n_poses) {
  Ref* r = (Ref*)h;
  if (lambdas) std::memcpy(lambdas, r->tr_lambda.data(), r->tr_lambda.size() * sizeof(double));
  if (rhos) std::memcpy(rhos, r->tr_rho.data(), r->tr_rho.size() * sizeof(double));
  if (y0s) std::memcpy(y0s, r->tr_y0.data(), r->tr_y0.size() * sizeof(double));
  if (yis) std::memcpy(yis, r->tr_yi.data(), r->tr_yi.size() * sizeof(double));
  if (poses16) std::memcpy(poses16, r->tr_poses.data(), r->tr_poses.size() * sizeof(double));
  if (n_poses) *n_poses = (int)(r->tr_poses.size() / 16);
  return (int)r->tr_lambda.size();
}
// the atan2f every angle of the sensor model goes through (include/apd_atan2f.h), for the tests
void ref_atan2f(const float* y, const float* x, float* out, long long n) {
  for (long long i = 0; i < n; i++) out[i] = apd::apd_atan2f(y[i], x[i]);
}
void ref_get_final_hessian(void* h, double* H36) { std::memcpy(H36, ((Ref*)h)->final_hessian, 36 * sizeof(double)); }

// brute-force probes used by the tests to validate the kd-tree itself
int ref_knn_bruteforce(const float* xyz, int n, const float* q, int k, int* out_idx) {
  std::vector<Cand> all(n);
  const F3 qq{q[0], q[1], q[2]};
  for (int i = 0; i < n; i++) all[i] = Cand{sqdist(qq, F3{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]}), i};
  std::partial_sort(all.begin(), all.begin() + k, all.end(), cand_less);
  for (int i = 0; i < k; i++) out_idx[i] = all[i].idx;
  return 0;
}
int ref_knn_kdtree(void* h, int which, const float* q, int k, int* out_idx, float* out_d) {
  Ref* r = (Ref*)h;
  Cloud& c = which == 0 ? r->src : r->tgt;
  ensure_tree(c);
  std::vector<Cand> heap;
  c.tree.search(F3{q[0], q[1], q[2]}, k, heap);
  for (int i = 0; i < (int)heap.size(); i++) out_idx[i] = heap[i].idx, out_d[i] = heap[i].d;
  return (int)heap.size();
}

// many queries at once (nq x 3): k entries per query in (distance, index) order, missing ones -1 / +inf
int ref_knn_kdtree_batch(void* h, int which, const float* q, int nq, int k, int* out_idx, float* out_d) {
  Ref* r = (Ref*)h;
  Cloud& c = which == 0 ? r->src : r->tgt;
  ensure_tree(c);
#pragma omp parallel for schedule(guided, 8)
  for (int i = 0; i < nq; i++) {
    std::vector<Cand> heap;
    c.tree.search(F3{q[3 * i], q[3 * i + 1], q[3 * i + 2]}, k, heap);
    for (int j = 0; j < k; j++) {
      out_idx[(size_t)i * k + j] = j < (int)heap.size() ? heap[j].idx : -1;
      out_d[(size_t)i * k + j] = j < (int)heap.size() ? heap[j].d : __builtin_inff();
    }
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Scan-to-submap target assembly, restated from
//   /root/reference/radar_graph_slam/apps/scan_matching_odometry_nodelet.cpp:606-618 (transform + concatenate)
//   and :412-422 (downsample() -> pcl::VoxelGrid, preprocessing_nodelet.cpp:137-144)
// PCL is not part of the reference tree (ROS noetic: PCL 1.10); its algorithm as published:
//   common/impl/transforms.hpp  detail::Transformer<double>::se3   out = (float)(m00*x + m01*y + m02*z + m03) in double
//   filters/impl/voxel_grid.hpp VoxelGrid<PointT>::applyFilter      getMinMax3D over finite points, inverse_leaf = 1/leaf
//       (float), min_b = floor(min*inverse_leaf), div_b = max_b - min_b + 1, idx = ijk . (1, div_b.x, div_b.x*div_b.y),
//       std::sort on idx (operator< compares idx only), one CentroidPoint (float sums of x, y, z, intensity, then / n)
//       per run of equal idx, in ascending idx; "Leaf size is too small" when dx*dy*dz > INT_MAX: a warning, output = input
//       (returned here as -(N + 1) with the unfiltered cloud in `out`).
// clouds: n_clouds arrays of n[c] x 4 floats {x, y, z, intensity}; poses: n_clouds x 16 doubles column-major (null: identity);
// out: room for sum(n) x 4 floats; out_idx / out_cnt (optional): voxel index and population of every output point.
struct VoxIdx {
  unsigned idx, pt;
  bool operator<(const VoxIdx& o) const { return idx < o.idx; }
};
long long ref_submap_assemble(int n_clouds, const float* const* clouds, const long long* n, const double* poses, const float* leaf, float* out,
                              int* out_idx, int* out_cnt) {
  std::vector<float> cat;  // x, y, z, intensity
  for (int c = 0; c < n_clouds; c++) {
    double T[16];
    for (int q = 0; q < 16; q++) T[q] = poses ? poses[16 * c + q] : (q % 5 == 0 ? 1.0 : 0.0);
    for (long long i = 0; i < n[c]; i++) {
      const float* p = clouds[c] + 4 * i;
      const double x = p[0], y = p[1], z = p[2];
      cat.push_back((float)(T[0] * x + T[4] * y + T[8] * z + T[12]));
      cat.push_back((float)(T[1] * x + T[5] * y + T[9] * z + T[13]));
      cat.push_back((float)(T[2] * x + T[6] * y + T[10] * z + T[14]));
      cat.push_back(p[3]);
    }
  }
  const long long N = (long long)cat.size() / 4;
  if (!leaf || !(leaf[0] > 0.f)) {  // downsample(): no filter configured -> the cloud itself
    std::memcpy(out, cat.data(), cat.size() * sizeof(float));
    return N;
  }
  const float inv[3] = {1.f / leaf[0], 1.f / leaf[1], 1.f / leaf[2]};
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  bool any = false;
  for (long long i = 0; i < N; i++) {
    const float* p = &cat[4 * i];
    if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) continue;
    any = true;
    for (int a = 0; a < 3; a++) mn[a] = std::min(mn[a], p[a]), mx[a] = std::max(mx[a], p[a]);
  }
  if (!any) return 0;
  long long d[3];
  int min_b[3], div_b[3];
  for (int a = 0; a < 3; a++) {
    d[a] = (long long)((mx[a] - mn[a]) * inv[a]) + 1;
    min_b[a] = (int)std::floor(mn[a] * inv[a]);
    div_b[a] = (int)std::floor(mx[a] * inv[a]) - min_b[a] + 1;
  }
  if ((double)d[0] * (double)d[1] * (double)d[2] > (double)INT32_MAX) {  // PCL multiplies in int64 (wraps for absurd leaves)
    // "Leaf size is too small for the input dataset. Integer indices would overflow." -- a PCL_WARN, and `output = *input_`: the
    // unfiltered cloud.  Signalled to the caller as -(N + 1).
    std::memcpy(out, cat.data(), cat.size() * sizeof(float));
    return -(N + 1);
  }
  const int mul[3] = {1, div_b[0], div_b[0] * div_b[1]};
  std::vector<VoxIdx> iv;
  iv.reserve(N);
  for (long long i = 0; i < N; i++) {
    const float* p = &cat[4 * i];
    if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) continue;
    const int i0 = (int)std::floor(p[0] * inv[0]) - min_b[0], i1 = (int)std::floor(p[1] * inv[1]) - min_b[1],
              i2 = (int)std::floor(p[2] * inv[2]) - min_b[2];
    iv.push_back(VoxIdx{(unsigned)(i0 * mul[0] + i1 * mul[1] + i2 * mul[2]), (unsigned)i});
  }
  std::sort(iv.begin(), iv.end(), std::less<VoxIdx>());
  long long nout = 0;
  for (size_t a = 0; a < iv.size();) {
    size_t b = a;
    float sx = 0.f, sy = 0.f, sz = 0.f, sw = 0.f;
    while (b < iv.size() && iv[b].idx == iv[a].idx) {
      const float* p = &cat[4 * (size_t)iv[b].pt];
      sx += p[0], sy += p[1], sz += p[2], sw += p[3];
      b++;
    }
    const float fn = (float)(b - a);
    out[4 * nout] = sx / fn, out[4 * nout + 1] = sy / fn, out[4 * nout + 2] = sz / fn, out[4 * nout + 3] = sw / fn;
    if (out_idx) out_idx[nout] = (int)iv[a].idx;
    if (out_cnt) out_cnt[nout] = (int)(b - a);
    nout++;
    a = b;
  }
  return nout;
}
}
