// ORACLE-side harness (test infrastructure, CPU only): exact k-NN through the one nearest-neighbour implementation the
// reference tree itself holds that compiles in this image -- /root/reference/radar_graph_slam/include/scan_context/nanoflann.hpp
// (header-only, STL only), with nanoflann::L2_Simple_Adaptor<float> (:423-446), whose accumulation order
// `result += diff * diff` over x, y, z is FLANN's L2_Simple<float>, the functor behind pcl::search::KdTree on the APD-GICP path
// (fast_apdgicp_impl.hpp:149-153, :318).  It is ScanContext's tree, NOT the path's FLANN: it pins the arithmetic and the
// exactness of oracle/apdgicp_ref.cpp's kd-tree, not FLANN's tie order.  The header is compiled where it lies
// (-I/root/reference/...; oracle/Makefile target _ref/libnanoflann_nn.so); nothing of it is copied into this repository and
// nothing of it travels to the GPU box.  Used by tests/test_oracle.py only.
#include <nanoflann.hpp>

#include <cstddef>
#include <vector>

namespace {
struct PointsXYZ {  // dataset adaptor over a packed float[n][3] array
  const float* p;
  size_t n;
  inline size_t kdtree_get_point_count() const { return n; }
  inline float kdtree_get_pt(const size_t idx, const size_t dim) const { return p[3 * idx + dim]; }
  template <class BBOX>
  bool kdtree_get_bbox(BBOX&) const { return false; }
};
using Tree = nanoflann::KDTreeSingleIndexAdaptor<nanoflann::L2_Simple_Adaptor<float, PointsXYZ>, PointsXYZ, 3, int>;
}  // namespace

extern "C" {
// k nearest neighbours of every query (nq x 3) among xyz (n x 3): indices and fp32 squared distances in ascending order,
// k entries per query (missing ones: -1 / +inf).  leaf_max_size: nanoflann's default is 10, PCL's FLANN index uses 15.
int nf_knn(const float* xyz, int n, const float* q, int nq, int k, int leaf_max_size, int* out_idx, float* out_d) {
  PointsXYZ pts{xyz, (size_t)n};
  Tree tree(3, pts, nanoflann::KDTreeSingleIndexAdaptorParams(leaf_max_size > 0 ? leaf_max_size : 10));
  tree.buildIndex();
  std::vector<int> idx(k);
  std::vector<float> d(k);
  for (int i = 0; i < nq; i++) {
    const size_t found = tree.knnSearch(q + 3 * (size_t)i, (size_t)k, idx.data(), d.data());
    for (int j = 0; j < k; j++) {
      out_idx[(size_t)i * k + j] = j < (int)found ? idx[j] : -1;
      out_d[(size_t)i * k + j] = j < (int)found ? d[j] : __builtin_inff();
    }
  }
  return 0;
}
}
