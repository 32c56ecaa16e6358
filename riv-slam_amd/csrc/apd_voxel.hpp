// Scan-to-submap target assembly on the device:
//   scan_matching_odometry_nodelet.cpp:606-618  transform the last <= max_submap_frames keyframe clouds by
//                                               their relative poses and concatenate them,
//   scan_matching_odometry_nodelet.cpp:412-422  downsample() with the configured filter -- pcl::VoxelGrid
//                                               (downsample_method "VOXELGRID", preprocessing_nodelet.cpp:137-144),
// so the submap never leaves HBM before it becomes the registration target.
//
// PCL is a third-party dependency that is not part of the reference tree (ROS noetic ships PCL 1.10);
// the kernels follow its published algorithm:
//   pcl::transformPointCloud(in, out, Matrix4d)   common/impl/transforms.hpp, detail::Transformer<double>::se3:
//        out = (float)(m00*x + m01*y + m02*z + m03), evaluated in double, left to right
//   pcl::VoxelGrid<PointT>::applyFilter           filters/impl/voxel_grid.hpp:
//        min/max over the finite points; inverse_leaf = 1/leaf (float); min_b = floor(min * inverse_leaf);
//        div_b = max_b - min_b + 1; idx = ijk . (1, div_b.x, div_b.x*div_b.y) with ijk = floor(p * inverse_leaf) - min_b
//        (float products, int32 index); sort by idx; one output point per occupied voxel in ascending idx,
//        the centroid of all fields (float sums, then / n).
// PCL sorts with std::sort on idx alone, so the order of the float additions inside a voxel is whatever
// introsort leaves; here the points of a voxel are added in input order (the sort key carries the point
// index).  Voxel membership, count and order are identical; centroids can differ in the last bit.
#pragma once
#include <hip/hip_runtime.h>

#include "apd_sort.hpp"

namespace apd {

struct SubmapJob {
  const float* xyz;   // first coordinate of point 0
  long long n;
  int stride;         // floats between points
  int intensity_off;  // floats from x to the intensity field, < 0: none (0 is stored)
  int out_off;        // first slot in the concatenated cloud
  int pad_;
  double T[12];       // rows 0..2 of the relative pose, row-major
};

// pcl::transformPointCloud + operator+= : one float4 {x, y, z, intensity} per input point
__global__ void k_submap_transform(const SubmapJob* jobs, float4* cat) {
  const SubmapJob j = jobs[blockIdx.y];
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= j.n) return;
  const float* p = j.xyz + i * j.stride;
  const double x = (double)p[0], y = (double)p[1], z = (double)p[2];
  float4 o;
  o.x = (float)(j.T[0] * x + j.T[1] * y + j.T[2] * z + j.T[3]);
  o.y = (float)(j.T[4] * x + j.T[5] * y + j.T[6] * z + j.T[7]);
  o.z = (float)(j.T[8] * x + j.T[9] * y + j.T[10] * z + j.T[11]);
  o.w = j.intensity_off >= 0 ? p[j.intensity_off] : 0.f;
  cat[j.out_off + i] = o;
}

__device__ __forceinline__ bool finite3(const float4& p) { return isfinite(p.x) && isfinite(p.y) && isfinite(p.z); }

// getMinMax3D over the finite points (ordered-int atomics, box6 initialised to +inf x3, -inf x3)
__global__ void k_vox_bbox(const float4* pts, int n, int* box6) {
  __shared__ float red[256 / 64];
  const int i = blockIdx.x * blockDim.x + threadIdx.x, tid = threadIdx.x;
  const float inf = __builtin_inff();
  float v[6] = {inf, inf, inf, -inf, -inf, -inf};
  for (int e = i; e < n; e += gridDim.x * blockDim.x) {  // few blocks: the six atomics per block are serialised
    const float4 p = pts[e];
    if (finite3(p)) {
      v[0] = fminf(v[0], p.x), v[1] = fminf(v[1], p.y), v[2] = fminf(v[2], p.z);
      v[3] = fmaxf(v[3], p.x), v[4] = fmaxf(v[4], p.y), v[5] = fmaxf(v[5], p.z);
    }
  }
  for (int q = 0; q < 6; q++) {
    const float r = block_reduce_minmax(v[q], q >= 3, red, tid, 256);
    if (tid == 0) {
      int b = __float_as_int(r);
      b = b >= 0 ? b : b ^ 0x7fffffff;
      if (q < 3) atomicMin(box6 + q, b);
      else atomicMax(box6 + q, b);
    }
  }
}

struct VoxGrid {
  int min_b[3], div_b[3], mul[3];
  bool overflow, empty;
};
__device__ __forceinline__ VoxGrid vox_grid(const int* box6, const float inv[3]) {
  VoxGrid g;
  float lo[3], hi[3];
  for (int a = 0; a < 3; a++) lo[a] = ordered_int_to_float(box6[a]), hi[a] = ordered_int_to_float(box6[3 + a]);
  g.empty = !(lo[0] <= hi[0]);
  long long d[3];
  for (int a = 0; a < 3; a++) {
    d[a] = (long long)((hi[a] - lo[a]) * inv[a]) + 1;  // voxel_grid.hpp: dx = int64((max - min) * inverse_leaf) + 1
    g.min_b[a] = (int)floorf(lo[a] * inv[a]);
    const int max_b = (int)floorf(hi[a] * inv[a]);
    g.div_b[a] = max_b - g.min_b[a] + 1;
  }
  // "Leaf size is too small for the input dataset" (PCL multiplies in int64, which wraps for absurd leaves; double does not)
  g.overflow = !g.empty && (double)d[0] * (double)d[1] * (double)d[2] > 2147483647.0;
  g.mul[0] = 1, g.mul[1] = g.div_b[0], g.mul[2] = g.div_b[0] * g.div_b[1];
  return g;
}

// sort key = voxel index << 32 | point index; padding and non-finite points get ~0 and sort to the end
__global__ void k_vox_keys(const float4* pts, int n, int np2, const int* box6, float ilx, float ily, float ilz, unsigned long long* keys,
                           int* err_flag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= np2) return;
  unsigned long long key = ~0ull;
  if (i < n) {
    const float inv[3] = {ilx, ily, ilz};
    const VoxGrid g = vox_grid(box6, inv);
    if (g.overflow) {
      if (i == 0) atomicExch(err_flag, 5);
    } else {
      const float4 p = pts[i];
      if (finite3(p)) {
        const int i0 = (int)floorf(p.x * ilx) - g.min_b[0], i1 = (int)floorf(p.y * ily) - g.min_b[1], i2 = (int)floorf(p.z * ilz) - g.min_b[2];
        const int idx = i0 * g.mul[0] + i1 * g.mul[1] + i2 * g.mul[2];
        key = ((unsigned long long)(unsigned)idx << 32) | (unsigned)i;
      }
    }
  }
  keys[i] = key;
}

// (the u64 bitonic sort -- k_bitonic_tile_sort / k_bitonic_global / k_bitonic_tile_merge -- lives in apd_sort.hpp)

// ---- voxel heads and their output slots: exclusive scan of the head flags (4096 per block, then the block sums)
constexpr int SCAN_ITEMS = 4;
constexpr int SCAN_BLK = 1024;
__device__ __forceinline__ bool vox_head(const unsigned long long* keys, int i) {
  const unsigned hi = (unsigned)(keys[i] >> 32);
  return hi != 0xFFFFFFFFu && (i == 0 || (unsigned)(keys[i - 1] >> 32) != hi);
}
__device__ __forceinline__ int block_exclusive_scan(int v, int* lds, int tid, int* total) {  // SCAN_BLK threads
  const int lane = tid & 63, wave = tid >> 6;
  int inc = v;
  for (int off = 1; off < 64; off <<= 1) {
    const int o = __shfl_up(inc, off, 64);
    if (lane >= off) inc += o;
  }
  if (lane == 63) lds[wave] = inc;
  __syncthreads();
  if (wave == 0) {
    int s = lane < SCAN_BLK / 64 ? lds[lane] : 0;
    for (int off = 1; off < SCAN_BLK / 64; off <<= 1) {
      const int o = __shfl_up(s, off, 64);
      if (lane >= off) s += o;
    }
    if (lane < SCAN_BLK / 64) lds[lane] = s;  // inclusive over waves
  }
  __syncthreads();
  const int before = wave ? lds[wave - 1] : 0;
  *total = lds[SCAN_BLK / 64 - 1];
  __syncthreads();
  return before + inc - v;
}
__global__ __launch_bounds__(SCAN_BLK) void k_vox_heads(const unsigned long long* keys, int np2, int* pos, int* bsum) {
  __shared__ int lds[SCAN_BLK / 64];
  const int tid = threadIdx.x, i0 = (blockIdx.x * SCAN_BLK + tid) * SCAN_ITEMS;
  int f[SCAN_ITEMS], s = 0;
  for (int u = 0; u < SCAN_ITEMS; u++) f[u] = (i0 + u < np2 && vox_head(keys, i0 + u)) ? 1 : 0, s += f[u];
  int total;
  int ex = block_exclusive_scan(s, lds, tid, &total);
  for (int u = 0; u < SCAN_ITEMS; u++) {
    if (i0 + u < np2) pos[i0 + u] = ex;
    ex += f[u];
  }
  if (tid == 0) bsum[blockIdx.x] = total;
}
// exclusive scan of up to SCAN_BLK * 16 block sums in one block; out_total = number of voxels
__global__ __launch_bounds__(SCAN_BLK) void k_scan_bsum(int* bsum, int nb, int* out_total) {
  __shared__ int lds[SCAN_BLK / 64];
  const int tid = threadIdx.x;
  int carry = 0;
  for (int b0 = 0; b0 < nb; b0 += SCAN_BLK) {
    const int i = b0 + tid;
    const int v = i < nb ? bsum[i] : 0;
    int total;
    const int ex = block_exclusive_scan(v, lds, tid, &total);
    if (i < nb) bsum[i] = carry + ex;
    carry += total;
  }
  if (tid == 0) *out_total = carry;
}

// CentroidPoint<PointXYZI>: float sums of x, y, z, intensity over the voxel, then / n
__global__ void k_vox_centroids(const unsigned long long* keys, const float4* pts, const int* pos, const int* bsum, int np2, float4* out,
                                int out_cap) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= np2 || !vox_head(keys, i)) return;
  const unsigned idx = (unsigned)(keys[i] >> 32);
  float sx = 0.f, sy = 0.f, sz = 0.f, sw = 0.f;
  int cnt = 0;
  for (int j = i; j < np2 && (unsigned)(keys[j] >> 32) == idx; j++) {
    const float4 p = pts[(unsigned)keys[j]];
    sx += p.x, sy += p.y, sz += p.z, sw += p.w;
    cnt++;
  }
  const int slot = pos[i] + bsum[i / (SCAN_BLK * SCAN_ITEMS)];
  const float fn = (float)cnt;
  if (slot < out_cap) out[slot] = make_float4(sx / fn, sy / fn, sz / fn, sw / fn);
}

}  // namespace apd
