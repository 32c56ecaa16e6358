// Spatial pre-ordering of a cloud (done once per set_cloud): Morton (Z-order) sort + per-chunk and
// per-group axis-aligned bounding boxes.  Nothing here changes any result: the nearest-neighbour and
// k-NN kernels use the boxes only to SKIP chunks whose fp32 lower bound proves they cannot contain a
// (better) neighbour, and ties are still resolved by the ORIGINAL index (perm).
//
//   spts[s]  = pts[perm[s]]                     sorted copy, s = position on the curve; .w = the bits of perm[s]
//   cbox[c]  = {lo.xyz, hi.xyz} of sorted points [16c, 16c+16)          (kChunk = 16)
//   gbox[g]  = box of chunks [8g, 8g+8)  = 128 sorted points            (kGroup = 8 chunks)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace apd {

constexpr int kGroupChunks = 8;            // chunks per group
constexpr int kGroupPts = 16 * kGroupChunks;  // 128 points per group
constexpr int SORT_BLK = 1024;
constexpr int SORT_LDS_MAX_N = 16384;      // single-block LDS sort up to this many points (128 KB of u64 keys)

struct Box {
  float lx, ly, lz, hx, hy, hz;
};

__device__ __forceinline__ unsigned expand10(unsigned v) {  // 10 bits -> every third bit
  v &= 0x3ffu;
  v = (v | (v << 16)) & 0x030000ffu;
  v = (v | (v << 8)) & 0x0300f00fu;
  v = (v | (v << 4)) & 0x030c30c3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}

#ifndef APD_CURVE_HILBERT
#define APD_CURVE_HILBERT 1
#endif
// 30-bit position of a point on a space-filling curve through the 1024^3 grid over the cloud's bounding cube.  The order
// only decides which points share a chunk / group box -- never a result -- so any curve is exact; the Hilbert curve has no
// jumps (consecutive cells are face neighbours), which makes the boxes of 16 / 128 consecutive points tighter than along the
// Z-curve.  Skilling's axes-to-transpose transform, then the usual bit interleave.
__device__ __forceinline__ unsigned morton30(float x, float y, float z, float lx, float ly, float lz, float scale) {
  const float fx = fminf(fmaxf((x - lx) * scale, 0.f), 1023.f);
  const float fy = fminf(fmaxf((y - ly) * scale, 0.f), 1023.f);
  const float fz = fminf(fmaxf((z - lz) * scale, 0.f), 1023.f);
  unsigned X0 = (unsigned)fx, X1 = (unsigned)fy, X2 = (unsigned)fz;
#if APD_CURVE_HILBERT
#pragma unroll
  for (unsigned Q = 512u; Q > 1u; Q >>= 1) {
    const unsigned P = Q - 1u;
    if (X0 & Q) X0 ^= P;  // (i = 0: invert)
    if (X1 & Q) X0 ^= P;
    else { const unsigned t = (X0 ^ X1) & P; X0 ^= t, X1 ^= t; }
    if (X2 & Q) X0 ^= P;
    else { const unsigned t = (X0 ^ X2) & P; X0 ^= t, X2 ^= t; }
  }
  X1 ^= X0, X2 ^= X1;  // Gray encode
  unsigned t = 0;
#pragma unroll
  for (unsigned Q = 512u; Q > 1u; Q >>= 1)
    if (X2 & Q) t ^= Q - 1u;
  X0 ^= t, X1 ^= t, X2 ^= t;
#endif
  return (expand10(X0) << 2) | (expand10(X1) << 1) | expand10(X2);
}

__device__ __forceinline__ float block_reduce_minmax(float v, bool is_max, float* lds, int tid, int nthreads) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float o = __shfl_down(v, off, 64);
    v = is_max ? fmaxf(v, o) : fminf(v, o);
  }
  __syncthreads();
  if ((tid & 63) == 0) lds[tid >> 6] = v;
  __syncthreads();
  float r = lds[0];
  for (int w = 1; w < nthreads / 64; w++) r = is_max ? fmaxf(r, lds[w]) : fminf(r, lds[w]);
  return r;
}

struct SortJob {
  const float4* pts;  // original order
  float4* spts;       // out: sorted
  int* perm;          // out: sorted position -> original index
  Box* cbox;          // out: ceil(n/16) chunk boxes
  Box* gbox;          // out: ceil(n/128) group boxes
  int n;
  int pad_;
};

// One block per cloud, everything in LDS (n <= SORT_LDS_MAX_N).
__global__ __launch_bounds__(SORT_BLK) void k_sort_cloud_lds(const SortJob* jobs) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
  __shared__ float red[SORT_BLK / 64];
  const SortJob job = jobs[blockIdx.x];
  const int n = job.n, tid = threadIdx.x;
  int np2 = 1;
  while (np2 < n) np2 <<= 1;
  // bounding box
  const float inf = __builtin_inff();
  float lx = inf, ly = inf, lz = inf, hx = -inf, hy = -inf, hz = -inf;
  for (int i = tid; i < n; i += SORT_BLK) {
    const float4 p = job.pts[i];
    lx = fminf(lx, p.x), ly = fminf(ly, p.y), lz = fminf(lz, p.z);
    hx = fmaxf(hx, p.x), hy = fmaxf(hy, p.y), hz = fmaxf(hz, p.z);
  }
  lx = block_reduce_minmax(lx, false, red, tid, SORT_BLK);
  ly = block_reduce_minmax(ly, false, red, tid, SORT_BLK);
  lz = block_reduce_minmax(lz, false, red, tid, SORT_BLK);
  hx = block_reduce_minmax(hx, true, red, tid, SORT_BLK);
  hy = block_reduce_minmax(hy, true, red, tid, SORT_BLK);
  hz = block_reduce_minmax(hz, true, red, tid, SORT_BLK);
  const float ext = fmaxf(fmaxf(hx - lx, hy - ly), fmaxf(hz - lz, 1e-30f));
  const float scale = 1023.f / ext;
  for (int i = tid; i < np2; i += SORT_BLK) {
    if (i < n) {
      const float4 p = job.pts[i];
      keys[i] = ((unsigned long long)morton30(p.x, p.y, p.z, lx, ly, lz, scale) << 32) | (unsigned)i;
    } else {
      keys[i] = ~0ull;
    }
  }
  __syncthreads();
  for (int k = 2; k <= np2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < np2 / 2; t += SORT_BLK) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));  // index with bit j cleared
        const int l = i | j;
        const bool up = (i & k) == 0;
        const unsigned long long a = keys[i], b = keys[l];
        if ((a > b) == up) keys[i] = b, keys[l] = a;
      }
      __syncthreads();
    }
  }
  for (int s = tid; s < n; s += SORT_BLK) {
    const int o = (int)(unsigned)keys[s];
    job.perm[s] = o;
    float4 q = job.pts[o];
    q.w = __int_as_float(o);  // the sorted copy carries the original index in .w (only x, y, z are coordinates on this path)
    job.spts[s] = q;
  }
  __syncthreads();
  // chunk boxes from the sorted copy just written by this block (same block -> visible after the barrier
  // only through L1/L2 of this CU: we re-read our own writes, which is coherent within a workgroup)
  __threadfence_block();
  const int nchunks = (n + 15) / 16;
  for (int c = tid; c < nchunks; c += SORT_BLK) {
    Box b{inf, inf, inf, -inf, -inf, -inf};
    for (int e = 0; e < 16; e++) {
      const int s = c * 16 + e;
      if (s < n) {
        const int o = (int)(unsigned)keys[s];
        const float4 p = job.pts[o];
        b.lx = fminf(b.lx, p.x), b.ly = fminf(b.ly, p.y), b.lz = fminf(b.lz, p.z);
        b.hx = fmaxf(b.hx, p.x), b.hy = fmaxf(b.hy, p.y), b.hz = fmaxf(b.hz, p.z);
      }
    }
    job.cbox[c] = b;
  }
  const int ngroups = (n + kGroupPts - 1) / kGroupPts;
  for (int g = tid; g < ngroups; g += SORT_BLK) {
    Box b{inf, inf, inf, -inf, -inf, -inf};
    for (int e = 0; e < kGroupPts; e++) {
      const int s = g * kGroupPts + e;
      if (s < n) {
        const int o = (int)(unsigned)keys[s];
        const float4 p = job.pts[o];
        b.lx = fminf(b.lx, p.x), b.ly = fminf(b.ly, p.y), b.lz = fminf(b.lz, p.z);
        b.hx = fmaxf(b.hx, p.x), b.hy = fmaxf(b.hy, p.y), b.hz = fmaxf(b.hz, p.z);
      }
    }
    job.gbox[g] = b;
  }
}

// ----------------------------------------------------------------------------------------------
// The register sort network of k_sort_tiles<E> (below; its one-block-per-cloud predecessor k_sort_cloud_reg<E> measured 58 us for
// 8192 keys against the tiled 29 us and is in the history of round 2): 1024*E keys per block with the keys in REGISTERS:
// thread t owns the E keys of positions t*E .. t*E+E-1.  A bitonic stage with stride j
//   j < E          is a compare-exchange between two registers of the same thread,
//   E <= j < 64*E  exchanges with lane (lane ^ j/E) of the same wave -- DPP for lane distances 1, 2, 4, 8 (pure VALU),
//                  ds_bpermute for 16 and 32 -- no barrier,
//   j >= 64*E      goes through LDS (slot e*1024 + t: conflict-free) with two barriers;
// for 8192 keys in one block that leaves 10 barrier stages out of 91 (k_sort_cloud_lds pays a barrier in every one of them).
// Every stride is a compile-time constant (fully unrolled stage loops): key[] is only ever indexed statically and the
// DPP controls are immediates.  118 us (LDS version) -> 58 us for 8192 keys, bound by the one CU a block runs on.
// The sort is a total order of unique keys, so the permutation is the one k_sort_cloud_lds produces.
// The sorted points are gathered once into registers and serve the copy, the chunk boxes (16/E neighbouring lanes)
// and the group boxes (128/E lanes) through xor-shuffles.
template <int CTRL>
__device__ __forceinline__ unsigned dpp_mov(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, false); }
// value of lane (lane ^ m): DPP (pure VALU) for m = 1, 2 (quad_perm), 4 (row_shl/row_shr 4, selected by lane bit 2) and
// 8 (row_ror 8); ds_bpermute only for 16 and 32
__device__ __forceinline__ unsigned lane_xor_u32(unsigned v, int m, int lane) {
  switch (m) {
    case 1: return dpp_mov<0xB1>(v);
    case 2: return dpp_mov<0x4E>(v);
    case 4: {
      const unsigned up = dpp_mov<0x104>(v), dn = dpp_mov<0x114>(v);  // row_shl:4 = lane + 4, row_shr:4 = lane - 4
      return (lane & 4) ? dn : up;
    }
    case 8: return dpp_mov<0x128>(v);  // row_ror:8
    default: return __shfl_xor(v, m, 64);
  }
}
__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int m, int lane) {
  const unsigned lo = lane_xor_u32((unsigned)v, m, lane), hi = lane_xor_u32((unsigned)(v >> 32), m, lane);
  return ((unsigned long long)hi << 32) | lo;
}

// ----------------------------------------------------------------------------------------------
// The same sort spread over several CUs (2048 < n <= 16384): one block per QUARTER of the padded cloud sorts its tile with
// the register network above (E = 1, 2 or 4 keys per thread; every block first reduces the bounding box of the WHOLE cloud,
// so all tiles use the same keys as the one-block kernels), a second launch merges the four sorted tiles by rank -- a key's
// final position is its position in its own tile plus the number of smaller keys in each of the other three, three
// branch-free binary searches over tiles staged in LDS -- and scatters the points, a third one builds the boxes.  The keys
// are unique, so the permutation is exactly the one-block kernels': 64 us on one CU -> three short launches (a single
// registration spends a third of its time in the sort; a batch sorts on 4 x as many CUs).
struct TileJob {
  SortJob job;
  unsigned long long* keys;  // [4][NT] sorted tiles
  int nt;                    // tile size = padded size / 4 (1024, 2048 or 4096)
  int pad_;
  const float4* staged;      // null, or the cloud as the HOST packed it, in pinned memory: n points, then the bounding box
                             // (low corner, high corner).  The tile blocks read it from there -- each its own quarter -- and
                             // write the device copy job.pts themselves: no copy in front of the sort (8.6 us on the copy
                             // engine + 12.7 us until the first kernel starts behind it, measured on the odometry frame)
};

template <int E>
__global__ __launch_bounds__(SORT_BLK) void k_sort_tiles(const TileJob* jobs) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long xch[];  // [E][1024]
  __shared__ float red[SORT_BLK / 64];
  const TileJob tj = jobs[blockIdx.y];
  const SortJob& job = tj.job;
  constexpr int NT = SORT_BLK * E;
  if (tj.nt != NT) return;  // (a launch covers the clouds of one size class)
  const int n = job.n, tid = threadIdx.x, tile = blockIdx.x;
  const float inf = __builtin_inff();
  float lx = inf, ly = inf, lz = inf, hx = -inf, hy = -inf, hz = -inf;
  float4 mine[E];
  if (tj.staged) {  // the host reduced the box while it packed the points (min and max are exact: the same box)
#pragma unroll
    for (int e = 0; e < E; e++) {
      const int i = tile * NT + tid * E + e;
      if (i < n) mine[e] = tj.staged[i];
    }
    const float4 lo = tj.staged[n], hi = tj.staged[n + 1];
    lx = lo.x, ly = lo.y, lz = lo.z, hx = hi.x, hy = hi.y, hz = hi.z;
#pragma unroll
    for (int e = 0; e < E; e++) {
      const int i = tile * NT + tid * E + e;
      if (i < n) const_cast<float4*>(job.pts)[i] = mine[e];
    }
  } else {
    for (int i = tid; i < n; i += SORT_BLK) {
      const float4 q = job.pts[i];
      lx = fminf(lx, q.x), ly = fminf(ly, q.y), lz = fminf(lz, q.z);
      hx = fmaxf(hx, q.x), hy = fmaxf(hy, q.y), hz = fmaxf(hz, q.z);
    }
    lx = block_reduce_minmax(lx, false, red, tid, SORT_BLK);
    ly = block_reduce_minmax(ly, false, red, tid, SORT_BLK);
    lz = block_reduce_minmax(lz, false, red, tid, SORT_BLK);
    hx = block_reduce_minmax(hx, true, red, tid, SORT_BLK);
    hy = block_reduce_minmax(hy, true, red, tid, SORT_BLK);
    hz = block_reduce_minmax(hz, true, red, tid, SORT_BLK);
#pragma unroll
    for (int e = 0; e < E; e++) {
      const int i = tile * NT + tid * E + e;
      if (i < n) mine[e] = job.pts[i];
    }
  }
  const float ext = fmaxf(fmaxf(hx - lx, hy - ly), fmaxf(hz - lz, 1e-30f));
  const float scale = 1023.f / ext;
  unsigned long long key[E];
#pragma unroll
  for (int e = 0; e < E; e++) {
    const int i = tile * NT + tid * E + e;
    key[e] = i < n ? ((unsigned long long)morton30(mine[e].x, mine[e].y, mine[e].z, lx, ly, lz, scale) << 32) | (unsigned)i : ~0ull;
  }
#pragma unroll
  for (int k = 2; k <= NT; k <<= 1) {
    for (int j = k >> 1; j >= 64 * E; j >>= 1) {  // partner in another wave: through LDS
      const int pt = tid ^ (j / E);
      __syncthreads();
#pragma unroll
      for (int e = 0; e < E; e++) xch[e * SORT_BLK + tid] = key[e];
      __syncthreads();
#pragma unroll
      for (int e = 0; e < E; e++) {
        const int i = tid * E + e;
        const unsigned long long o = xch[e * SORT_BLK + pt];
        const bool take_min = ((i & j) == 0) == ((i & k) == 0);
        key[e] = ((o < key[e]) == take_min) ? o : key[e];
      }
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
      const int j = m * E;
      if (j < k) {
#pragma unroll
        for (int e = 0; e < E; e++) {
          const int i = tid * E + e;
          const unsigned long long o = shfl_xor_u64(key[e], m, tid & 63);
          const bool take_min = ((i & j) == 0) == ((i & k) == 0);
          key[e] = ((o < key[e]) == take_min) ? o : key[e];
        }
      }
    }
#pragma unroll
    for (int j = E / 2; j > 0; j >>= 1) {
      if (j < k) {
#pragma unroll
        for (int e = 0; e < E; e++) {
          if ((e & j) == 0) {
            const int i = tid * E + e;
            const bool up = (i & k) == 0;
            const unsigned long long a = key[e], b = key[e | j];
            if ((a > b) == up) key[e] = b, key[e | j] = a;
          }
        }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < E; e++) tj.keys[(size_t)tile * NT + tid * E + e] = key[e];
}

// grid (4 * nt / 1024, clouds): thread g of a cloud owns key g of the concatenated tiles
__global__ __launch_bounds__(SORT_BLK) void k_merge_tiles(const TileJob* jobs) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long tl[];  // [4][nt]
  const TileJob tj = jobs[blockIdx.y];
  const SortJob& job = tj.job;
  const int nt = tj.nt, tid = threadIdx.x;
  if ((int)(blockIdx.x * SORT_BLK) >= 4 * nt) return;
  for (int e = tid; e < 4 * nt; e += SORT_BLK) tl[e] = tj.keys[e];
  __syncthreads();
  const int g = blockIdx.x * SORT_BLK + tid, tile = g / nt, p = g - tile * nt;
  const unsigned long long key = tl[g];
  if (key == ~0ull) return;  // padding
  int rank = p;
  int pos[3] = {0, 0, 0};
  const unsigned long long* other[3];
#pragma unroll
  for (int u = 0; u < 3; u++) other[u] = tl + (size_t)((tile + 1 + u) & 3) * nt;
  for (int step = nt >> 1; step > 0; step >>= 1) {  // three searches side by side
#pragma unroll
    for (int u = 0; u < 3; u++) pos[u] += other[u][pos[u] + step - 1] < key ? step : 0;
  }
#pragma unroll
  for (int u = 0; u < 3; u++) rank += pos[u] + (other[u][pos[u]] < key ? 1 : 0);
  const int o = (int)(unsigned)key;
  float4 q = job.pts[o];
  q.w = __int_as_float(o);
  job.perm[rank] = o;
  job.spts[rank] = q;
}

// chunk and group boxes of the sorted copy: one thread per chunk, 8 neighbouring lanes per group.  grid (ceil(nchunks / 256), clouds)
__global__ __launch_bounds__(256) void k_boxes_sorted(const TileJob* jobs) {
  const SortJob job = jobs[blockIdx.y].job;
  const int n = job.n, c = blockIdx.x * 256 + threadIdx.x;
  const float inf = __builtin_inff();
  Box bx{inf, inf, inf, -inf, -inf, -inf};
  if (c * 16 < n) {
#pragma unroll 4
    for (int e = 0; e < 16; e++) {
      const int sidx = c * 16 + e;
      if (sidx < n) {
        const float4 q = job.spts[sidx];
        bx.lx = fminf(bx.lx, q.x), bx.ly = fminf(bx.ly, q.y), bx.lz = fminf(bx.lz, q.z);
        bx.hx = fmaxf(bx.hx, q.x), bx.hy = fmaxf(bx.hy, q.y), bx.hz = fmaxf(bx.hz, q.z);
      }
    }
    job.cbox[c] = bx;
  }
#pragma unroll
  for (int m = 1; m < kGroupChunks; m <<= 1) {
    bx.lx = fminf(bx.lx, __shfl_xor(bx.lx, m, 64)), bx.ly = fminf(bx.ly, __shfl_xor(bx.ly, m, 64)), bx.lz = fminf(bx.lz, __shfl_xor(bx.lz, m, 64));
    bx.hx = fmaxf(bx.hx, __shfl_xor(bx.hx, m, 64)), bx.hy = fmaxf(bx.hy, __shfl_xor(bx.hy, m, 64)), bx.hz = fmaxf(bx.hz, __shfl_xor(bx.hz, m, 64));
  }
  if (c % kGroupChunks == 0 && c * 16 < n) job.gbox[c / kGroupChunks] = bx;
}

// ---- generic path for large clouds (n > SORT_LDS_MAX_N): keys in global memory
__global__ void k_bbox_atomic(const float4* pts, int n, int* box6 /* ordered-int encoded, init lo=+inf hi=-inf */) {
  __shared__ float red[256 / 64];
  const int i = blockIdx.x * blockDim.x + threadIdx.x, tid = threadIdx.x;
  const float inf = __builtin_inff();
  float v[6] = {inf, inf, inf, -inf, -inf, -inf};
  for (int e = i; e < n; e += gridDim.x * blockDim.x) {  // few blocks, many points each: the six atomics per block are serialised
    const float4 p = pts[e];
    v[0] = fminf(v[0], p.x), v[1] = fminf(v[1], p.y), v[2] = fminf(v[2], p.z);
    v[3] = fmaxf(v[3], p.x), v[4] = fmaxf(v[4], p.y), v[5] = fmaxf(v[5], p.z);
  }
  for (int q = 0; q < 6; q++) {
    const float r = block_reduce_minmax(v[q], q >= 3, red, tid, 256);
    if (tid == 0) {
      // monotone float -> int mapping so that integer atomics order like floats
      int b = __float_as_int(r);
      b = b >= 0 ? b : b ^ 0x7fffffff;
      if (q < 3) atomicMin(box6 + q, b);
      else atomicMax(box6 + q, b);
    }
  }
}
__device__ __forceinline__ float ordered_int_to_float(int b) { return __int_as_float(b >= 0 ? b : b ^ 0x7fffffff); }

// Large clouds are dense: with 10 bits per axis many points share a Morton cell and the 128-point groups stop being
// compact.  The key is [Morton code, `bits` per axis][index, idx_bits], bits = min(21, (64 - idx_bits) / 3).
__device__ __forceinline__ unsigned long long expand21(unsigned long long x) {
  x &= 0x1fffffull;
  x = (x | x << 32) & 0x1f00000000ffffull;
  x = (x | x << 16) & 0x1f0000ff0000ffull;
  x = (x | x << 8) & 0x100f00f00f00f00full;
  x = (x | x << 4) & 0x10c30c30c30c30c3ull;
  x = (x | x << 2) & 0x1249249249249249ull;
  return x;
}
__global__ void k_morton_keys(const float4* pts, int n, int np2, const int* box6, unsigned long long* keys, int bits, int idx_bits) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= np2) return;
  if (i >= n) {
    keys[i] = ~0ull;
    return;
  }
  const float lx = ordered_int_to_float(box6[0]), ly = ordered_int_to_float(box6[1]), lz = ordered_int_to_float(box6[2]);
  const float hx = ordered_int_to_float(box6[3]), hy = ordered_int_to_float(box6[4]), hz = ordered_int_to_float(box6[5]);
  const float ext = fmaxf(fmaxf(hx - lx, hy - ly), fmaxf(hz - lz, 1e-30f));
  const float top = (float)((1u << bits) - 1u), scale = top / ext;
  const float4 p = pts[i];
  unsigned X0 = (unsigned)fminf(fmaxf((p.x - lx) * scale, 0.f), top);
  unsigned X1 = (unsigned)fminf(fmaxf((p.y - ly) * scale, 0.f), top);
  unsigned X2 = (unsigned)fminf(fmaxf((p.z - lz) * scale, 0.f), top);
#if APD_CURVE_HILBERT
  for (unsigned Q = 1u << (bits - 1); Q > 1u; Q >>= 1) {  // Hilbert curve, as in morton30
    const unsigned P = Q - 1u;
    if (X0 & Q) X0 ^= P;
    if (X1 & Q) X0 ^= P;
    else { const unsigned t = (X0 ^ X1) & P; X0 ^= t, X1 ^= t; }
    if (X2 & Q) X0 ^= P;
    else { const unsigned t = (X0 ^ X2) & P; X0 ^= t, X2 ^= t; }
  }
  X1 ^= X0, X2 ^= X1;
  unsigned t = 0;
  for (unsigned Q = 1u << (bits - 1); Q > 1u; Q >>= 1)
    if (X2 & Q) t ^= Q - 1u;
  X0 ^= t, X1 ^= t, X2 ^= t;
#endif
  const unsigned long long m = (expand21(X0) << 2) | (expand21(X1) << 1) | expand21(X2);
  keys[i] = (m << idx_bits) | (unsigned long long)(unsigned)i;
}

__global__ void k_bitonic_global(unsigned long long* keys, int np2, int k, int j) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= np2 / 2) return;
  const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
  const int l = i | j;
  const bool up = (i & k) == 0;
  const unsigned long long a = keys[i], b = keys[l];
  if ((a > b) == up) keys[i] = b, keys[l] = a;
}

// ---- the same sort with LDS tiles of VOX_TILE keys: k_bitonic_global only for strides >= VOX_TILE, one tile kernel for the
// rest of every merge stage (35 launches for 2^19 keys instead of 190); also used by the voxel filter (apd_voxel.hpp)
constexpr int VOX_TILE = 4096;
__device__ __forceinline__ void tile_steps(unsigned long long* t, int base, int k, int jmax, int tid) {
  for (int j = jmax; j > 0; j >>= 1) {
    for (int q = tid; q < VOX_TILE / 2; q += 1024) {
      const int i = ((q & ~(j - 1)) << 1) | (q & (j - 1)), l = i | j;
      const bool up = ((base + i) & k) == 0;
      const unsigned long long a = t[i], b = t[l];
      if ((a > b) == up) t[i] = b, t[l] = a;
    }
    __syncthreads();
  }
}
// full sort of every tile (the directions alternate as the later merges expect)
__global__ __launch_bounds__(1024) void k_bitonic_tile_sort(unsigned long long* keys) {
  __shared__ unsigned long long t[VOX_TILE];
  const int base = blockIdx.x * VOX_TILE, tid = threadIdx.x;
  for (int q = tid; q < VOX_TILE; q += 1024) t[q] = keys[base + q];
  __syncthreads();
  for (int k = 2; k <= VOX_TILE; k <<= 1) tile_steps(t, base, k, k >> 1, tid);
  for (int q = tid; q < VOX_TILE; q += 1024) keys[base + q] = t[q];
}
// the strides < VOX_TILE of merge stage k
__global__ __launch_bounds__(1024) void k_bitonic_tile_merge(unsigned long long* keys, int k) {
  __shared__ unsigned long long t[VOX_TILE];
  const int base = blockIdx.x * VOX_TILE, tid = threadIdx.x;
  for (int q = tid; q < VOX_TILE; q += 1024) t[q] = keys[base + q];
  __syncthreads();
  tile_steps(t, base, k, VOX_TILE >> 1, tid);
  for (int q = tid; q < VOX_TILE; q += 1024) keys[base + q] = t[q];
}

__global__ void k_gather_sorted(const unsigned long long* keys, const float4* pts, int n, float4* spts, int* perm, int idx_bits) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  const int o = (int)(keys[s] & ((1ull << idx_bits) - 1ull));
  perm[s] = o;
  float4 q = pts[o];
  q.w = __int_as_float(o);
  spts[s] = q;
}

__global__ void k_boxes(const float4* spts, int n, int pts_per_box, Box* out, int nboxes) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nboxes) return;
  const float inf = __builtin_inff();
  Box b{inf, inf, inf, -inf, -inf, -inf};
  for (int e = 0; e < pts_per_box; e++) {
    const int s = c * pts_per_box + e;
    if (s < n) {
      const float4 p = spts[s];
      b.lx = fminf(b.lx, p.x), b.ly = fminf(b.ly, p.y), b.lz = fminf(b.lz, p.z);
      b.hx = fmaxf(b.hx, p.x), b.hy = fmaxf(b.hy, p.y), b.hz = fmaxf(b.hz, p.z);
    }
  }
  out[c] = b;
}

// Third level for large clouds (n > SORT_LDS_MAX_N): one box per kSuperGroups consecutive group boxes (8192 sorted points),
// stored behind the group boxes in the same buffer (gbox[ngroups + s]).  The searches test these first, so a wave looks
// at 64 group boxes only where its neighbourhood can be.
constexpr int kSuperGroups = 64;
__global__ void k_super_boxes(Box* gbox, int ngroups, int nsuper) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nsuper) return;
  const float inf = __builtin_inff();
  Box b{inf, inf, inf, -inf, -inf, -inf};
  for (int g = s * kSuperGroups; g < min((s + 1) * kSuperGroups, ngroups); g++) {
    const Box a = gbox[g];
    b.lx = fminf(b.lx, a.lx), b.ly = fminf(b.ly, a.ly), b.lz = fminf(b.lz, a.lz);
    b.hx = fmaxf(b.hx, a.hx), b.hy = fmaxf(b.hy, a.hy), b.hz = fmaxf(b.hz, a.hz);
  }
  gbox[ngroups + s] = b;
}

// un-permute helpers for the getters: out[perm[s]] = in[s]
template <typename T>
__global__ void k_scatter_by_perm(const T* in, const int* perm, int n, T* out) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < n) out[perm[s]] = in[s];
}

}  // namespace apd
