// lfx_kernels_downsample.hpp -- voxel-grid Downsample.
#pragma once

#include "lfx_kernels_common.hpp"

#pragma clang fp contract(off)

namespace lfx
{

// ------------------------------------------------------------------------------------------
// Voxel-grid Downsample: lib/include/lidar_feature_library/downsample.hpp:37-51 = pcl::VoxelGrid with
// one leaf size, which the localizer applies to the surface scan (localization/.../surface.hpp:111).  The arithmetic
// is PCL's (third party: parity unpinned): float bounds of the cloud, cell index =
// floor(x * (1 / leaf)) - min cell per axis, linear index, cells in ascending index order, centroid = float sum / count.
// One workgroup per cloud: bounds -> cell index per point -> stable LSD radix sort of (cell, point) by 8-bit digits
// (stable, so the points of a cell stay in input order: PCL leaves that order to an unstable sort, here it is defined)
// -> cell heads -> one thread per cell sums its points in that order.  Scratch: two (key, value) arrays per point.
constexpr int kVoxThreads = 1024;
__global__ __launch_bounds__(kVoxThreads) void voxel_downsample_kernel(
  const float4 * __restrict__ pts, const uint32_t * __restrict__ begin, const uint32_t * __restrict__ count,
  uint32_t count_stride, float leaf, uint32_t * __restrict__ key_a, uint32_t * __restrict__ key_b,
  uint32_t * __restrict__ val_a, uint32_t * __restrict__ val_b, float4 * __restrict__ out,
  uint32_t * __restrict__ out_count, uint32_t * __restrict__ status)
{
  constexpr int T = kVoxThreads, W = T / 64;
  const uint32_t s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t b = begin[s], n = count[(size_t)s * count_stride];
  __shared__ float red[6][W];
  __shared__ int geo[8];                      // min cell x, y, z; multipliers of y, z; radix passes; leaf too small
  __shared__ uint32_t hist[256], base[256], wtot[W];
  __shared__ uint16_t wcnt[W][256];
  if (n == 0) {
    if (tid == 0) {out_count[s] = 0; status[s] = 0;}
    return;
  }
  const float inv = 1.0f / leaf;              // inverse_leaf_size_
  // ---- bounds (getMinMax3D)
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (uint32_t i = tid; i < n; i += T) {
    const float4 p = pts[b + i];
    mn[0] = fminf(mn[0], p.x); mn[1] = fminf(mn[1], p.y); mn[2] = fminf(mn[2], p.z);
    mx[0] = fmaxf(mx[0], p.x); mx[1] = fmaxf(mx[1], p.y); mx[2] = fmaxf(mx[2], p.z);
  }
#pragma unroll
  for (int a = 0; a < 3; a++) {
    for (int o = 32; o > 0; o >>= 1) {
      mn[a] = fminf(mn[a], __shfl_xor(mn[a], o));
      mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o));
    }
    if (lane == 0) {red[a][wave] = mn[a]; red[3 + a][wave] = mx[a];}
  }
  __syncthreads();
  if (tid == 0) {
    float lo[3], hi[3];
    for (int a = 0; a < 3; a++) {
      lo[a] = red[a][0]; hi[a] = red[3 + a][0];
      for (int w = 1; w < W; w++) {lo[a] = fminf(lo[a], red[a][w]); hi[a] = fmaxf(hi[a], red[3 + a][w]);}
    }
    const long long dx = (long long)((hi[0] - lo[0]) * inv) + 1, dy = (long long)((hi[1] - lo[1]) * inv) + 1,
      dz = (long long)((hi[2] - lo[2]) * inv) + 1;
    // PCL: "leaf size is too small for the input dataset" (also catches non-finite bounds); factor by factor, three extents of
    // a few million cells overflow 64 bits
    const long long lim = 2147483647LL;
    int bad = !(dx >= 1 && dy >= 1 && dz >= 1 && dx <= lim && dy <= lim && dz <= lim && dx * dy <= lim && dx * dy * dz <= lim);
    int min_b[3], div_b[3];
    for (int a = 0; a < 3; a++) {
      min_b[a] = (int)floorf(lo[a] * inv);
      div_b[a] = (int)floorf(hi[a] * inv) - min_b[a] + 1;
    }
    long long cells = 0;
    if (!bad && div_b[0] > 0 && div_b[1] > 0 && div_b[2] > 0 && (long long)div_b[0] * div_b[1] <= lim) {
      cells = (long long)div_b[0] * div_b[1] * div_b[2];
    }
    if (!(cells > 0 && cells <= lim)) {bad = 1;}
    const uint32_t maxkey = bad ? 0u : (uint32_t)(cells - 1);
    geo[0] = min_b[0]; geo[1] = min_b[1]; geo[2] = min_b[2];
    geo[3] = div_b[0]; geo[4] = div_b[0] * div_b[1];
    geo[5] = maxkey == 0u ? 1 : (32 - __clz((int)maxkey) + 7) / 8;
    geo[6] = bad;
  }
  __syncthreads();
  if (geo[6]) {
    if (tid == 0) {out_count[s] = 0; status[s] = 1;}
    return;
  }
  const float fb0 = (float)geo[0], fb1 = (float)geo[1], fb2 = (float)geo[2];
  const int mul1 = geo[3], mul2 = geo[4], passes = geo[5];
  // ---- cell index per point
  for (uint32_t i = tid; i < n; i += T) {
    const float4 p = pts[b + i];
    const int i0 = (int)(floorf(p.x * inv) - fb0), i1 = (int)(floorf(p.y * inv) - fb1), i2 = (int)(floorf(p.z * inv) - fb2);
    key_a[b + i] = (uint32_t)(i0 + i1 * mul1 + i2 * mul2);
    val_a[b + i] = i;
  }
  __syncthreads();
  // ---- stable LSD radix sort, 8 bits per pass
  uint32_t * ks = key_a, * kd = key_b, * vs = val_a, * vd = val_b;
  for (int pass = 0; pass < passes; pass++) {
    const int shift = 8 * pass;
    if (tid < 256) {hist[tid] = 0;}
    __syncthreads();
    for (uint32_t i = tid; i < n; i += T) {atomicAdd(&hist[(ks[b + i] >> shift) & 255u], 1u);}
    __syncthreads();
    if (tid < 64) {                                           // exclusive scan of the 256 bins by one wave, four bins per lane
      const uint32_t h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
      uint32_t incl = h0 + h1 + h2 + h3;
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(incl, d);
        if ((int)tid >= d) {incl += t;}
      }
      const uint32_t ex = incl - (h0 + h1 + h2 + h3);
      base[4 * tid] = ex; base[4 * tid + 1] = ex + h0; base[4 * tid + 2] = ex + h0 + h1; base[4 * tid + 3] = ex + h0 + h1 + h2;
    }
    __syncthreads();
    for (uint32_t t0 = 0; t0 < n; t0 += T) {
      for (int z = tid; z < W * 256; z += T) {(&wcnt[0][0])[z] = 0;}
      __syncthreads();
      const uint32_t i = t0 + tid;
      const bool valid = i < n;
      const uint32_t k = valid ? ks[b + i] : 0u, v = valid ? vs[b + i] : 0u;
      const uint32_t d = valid ? (k >> shift) & 255u : 256u;
      uint64_t peers = ~0ull;
#pragma unroll
      for (int bit = 0; bit < 9; bit++) {
        const bool set = (d >> bit) & 1u;
        const uint64_t m = __ballot(set);
        peers &= set ? m : ~m;
      }
      const uint32_t rank = __popcll(peers & ((1ull << lane) - 1ull));
      if (valid && rank == 0) {wcnt[wave][d] = (uint16_t)__popcll(peers);}
      __syncthreads();
      uint32_t tile_total = 0;
      if (tid < 256) {
        for (int w = 0; w < W; w++) {
          const uint32_t c = wcnt[w][tid];
          wcnt[w][tid] = (uint16_t)tile_total;
          tile_total += c;
        }
      }
      __syncthreads();
      if (valid) {
        const uint32_t pos = base[d] + wcnt[wave][d] + rank;
        kd[b + pos] = k;
        vd[b + pos] = v;
      }
      __syncthreads();
      if (tid < 256) {base[tid] += tile_total;}
    }
    __syncthreads();
    uint32_t * t = ks; ks = kd; kd = t;
    t = vs; vs = vd; vd = t;
  }
  // ---- cell heads: position of every cell's first point (kept in kd), number of cells
  uint32_t cells_before = 0;
  for (uint32_t t0 = 0; t0 < n; t0 += T) {
    const uint32_t i = t0 + tid;
    const bool head = i < n && (i == 0 || ks[b + i] != ks[b + i - 1]);
    const uint64_t hm = __ballot(head);
    if (lane == 0) {wtot[wave] = __popcll(hm);}
    __syncthreads();
    uint32_t before = cells_before, total = 0;
    for (int w = 0; w < W; w++) {
      if (w < (int)wave) {before += wtot[w];}
      total += wtot[w];
    }
    if (head) {kd[b + before + __popcll(hm & ((1ull << lane) - 1ull))] = i;}
    cells_before += total;
    __syncthreads();
  }
  const uint32_t m = cells_before;
  __syncthreads();
  // ---- centroids: the points of a cell in input order (AccumulatorXYZ: float sums, then / count)
  for (uint32_t c = tid; c < m; c += T) {
    const uint32_t a = kd[b + c], e = c + 1 < m ? kd[b + c + 1] : n;
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (uint32_t jdx = a; jdx < e; jdx++) {
      const float4 p = pts[b + vs[b + jdx]];
      sx += p.x; sy += p.y; sz += p.z;
    }
    const float cnt = (float)(e - a);
    out[b + c] = make_float4(sx / cnt, sy / cnt, sz / cnt, 1.0f);
  }
  if (tid == 0) {out_count[s] = m; status[s] = 0;}
}

}  // namespace lfx
