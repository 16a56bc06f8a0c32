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
constexpr int kVoxThreads = 1024, kVoxItems = 12;

// bounds of the cloud (per-wave minima and maxima in red[0..2][w], red[3..5][w]) -> geo: min cell x, y, z; multipliers of y, z;
// radix passes; leaf too small
__device__ inline void voxel_geometry(const float (*red)[kVoxThreads / 64], int W, float inv, int * geo)
{
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

__global__ __launch_bounds__(kVoxThreads) void voxel_downsample_kernel(
  const float4 * __restrict__ pts, const uint32_t * __restrict__ begin, const uint32_t * __restrict__ count,
  uint32_t count_stride, float leaf, uint32_t * __restrict__ key_a, uint32_t * __restrict__ key_b,
  uint32_t * __restrict__ val_a, uint32_t * __restrict__ val_b, float4 * __restrict__ out,
  uint32_t * __restrict__ out_count, uint32_t * __restrict__ status,
  uint32_t unfiltered /* PCL hands a cloud whose leaf is too small back as it is: copy it to `out` (status stays 1) */,
  const uint32_t * __restrict__ other_count, uint32_t * __restrict__ lengths /* null, or [clouds][2] in pinned host memory:
     other_count[s * count_stride] and this cloud's output length, for lfx_localize_batch's next call */)
{
  constexpr int T = kVoxThreads, W = T / 64;
  const uint32_t s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t b = begin[s], n = count[(size_t)s * count_stride];
  __shared__ float red[6][W];
  __shared__ int geo[8];                      // min cell x, y, z; multipliers of y, z; radix passes; leaf too small
  __shared__ uint32_t hist[256], base[256], wtot[W];
  __shared__ uint16_t wcnt[W][256];
  extern __shared__ uint16_t table[];         // [tiles of the cloud][W][256], then the sorted points: the form for clouds of up to kVoxItems * T points
  auto finish = [&](uint32_t n_out, uint32_t st) __attribute__((always_inline)) {      // every thread calls it, at the end
      if (st == 1u && unfiltered) {
        for (uint32_t i = tid; i < n; i += T) {
          const float4 p = pts[b + i];
          out[b + i] = make_float4(p.x, p.y, p.z, 1.f);
        }
        n_out = n;
      }
      if (tid == 0) {
        out_count[s] = n_out; status[s] = st;
        if (lengths) {lengths[2 * s] = other_count[(size_t)s * count_stride]; lengths[2 * s + 1] = n_out;}
      }
    };
  if (n == 0) {
    finish(0u, 0u);
    return;
  }
  const float inv = 1.0f / leaf;              // inverse_leaf_size_
  if (n <= (uint32_t)(kVoxItems * T)) {
    // ---- A cloud of up to 12 288 points (a scan's surface cloud): the same steps with each thread's points and sort items in
    // registers -- every pass over the cloud is ONE round trip to memory with all its loads in flight, where the general
    // form below pays one per 1 024 points (its loops wait for each load: 110 round trips, 72 us for 10 000 points).
    constexpr int VI = kVoxItems;
    float4 p[VI];
#pragma unroll
    for (int t = 0; t < VI; t++) {
      const uint32_t i = (uint32_t)t * T + tid;
      p[t] = pts[b + (i < n ? i : 0u)];
    }
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int t = 0; t < VI; t++) {
      const uint32_t i = (uint32_t)t * T + tid;
      if (i < n) {
        mn[0] = fminf(mn[0], p[t].x); mn[1] = fminf(mn[1], p[t].y); mn[2] = fminf(mn[2], p[t].z);
        mx[0] = fmaxf(mx[0], p[t].x); mx[1] = fmaxf(mx[1], p[t].y); mx[2] = fmaxf(mx[2], p[t].z);
      }
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
    if (tid == 0) {voxel_geometry(red, W, inv, geo);}
    __syncthreads();
    if (geo[6]) {
      finish(0u, 1u);
      return;
    }
    const float fb0 = (float)geo[0], fb1 = (float)geo[1], fb2 = (float)geo[2];
    const int mul1 = geo[3], mul2 = geo[4], passes = geo[5];
    uint32_t k[VI], v[VI];
#pragma unroll
    for (int t = 0; t < VI; t++) {
      const int i0 = (int)(floorf(p[t].x * inv) - fb0), i1 = (int)(floorf(p[t].y * inv) - fb1), i2 = (int)(floorf(p[t].z * inv) - fb2);
      k[t] = (uint32_t)(i0 + i1 * mul1 + i2 * mul2);
      v[t] = (uint32_t)t * T + tid;
    }
    uint32_t * ks = key_a, * kd = key_b, * vs = val_a, * vd = val_b;
    for (int pass = 0; pass < passes; pass++) {
      const int shift = 8 * pass;
      if (pass > 0) {                                            // what the previous pass wrote, all of it at once
#pragma unroll
        for (int t = 0; t < VI; t++) {
          const uint32_t i = (uint32_t)t * T + tid;
          k[t] = ks[b + (i < n ? i : 0u)]; v[t] = vs[b + (i < n ? i : 0u)];
        }
      }
      // every tile's ranking at once: table[(tile, wave)][digit] = points of that digit in that wave of that tile (the waves'
      // ballots, as below), then ONE walk per digit over the (tile, wave) pairs in order turns the counts into the places
      // where each group starts inside its digit and leaves the digit's total -- four barriers per pass where a barrier-
      // ridden loop over the tiles took four per tile
      const uint32_t n_tiles = (n + T - 1) / T;
      for (uint32_t z = tid; z < n_tiles * W * 128u; z += T) {reinterpret_cast<uint32_t *>(table)[z] = 0u;}
      __syncthreads();
      uint32_t rank[VI], dig[VI];
#pragma unroll
      for (int t = 0; t < VI; t++) {
        rank[t] = 0u; dig[t] = 256u;
        if ((uint32_t)t * T >= n) {continue;}                  // (the same in every thread)
        const bool valid = (uint32_t)t * T + tid < n;
        const uint32_t d = valid ? (k[t] >> shift) & 255u : 256u;
        uint64_t peers = ~0ull;
#pragma unroll
        for (int bit = 0; bit < 9; bit++) {
          const bool set = (d >> bit) & 1u;
          const uint64_t m = __ballot(set);
          peers &= set ? m : ~m;
        }
        rank[t] = __popcll(peers & ((1ull << lane) - 1ull));
        dig[t] = d;
        if (valid && rank[t] == 0) {table[((uint32_t)t * W + wave) * 256u + d] = (uint16_t)__popcll(peers);}
      }
      __syncthreads();
      if (tid < 256) {
        uint32_t acc = 0;
        const uint32_t groups = n_tiles * W;
#pragma unroll 8
        for (uint32_t g = 0; g < groups; g++) {
          const uint32_t c = table[g * 256u + tid];
          table[g * 256u + tid] = (uint16_t)acc;
          acc += c;
        }
        hist[tid] = acc;
      }
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
#pragma unroll
      for (int t = 0; t < VI; t++) {
        if (dig[t] < 256u) {
          const uint32_t pos = base[dig[t]] + table[((uint32_t)t * W + wave) * 256u + dig[t]] + rank[t];
          kd[b + pos] = k[t];
          vd[b + pos] = v[t];
        }
      }
      __syncthreads();
      uint32_t * t_ = ks; ks = kd; kd = t_;
      t_ = vs; vs = vd; vd = t_;
    }
    // ---- the sorted keys with their predecessors, and the points in sorted order (two round trips)
    uint32_t kp[VI];
#pragma unroll
    for (int t = 0; t < VI; t++) {
      const uint32_t i = (uint32_t)t * T + tid, ii = i < n ? i : 0u;
      k[t] = ks[b + ii]; kp[t] = ks[b + (ii ? ii - 1u : 0u)]; v[t] = vs[b + ii];
    }
#pragma unroll
    for (int t = 0; t < VI; t++) {p[t] = pts[b + v[t]];}
    // (the ranking table is done with: its place in LDS takes the points in sorted order, x y z planes)
    float * sx_l = reinterpret_cast<float *>(table), * sy_l = sx_l + VI * T, * sz_l = sy_l + VI * T;
#pragma unroll
    for (int t = 0; t < VI; t++) {
      const uint32_t i = (uint32_t)t * T + tid;
      if (i < n) {sx_l[i] = p[t].x; sy_l[i] = p[t].y; sz_l[i] = p[t].z;}
    }
    // ---- cell heads, every tile at once: the waves' head counts per (tile, wave), one wave's prefix over them, then each
    // head's place; the places go to LDS when there are at most 2 047 cells (a scan's surface cloud has a few hundred)
    uint64_t hm[VI];
#pragma unroll
    for (int t = 0; t < VI; t++) {
      const uint32_t i = (uint32_t)t * T + tid;
      const bool head = i < n && (i == 0 || k[t] != kp[t]);
      hm[t] = __ballot(head);
      if (lane == 0) {hist[t * W + wave] = (uint32_t)__popcll(hm[t]);}
    }
    __syncthreads();
    static_assert(VI * W <= 256 && VI * W <= 4 * 64, "the head counts fit the bins' place and one wave's lanes, four each");
    if (tid < 64) {
      uint32_t c[4], sum = 0;
#pragma unroll
      for (int e = 0; e < 4; e++) {c[e] = 4 * tid + e < VI * W ? hist[4 * tid + e] : 0u; sum += c[e];}
      uint32_t incl = sum;
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d);
        if ((int)tid >= d) {incl += o;}
      }
      uint32_t ex = incl - sum;
#pragma unroll
      for (int e = 0; e < 4; e++) {if (4 * tid + e < VI * W) {hist[4 * tid + e] = ex;} ex += c[e];}
      if (tid == 63) {base[0] = incl;}
    }
    __syncthreads();
    const uint32_t m = base[0];
    uint32_t * heads_l = reinterpret_cast<uint32_t *>(&wcnt[0][0]);
    const bool heads_in_lds = m < (uint32_t)(W * 256 / 2);
#pragma unroll
    for (int t = 0; t < VI; t++) {
      const uint32_t i = (uint32_t)t * T + tid;
      if ((hm[t] >> lane) & 1ull) {
        const uint32_t at = hist[t * W + wave] + (uint32_t)__popcll(hm[t] & ((1ull << lane) - 1ull));
        if (heads_in_lds) {heads_l[at] = i;} else {kd[b + at] = i;}
      }
    }
    if (tid == 0) {if (heads_in_lds) {heads_l[m] = n;}}
    __syncthreads();
    // ---- centroids: the points of a cell in input order (AccumulatorXYZ: float sums, then / count), out of LDS: a cell of
    // a few hundred points near the sensor is a few hundred LDS reads for its thread, not as many trips to memory
    for (uint32_t c = tid; c < m; c += T) {
      const uint32_t a = heads_in_lds ? heads_l[c] : kd[b + c], e = heads_in_lds ? heads_l[c + 1] : (c + 1 < m ? kd[b + c + 1] : n);
      float sx = 0.f, sy = 0.f, sz = 0.f;
      for (uint32_t j = a; j < e; j++) {sx += sx_l[j]; sy += sy_l[j]; sz += sz_l[j];}
      const float cnt = (float)(e - a);
      out[b + c] = make_float4(sx / cnt, sy / cnt, sz / cnt, 1.0f);
    }
    finish(m, 0u);
    return;
  }
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
  if (tid == 0) {voxel_geometry(red, W, inv, geo);}
  __syncthreads();
  if (geo[6]) {
    finish(0u, 1u);
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
  finish(m, 0u);
}

}  // namespace lfx
