// lfx_kernels_wire.hpp -- packed clouds for the gather and the PointCloud2 payloads, colored_scan.
#pragma once

#include "lfx_kernels_common.hpp"

#pragma clang fp contract(off)

namespace lfx
{

// ------------------------------------------------------------------------------------------
// Batch-level packing for the multi-GPU gather: exclusive prefix of the per-scan feature counts
// (one workgroup walks the batch), then a copy of every scan's clouds to its packed offset.
__global__ __launch_bounds__(256) void feature_offsets_kernel(
  const uint32_t * __restrict__ scan_info, uint32_t batch, uint32_t * __restrict__ offsets /* [2][batch+1] */)
{
  __shared__ uint32_t part[2][256];
  const uint32_t tid = threadIdx.x;
  const uint32_t per = (batch + 255) / 256;
  const uint32_t lo = tid * per, hi = (lo + per < batch) ? lo + per : batch;
  uint32_t e = 0, s = 0;
  for (uint32_t k = lo; k < hi; k++) {e += scan_info[k * 4 + kInfoEdge]; s += scan_info[k * 4 + kInfoSurface];}
  part[0][tid] = e;
  part[1][tid] = s;
  __syncthreads();
  for (uint32_t d = 1; d < 256; d <<= 1) {
    const uint32_t a = tid >= d ? part[0][tid - d] : 0u, b = tid >= d ? part[1][tid - d] : 0u;
    __syncthreads();
    part[0][tid] += a;
    part[1][tid] += b;
    __syncthreads();
  }
  uint32_t ce = part[0][tid] - e, cs = part[1][tid] - s;
  for (uint32_t k = lo; k < hi; k++) {
    offsets[k] = ce;
    offsets[batch + 1 + k] = cs;
    ce += scan_info[k * 4 + kInfoEdge];
    cs += scan_info[k * 4 + kInfoSurface];
  }
  if (tid == 255) {
    offsets[batch] = part[0][255];
    offsets[2 * batch + 1] = part[1][255];
  }
}

__global__ __launch_bounds__(256) void feature_pack_kernel(
  const uint32_t * __restrict__ scan_begin, const uint32_t * __restrict__ scan_info,
  const uint32_t * __restrict__ offsets, uint32_t batch, const float4 * __restrict__ edge_pts,
  const float4 * __restrict__ surf_pts, float4 * __restrict__ edge_out, float4 * __restrict__ surf_out,
  uint32_t capacity, uint32_t xyz_wire)
{
  const uint32_t s = blockIdx.y;
  const uint32_t ne = scan_info[s * 4 + kInfoEdge], ns = scan_info[s * 4 + kInfoSurface];
  const size_t b = scan_begin[s];
  const uint32_t oe = offsets[s], os = offsets[batch + 1 + s];
  for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < ne + ns; k += gridDim.x * blockDim.x) {
    const bool edge = k < ne;
    const uint32_t q = edge ? k : k - ne;
    float4 v = edge ? edge_pts[b + q] : surf_pts[b + q];
    if (xyz_wire == 2u) {
      // tight x, y, z (12 bytes per point): what has to travel when the clouds are gathered to one GPU
      float * o = reinterpret_cast<float *>(edge ? edge_out : surf_out) + 3 * (size_t)((edge ? oe : os) + q);
      if ((edge ? oe : os) + q < capacity) {o[0] = v.x; o[1] = v.y; o[2] = v.z;}
      continue;
    }
    if (xyz_wire) {v.w = 1.0f;}          // pcl::PointXYZ: data[3] = 1 (the curvature travels in lfx_pack_features only)
    if (edge) {
      if (oe + q < capacity) {edge_out[oe + q] = v;}
    } else {
      if (os + q < capacity) {surf_out[os + q] = v;}
    }
  }
}

// colored_scan (feature_extraction.cpp:153,161; color_points.hpp:60-74): per scan the points of every
// labelled ring, rings ascending, angle ascending, as 32-byte pcl::PointXYZRGB wire records.
__global__ __launch_bounds__(256) void colored_offsets_kernel(
  const uint32_t * __restrict__ ring_count, const uint8_t * __restrict__ ring_status, uint32_t batch, uint32_t max_rings,
  uint32_t * __restrict__ offsets /* [batch+1] */)
{
  __shared__ uint32_t part[256];
  const uint32_t tid = threadIdx.x;
  const uint32_t per = (batch + 255) / 256;
  const uint32_t lo = tid * per, hi = (lo + per < batch) ? lo + per : batch;
  auto scan_total = [&](uint32_t s) {
      uint32_t t = 0;
      for (uint32_t r = 0; r < max_rings; r++) {
        if (ring_status[s * kRings + r] == kOk) {t += ring_count[s * kRings + r];}
      }
      return t;
    };
  uint32_t sum = 0;
  for (uint32_t k = lo; k < hi; k++) {sum += scan_total(k);}
  part[tid] = sum;
  __syncthreads();
  for (uint32_t d = 1; d < 256; d <<= 1) {
    const uint32_t a = tid >= d ? part[tid - d] : 0u;
    __syncthreads();
    part[tid] += a;
    __syncthreads();
  }
  uint32_t c = part[tid] - sum;
  for (uint32_t k = lo; k < hi; k++) {
    offsets[k] = c;
    c += scan_total(k);
  }
  if (tid == 255) {offsets[batch] = part[255];}
}

__global__ __launch_bounds__(256) void colored_pack_kernel(
  const uint32_t * __restrict__ ring_count, const uint8_t * __restrict__ ring_status,
  const uint32_t * __restrict__ offsets, const float2 * __restrict__ sxy, const uint32_t * __restrict__ sidx,
  const uint8_t * __restrict__ label_s, const uint8_t * __restrict__ pts, Layout L,
  const uint32_t * __restrict__ scan_begin, uint32_t max_rings, uint32_t cap, float4 * __restrict__ out,
  uint32_t capacity, const uint32_t * __restrict__ scan_info, const uint32_t * __restrict__ xform)
{
  const uint32_t s = blockIdx.y, ring = blockIdx.x, tid = threadIdx.x;
  if (ring_status[s * kRings + ring] != kOk) {return;}
  const bool org = scan_is_organised(scan_info[s * 4 + kInfoError]) || scan_took_holes(scan_info[s * 4 + kInfoError]);     // nothing was staged: every field from the record
  const bool grid = scan_is_grid(scan_info[s * 4 + kInfoError]);         // ... and the index from the position (the holes form keeps sidx)
  __shared__ uint32_t before;
  if (tid == 0) {before = 0;}
  __syncthreads();
  if (tid < ring && ring_status[s * kRings + tid] == kOk) {atomicAdd(&before, ring_count[s * kRings + tid]);}
  __syncthreads();
  const uint32_t n = ring_count[s * kRings + ring];
  const size_t off = ring_base(s, ring, max_rings, cap);
  const uint32_t at = offsets[s] + before;
  // color_points.cpp:39-68, indexed by label; a = 255 as in a default-constructed pcl::PointXYZRGB
  const uint32_t table[8] = {0xFFFFFFFFu, 0xFFFF0000u, 0xFFFF3F00u, 0xFFFF0000u, 0xFFFF3F00u, 0xFF7F7F7Fu, 0xFFFF00FFu, 0xFF00FF00u};
  for (uint32_t i = tid; i < n; i += blockDim.x) {
    if (at + i >= capacity) {break;}
    const uint32_t orig = grid ? ring_column(xform[s * kRings + ring], i, n) * max_rings + ring : sidx[off + i];
    const uint8_t * rec = pts + ((size_t)scan_begin[s] + orig) * L.step;
    const float2 xy = org ? make_float2(load_f32(rec + L.ox, L.be), load_f32(rec + L.oy, L.be)) : sxy[off + i];
    // z from the input record (the staged z of a ring the workgroup-per-ring kernel sorted itself is not re-ordered)
    const float z = load_f32(rec + L.oz, L.be);
    out[2 * (size_t)(at + i)] = make_float4(xy.x, xy.y, z, 1.0f);
    out[2 * (size_t)(at + i) + 1] = make_float4(__uint_as_float(table[label_s[off + i] & 7u]), 0.f, 0.f, 0.f);
  }
}

}  // namespace lfx
