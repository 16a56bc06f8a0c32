// lfx_wire.hip -- the clouds as they travel: packed per-batch buffers for the gather, PointCloud2 payloads of scan_edge /
// scan_surface, colored_scan (lfx_kernels_wire.hpp).
#include "lfx_internal.hpp"
#include "lfx_kernels_wire.hpp"

using namespace lfx_host;

extern "C" {

namespace
{
int pack_clouds(
  lfx_ctx * c, float * d_edge_out, float * d_surface_out, uint32_t * d_offsets_out, size_t capacity_points,
  void * stream, uint32_t xyz_wire);
}

int lfx_pack_features(
  lfx_ctx * c, float * d_edge_out, float * d_surface_out, uint32_t * d_offsets_out, size_t capacity_points,
  void * stream)
{
  return pack_clouds(c, d_edge_out, d_surface_out, d_offsets_out, capacity_points, stream, 0u);
}

int lfx_pack_xyz(
  lfx_ctx * c, float * d_edge_out, float * d_surface_out, uint32_t * d_offsets_out, size_t capacity_points,
  void * stream)
{
  return pack_clouds(c, d_edge_out, d_surface_out, d_offsets_out, capacity_points, stream, 1u);
}

int lfx_pack_xyz12(
  lfx_ctx * c, float * d_edge_out, float * d_surface_out, uint32_t * d_offsets_out, size_t capacity_points,
  void * stream)
{
  return pack_clouds(c, d_edge_out, d_surface_out, d_offsets_out, capacity_points, stream, 2u);
}

int lfx_pack_colored(lfx_ctx * c, float * d_colored_out, uint32_t * d_offsets_out, size_t capacity_points, void * stream)
{
  if (!c || !d_colored_out || !d_offsets_out) {return LFX_ERR_INVALID_ARGUMENT;}
  if (c->last_batch == 0 || !c->last_points) {return fail(c, LFX_ERR_INVALID_ARGUMENT, "no batch has been extracted yet");}
  LFX_HIP(c, hipSetDevice(c->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  const uint32_t batch = c->last_batch;
  const uint32_t capacity = capacity_points > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)capacity_points;
  hipLaunchKernelGGL(lfx::colored_offsets_kernel, dim3(1), dim3(256), 0, st, c->ring_count.p, c->ring_status.p, batch,
    c->max_rings, d_offsets_out);
  hipLaunchKernelGGL(lfx::colored_pack_kernel, dim3(c->max_rings, batch), dim3(256), 0, st,
    c->ring_count.p, c->ring_status.p, d_offsets_out, c->sxy.p, c->sidx.p, c->label_s.p,
    static_cast<const uint8_t *>(c->last_points), c->layout, c->scan_begin.p, c->max_rings, c->cap,
    reinterpret_cast<float4 *>(d_colored_out), capacity, c->scan_info.p, c->xform.p);
  LFX_HIP(c, hipGetLastError());
  return LFX_OK;
}

namespace
{
int pack_clouds(
  lfx_ctx * c, float * d_edge_out, float * d_surface_out, uint32_t * d_offsets_out, size_t capacity_points,
  void * stream, uint32_t xyz_wire)
{
  if (!c || !d_edge_out || !d_surface_out || !d_offsets_out) {return LFX_ERR_INVALID_ARGUMENT;}
  if (c->last_batch == 0) {return fail(c, LFX_ERR_INVALID_ARGUMENT, "no batch has been extracted yet");}
  LFX_HIP(c, hipSetDevice(c->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  const uint32_t batch = c->last_batch;
  const uint32_t capacity = capacity_points > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)capacity_points;
  hipLaunchKernelGGL(lfx::feature_offsets_kernel, dim3(1), dim3(256), 0, st, c->scan_info.p, batch, d_offsets_out);
  hipLaunchKernelGGL(lfx::feature_pack_kernel, dim3(8, batch), dim3(256), 0, st,
    c->scan_begin.p, c->scan_info.p, d_offsets_out, batch, c->edge_pts.p, c->surf_pts.p,
    reinterpret_cast<float4 *>(d_edge_out), reinterpret_cast<float4 *>(d_surface_out), capacity, xyz_wire);
  LFX_HIP(c, hipGetLastError());
  return LFX_OK;
}
}  // namespace

// ---------------------------------------------------------------------------- colored_scan
int lfx_label_to_color(uint8_t label, uint8_t rgb[3])   // color_points.cpp:39-68
{
  static const uint8_t table[8][3] = {
    {255, 255, 255},   // Default
    {255, 0, 0},       // Edge
    {255, 63, 0},      // EdgeNeighbor
    {255, 0, 0},       // Surface
    {255, 63, 0},      // SurfaceNeighbor
    {127, 127, 127},   // OutOfRange
    {255, 0, 255},     // Occluded
    {0, 255, 0}};      // ParallelBeam
  if (!rgb || label > LFX_LABEL_PARALLEL_BEAM) {return LFX_ERR_INVALID_ARGUMENT;}
  rgb[0] = table[label][0]; rgb[1] = table[label][1]; rgb[2] = table[label][2];
  return LFX_OK;
}

int lfx_color_points_by_label(const lfx_ctx * c, const void * points, size_t n_points, const uint8_t * labels, float * out)
{
  if (!c || (!points && n_points) || !labels || !out) {return LFX_ERR_INVALID_ARGUMENT;}
  const uint8_t * p = static_cast<const uint8_t *>(points);
  for (size_t i = 0; i < n_points; i++) {
    uint8_t rgb[3];
    if (lfx_label_to_color(labels[i], rgb) != LFX_OK) {return LFX_ERR_INVALID_ARGUMENT;}
    const uint8_t * q = p + i * c->layout.step;
    for (int a = 0; a < 3; a++) {
      uint32_t v;
      std::memcpy(&v, q + (a == 0 ? c->layout.ox : (a == 1 ? c->layout.oy : c->layout.oz)), 4);
      if (c->layout.be) {v = __builtin_bswap32(v);}
      std::memcpy(&out[4 * i + a], &v, 4);
    }
    const uint32_t packed = 0xFF000000u | ((uint32_t)rgb[0] << 16) | ((uint32_t)rgb[1] << 8) | (uint32_t)rgb[2];
    std::memcpy(&out[4 * i + 3], &packed, 4);
  }
  return LFX_OK;
}

}  // extern "C"
