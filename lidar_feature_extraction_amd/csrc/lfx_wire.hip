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

// ------------------------------------------------------------------------------------------
// lfx_box_calibration: what this device gives right now -- a plain copy's rate and the shader clock under load -- printed
// by bench.py beside its numbers, so that a slower box and a slower kernel can be told apart in the driver's record.
namespace
{
typedef float calib_f4 __attribute__((ext_vector_type(4)));

// (one float4 per thread, no loop: the form that reaches the highest rate here -- 6.2 TB/s against 4.8-5.7 for grid-stride
// loops with 4 or 8 loads in flight, tools/membench)
__global__ __launch_bounds__(256) void calib_copy_kernel(const calib_f4 * __restrict__ src, calib_f4 * __restrict__ dst, size_t n)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {dst[i] = src[i];}
}

constexpr int kCalibChain = 1 << 17;       // dependent adds per wave
__global__ __launch_bounds__(64) void calib_clock_kernel(uint32_t * __restrict__ out)
{
  uint32_t a = threadIdx.x;
  for (int i = 0; i < kCalibChain / 64; i++) {
#pragma unroll
    for (int u = 0; u < 64; u++) {asm volatile ("v_add_u32 %0, %0, %0" : "+v"(a));}
  }
  if (a == 0x12345u) {out[0] = a;}
}
}  // namespace

extern "C" int lfx_box_calibration(lfx_ctx * c, size_t bytes, void * stream, double * copy_gbs, double * clock_mhz)
{
  if (!c || !copy_gbs || !clock_mhz) {return LFX_ERR_INVALID_ARGUMENT;}
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (bytes == 0) {bytes = (size_t)1 << 30;}
  bytes &= ~(size_t)4095;
  if (bytes < 4096) {return LFX_ERR_INVALID_ARGUMENT;}
  LFX_HIP(c, hipSetDevice(c->device));
  uint8_t * a = nullptr, * b = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipError_t err = hipMalloc(reinterpret_cast<void **>(&a), bytes);
  if (err == hipSuccess) {err = hipMalloc(reinterpret_cast<void **>(&b), bytes);}
  if (err == hipSuccess) {err = hipMemsetAsync(a, 1, bytes, st);}
  if (err == hipSuccess) {err = hipEventCreate(&e0);}
  if (err == hipSuccess) {err = hipEventCreate(&e1);}
  float best_copy = 0.f, best_clock = 0.f;
  if (err == hipSuccess) {
    const size_t n = bytes / 16;
    for (int rep = 0; rep < 4 && err == hipSuccess; rep++) {          // (the first one warms the pages up)
      (void)hipEventRecord(e0, st);
      hipLaunchKernelGGL(calib_copy_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const calib_f4 *>(a), reinterpret_cast<calib_f4 *>(b), n);
      (void)hipEventRecord(e1, st);
      err = hipEventSynchronize(e1);
      float ms = 0.f;
      if (err == hipSuccess) {err = hipEventElapsedTime(&ms, e0, e1);}
      if (rep > 0 && ms > 0.f && (best_copy == 0.f || ms < best_copy)) {best_copy = ms;}
    }
    // one wave per SIMD slot of every CU, four per SIMD: the clock the device holds with all its vector units busy
    hipDeviceProp_t prop;
    if (err == hipSuccess) {err = hipGetDeviceProperties(&prop, c->device);}
    for (int rep = 0; rep < 3 && err == hipSuccess; rep++) {
      (void)hipEventRecord(e0, st);
      hipLaunchKernelGGL(calib_clock_kernel, dim3((uint32_t)prop.multiProcessorCount * 16u), dim3(64), 0, st, reinterpret_cast<uint32_t *>(b));
      (void)hipEventRecord(e1, st);
      err = hipEventSynchronize(e1);
      float ms = 0.f;
      if (err == hipSuccess) {err = hipEventElapsedTime(&ms, e0, e1);}
      if (rep > 0 && ms > 0.f && (best_clock == 0.f || ms < best_clock)) {best_clock = ms;}
    }
  }
  if (e0) {(void)hipEventDestroy(e0);}
  if (e1) {(void)hipEventDestroy(e1);}
  if (a) {(void)hipFree(a);}
  if (b) {(void)hipFree(b);}
  if (err != hipSuccess) {
    c->err = std::string("lfx_box_calibration: ") + hipGetErrorString(err);
    return LFX_ERR_HIP;
  }
  *copy_gbs = best_copy > 0.f ? 2.0 * (double)bytes / (1e-3 * best_copy) / 1e9 : 0.0;
  // four waves share a SIMD-32, which takes a wave's 64 lanes in two cycles: a wave's chain advances one add every 8 cycles
  // (MI355X_MICROARCH.md, "Each CU has 4 SIMD-32 units"; measured: 2.17 GHz by this count on an MI355X)
  *clock_mhz = best_clock > 0.f ? (double)kCalibChain * 8.0 / (1e-3 * best_clock) / 1e6 : 0.0;
  return LFX_OK;
}
