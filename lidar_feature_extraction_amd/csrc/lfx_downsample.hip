// lfx_downsample.hip -- voxel-grid Downsample (lfx_kernels_downsample.hpp).
#include "lfx_internal.hpp"
#include "lfx_kernels_downsample.hpp"

using namespace lfx_host;

// ---------------------------------------------------------------------------- voxel-grid Downsample
namespace lfx_host
{
// lfx_voxel_downsample, with what lfx_localize_batch adds: a cloud PCL would hand back unfiltered is copied to the output
// (status still says so), and the clouds' lengths go to pinned host memory for the next call's launch sizes
int voxel_downsample(
  lfx_ctx * c, const float * d_points, const uint32_t * d_begin, const uint32_t * d_count, uint32_t count_stride,
  uint32_t n_clouds, size_t total_points, float leaf, float * d_out, uint32_t * d_out_count, uint32_t * d_status, void * stream,
  bool unfiltered, const uint32_t * d_other_count, uint32_t * lengths)
{
  if (!c || !d_points || !d_begin || !d_count || !d_out || !d_out_count || !d_status || n_clouds == 0 || count_stride == 0) {
    return LFX_ERR_INVALID_ARGUMENT;
  }
  if (!(leaf > 0.f)) {return fail(c, LFX_ERR_INVALID_ARGUMENT, "leaf size must be > 0");}
  LFX_HIP(c, hipSetDevice(c->device));
  if (c->vox_scratch.n < 4 * total_points) {          // (key, value) x 2 per point, grown on demand
    c->vox_scratch.release();
    if (c->vox_scratch.alloc(4 * total_points) != hipSuccess) {
      c->vox_scratch.n = 0;
      return fail(c, LFX_ERR_OUT_OF_MEMORY, "cannot allocate the sort scratch of the voxel grid");
    }
  }
  uint32_t * w = c->vox_scratch.p;
  // the small-cloud form's ranking table (96 KB), then its sorted points (144 KB) in LDS: more than a kernel gets without asking
  constexpr size_t table_bytes = (size_t)lfx::kVoxItems * lfx::kVoxThreads * 3 * sizeof(float);
  static_assert(table_bytes >= (size_t)lfx::kVoxItems * (lfx::kVoxThreads / 64) * 256 * sizeof(uint16_t), "the table fits where the points go");
  if (!c->vox_lds_asked) {
    LFX_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&lfx::voxel_downsample_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
      (int)table_bytes));
    c->vox_lds_asked = true;
  }
  hipLaunchKernelGGL(lfx::voxel_downsample_kernel, dim3(n_clouds), dim3(lfx::kVoxThreads), table_bytes, static_cast<hipStream_t>(stream),
    reinterpret_cast<const float4 *>(d_points), d_begin, d_count, count_stride, leaf, w, w + total_points, w + 2 * total_points,
    w + 3 * total_points, reinterpret_cast<float4 *>(d_out), d_out_count, d_status, unfiltered ? 1u : 0u, d_other_count, lengths);
  LFX_HIP(c, hipGetLastError());
  return LFX_OK;
}
}  // namespace lfx_host

extern "C" {

int lfx_voxel_downsample(
  lfx_ctx * c, const float * d_points, const uint32_t * d_begin, const uint32_t * d_count, uint32_t count_stride,
  uint32_t n_clouds, size_t total_points, float leaf, float * d_out, uint32_t * d_out_count, uint32_t * d_status, void * stream)
{
  return voxel_downsample(c, d_points, d_begin, d_count, count_stride, n_clouds, total_points, leaf, d_out, d_out_count, d_status, stream,
           false, nullptr, nullptr);
}

int lfx_downsample_surface(lfx_ctx * c, float leaf, float * d_out, uint32_t * d_out_count, uint32_t * d_status, void * stream)
{
  if (!c) {return LFX_ERR_INVALID_ARGUMENT;}
  if (c->last_batch == 0) {return fail(c, LFX_ERR_INVALID_ARGUMENT, "no batch has been extracted yet");}
  return lfx_voxel_downsample(c, reinterpret_cast<const float *>(c->surf_pts.p), c->scan_begin.p,
           c->scan_info.p + lfx::kInfoSurface, 4, c->last_batch, c->h_scan_begin[c->last_batch], leaf, d_out, d_out_count,
           d_status, stream);
}

}  // extern "C"
