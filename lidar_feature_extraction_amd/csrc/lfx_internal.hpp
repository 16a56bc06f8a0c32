// lfx_internal.hpp -- what the translation units of liblfx.so share: the context behind lfx_ctx, device / pinned
// buffers, error plumbing.  Not part of the C ABI (include/lfx.h is).
#pragma once

#include "../../include/lfx.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "lfx_kernels_common.hpp"

namespace lfx_host
{

// The unit kernels live in four translation units of their own, one per variant of the parameters (lfx_unit_v0.hip ...
// lfx_unit_v3.hip over lfx_unit_variant.inl; UnitVariant in lfx_kernels_unit.hpp); run_batch launches them through these.
// the padding the unit kernels of variant V are compiled for (lfx::UnitVariant<V>::kPT: lfx_unit_variant.inl checks; 0 = at run time)
constexpr int kUnitVariantPadding[4] = {5, 5, 2, 0};

struct UnitOrgArgs
{
  lfx::Params prm;
  uint32_t cap, flags, max_rings, drop_zero;
  const uint8_t * pts;
  const uint32_t * scan_begin;
  uint32_t * ring_count;
  const lfx::UnitTables * tab;
  const uint32_t * xform, * geom;
};
struct UnitArgs
{
  lfx::Params prm;
  uint32_t cap, flags, max_rings;
  const uint32_t * ring_count;
  const float2 * sxy;
  const float * sz;
  const uint32_t * sidx;
  const lfx::UnitTables * tab;
  uint32_t * defer_count, * defer_list;
  const uint32_t * list_count, * list;
  uint32_t redo_cap;
};
#define LFX_DECLARE_UNIT_LAUNCHERS(V) \
  void launch_unit_org_v##V(int chunks, bool xf, bool holes, dim3 grid, uint32_t lds_pad, hipStream_t st, const UnitOrgArgs & a); \
  void launch_unit_v##V(bool second, int chunks, bool loop, dim3 grid, uint32_t lds_pad, hipStream_t st, const UnitArgs & a);
LFX_DECLARE_UNIT_LAUNCHERS(0) LFX_DECLARE_UNIT_LAUNCHERS(1) LFX_DECLARE_UNIT_LAUNCHERS(2) LFX_DECLARE_UNIT_LAUNCHERS(3)
#undef LFX_DECLARE_UNIT_LAUNCHERS
inline void launch_unit_org(int variant, int chunks, bool xf, bool holes, dim3 grid, uint32_t lds_pad, hipStream_t st, const UnitOrgArgs & a)
{
  switch (variant) {
    case 0: launch_unit_org_v0(chunks, xf, holes, grid, lds_pad, st, a); break;
    case 1: launch_unit_org_v1(chunks, xf, holes, grid, lds_pad, st, a); break;
    case 2: launch_unit_org_v2(chunks, xf, holes, grid, lds_pad, st, a); break;
    default: launch_unit_org_v3(chunks, xf, holes, grid, lds_pad, st, a); break;
  }
}
inline void launch_unit(int variant, bool second, int chunks, bool loop, dim3 grid, uint32_t lds_pad, hipStream_t st, const UnitArgs & a)
{
  switch (variant) {
    case 0: launch_unit_v0(second, chunks, loop, grid, lds_pad, st, a); break;
    case 1: launch_unit_v1(second, chunks, loop, grid, lds_pad, st, a); break;
    case 2: launch_unit_v2(second, chunks, loop, grid, lds_pad, st, a); break;
    default: launch_unit_v3(second, chunks, loop, grid, lds_pad, st, a); break;
  }
}

template<typename T>
struct DevBuf
{
  T * p = nullptr;
  size_t n = 0;
  hipError_t alloc(size_t count)
  {
    n = count;
    return hipMalloc(reinterpret_cast<void **>(&p), count * sizeof(T) + 16);
  }
  void release()
  {
    if (p) {(void)hipFree(p);}
    p = nullptr;
  }
};

struct HostScan          // the per-ring lists of one scan (the large arrays live in the pinned result block)
{
  std::vector<uint8_t> ring_status;
  std::vector<uint32_t> ring_count, ring_offset;
  std::vector<uint16_t> ring_id;
};

// Pinned host memory that only ever grows (the results handed to the caller stay valid until the next call).
struct PinnedBuf
{
  uint8_t * p = nullptr;
  size_t bytes = 0;
  hipError_t reserve(size_t need)
  {
    if (need <= bytes) {return hipSuccess;}
    // (growing is rare; a copy queued on the old block by a call that returned early with an error must not outlive it)
    if (p) {(void)hipDeviceSynchronize(); (void)hipHostFree(p);}
    p = nullptr;
    bytes = 0;
    const size_t want = need + need / 4 + 4096;
    const hipError_t e = hipHostMalloc(reinterpret_cast<void **>(&p), want, hipHostMallocDefault);
    if (e == hipSuccess) {bytes = want;}
    return e;
  }
  void release()
  {
    if (p) {(void)hipHostFree(p);}
    p = nullptr;
    bytes = 0;
  }
};

}  // namespace lfx_host

// The route selection of run_batch (lfx_api.hip, choose_route): what it remembers between batches, the switches that pin
// its choices (tests), what it decides for one batch.
struct RouteState
{
  uint32_t report[lfx::kCounters] = {};  // the counters block of the last batch whose report has landed (lfx_kernels_common.hpp)
  uint32_t report_rings = 0;             // rings of that batch (0 = no report yet)
  bool use_xform = false;                // rings arrive rotated / reversed: ring_cut_kernel ahead of the organised-scan kernel
  bool use_holes = false;                // the grid holds (0, 0, 0) records the zero filter drops: grid_count_kernel + the holes form
  bool bucket_all = false;               // the stream is not organised: bucketing route for every scan
  uint32_t retry_in = 0;                 // ... and the organised-scan kernel is tried again in so many batches
  bool pre_order = false;                // order repair BEFORE the first unit pass (a stream that keeps arriving out of order)
};
struct RoutePins                         // LFX_DEBUG_FUSED / _XFORM / _SHORT_TAIL / _PRE_ORDER (0 / 1; -1 = not pinned), _REDO_CAP (0 = not)
{
  int fused = -1, xform = -1, short_tail = -1, pre_order = -1, holes = -1;
  uint32_t redo_cap = 0;
};
struct RouteChoice
{
  bool fused = false, xform = false, short_tail = false, pre_order = false, holes = false;
  uint32_t fb_grid = 0, redo_cap = 0;
};

struct lfx_ctx
{
  int device = 0;
  lfx_params params{};
  lfx::Params dev{};
  lfx::Layout layout{};
  uint32_t max_points = 0, max_batch = 0, cap = 0, max_chunks = 0, max_rings = 0, ring_threads = 0, slow_grid = 0;
  size_t total_cap = 0, ring_lds = 0, order_lds = 0;
  uint32_t stage_flags = LFX_STAGE_ALL;  // LFX_DEBUG_RING_FLAGS overrides it for slow-kernel ablations (wrong results)
  uint32_t unit_lds_pad = 0;             // LFX_DEBUG_UNIT_LDS_PAD: extra LDS per workgroup (occupancy experiments)
  int unit_variant = 3;                  // which compilation of the unit kernels serves the parameters (UnitVariant, lfx_kernels_unit.hpp)
  uint32_t unit_chunks = 6;              // chunks of 64 positions per unit wave (3..6), from the configured ring length
  uint32_t unit_flags = 65u;             // LFX_DEBUG_UNIT_FLAGS: 1 edge pass, 64 surface pass (ablations only)
  uint32_t drop_zero = 0;                // lfx_config::drop_zero_points
  // What a batch reports about its stream (the counters block behind ring_flags) is copied to pinned memory at the end of
  // the batch, nobody waiting; the next batches' route is chosen from the last report that has LANDED.
  // No event stands behind it: a host that runs ahead of the device (a loop of lfx_extract_batch_device calls, the pipelined
  // pair) would ask about the batch it has just queued and never find that event passed.  The block carries the batch's
  // serial number before and after the counters (feature_compact_kernel writes begin, fence, counters, fence, end; the
  // host reads end, counters, begin): a block whose two serials agree is one batch's report, whichever batch has got that
  // far -- the route follows the stream a few batches behind instead of not at all.
  uint32_t * h_counters = nullptr;       // pinned [1 + lfx::kCounters + 1]
  uint32_t batch_serial = 0;             // serial of the batch queued last (never 0 in the block)
  uint32_t report_taken = 0;             // serial of the report the route state was last fed with
  RouteState route;
  RoutePins route_pins;
  bool pre_order = false;                // this batch's choice (run_batch)
  // The organised-scan kernel (lfx_kernels_extract.hpp, unit_body<ORG>) reads a driver's column-major scan directly; scans
  // that are not of that form fall back to the bucketing route inside the same call (choose_route decides per batch).
  bool fused_possible = false;
  bool last_used_xform = false;          // the last batch's organised-scan kernel ran with the ring transforms
  int totals_env = -1;                   // LFX_DEBUG_TOTALS_KERNEL=1: ring_totals_kernel also for small batches (A/B)
  bool fast_path = true;                 // wave-per-unit kernel first, workgroup-per-ring kernel for what it defers
  std::string err;
  lfx_log_fn log_cb = nullptr;           // lfx_set_log_callback
  void * log_user = nullptr;

  // device scratch
  lfx_host::DevBuf<uint32_t> scan_begin, scan_geom, scan_info, chunk_base, chunk_flags, ring_count, ring_nedge,
    ring_nsurf, ring_ebase, ring_sbase, ring_flags, unit_ne, unit_ns, unit_span, slow_list, defer_list, redo_list, fb_list, xform,
    sidx, rec_idx, edge_idx,
    surf_idx, d_sidx, counters, scan_flags, tail_ticket;
  // Ring ids that are not 0 .. max_rings-1 (lfx_config.ring_ids, lfx_set_ring_ids, or found by the host entry points in a
  // scan that carried another id): slot_id[k] = the id of slot k (ascending), ring_slot the device table id -> slot the
  // bucketing kernel reads; empty = the id is the slot.  ids_given: the caller said so (no looking up by the library).
  lfx_host::DevBuf<uint16_t> ring_slot;
  std::vector<uint16_t> slot_id;
  bool ids_given = false, organised_by_config = false;
  lfx_host::DevBuf<uint16_t> cum16;      // grid_count_kernel's prefix table (the holes form; contexts with the zero filter on)
  lfx_host::DevBuf<uint4> hole_desc;     // ... and its unit descriptors
  // The batch's accumulators (counters, scan_flags, ring_nedge / ring_nsurf) exist twice: batch k uses set `parity`, its
  // compaction zeroes the other one over the scans the batch before last left dirty there (par_dirty).  aux_dirty: scans whose
  // bucketing-route tables (look-back flags, ring flags, ring transforms) a batch since the last reset may have touched.
  uint32_t parity = 0, par_dirty[2] = {0u, 0u}, aux_dirty = 0;
  uint32_t scan_count_from = 256;        // batches of so many scans take scan_count_kernel for the holes form's count pass (LFX_DEBUG_SCAN_COUNT_FROM)
  lfx_host::DevBuf<uint8_t> ring_status, label_s, staging, d_label;
  lfx_host::DevBuf<double> d_curv;
  lfx_host::DevBuf<float2> sxy;
  lfx_host::DevBuf<float> sz;
  lfx_host::DevBuf<double> curv_s;
  lfx_host::DevBuf<float4> edge_pts, surf_pts, rec_pts;
  lfx_host::DevBuf<float4> rec32;         // the record slots of the unit kernels (lfx_kernels_common.hpp rec_slot_places)
  uint32_t slot_places = 64;              // places per slot: what the context's unit kernels are compiled for
  lfx_host::DevBuf<lfx::UnitTables> unit_tab;      // the unit kernel's output pointers (one element)
  lfx_host::DevBuf<uint32_t> vox_scratch;          // lfx_voxel_downsample: sort keys / values, allocated on first use
  lfx_host::DevBuf<double> align_scratch;          // lfx_scan_to_map_align: states, rows, errors; allocated on first use
  lfx_host::DevBuf<float> align_surface;           // lfx_localize_batch: the downsampled surface clouds (+ counts, status)
  lfx_host::PinnedBuf h_align;                     // the alignment's small copies to and from the host (poses, counts, states)
  lfx_host::PinnedBuf h_loc;                       // lfx_localize_batch: the clouds' lengths as the device saw them, [batch][2]
  bool vox_lds_asked = false;                      // the voxel-grid kernel has been granted its 144 KB of dynamic LDS on this device
  int align_guess = 4;                             // iterations the previous alignment needed: so many are queued before the host looks
  uint32_t loc_guess[2] = {0u, 0u};                // longest edge / downsampled surface cloud of the previous lfx_localize_batch

  hipStream_t stream = nullptr;          // used by the synchronous host entry points
  std::vector<uint32_t> h_scan_begin;    // of the last batch
  std::vector<uint32_t> h_scan_geom;     // [batch][kGeomStride]: columns per ring and block boundaries of every scan (organised-scan kernel)
  std::vector<uint32_t> uploaded_begin;  // what scan_begin on the device currently holds
  uint32_t last_batch = 0;
  const void * last_points = nullptr;
  std::vector<lfx_host::HostScan> host;
  uint32_t outputs = LFX_OUT_ALL;        // lfx_config::outputs
  lfx_host::PinnedBuf h_in, h_out;                 // staging of the synchronous host API (allocated on first use)
  // lfx_extract_submit / lfx_extract_wait: two scans in flight, each with its own device input, pinned staging and result
  // block; the uploads run on copy_stream beside the kernels of the scan before
  struct Slot
  {
    lfx_host::DevBuf<uint8_t> in;
    lfx_host::PinnedBuf hin, hout;
    hipEvent_t uploaded = nullptr, done = nullptr;
    uint64_t ticket = 0;
    bool busy = false;
    void * plan = nullptr;               // FetchPlan (lfx_api.hip)
    lfx_host::HostScan host;
  };
  Slot slots[2];
  hipStream_t copy_stream = nullptr;
  uint64_t next_ticket = 1, next_wait = 1;
  uint32_t * h_status = nullptr;         // pinned, lfx_batch_status

  bool profiling = false;
  uint32_t profile_every = 1, batch_no = 0;   // lfx_set_profiling_interval: events around every n-th batch only
  bool profile_now = false;
  struct Span { hipEvent_t a, b; int k; };
  std::vector<Span> spans;
  std::vector<hipEvent_t> free_events;
  double ms[LFX_N_KERNELS] = {};
  uint64_t launches[LFX_N_KERNELS] = {};
};

namespace lfx_host
{

std::string & create_error();          // what lfx_last_error(NULL) reports (lfx_api.hip)

#define LFX_HIP(ctx, call) \
  do { \
    const hipError_t e_ = (call); \
    if (e_ != hipSuccess) { \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_); \
      return LFX_ERR_HIP; \
    } \
  } while (0)

inline int fail(lfx_ctx * ctx, int code, const std::string & msg)
{
  if (ctx) {ctx->err = msg;}
  return code;
}

// lfx_downsample.hip: lfx_voxel_downsample with the extras of lfx_localize_batch
int voxel_downsample(
  lfx_ctx * c, const float * d_points, const uint32_t * d_begin, const uint32_t * d_count, uint32_t count_stride,
  uint32_t n_clouds, size_t total_points, float leaf, float * d_out, uint32_t * d_out_count, uint32_t * d_status, void * stream,
  bool unfiltered, const uint32_t * d_other_count, uint32_t * lengths);

}  // namespace lfx_host
