// lfx_api.hip -- host side of the C ABI declared in include/lfx.h.
//
// Owns device scratch, turns the nine node parameters into kernel constants, launches the five
// kernels of lfx_kernels.hpp on the caller's stream and moves results.  There is no CPU
// implementation of the path here: without a gfx950 device every entry point fails.
#include "../../include/lfx.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>      // types and prototypes only: librccl is opened at run time (lfx_comm_*), not linked
#include <dlfcn.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "lfx_kernels.hpp"

namespace
{

std::string g_create_error;

const char * kKernelNames[LFX_N_KERNELS] = {
  "ring_histogram_kernel", "ring_scan_kernel", "ring_scatter_kernel", "ring_unit_kernel",
  "ring_order_kernel", "ring_unit_kernel(second pass)", "ring_extract_kernel", "ring_totals_kernel",
  "feature_compact_kernel", "ring_unit_org_kernel", "ring_cut_kernel", "ring_stream_kernel"};

// IsNeighborXY compares acos(cos_angle) with the threshold (neighbor.hpp:44-48, math.cpp:45).
// acos is monotone, so that test is a bound on cos_angle itself: the smallest double c with
// acos(c) < threshold, found by bisection over the ordered doubles with the HOST's acos -- the
// same libm call the reference node makes on this machine.  (+inf: never a neighbour.)
int64_t ordered_key(double d)
{
  int64_t k;
  std::memcpy(&k, &d, 8);
  return k < 0 ? std::numeric_limits<int64_t>::min() - k : k;
}

double from_ordered_key(int64_t k)
{
  const int64_t b = k < 0 ? std::numeric_limits<int64_t>::min() - k : k;
  double d;
  std::memcpy(&d, &b, 8);
  return d;
}

double cos_bound(double radian_threshold)
{
  if (!(std::acos(1.0) < radian_threshold)) {return std::numeric_limits<double>::infinity();}
  if (std::acos(-1.0) < radian_threshold) {return -1.0;}
  int64_t lo = ordered_key(-1.0), hi = ordered_key(1.0);   // predicate false at lo, true at hi
  while (hi - lo > 1) {
    const int64_t mid = lo + (hi - lo) / 2;
    if (std::acos(from_ordered_key(mid)) < radian_threshold) {hi = mid;} else {lo = mid;}
  }
  return from_ordered_key(hi);
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
    if (p) {(void)hipHostFree(p);}
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

}  // namespace

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
  bool default_thresholds = false;       // padding 5 and the seven thresholds of hyper_parameter.hpp:35-43: literal-threshold unit kernel
  uint32_t unit_chunks = 6;              // chunks of 64 positions per unit wave (3..6), from the configured ring length
  uint32_t unit_flags = 65u;             // LFX_DEBUG_UNIT_FLAGS: 1 edge pass, 64 surface pass (ablations only)
  uint32_t drop_zero = 0;                // lfx_config::drop_zero_points
  bool single_pass = true;               // look-back bucketing; LFX_DEBUG_TWO_PASS selects histogram + scan + scatter
  // Order repair BEFORE the unit kernel (ring_order_kernel over every ring), switched on while the stream keeps
  // arriving rotated / reversed: decided from the counters of earlier batches, which arrive in pinned host
  // memory without anyone waiting for them.  LFX_DEBUG_PRE_ORDER=0/1 pins it.
  int pre_order_env = -1;
  bool pre_order = false;
  uint32_t * h_counters = nullptr;       // pinned [lfx::kCounters]: deferred, repaired after the first pass, slow, repaired before it,
                                         // scans on the fall-back list, organised-scan kernel ran, scans of that batch
  // The organised-scan kernel (lfx_kernels.hpp, unit_body<ORG>) reads a driver's column-major scan directly; scans that are
  // not of that form fall back to the bucketing route inside the same call.  While most scans of a stream fall back the
  // kernel is not launched at all (decided from the counters of earlier batches; every 16th batch tries again).
  // LFX_DEBUG_FUSED=0/1 pins it.
  bool fused_possible = false;
  int fused_env = -1;
  bool walk_rings = false;                // the organised-scan kernel in its streaming form (ring_stream_kernel); LFX_DEBUG_STREAM=0: one wave per unit
  // Rings that arrive rotated / reversed (a driver that does not cut its scans at -pi, a clockwise sensor): while the
  // organised-scan kernel keeps giving scans up for their angle order alone, ring_cut_kernel finds every ring's
  // transform first and the kernel applies it in its loads (LFX_DEBUG_XFORM=0/1 pins it).
  int xform_env = -1;
  bool use_xform = false;
  bool last_used_xform = false;          // the last batch's organised-scan kernel ran with the transforms
  int short_tail_env = -1;               // LFX_DEBUG_SHORT_TAIL=0/1 pins the two-launch tail of the bucketing route (tests)
  bool bucket_all = false;               // the stream is not organised: bucketing route for every scan
  uint32_t retry_in = 0;
  uint32_t redo_cap_env = 0;             // LFX_DEBUG_REDO_CAP: rings the second unit pass is launched for (tests)
  uint32_t h_rings_seen = 0;             // rings of the batch those counters belong to
  bool fast_path = true;                 // wave-per-unit kernel first, workgroup-per-ring kernel for what it defers
  std::string err;

  // device scratch
  DevBuf<uint32_t> scan_begin, scan_info, chunk_base, chunk_flags, ring_count, ring_nedge,
    ring_nsurf, ring_ebase, ring_sbase, ring_flags, unit_ne, unit_ns, unit_span, slow_list, defer_list, redo_list, fb_list, xform,
    sidx, rec_idx, edge_idx,
    surf_idx, d_sidx;
  DevBuf<uint16_t> chunk_hist;
  DevBuf<uint8_t> ring_status, label_s, staging, d_label;
  DevBuf<double> d_curv;
  DevBuf<float2> sxy;
  DevBuf<float> sz;
  DevBuf<double> curv_s;
  DevBuf<float4> edge_pts, surf_pts, rec_pts;
  DevBuf<lfx::UnitTables> unit_tab;      // the unit kernel's output pointers (one element)
  DevBuf<uint32_t> vox_scratch;          // lfx_voxel_downsample: sort keys / values, allocated on first use
  DevBuf<double> align_scratch;          // lfx_scan_to_map_align: states, rows, errors; allocated on first use
  DevBuf<float> align_surface;           // lfx_localize_batch: the downsampled surface clouds (+ counts, status)
  PinnedBuf h_align;                     // the alignment's small copies to and from the host (poses, counts, states)

  hipStream_t stream = nullptr;          // used by the synchronous host entry points
  std::vector<uint32_t> h_scan_begin;    // of the last batch
  std::vector<uint32_t> uploaded_begin;  // what scan_begin on the device currently holds
  uint32_t last_batch = 0;
  const void * last_points = nullptr;
  std::vector<HostScan> host;
  uint32_t outputs = LFX_OUT_ALL;        // lfx_config::outputs
  PinnedBuf h_in, h_out;                 // staging of the synchronous host API (allocated on first use)
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

namespace
{

#define LFX_HIP(ctx, call) \
  do { \
    const hipError_t e_ = (call); \
    if (e_ != hipSuccess) { \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_); \
      return LFX_ERR_HIP; \
    } \
  } while (0)

int fail(lfx_ctx * ctx, int code, const std::string & msg)
{
  if (ctx) {ctx->err = msg;}
  return code;
}

hipEvent_t take_event(lfx_ctx * c)
{
  if (!c->free_events.empty()) {
    hipEvent_t e = c->free_events.back();
    c->free_events.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}

struct Timed
{
  Timed(lfx_ctx * c, int k, hipStream_t s)
  : c_(c), k_(k), s_(s)
  {
    if (c_->profile_now) {
      a_ = take_event(c_);
      b_ = take_event(c_);
      (void)hipEventRecord(a_, s_);
    }
  }
  ~Timed()
  {
    if (c_->profile_now) {
      (void)hipEventRecord(b_, s_);
      c_->spans.push_back({a_, b_, k_});
    }
  }
  lfx_ctx * c_;
  int k_;
  hipStream_t s_;
  hipEvent_t a_ = nullptr, b_ = nullptr;
};

int drain_spans(lfx_ctx * c)
{
  for (auto & sp : c->spans) {
    LFX_HIP(c, hipEventSynchronize(sp.b));
    float t = 0.f;
    LFX_HIP(c, hipEventElapsedTime(&t, sp.a, sp.b));
    c->ms[sp.k] += t;
    c->launches[sp.k] += 1;
    c->free_events.push_back(sp.a);
    c->free_events.push_back(sp.b);
  }
  c->spans.clear();
  return LFX_OK;
}

int validate_params(const lfx_params * p, std::string & why)
{
  // hyper_parameter.hpp:45-53 asserts every parameter > 0
  if (!p) {why = "params is NULL"; return LFX_ERR_INVALID_ARGUMENT;}
  if (p->padding <= 0 || p->padding > LFX_MAX_PADDING) {
    why = "convolution_padding must be in [1, " + std::to_string(LFX_MAX_PADDING) + "]";
    return LFX_ERR_INVALID_ARGUMENT;
  }
  if (p->n_blocks <= 0) {why = "n_blocks must be > 0"; return LFX_ERR_INVALID_ARGUMENT;}
  if (!(p->neighbor_degree_threshold > 0) || !(p->distance_diff_threshold > 0) ||
    !(p->parallel_beam_min_range_ratio > 0) || !(p->edge_threshold > 0) || !(p->surface_threshold > 0) ||
    !(p->min_range > 0) || !(p->max_range > 0))
  {
    why = "every threshold / range parameter must be > 0 (hyper_parameter.hpp:45-53)";
    return LFX_ERR_INVALID_ARGUMENT;
  }
  return LFX_OK;
}

lfx::Params device_params(const lfx_params & p)
{
  lfx::Params d;
  d.P = p.padding;
  d.B = p.n_blocks;
  d.cos_bound = cos_bound(p.neighbor_degree_threshold * M_PI / 180.0);   // degree_to_radian.hpp:34-37
  d.cos_bound_f = (float)d.cos_bound;
  d.dist_diff = p.distance_diff_threshold;
  d.pb_ratio = p.parallel_beam_min_range_ratio;
  d.pb_ratio_f = (float)d.pb_ratio;
  d.edge_thr = p.edge_threshold;
  d.surf_thr = p.surface_threshold;
  d.min_range = p.min_range;
  d.max_range = p.max_range;
  return d;
}

uint32_t ring_threads_for(uint32_t cap)
{
  return cap > 1024 ? 512u : 256u;
}

// Launch the five kernels for `batch` scans whose records lie back to back at d_points.
int run_batch(lfx_ctx * c, const void * d_points, const uint32_t * n_points, uint32_t batch, hipStream_t st)
{
  if (!d_points || !n_points || batch == 0) {return fail(c, LFX_ERR_INVALID_ARGUMENT, "empty batch");}
  if (batch > c->max_batch) {return fail(c, LFX_ERR_CAPACITY, "batch exceeds max_batch");}
  c->h_scan_begin.resize(batch + 1);
  uint32_t longest = 0;
  size_t total = 0;
  for (uint32_t s = 0; s < batch; s++) {
    if (n_points[s] > c->max_points) {return fail(c, LFX_ERR_CAPACITY, "scan exceeds max_points_per_scan");}
    c->h_scan_begin[s] = (uint32_t)total;
    total += n_points[s];
    longest = n_points[s] > longest ? n_points[s] : longest;
  }
  c->h_scan_begin[batch] = (uint32_t)total;
  if (total > c->total_cap) {return fail(c, LFX_ERR_CAPACITY, "batch exceeds the context's point capacity");}
  LFX_HIP(c, hipSetDevice(c->device));
  if (c->uploaded_begin != c->h_scan_begin) {
    // pageable source: the runtime stages it before returning, so the vector may change afterwards
    LFX_HIP(c, hipMemcpyAsync(c->scan_begin.p, c->h_scan_begin.data(), (batch + 1) * 4, hipMemcpyHostToDevice, st));
    c->uploaded_begin = c->h_scan_begin;
  }
  c->last_batch = batch;
  c->last_points = d_points;
  c->profile_now = c->profiling && (c->batch_no++ % c->profile_every) == 0u;
  uint32_t * counters = c->ring_flags.p + (size_t)c->max_batch * lfx::kRings;     // behind ring_flags[max_batch][256]
  uint32_t * defer_count = counters + lfx::kCntDefer, * redo_count = counters + lfx::kCntRedo,
    * slow_count = counters + lfx::kCntSlow, * fb_count = counters + lfx::kCntFallback;
  const uint8_t * pts = static_cast<const uint8_t *>(d_points);
  const uint32_t chunks = (longest + lfx::kChunkPoints - 1) / lfx::kChunkPoints;
  const bool canon = c->layout.step == 32 && c->layout.ox == 0 && c->layout.oy == 4 && c->layout.oz == 8 &&
    c->layout.oring == 20 && c->layout.rtype == LFX_FIELD_UINT16 && c->layout.be == 0 &&
    (reinterpret_cast<uintptr_t>(pts) & 15u) == 0;
  // ---- which route: the organised-scan kernel first (scans it cannot take fall back inside this call), or
  //      bucketing for every scan.  What earlier batches reported arrives in pinned memory unasked.
  bool fused = c->fused_possible && canon && chunks != 0;
  uint32_t fb_grid = batch;                            // list entries the bucketing kernels are launched for
  bool short_tail = false;                             // bucketing route = bucketing + the workgroup-per-ring kernel only
  if (fused) {
    const uint32_t was_fused = c->h_counters[lfx::kCntFusedRan], fell = c->h_counters[lfx::kCntFallback],
      of = c->h_counters[lfx::kCntBatch];
    const uint32_t order_fell = c->h_counters[lfx::kCntOrderFell], cut_ran = c->h_counters[lfx::kCntCutRan],
      turned = c->h_counters[lfx::kCntTurned];
    if (was_fused && of) {
      // most of what fell back did so for the angle order of its rings alone: find the rings' transforms first from now
      // on; back to plain loads once (almost) no ring needs one any more
      if (!cut_ran && 4u * order_fell > of && 2u * order_fell > fell) {c->use_xform = true;}
      if (cut_ran && 50u * turned < of * c->max_rings) {c->use_xform = false;}
    }
    if (c->xform_env >= 0) {c->use_xform = c->xform_env != 0;}
    if (c->fused_env >= 0) {
      fused = c->fused_env != 0;
    } else {
      if (was_fused && of) {
        // (a report from before the transforms were switched on says nothing about the route with them)
        const bool mostly_not = 4u * fell > of && !(c->use_xform && !cut_ran);
        if (mostly_not && !c->bucket_all) {c->retry_in = 16;}
        c->bucket_all = mostly_not;
      }
      if (c->bucket_all) {
        fused = false;
        if (c->retry_in == 0 || --c->retry_in == 0) {fused = true; c->retry_in = 16;}     // the stream may have changed
      }
    }
    if (fused) {
      const uint32_t guess = (was_fused ? 2u * fell : 0u) + 8u;
      fb_grid = guess < batch && !c->bucket_all ? guess : batch;      // (a retry on a stream that has been falling back: expect all of it)
      // a stream that has not been falling back: its odd scan out (if one turns up) is redone by the workgroup-per-ring
      // kernel straight from the bucketed arrays -- two near-empty launches per batch instead of five
      short_tail = was_fused && of && fell == 0 && c->pre_order_env < 0 && c->redo_cap_env == 0;
      if (c->short_tail_env >= 0) {short_tail = c->short_tail_env != 0;}
    }
  }
  hipLaunchKernelGGL(lfx::batch_reset_kernel, dim3(64), dim3(256), 0, st,
    c->scan_info.p, batch * 4u, c->ring_count.p, batch * (uint32_t)lfx::kRings, c->chunk_flags.p,
    c->single_pass ? batch * c->max_chunks : 0u, c->ring_flags.p, batch * (uint32_t)lfx::kRings, counters, c->fb_list.p, batch,
    fused ? 0u : 1u, c->xform.p);
  if (chunks == 0) {return LFX_OK;}
  c->last_used_xform = fused && c->use_xform;
  if (fused) {
    const bool xf = c->use_xform;
    if (xf) {
      Timed t(c, 10, st);
      hipLaunchKernelGGL(lfx::ring_cut_kernel, dim3(batch), dim3(lfx::kCutThreads), 0, st,
        pts, c->scan_begin.p, c->max_rings, c->cap, c->xform.p, counters);
    }
    Timed t(c, c->walk_rings ? 11 : 9, st);
    const uint32_t groups = (c->max_rings + 3u) / 4u;
    void (*kern)(lfx::Params, uint32_t, uint32_t, uint32_t, uint32_t, const uint8_t *, const uint32_t *, uint32_t *,
      const lfx::UnitTables *, const uint32_t *) = nullptr;
#define LFX_PICK_ORG(DEFV, XFV) \
    (c->unit_chunks == 5 ? &lfx::ring_unit_org_kernel<5, DEFV, XFV> : c->unit_chunks == 4 ? &lfx::ring_unit_org_kernel<4, DEFV, XFV> : \
     c->unit_chunks == 3 ? &lfx::ring_unit_org_kernel<3, DEFV, XFV> : &lfx::ring_unit_org_kernel<6, DEFV, XFV>)
    if (c->default_thresholds) {
      kern = xf ? LFX_PICK_ORG(true, true) : LFX_PICK_ORG(true, false);
    } else {
      kern = xf ? LFX_PICK_ORG(false, true) : LFX_PICK_ORG(false, false);
    }
#undef LFX_PICK_ORG
#define LFX_PICK_STREAM(DEFV, XFV) \
    (c->unit_chunks == 5 ? &lfx::ring_stream_kernel<5, DEFV, XFV> : c->unit_chunks == 4 ? &lfx::ring_stream_kernel<4, DEFV, XFV> : \
     c->unit_chunks == 3 ? &lfx::ring_stream_kernel<3, DEFV, XFV> : &lfx::ring_stream_kernel<6, DEFV, XFV>)
    if (c->walk_rings) {
      // one workgroup per (ring group, scan): the waves walk their rings block by block
      if (c->default_thresholds) {
        kern = xf ? LFX_PICK_STREAM(true, true) : LFX_PICK_STREAM(true, false);
      } else {
        kern = xf ? LFX_PICK_STREAM(false, true) : LFX_PICK_STREAM(false, false);
      }
    }
#undef LFX_PICK_STREAM
    hipLaunchKernelGGL(kern, dim3(c->walk_rings ? groups : groups * (uint32_t)c->dev.B, batch), dim3(64 * lfx::kUnitWaves), c->unit_lds_pad, st,
      c->dev, c->cap, c->unit_flags, c->max_rings, c->drop_zero, pts, c->scan_begin.p, c->ring_count.p, c->unit_tab.p, c->xform.p);
  }
  // ---- the bucketing route, over the scans on the fall-back list
  if (c->single_pass) {
    Timed t(c, 2, st);
    auto kern = &lfx::ring_scatter_kernel<false, true>;
    if (canon) {kern = &lfx::ring_scatter_kernel<true, true>;}
    if (fb_grid == batch) {                            // a row per scan: the form without the loop over list entries
      kern = canon ? &lfx::ring_scatter_kernel<true, true, true> : &lfx::ring_scatter_kernel<false, true, true>;
    }
    hipLaunchKernelGGL(kern, dim3(chunks, fb_grid), dim3(lfx::kChunkThreads), 0, st,
      pts, c->layout, c->scan_begin.p, c->chunk_base.p, c->chunk_flags.p, c->ring_count.p, c->scan_info.p,
      c->sxy.p, c->sz.p, c->sidx.p, c->max_chunks, c->max_rings, c->cap, c->drop_zero, fb_count, c->fb_list.p);
  } else {
    // (two-pass bucketing, LFX_DEBUG_TWO_PASS: never together with the organised-scan kernel, so the list is every scan in order)
    {
      Timed t(c, 0, st);
      hipLaunchKernelGGL(lfx::ring_histogram_kernel, dim3(chunks, batch), dim3(lfx::kChunkThreads), 0, st,
        pts, c->layout, c->scan_begin.p, c->chunk_hist.p, c->scan_info.p, c->max_chunks, c->max_rings, c->drop_zero);
    }
    {
      Timed t(c, 1, st);
      hipLaunchKernelGGL(lfx::ring_scan_kernel, dim3(batch), dim3(lfx::kRings), 0, st,
        c->scan_begin.p, c->chunk_hist.p, c->chunk_base.p, c->ring_count.p, c->scan_info.p, c->max_chunks);
    }
    {
      Timed t(c, 2, st);
      auto kern = &lfx::ring_scatter_kernel<false, false>;
      if (canon) {kern = &lfx::ring_scatter_kernel<true, false>;}
      hipLaunchKernelGGL(kern, dim3(chunks, batch), dim3(lfx::kChunkThreads), 0, st,
        pts, c->layout, c->scan_begin.p, c->chunk_base.p, c->chunk_flags.p, c->ring_count.p, c->scan_info.p,
        c->sxy.p, c->sz.p, c->sidx.p, c->max_chunks, c->max_rings, c->cap, c->drop_zero, fb_count, c->fb_list.p);
    }
  }
  // the near-empty launches of the bucketing route are kept small while the organised-scan kernel takes the stream
  const uint32_t list_grid = fused ? (c->slow_grid < 4u * fb_grid ? c->slow_grid : 4u * fb_grid) : c->slow_grid;
  if (c->fast_path && !short_tail) {
    if (c->pre_order_env >= 0) {
      c->pre_order = c->pre_order_env != 0;
    } else if (c->h_counters && c->h_rings_seen) {
      // more than a twentieth of the rings of an earlier batch needed their order repaired: expect the same now
      c->pre_order = 20u * (c->h_counters[lfx::kCntRedo] + c->h_counters[lfx::kCntPreFixed]) > c->h_rings_seen;
    }
    if (c->pre_order) {
      Timed t(c, 4, st);
      hipLaunchKernelGGL(lfx::ring_order_kernel, dim3(fused ? list_grid : 4 * c->slow_grid), dim3(512), c->order_lds, st,
        c->cap, c->max_rings, c->ring_count.p, c->sxy.p, c->sz.p, c->sidx.p, c->ring_flags.p, defer_count,
        c->defer_list.p, redo_count, c->redo_list.p, slow_count, c->slow_list.p, 1u, counters + lfx::kCntPreFixed, 0u,
        fb_count, c->fb_list.p, 0u);
    }
    {
      Timed t(c, 3, st);
      const uint32_t units = c->max_rings * (uint32_t)c->dev.B;
      // the looping form only where the list's length is a guess (behind the organised-scan kernel)
#define LFX_PICK_UNIT(DEFV, LOOPV) \
      (c->unit_chunks == 5 ? &lfx::ring_unit_kernel<false, 5, DEFV, LOOPV> : c->unit_chunks == 4 ? &lfx::ring_unit_kernel<false, 4, DEFV, LOOPV> : \
       c->unit_chunks == 3 ? &lfx::ring_unit_kernel<false, 3, DEFV, LOOPV> : &lfx::ring_unit_kernel<false, 6, DEFV, LOOPV>)
      auto kern = c->default_thresholds ? (fused ? LFX_PICK_UNIT(true, true) : LFX_PICK_UNIT(true, false)) :
        (fused ? LFX_PICK_UNIT(false, true) : LFX_PICK_UNIT(false, false));
#undef LFX_PICK_UNIT
      hipLaunchKernelGGL(kern, dim3((units + lfx::kUnitWaves - 1) / lfx::kUnitWaves, fb_grid),
        dim3(64 * lfx::kUnitWaves), c->unit_lds_pad, st,
        c->dev, c->cap, c->unit_flags, c->max_rings, c->ring_count.p, c->sxy.p, c->sz.p, c->sidx.p, c->unit_tab.p,
        defer_count, c->defer_list.p, fb_count, c->fb_list.p, 0u);
    }
    // The second pass is launched for as many rings as earlier batches had repaired after their first pass, twice
    // over and at least 256 (a launch that covers every unit of a large batch costs ~20 us to find nothing to do);
    // the order kernel hands what does not fit to the workgroup-per-ring kernel.
    uint32_t redo_cap = batch * c->max_rings;
    if (c->redo_cap_env) {
      redo_cap = c->redo_cap_env;
    } else if (c->h_counters && c->h_rings_seen) {
      const uint32_t want = 2u * c->h_counters[lfx::kCntRedo] + 256u;
      redo_cap = want < redo_cap ? want : redo_cap;
    }
    {
      // rings out of angle order: repaired in place, then a second pass of the unit kernel over them
      Timed t(c, 4, st);
      hipLaunchKernelGGL(lfx::ring_order_kernel, dim3(list_grid), dim3(512), c->order_lds, st,
        c->cap, c->max_rings, c->ring_count.p, c->sxy.p, c->sz.p, c->sidx.p, c->ring_flags.p, defer_count,
        c->defer_list.p, redo_count, c->redo_list.p, slow_count, c->slow_list.p, 0u, counters + lfx::kCntPreFixed, redo_cap,
        fb_count, c->fb_list.p, 0xFFFFFFFFu /* the first unit pass covers the whole list (it loops where it has to) */);
    }
    {
      Timed t(c, 5, st);
      const uint32_t units = redo_cap * (uint32_t)c->dev.B;
      auto kern = &lfx::ring_unit_kernel<true, 6, false>;
      if (c->unit_chunks == 5) {kern = &lfx::ring_unit_kernel<true, 5, false>;}
      if (c->unit_chunks == 4) {kern = &lfx::ring_unit_kernel<true, 4, false>;}
      if (c->unit_chunks == 3) {kern = &lfx::ring_unit_kernel<true, 3, false>;}
      hipLaunchKernelGGL(kern, dim3((units + lfx::kUnitWaves - 1) / lfx::kUnitWaves),
        dim3(64 * lfx::kUnitWaves), c->unit_lds_pad, st,
        c->dev, c->cap, c->unit_flags, c->max_rings, c->ring_count.p, c->sxy.p, c->sz.p, c->sidx.p, c->unit_tab.p,
        slow_count, c->slow_list.p, redo_count, c->redo_list.p, redo_cap);
    }
  }
  {
    Timed t(c, 6, st);
    const dim3 grid = c->fast_path ? dim3(list_grid) : dim3(c->max_rings, batch);
    hipLaunchKernelGGL(lfx::ring_extract_kernel, grid, dim3(c->ring_threads), c->ring_lds, st,
      c->dev, c->cap, c->stage_flags, short_tail ? 2u : (c->fast_path ? 1u : 0u), pts, c->layout, c->scan_begin.p,
      c->ring_count.p, c->sxy.p, c->sz.p, c->sidx.p, c->label_s.p, c->curv_s.p, c->rec_pts.p,
      c->rec_idx.p, c->ring_status.p, c->unit_ne.p, c->unit_ns.p, c->unit_span.p, c->ring_flags.p,
      short_tail ? fb_count : slow_count, short_tail ? c->fb_list.p : c->slow_list.p, c->max_rings);
  }
  {
    Timed t(c, 7, st);
    hipLaunchKernelGGL(lfx::ring_totals_kernel, dim3(batch), dim3(lfx::kRings), 0, st,
      c->scan_info.p, c->ring_count.p, c->unit_ne.p, c->unit_ns.p, c->ring_nedge.p, c->ring_nsurf.p,
      c->ring_ebase.p, c->ring_sbase.p, c->fast_path ? (uint32_t)c->dev.B : 1u, c->max_rings);
  }
  {
    Timed t(c, 8, st);
    const uint32_t n_units = c->fast_path ? (uint32_t)c->dev.B : 1u;
    hipLaunchKernelGGL(lfx::feature_compact_kernel, dim3((c->max_rings + 3) / 4, batch), dim3(256), 0, st,
      n_units, c->cap, c->scan_begin.p, c->ring_count.p, c->ring_ebase.p, c->ring_sbase.p, c->unit_ne.p,
      c->unit_ns.p, c->unit_span.p, c->rec_pts.p, c->rec_idx.p, c->edge_pts.p, c->edge_idx.p, c->surf_pts.p,
      c->surf_idx.p, c->max_rings);
  }
  if (c->h_counters) {
    // for the next batches' decision about the order pre-pass; nobody waits for this copy
    LFX_HIP(c, hipMemcpyAsync(c->h_counters, counters, 4 * lfx::kCounters, hipMemcpyDeviceToHost, st));
    c->h_rings_seen = batch * c->max_rings;
  }
  LFX_HIP(c, hipGetLastError());
  return LFX_OK;
}

size_t align16(size_t v) {return (v + 15u) & ~(size_t)15u;}

// Results of scans [first, first + count) of the last batch to pinned host memory: everything is queued on `st`
// (the pack kernel writes headers and clouds straight into the pinned block; labels / curvature / sorted_index, when
// asked for, are un-permuted on the device and copied) and the host waits ONCE.
int fetch(lfx_ctx * c, uint32_t first, uint32_t count, hipStream_t st, uint32_t mask, lfx_scan_result * out)
{
  if (count == 0 || first + count > c->last_batch) {return fail(c, LFX_ERR_INVALID_ARGUMENT, "scan index outside the last batch");}
  if (c->host.size() < c->last_batch) {c->host.resize(c->last_batch);}
  const uint32_t p0 = c->h_scan_begin[first];
  const size_t P = c->h_scan_begin[first + count] - p0;
  const bool want_lab = mask & LFX_OUT_LABELS, want_curv = mask & LFX_OUT_CURVATURE, want_sidx = mask & LFX_OUT_SORTED_INDEX;
  // pinned block: headers | edge_pts | surf_pts | curvature | edge_idx | surf_idx | sorted_index | labels
  const size_t o_hdr = 0, o_ep = align16((size_t)count * lfx::kResultHeaderBytes), o_sp = o_ep + P * 16, o_cv = o_sp + P * 16,
    o_ei = o_cv + (want_curv ? P * 8 : 0), o_si = o_ei + P * 4, o_sx = o_si + P * 4, o_lb = o_sx + (want_sidx ? P * 4 : 0),
    total = o_lb + (want_lab ? P : 0) + 16;
  if (c->h_out.reserve(total) != hipSuccess) {return fail(c, LFX_ERR_OUT_OF_MEMORY, "cannot allocate the pinned result block");}
  uint8_t * H = c->h_out.p;
  hipLaunchKernelGGL(lfx::result_pack_kernel, dim3(8, count), dim3(256), 0, st,
    first, p0, c->scan_begin.p, c->scan_info.p, c->ring_count.p, c->ring_status.p, c->edge_pts.p, c->edge_idx.p,
    c->surf_pts.p, c->surf_idx.p, H + o_hdr, reinterpret_cast<float4 *>(H + o_ep), reinterpret_cast<float4 *>(H + o_sp),
    reinterpret_cast<uint32_t *>(H + o_ei), reinterpret_cast<uint32_t *>(H + o_si));
  LFX_HIP(c, hipGetLastError());
  if ((want_lab || want_curv || want_sidx) && P) {
    // ring-major (fixed capacity per ring) -> the caller's point order (labels, curvature) and the dense list of
    // angle-sorted indices, rings ascending; points that are in no ring (zero filter, over-long ring) stay Default / 0.
    // All scans of the range in one launch and one copy per array (the buffers grow to the largest range asked for).
    if (c->d_label.n < P) {
      LFX_HIP(c, hipStreamSynchronize(st));
      c->d_label.release(); c->d_curv.release(); c->d_sidx.release();
      if (c->d_label.alloc(P) != hipSuccess || c->d_curv.alloc(P) != hipSuccess || c->d_sidx.alloc(P) != hipSuccess) {
        c->d_label.n = 0;
        return fail(c, LFX_ERR_OUT_OF_MEMORY, "cannot allocate the per-point output buffers");
      }
    }
    LFX_HIP(c, hipMemsetAsync(c->d_label.p, 0, P, st));
    LFX_HIP(c, hipMemsetAsync(c->d_curv.p, 0, P * 8, st));
    hipLaunchKernelGGL(lfx::densify_kernel, dim3(c->max_rings, count), dim3(256), 0, st,
      first, p0, c->scan_begin.p, c->max_rings, c->cap, c->ring_count.p, c->label_s.p, c->curv_s.p, c->sidx.p, c->d_label.p,
      c->d_curv.p, c->d_sidx.p, c->scan_info.p, c->xform.p);
    LFX_HIP(c, hipGetLastError());
    if (want_lab) {LFX_HIP(c, hipMemcpyAsync(H + o_lb, c->d_label.p, P, hipMemcpyDeviceToHost, st));}
    if (want_curv) {LFX_HIP(c, hipMemcpyAsync(H + o_cv, c->d_curv.p, P * 8, hipMemcpyDeviceToHost, st));}
    if (want_sidx) {LFX_HIP(c, hipMemcpyAsync(H + o_sx, c->d_sidx.p, P * 4, hipMemcpyDeviceToHost, st));}
  }
  LFX_HIP(c, hipStreamSynchronize(st));
  for (uint32_t k = 0; k < count; k++) {
    const uint32_t s = first + k, b = c->h_scan_begin[s] - p0, n = c->h_scan_begin[s + 1] - c->h_scan_begin[s];
    const uint32_t * hdr = reinterpret_cast<const uint32_t *>(H + o_hdr + (size_t)k * lfx::kResultHeaderBytes);
    const uint32_t * rcount = hdr + 4;
    const uint8_t * rstat = reinterpret_cast<const uint8_t *>(hdr + 4 + lfx::kRings);
    if (hdr[lfx::kInfoError] & lfx::kErrRingId) {
      return fail(c, LFX_ERR_RING_ID, "a point carries a ring id the context was not created for (max_rings / LFX_MAX_RING_ID)");
    }
    if (hdr[lfx::kInfoError] & lfx::kErrTimeout) {
      return fail(c, LFX_ERR_HIP, "ring bucketing timed out waiting for an earlier chunk (set LFX_DEBUG_TWO_PASS=1)");
    }
    HostScan & h = c->host[s];
    h.ring_id.clear(); h.ring_count.clear(); h.ring_offset.clear(); h.ring_status.clear();
    uint32_t dense = 0, nr = 0;
    for (uint32_t r = 0; r < c->max_rings; r++) {
      if (rcount[r] == 0) {continue;}
      h.ring_id.push_back((uint16_t)r);
      h.ring_count.push_back(rcount[r]);
      h.ring_offset.push_back(dense);
      h.ring_status.push_back(rstat[r]);
      dense += rcount[r];
      nr++;
    }
    if (dense > n || (dense != n && !c->drop_zero)) {
      return fail(c, LFX_ERR_HIP, "internal: ring counts do not add up to the scan");
    }
    lfx_scan_result & o = out[k];
    o.n_points = n;
    o.n_sorted = dense;             // = n less the points the zero filter dropped
    o.labels = want_lab ? H + o_lb + b : nullptr;
    o.curvature = want_curv ? reinterpret_cast<const double *>(H + o_cv) + b : nullptr;
    o.sorted_index = want_sidx ? reinterpret_cast<const uint32_t *>(H + o_sx) + b : nullptr;
    o.n_rings = nr;
    o.ring_id = h.ring_id.data();
    o.ring_count = h.ring_count.data();
    o.ring_offset = h.ring_offset.data();
    o.ring_status = h.ring_status.data();
    o.n_edge = hdr[lfx::kInfoEdge];
    o.edge_points = reinterpret_cast<const float *>(H + o_ep) + (size_t)b * 4;
    o.edge_index = reinterpret_cast<const uint32_t *>(H + o_ei) + b;
    o.n_surface = hdr[lfx::kInfoSurface];
    o.surface_points = reinterpret_cast<const float *>(H + o_sp) + (size_t)b * 4;
    o.surface_index = reinterpret_cast<const uint32_t *>(H + o_si) + b;
  }
  return LFX_OK;
}

}  // namespace

// =========================================================================================
extern "C" {

void lfx_default_params(lfx_params * p)   // hyper_parameter.hpp:35-43
{
  if (!p) {return;}
  *p = lfx_params{5, 2.0, 0.3, 0.02, 0.05, 0.05, 0.1, 100.0, 6};
}

void lfx_launch_params(lfx_params * p)   // lidar_feature_launch/config/lidar_feature_extraction.param.yaml:3-10
{
  if (!p) {return;}
  *p = lfx_params{2, 3.0, 0.3, 0.02, 50.0, 0.05, 0.1, 1000.0, 6};
}

const char * lfx_status_string(int s)
{
  switch (s) {
    case LFX_RING_OK: return "ok";
    case LFX_RING_SPARSE: return "ring has fewer than padding+1 points (removed)";
    case LFX_RING_TOO_FEW_CONV: return "ring has fewer than 2*padding+1 points (convolution)";
    case LFX_RING_TOO_FEW_BLOCKS: return "ring has fewer than n_blocks points between its borders";
    case LFX_RING_BLOCK_TOO_SMALL: return "a block of the ring holds fewer than 2 points";
    case LFX_RING_ZERO_NORM_PAIR: return "two adjacent points are both (0,0) in xy";
    case LFX_RING_TOO_LARGE: return "ring holds more points than max_points_per_ring";
    default: return "unknown";
  }
}

int lfx_range_message(
  int kind, const char * value_name, const char * range_name, long long value, long long range, char * buf, size_t len)
{
  static const char * const op[4] = {">=", "<=", ">", "<"};      // range_message.hpp:37-83
  if (kind < 0 || kind > 3 || !value_name || !range_name || (!buf && len)) {return -1;}
  return std::snprintf(buf, len, "%s (which is %lld) %s %s (which is %lld)", value_name, value, op[kind], range_name, range);
}

int lfx_ring_message(int ring_status, uint32_t n_points, const lfx_params * p, char * buf, size_t len)
{
  if (!p || (!buf && len)) {return -1;}
  const int N = (int)n_points, P = p->padding, B = p->n_blocks;
  switch (ring_status) {
    case LFX_RING_TOO_FEW_CONV:       // convolution.cpp:40-41
      return std::snprintf(buf, len, "Input array size %d cannot be smaller than weight size %d", N, 2 * P + 1);
    case LFX_RING_TOO_FEW_BLOCKS:     // index_range.cpp:36-38 (the reference's text lacks the closing parenthesis)
      return std::snprintf(buf, len, "end_index - start_index (which is %d) cannot be smaller than n_blocks (which is %d", N - 2 * P, B);
    case LFX_RING_BLOCK_TOO_SMALL: {  // neighbor.hpp:72-73: the first block slice with fewer than two points
      for (int j = 0; j < B; j++) {
        const double s = (double)P, e = (double)(N - P), n = (double)B;     // index_range.cpp:60-66
        const int size = (int)(s * (1. - (j + 1) / n) + e * (j + 1) / n) - (int)(s * (1. - j / n) + e * j / n);
        if (size < 2) {return std::snprintf(buf, len, "The input point size (which is %d) cannot be smaller than 2", size);}
      }
      return std::snprintf(buf, len, "%s", "");
    }
    case LFX_RING_ZERO_NORM_PAIR:     // math.cpp:41
      return std::snprintf(buf, len, "All input values are zero. Angle cannot be calculated");
    default:
      return std::snprintf(buf, len, "%s", "");
  }
}

const char * lfx_kernel_name(int k) {return (k >= 0 && k < LFX_N_KERNELS) ? kKernelNames[k] : "";}

const char * lfx_last_error(const lfx_ctx * ctx) {return ctx ? ctx->err.c_str() : g_create_error.c_str();}

namespace
{
uint32_t field_size(uint32_t datatype)
{
  switch (datatype) {
    case LFX_FIELD_INT8: case LFX_FIELD_UINT8: return 1;
    case LFX_FIELD_INT16: case LFX_FIELD_UINT16: return 2;
    case LFX_FIELD_INT32: case LFX_FIELD_UINT32: case LFX_FIELD_FLOAT32: return 4;
    case LFX_FIELD_FLOAT64: return 8;
    default: return 0;
  }
}
}  // namespace

int lfx_layout_from_fields(
  const lfx_point_field * fields, uint32_t n_fields, uint32_t point_step, int is_bigendian, lfx_layout * out)
{
  if (!out || (!fields && n_fields) || point_step == 0) {return LFX_ERR_INVALID_ARGUMENT;}
  const lfx_point_field * fx = nullptr, * fy = nullptr, * fz = nullptr, * fr = nullptr;
  for (uint32_t i = 0; i < n_fields; i++) {
    const lfx_point_field & f = fields[i];
    if (!f.name) {return LFX_ERR_INVALID_ARGUMENT;}
    if (!std::strcmp(f.name, "x")) {fx = &f;}
    if (!std::strcmp(f.name, "y")) {fy = &f;}
    if (!std::strcmp(f.name, "z")) {fz = &f;}
    if (!std::strcmp(f.name, "ring")) {fr = &f;}       // RingIsAvailable, ring.cpp:36-44
  }
  if (!fr) {return LFX_ERR_NO_RING_FIELD;}
  for (const lfx_point_field * f : {fx, fy, fz}) {
    if (!f || f->datatype != LFX_FIELD_FLOAT32 || (uint64_t)f->offset + 4u > point_step) {return LFX_ERR_UNSUPPORTED_FIELD;}
  }
  if (fr->datatype < LFX_FIELD_INT8 || fr->datatype > LFX_FIELD_UINT32 ||
    (uint64_t)fr->offset + field_size(fr->datatype) > point_step)      // (64-bit sums: an offset near 2^32 must not wrap past the test)
  {
    return LFX_ERR_UNSUPPORTED_FIELD;
  }
  *out = lfx_layout{point_step, fx->offset, fy->offset, fz->offset, fr->offset, fr->datatype, is_bigendian ? 1u : 0u};
  return LFX_OK;
}

int lfx_create(lfx_ctx ** out, int device_id, const lfx_params * params, const lfx_config * config)
{
  if (!out) {return LFX_ERR_INVALID_ARGUMENT;}
  *out = nullptr;
  std::string why;
  if (validate_params(params, why) != LFX_OK) {g_create_error = why; return LFX_ERR_INVALID_ARGUMENT;}
  if (!config || config->max_points_per_scan == 0 || config->max_batch == 0) {
    g_create_error = "config must give max_points_per_scan and max_batch";
    return LFX_ERR_INVALID_ARGUMENT;
  }
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device_id < 0 || device_id >= count) {
    g_create_error = "no HIP device: this library has no CPU path";
    return LFX_ERR_NO_DEVICE;
  }
  hipDeviceProp_t prop{};
  if (hipGetDeviceProperties(&prop, device_id) != hipSuccess || std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    g_create_error = std::string("device is not gfx950 (MI355X): ") + prop.gcnArchName;
    return LFX_ERR_NO_DEVICE;
  }
  lfx_ctx * c = new lfx_ctx();
  c->device = device_id;
  c->params = *params;
  c->dev = device_params(*params);
  {
    lfx_params d;
    lfx_default_params(&d);
    c->default_thresholds = params->padding == 5 && params->distance_diff_threshold == d.distance_diff_threshold &&
      params->parallel_beam_min_range_ratio == d.parallel_beam_min_range_ratio && params->edge_threshold == d.edge_threshold &&
      params->surface_threshold == d.surface_threshold && params->min_range == d.min_range && params->max_range == d.max_range &&
      std::getenv("LFX_DEBUG_GENERIC_THRESHOLDS") == nullptr;
  }
  const lfx_layout & L = config->layout;
  if (L.point_step == 0) {
    c->layout = lfx::Layout{32, 0, 4, 8, 20, LFX_FIELD_UINT16, 0};     // PointXYZIR, point_type.hpp:62-86
  } else {
    const uint32_t rtype = L.ring_datatype ? L.ring_datatype : (uint32_t)LFX_FIELD_UINT16;
    const uint32_t rsize = field_size(rtype);
    if (rtype < LFX_FIELD_INT8 || rtype > LFX_FIELD_UINT32 ||
      (uint64_t)L.off_x + 4u > L.point_step || (uint64_t)L.off_y + 4u > L.point_step || (uint64_t)L.off_z + 4u > L.point_step ||
      (uint64_t)L.off_ring + rsize > L.point_step)
    {
      delete c;
      g_create_error = "layout: x, y, z (FLOAT32) and ring (an integer type) must lie inside point_step";
      return LFX_ERR_INVALID_ARGUMENT;
    }
    c->layout = lfx::Layout{L.point_step, L.off_x, L.off_y, L.off_z, L.off_ring, rtype, L.big_endian ? 1u : 0u};
  }
  c->drop_zero = config->drop_zero_points ? 1u : 0u;
  c->outputs = (config->outputs ? config->outputs : (uint32_t)LFX_OUT_ALL) | LFX_OUT_FEATURES;
  c->max_points = config->max_points_per_scan;
  c->max_batch = config->max_batch;
  uint32_t ring_cap = config->max_points_per_ring ? config->max_points_per_ring : LFX_MAX_RING_POINTS;
  ring_cap = ring_cap > c->max_points ? c->max_points : ring_cap;
  c->cap = ((ring_cap < 64 ? 64 : ring_cap) + 63u) & ~63u;
  if (c->cap > LFX_MAX_RING_POINTS) {
    delete c;
    g_create_error = "max_points_per_ring exceeds LFX_MAX_RING_POINTS";
    return LFX_ERR_INVALID_ARGUMENT;
  }
  c->max_rings = config->max_rings ? (config->max_rings > lfx::kRings ? lfx::kRings : config->max_rings) : lfx::kRings;
  c->ring_threads = ring_threads_for(c->cap);
  {
    // longest span (owned positions + halo) a unit of a ring of `ring_cap` points can have:
    // block <= ceil((N - 2P) / B) + 1, plus a border of P for the first / last unit, plus 2 (P + 1) halo
    const int P = c->dev.P, B = c->dev.B, N = (int)ring_cap;
    const int span = (N - 2 * P + B - 1) / B + 1 + 3 * P + 2;
    const int ch = (span + 63) / 64;
    c->unit_chunks = (uint32_t)(ch < 3 ? 3 : (ch > 6 ? 6 : ch));
    if (const char * dbg = std::getenv("LFX_DEBUG_UNIT_CHUNKS")) {c->unit_chunks = (uint32_t)std::atoi(dbg);}
    if (c->unit_chunks < 3 || c->unit_chunks > 6) {c->unit_chunks = 6;}
  }
  if (const char * dbg = std::getenv("LFX_DEBUG_RING_FLAGS")) {c->stage_flags = (uint32_t)std::atoi(dbg);}
  c->fast_path = c->dev.B <= lfx::kUnitMaxBlocks && std::getenv("LFX_DEBUG_NO_FAST_PATH") == nullptr;
  c->single_pass = std::getenv("LFX_DEBUG_TWO_PASS") == nullptr;
  // the organised-scan kernel needs to know the sensor's ring count (max_rings given) and reads PointXYZIR records
  c->fused_possible = c->fast_path && c->single_pass && config->max_rings != 0 && c->max_points < (1u << 27) && c->layout.step == 32 && c->layout.ox == 0 &&
    c->layout.oy == 4 && c->layout.oz == 8 && c->layout.oring == 20 && c->layout.rtype == LFX_FIELD_UINT16 && c->layout.be == 0;
  if (const char * dbg = std::getenv("LFX_DEBUG_FUSED")) {c->fused_env = std::atoi(dbg) != 0 ? 1 : 0;}
  if (const char * dbg = std::getenv("LFX_DEBUG_STREAM")) {c->walk_rings = std::atoi(dbg) != 0;}
  if (const char * dbg = std::getenv("LFX_DEBUG_SHORT_TAIL")) {c->short_tail_env = std::atoi(dbg) != 0 ? 1 : 0;}
  if (const char * dbg = std::getenv("LFX_DEBUG_XFORM")) {c->xform_env = std::atoi(dbg) != 0 ? 1 : 0;}
  c->slow_grid = 1024;
  if (const char * dbg = std::getenv("LFX_DEBUG_REDO_CAP")) {c->redo_cap_env = (uint32_t)std::atoi(dbg);}
  if (const char * dbg = std::getenv("LFX_DEBUG_PRE_ORDER")) {c->pre_order_env = std::atoi(dbg) != 0 ? 1 : 0;}
  if (const char * dbg = std::getenv("LFX_DEBUG_UNIT_FLAGS")) {c->unit_flags = (uint32_t)std::atoi(dbg);}
  if (const char * dbg = std::getenv("LFX_DEBUG_UNIT_LDS_PAD")) {c->unit_lds_pad = (uint32_t)std::atoi(dbg);}
  if (const char * dbg = std::getenv("LFX_DEBUG_RING_THREADS")) {c->ring_threads = (uint32_t)std::atoi(dbg);}
  c->ring_lds = lfx::ring_lds_bytes(c->cap);
  c->order_lds = lfx::order_lds_bytes(c->cap);
  c->max_chunks = (c->max_points + lfx::kChunkPoints - 1) / lfx::kChunkPoints;
  c->total_cap = (size_t)c->max_points * c->max_batch;
  if (c->total_cap >= (1ull << 32)) {
    delete c;
    g_create_error = "max_points_per_scan * max_batch must stay below 2^32";
    return LFX_ERR_CAPACITY;
  }

  hipError_t e = hipSetDevice(device_id);
  const size_t nb = c->max_batch, tc = c->total_cap, tables = nb * lfx::kRings, chunk_tab = nb * c->max_chunks * lfx::kRings;
  auto ok = [&](hipError_t r) {if (e == hipSuccess) {e = r;}};
  ok(c->scan_begin.alloc(nb + 1)); ok(c->scan_info.alloc(nb * 4));
  ok(c->chunk_hist.alloc(chunk_tab)); ok(c->chunk_base.alloc(chunk_tab));
  ok(c->ring_count.alloc(tables)); ok(c->chunk_flags.alloc(nb * c->max_chunks));
  ok(c->ring_status.alloc(tables)); ok(c->ring_nedge.alloc(tables));
  ok(c->ring_nsurf.alloc(tables)); ok(c->ring_ebase.alloc(tables)); ok(c->ring_sbase.alloc(tables));
  ok(c->ring_flags.alloc(tables + lfx::kCounters)); ok(c->slow_list.alloc(tables)); ok(c->defer_list.alloc(tables));
  ok(c->fb_list.alloc(nb)); ok(c->xform.alloc(tables));
  ok(c->redo_list.alloc(tables));
  ok(c->unit_ne.alloc(tables * lfx::kUnitMaxBlocks)); ok(c->unit_ns.alloc(tables * lfx::kUnitMaxBlocks));
  ok(c->unit_span.alloc(tables * lfx::kUnitMaxBlocks));
  const size_t rc = nb * c->max_rings * c->cap;      // ring-major arrays: fixed capacity per ring id
  ok(c->sxy.alloc(rc)); ok(c->sz.alloc(rc)); ok(c->sidx.alloc(rc)); ok(c->rec_pts.alloc(rc)); ok(c->rec_idx.alloc(rc));
  ok(c->label_s.alloc(rc)); ok(c->curv_s.alloc(rc));
  ok(c->d_label.alloc(c->max_points)); ok(c->d_curv.alloc(c->max_points)); ok(c->d_sidx.alloc(c->max_points));
  ok(c->edge_pts.alloc(tc)); ok(c->surf_pts.alloc(tc)); ok(c->edge_idx.alloc(tc)); ok(c->surf_idx.alloc(tc));
  ok(c->unit_tab.alloc(1));
  if (e == hipSuccess) {
    e = hipHostMalloc(reinterpret_cast<void **>(&c->h_counters), 4 * lfx::kCounters, hipHostMallocDefault);
    if (e == hipSuccess) {std::memset(c->h_counters, 0, 4 * lfx::kCounters);}
  }
  if (e == hipSuccess) {
    const lfx::UnitTables t{c->label_s.p, c->curv_s.p, c->rec_pts.p, c->rec_idx.p, c->ring_status.p, c->unit_ne.p,
      c->unit_ns.p, c->unit_span.p, c->ring_flags.p, c->scan_info.p,
      c->ring_flags.p + (size_t)c->max_batch * lfx::kRings + lfx::kCntFallback, c->fb_list.p};
    e = hipMemcpy(c->unit_tab.p, &t, sizeof(t), hipMemcpyHostToDevice);
  }
  if (e == hipSuccess) {e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);}
  // (the attribute is per function, not per context: always the worst case, so that contexts of different ring
  // capacities can live side by side)
  if (e == hipSuccess) {
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(lfx::ring_extract_kernel),
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lfx::ring_lds_bytes(LFX_MAX_RING_POINTS));
  }
  if (e == hipSuccess) {
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(lfx::ring_order_kernel),
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lfx::order_lds_bytes(LFX_MAX_RING_POINTS));
  }
  if (e == hipSuccess) {
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(lfx::ring_stage_kernel),
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lfx::ring_lds_bytes(LFX_MAX_RING_POINTS));
  }
  if (e != hipSuccess) {
    g_create_error = std::string("device setup failed: ") + hipGetErrorString(e);
    lfx_destroy(c);
    return e == hipErrorOutOfMemory ? LFX_ERR_OUT_OF_MEMORY : LFX_ERR_HIP;
  }
  *out = c;
  return LFX_OK;
}

void lfx_destroy(lfx_ctx * c)
{
  if (!c) {return;}
  (void)hipSetDevice(c->device);
  if (c->stream) {(void)hipStreamSynchronize(c->stream);}
  for (auto & sp : c->spans) {(void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b);}
  for (auto & ev : c->free_events) {(void)hipEventDestroy(ev);}
  c->scan_begin.release(); c->scan_info.release(); c->chunk_hist.release(); c->chunk_base.release();
  c->ring_count.release(); c->chunk_flags.release(); c->d_label.release(); c->d_curv.release(); c->d_sidx.release();
  c->ring_status.release(); c->ring_nedge.release(); c->ring_nsurf.release(); c->ring_ebase.release();
  c->ring_sbase.release(); c->ring_flags.release(); c->slow_list.release(); c->defer_list.release(); c->redo_list.release(); c->fb_list.release(); c->xform.release(); c->unit_ne.release(); c->unit_ns.release(); c->unit_span.release();
  c->sxy.release(); c->sz.release(); c->sidx.release(); c->rec_pts.release(); c->rec_idx.release(); c->label_s.release();
  c->unit_tab.release();
  if (c->h_counters) {(void)hipHostFree(c->h_counters); c->h_counters = nullptr;}
  c->curv_s.release(); c->edge_pts.release(); c->surf_pts.release(); c->edge_idx.release(); c->surf_idx.release();
  c->staging.release();
  c->h_in.release(); c->h_out.release(); c->vox_scratch.release(); c->align_scratch.release(); c->align_surface.release(); c->h_align.release();
  if (c->h_status) {(void)hipHostFree(c->h_status); c->h_status = nullptr;}
  if (c->stream) {(void)hipStreamDestroy(c->stream);}
  delete c;
}

int lfx_extract_batch_device(lfx_ctx * c, const void * d_points, const uint32_t * n_points, uint32_t batch, void * stream)
{
  if (!c) {return LFX_ERR_INVALID_ARGUMENT;}
  return run_batch(c, d_points, n_points, batch, static_cast<hipStream_t>(stream));
}

int lfx_device_results(const lfx_ctx * c, lfx_device_view * v)
{
  if (!c || !v) {return LFX_ERR_INVALID_ARGUMENT;}
  v->batch = c->last_batch;
  v->max_rings = c->max_rings;
  v->ring_capacity = c->cap;
  v->scan_begin = c->scan_begin.p;
  v->labels_sorted = c->label_s.p;
  v->curvature_sorted = c->curv_s.p;
  v->sorted_index = c->sidx.p;
  v->scan_info = c->scan_info.p;
  v->ring_count = c->ring_count.p;
  v->ring_status = c->ring_status.p;
  v->edge_points = reinterpret_cast<const float *>(c->edge_pts.p);
  v->edge_index = c->edge_idx.p;
  v->surface_points = reinterpret_cast<const float *>(c->surf_pts.p);
  v->surface_index = c->surf_idx.p;
  return LFX_OK;
}

int lfx_batch_status(lfx_ctx * c, void * stream, uint32_t * first_bad)
{
  if (!c) {return LFX_ERR_INVALID_ARGUMENT;}
  if (c->last_batch == 0) {return fail(c, LFX_ERR_INVALID_ARGUMENT, "no batch has been extracted yet");}
  LFX_HIP(c, hipSetDevice(c->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (!c->h_status) {
    LFX_HIP(c, hipHostMalloc(reinterpret_cast<void **>(&c->h_status), (size_t)c->max_batch * 16, hipHostMallocDefault));
  }
  LFX_HIP(c, hipMemcpyAsync(c->h_status, c->scan_info.p, (size_t)c->last_batch * 16, hipMemcpyDeviceToHost, st));
  LFX_HIP(c, hipStreamSynchronize(st));
  for (uint32_t s = 0; s < c->last_batch; s++) {
    const uint32_t e = c->h_status[s * 4 + lfx::kInfoError];
    if (e & (lfx::kErrRingId | lfx::kErrTimeout)) {
      if (first_bad) {*first_bad = s;}
      return (e & lfx::kErrRingId) ?
             fail(c, LFX_ERR_RING_ID, "a point carries a ring id the context was not created for (max_rings / LFX_MAX_RING_ID)") :
             fail(c, LFX_ERR_HIP, "ring bucketing timed out waiting for an earlier chunk (set LFX_DEBUG_TWO_PASS=1)");
    }
  }
  return LFX_OK;
}

int lfx_scan_routes(lfx_ctx * c, void * stream, uint8_t * routes)
{
  if (!c || !routes) {return LFX_ERR_INVALID_ARGUMENT;}
  if (c->last_batch == 0) {return fail(c, LFX_ERR_INVALID_ARGUMENT, "no batch has been extracted yet");}
  LFX_HIP(c, hipSetDevice(c->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (!c->h_status) {
    LFX_HIP(c, hipHostMalloc(reinterpret_cast<void **>(&c->h_status), (size_t)c->max_batch * 16, hipHostMallocDefault));
  }
  LFX_HIP(c, hipMemcpyAsync(c->h_status, c->scan_info.p, (size_t)c->last_batch * 16, hipMemcpyDeviceToHost, st));
  LFX_HIP(c, hipStreamSynchronize(st));
  for (uint32_t s = 0; s < c->last_batch; s++) {
    const uint32_t e = c->h_status[s * 4 + lfx::kInfoError];
    routes[s] = lfx::scan_is_organised(e) ? (c->last_used_xform ? 2 : 1) : 0;
  }
  return LFX_OK;
}

int lfx_host_alloc(lfx_ctx * c, size_t bytes, void ** out)
{
  if (!c || !out || bytes == 0) {return LFX_ERR_INVALID_ARGUMENT;}
  LFX_HIP(c, hipSetDevice(c->device));
  if (hipHostMalloc(out, bytes, hipHostMallocDefault) != hipSuccess) {
    *out = nullptr;
    return fail(c, LFX_ERR_OUT_OF_MEMORY, "cannot allocate pinned host memory");
  }
  return LFX_OK;
}

void lfx_host_free(lfx_ctx *, void * ptr)
{
  if (ptr) {(void)hipHostFree(ptr);}
}

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

int lfx_download_scan(lfx_ctx * c, uint32_t scan, void * stream, lfx_scan_result * out)
{
  if (!c || !out) {return LFX_ERR_INVALID_ARGUMENT;}
  LFX_HIP(c, hipSetDevice(c->device));
  return fetch(c, scan, 1, static_cast<hipStream_t>(stream), LFX_OUT_ALL, out);
}

namespace
{
int extract_batch_impl(
  lfx_ctx * c, const void * const * points, const size_t * n_points, uint32_t batch, uint32_t mask, lfx_scan_result * out)
{
  if (!c || !points || !n_points || !out || batch == 0) {return LFX_ERR_INVALID_ARGUMENT;}
  if (batch > c->max_batch) {return fail(c, LFX_ERR_CAPACITY, "batch exceeds max_batch");}
  LFX_HIP(c, hipSetDevice(c->device));
  std::vector<uint32_t> n32(batch);
  size_t total = 0;
  for (uint32_t s = 0; s < batch; s++) {
    if (n_points[s] > c->max_points) {return fail(c, LFX_ERR_CAPACITY, "scan exceeds max_points_per_scan");}
    if (n_points[s] && !points[s]) {return fail(c, LFX_ERR_INVALID_ARGUMENT, "points[s] is NULL");}
    n32[s] = (uint32_t)n_points[s];
    total += n_points[s];
  }
  if (total == 0) {
    for (uint32_t s = 0; s < batch; s++) {std::memset(&out[s], 0, sizeof(out[s]));}
    return LFX_OK;
  }
  if (!c->staging.p) {
    if (c->staging.alloc(c->total_cap * c->layout.step) != hipSuccess) {
      return fail(c, LFX_ERR_OUT_OF_MEMORY, "cannot allocate the input staging buffer");
    }
  }
  // Input: a pinned buffer (lfx_host_alloc, or anything the caller registered with HIP) goes to the device by DMA;
  // pageable memory is copied through the context's pinned staging buffer in pieces, each piece's DMA running while
  // the next is being copied.
  size_t at = 0;
  for (uint32_t s = 0; s < batch; s++) {
    const size_t bytes = n_points[s] * c->layout.step;
    if (bytes) {
      hipPointerAttribute_t attr;
      const bool pinned = hipPointerGetAttributes(&attr, points[s]) == hipSuccess && attr.type == hipMemoryTypeHost;
      if (!pinned) {(void)hipGetLastError();}              // a plain pointer is "invalid value" to the query: not an error here
      if (pinned) {
        LFX_HIP(c, hipMemcpyAsync(c->staging.p + at, points[s], bytes, hipMemcpyHostToDevice, c->stream));
      } else {
        if (c->h_in.bytes < at + bytes) {
          // (growing the staging buffer must not pull the rug from under copies already queued)
          LFX_HIP(c, hipStreamSynchronize(c->stream));
          if (c->h_in.reserve(total * c->layout.step) != hipSuccess) {
            return fail(c, LFX_ERR_OUT_OF_MEMORY, "cannot allocate the pinned input staging buffer");
          }
        }
        constexpr size_t kPiece = 1u << 20;
        const uint8_t * src = static_cast<const uint8_t *>(points[s]);
        for (size_t o = 0; o < bytes; o += kPiece) {
          const size_t len = bytes - o < kPiece ? bytes - o : kPiece;
          std::memcpy(c->h_in.p + at + o, src + o, len);
          LFX_HIP(c, hipMemcpyAsync(c->staging.p + at + o, c->h_in.p + at + o, len, hipMemcpyHostToDevice, c->stream));
        }
      }
    }
    at += bytes;
  }
  const int rc = run_batch(c, c->staging.p, n32.data(), batch, c->stream);
  if (rc != LFX_OK) {return rc;}
  return fetch(c, 0, batch, c->stream, mask, out);
}
}  // namespace

int lfx_extract_batch(
  lfx_ctx * c, const void * const * points, const size_t * n_points, uint32_t batch, lfx_scan_result * out)
{
  if (!c) {return LFX_ERR_INVALID_ARGUMENT;}
  return extract_batch_impl(c, points, n_points, batch, c->outputs, out);
}

int lfx_extract(lfx_ctx * c, const void * points, size_t n_points, lfx_scan_result * out)
{
  const void * p[1] = {points};
  const size_t n[1] = {n_points};
  return lfx_extract_batch(c, p, n, 1, out);
}

// ---------------------------------------------------------------------------- multi-GPU gather over RCCL
namespace
{
struct Rccl
{
  void * lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string why;
};

Rccl * rccl()
{
  static Rccl r;
  if (r.lib || !r.why.empty()) {return &r;}
  // the copy already in the process (PyTorch ships one under the same SONAME) or the ROCm installation's
  for (const char * name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (r.lib) {break;}
  }
  if (!r.lib) {r.why = std::string("cannot open librccl: ") + dlerror(); return &r;}
  bool ok = true;
  auto sym = [&](const char * n) {void * p = dlsym(r.lib, n); if (!p) {ok = false; r.why = std::string("librccl lacks ") + n;} return p;};
  r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
  r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
  r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
  r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
  r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
  r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
  r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
  r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
  r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
  if (!ok) {dlclose(r.lib); r.lib = nullptr;}
  return &r;
}
}  // namespace

struct lfx_comm
{
  lfx_ctx * ctx = nullptr;
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  uint32_t * d_mine = nullptr;      // [2] this rank's totals
  uint32_t * d_totals = nullptr;    // [world][2]
  uint32_t * h_totals = nullptr;    // pinned [world][2]
  hipEvent_t landed = nullptr;      // the totals are in h_totals
  bool counts_pending = false;
};

#define LFX_NCCL(ctx, call) \
  do { \
    const ncclResult_t r_ = (call); \
    if (r_ != ncclSuccess) { \
      (ctx)->err = std::string(#call) + ": " + rccl()->GetErrorString(r_); \
      return LFX_ERR_HIP; \
    } \
  } while (0)

extern "C" {

int lfx_comm_unique_id(uint8_t id[LFX_COMM_ID_BYTES])
{
  static_assert(sizeof(ncclUniqueId) == LFX_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
  if (!id) {return LFX_ERR_INVALID_ARGUMENT;}
  Rccl * r = rccl();
  if (!r->lib) {g_create_error = r->why; return LFX_ERR_NO_DEVICE;}
  ncclUniqueId u;
  if (r->GetUniqueId(&u) != ncclSuccess) {g_create_error = "ncclGetUniqueId failed"; return LFX_ERR_HIP;}
  std::memcpy(id, &u, LFX_COMM_ID_BYTES);
  return LFX_OK;
}

int lfx_comm_create(lfx_ctx * c, const uint8_t id[LFX_COMM_ID_BYTES], int rank, int world, lfx_comm ** out)
{
  if (!c || !id || !out || world < 1 || rank < 0 || rank >= world) {return LFX_ERR_INVALID_ARGUMENT;}
  *out = nullptr;
  Rccl * r = rccl();
  if (!r->lib) {return fail(c, LFX_ERR_NO_DEVICE, r->why);}
  LFX_HIP(c, hipSetDevice(c->device));
  lfx_comm * m = new lfx_comm();
  m->ctx = c; m->rank = rank; m->world = world;
  ncclUniqueId u;
  std::memcpy(&u, id, LFX_COMM_ID_BYTES);
  const ncclResult_t nr = r->CommInitRank(&m->comm, world, u, rank);
  if (nr != ncclSuccess) {
    c->err = std::string("ncclCommInitRank: ") + r->GetErrorString(nr);
    delete m;
    return LFX_ERR_HIP;
  }
  hipError_t e = hipMalloc(reinterpret_cast<void **>(&m->d_mine), 16);
  if (e == hipSuccess) {e = hipMalloc(reinterpret_cast<void **>(&m->d_totals), (size_t)world * 8 + 16);}
  if (e == hipSuccess) {e = hipHostMalloc(reinterpret_cast<void **>(&m->h_totals), (size_t)world * 8 + 16, hipHostMallocDefault);}
  if (e == hipSuccess) {e = hipEventCreateWithFlags(&m->landed, hipEventDisableTiming);}
  if (e != hipSuccess) {
    c->err = std::string("lfx_comm_create: ") + hipGetErrorString(e);
    lfx_comm_destroy(m);
    return LFX_ERR_HIP;
  }
  *out = m;
  return LFX_OK;
}

void lfx_comm_destroy(lfx_comm * m)
{
  if (!m) {return;}
  if (m->comm && rccl()->lib) {(void)rccl()->CommDestroy(m->comm);}
  if (m->d_mine) {(void)hipFree(m->d_mine);}
  if (m->d_totals) {(void)hipFree(m->d_totals);}
  if (m->h_totals) {(void)hipHostFree(m->h_totals);}
  if (m->landed) {(void)hipEventDestroy(m->landed);}
  delete m;
}

int lfx_gather_counts(lfx_ctx * c, lfx_comm * m, const uint32_t * d_offsets, uint32_t batch, void * stream)
{
  if (!c || !m || !d_offsets || batch == 0) {return LFX_ERR_INVALID_ARGUMENT;}
  Rccl * r = rccl();
  hipStream_t st = static_cast<hipStream_t>(stream);
  LFX_HIP(c, hipSetDevice(c->device));
  // totals: d_offsets[batch] (edge) and d_offsets[2 * batch + 1] (surface)
  LFX_HIP(c, hipMemcpyAsync(m->d_mine, d_offsets + batch, 4, hipMemcpyDeviceToDevice, st));
  LFX_HIP(c, hipMemcpyAsync(m->d_mine + 1, d_offsets + 2 * batch + 1, 4, hipMemcpyDeviceToDevice, st));
  LFX_NCCL(c, r->AllGather(m->d_mine, m->d_totals, 2, ncclUint32, m->comm, st));
  LFX_HIP(c, hipMemcpyAsync(m->h_totals, m->d_totals, (size_t)m->world * 8, hipMemcpyDeviceToHost, st));
  LFX_HIP(c, hipEventRecord(m->landed, st));
  m->counts_pending = true;
  return LFX_OK;
}

int lfx_gather_payload(
  lfx_ctx * c, lfx_comm * m, int dst, const float * d_edge, const float * d_surface, const uint32_t * d_offsets,
  uint32_t batch, uint32_t fpp, float * d_edge_all, float * d_surface_all, uint32_t * d_offsets_all, size_t capacity_points,
  uint64_t * counts_out, void * stream)
{
  if (!c || !m || !d_edge || !d_surface || !d_offsets || batch == 0 || dst < 0 || dst >= m->world || (fpp != 3 && fpp != 4)) {
    return LFX_ERR_INVALID_ARGUMENT;
  }
  if (m->rank == dst && (!d_edge_all || !d_surface_all || !d_offsets_all)) {return LFX_ERR_INVALID_ARGUMENT;}
  if (!m->counts_pending) {return fail(c, LFX_ERR_INVALID_ARGUMENT, "lfx_gather_payload without lfx_gather_counts");}
  Rccl * r = rccl();
  hipStream_t st = static_cast<hipStream_t>(stream);
  LFX_HIP(c, hipSetDevice(c->device));
  LFX_HIP(c, hipEventSynchronize(m->landed));
  m->counts_pending = false;
  uint64_t sum_e = 0, sum_s = 0;
  for (int k = 0; k < m->world; k++) {
    if (counts_out) {counts_out[2 * k] = m->h_totals[2 * k]; counts_out[2 * k + 1] = m->h_totals[2 * k + 1];}
    sum_e += m->h_totals[2 * k];
    sum_s += m->h_totals[2 * k + 1];
  }
  // every rank takes the same decision from the same totals and the same capacity_points, so nobody is left waiting
  // in a send or a receive that the other side never posts
  if (sum_e > capacity_points || sum_s > capacity_points) {
    return fail(c, LFX_ERR_CAPACITY, "gathered clouds exceed capacity_points of the destination rank");
  }
  const size_t tab = 2 * ((size_t)batch + 1);
  const uint32_t me = m->h_totals[2 * m->rank], ms = m->h_totals[2 * m->rank + 1];
  if (m->rank != dst) {
    LFX_NCCL(c, r->GroupStart());
    LFX_NCCL(c, r->Send(d_edge, (size_t)me * fpp, ncclFloat32, dst, m->comm, st));
    LFX_NCCL(c, r->Send(d_surface, (size_t)ms * fpp, ncclFloat32, dst, m->comm, st));
    LFX_NCCL(c, r->Send(d_offsets, tab, ncclUint32, dst, m->comm, st));
    LFX_NCCL(c, r->GroupEnd());
    return LFX_OK;
  }
  size_t at_e = 0, at_s = 0;
  LFX_NCCL(c, r->GroupStart());
  for (int k = 0; k < m->world; k++) {
    const uint32_t ne = m->h_totals[2 * k], ns = m->h_totals[2 * k + 1];
    if (k != dst) {
      LFX_NCCL(c, r->Recv(d_edge_all + at_e * fpp, (size_t)ne * fpp, ncclFloat32, k, m->comm, st));
      LFX_NCCL(c, r->Recv(d_surface_all + at_s * fpp, (size_t)ns * fpp, ncclFloat32, k, m->comm, st));
      LFX_NCCL(c, r->Recv(d_offsets_all + (size_t)k * tab, tab, ncclUint32, k, m->comm, st));
    }
    at_e += ne;
    at_s += ns;
  }
  LFX_NCCL(c, r->GroupEnd());
  // dst's own part: plain device copies on the same stream
  at_e = 0; at_s = 0;
  for (int k = 0; k < dst; k++) {at_e += m->h_totals[2 * k]; at_s += m->h_totals[2 * k + 1];}
  if (me) {LFX_HIP(c, hipMemcpyAsync(d_edge_all + at_e * fpp, d_edge, (size_t)me * fpp * 4, hipMemcpyDeviceToDevice, st));}
  if (ms) {LFX_HIP(c, hipMemcpyAsync(d_surface_all + at_s * fpp, d_surface, (size_t)ms * fpp * 4, hipMemcpyDeviceToDevice, st));}
  LFX_HIP(c, hipMemcpyAsync(d_offsets_all + (size_t)dst * tab, d_offsets, tab * 4, hipMemcpyDeviceToDevice, st));
  return LFX_OK;
}

int lfx_gather(
  lfx_ctx * c, lfx_comm * m, int dst, const float * d_edge, const float * d_surface, const uint32_t * d_offsets,
  uint32_t batch, uint32_t fpp, float * d_edge_all, float * d_surface_all, uint32_t * d_offsets_all, size_t capacity_points,
  uint64_t * counts_out, void * stream)
{
  const int rc = lfx_gather_counts(c, m, d_offsets, batch, stream);
  if (rc != LFX_OK) {return rc;}
  return lfx_gather_payload(c, m, dst, d_edge, d_surface, d_offsets, batch, fpp, d_edge_all, d_surface_all, d_offsets_all,
           capacity_points, counts_out, stream);
}

}  // extern "C"

// ---------------------------------------------------------------------------- voxel-grid Downsample
extern "C" {

int lfx_voxel_downsample(
  lfx_ctx * c, const float * d_points, const uint32_t * d_begin, const uint32_t * d_count, uint32_t count_stride,
  uint32_t n_clouds, size_t total_points, float leaf, float * d_out, uint32_t * d_out_count, uint32_t * d_status, void * stream)
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
  hipLaunchKernelGGL(lfx::voxel_downsample_kernel, dim3(n_clouds), dim3(lfx::kVoxThreads), 0, static_cast<hipStream_t>(stream),
    reinterpret_cast<const float4 *>(d_points), d_begin, d_count, count_stride, leaf, w, w + total_points, w + 2 * total_points,
    w + 3 * total_points, reinterpret_cast<float4 *>(d_out), d_out_count, d_status);
  LFX_HIP(c, hipGetLastError());
  return LFX_OK;
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

// ---------------------------------------------------------------------------- the map (KDTreeEigen's place)
struct lfx_map
{
  int device = 0;
  DevBuf<float4> pts;                    // the map's own copy of the points (sorted by cell when there is a grid)
  DevBuf<uint32_t> start;                // first point of every cell, + 1
  lfx::MapIndex index{};
  float cell = 0.f;
};

namespace
{
void launch_rows(bool surface, const lfx::MapIndex & mi, const lfx::MapPose & P, uint32_t k, const float * d_points,
  const uint32_t * d_begin, const uint32_t * d_count, uint32_t count_stride, uint32_t n_clouds, uint32_t longest, double * d_residual,
  double * d_jacobian, const lfx::AlignState * states, hipStream_t st)
{
  const bool wave = mi.start != nullptr;              // a grid: one query per wave; no grid: one per thread, the map through LDS
  const dim3 grid(wave ? longest : (longest + 127u) / 128u, n_clouds), block(wave ? 64 : 128);
  const float4 * pts = reinterpret_cast<const float4 *>(d_points);
#define LFX_ROWS(S, M) hipLaunchKernelGGL((lfx::scan_to_map_kernel<S, M>), grid, block, 0, st, mi, P, k, pts, d_begin, d_count, \
    count_stride, d_residual, d_jacobian, states)
  if (wave) {
    if (surface) {LFX_ROWS(true, lfx::kSearchGridWave);} else {LFX_ROWS(false, lfx::kSearchGridWave);}
  } else {
    if (surface) {LFX_ROWS(true, lfx::kSearchWholeMap);} else {LFX_ROWS(false, lfx::kSearchWholeMap);}
  }
#undef LFX_ROWS
}
}  // namespace

extern "C" {

int lfx_map_create(lfx_ctx * c, const float * d_points, uint32_t n_points, float cell_size, lfx_map ** out, void * stream)
{
  if (!c || !d_points || !out || n_points == 0) {return LFX_ERR_INVALID_ARGUMENT;}
  if (!(cell_size >= 0.f) || !std::isfinite(cell_size)) {return fail(c, LFX_ERR_INVALID_ARGUMENT, "cell_size must be >= 0 (0: no grid)");}
  LFX_HIP(c, hipSetDevice(c->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  lfx_map * m = new (std::nothrow) lfx_map();
  if (!m) {return fail(c, LFX_ERR_OUT_OF_MEMORY, "cannot allocate the map");}
  m->device = c->device;
  auto give_up = [&](int code, const char * why) {lfx_map_destroy(m); return fail(c, code, why);};
  if (m->pts.alloc(n_points) != hipSuccess) {m->pts.p = nullptr; return give_up(LFX_ERR_OUT_OF_MEMORY, "cannot allocate the map's points");}
  lfx::MapIndex & mi = m->index;
  mi.pts = m->pts.p; mi.start = nullptr; mi.n = n_points;
  mi.ox = mi.oy = mi.oz = 0.; mi.h = 0.; mi.inv_h = 0.; mi.nx = mi.ny = mi.nz = 1;
  const float4 * src = reinterpret_cast<const float4 *>(d_points);
  if (cell_size == 0.f) {
    hipError_t e = hipMemcpyAsync(m->pts.p, src, sizeof(float4) * (size_t)n_points, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) {e = hipStreamSynchronize(st);}
    if (e != hipSuccess) {return give_up(LFX_ERR_HIP, hipGetErrorString(e));}
    *out = m;
    return LFX_OK;
  }
  // bounds of the map
  uint32_t * d_bounds = nullptr;
  if (hipMalloc(reinterpret_cast<void **>(&d_bounds), 6 * sizeof(uint32_t)) != hipSuccess) {return give_up(LFX_ERR_OUT_OF_MEMORY, "cannot allocate the map's bounds");}
  const uint32_t init[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
  uint32_t got[6];
  hipError_t e = hipMemcpyAsync(d_bounds, init, sizeof(init), hipMemcpyHostToDevice, st);
  if (e == hipSuccess) {
    const uint32_t blocks = std::min<uint32_t>((n_points + 255u) / 256u, 2048u);
    hipLaunchKernelGGL(lfx::map_bounds_kernel, dim3(blocks), dim3(256), 0, st, src, n_points, d_bounds);
    e = hipMemcpyAsync(got, d_bounds, sizeof(got), hipMemcpyDeviceToHost, st);
  }
  if (e == hipSuccess) {e = hipStreamSynchronize(st);}
  (void)hipFree(d_bounds);
  if (e != hipSuccess) {return give_up(LFX_ERR_HIP, hipGetErrorString(e));}
  auto back = [](uint32_t u) {
      const uint32_t b = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
      float f;
      std::memcpy(&f, &b, 4);
      return (double)f;
    };
  const double lo[3] = {back(got[0]), back(got[1]), back(got[2])}, hi[3] = {back(got[3]), back(got[4]), back(got[5])};
  for (int a = 0; a < 3; a++) {
    if (!std::isfinite(lo[a]) || !std::isfinite(hi[a])) {return give_up(LFX_ERR_INVALID_ARGUMENT, "the map holds a point that is not finite");}
  }
  // the grid: cubic cells of the asked size, grown until the grid has at most 2^25 cells
  double h = (double)cell_size;
  const double limit = 33554432.;
  int dims[3];
  for (;;) {
    double cells = 1.;
    for (int a = 0; a < 3; a++) {
      const double na = std::floor((hi[a] - lo[a]) / h) + 1.;
      dims[a] = na > 2147483647. ? 2147483647 : (int)na;
      cells *= na;
    }
    if (cells <= limit) {break;}
    h *= std::max(1.05, std::cbrt(cells / limit));
  }
  mi.ox = lo[0]; mi.oy = lo[1]; mi.oz = lo[2]; mi.h = h; mi.inv_h = 1. / h; mi.nx = dims[0]; mi.ny = dims[1]; mi.nz = dims[2];
  m->cell = (float)h;
  const size_t cells = (size_t)dims[0] * dims[1] * dims[2];
  const uint32_t n_blocks = (uint32_t)((cells + lfx::kScanItems - 1) / lfx::kScanItems);
  DevBuf<uint32_t> cell_count, partial;
  if (m->start.alloc(cells + 1) != hipSuccess) {m->start.p = nullptr; return give_up(LFX_ERR_OUT_OF_MEMORY, "cannot allocate the map's cells");}
  if (cell_count.alloc(cells) != hipSuccess || partial.alloc(n_blocks + 1) != hipSuccess) {
    cell_count.release(); partial.release();
    return give_up(LFX_ERR_OUT_OF_MEMORY, "cannot allocate the map's cells");
  }
  e = hipMemsetAsync(cell_count.p, 0, cells * sizeof(uint32_t), st);
  if (e == hipSuccess) {
    const dim3 per_point((n_points + 255u) / 256u);
    hipLaunchKernelGGL(lfx::map_count_kernel, per_point, dim3(256), 0, st, mi, src, cell_count.p);
    hipLaunchKernelGGL(lfx::cell_block_sum_kernel, dim3(n_blocks), dim3(lfx::kScanThreads), 0, st, cell_count.p, cells, partial.p);
    hipLaunchKernelGGL(lfx::cell_partial_scan_kernel, dim3(1), dim3(lfx::kScanThreads), 0, st, partial.p, n_blocks);
    hipLaunchKernelGGL(lfx::cell_start_kernel, dim3(n_blocks), dim3(lfx::kScanThreads), 0, st, cell_count.p, cells, partial.p, m->start.p, n_points);
    hipLaunchKernelGGL(lfx::map_scatter_kernel, per_point, dim3(256), 0, st, mi, src, cell_count.p, m->start.p, m->pts.p);
    e = hipGetLastError();
  }
  if (e == hipSuccess) {e = hipStreamSynchronize(st);}
  cell_count.release(); partial.release();
  if (e != hipSuccess) {return give_up(LFX_ERR_HIP, hipGetErrorString(e));}
  mi.start = m->start.p;
  *out = m;
  return LFX_OK;
}

void lfx_map_destroy(lfx_map * m)
{
  if (!m) {return;}
  (void)hipSetDevice(m->device);
  m->pts.release(); m->start.release();
  delete m;
}

int lfx_map_create_host(lfx_ctx * c, const float * points, uint32_t n_points, float cell_size, lfx_map ** out, void * stream)
{
  if (!c || !points || !out || n_points == 0) {return LFX_ERR_INVALID_ARGUMENT;}
  LFX_HIP(c, hipSetDevice(c->device));
  DevBuf<float> staged;
  if (staged.alloc(4 * (size_t)n_points) != hipSuccess) {return fail(c, LFX_ERR_OUT_OF_MEMORY, "cannot stage the map's points");}
  hipError_t e = hipMemcpyAsync(staged.p, points, sizeof(float) * 4 * (size_t)n_points, hipMemcpyHostToDevice, static_cast<hipStream_t>(stream));
  if (e == hipSuccess) {e = hipStreamSynchronize(static_cast<hipStream_t>(stream));}
  if (e != hipSuccess) {staged.release(); return fail(c, LFX_ERR_HIP, hipGetErrorString(e));}
  const int rc = lfx_map_create(c, staged.p, n_points, cell_size, out, stream);
  staged.release();
  return rc;
}

int lfx_map_info(const lfx_map * m, uint32_t * n_points, float * cell_size, int32_t dims[3])
{
  if (!m) {return LFX_ERR_INVALID_ARGUMENT;}
  if (n_points) {*n_points = m->index.n;}
  if (cell_size) {*cell_size = m->index.start ? m->cell : 0.f;}
  if (dims) {dims[0] = m->index.nx; dims[1] = m->index.ny; dims[2] = m->index.nz;}
  return LFX_OK;
}

int lfx_map_nearest(
  lfx_ctx * c, const lfx_map * m, const double * d_queries, uint32_t n_queries, uint32_t k, double * d_neighbours,
  double * d_squared_distances, uint32_t * d_indices, void * stream)
{
  if (!c || !m || !d_queries) {return LFX_ERR_INVALID_ARGUMENT;}
  if (k == 0 || k > (uint32_t)lfx::kNearestMax || m->index.n < k) {
    return fail(c, LFX_ERR_INVALID_ARGUMENT, "k must be in [1, 16] and the map must hold that many points");
  }
  if (m->device != c->device) {return fail(c, LFX_ERR_INVALID_ARGUMENT, "the map lives on another device");}
  if (n_queries == 0) {return LFX_OK;}
  LFX_HIP(c, hipSetDevice(c->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (m->index.start) {
    hipLaunchKernelGGL(lfx::map_nearest_kernel<lfx::kSearchGridWave>, dim3(n_queries), dim3(64), 0, st, m->index, d_queries,
      n_queries, k, d_neighbours, d_squared_distances, d_indices);
  } else {
    hipLaunchKernelGGL(lfx::map_nearest_kernel<lfx::kSearchWholeMap>, dim3((n_queries + 127u) / 128u), dim3(128), 0, st, m->index,
      d_queries, n_queries, k, d_neighbours, d_squared_distances, d_indices);
  }
  LFX_HIP(c, hipGetLastError());
  return LFX_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------- scan-to-map residuals
extern "C" {

int lfx_scan_to_map_residuals(
  lfx_ctx * c, int kind, const lfx_map * map, const double pose[12], uint32_t n_neighbors,
  const float * d_points, const uint32_t * d_begin, const uint32_t * d_count, uint32_t count_stride, uint32_t n_clouds,
  uint32_t max_points_per_cloud, double * d_residual, double * d_jacobian, void * stream)
{
  if (!c || !map || !pose || !d_points || !d_begin || !d_count || !d_residual || !d_jacobian || n_clouds == 0 || count_stride == 0 ||
    (kind != LFX_RESIDUAL_EDGE && kind != LFX_RESIDUAL_SURFACE))
  {
    return LFX_ERR_INVALID_ARGUMENT;
  }
  if (n_neighbors == 0 || n_neighbors > (uint32_t)lfx::kNearestMax || map->index.n < n_neighbors || (kind == LFX_RESIDUAL_SURFACE && n_neighbors < 3)) {
    return fail(c, LFX_ERR_INVALID_ARGUMENT, "n_neighbors must be in [1, 16] (>= 3 for planes) and the map must hold that many points");
  }
  if (map->device != c->device) {return fail(c, LFX_ERR_INVALID_ARGUMENT, "the map lives on another device");}
  if (max_points_per_cloud == 0) {return LFX_OK;}
  LFX_HIP(c, hipSetDevice(c->device));
  lfx::MapPose P;
  for (int i = 0; i < 12; i++) {P.m[i] = pose[i];}
  {
    // Eigen::Quaterniond(Matrix3d): the branch on the trace, then on the largest diagonal entry
    auto M = [&](int r, int col) {return pose[4 * r + col];};
    double q[3], w, t = M(0, 0) + M(1, 1) + M(2, 2);
    if (t > 0.) {
      t = std::sqrt(t + 1.0);
      w = 0.5 * t;
      t = 0.5 / t;
      q[0] = (M(2, 1) - M(1, 2)) * t; q[1] = (M(0, 2) - M(2, 0)) * t; q[2] = (M(1, 0) - M(0, 1)) * t;
    } else {
      int i = 0;
      if (M(1, 1) > M(0, 0)) {i = 1;}
      if (M(2, 2) > M(i, i)) {i = 2;}
      const int j = (i + 1) % 3, k = (j + 1) % 3;
      t = std::sqrt(M(i, i) - M(j, j) - M(k, k) + 1.0);
      q[i] = 0.5 * t;
      t = 0.5 / t;
      w = (M(k, j) - M(j, k)) * t;
      q[j] = (M(j, i) + M(i, j)) * t;
      q[k] = (M(k, i) + M(i, k)) * t;
    }
    P.qw = w; P.qx = q[0]; P.qy = q[1]; P.qz = q[2];
  }
  launch_rows(kind == LFX_RESIDUAL_SURFACE, map->index, P, n_neighbors, d_points, d_begin, d_count, count_stride, n_clouds,
    max_points_per_cloud, d_residual, d_jacobian, nullptr, static_cast<hipStream_t>(stream));
  LFX_HIP(c, hipGetLastError());
  return LFX_OK;
}

int lfx_edge_residuals(
  lfx_ctx * c, const lfx_map * map, const double pose[12], uint32_t n_neighbors, double * d_residual, double * d_jacobian,
  void * stream)
{
  if (!c) {return LFX_ERR_INVALID_ARGUMENT;}
  if (c->last_batch == 0) {return fail(c, LFX_ERR_INVALID_ARGUMENT, "no batch has been extracted yet");}
  uint32_t longest = 0;
  for (uint32_t s = 0; s < c->last_batch; s++) {
    const uint32_t n = c->h_scan_begin[s + 1] - c->h_scan_begin[s];
    longest = n > longest ? n : longest;                 // a scan has no more edge points than points
  }
  return lfx_scan_to_map_residuals(c, LFX_RESIDUAL_EDGE, map, pose, n_neighbors, reinterpret_cast<const float *>(c->edge_pts.p),
           c->scan_begin.p, c->scan_info.p + lfx::kInfoEdge, 4, c->last_batch, longest, d_residual, d_jacobian, stream);
}

}  // extern "C"

// ---------------------------------------------------------------------------- the optimizer around the rows
namespace
{
struct AlignProblem                     // what Problem::Make reads, per kind
{
  // rows of dimension 3: the edge clouds, or the point pairs
  const lfx_map * edge_map = nullptr;
  const float * edge_points = nullptr; const double * X = nullptr, * Y = nullptr;
  const uint32_t * begin3 = nullptr, * count3 = nullptr; uint32_t stride3 = 1, longest3 = 0; size_t total3 = 0;
  // rows of dimension 1: the downsampled surface clouds
  const lfx_map * surface_map = nullptr;
  const float * surface_points = nullptr;
  const uint32_t * begin1 = nullptr, * count1 = nullptr; uint32_t stride1 = 1, longest1 = 0; size_t total1 = 0;
  uint32_t n_neighbors = 0;
};

int run_align(lfx_ctx * c, const AlignProblem & pr, uint32_t n_clouds, int max_iter, const double * initial_poses,
  lfx_align_result * results, hipStream_t st)
{
  static_assert(sizeof(lfx::AlignState) % 8 == 0, "AlignState is an array of doubles' worth");
  if (n_clouds > 65535u) {return fail(c, LFX_ERR_INVALID_ARGUMENT, "at most 65535 scans per alignment call");}   // (a launch's y extent)
  const size_t state_d = sizeof(lfx::AlignState) / 8 * (size_t)n_clouds, pose_d = 12 * (size_t)n_clouds;
  const size_t rows = pr.total3 + pr.total1;
  const size_t partial_d = (size_t)n_clouds * lfx::kAlignSlices * 64;
  const size_t need = state_d + pose_d + 24 * pr.total3 + 8 * pr.total1 + rows + partial_d + (n_clouds + 1) / 2 + 9;
  if (c->align_scratch.n < need) {
    c->align_scratch.release();
    if (c->align_scratch.alloc(need) != hipSuccess) {
      c->align_scratch.n = 0;
      return fail(c, LFX_ERR_OUT_OF_MEMORY, "cannot allocate the rows of the scan-to-map alignment");
    }
  }
  double * w = c->align_scratch.p;
  lfx::AlignState * states = reinterpret_cast<lfx::AlignState *>(w); w += state_d;
  double * d_initial = w; w += pose_d;
  double * r3 = w; w += 3 * pr.total3;
  double * J3 = w; w += 21 * pr.total3;
  double * r1 = w; w += pr.total1;
  double * J1 = w; w += 7 * pr.total1;
  double * d_weights = w; w += rows;
  double * d_partials = w; w += partial_d;
  uint32_t * d_tickets = reinterpret_cast<uint32_t *>(w); w += (n_clouds + 1) / 2;
  uint32_t * d_active = reinterpret_cast<uint32_t *>(w);
  LFX_HIP(c, hipMemsetAsync(d_tickets, 0, sizeof(uint32_t) * n_clouds, st));
  // small copies through pinned memory: [poses | states | active]
  const size_t h_states_at = pose_d * 8, h_active_at = h_states_at + sizeof(lfx::AlignState) * n_clouds;
  LFX_HIP(c, c->h_align.reserve(h_active_at + 16 + 20 * (size_t)n_clouds));
  std::memcpy(c->h_align.p, initial_poses, pose_d * 8);
  LFX_HIP(c, hipMemcpyAsync(d_initial, c->h_align.p, pose_d * 8, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(lfx::align_begin_kernel, dim3((n_clouds + 63u) / 64u), dim3(64), 0, st, states, d_initial, n_clouds, d_active);
  const lfx::MapPose none{};
  for (int iter = 0; iter < max_iter; iter++) {
    if (pr.X) {
      if (pr.longest3) {
        hipLaunchKernelGGL(lfx::pair_rows_kernel, dim3((pr.longest3 + 127u) / 128u, n_clouds), dim3(128), 0, st, pr.X, pr.Y,
          pr.begin3, pr.count3, r3, J3, states);
      }
    } else {
      const bool both_grids = pr.edge_map->index.start && pr.surface_map->index.start;
      // a few scans: edge and surface rows in one launch, side by side (the short surface part otherwise runs after the edge
      // part on a mostly idle chip).  Many scans fill the chip anyway, and the one kernel's register count (the surface
      // rows' QR) would halve the edge searches' occupancy: 64 scans took 13.4 ms that way against 8.4 ms.
      const bool few = (uint64_t)n_clouds * ((uint64_t)pr.longest3 + pr.longest1) <= 32768u;
      if (pr.longest3 && pr.longest1 && both_grids && few) {
        const lfx::RowsOfKind e{pr.edge_map->index, reinterpret_cast<const float4 *>(pr.edge_points), pr.begin3, pr.count3, pr.stride3, r3, J3};
        const lfx::RowsOfKind f{pr.surface_map->index, reinterpret_cast<const float4 *>(pr.surface_points), pr.begin1, pr.count1, pr.stride1,
          r1, J1};
        hipLaunchKernelGGL(lfx::scan_to_map_both_kernel<lfx::kSearchGridWave>, dim3(pr.longest3 + pr.longest1, n_clouds), dim3(64), 0, st,
          e, f, pr.longest3, none, pr.n_neighbors, states);
      } else {
        if (pr.longest3) {
          launch_rows(false, pr.edge_map->index, none, pr.n_neighbors, pr.edge_points, pr.begin3, pr.count3, pr.stride3, n_clouds,
            pr.longest3, r3, J3, states, st);
        }
        if (pr.longest1) {
          launch_rows(true, pr.surface_map->index, none, pr.n_neighbors, pr.surface_points, pr.begin1, pr.count1, pr.stride1, n_clouds,
            pr.longest1, r1, J1, states, st);
        }
      }
    }
    hipLaunchKernelGGL(lfx::align_scale_kernel, dim3(n_clouds), dim3(lfx::kAlignThreads), 0, st, states, iter, r3, pr.begin3, pr.count3,
      pr.stride3, r1, pr.begin1, pr.count1, pr.stride1, d_weights, d_active);
    hipLaunchKernelGGL(lfx::align_update_kernel, dim3(lfx::kAlignSlices, n_clouds), dim3(lfx::kAlignThreads), 0, st, states, iter,
      max_iter, r3, J3, pr.begin3, pr.count3, pr.stride3, r1, J1, pr.begin1, pr.count1, pr.stride1, d_weights, d_partials, d_tickets,
      d_active);
    // the kernels of a finished scan return at once, but a launch is a launch: now and then ask whether any scan still iterates
    if ((iter == 2 || iter == 4 || iter == 7 || iter == 11 || iter == 15) && iter + 1 < max_iter) {
      uint32_t * active = reinterpret_cast<uint32_t *>(c->h_align.p + h_active_at);
      LFX_HIP(c, hipMemcpyAsync(active, d_active, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
      LFX_HIP(c, hipStreamSynchronize(st));
      if (*active == 0) {break;}
    }
  }
  LFX_HIP(c, hipGetLastError());
  const lfx::AlignState * h = reinterpret_cast<const lfx::AlignState *>(c->h_align.p + h_states_at);
  LFX_HIP(c, hipMemcpyAsync(c->h_align.p + h_states_at, states, sizeof(lfx::AlignState) * n_clouds, hipMemcpyDeviceToHost, st));
  LFX_HIP(c, hipStreamSynchronize(st));
  for (uint32_t s = 0; s < n_clouds; s++) {
    for (int i = 0; i < 12; i++) {results[s].pose[i] = h[s].pose.m[i];}
    results[s].error = h[s].error; results[s].error_scale = h[s].scale;
    results[s].iteration = h[s].iteration; results[s].code = h[s].code;
  }
  return LFX_OK;
}
}  // namespace

extern "C" {

const char * lfx_align_message(int code)
{
  switch (code) {                         // the texts of optimization_result.hpp:43-79
    case LFX_ALIGN_CONVERGED: return "Optimization successfully converged";
    case LFX_ALIGN_LARGER_ERROR: return "The error is larger than previous iteration";
    case LFX_ALIGN_LARGER_SCALE: return "The scale is larger than previous iteration";
    case LFX_ALIGN_MAX_ITERATION: return "The iteration reached the maximum value";
    case LFX_ALIGN_EMPTY_INPUT: return "The input data is empty";
    default: return "unknown";
  }
}

int lfx_scan_to_map_align(
  lfx_ctx * c, const lfx_map * edge_map, const lfx_map * surface_map, uint32_t n_neighbors, int max_iter,
  const float * d_edge_points, const uint32_t * d_edge_begin, const uint32_t * d_edge_count, uint32_t edge_count_stride,
  uint32_t max_edge_points_per_cloud, size_t total_edge_points,
  const float * d_surface_points, const uint32_t * d_surface_begin, const uint32_t * d_surface_count,
  uint32_t surface_count_stride, uint32_t max_surface_points_per_cloud, size_t total_surface_points,
  uint32_t n_clouds, const double * initial_poses, lfx_align_result * results, void * stream)
{
  if (!c || !edge_map || !surface_map || !d_edge_points || !d_edge_begin || !d_edge_count || !d_surface_points ||
    !d_surface_begin || !d_surface_count || !initial_poses || !results || n_clouds == 0 || edge_count_stride == 0 ||
    surface_count_stride == 0)
  {
    return LFX_ERR_INVALID_ARGUMENT;
  }
  if (max_iter < 1) {return fail(c, LFX_ERR_INVALID_ARGUMENT, "max_iter must be >= 1");}
  if (n_neighbors < 3 || n_neighbors > (uint32_t)lfx::kNearestMax || edge_map->index.n < n_neighbors || surface_map->index.n < n_neighbors) {
    return fail(c, LFX_ERR_INVALID_ARGUMENT, "n_neighbors must be in [3, 16] and both maps must hold that many points");
  }
  if (edge_map->device != c->device || surface_map->device != c->device) {return fail(c, LFX_ERR_INVALID_ARGUMENT, "a map lives on another device");}
  LFX_HIP(c, hipSetDevice(c->device));
  AlignProblem pr;
  pr.edge_map = edge_map; pr.edge_points = d_edge_points;
  pr.begin3 = d_edge_begin; pr.count3 = d_edge_count; pr.stride3 = edge_count_stride; pr.longest3 = max_edge_points_per_cloud;
  pr.total3 = total_edge_points;
  pr.surface_map = surface_map; pr.surface_points = d_surface_points;
  pr.begin1 = d_surface_begin; pr.count1 = d_surface_count; pr.stride1 = surface_count_stride;
  pr.longest1 = max_surface_points_per_cloud; pr.total1 = total_surface_points;
  pr.n_neighbors = n_neighbors;
  return run_align(c, pr, n_clouds, max_iter, initial_poses, results, static_cast<hipStream_t>(stream));
}

int lfx_align_point_pairs(
  lfx_ctx * c, const double * d_source, const double * d_target, const uint32_t * d_begin, const uint32_t * d_count,
  uint32_t max_points_per_cloud, size_t total_points, uint32_t n_clouds, int max_iter, const double * initial_poses,
  lfx_align_result * results, void * stream)
{
  if (!c || !d_source || !d_target || !d_begin || !d_count || !initial_poses || !results || n_clouds == 0) {return LFX_ERR_INVALID_ARGUMENT;}
  if (max_iter < 1) {return fail(c, LFX_ERR_INVALID_ARGUMENT, "max_iter must be >= 1");}
  LFX_HIP(c, hipSetDevice(c->device));
  AlignProblem pr;
  pr.X = d_source; pr.Y = d_target; pr.begin3 = d_begin; pr.count3 = d_count; pr.stride3 = 1; pr.longest3 = max_points_per_cloud;
  pr.total3 = total_points;
  return run_align(c, pr, n_clouds, max_iter, initial_poses, results, static_cast<hipStream_t>(stream));
}

int lfx_localize_batch(
  lfx_ctx * c, const lfx_map * edge_map, const lfx_map * surface_map, uint32_t n_neighbors, int max_iter, float surface_leaf, const double * initial_poses, lfx_align_result * results, void * stream)
{
  if (!c || !initial_poses || !results) {return LFX_ERR_INVALID_ARGUMENT;}
  if (c->last_batch == 0) {return fail(c, LFX_ERR_INVALID_ARGUMENT, "no batch has been extracted yet");}
  LFX_HIP(c, hipSetDevice(c->device));
  const uint32_t batch = c->last_batch;
  const size_t total = c->h_scan_begin[batch];
  const size_t need = 4 * total + 2 * (size_t)batch;
  if (c->align_surface.n < need) {
    c->align_surface.release();
    if (c->align_surface.alloc(need) != hipSuccess) {
      c->align_surface.n = 0;
      return fail(c, LFX_ERR_OUT_OF_MEMORY, "cannot allocate the downsampled surface clouds");
    }
  }
  float * down = c->align_surface.p;
  uint32_t * down_count = reinterpret_cast<uint32_t *>(down + 4 * total), * down_status = down_count + batch;
  const int rc = lfx_downsample_surface(c, surface_leaf, down, down_count, down_status, stream);
  if (rc != LFX_OK) {return rc;}
  // where PCL gives the cloud back unfiltered (leaf too small for its extent) the rows are built from all surface points
  hipLaunchKernelGGL(lfx::downsample_passthrough_kernel, dim3(batch), dim3(256), 0, static_cast<hipStream_t>(stream),
    reinterpret_cast<const float4 *>(c->surf_pts.p), c->scan_begin.p, c->scan_info.p + lfx::kInfoSurface, 4u,
    reinterpret_cast<float4 *>(down), down_count, down_status);
  // the longest edge cloud and the longest downsampled surface cloud size the launches (and choose between one query per
  // thread and one per wave): two small copies, and the call is synchronous anyway
  hipStream_t st = static_cast<hipStream_t>(stream);
  LFX_HIP(c, c->h_align.reserve(20 * (size_t)batch));
  uint32_t * info = reinterpret_cast<uint32_t *>(c->h_align.p), * down_n = info + 4 * (size_t)batch;
  LFX_HIP(c, hipMemcpyAsync(info, c->scan_info.p, sizeof(uint32_t) * 4 * batch, hipMemcpyDeviceToHost, st));
  LFX_HIP(c, hipMemcpyAsync(down_n, down_count, sizeof(uint32_t) * batch, hipMemcpyDeviceToHost, st));
  LFX_HIP(c, hipStreamSynchronize(st));
  uint32_t longest_edge = 0, longest_surface = 0;
  for (uint32_t s = 0; s < batch; s++) {
    longest_edge = std::max(longest_edge, info[4 * s + lfx::kInfoEdge]);
    longest_surface = std::max(longest_surface, down_n[s]);
  }
  return lfx_scan_to_map_align(c, edge_map, surface_map, n_neighbors, max_iter,
           reinterpret_cast<const float *>(c->edge_pts.p), c->scan_begin.p, c->scan_info.p + lfx::kInfoEdge, 4, longest_edge, total,
           down, c->scan_begin.p, down_count, 1, longest_surface, total, batch, initial_poses, results, stream);
}

int lfx_localize_host(
  lfx_ctx * c, const lfx_map * edge_map, const lfx_map * surface_map, uint32_t n_neighbors, int max_iter, float surface_leaf,
  const float * edge_points, uint32_t n_edge, const float * surface_points, uint32_t n_surface, const double initial_pose[12],
  lfx_align_result * result, void * stream)
{
  if (!c || !edge_map || !surface_map || !initial_pose || !result || (n_edge && !edge_points) || (n_surface && !surface_points)) {
    return LFX_ERR_INVALID_ARGUMENT;
  }
  LFX_HIP(c, hipSetDevice(c->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  // [edge | surface | downsampled surface] records of 4 floats, then begin / count words
  const size_t ne = n_edge, ns = n_surface, words = 8;
  const size_t need = 4 * (ne + 2 * ns + 2) + words;
  if (c->align_surface.n < need) {
    c->align_surface.release();
    if (c->align_surface.alloc(need) != hipSuccess) {
      c->align_surface.n = 0;
      return fail(c, LFX_ERR_OUT_OF_MEMORY, "cannot allocate the scan's clouds");
    }
  }
  float * d_edge = c->align_surface.p, * d_surface = d_edge + 4 * (ne + 1), * d_down = d_surface + 4 * (ns + 1);
  uint32_t * d_words = reinterpret_cast<uint32_t *>(d_down + 4 * ns);
  // words: [0] begin (0), [1] n_edge, [2] n_surface, [3] downsampled count, [4] downsample status
  LFX_HIP(c, c->h_align.reserve(sizeof(uint32_t) * words));
  uint32_t * h_words = reinterpret_cast<uint32_t *>(c->h_align.p);
  h_words[0] = 0; h_words[1] = n_edge; h_words[2] = n_surface; h_words[3] = 0; h_words[4] = 0;
  LFX_HIP(c, hipMemcpyAsync(d_words, h_words, sizeof(uint32_t) * words, hipMemcpyHostToDevice, st));
  if (n_edge) {LFX_HIP(c, hipMemcpyAsync(d_edge, edge_points, sizeof(float) * 4 * ne, hipMemcpyHostToDevice, st));}
  uint32_t n_down = 0;
  if (n_surface) {
    LFX_HIP(c, hipMemcpyAsync(d_surface, surface_points, sizeof(float) * 4 * ns, hipMemcpyHostToDevice, st));
    const int rc = lfx_voxel_downsample(c, d_surface, d_words, d_words + 2, 1, 1, ns, surface_leaf, d_down, d_words + 3, d_words + 4, stream);
    if (rc != LFX_OK) {return rc;}
    hipLaunchKernelGGL(lfx::downsample_passthrough_kernel, dim3(1), dim3(256), 0, st, reinterpret_cast<const float4 *>(d_surface),
      d_words, d_words + 2, 1u, reinterpret_cast<float4 *>(d_down), d_words + 3, d_words + 4);
    LFX_HIP(c, hipMemcpyAsync(h_words + 3, d_words + 3, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    LFX_HIP(c, hipStreamSynchronize(st));
    n_down = h_words[3];
  } else {
    LFX_HIP(c, hipStreamSynchronize(st));             // the words have left the pinned block: the alignment stages through it too
  }
  return lfx_scan_to_map_align(c, edge_map, surface_map, n_neighbors, max_iter, d_edge, d_words, d_words + 1, 1, n_edge, ne,
           d_down, d_words, d_words + 3, 1, n_down, ns, 1, initial_pose, result, stream);
}

}  // extern "C"

// ---------------------------------------------------------------------------- stage entry points
int lfx_stage_ring(
  lfx_ctx * c, const lfx_params * params, uint32_t flags, uint32_t n, const float * x, const float * y,
  const int32_t * groups, const double * curvature_in, const double * range_in, double * range_out,
  double * curvature_out, uint8_t * link_out, uint8_t * labels_out, int32_t * ring_status_out)
{
  if (!c || !x || !y || n == 0) {return LFX_ERR_INVALID_ARGUMENT;}
  const lfx_params * pp = params ? params : &c->params;
  std::string why;
  if (validate_params(pp, why) != LFX_OK) {return fail(c, LFX_ERR_INVALID_ARGUMENT, why);}
  if (n > LFX_MAX_RING_POINTS) {return fail(c, LFX_ERR_CAPACITY, "ring longer than LFX_MAX_RING_POINTS");}
  LFX_HIP(c, hipSetDevice(c->device));
  const lfx::Params dp = device_params(*pp);
  const uint32_t cap = ((n < 64 ? 64 : n) + 63u) & ~63u;
  float * dx = nullptr, * dy = nullptr;
  int32_t * dg = nullptr, * dstat = nullptr;
  double * dci = nullptr, * dri = nullptr, * dr = nullptr, * dc = nullptr;
  uint8_t * dl = nullptr, * dlab = nullptr;
  auto cleanup = [&] {
      (void)hipFree(dx); (void)hipFree(dy); (void)hipFree(dg); (void)hipFree(dstat); (void)hipFree(dci); (void)hipFree(dri);
      (void)hipFree(dr); (void)hipFree(dc); (void)hipFree(dl); (void)hipFree(dlab);
    };
  hipError_t e = hipSuccess;
  auto ok = [&](hipError_t r) {if (e == hipSuccess) {e = r;}};
  ok(hipMalloc(&dx, n * 4)); ok(hipMalloc(&dy, n * 4)); ok(hipMalloc(&dstat, 4));
  ok(hipMalloc(&dr, n * 8)); ok(hipMalloc(&dc, n * 8)); ok(hipMalloc(&dl, n)); ok(hipMalloc(&dlab, n));
  if (groups) {ok(hipMalloc(&dg, n * 4));}
  if (curvature_in) {ok(hipMalloc(&dci, n * 8));}
  if (range_in) {ok(hipMalloc(&dri, n * 8));}
  if (e == hipSuccess) {
    ok(hipMemcpy(dx, x, n * 4, hipMemcpyHostToDevice));
    ok(hipMemcpy(dy, y, n * 4, hipMemcpyHostToDevice));
    if (groups) {ok(hipMemcpy(dg, groups, n * 4, hipMemcpyHostToDevice));}
    if (curvature_in) {ok(hipMemcpy(dci, curvature_in, n * 8, hipMemcpyHostToDevice));}
    if (range_in) {ok(hipMemcpy(dri, range_in, n * 8, hipMemcpyHostToDevice));}
  }
  if (e == hipSuccess) {
    hipLaunchKernelGGL(lfx::ring_stage_kernel, dim3(1), dim3(256), lfx::ring_lds_bytes(cap), c->stream,
      dp, cap, flags, (int)n, dx, dy, dg, dci, dri, dr, dc, dl, dlab, dstat);
    ok(hipGetLastError());
    ok(hipStreamSynchronize(c->stream));
  }
  if (e == hipSuccess) {
    if (range_out) {ok(hipMemcpy(range_out, dr, n * 8, hipMemcpyDeviceToHost));}
    if (curvature_out) {ok(hipMemcpy(curvature_out, dc, n * 8, hipMemcpyDeviceToHost));}
    if (link_out && n > 1) {ok(hipMemcpy(link_out, dl, n - 1, hipMemcpyDeviceToHost));}
    if (labels_out) {ok(hipMemcpy(labels_out, dlab, n, hipMemcpyDeviceToHost));}
    if (ring_status_out) {ok(hipMemcpy(ring_status_out, dstat, 4, hipMemcpyDeviceToHost));}
  }
  cleanup();
  if (e != hipSuccess) {return fail(c, LFX_ERR_HIP, std::string("lfx_stage_ring: ") + hipGetErrorString(e));}
  return LFX_OK;
}

int lfx_stage_convolution1d(lfx_ctx * c, const double * input, uint32_t n, const double * weight, uint32_t m, double * out)
{
  if (!c || !input || !weight || !out || m == 0 || (m % 2) == 0) {return LFX_ERR_INVALID_ARGUMENT;}
  if (n < m) {   // convolution.cpp:39-43 throws std::invalid_argument
    return fail(c, LFX_ERR_INVALID_ARGUMENT, "Input array size " + std::to_string(n) + " cannot be smaller than weight size " + std::to_string(m));
  }
  LFX_HIP(c, hipSetDevice(c->device));
  double * di = nullptr, * dw = nullptr, * dout = nullptr;
  hipError_t e = hipSuccess;
  auto ok = [&](hipError_t r) {if (e == hipSuccess) {e = r;}};
  ok(hipMalloc(&di, n * 8)); ok(hipMalloc(&dw, m * 8)); ok(hipMalloc(&dout, n * 8));
  if (e == hipSuccess) {
    ok(hipMemcpy(di, input, n * 8, hipMemcpyHostToDevice));
    ok(hipMemcpy(dw, weight, m * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(lfx::convolution1d_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream,
      di, (int)n, dw, (int)m, dout);
    ok(hipGetLastError());
    ok(hipStreamSynchronize(c->stream));
    ok(hipMemcpy(out, dout, n * 8, hipMemcpyDeviceToHost));
  }
  (void)hipFree(di); (void)hipFree(dw); (void)hipFree(dout);
  if (e != hipSuccess) {return fail(c, LFX_ERR_HIP, std::string("lfx_stage_convolution1d: ") + hipGetErrorString(e));}
  return LFX_OK;
}

int lfx_stage_ring_projection(
  lfx_ctx * c, const void * points, size_t n_points, uint32_t * sorted_index, uint32_t * n_rings,
  uint16_t * ring_id, uint32_t * ring_count)
{
  if (!c || !sorted_index) {return LFX_ERR_INVALID_ARGUMENT;}
  lfx_scan_result r;
  const void * p[1] = {points};
  const size_t n[1] = {n_points};
  const int rc = extract_batch_impl(c, p, n, 1, LFX_OUT_SORTED_INDEX, &r);
  if (rc != LFX_OK) {return rc;}
  if (r.n_sorted) {std::memcpy(sorted_index, r.sorted_index, (size_t)r.n_sorted * 4);}     // n_sorted <= n_points: the zero filter drops points
  if (n_rings) {*n_rings = r.n_rings;}
  for (uint32_t k = 0; k < r.n_rings; k++) {
    if (ring_id) {ring_id[k] = r.ring_id[k];}
    if (ring_count) {ring_count[k] = r.ring_count[k];}
  }
  return LFX_OK;
}

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

// ---------------------------------------------------------------------------- measurement
int lfx_set_profiling(lfx_ctx * c, int enabled)
{
  if (!c) {return LFX_ERR_INVALID_ARGUMENT;}
  const int rc = drain_spans(c);
  c->profiling = enabled != 0;
  for (int k = 0; k < LFX_N_KERNELS; k++) {c->ms[k] = 0; c->launches[k] = 0;}
  return rc;
}

int lfx_set_profiling_interval(lfx_ctx * c, uint32_t every_n_batches)
{
  if (!c || every_n_batches == 0) {return LFX_ERR_INVALID_ARGUMENT;}
  c->profile_every = every_n_batches;
  c->batch_no = 0;
  return LFX_OK;
}

int lfx_kernel_times(lfx_ctx * c, double ms[LFX_N_KERNELS], uint64_t launches[LFX_N_KERNELS])
{
  if (!c) {return LFX_ERR_INVALID_ARGUMENT;}
  const int rc = drain_spans(c);
  for (int k = 0; k < LFX_N_KERNELS; k++) {
    if (ms) {ms[k] = c->ms[k];}
    if (launches) {launches[k] = c->launches[k];}
  }
  return rc;
}

}  // extern "C"

#ifdef LFX_STAMPS
// Diagnostic build only: copy the unit kernel's stage stamps (see LFX_STAMP) to the host.
extern "C" int lfx_debug_read_stamps(unsigned long long * out, int n)
{
  const int total = lfx::kStampUnits * lfx::kStampSlots;
  if (!out || n < total) {return -total;}
  if (hipDeviceSynchronize() != hipSuccess) {return LFX_ERR_HIP;}
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(lfx::g_unit_stamps), sizeof(unsigned long long) * total) != hipSuccess) {return LFX_ERR_HIP;}
  return total;
}
#endif

