// lfx_api.hip -- host side of the extraction path of the C ABI declared in include/lfx.h: the context, the launches of
// lfx_kernels_extract.hpp on the caller's stream, results, messages, the per-stage mirrors, measurement.  (Wire formats:
// lfx_wire.hip; the RCCL gather: lfx_gather.hip; Downsample: lfx_downsample.hip; the localizer: lfx_localize.hip.)
// There is no CPU implementation of the path here: without a gfx950 device every entry point fails.
#include "lfx_internal.hpp"

#include <algorithm>
#include <cstddef>
#include "lfx_kernels_extract.hpp"

// The switches tests and experiments pin routes and variants with exist in the test-hooks build of the library
// only (liblfx_testhooks.so, -DLFX_TEST_HOOKS): the shipped library reads no environment variable in lfx_create.
#ifdef LFX_TEST_HOOKS
#define LFX_DEBUG_ENV(name) std::getenv("LFX_DEBUG_" name)
#else
#define LFX_DEBUG_ENV(name) static_cast<const char *>(nullptr)
#endif

using namespace lfx_host;

namespace
{

std::string g_create_error;

const char * kKernelNames[LFX_N_KERNELS] = {
  "ring_scatter_kernel", "ring_unit_kernel", "ring_order_kernel", "ring_unit_kernel(second pass)", "ring_extract_kernel",
  "ring_totals_kernel", "feature_compact_kernel", "ring_unit_org_kernel", "ring_cut_kernel", "fallback_tail_kernel", "grid_count_kernel",
  "batch_reset_kernel"};

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

}  // namespace

std::string & lfx_host::create_error() {return g_create_error;}

namespace
{

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

// Which route a batch takes, from what earlier batches reported (RouteState::report = the counters block of the last
// batch whose report has landed) and the state the choices before left behind.  No device, no context: the function the
// table-driven test drives (lfx_route_choice, tests/test_route_choice.py).
//   fused       the organised-scan kernel runs first; the bucketing kernels then take the fall-back list only
//   xform       ... after ring_cut_kernel has found every ring's rotation / reversal
//   fb_grid     list entries the bucketing kernels are launched for (they loop where the list turns out longer)
//   short_tail  bucketing route = bucketing + the workgroup-per-ring kernel (two near-empty launches instead of five)
//   pre_order   order repair ahead of the first unit pass;  redo_cap  rings the second unit pass is launched for
RouteChoice choose_route(RouteState & st, const RoutePins & pin, bool organised_possible, uint32_t batch, uint32_t max_rings)
{
  RouteChoice ch;
  const uint32_t * rep = st.report;
  bool fused = organised_possible;
  ch.fb_grid = batch;
  if (fused) {
    const uint32_t was_fused = rep[lfx::kCntFusedRan], fell = rep[lfx::kCntFallback], of = rep[lfx::kCntBatch];
    const uint32_t order_fell = rep[lfx::kCntOrderFell], cut_ran = rep[lfx::kCntCutRan], turned = rep[lfx::kCntTurned];
    if (was_fused && of) {
      // most of what fell back did so for the angle order of its rings alone: find the rings' transforms first from now
      // on; back to plain loads once (almost) no ring needs one any more
      if (!cut_ran && 4u * order_fell > of && 2u * order_fell > fell) {st.use_xform = true;}
      if (cut_ran && 50u * turned < of * max_rings) {st.use_xform = false;}
    }
    if (was_fused && of) {
      // most of what fell back did so for a (0, 0, 0) record alone (zero filter on): a grid with holes -- count its valid
      // returns first and take it in place; back to the plain form once (almost) no ring group holds such a record
      const uint32_t zero_fell = rep[lfx::kCntZeroFell], holes_ran = rep[lfx::kCntHolesRan], zero_groups = rep[lfx::kCntZeroGroups];
      if (!holes_ran && 4u * zero_fell > of && 2u * zero_fell > fell) {st.use_holes = true;}
      if (holes_ran && 50u * zero_groups < of) {st.use_holes = false;}
    }
    if (pin.xform >= 0) {st.use_xform = pin.xform != 0;}
    if (pin.holes >= 0) {st.use_holes = pin.holes != 0;}
    if (pin.fused >= 0) {
      fused = pin.fused != 0;
    } else {
      if (was_fused && of) {
        // (a report from before the transforms were switched on says nothing about the route with them)
        const bool mostly_not = 4u * fell > of && !(st.use_xform && !cut_ran) && !(st.use_holes && !rep[lfx::kCntHolesRan]);
        if (mostly_not && !st.bucket_all) {st.retry_in = 16;}
        st.bucket_all = mostly_not;
      }
      if (st.bucket_all) {
        fused = false;
        if (st.retry_in == 0 || --st.retry_in == 0) {fused = true; st.retry_in = 16;}     // the stream may have changed
      }
    }
    if (fused) {
      const uint32_t guess = (was_fused ? 2u * fell : 0u) + 8u;
      ch.fb_grid = guess < batch && !st.bucket_all ? guess : batch;      // (a retry on a stream that has been falling back: expect all of it)
      // a stream that has not been falling back: its odd scan out (if one turns up) is redone by the workgroup-per-ring
      // kernel straight from the bucketed arrays -- two near-empty launches per batch instead of five
      ch.short_tail = was_fused && of && fell == 0 && pin.pre_order < 0 && pin.redo_cap == 0;
      if (pin.short_tail >= 0) {ch.short_tail = pin.short_tail != 0;}
    }
  }
  ch.fused = fused;
  ch.xform = fused && st.use_xform;
  ch.holes = fused && st.use_holes && !ch.xform;       // (the holes form takes rings as they stand)
  // more than a twentieth of the rings of an earlier batch needed their order repaired: expect the same now
  ch.pre_order = st.pre_order;
  if (pin.pre_order >= 0) {
    ch.pre_order = pin.pre_order != 0;
  } else if (st.report_rings) {
    ch.pre_order = 20u * (rep[lfx::kCntRedo] + rep[lfx::kCntPreFixed]) > st.report_rings;
  }
  st.pre_order = ch.pre_order;
  // The second pass is launched for as many rings as earlier batches had repaired after their first pass, twice
  // over and at least 256 (a launch that covers every unit of a large batch costs ~20 us to find nothing to do);
  // the order kernel hands what does not fit to the workgroup-per-ring kernel.
  ch.redo_cap = batch * max_rings;
  if (pin.redo_cap) {
    ch.redo_cap = pin.redo_cap;
  } else if (st.report_rings) {
    const uint32_t want = 2u * rep[lfx::kCntRedo] + 256u;
    ch.redo_cap = want < ch.redo_cap ? want : ch.redo_cap;
  }
  return ch;
}

// The sensor's ring ids (lfx_config.ring_ids, lfx_set_ring_ids, or looked up in a scan by the host entry points): slots by
// id ascending -- the order results list rings in.  Ids 0 .. n-1 are the default (no table; the organised route stays
// possible); anything else goes through ring_slot in the bucketing kernel.
int install_ring_ids(lfx_ctx * c, const uint16_t * ids, uint32_t n, bool given)
{
  std::vector<uint16_t> v(ids, ids + n);
  std::sort(v.begin(), v.end());
  v.erase(std::unique(v.begin(), v.end()), v.end());
  if (v.size() > c->max_rings) {
    return fail(c, LFX_ERR_RING_ID, "the scan carries " + std::to_string(v.size()) + " distinct ring ids, the context takes " + std::to_string(c->max_rings) +
                  " (max_rings; at most LFX_MAX_RINGS)");
  }
  bool identity = true;
  for (size_t k = 0; k < v.size(); k++) {identity = identity && v[k] == k;}
  LFX_HIP(c, hipSetDevice(c->device));
  if (c->stream) {LFX_HIP(c, hipStreamSynchronize(c->stream));}
  c->ids_given = given;
  if (identity) {
    c->slot_id.clear();
    c->fused_possible = c->organised_by_config;
    return LFX_OK;
  }
  if (!c->ring_slot.p && c->ring_slot.alloc(65536) != hipSuccess) {
    c->ring_slot.p = nullptr;
    return fail(c, LFX_ERR_OUT_OF_MEMORY, "cannot allocate the ring id table");
  }
  std::vector<uint16_t> table(65536, (uint16_t)0xFFFFu);
  for (size_t k = 0; k < v.size(); k++) {table[v[k]] = (uint16_t)k;}
  LFX_HIP(c, hipMemcpy(c->ring_slot.p, table.data(), table.size() * 2, hipMemcpyHostToDevice));
  c->slot_id = v;
  c->fused_possible = false;               // (the organised-scan kernel reads ring r at column offset r)
  return LFX_OK;
}

// Launch the kernels for `batch` scans whose records lie back to back at d_points.
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
    if (c->fused_possible) {
      // what every unit of the organised-scan kernel would otherwise work out for itself, the same for all 1 536 units of
      // a scan: the columns per ring (an integer division) and its block's two boundaries (two f64 divisions each:
      // index_range.cpp:60-66) -- 6 % of a wave's life.  Same arithmetic here (IEEE double, no contraction).
      c->h_scan_geom.assign((size_t)batch * lfx::kGeomStride, 0u);
      const uint32_t R = c->max_rings;
      const int P = c->dev.P, B = c->dev.B;
      for (uint32_t s = 0; s < batch; s++) {
        const uint32_t n = n_points[s], C = n / R;
        if (C * R != n || C == 0u || C > c->cap) {continue;}               // not R rings x C columns: entry 0 stays 0
        uint32_t * g = c->h_scan_geom.data() + (size_t)s * lfx::kGeomStride;
        g[0] = C;
        for (int j = 0; j <= B && j <= lfx::kUnitMaxBlocks; j++) {g[1 + j] = (uint32_t)lfx::block_boundary((int)C, P, B, j);}
      }
      LFX_HIP(c, hipMemcpyAsync(c->scan_geom.p, c->h_scan_geom.data(), c->h_scan_geom.size() * 4, hipMemcpyHostToDevice, st));
    }
  }
  c->last_batch = batch;
  c->last_points = d_points;
  c->profile_now = c->profiling && (c->batch_no++ % c->profile_every) == 0u;
  // this batch's set of accumulators (lfx_kernels_common.hpp kParityCounters); the other one is zeroed by this batch's
  // compaction for the batch after it
  const uint32_t par = c->parity;
  const size_t tables = (size_t)c->max_batch * lfx::kRings;
  uint32_t * counters = c->counters.p + par * lfx::kParityCounters;
  uint32_t * scan_flags = c->scan_flags.p + (size_t)par * c->max_batch;
  uint32_t * ring_nedge = c->ring_nedge.p + par * tables, * ring_nsurf = c->ring_nsurf.p + par * tables;
  const lfx::UnitTables * unit_tab = c->unit_tab.p + par;
  uint32_t * defer_count = counters + lfx::kCntDefer, * redo_count = counters + lfx::kCntRedo,
    * slow_count = counters + lfx::kCntSlow, * fb_count = counters + lfx::kCntFallback;
  const uint8_t * pts = static_cast<const uint8_t *>(d_points);
  const uint32_t chunks = (longest + lfx::kChunkPoints - 1) / lfx::kChunkPoints;
  const bool canon = c->layout.step == 32 && c->layout.ox == 0 && c->layout.oy == 4 && c->layout.oz == 8 &&
    c->layout.oring == 20 && c->layout.rtype == LFX_FIELD_UINT16 && c->layout.be == 0 &&
    (reinterpret_cast<uintptr_t>(pts) & 15u) == 0;
  // ---- which route: the organised-scan kernel first (scans it cannot take fall back inside this call), or
  //      bucketing for every scan: choose_route() over what earlier batches reported.  A batch's last kernel writes its
  //      report into pinned memory unasked; it is read only once the event behind that kernel has passed (with two scans in
  //      flight the previous batch may still be running), otherwise the report before it stands.
  if (c->h_counters) {
    // (a seqlock read of the pinned block: end serial, the counters, begin serial -- the device writes them the other way round)
    const uint32_t end = __atomic_load_n(c->h_counters + 1 + lfx::kCounters, __ATOMIC_ACQUIRE);
    uint32_t got[lfx::kCounters];
    for (int i = 0; i < lfx::kCounters; i++) {got[i] = __atomic_load_n(c->h_counters + 1 + i, __ATOMIC_RELAXED);}
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    const uint32_t begin = __atomic_load_n(c->h_counters, __ATOMIC_RELAXED);
    if (end != 0u && end == begin && end != c->report_taken) {
      static_assert(sizeof(c->route.report) == sizeof(got), "the report is the counters block");
      std::memcpy(c->route.report, got, sizeof(got));
      c->route.report_rings = got[lfx::kCntBatch] * c->max_rings;
      c->report_taken = end;
    }
  }
  const RouteChoice choice = choose_route(c->route, c->route_pins, c->fused_possible && canon && chunks != 0, batch, c->max_rings);
  const bool fused = choice.fused, short_tail = choice.short_tail;
  const bool holes = choice.holes && c->drop_zero != 0u && c->cum16.p != nullptr;
  const uint32_t fb_grid = choice.fb_grid;             // list entries the bucketing kernels are launched for
  if (chunks == 0) {
    // every scan of the batch is empty: no kernel runs; the result tables say so
    LFX_HIP(c, hipMemsetAsync(c->scan_info.p, 0, (size_t)batch * 16, st));
    LFX_HIP(c, hipMemsetAsync(c->ring_count.p, 0, (size_t)batch * lfx::kRings * 4, st));
    return LFX_OK;
  }
  // The tail of the organised route is ONE launch that leaves the bucketing route's tables as it found them
  // (fallback_tail_kernel): a stream that stays on it needs no reset launch.  Any other route dirties them over its scans;
  // the reset kernel then runs ahead of the next batch, over everything a batch since the last reset may have touched.
  const bool lazy = fused && short_tail && !choice.xform;
  if (!lazy || c->aux_dirty != 0u) {
    const uint32_t n_aux = c->aux_dirty > batch ? c->aux_dirty : batch;
    Timed t(c, 11, st);
    hipLaunchKernelGGL(lfx::batch_reset_kernel, dim3(64), dim3(256), 0, st,
      c->chunk_flags.p, n_aux * c->max_chunks, c->ring_flags.p, n_aux * (uint32_t)lfx::kRings, counters, c->fb_list.p, batch,
      fused ? 0u : 1u, c->xform.p);
    c->aux_dirty = lazy ? 0u : batch;
  }
  c->last_used_xform = fused && choice.xform;
  if (fused) {
    const bool xf = choice.xform;
    if (xf) {
      Timed t(c, 8, st);
      hipLaunchKernelGGL(lfx::ring_cut_kernel, dim3(batch), dim3(lfx::kCutThreads), 0, st,
        pts, c->scan_begin.p, c->max_rings, c->cap, c->xform.p, counters);
    }
    const uint32_t groups = (c->max_rings + 3u) / 4u;
    if (holes) {
      // the count pass of the holes form: valid returns per ring and piece of 16 columns, every ring's length
      Timed t(c, 10, st);
      // (a workgroup per scan, reading its records as they lie, where the batch fills the device with those; else a
      // workgroup per ring group and scan)
      const uint32_t row_words = c->cap / lfx::kPieceCols + 1u;
      const size_t count_lds = ((size_t)c->max_rings * row_words + 2u * groups * (size_t)c->dev.B) * 4u;
      if (batch >= c->scan_count_from && count_lds <= 144u * 1024u) {
        hipLaunchKernelGGL(lfx::scan_count_kernel, dim3(batch), dim3(lfx::kScanCountThreads), count_lds, st,
          pts, c->scan_begin.p, c->scan_geom.p, c->max_rings, lfx::cum_stride(c->cap), c->cum16.p, c->ring_count.p, unit_tab, counters,
          c->hole_desc.p, c->cap, 64u * c->unit_chunks, 4u * (uint32_t)lfx::holes_loads((int)c->unit_chunks), row_words);
      } else {
      hipLaunchKernelGGL(lfx::grid_count_kernel, dim3(groups, batch), dim3(256), 0, st,
        pts, c->scan_begin.p, c->scan_geom.p, c->max_rings, lfx::cum_stride(c->cap), c->cum16.p, c->ring_count.p, unit_tab, counters,
        c->hole_desc.p, c->cap, 64u * c->unit_chunks, 4u * (uint32_t)lfx::holes_loads((int)c->unit_chunks));
      }
    }
    {
      Timed t(c, 7, st);
      const UnitOrgArgs a{c->dev, c->cap, c->unit_flags, c->max_rings, c->drop_zero, pts, c->scan_begin.p, c->ring_count.p, unit_tab,
        c->xform.p, c->scan_geom.p};
      launch_unit_org(c->unit_variant, (int)c->unit_chunks, xf, holes, dim3(groups, (uint32_t)c->dev.B, batch), c->unit_lds_pad, st, a);
    }
  }
  const lfx::RingExtractArgs ex{c->dev, c->cap, c->stage_flags, c->max_rings, pts, c->layout, c->scan_begin.p, c->ring_count.p, c->sxy.p, c->sz.p,
    c->sidx.p, c->label_s.p, (c->outputs & LFX_OUT_CURVATURE) ? c->curv_s.p : nullptr, c->rec_pts.p, c->rec_idx.p, c->ring_status.p,
    c->unit_ne.p, c->unit_ns.p, c->unit_span.p, c->ring_flags.p};
  if (lazy) {
    // ---- the organised route's tail in one launch: the fall-back list is empty as a rule
    Timed t(c, 9, st);
    const lfx::ScatterArgs sc{pts, c->layout, c->scan_begin.p, c->chunk_base.p, c->chunk_flags.p, c->ring_count.p, c->scan_info.p, scan_flags,
      c->sxy.p, c->sz.p, c->sidx.p, c->max_chunks, c->max_rings, c->cap, c->drop_zero};
    const uint32_t turns = (batch + lfx::kTailMaxTurns - 1u) / lfx::kTailMaxTurns;
    // (two rows of workgroups, not the eight the list routes guess: this route is only taken while nothing has been falling
    // back, the rows walk a list that turns out longer, and 1 024 workgroups of 80 KB of LDS that read one word and leave
    // were 6 us of a 98 us step at 128 x 2048 x 32)
    const uint32_t rows = fb_grid < 2u ? fb_grid : 2u;
    const dim3 grid(chunks, rows > turns ? rows : turns);
    if (canon) {
      hipLaunchKernelGGL(lfx::fallback_tail_kernel<true>, grid, dim3(lfx::kChunkThreads), c->ring_lds, st, sc, ex, fb_count, c->fb_list.p, c->tail_ticket.p);
    } else {
      hipLaunchKernelGGL(lfx::fallback_tail_kernel<false>, grid, dim3(lfx::kChunkThreads), c->ring_lds, st, sc, ex, fb_count, c->fb_list.p, c->tail_ticket.p);
    }
  } else {
  // ---- the bucketing route, over the scans on the fall-back list
  {
    Timed t(c, 0, st);
    auto kern = &lfx::ring_scatter_kernel<false>;
    if (canon) {kern = &lfx::ring_scatter_kernel<true>;}
    if (fb_grid == batch) {                            // a row per scan: the form without the loop over list entries
      kern = canon ? &lfx::ring_scatter_kernel<true, true> : &lfx::ring_scatter_kernel<false, true>;
    }
    hipLaunchKernelGGL(kern, dim3(chunks, fb_grid), dim3(lfx::kChunkThreads), 0, st,
      pts, c->layout, c->scan_begin.p, c->chunk_base.p, c->chunk_flags.p, c->ring_count.p, c->scan_info.p, scan_flags,
      c->sxy.p, c->sz.p, c->sidx.p, c->max_chunks, c->max_rings, c->cap, c->drop_zero, fb_count, c->fb_list.p,
      c->slot_id.empty() ? nullptr : c->ring_slot.p);
  }
  // the near-empty launches of the bucketing route are kept small while the organised-scan kernel takes the stream
  const uint32_t list_grid = fused ? (c->slow_grid < 4u * fb_grid ? c->slow_grid : 4u * fb_grid) : c->slow_grid;
  if (c->fast_path && !short_tail) {
    c->pre_order = choice.pre_order;
    if (c->pre_order) {
      Timed t(c, 2, st);
      hipLaunchKernelGGL(lfx::ring_order_kernel, dim3(fused ? list_grid : 4 * c->slow_grid), dim3(512), c->order_lds, st,
        c->cap, c->max_rings, c->ring_count.p, c->sxy.p, c->sz.p, c->sidx.p, c->ring_flags.p, defer_count,
        c->defer_list.p, redo_count, c->redo_list.p, slow_count, c->slow_list.p, 1u, counters + lfx::kCntPreFixed, 0u,
        fb_count, c->fb_list.p, 0u);
    }
    {
      Timed t(c, 1, st);
      const uint32_t units = c->max_rings * (uint32_t)c->dev.B;
      // the looping form only where the list's length is a guess (behind the organised-scan kernel)
      const UnitArgs a{c->dev, c->cap, c->unit_flags, c->max_rings, c->ring_count.p, c->sxy.p, c->sz.p, c->sidx.p, unit_tab,
        defer_count, c->defer_list.p, fb_count, c->fb_list.p, 0u};
      launch_unit(c->unit_variant, false, (int)c->unit_chunks, fused, dim3((units + lfx::kUnitWaves - 1) / lfx::kUnitWaves, fb_grid),
        c->unit_lds_pad, st, a);
    }
    const uint32_t redo_cap = choice.redo_cap;
    {
      // rings out of angle order: repaired in place, then a second pass of the unit kernel over them
      Timed t(c, 2, st);
      hipLaunchKernelGGL(lfx::ring_order_kernel, dim3(list_grid), dim3(512), c->order_lds, st,
        c->cap, c->max_rings, c->ring_count.p, c->sxy.p, c->sz.p, c->sidx.p, c->ring_flags.p, defer_count,
        c->defer_list.p, redo_count, c->redo_list.p, slow_count, c->slow_list.p, 0u, counters + lfx::kCntPreFixed, redo_cap,
        fb_count, c->fb_list.p, 0xFFFFFFFFu /* the first unit pass covers the whole list (it loops where it has to) */);
    }
    {
      Timed t(c, 3, st);
      const uint32_t units = redo_cap * (uint32_t)c->dev.B;
      const UnitArgs a{c->dev, c->cap, c->unit_flags, c->max_rings, c->ring_count.p, c->sxy.p, c->sz.p, c->sidx.p, unit_tab,
        slow_count, c->slow_list.p, redo_count, c->redo_list.p, redo_cap};
      launch_unit(c->unit_variant, true, (int)c->unit_chunks, false, dim3((units + lfx::kUnitWaves - 1) / lfx::kUnitWaves), c->unit_lds_pad, st, a);
    }
  }
  {
    Timed t(c, 4, st);
    const dim3 grid = c->fast_path ? dim3(list_grid) : dim3(c->max_rings, batch);
    hipLaunchKernelGGL(lfx::ring_extract_kernel, grid, dim3(c->ring_threads), c->ring_lds, st,
      ex, short_tail ? 2u : (c->fast_path ? 1u : 0u), short_tail ? fb_count : slow_count, short_tail ? c->fb_list.p : c->slow_list.p);
  }
  }
  {
    // compaction: every scan of the batch, whoever labelled it.  Small batches: the compaction kernel finds every ring's
    // place itself (a launch costs more than the sums it saves; at 65 536 rings the same was measured at +75 us)
    const uint32_t n_units = c->fast_path ? (uint32_t)c->dev.B : 1u;
    // On the organised route the rings' totals are there already (its units added them up): no totals launch whatever the
    // batch; a scan that route gave up is summed from its unit tables inside the compaction kernel.
    const bool self_totals = (fused || (uint64_t)batch * c->max_rings <= 8192u) && c->totals_env != 1;
    if (!self_totals) {
      Timed t(c, 5, st);
      hipLaunchKernelGGL(lfx::ring_totals_kernel, dim3(batch), dim3(lfx::kRings), 0, st,
        c->scan_info.p, c->ring_count.p, c->unit_ne.p, c->unit_ns.p, ring_nedge, ring_nsurf,
        c->ring_ebase.p, c->ring_sbase.p, n_units, c->max_rings);
    }
    c->batch_serial = c->batch_serial + 1u == 0u ? 1u : c->batch_serial + 1u;
    {
      Timed t(c, 6, st);
      hipLaunchKernelGGL(lfx::feature_compact_kernel, dim3((c->max_rings + 3) / 4, batch), dim3(256), 0, st,
        n_units, c->cap, c->scan_begin.p, c->ring_count.p, self_totals ? nullptr : c->ring_ebase.p, c->ring_sbase.p, c->unit_ne.p,
        c->unit_ns.p, c->unit_span.p, c->rec_pts.p, c->rec_idx.p, c->edge_pts.p, c->edge_idx.p, c->surf_pts.p,
        c->surf_idx.p, c->max_rings, c->scan_info.p, counters, c->h_counters, c->batch_serial, ring_nedge, ring_nsurf, c->rec32.p, c->slot_places,
        scan_flags, c->ring_count.p, batch, fused ? 1u : 0u,
        c->counters.p + (par ^ 1u) * lfx::kParityCounters, c->scan_flags.p + (size_t)(par ^ 1u) * c->max_batch,
        c->ring_nedge.p + (par ^ 1u) * tables, c->ring_nsurf.p + (par ^ 1u) * tables, c->par_dirty[par ^ 1u]);
    }
    // (this batch's set is dirty over its scans from here on; the other one is clean once the compaction has run)
    c->par_dirty[par ^ 1u] = 0u;
    c->par_dirty[par] = batch;
    c->parity = par ^ 1u;
  }
  LFX_HIP(c, hipGetLastError());
  return LFX_OK;
}

size_t align16(size_t v) {return (v + 15u) & ~(size_t)15u;}

// Results of scans [first, first + count) of the last batch to pinned host memory, in two halves: fetch_queue puts
// everything on `st` (the pack kernel writes headers and clouds straight into the pinned block; labels / curvature /
// sorted_index, when asked for, are un-permuted on the device and copied); fetch_finish, once the stream has got there,
// reads the headers and fills the caller's structures.  The synchronous entry points run one after the other around ONE
// wait; the pipelined ones (lfx_extract_submit / _wait) keep a plan per scan in flight.
struct FetchPlan
{
  uint32_t first = 0, count = 0, p0 = 0, mask = 0;
  size_t o_hdr = 0, o_ep = 0, o_sp = 0, o_cv = 0, o_ei = 0, o_si = 0, o_sx = 0, o_lb = 0;
  uint8_t * H = nullptr;
  std::vector<uint32_t> begin;        // scan_begin[first .. first + count] as the batch had it
};

int fetch_queue(lfx_ctx * c, uint32_t first, uint32_t count, hipStream_t st, uint32_t mask, PinnedBuf & block, FetchPlan & plan)
{
  if (count == 0 || first + count > c->last_batch) {return fail(c, LFX_ERR_INVALID_ARGUMENT, "scan index outside the last batch");}
  const uint32_t p0 = c->h_scan_begin[first];
  const size_t P = c->h_scan_begin[first + count] - p0;
  mask &= c->outputs | ~(uint32_t)LFX_OUT_CURVATURE;        // (a context created without the per-point curvature has none to fetch)
  const bool want_lab = mask & LFX_OUT_LABELS, want_curv = mask & LFX_OUT_CURVATURE, want_sidx = mask & LFX_OUT_SORTED_INDEX;
  // pinned block: headers | edge_pts | surf_pts | curvature | edge_idx | surf_idx | sorted_index | labels
  const size_t o_hdr = 0, o_ep = align16((size_t)count * lfx::kResultHeaderBytes), o_sp = o_ep + P * 16, o_cv = o_sp + P * 16,
    o_ei = o_cv + (want_curv ? P * 8 : 0), o_si = o_ei + P * 4, o_sx = o_si + P * 4, o_lb = o_sx + (want_sidx ? P * 4 : 0),
    total = o_lb + (want_lab ? P : 0) + 16;
  if (block.reserve(total) != hipSuccess) {return fail(c, LFX_ERR_OUT_OF_MEMORY, "cannot allocate the pinned result block");}
  uint8_t * H = block.p;
  plan.first = first; plan.count = count; plan.p0 = p0; plan.mask = mask; plan.H = H;
  plan.o_hdr = o_hdr; plan.o_ep = o_ep; plan.o_sp = o_sp; plan.o_cv = o_cv; plan.o_ei = o_ei; plan.o_si = o_si; plan.o_sx = o_sx; plan.o_lb = o_lb;
  plan.begin.assign(c->h_scan_begin.begin() + first, c->h_scan_begin.begin() + first + count + 1);
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
      first, p0, c->scan_begin.p, c->max_rings, c->cap, c->ring_count.p, c->label_s.p, want_curv ? c->curv_s.p : nullptr, c->sidx.p, c->d_label.p,
      c->d_curv.p, c->d_sidx.p, c->scan_info.p, c->xform.p);
    LFX_HIP(c, hipGetLastError());
    if (want_lab) {LFX_HIP(c, hipMemcpyAsync(H + o_lb, c->d_label.p, P, hipMemcpyDeviceToHost, st));}
    if (want_curv) {LFX_HIP(c, hipMemcpyAsync(H + o_cv, c->d_curv.p, P * 8, hipMemcpyDeviceToHost, st));}
    if (want_sidx) {LFX_HIP(c, hipMemcpyAsync(H + o_sx, c->d_sidx.p, P * 4, hipMemcpyDeviceToHost, st));}
  }
  return LFX_OK;
}

// (hosts: one HostScan per scan of the plan; the ring lists the results point into live there)
int fetch_finish(lfx_ctx * c, const FetchPlan & plan, HostScan * hosts, lfx_scan_result * out)
{
  uint8_t * H = plan.H;
  const bool want_lab = plan.mask & LFX_OUT_LABELS, want_curv = plan.mask & LFX_OUT_CURVATURE, want_sidx = plan.mask & LFX_OUT_SORTED_INDEX;
  for (uint32_t k = 0; k < plan.count; k++) {
    const uint32_t b = plan.begin[k] - plan.p0, n = plan.begin[k + 1] - plan.begin[k];
    const uint32_t * hdr = reinterpret_cast<const uint32_t *>(H + plan.o_hdr + (size_t)k * lfx::kResultHeaderBytes);
    const uint32_t * rcount = hdr + 4;
    const uint8_t * rstat = reinterpret_cast<const uint8_t *>(hdr + 4 + lfx::kRings);
    if (hdr[lfx::kInfoError] & lfx::kErrRingId) {
      return fail(c, LFX_ERR_RING_ID, "a point carries a ring id the context does not know (max_rings, lfx_config.ring_ids, lfx_set_ring_ids)");
    }
    if (hdr[lfx::kInfoError] & lfx::kErrTimeout) {
      return fail(c, LFX_ERR_HIP, "ring bucketing timed out waiting for an earlier chunk (the workgroups of a scan were not dispatched in index order)");
    }
    HostScan & h = hosts[k];
    h.ring_id.clear(); h.ring_count.clear(); h.ring_offset.clear(); h.ring_status.clear();
    uint32_t dense = 0, nr = 0;
    for (uint32_t r = 0; r < c->max_rings; r++) {
      if (rcount[r] == 0) {continue;}
      h.ring_id.push_back(c->slot_id.empty() ? (uint16_t)r : c->slot_id[r]);
      h.ring_count.push_back(rcount[r]);
      h.ring_offset.push_back(dense);
      h.ring_status.push_back(rstat[r]);
      dense += rcount[r];
      nr++;
    }
    if (dense > n || (dense != n && !c->drop_zero)) {
      return fail(c, LFX_ERR_HIP, "internal: ring counts do not add up to the scan");
    }
    if (c->log_cb) {
      // what the node logs per abandoned ring (feature_extraction.cpp:154-156)
      for (uint32_t r = 0; r < nr; r++) {
        const int stat = h.ring_status[r];
        if (stat == LFX_RING_OK || stat == LFX_RING_SPARSE) {continue;}
        char text[160];
        if (stat == LFX_RING_TOO_LARGE) {
          std::snprintf(text, sizeof(text), "ring %u holds %u points, more than max_points_per_ring (%u)", (unsigned)h.ring_id[r], h.ring_count[r], c->cap);
        } else {
          lfx_ring_message(stat, h.ring_count[r], &c->params, text, sizeof(text));
        }
        c->log_cb(LFX_LOG_WARN, text, c->log_user);
      }
    }
    lfx_scan_result & o = out[k];
    o.n_points = n;
    o.n_sorted = dense;             // = n less the points the zero filter dropped
    o.labels = want_lab ? H + plan.o_lb + b : nullptr;
    o.curvature = want_curv ? reinterpret_cast<const double *>(H + plan.o_cv) + b : nullptr;
    o.sorted_index = want_sidx ? reinterpret_cast<const uint32_t *>(H + plan.o_sx) + b : nullptr;
    o.n_rings = nr;
    o.ring_id = h.ring_id.data();
    o.ring_count = h.ring_count.data();
    o.ring_offset = h.ring_offset.data();
    o.ring_status = h.ring_status.data();
    o.n_edge = hdr[lfx::kInfoEdge];
    o.edge_points = reinterpret_cast<const float *>(H + plan.o_ep) + (size_t)b * 4;
    o.edge_index = reinterpret_cast<const uint32_t *>(H + plan.o_ei) + b;
    o.n_surface = hdr[lfx::kInfoSurface];
    o.surface_points = reinterpret_cast<const float *>(H + plan.o_sp) + (size_t)b * 4;
    o.surface_index = reinterpret_cast<const uint32_t *>(H + plan.o_si) + b;
  }
  return LFX_OK;
}

int fetch(lfx_ctx * c, uint32_t first, uint32_t count, hipStream_t st, uint32_t mask, lfx_scan_result * out)
{
  FetchPlan plan;
  const int rc = fetch_queue(c, first, count, st, mask, c->h_out, plan);
  if (rc != LFX_OK) {return rc;}
  LFX_HIP(c, hipStreamSynchronize(st));
  if (c->host.size() < c->last_batch) {c->host.resize(c->last_batch);}
  return fetch_finish(c, plan, c->host.data() + first, out);
}

// Copy one scan's records to a device buffer on `st`: a pinned buffer (lfx_host_alloc, or anything the caller registered
// with HIP) goes by DMA as it stands; pageable memory is copied through `stage` (pinned) in pieces, each piece's DMA
// running while the next is being copied.
int upload_scan(lfx_ctx * c, uint8_t * d_dst, const void * points, size_t bytes, PinnedBuf & stage, size_t stage_at, hipStream_t st)
{
  if (bytes == 0) {return LFX_OK;}
  hipPointerAttribute_t attr;
  const bool pinned = hipPointerGetAttributes(&attr, points) == hipSuccess && attr.type == hipMemoryTypeHost;
  if (!pinned) {(void)hipGetLastError();}              // a plain pointer is "invalid value" to the query: not an error here
  if (pinned) {
    LFX_HIP(c, hipMemcpyAsync(d_dst, points, bytes, hipMemcpyHostToDevice, st));
    return LFX_OK;
  }
  constexpr size_t kPiece = 1u << 20;
  const uint8_t * src = static_cast<const uint8_t *>(points);
  for (size_t o = 0; o < bytes; o += kPiece) {
    const size_t len = bytes - o < kPiece ? bytes - o : kPiece;
    std::memcpy(stage.p + stage_at + o, src + o, len);
    LFX_HIP(c, hipMemcpyAsync(d_dst + o, stage.p + stage_at + o, len, hipMemcpyHostToDevice, st));
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

int lfx_set_log_callback(lfx_ctx * c, lfx_log_fn cb, void * user)
{
  if (!c) {return LFX_ERR_INVALID_ARGUMENT;}
  c->log_cb = cb;
  c->log_user = user;
  return LFX_OK;
}

int lfx_route_choice(const uint32_t report[LFX_ROUTE_REPORT_WORDS], uint32_t report_rings, uint32_t state[LFX_ROUTE_STATE_WORDS],
                     int organised_possible, uint32_t batch, uint32_t max_rings, uint32_t choice[LFX_ROUTE_CHOICE_WORDS])
{
  static_assert(LFX_ROUTE_REPORT_WORDS == lfx::kCounters, "the report is the counters block");
  if (!report || !state || !choice) {return LFX_ERR_INVALID_ARGUMENT;}
  RouteState st;
  std::memcpy(st.report, report, sizeof(st.report));
  st.report_rings = report_rings;
  st.use_xform = state[0] != 0; st.bucket_all = state[1] != 0; st.retry_in = state[2]; st.pre_order = state[3] != 0; st.use_holes = state[4] != 0;
  const RouteChoice ch = choose_route(st, RoutePins(), organised_possible != 0, batch, max_rings);
  state[0] = st.use_xform; state[1] = st.bucket_all; state[2] = st.retry_in; state[3] = st.pre_order; state[4] = st.use_holes;
  choice[0] = ch.fused; choice[1] = ch.xform; choice[2] = ch.fb_grid; choice[3] = ch.short_tail; choice[4] = ch.pre_order;
  choice[5] = ch.redo_cap; choice[6] = ch.holes;
  return LFX_OK;
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

int lfx_create(lfx_ctx ** out, int device_id, const lfx_params * params, const lfx_config * caller_config)
{
  if (!out) {return LFX_ERR_INVALID_ARGUMENT;}
  *out = nullptr;
  std::string why;
  if (validate_params(params, why) != LFX_OK) {g_create_error = why; return LFX_ERR_INVALID_ARGUMENT;}
  // the caller's struct may be shorter (built against an older header) or longer (a newer one) than this library's: only
  // the bytes both know are read, the rest of ours are zero
  if (!caller_config || caller_config->struct_size < offsetof(lfx_config, max_batch) + sizeof(uint32_t)) {
    g_create_error = "config->struct_size must be sizeof(lfx_config)";
    return LFX_ERR_INVALID_ARGUMENT;
  }
  lfx_config own{};
  std::memcpy(&own, caller_config, caller_config->struct_size < sizeof(own) ? caller_config->struct_size : sizeof(own));
  const lfx_config * config = &own;
  if (config->max_points_per_scan == 0 || config->max_batch == 0) {
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
    const bool defaults = params->padding == 5 && params->distance_diff_threshold == d.distance_diff_threshold &&
      params->parallel_beam_min_range_ratio == d.parallel_beam_min_range_ratio && params->edge_threshold == d.edge_threshold &&
      params->surface_threshold == d.surface_threshold && params->min_range == d.min_range && params->max_range == d.max_range &&
      LFX_DEBUG_ENV("GENERIC_THRESHOLDS") == nullptr;
    c->unit_variant = defaults ? 0 : (params->padding == 5 ? 1 : (params->padding == 2 ? 2 : 3));
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
  // what the caller knows about its stream: the state the route selection would otherwise reach after a batch or two
  if (config->stream_hint == LFX_STREAM_TURNED_RINGS) {c->route.use_xform = true;}
  if (config->stream_hint == LFX_STREAM_NO_GRID) {c->route.bucket_all = true; c->route.retry_in = 16;}
  if (config->stream_hint == LFX_STREAM_GRID_WITH_HOLES) {c->route.use_holes = true;}
  if (config->stream_hint > LFX_STREAM_GRID_WITH_HOLES) {
    g_create_error = "stream_hint must be LFX_STREAM_UNKNOWN, LFX_STREAM_TURNED_RINGS, LFX_STREAM_NO_GRID or LFX_STREAM_GRID_WITH_HOLES";
    delete c;
    return LFX_ERR_INVALID_ARGUMENT;
  }
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
    // (3 .. 6 chunks, or the long form for anything above -- blocks of up to 768 positions; longer ones are the
    // workgroup-per-ring kernel's)
    c->unit_chunks = (uint32_t)(ch < 3 ? 3 : (ch > 6 ? lfx::kUnitMaxChunks : ch));
    if (const char * dbg = LFX_DEBUG_ENV("UNIT_CHUNKS")) {c->unit_chunks = (uint32_t)std::atoi(dbg);}
    if (c->unit_chunks < 3 || c->unit_chunks > 6) {c->unit_chunks = (uint32_t)lfx::kUnitMaxChunks;}
  }
  if (const char * dbg = LFX_DEBUG_ENV("RING_FLAGS")) {c->stage_flags = (uint32_t)std::atoi(dbg);}
  // (a padding beyond the kernels' 32-position windows: the workgroup-per-ring kernel for every ring, lfx_kernels_extract.hpp label_pass_wide)
  c->fast_path = c->dev.B <= lfx::kUnitMaxBlocks && c->dev.P <= lfx::kWindowPadding && LFX_DEBUG_ENV("NO_FAST_PATH") == nullptr;
  // the organised-scan kernel needs to know the sensor's ring count (max_rings given) and reads PointXYZIR records
  c->fused_possible = c->fast_path && config->max_rings != 0 && c->max_points < (1u << 27) && c->layout.step == 32 && c->layout.ox == 0 &&
    c->layout.oy == 4 && c->layout.oz == 8 && c->layout.oring == 20 && c->layout.rtype == LFX_FIELD_UINT16 && c->layout.be == 0;
  c->organised_by_config = c->fused_possible;
  if (const char * dbg = LFX_DEBUG_ENV("FUSED")) {c->route_pins.fused = std::atoi(dbg) != 0 ? 1 : 0;}
  if (const char * dbg = LFX_DEBUG_ENV("TOTALS_KERNEL")) {c->totals_env = std::atoi(dbg) != 0 ? 1 : 0;}
  if (const char * dbg = LFX_DEBUG_ENV("SHORT_TAIL")) {c->route_pins.short_tail = std::atoi(dbg) != 0 ? 1 : 0;}
  if (const char * dbg = LFX_DEBUG_ENV("XFORM")) {c->route_pins.xform = std::atoi(dbg) != 0 ? 1 : 0;}
  if (const char * dbg = LFX_DEBUG_ENV("HOLES")) {c->route_pins.holes = std::atoi(dbg) != 0 ? 1 : 0;}
  if (const char * dbg = LFX_DEBUG_ENV("SCAN_COUNT_FROM")) {c->scan_count_from = (uint32_t)std::atoi(dbg);}      // batch from which the count pass is one workgroup per scan
  c->slow_grid = 1024;
  if (const char * dbg = LFX_DEBUG_ENV("REDO_CAP")) {c->route_pins.redo_cap = (uint32_t)std::atoi(dbg);}
  if (const char * dbg = LFX_DEBUG_ENV("PRE_ORDER")) {c->route_pins.pre_order = std::atoi(dbg) != 0 ? 1 : 0;}
  if (const char * dbg = LFX_DEBUG_ENV("UNIT_FLAGS")) {c->unit_flags = (uint32_t)std::atoi(dbg);}
  if (const char * dbg = LFX_DEBUG_ENV("UNIT_LDS_PAD")) {c->unit_lds_pad = (uint32_t)std::atoi(dbg);}
  if (const char * dbg = LFX_DEBUG_ENV("RING_THREADS")) {c->ring_threads = (uint32_t)std::atoi(dbg);}
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
  ok(c->scan_begin.alloc(nb + 1)); ok(c->scan_info.alloc(nb * 4)); ok(c->scan_geom.alloc(nb * (size_t)lfx::kGeomStride));
  ok(c->chunk_base.alloc(chunk_tab));
  ok(c->ring_count.alloc(tables)); ok(c->chunk_flags.alloc(nb * c->max_chunks));
  // (the batch's accumulators exist twice, lfx_kernels_common.hpp kParityCounters)
  ok(c->ring_status.alloc(tables)); ok(c->ring_nedge.alloc(2 * tables));
  ok(c->ring_nsurf.alloc(2 * tables)); ok(c->ring_ebase.alloc(tables)); ok(c->ring_sbase.alloc(tables));
  ok(c->counters.alloc(2 * lfx::kParityCounters)); ok(c->scan_flags.alloc(2 * nb)); ok(c->tail_ticket.alloc(nb));
  ok(c->ring_flags.alloc(tables)); ok(c->slow_list.alloc(tables)); ok(c->defer_list.alloc(tables));
  ok(c->fb_list.alloc(nb)); ok(c->xform.alloc(tables));
  ok(c->redo_list.alloc(tables));
  ok(c->unit_ne.alloc(tables * lfx::kUnitMaxBlocks)); ok(c->unit_ns.alloc(tables * lfx::kUnitMaxBlocks));
  ok(c->unit_span.alloc(tables * lfx::kUnitMaxBlocks));
  const size_t rc = nb * c->max_rings * c->cap;      // ring-major arrays: fixed capacity per ring id
  ok(c->sxy.alloc(rc)); ok(c->sz.alloc(rc)); ok(c->sidx.alloc(rc)); ok(c->rec_pts.alloc(rc)); ok(c->rec_idx.alloc(rc));
  ok(c->label_s.alloc(rc));
  // (a context created without LFX_OUT_CURVATURE has no per-point curvature array at all: 8 bytes per ring position)
  if (c->outputs & LFX_OUT_CURVATURE) {ok(c->curv_s.alloc(rc));}
  ok(c->d_label.alloc(c->max_points)); ok(c->d_curv.alloc(c->max_points)); ok(c->d_sidx.alloc(c->max_points));
  ok(c->edge_pts.alloc(tc)); ok(c->surf_pts.alloc(tc)); ok(c->edge_idx.alloc(tc)); ok(c->surf_idx.alloc(tc));
  ok(c->unit_tab.alloc(2));
  if (c->drop_zero && c->fused_possible) {
    // the tables of the holes form of the organised route (grid_count_kernel): prefix rows and unit descriptors
    ok(c->cum16.alloc(nb * c->max_rings * lfx::cum_stride(c->cap) + 1024u));
    ok(c->hole_desc.alloc(nb * c->max_rings * (size_t)c->dev.B));
  }
  if (c->fast_path) {
    // the unit kernels' record slots: 20 bytes per place (a point and its index), 64 or 128 places per unit, units back to back
    const size_t slots = nb * c->max_rings * (size_t)c->dev.B;
    c->slot_places = lfx::rec_slot_places(kUnitVariantPadding[c->unit_variant & 3], (int)c->unit_chunks);
    const size_t slot_bytes = (size_t)c->slot_places * lfx::kRecBytes;
    if (slots * slot_bytes <= ((size_t)16 << 30)) {
      ok(c->rec32.alloc(slots * (slot_bytes / 16u)));
    } else {
      c->fast_path = false;                // (hundreds of blocks per ring on a large batch without the sensor's ring count:
      c->fused_possible = false;           // the workgroup-per-ring kernel takes every ring)
    }
  }
  if (e == hipSuccess) {
    e = hipHostMalloc(reinterpret_cast<void **>(&c->h_counters), 4 * (lfx::kCounters + 2), hipHostMallocDefault);
    if (e == hipSuccess) {std::memset(c->h_counters, 0, 4 * (lfx::kCounters + 2));}
  }
  if (e == hipSuccess) {
    // everything a batch's kernels add to or OR into starts clean (and is left clean by the batch before, run_batch); the
    // tables every route writes before anyone reads them start at zero for the readers of a context that has run nothing
    ok(hipMemset(c->counters.p, 0, 2 * lfx::kParityCounters * 4)); ok(hipMemset(c->scan_flags.p, 0, 2 * nb * 4));
    ok(hipMemset(c->ring_nedge.p, 0, 2 * tables * 4)); ok(hipMemset(c->ring_nsurf.p, 0, 2 * tables * 4));
    ok(hipMemset(c->tail_ticket.p, 0, nb * 4)); ok(hipMemset(c->chunk_flags.p, 0, nb * c->max_chunks * 4));
    ok(hipMemset(c->ring_flags.p, 0, tables * 4)); ok(hipMemset(c->xform.p, 0, tables * 4));
    ok(hipMemset(c->scan_info.p, 0, nb * 16)); ok(hipMemset(c->ring_count.p, 0, tables * 4));
  }
  if (e == hipSuccess) {
    // (no per-point curvature asked for: the kernels find no array to write it to -- a fifth of the unit kernel's HBM traffic)
    for (uint32_t par = 0; par < 2u && e == hipSuccess; par++) {
      lfx::UnitTables t{};
      t.label_s = c->label_s.p; t.curv_s = (c->outputs & LFX_OUT_CURVATURE) ? c->curv_s.p : nullptr;
      t.rec_pts = c->rec_pts.p; t.rec_idx = c->rec_idx.p; t.ring_status = c->ring_status.p;
      t.unit_ne = c->unit_ne.p; t.unit_ns = c->unit_ns.p; t.unit_span = c->unit_span.p; t.ring_flags = c->ring_flags.p;
      t.scan_info = c->scan_info.p;
      t.fb_count = c->counters.p + par * lfx::kParityCounters + lfx::kCntFallback;
      t.fb_list = c->fb_list.p;
      t.scan_flags = c->scan_flags.p + (size_t)par * nb;
      t.sidx = c->sidx.p; t.cum16 = c->cum16.p; t.hole_desc = c->hole_desc.p;
      t.ring_nedge = c->ring_nedge.p + par * tables; t.ring_nsurf = c->ring_nsurf.p + par * tables;
      t.rec32 = c->rec32.p; t.prm = c->dev;
      e = hipMemcpy(c->unit_tab.p + par, &t, sizeof(t), hipMemcpyHostToDevice);
    }
  }
  if (e == hipSuccess) {e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);}
  // (the attribute is per function, not per context: always the worst case, so that contexts of different ring
  // capacities can live side by side)
  if (e == hipSuccess) {
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(lfx::ring_extract_kernel),
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lfx::ring_lds_bytes(LFX_MAX_RING_POINTS));
  }
  if (e == hipSuccess) {
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(lfx::scan_count_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
  }
  if (e == hipSuccess) {
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(lfx::fallback_tail_kernel<true>),
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lfx::ring_lds_bytes(LFX_MAX_RING_POINTS));
  }
  if (e == hipSuccess) {
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(lfx::fallback_tail_kernel<false>),
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
  c->organised_by_config = c->fused_possible;
  if (config->ring_ids != nullptr && config->n_ring_ids != 0u) {
    const int rc = install_ring_ids(c, config->ring_ids, config->n_ring_ids, true);
    if (rc != LFX_OK) {
      g_create_error = c->err;
      lfx_destroy(c);
      return rc;
    }
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
  c->scan_begin.release(); c->scan_info.release(); c->scan_geom.release(); c->chunk_base.release();
  c->ring_count.release(); c->chunk_flags.release(); c->d_label.release(); c->d_curv.release(); c->d_sidx.release();
  c->ring_status.release(); c->ring_nedge.release(); c->ring_nsurf.release(); c->ring_ebase.release();
  c->ring_sbase.release(); c->ring_flags.release(); c->counters.release(); c->scan_flags.release(); c->tail_ticket.release(); c->cum16.release(); c->hole_desc.release(); c->ring_slot.release(); c->slow_list.release(); c->defer_list.release(); c->redo_list.release(); c->fb_list.release(); c->xform.release(); c->unit_ne.release(); c->unit_ns.release(); c->unit_span.release();
  c->sxy.release(); c->sz.release(); c->sidx.release(); c->rec_pts.release(); c->rec_idx.release(); c->label_s.release();
  c->unit_tab.release();
  if (c->h_counters) {(void)hipHostFree(c->h_counters); c->h_counters = nullptr;}
  c->rec32.release();
  c->curv_s.release(); c->edge_pts.release(); c->surf_pts.release(); c->edge_idx.release(); c->surf_idx.release();
  c->staging.release();
  c->h_in.release(); c->h_out.release(); c->vox_scratch.release(); c->align_scratch.release(); c->align_surface.release(); c->h_align.release(); c->h_loc.release();
  if (c->h_status) {(void)hipHostFree(c->h_status); c->h_status = nullptr;}
  if (c->copy_stream) {(void)hipStreamSynchronize(c->copy_stream);}
  for (auto & sl : c->slots) {
    sl.in.release(); sl.hin.release(); sl.hout.release();
    if (sl.uploaded) {(void)hipEventDestroy(sl.uploaded);}
    if (sl.done) {(void)hipEventDestroy(sl.done);}
    delete static_cast<FetchPlan *>(sl.plan);
  }
  if (c->copy_stream) {(void)hipStreamDestroy(c->copy_stream);}
  if (c->stream) {(void)hipStreamDestroy(c->stream);}
  delete c;
}

int lfx_extract_batch_device(lfx_ctx * c, const void * d_points, const uint32_t * n_points, uint32_t batch, void * stream)
{
  if (!c) {return LFX_ERR_INVALID_ARGUMENT;}
  if (c->slots[0].busy || c->slots[1].busy) {return fail(c, LFX_ERR_INVALID_ARGUMENT, "a submitted scan is still in flight (lfx_extract_wait)");}
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
  v->curvature_sorted = (c->outputs & LFX_OUT_CURVATURE) ? c->curv_s.p : nullptr;
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
             fail(c, LFX_ERR_HIP, "ring bucketing timed out waiting for an earlier chunk (the workgroups of a scan were not dispatched in index order)");
    }
  }
  return LFX_OK;
}

int lfx_set_ring_ids(lfx_ctx * c, const uint16_t * ids, uint32_t n)
{
  if (!c || (ids == nullptr && n != 0u) || n > LFX_MAX_RINGS) {return LFX_ERR_INVALID_ARGUMENT;}
  if (c->slots[0].busy || c->slots[1].busy) {return fail(c, LFX_ERR_INVALID_ARGUMENT, "a submitted scan is still in flight (lfx_extract_wait)");}
  if (ids == nullptr) {
    c->slot_id.clear();
    c->ids_given = false;
    c->fused_possible = c->organised_by_config;
    return LFX_OK;
  }
  return install_ring_ids(c, ids, n, true);
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
    routes[s] = lfx::scan_took_holes(e) ? 3 : (lfx::scan_is_organised(e) ? (c->last_used_xform ? 2 : 1) : 0);
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
  if (c->slots[0].busy || c->slots[1].busy) {return fail(c, LFX_ERR_INVALID_ARGUMENT, "a submitted scan is still in flight (lfx_extract_wait)");}
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
  // Input: pinned buffers go to the device by DMA, pageable ones through the context's pinned staging buffer (upload_scan)
  if (c->h_in.bytes < total * c->layout.step) {
    bool pageable = false;
    for (uint32_t s = 0; s < batch && !pageable; s++) {
      hipPointerAttribute_t attr;
      pageable = n_points[s] && !(hipPointerGetAttributes(&attr, points[s]) == hipSuccess && attr.type == hipMemoryTypeHost);
    }
    (void)hipGetLastError();
    if (pageable && c->h_in.reserve(total * c->layout.step) != hipSuccess) {     // (reserve waits for the device before it frees)
      return fail(c, LFX_ERR_OUT_OF_MEMORY, "cannot allocate the pinned input staging buffer");
    }
  }
  size_t at = 0;
  for (uint32_t s = 0; s < batch; s++) {
    const size_t bytes = n_points[s] * c->layout.step;
    const int urc = upload_scan(c, c->staging.p + at, points[s], bytes, c->h_in, at, c->stream);
    if (urc != LFX_OK) {return urc;}
    at += bytes;
  }
  int rc = run_batch(c, c->staging.p, n32.data(), batch, c->stream);
  if (rc != LFX_OK) {return rc;}
  rc = fetch(c, 0, batch, c->stream, mask, out);
  if (rc == LFX_ERR_RING_ID && !c->ids_given) {
    // A point carries an id that is not 0 .. max_rings-1 and the caller has named none: the reference buckets by whatever
    // uint16 a point carries (ring.hpp:114-125), so the ids of these scans are looked up here, on the host's copy of the
    // records -- a pass over one field; nothing of the path itself runs on the host -- and the batch runs again with them.
    std::vector<uint8_t> seen(65536, 0);
    bool wide = false;
    for (uint32_t s = 0; s < batch; s++) {
      const uint8_t * p = static_cast<const uint8_t *>(points[s]);
      for (size_t k = 0; k < n_points[s]; k++) {
        const uint8_t * f = p + k * c->layout.step + c->layout.oring;
        uint32_t id;
        switch (c->layout.rtype) {
          case LFX_FIELD_INT8: id = (uint32_t)(int32_t)*reinterpret_cast<const int8_t *>(f); break;
          case LFX_FIELD_UINT8: id = *f; break;
          case LFX_FIELD_INT32: case LFX_FIELD_UINT32: {uint32_t v; std::memcpy(&v, f, 4); id = c->layout.be ? __builtin_bswap32(v) : v; break;}
          default: {
            uint16_t v; std::memcpy(&v, f, 2); v = c->layout.be ? __builtin_bswap16(v) : v;
            id = c->layout.rtype == LFX_FIELD_INT16 ? (uint32_t)(int32_t)(int16_t)v : v;
          }
        }
        if (id > 65535u) {wide = true;} else {seen[id] = 1;}
      }
    }
    if (wide) {return fail(c, LFX_ERR_RING_ID, "a point carries a ring id above 65535 (the reference's ring field is a uint16, point_type.hpp:62-86)");}
    std::vector<uint16_t> ids;
    for (uint32_t id = 0; id < 65536u; id++) {if (seen[id]) {ids.push_back((uint16_t)id);}}
    // (ids the context already knows stay while there is room: a sensor's rings need not all show up in every scan)
    {
      std::vector<uint16_t> both = ids;
      for (uint16_t id : c->slot_id) {if (!seen[id]) {both.push_back(id);}}
      if (both.size() <= c->max_rings) {ids = both;}
    }
    rc = install_ring_ids(c, ids.data(), (uint32_t)ids.size(), false);
    if (rc != LFX_OK) {return rc;}
    rc = run_batch(c, c->staging.p, n32.data(), batch, c->stream);
    if (rc != LFX_OK) {return rc;}
    rc = fetch(c, 0, batch, c->stream, mask, out);
  }
  return rc;
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

// The pipelined pair.  The kernels of consecutive scans run in order on the context's stream (they share the device
// scratch); what overlaps them is the NEXT scan's upload (copy_stream, a device input buffer per slot) and the host's own
// work of queueing it -- nothing here waits for the device.
int lfx_extract_submit(lfx_ctx * c, const void * points, size_t n_points, uint64_t * ticket)
{
  if (!c || !ticket || (n_points && !points)) {return LFX_ERR_INVALID_ARGUMENT;}
  if (n_points == 0) {return fail(c, LFX_ERR_INVALID_ARGUMENT, "an empty scan");}
  if (n_points > c->max_points) {return fail(c, LFX_ERR_CAPACITY, "scan exceeds max_points_per_scan");}
  lfx_ctx::Slot & sl = c->slots[c->next_ticket % 2];
  if (sl.busy) {return fail(c, LFX_ERR_INVALID_ARGUMENT, "two scans are in flight: lfx_extract_wait for the older one first");}
  LFX_HIP(c, hipSetDevice(c->device));
  if (!c->copy_stream) {LFX_HIP(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));}
  if (!sl.uploaded) {
    LFX_HIP(c, hipEventCreateWithFlags(&sl.uploaded, hipEventDisableTiming));
    LFX_HIP(c, hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    sl.plan = new FetchPlan();
  }
  const size_t bytes = n_points * c->layout.step;
  if (!sl.in.p && sl.in.alloc((size_t)c->max_points * c->layout.step) != hipSuccess) {
    sl.in.p = nullptr;
    return fail(c, LFX_ERR_OUT_OF_MEMORY, "cannot allocate the slot's input buffer");
  }
  if (sl.hin.bytes < bytes) {
    hipPointerAttribute_t attr;
    const bool pinned = hipPointerGetAttributes(&attr, points) == hipSuccess && attr.type == hipMemoryTypeHost;
    (void)hipGetLastError();
    if (!pinned && sl.hin.reserve((size_t)c->max_points * c->layout.step) != hipSuccess) {
      return fail(c, LFX_ERR_OUT_OF_MEMORY, "cannot allocate the slot's pinned staging buffer");
    }
  }
  // (the slot's previous scan has been waited for, so nothing reads its input buffer or writes its result block any more)
  int rc = upload_scan(c, sl.in.p, points, bytes, sl.hin, 0, c->copy_stream);
  if (rc != LFX_OK) {return rc;}
  LFX_HIP(c, hipEventRecord(sl.uploaded, c->copy_stream));
  LFX_HIP(c, hipStreamWaitEvent(c->stream, sl.uploaded, 0));
  const uint32_t n32 = (uint32_t)n_points;
  rc = run_batch(c, sl.in.p, &n32, 1, c->stream);
  if (rc != LFX_OK) {return rc;}
  rc = fetch_queue(c, 0, 1, c->stream, c->outputs, sl.hout, *static_cast<FetchPlan *>(sl.plan));
  if (rc != LFX_OK) {return rc;}
  LFX_HIP(c, hipEventRecord(sl.done, c->stream));
  sl.busy = true;
  sl.ticket = c->next_ticket;
  *ticket = c->next_ticket++;
  return LFX_OK;
}

int lfx_extract_wait(lfx_ctx * c, uint64_t ticket, lfx_scan_result * out)
{
  if (!c || !out) {return LFX_ERR_INVALID_ARGUMENT;}
  lfx_ctx::Slot & sl = c->slots[ticket % 2];
  if (!sl.busy || sl.ticket != ticket) {return fail(c, LFX_ERR_INVALID_ARGUMENT, "no such ticket in flight");}
  if (ticket != c->next_wait) {return fail(c, LFX_ERR_INVALID_ARGUMENT, "tickets are waited for in the order they were issued");}
  LFX_HIP(c, hipSetDevice(c->device));
  LFX_HIP(c, hipEventSynchronize(sl.done));
  sl.busy = false;
  c->next_wait = ticket + 1;
  return fetch_finish(c, *static_cast<FetchPlan *>(sl.plan), &sl.host, out);
}

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


