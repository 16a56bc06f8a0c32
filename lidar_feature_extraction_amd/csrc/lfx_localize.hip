// lfx_localize.hip -- the consumer of the two clouds: map index, residual rows, the optimizer, Localizer::Update
// (lfx_kernels_localize.hpp).
#include "lfx_internal.hpp"
#include "lfx_kernels_localize.hpp"

using namespace lfx_host;

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
  double * d_jacobian, const lfx::AlignState * states, hipStream_t st, const uint32_t * d_row_begin = nullptr)
{
  const bool wave = mi.start != nullptr;              // a grid: one query per wave; no grid: one per thread, the map through LDS
  const dim3 grid(wave ? longest : (longest + 127u) / 128u, n_clouds), block(wave ? 64 : 128);
  const float4 * pts = reinterpret_cast<const float4 *>(d_points);
#define LFX_ROWS(S, M) hipLaunchKernelGGL((lfx::scan_to_map_kernel<S, M>), grid, block, 0, st, mi, P, k, pts, d_begin, d_count, \
    count_stride, d_residual, d_jacobian, states, d_row_begin)
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
  // where each cloud's ROWS start in r3 / J3 and r1 / J1 (device, [n_clouds]); null: where its points start.  total3 /
  // total1 count rows: with compact row starts the scratch is sized by the clouds' real lengths, not by the layout the
  // points happen to lie in (lfx_localize_batch: scan s's clouds start at its first input point)
  const uint32_t * rbegin3 = nullptr, * rbegin1 = nullptr;
};

int run_align(lfx_ctx * c, const AlignProblem & pr, uint32_t n_clouds, int max_iter, const double * initial_poses,
  lfx_align_result * results, hipStream_t st)
{
  static_assert(sizeof(lfx::AlignState) % 8 == 0, "AlignState is an array of doubles' worth");
  if (n_clouds > 65535u) {return fail(c, LFX_ERR_INVALID_ARGUMENT, "at most 65535 scans per alignment call");}   // (a launch's y extent)
  const size_t state_d = sizeof(lfx::AlignState) / 8 * (size_t)n_clouds;
  const size_t rows = pr.total3 + pr.total1;
  const size_t partial_d = (size_t)n_clouds * lfx::kAlignSlices * lfx::kAlignTile;
  const size_t nbr_d = (size_t)lfx::kNearestMax / 2 * rows;                   // the searches' results: 16 words per row
  const size_t reach_d = rows;                                               // and how far each row's 16th neighbour was
  const size_t need = state_d + 24 * pr.total3 + 8 * pr.total1 + rows + partial_d + nbr_d + reach_d + (n_clouds + 1) / 2 + 9;
  if (c->align_scratch.n < need) {
    c->align_scratch.release();
    if (c->align_scratch.alloc(need) != hipSuccess) {
      c->align_scratch.n = 0;
      return fail(c, LFX_ERR_OUT_OF_MEMORY, "cannot allocate the rows of the scan-to-map alignment");
    }
  }
  double * w = c->align_scratch.p;
  lfx::AlignState * states = reinterpret_cast<lfx::AlignState *>(w); w += state_d;
  double * r3 = w; w += 3 * pr.total3;
  double * J3 = w; w += 21 * pr.total3;
  double * r1 = w; w += pr.total1;
  double * J1 = w; w += 7 * pr.total1;
  double * d_weights = w; w += rows;
  double * d_partials = w; w += partial_d;
  uint32_t * nbr3 = reinterpret_cast<uint32_t *>(w), * nbr1 = nbr3 + (size_t)lfx::kNearestMax * pr.total3; w += nbr_d;
  double * reach3 = w, * reach1 = w + pr.total3; w += reach_d;
  uint32_t * d_tickets = reinterpret_cast<uint32_t *>(w); w += (n_clouds + 1) / 2;
  uint32_t * d_active = reinterpret_cast<uint32_t *>(w);
  // Pinned host memory, read and written by the kernels themselves: [the caller's poses | a result record per scan].  No
  // copy is queued in either direction; the thread that ends a scan's iterations writes its record.
  const size_t pose_bytes = 96 * (size_t)n_clouds;
  LFX_HIP(c, c->h_align.reserve(pose_bytes + sizeof(lfx::AlignOut) * (size_t)n_clouds));
  std::memcpy(c->h_align.p, initial_poses, pose_bytes);
  volatile lfx::AlignOut * out = reinterpret_cast<volatile lfx::AlignOut *>(c->h_align.p + pose_bytes);
  void * d_pinned = nullptr;
  LFX_HIP(c, hipHostGetDevicePointer(&d_pinned, c->h_align.p, 0));
  const double * d_initial = static_cast<const double *>(d_pinned);
  lfx::AlignOut * d_out = reinterpret_cast<lfx::AlignOut *>(static_cast<uint8_t *>(d_pinned) + pose_bytes);
  hipLaunchKernelGGL(lfx::align_begin_kernel, dim3((n_clouds + 63u) / 64u), dim3(64), 0, st, states, d_initial, n_clouds, d_active,
    d_tickets, d_out);
  const lfx::MapPose none{};
  auto iteration = [&](int iter) {
      if (pr.X) {
        if (pr.longest3) {
          hipLaunchKernelGGL(lfx::pair_rows_kernel, dim3((pr.longest3 + 127u) / 128u, n_clouds), dim3(128), 0, st, pr.X, pr.Y,
            pr.begin3, pr.count3, r3, J3, states);
        }
      } else {
        const bool both_grids = pr.edge_map->index.start && pr.surface_map->index.start;
        if (both_grids) {
          // the searches of both kinds in one launch, one wave per query; then the rows, one thread per query
          const lfx::RowsOfKind e{pr.edge_map->index, reinterpret_cast<const float4 *>(pr.edge_points), pr.begin3, pr.count3, pr.stride3, r3, J3,
            pr.rbegin3, nbr3, reach3};
          const lfx::RowsOfKind f{pr.surface_map->index, reinterpret_cast<const float4 *>(pr.surface_points), pr.begin1, pr.count1, pr.stride1,
            r1, J1, pr.rbegin1, nbr1, reach1};
          if (pr.longest3 + pr.longest1) {
            const uint32_t w3 = (pr.longest3 + lfx::kSearchWaves - 1u) / lfx::kSearchWaves, w1 = (pr.longest1 + lfx::kSearchWaves - 1u) / lfx::kSearchWaves;
            hipLaunchKernelGGL(lfx::map_search_kernel, dim3(w3 + w1, n_clouds), dim3(64 * lfx::kSearchWaves), 0, st, e, f, w3,
              pr.n_neighbors, states, iter);
            const uint32_t g3 = (pr.longest3 + lfx::kRowThreads - 1u) / lfx::kRowThreads, g1 = (pr.longest1 + lfx::kRowThreads - 1u) / lfx::kRowThreads;
            hipLaunchKernelGGL(lfx::rows_from_neighbours_kernel, dim3(g3 + g1, n_clouds), dim3(lfx::kRowThreads), 0, st, e, f, g3,
              pr.n_neighbors, states);
          }
        } else {
          if (pr.longest3) {
            launch_rows(false, pr.edge_map->index, none, pr.n_neighbors, pr.edge_points, pr.begin3, pr.count3, pr.stride3, n_clouds,
              pr.longest3, r3, J3, states, st, pr.rbegin3);
          }
          if (pr.longest1) {
            launch_rows(true, pr.surface_map->index, none, pr.n_neighbors, pr.surface_points, pr.begin1, pr.count1, pr.stride1, n_clouds,
              pr.longest1, r1, J1, states, st, pr.rbegin1);
          }
        }
      }
      // (the step kernels only address rows)
      const uint32_t * rb3 = pr.rbegin3 ? pr.rbegin3 : pr.begin3, * rb1 = pr.rbegin1 ? pr.rbegin1 : pr.begin1;
      hipLaunchKernelGGL(lfx::align_scale_kernel, dim3(n_clouds), dim3(lfx::kScaleThreads), 0, st, states, iter, r3, rb3, pr.count3,
        pr.stride3, r1, rb1, pr.count1, pr.stride1, d_weights, d_active, d_out);
      hipLaunchKernelGGL(lfx::align_update_kernel, dim3(lfx::kAlignSlices, n_clouds), dim3(lfx::kAlignThreads), 0, st, states, iter,
        max_iter, r3, J3, rb3, pr.count3, pr.stride3, r1, J1, rb1, pr.count1, pr.stride1, d_weights, d_partials, d_tickets,
        d_active, d_out);
    };
  // As many iterations as the previous call needed are queued at once (a finished scan's kernels return at once, but a launch
  // is a launch); only then does the host look -- at the records in its own memory -- and, where a scan still iterates,
  // goes on one iteration at a time.
  int launched = 0, target = std::min(max_iter, std::max(1, c->align_guess));
  for (;;) {
    for (; launched < target; launched++) {iteration(launched);}
    LFX_HIP(c, hipGetLastError());
    // the wait: on the records themselves first (the thread that ends a scan sets `done` behind a system-scope fence; a
    // blocking wait on the stream wakes 30-50 us late, a third of what a whole scan's alignment takes), then until the stream
    // has drained what was queued behind them
    auto all_done = [&]() {
        for (uint32_t s = 0; s < n_clouds; s++) {if (out[s].done == 0) {return false;}}
        return true;
      };
    hipError_t q = hipErrorNotReady;
    for (uint32_t spins = 0; q == hipErrorNotReady; spins++) {
      if (all_done() || (spins & 63u) == 63u) {q = hipStreamQuery(st);}
      if (spins > (1u << 24)) {q = hipStreamSynchronize(st);}         // (seconds: something else holds the stream)
    }
    LFX_HIP(c, q);
    if (all_done()) {break;}
    if (launched >= max_iter) {return fail(c, LFX_ERR_HIP, "the alignment did not finish within its iterations");}   // (cannot happen)
    target = launched + 1;
  }
  int needed = 1;
  for (uint32_t s = 0; s < n_clouds; s++) {
    for (int i = 0; i < 12; i++) {results[s].pose[i] = out[s].pose[i];}
    results[s].error = out[s].error; results[s].error_scale = out[s].scale;
    results[s].iteration = out[s].iteration; results[s].code = out[s].code;
    needed = std::max(needed, std::min(out[s].iteration + 1, max_iter));
  }
  c->align_guess = needed;
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
    case LFX_ALIGN_NO_PLANE: return "No surface neighbourhood spans a plane";
    default: return "unknown";
  }
}

namespace
{
int align_clouds(
  lfx_ctx * c, const lfx_map * edge_map, const lfx_map * surface_map, uint32_t n_neighbors, int max_iter,
  const float * d_edge_points, const uint32_t * d_edge_begin, const uint32_t * d_edge_count, uint32_t edge_count_stride,
  uint32_t max_edge_points_per_cloud, size_t total_edge_points,
  const float * d_surface_points, const uint32_t * d_surface_begin, const uint32_t * d_surface_count,
  uint32_t surface_count_stride, uint32_t max_surface_points_per_cloud, size_t total_surface_points,
  uint32_t n_clouds, const double * initial_poses, lfx_align_result * results, void * stream,
  const uint32_t * d_edge_row_begin, const uint32_t * d_surface_row_begin)
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
  pr.rbegin3 = d_edge_row_begin; pr.rbegin1 = d_surface_row_begin;
  return run_align(c, pr, n_clouds, max_iter, initial_poses, results, static_cast<hipStream_t>(stream));
}
}  // namespace

int lfx_scan_to_map_align(
  lfx_ctx * c, const lfx_map * edge_map, const lfx_map * surface_map, uint32_t n_neighbors, int max_iter,
  const float * d_edge_points, const uint32_t * d_edge_begin, const uint32_t * d_edge_count, uint32_t edge_count_stride,
  uint32_t max_edge_points_per_cloud, size_t total_edge_points,
  const float * d_surface_points, const uint32_t * d_surface_begin, const uint32_t * d_surface_count,
  uint32_t surface_count_stride, uint32_t max_surface_points_per_cloud, size_t total_surface_points,
  uint32_t n_clouds, const double * initial_poses, lfx_align_result * results, void * stream)
{
  return align_clouds(c, edge_map, surface_map, n_neighbors, max_iter, d_edge_points, d_edge_begin, d_edge_count, edge_count_stride,
           max_edge_points_per_cloud, total_edge_points, d_surface_points, d_surface_begin, d_surface_count, surface_count_stride,
           max_surface_points_per_cloud, total_surface_points, n_clouds, initial_poses, results, stream, nullptr, nullptr);
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
  lfx_ctx * c, const lfx_map * edge_map, const lfx_map * surface_map, uint32_t n_neighbors, int max_iter, float surface_leaf,
  uint32_t n_scans, const double * initial_poses, lfx_align_result * results, void * stream)
{
  if (!c || !initial_poses || !results) {return LFX_ERR_INVALID_ARGUMENT;}
  if (c->last_batch == 0) {return fail(c, LFX_ERR_INVALID_ARGUMENT, "no batch has been extracted yet");}
  // the caller's arrays are sized by ITS count: one pose after a batch of 16 would be read and written 15 entries too far
  if (n_scans != c->last_batch) {
    return fail(c, LFX_ERR_INVALID_ARGUMENT, "n_scans (" + std::to_string(n_scans) + ") is not the number of scans of the last batch (" +
             std::to_string(c->last_batch) + ")");
  }
  LFX_HIP(c, hipSetDevice(c->device));
  const uint32_t batch = c->last_batch;
  const size_t total = c->h_scan_begin[batch];
  const size_t need = 4 * total + 4 * (size_t)batch;        // the clouds, then counts, status and the two tables of row starts
  if (c->align_surface.n < need) {
    c->align_surface.release();
    if (c->align_surface.alloc(need) != hipSuccess) {
      c->align_surface.n = 0;
      return fail(c, LFX_ERR_OUT_OF_MEMORY, "cannot allocate the downsampled surface clouds");
    }
  }
  float * down = c->align_surface.p;
  uint32_t * down_count = reinterpret_cast<uint32_t *>(down + 4 * total), * down_status = down_count + batch;
  uint32_t * d_row3 = down_status + batch, * d_row1 = d_row3 + batch;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // A few scans: nothing is asked of the device before the alignment.  The rows' scratch is sized by a bound (a scan has no
  // more edge points, and no more surface points, than points: 400 bytes per input point), the rows of scan s start where
  // its points do, and the launches are sized by the previous call's longest clouds (the kernels stride over what there is).
  const bool by_bound = 400u * total <= ((size_t)192 << 20);      // (four 64 x 1800 scans; the scratch only ever grows)
  LFX_HIP(c, c->h_loc.reserve(8 * (size_t)batch));
  volatile uint32_t * lengths = reinterpret_cast<volatile uint32_t *>(c->h_loc.p);
  void * d_lengths = nullptr;
  LFX_HIP(c, hipHostGetDevicePointer(&d_lengths, c->h_loc.p, 0));
  // Downsample of the surface clouds (where PCL gives a cloud back unfiltered -- leaf too small for its extent -- the rows are
  // built from all its points: the kernel copies it), the clouds' lengths left in pinned memory on the way
  const int rc = voxel_downsample(c, reinterpret_cast<const float *>(c->surf_pts.p), c->scan_begin.p, c->scan_info.p + lfx::kInfoSurface, 4,
    batch, total, surface_leaf, down, down_count, down_status, stream, true, c->scan_info.p + lfx::kInfoEdge, static_cast<uint32_t *>(d_lengths));
  if (rc != LFX_OK) {return rc;}
  auto remember = [&]() {
      uint32_t e = 0, f = 0;
      for (uint32_t s = 0; s < batch; s++) {e = std::max(e, (uint32_t)lengths[2 * s]); f = std::max(f, (uint32_t)lengths[2 * s + 1]);}
      c->loc_guess[0] = e; c->loc_guess[1] = f;
    };
  if (by_bound) {
    const uint32_t guess3 = c->loc_guess[0] ? c->loc_guess[0] + c->loc_guess[0] / 8u : 4096u;
    const uint32_t guess1 = c->loc_guess[1] ? c->loc_guess[1] + c->loc_guess[1] / 8u : 2048u;
    const int ra = align_clouds(c, edge_map, surface_map, n_neighbors, max_iter,
      reinterpret_cast<const float *>(c->edge_pts.p), c->scan_begin.p, c->scan_info.p + lfx::kInfoEdge, 4, guess3, total,
      down, c->scan_begin.p, down_count, 1, guess1, total, batch, initial_poses, results, stream, nullptr, nullptr);
    if (ra == LFX_OK) {remember();}
    return ra;
  }
  // Many scans: the clouds' real lengths first (one wait), so that the rows are packed -- scan s's rows start at the number of
  // edge (downsampled surface) points of the scans before it, 400 bytes per row of a few thousand rows per scan instead of
  // per input point
  LFX_HIP(c, hipStreamSynchronize(st));
  LFX_HIP(c, c->h_align.reserve(8 * (size_t)batch));
  uint32_t * rows = reinterpret_cast<uint32_t *>(c->h_align.p);
  uint32_t longest_edge = 0, longest_surface = 0;
  size_t rows3 = 0, rows1 = 0;
  for (uint32_t s = 0; s < batch; s++) {
    rows[s] = (uint32_t)rows3; rows[batch + s] = (uint32_t)rows1;
    rows3 += lengths[2 * s]; rows1 += lengths[2 * s + 1];
    longest_edge = std::max(longest_edge, (uint32_t)lengths[2 * s]);
    longest_surface = std::max(longest_surface, (uint32_t)lengths[2 * s + 1]);
  }
  c->loc_guess[0] = longest_edge; c->loc_guess[1] = longest_surface;
  LFX_HIP(c, hipMemcpyAsync(d_row3, rows, sizeof(uint32_t) * 2 * batch, hipMemcpyHostToDevice, st));
  LFX_HIP(c, hipStreamSynchronize(st));              // (run_align lays its own records over the pinned block)
  return align_clouds(c, edge_map, surface_map, n_neighbors, max_iter,
           reinterpret_cast<const float *>(c->edge_pts.p), c->scan_begin.p, c->scan_info.p + lfx::kInfoEdge, 4, longest_edge, rows3,
           down, c->scan_begin.p, down_count, 1, longest_surface, rows1, batch, initial_poses, results, stream, d_row3, d_row1);
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
      d_words, d_words + 2, 1u, reinterpret_cast<float4 *>(d_down), d_words + 3, d_words + 4, d_words + 1, static_cast<uint32_t *>(nullptr));
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
