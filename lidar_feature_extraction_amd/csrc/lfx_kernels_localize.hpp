// lfx_kernels_localize.hpp -- the consumer: map index, nearest neighbours, residual rows, the optimizer.
#pragma once

#include "lfx_kernels_common.hpp"

#pragma clang fp contract(off)

namespace lfx
{

// ------------------------------------------------------------------------------------------
// Scan-to-map residual build (first slice): what the reference's localizer does with the two clouds this
// path emits -- for every edge point the line through its k nearest edge-map points (mean + principal direction of their
// covariance), residual (p - p1) x (p - p2) and its 3 x 7 Jacobian row (localization: edge.hpp:86-124, src/edge.cpp:38-84);
// for every (downsampled) surface point the plane through its k nearest surface-map points (least squares X w = -1),
// residual = signed point-plane distance and its 1 x 7 row (surface.hpp:40-139, math.hpp:36-40); quaternion derivative
// rotationlib/src/jacobian/quaternion.cpp:35-52.  One thread per scan point; the map streams through LDS in tiles and
// every thread keeps its 16 nearest candidates (exact search, squared L2 in f64, ties by the lower map index).
// PARITY UNPINNED beyond the vectors of localization/test/test_edge.cpp / test_math.cpp: Eigen's and nanoflann's own
// arithmetic (reduction orders, computeDirect, householderQr, order of equidistant neighbours) is not available here;
// results agree with the CPU restatement to ~1e-9 relative, the edge rows up to the sign of the principal direction
// (residual and Jacobian flip together, J^T r does not).
constexpr int kNearestMax = 16;          // the localizer uses N_NEIGHBORS = 15 (localizer.hpp:46)
struct MapPose
{
  double m[12];                           // point_to_map as [R | t], row-major 3 x 4
  double qw, qx, qy, qz;                  // Eigen::Quaterniond(R), computed by the host
};

// state of Optimizer::Run (optimizer.hpp:79-123) for one scan, kept on the device between the kernels of an iteration
struct AlignState
{
  MapPose pose;                           // MakePose(q, t) and Quaterniond(pose.rotation()), what Problem::Make is given
  double q[4], t[3];                      // the optimizer's q (w x y z) and t
  double prev_error, prev_scale;
  double error, scale;                    // OptimizationResult
  double cur_error, cur_scale;            // this iteration's, from align_scale_kernel to align_update_kernel
  int32_t iteration, code, done, surface_rows_with_plane;      // (the last one: counted by the row kernels of the iteration, reset by align_scale_kernel)
  double prev_m[12];                      // the pose of the iteration before (the searches bound how far a query has moved)
};

// what the host reads when a scan has stopped iterating: written by the thread that stops it, straight into pinned host memory
struct AlignOut
{
  double pose[12];
  double error, scale;
  int32_t iteration, code, done, pad;
};

struct D3 { double x, y, z; };
__device__ inline D3 d3_sub(D3 a, D3 b) {return {a.x - b.x, a.y - b.y, a.z - b.z};}
__device__ inline D3 d3_cross(D3 a, D3 b) {return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};}
__device__ inline double d3_dot(D3 a, D3 b) {return a.x * b.x + a.y * b.y + a.z * b.z;}

// unit eigenvector of the largest eigenvalue of a symmetric 3 x 3 matrix, closed form (trigonometric roots of the
// characteristic polynomial of the shifted, scaled matrix; the kernel of (A - lambda I) from the larger of two cross
// products of its columns).  (0, 0, 1) where the spectrum is isotropic.
__device__ inline D3 principal_direction(const double (&c)[6] /* xx xy xz yy yz zz */)
{
  const double shift = (c[0] + c[3] + c[5]) / 3.;
  double a00 = c[0] - shift, a11 = c[3] - shift, a22 = c[5] - shift, a01 = c[1], a02 = c[2], a12 = c[4];
  double scale = fmax(fmax(fabs(a00), fabs(a11)), fmax(fabs(a22), fmax(fabs(a01), fmax(fabs(a02), fabs(a12)))));
  if (!(scale > 0.)) {return {0., 0., 1.};}
  const double inv = 1. / scale;
  a00 *= inv; a11 *= inv; a22 *= inv; a01 *= inv; a02 *= inv; a12 *= inv;
  // roots of x^3 - c1' x - c0 (trace is zero after the shift)
  const double c0 = a00 * a11 * a22 + 2. * a01 * a02 * a12 - a00 * a12 * a12 - a11 * a02 * a02 - a22 * a01 * a01;
  const double c1 = a00 * a11 - a01 * a01 + a00 * a22 - a02 * a02 + a11 * a22 - a12 * a12;
  double a3 = -c1 / 3.;
  a3 = a3 > 0. ? a3 : 0.;
  const double half_b = 0.5 * c0;
  double qd = a3 * a3 * a3 - half_b * half_b;
  qd = qd > 0. ? qd : 0.;
  const double rho = sqrt(a3), theta = atan2(sqrt(qd), half_b) / 3.;
  const double lambda = 2. * rho * cos(theta);                       // the largest root
  const double lo = -rho * (cos(theta) + 1.7320508075688772 * sin(theta));
  if (!(lambda - lo > 1e-14)) {return {0., 0., 1.};}
  const double m00 = a00 - lambda, m11 = a11 - lambda, m22 = a22 - lambda;
  // columns of (A - lambda I); the one with the largest diagonal entry in magnitude is the representative
  const D3 col0{m00, a01, a02}, col1{a01, m11, a12}, col2{a02, a12, m22};
  const double d0 = fabs(m00), d1 = fabs(m11), d2 = fabs(m22);
  D3 rep = col0, o1 = col1, o2 = col2;
  if (d1 > d0 && d1 >= d2) {rep = col1; o1 = col2; o2 = col0;} else if (d2 > d0 && d2 > d1) {rep = col2; o1 = col0; o2 = col1;}
  const D3 x1 = d3_cross(rep, o1), x2 = d3_cross(rep, o2);
  const double n1 = d3_dot(x1, x1), n2 = d3_dot(x2, x2);
  const D3 v = n1 > n2 ? x1 : x2;
  const double nn = n1 > n2 ? n1 : n2;
  if (!(nn > 0.)) {return {0., 0., 1.};}
  const double s = 1. / sqrt(nn);
  return {v.x * s, v.y * s, v.z * s};
}

// rotationlib::DRpDq: 3 x 4, row-major
__device__ inline void drp_dq(const MapPose & P, D3 p, double (&d)[12])
{
  const D3 v{P.qx, P.qy, P.qz};
  const D3 vxp = d3_cross(v, p);
  const double vp = d3_dot(v, p), w = P.qw;
  const double c0[3] = {w * p.x + vxp.x, w * p.y + vxp.y, w * p.z + vxp.z};
  const double K[9] = {0., -p.z, p.y, p.z, 0., -p.x, -p.y, p.x, 0.};
  const double vv[3] = {v.x, v.y, v.z}, pp[3] = {p.x, p.y, p.z};
#pragma unroll
  for (int r = 0; r < 3; r++) {
    d[4 * r] = 2. * c0[r];
#pragma unroll
    for (int c = 0; c < 3; c++) {d[4 * r + 1 + c] = 2. * ((r == c ? vp : 0.) + vv[r] * pp[c] - pp[r] * vv[c] - w * K[3 * r + c]);}
  }
}

// The map a scan is matched against: the reference's KDTreeEigen (localization/include/lidar_feature_localization/
// kdtree.hpp:50-63: built once per map, exact k-nearest queries).  Here a uniform grid of cubic cells: the points sorted
// by cell (x fastest), start[] = the first point of every cell, so that the cells of one grid row between two x are one
// contiguous run of points.  start == nullptr: no grid, every query reads the whole map (small maps, and the check of
// the grid).  Both give the same neighbours in the same order: ascending distance, equal distances by the lower index
// of the point in the map as it was given.
struct MapIndex
{
  const float4 * pts;                     // grid: sorted by cell, w = the point's original index (bits); else as given
  const uint32_t * start;                 // [nx * ny * nz + 1] or nullptr
  double ox, oy, oz, h, inv_h;            // cell (ix, iy, iz) = floor((p - o) * inv_h), clamped into the grid
  int nx, ny, nz;
  uint32_t n;
};

// (a NaN coordinate converts to cell 0: a query that is not a number walks the grid from there and ends when its cube holds it)
__device__ inline int cell_coordinate(double p, double o, double inv_h)
{
  double u = floor((p - o) * inv_h);
  u = u < -268435456. ? -268435456. : (u > 268435456. ? 268435456. : u);
  return (int)u;
}

// candidate (d, at) into the ascending list of the no-grid search; candidates arrive by ascending map index, so `<` alone
// keeps equal distances in index order
__device__ inline void nearest_insert(double (&dist)[kNearestMax], uint32_t (&idx)[kNearestMax], double d, uint32_t at)
{
  constexpr int KM = kNearestMax;
  if (!(d < dist[KM - 1])) {return;}
  bool placed = false;
#pragma unroll
  for (int j = KM - 1; j > 0; j--) {
    if (!placed) {
      if (d < dist[j - 1]) {dist[j] = dist[j - 1]; idx[j] = idx[j - 1];} else {dist[j] = d; idx[j] = at; placed = true;}
    }
  }
  if (!placed) {dist[0] = d; idx[0] = at;}
}

// every thread of the workgroup (T threads) calls this; the map passes through LDS in tiles
template<int T>
__device__ __forceinline__ void nearest_whole_map(const MapIndex & mi, D3 q, double (&dist)[kNearestMax], uint32_t (&idx)[kNearestMax], float4 * tile)
{
  const uint32_t tid = threadIdx.x, n_map = mi.n;
  for (uint32_t t0 = 0; t0 < n_map; t0 += T) {
    __syncthreads();
    tile[tid] = mi.pts[t0 + tid < n_map ? t0 + tid : n_map - 1u];
    __syncthreads();
    const uint32_t lim = n_map - t0 < (uint32_t)T ? n_map - t0 : (uint32_t)T;
    for (uint32_t e = 0; e < lim; e++) {
      const float4 mpt = tile[e];
      const double dx = (double)mpt.x - q.x, dy = (double)mpt.y - q.y, dz = (double)mpt.z - q.z;
      nearest_insert(dist, idx, dx * dx + dy * dy + dz * dz, t0 + e);
    }
  }
}

// the cube of cells within rho of the query's cell, clipped to the grid, and the test that ends the search
struct GridCube
{
  double ux, uy, uz;
  int cx, cy, cz, rho;
  int xlo, xhi, ylo, yhi, zlo, zhi;
  __device__ __forceinline__ void begin(const MapIndex & mi, D3 q)
  {
    ux = (q.x - mi.ox) * mi.inv_h; uy = (q.y - mi.oy) * mi.inv_h; uz = (q.z - mi.oz) * mi.inv_h;
    cx = cell_coordinate(q.x, mi.ox, mi.inv_h); cy = cell_coordinate(q.y, mi.oy, mi.inv_h); cz = cell_coordinate(q.z, mi.oz, mi.inv_h);
    auto outside = [](int c, int n) {return c < 0 ? -c : (c > n - 1 ? c - (n - 1) : 0);};
    rho = max(max(outside(cx, mi.nx), outside(cy, mi.ny)), max(outside(cz, mi.nz), 1));
    clip(mi);
  }
  __device__ __forceinline__ void clip(const MapIndex & mi)
  {
    xlo = max(cx - rho, 0); xhi = min(cx + rho, mi.nx - 1); ylo = max(cy - rho, 0); yhi = min(cy + rho, mi.ny - 1);
    zlo = max(cz - rho, 0); zhi = min(cz + rho, mi.nz - 1);
  }
  __device__ __forceinline__ void grow(const MapIndex & mi) {rho = rho < 2 ? rho + 1 : 2 * rho; clip(mi);}
  __device__ __forceinline__ uint32_t candidates(const MapIndex & mi) const
  {
    uint32_t total = 0;
    if (xlo <= xhi) {
      for (int z = zlo; z <= zhi; z++) {
        for (int y = ylo; y <= yhi; y++) {
          const size_t row = ((size_t)z * mi.ny + y) * mi.nx;
          total += mi.start[row + xhi + 1] - mi.start[row + xlo];
        }
      }
    }
    return total;
  }
  // all points seen, or the kk-th distance inside the cube's inscribed sphere (no unseen point can be nearer): a point in a
  // cell outside the cube differs from the query by at least g cells along some axis (1e-7 cells: rounding of the cell
  // coordinates of points on a face)
  __device__ __forceinline__ bool done(const MapIndex & mi, double kth) const
  {
    if (cx - rho <= 0 && cx + rho >= mi.nx - 1 && cy - rho <= 0 && cy + rho >= mi.ny - 1 && cz - rho <= 0 && cz + rho >= mi.nz - 1) {return true;}
    const double r = (double)rho;
    const double g = fmin(fmin(fmin(ux - ((double)cx - r), ((double)cx + r + 1.) - ux), fmin(uy - ((double)cy - r), ((double)cy + r + 1.) - uy)),
        fmin(uz - ((double)cz - r), ((double)cz + r + 1.) - uz)) - 1e-7;
    const double reach = g * mi.h;
    return g > 0. && kth <= reach * reach;
  }
};

#ifndef LFX_GRID_UNROLL
#define LFX_GRID_UNROLL 2
#endif
constexpr int kGridUnroll = LFX_GRID_UNROLL;      // runs of 64 points loaded at once
#ifndef LFX_BULK_INSERT
#define LFX_BULK_INSERT 24
#endif
constexpr int kBulkInsert = LFX_BULK_INSERT;      // this many points passing the bar at once are merged in, not inserted one by one
#ifndef LFX_MERGE_INSERT
#define LFX_MERGE_INSERT 4
#endif
constexpr int kMergeInsert = LFX_MERGE_INSERT;    // this many (and fewer than kBulkInsert): placed by counting, in one step

__device__ __forceinline__ double wave_read(double v, int lane)
{
  const long long b = __double_as_longlong(v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, lane), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), lane);
  return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}

// ascending bitonic sort of (d, orig) with `at` carried along, over the 64 lanes of the wave
__device__ __forceinline__ void wave_sort_steps(double & d, uint32_t & orig, uint32_t & at, int lane, int k_first, int k_last)
{
  for (int k = k_first; k <= k_last; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      const double od = __shfl_xor(d, j, 64);
      const uint32_t oo = (uint32_t)__shfl_xor((int)orig, j, 64), oa = (uint32_t)__shfl_xor((int)at, j, 64);
      const bool want_min = ((lane & j) == 0) == ((lane & k) == 0);      // the lower lane of a pair in an ascending run
      const bool other_less = od < d || (od == d && oo < orig), self_less = d < od || (d == od && orig < oo);
      const bool take = want_min ? other_less : self_less;
      d = take ? od : d; orig = take ? oo : orig; at = take ? oa : at;
    }
  }
}

// one wave, one query (the same q in every lane): rho grown until GridCube::done.  The 64 lanes take 64 consecutive points
// of a run of cells at a time; the list of the KM nearest so far lives in lanes 0..KM-1 (distance, position, original
// index), its last distance is the bar a point has to pass, and the few points that pass are inserted one at a time (a
// shift along the lanes).  All 64 lanes must be here; the list comes back in lanes 0..KM-1.
// (Measured against one query per thread -- lists in registers with batched insertion, or heaps in LDS: a thread inserts
// for a few points in a hundred, but some thread of 64 does at nearly every point, so the wave paid the insertion at every
// point; this form was 2-4x faster from one scan to 64 and level at 256, and it is the only one kept.)
// known, reach2 (optional): a squared distance within which at least KM map points are KNOWN to lie (the caller's bound;
// infinity says nothing).  The list's bar then starts there instead of at infinity: nothing farther is ever looked at, so
// the first candidates are not all sorted into the list only to be pushed out again, and rows beyond it are not read.
// (A flag beside the value, not a NaN for "none": a wave-uniform double constant is what hipcc 7.2 gets wrong, see below.)
__device__ __forceinline__ void nearest_in_grid_wave(const MapIndex & mi, D3 q, uint32_t kk, double & ldist, uint32_t & lidx,
  bool known = false, double reach2 = 0.)
{
  constexpr int KM = kNearestMax;
  const int lane = threadIdx.x & 63;
  GridCube cube;
  cube.begin(mi, q);
  if (known && cube.rho == 1) {
    // With a bound the cube can be made large enough at once for the search to END with it (GridCube::done: the 16th distance
    // within the cube's inscribed sphere): the sphere of the bound has to fit, i.e. rho cells plus the query's distance to the
    // nearest face of its own cell.  A query in thin surroundings otherwise walks the small cube first, finds too little and
    // starts again.  Rows beyond the bound are never fetched, so the larger cube costs its row table only (up to 7 x 7 rows
    // fit the wave's lanes; a bound that needs more starts small as before).
    const double g0 = fmin(fmin(fmin(cube.ux - (double)cube.cx, (double)cube.cx + 1. - cube.ux), fmin(cube.uy - (double)cube.cy, (double)cube.cy + 1. - cube.uy)),
        fmin(cube.uz - (double)cube.cz, (double)cube.cz + 1. - cube.uz));
    const double need = ceil(sqrt(reach2) * mi.inv_h - fmax(g0, 0.) + 1e-6);
    if (need >= 2. && need <= 3.) {cube.rho = (int)need; cube.clip(mi);}
  }
  uint32_t lorig = 0u;
  for (;;) {
    ldist = INFINITY; lidx = 0u; lorig = 0xFFFFFFFFu;
    // (the bar read back from the list, through an empty asm the compiler cannot see through, rather than set to the
    // constant: hipcc 7.2 materialises a wave-uniform double constant with s_mov_b64 and a 64-bit literal, which gfx950
    // truncates to its low word -- infinity became 0.0 here)
    asm volatile("" : "+v"(ldist));
    const bool bounded = known && reach2 < ldist;              // (ldist is infinity here: a bound of infinity is none)
    // the bar: the list's last entry once that is within the caller's bound, the bound (any point AT it passes) until then
    auto set_bar = [&](double & bar_, uint32_t & bar_orig_) __attribute__((always_inline)) {
        const double d_last = wave_read(ldist, KM - 1);
        const uint32_t o_last = (uint32_t)__builtin_amdgcn_readlane((int)lorig, KM - 1);
        const bool mine = !bounded || d_last <= reach2;
        bar_ = mine ? d_last : reach2;
        bar_orig_ = mine ? o_last : 0xFFFFFFFFu;
      };
    double bar;
    uint32_t bar_orig;
    set_bar(bar, bar_orig);
    auto run = [&](uint32_t a, uint32_t b) __attribute__((always_inline)) {
        for (uint32_t base = a; base < b; base += 64u * kGridUnroll) {
          float4 mpts[kGridUnroll];
#pragma unroll
          for (int u = 0; u < kGridUnroll; u++) {
            const uint32_t at = base + 64u * u + (uint32_t)lane;
            mpts[u] = mi.pts[at < b ? at : b - 1u];
          }
#pragma unroll
          for (int u = 0; u < kGridUnroll; u++) {
            const uint32_t at = base + 64u * u + (uint32_t)lane;
            if (base + 64u * u >= b) {continue;}                                      // (the same in every lane)
            const bool live = at < b;
            const float4 mpt = mpts[u];
            const double dx = (double)mpt.x - q.x, dy = (double)mpt.y - q.y, dz = (double)mpt.z - q.z;
            const double d = dx * dx + dy * dy + dz * dz;
            const uint32_t orig = __float_as_uint(mpt.w);
            uint64_t pass = __ballot(live && (d < bar || (d == bar && orig < bar_orig)));
            if (__popcll(pass) >= kBulkInsert) {
              // many at once (the first points of a query, before there is a bar worth the name): sort the 64 of them, merge
              // their 16 smallest with the list -- the cost of about eight single insertions, whatever their number
              const bool mine = ((pass >> lane) & 1ull) != 0ull;
              double sd = mine ? d : INFINITY;
              uint32_t so = mine ? orig : 0xFFFFFFFFu, sa = mine ? at : 0u;
              wave_sort_steps(sd, so, sa, lane, 2, 64);
              // lanes 0..15 now hold the 16 smallest, ascending; against the list, reversed: the smaller of each pair are the
              // 16 smallest of the 32 and form a bitonic run, which four more steps put in order
              const int from = 15 - (lane & 15);
              const double rd = __shfl(sd, from, 64);
              const uint32_t ro = (uint32_t)__shfl((int)so, from, 64), ra = (uint32_t)__shfl((int)sa, from, 64);
              const bool cand_less = rd < ldist || (rd == ldist && ro < lorig);
              if (lane < KM) {ldist = cand_less ? rd : ldist; lorig = cand_less ? ro : lorig; lidx = cand_less ? ra : lidx;}
              wave_sort_steps(ldist, lorig, lidx, lane, 16, 16);             // (k = 16 within lanes 0..15: ascending)
              set_bar(bar, bar_orig);
              pass = 0;
            }
            if (__popcll(pass) >= kMergeInsert) {
              // a few at once: every passing candidate is told to all lanes in turn (no lane-to-lane traffic), the lanes count
              // for their list entry the candidates that come before it and for their candidate the list entries and the
              // candidates that come before it -- its place in the merged order --, and two pushes (the entries, then the
              // candidates: ds_permute hands a lane nothing written to it as 0) put the first 16 of the merged order in
              // place.  One after the other the same insertions are a chain: each has to see the list the last one left.
              const bool mine = ((pass >> lane) & 1ull) != 0ull;
              uint32_t shift = 0u, before = 0u, among = 0u;
              for (uint64_t it = pass; it; it &= it - 1ull) {
                const int src = __ffsll((unsigned long long)it) - 1;
                const double cd = wave_read(d, src);
                const uint32_t corig = (uint32_t)__builtin_amdgcn_readlane((int)orig, src);
                const bool entry_first = ldist < cd || (ldist == cd && lorig < corig);      // my list entry comes before it
                const uint32_t n_first = (uint32_t)__popcll(__ballot(lane < KM && entry_first));
                before = lane == src ? n_first : before;
                shift += entry_first ? 0u : 1u;
                among += (cd < d || (cd == d && corig < orig)) ? 1u : 0u;                 // it comes before my candidate
              }
              const uint32_t to_e = (uint32_t)lane + shift, to_c = before + among;
              const int dst_e = 4 * (int)(lane < KM && to_e < (uint32_t)KM ? to_e : 63u);
              const int dst_c = 4 * (int)(mine && to_c < (uint32_t)KM ? to_c : 63u);
              const long long db = __double_as_longlong(d), lb = __double_as_longlong(ldist);
              const uint32_t n_lo = (uint32_t)__builtin_amdgcn_ds_permute(dst_e, (int)(uint32_t)lb) | (uint32_t)__builtin_amdgcn_ds_permute(dst_c, (int)(uint32_t)db);
              const uint32_t n_hi = (uint32_t)__builtin_amdgcn_ds_permute(dst_e, (int)(uint32_t)(lb >> 32)) |
                (uint32_t)__builtin_amdgcn_ds_permute(dst_c, (int)(uint32_t)(db >> 32));
              const uint32_t n_i = (uint32_t)__builtin_amdgcn_ds_permute(dst_e, (int)lidx) | (uint32_t)__builtin_amdgcn_ds_permute(dst_c, (int)at);
              const uint32_t n_o = (uint32_t)__builtin_amdgcn_ds_permute(dst_e, (int)lorig) | (uint32_t)__builtin_amdgcn_ds_permute(dst_c, (int)orig);
              if (lane < KM) {ldist = __longlong_as_double((long long)(((uint64_t)n_hi << 32) | n_lo)); lidx = n_i; lorig = n_o;}
              set_bar(bar, bar_orig);
              pass = 0;
            }
            while (pass) {
              const int src = __ffsll((unsigned long long)pass) - 1;
              pass &= pass - 1;
              const double cd = wave_read(d, src);
              const uint32_t cat = (uint32_t)__builtin_amdgcn_readlane((int)at, src), corig = (uint32_t)__builtin_amdgcn_readlane((int)orig, src);
              if (!(cd < bar || (cd == bar && corig < bar_orig))) {continue;}         // the bar has moved since the ballot
              const uint64_t later = __ballot(lane < KM && (cd < ldist || (cd == ldist && corig < lorig))) & 0xFFFFull;
              const int place = __ffsll((unsigned long long)later) - 1;                 // the first entry the point comes before
              const double up_d = __shfl_up(ldist, 1, 64);
              const uint32_t up_i = (uint32_t)__shfl_up((int)lidx, 1, 64), up_o = (uint32_t)__shfl_up((int)lorig, 1, 64);
              const bool shifts = lane > place, lands = lane == place;
              ldist = shifts ? up_d : (lands ? cd : ldist);
              lidx = shifts ? up_i : (lands ? cat : lidx);
              lorig = shifts ? up_o : (lands ? corig : lorig);
              set_bar(bar, bar_orig);
            }
          }
        }
      };
    const int ny_c = cube.yhi - cube.ylo + 1, n_rows = ny_c * (cube.zhi - cube.zlo + 1);
    if (cube.xlo <= cube.xhi && n_rows <= 64) {
      // Lane r holds row r of the cube (a run of cells along x = one run of points): both ends of its run, fetched by all lanes
      // at once, and the squared distance below which none of its points can lie (its gaps to the query along y and z; 1e-7
      // cells of slack as in GridCube::done).  The rows are visited nearest first -- the query's own row sets a low bar at
      // once -- and the visit ends when the nearest row left lies beyond the bar: with 1 m cells and 15 neighbours within
      // half a metre that is most of the 9 rows.  Exact: a row is left out only if each of its points is farther than the
      // 16th nearest found so far.
      const int rz = lane / ny_c, ry = lane - rz * ny_c;
      bool has = lane < n_rows;
      const double y0 = (double)(cube.ylo + ry), z0 = (double)(cube.zlo + rz);
      const double gy = fmax(fmax(y0 - cube.uy, cube.uy - (y0 + 1.)) - 1e-7, 0.), gz = fmax(fmax(z0 - cube.uz, cube.uz - (z0 + 1.)) - 1e-7, 0.);
      const double far2 = (gy * gy + gz * gz) * (mi.h * mi.h);
      // With a bound from the caller, what is left of it after the row's gaps along y and z limits the cells of the row worth
      // reading along x as well (cell c holds nothing within the bound unless c - ux and ux - (c + 1) are both at most the
      // rest, in cells; 1e-6 cells of slack): the run's ends are fetched for those cells only.
      int xa = cube.xlo, xb = cube.xhi;
      if (bounded) {
        const double left2 = reach2 - far2;
        if (left2 < 0.) {
          has = false;
        } else {
          const double rx = sqrt(left2) * mi.inv_h + 1e-6;
          const double lo = fmin(fmax(ceil(cube.ux - 1. - rx), -1e9), 1e9), hi = fmin(fmax(floor(cube.ux + rx), -1e9), 1e9);
          xa = max(xa, (int)lo); xb = min(xb, (int)hi);
          has = has && xa <= xb;
        }
      }
      uint32_t a = 0u, b = 0u;
      if (has) {
        const size_t cell0 = ((size_t)(cube.zlo + rz) * mi.ny + (cube.ylo + ry)) * mi.nx;
        a = mi.start[cell0 + xa]; b = mi.start[cell0 + xb + 1];
      }
      uint64_t todo = __ballot(has && a != b);
      for (;;) {
        const uint64_t within = todo & __ballot(!(far2 > bar));      // the rows the bar has not passed by
        if (within == 0ull) {break;}
        int pick = __ffsll((unsigned long long)within) - 1;
        if (within & (within - 1ull)) {                               // more than one: the nearest of them
          double m = ((within >> lane) & 1ull) ? far2 : INFINITY;
#pragma unroll
          for (int off = 32; off >= 1; off >>= 1) {const double o = __shfl_xor(m, off, 64); m = o < m ? o : m;}
          pick = __ffsll((unsigned long long)(__ballot(far2 == m) & within)) - 1;
        }
        todo &= ~(1ull << pick);
        run((uint32_t)__builtin_amdgcn_readlane((int)a, pick), (uint32_t)__builtin_amdgcn_readlane((int)b, pick));
      }
    } else if (cube.xlo <= cube.xhi) {
      // (a cube of more than 64 rows: the query lies far outside the map or its surroundings are empty)
      for (int z = cube.zlo; z <= cube.zhi; z++) {
        for (int y = cube.ylo; y <= cube.yhi; y++) {
          const size_t cell0 = ((size_t)z * mi.ny + y) * mi.nx;
          run(mi.start[cell0 + cube.xlo], mi.start[cell0 + cube.xhi + 1]);
        }
      }
    }
    if (cube.done(mi, wave_read(ldist, (int)kk - 1))) {
      return;
    }
    cube.grow(mi);
  }
}

// The row(s) of one scan point from its k nearest map points, idx[] in order of distance (positions in mi.pts): an edge point's
// 3 x 7 Jacobian and residual at J_out[21] / R_out[3], a surface point's 1 x 7 row and residual at J_out[7] / R_out[1].
template<bool SURFACE>
__device__ __forceinline__ void row_from_neighbours(
  const MapPose & P, D3 p0, D3 q, uint32_t kk, const uint32_t (&idx)[kNearestMax], const float4 * __restrict__ map,
  double * __restrict__ J_out, double * __restrict__ R_out)
{
  constexpr int KM = kNearestMax;
  double d[12];
  drp_dq(P, p0, d);
  if (!SURFACE) {
    // mean and covariance of the k neighbours (edge.cpp:38-49), in order of distance
    double mx = 0., my = 0., mz = 0.;
#pragma unroll
    for (int j = 0; j < KM; j++) {
      if ((uint32_t)j < kk) {const float4 m4 = map[idx[j]]; mx += (double)m4.x; my += (double)m4.y; mz += (double)m4.z;}
    }
    const double nk = (double)kk;
    mx /= nk; my /= nk; mz /= nk;
    double c[6] = {0., 0., 0., 0., 0., 0.};
#pragma unroll
    for (int j = 0; j < KM; j++) {
      if ((uint32_t)j < kk) {
        const float4 m4 = map[idx[j]];
        const double ex = (double)m4.x - mx, ey = (double)m4.y - my, ez = (double)m4.z - mz;
        c[0] += ex * ex; c[1] += ex * ey; c[2] += ex * ez; c[3] += ey * ey; c[4] += ey * ez; c[5] += ez * ez;
      }
    }
#pragma unroll
    for (int a = 0; a < 6; a++) {c[a] /= nk;}
    const D3 u = principal_direction(c);
    const D3 p1{mx - u.x, my - u.y, mz - u.z}, p2{mx + u.x, my + u.y, mz + u.z};
    const D3 e = d3_sub(p2, p1);
    const double K[9] = {0., -e.z, e.y, e.z, 0., -e.x, -e.y, e.x, 0.};       // Hat(p2 - p1)
    double * J = J_out;
#pragma unroll
    for (int r = 0; r < 3; r++) {
#pragma unroll
      for (int cc = 0; cc < 4; cc++) {J[7 * r + cc] = K[3 * r] * d[cc] + K[3 * r + 1] * d[4 + cc] + K[3 * r + 2] * d[8 + cc];}
#pragma unroll
      for (int cc = 0; cc < 3; cc++) {J[7 * r + 4 + cc] = K[3 * r + cc];}
    }
    const D3 rr = d3_cross(d3_sub(q, p1), d3_sub(q, p2));                  // MakeEdgeResidual
    double * R = R_out;
    R[0] = rr.x; R[1] = rr.y; R[2] = rr.z;
  } else {
    // plane coefficients: least squares X w = -1 by Householder QR (surface.hpp:78-83, math.hpp:36-40)
    double X[KM][3], g[KM];
#pragma unroll
    for (int j = 0; j < KM; j++) {
      X[j][0] = 0.; X[j][1] = 0.; X[j][2] = 0.; g[j] = 0.;
      if ((uint32_t)j < kk) {const float4 m4 = map[idx[j]]; X[j][0] = (double)m4.x; X[j][1] = (double)m4.y; X[j][2] = (double)m4.z; g[j] = -1.0;}
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
      double norm = 0.;
#pragma unroll
      for (int r = 0; r < KM; r++) {if (r >= c) {norm += X[r][c] * X[r][c];}}       // rows >= kk hold zeros
      norm = sqrt(norm);
      const double alpha = X[c][c] > 0. ? -norm : norm;
      double v[KM];
      double vv = 0.;
#pragma unroll
      for (int r = 0; r < KM; r++) {v[r] = r >= c ? X[r][c] : 0.; if (r == c) {v[r] -= alpha;} vv += v[r] * v[r];}
      if (vv > 0.) {
#pragma unroll
        for (int cc = 0; cc < 3; cc++) {
          if (cc >= c) {
            double sdot = 0.;
#pragma unroll
            for (int r = 0; r < KM; r++) {sdot += v[r] * X[r][cc];}
            sdot = 2. * sdot / vv;
#pragma unroll
            for (int r = 0; r < KM; r++) {X[r][cc] -= sdot * v[r];}
          }
        }
        double sdot = 0.;
#pragma unroll
        for (int r = 0; r < KM; r++) {sdot += v[r] * g[r];}
        sdot = 2. * sdot / vv;
#pragma unroll
        for (int r = 0; r < KM; r++) {g[r] -= sdot * v[r];}
      }
    }
    // A neighbourhood without a plane (the k points coincide or lie on a line through ... anything that leaves R a zero
    // pivot): Eigen's householderQr().solve() divides by it all the same and hands back whatever comes out, which cannot
    // be reproduced here (its arithmetic is not in the image).  Such a row is given weight 0 instead -- the ZERO row
    // (residual 0, u = 0: it adds nothing to any of the optimizer's sums), which is also how a caller tells: every other
    // surface row has |u| = 1.
    const double r0 = fabs(X[0][0]), r1 = fabs(X[1][1]), r2 = fabs(X[2][2]);
    const double rmax = fmax(r0, fmax(r1, r2));
    const bool no_plane = !(fmin(r0, fmin(r1, r2)) > 1e-9 * rmax);
    double w[3];
    w[2] = g[2] / X[2][2];
    w[1] = (g[1] - X[1][2] * w[2]) / X[1][1];
    w[0] = (g[0] - X[0][1] * w[1] - X[0][2] * w[2]) / X[0][0];
    const double norm = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    const double u[3] = {no_plane ? 0. : w[0] / norm, no_plane ? 0. : w[1] / norm, no_plane ? 0. : w[2] / norm};
    double * J = J_out;
#pragma unroll
    for (int cc = 0; cc < 4; cc++) {J[cc] = u[0] * d[cc] + u[1] * d[4 + cc] + u[2] * d[8 + cc];}   // MakeJacobianRow, surface.hpp:85-93
    J[4] = u[0]; J[5] = u[1]; J[6] = u[2];
    R_out[0] = no_plane ? 0. : (w[0] * q.x + w[1] * q.y + w[2] * q.z + 1.0) / norm;         // SignedPointPlaneDistance
  }
}

// how a query finds its neighbours
enum : int {kSearchWholeMap = 0, kSearchGridWave = 2};

// A surface row whose neighbourhood spans a plane (|u| = 1; the others are zero rows, row_from_neighbours) is counted in its
// scan's state: align_scale_kernel ends a scan that HAS surface points but not one such row with a status of its own instead
// of letting rows that say nothing read as "converged" (the reference's arithmetic yields NaN there and fails differently).
__device__ inline void count_surface_row_with_a_plane(const AlignState * __restrict__ align, uint32_t s)
{
  (void)__hip_atomic_fetch_add(const_cast<int32_t *>(&align[s].surface_rows_with_plane), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the rows of one workgroup: `bx` = its index along x among the workgroups of its kind
template<bool SURFACE, int SEARCH>
__device__ __forceinline__ void scan_to_map_rows(
  uint32_t bx, const MapIndex & mi, MapPose P, uint32_t k, const float4 * __restrict__ pts,
  const uint32_t * __restrict__ begin, const uint32_t * __restrict__ count, uint32_t count_stride,
  double * __restrict__ residual, double * __restrict__ jacobian, const AlignState * __restrict__ align,
  const uint32_t * __restrict__ row_begin /* where the scan's rows start; null: where its points do */)
{
  constexpr int KM = kNearestMax, T = 128;
  const uint32_t s = blockIdx.y, tid = threadIdx.x;
  if (align) {                                         // inside lfx_scan_to_map_align: this scan's current pose
    if (align[s].done) {return;}
    P = align[s].pose;
  }
  const float4 * __restrict__ map = mi.pts;
  const uint32_t b = begin[s], n = count[(size_t)s * count_stride];
  const uint32_t rb = row_begin ? row_begin[s] : b;
  // kSearchGridWave: a workgroup is one wave and has one query, the same in every lane
  const uint32_t i = SEARCH == kSearchGridWave ? bx : bx * T + tid;
  if ((SEARCH == kSearchGridWave ? bx : bx * T) >= n) {return;}                        // the whole workgroup is beyond this cloud
  const bool valid = i < n;
  const float4 pf = pts[b + (valid ? i : 0u)];
  const D3 p0{(double)pf.x, (double)pf.y, (double)pf.z};
  const D3 q{P.m[0] * p0.x + P.m[1] * p0.y + P.m[2] * p0.z + P.m[3], P.m[4] * p0.x + P.m[5] * p0.y + P.m[6] * p0.z + P.m[7],
    P.m[8] * p0.x + P.m[9] * p0.y + P.m[10] * p0.z + P.m[11]};
  const uint32_t kk = k < (uint32_t)KM ? k : (uint32_t)KM;
  double dist[KM];
  uint32_t idx[KM];
  if (SEARCH == kSearchGridWave) {
    __shared__ double list_d[KM];
    __shared__ uint32_t list_i[KM];
    double ld;
    uint32_t li;
    nearest_in_grid_wave(mi, q, kk, ld, li);
    if (tid < (uint32_t)KM) {list_d[tid] = ld; list_i[tid] = li;}
    __syncthreads();
    if (tid != 0) {return;}                              // one lane does the rest
#pragma unroll
    for (int j = 0; j < KM; j++) {dist[j] = list_d[j]; idx[j] = list_i[j];}
  } else {
#pragma unroll
    for (int j = 0; j < KM; j++) {dist[j] = INFINITY; idx[j] = 0u;}
    __shared__ float4 tile[T];
    nearest_whole_map<T>(mi, q, dist, idx, tile);
  }
  if (!valid) {return;}
  if (SURFACE) {
    double * J = jacobian + 7 * (size_t)(rb + i);
    row_from_neighbours<true>(P, p0, q, kk, idx, map, J, residual + (size_t)(rb + i));
    if (align && (J[4] != 0. || J[5] != 0. || J[6] != 0.)) {count_surface_row_with_a_plane(align, s);}
  } else {
    row_from_neighbours<false>(P, p0, q, kk, idx, map, jacobian + 21 * (size_t)(rb + i), residual + 3 * (size_t)(rb + i));
  }
}

template<bool SURFACE, int SEARCH>
__global__ __launch_bounds__(128) void scan_to_map_kernel(
  MapIndex mi, MapPose P, uint32_t k, const float4 * __restrict__ pts,
  const uint32_t * __restrict__ begin, const uint32_t * __restrict__ count, uint32_t count_stride,
  double * __restrict__ residual, double * __restrict__ jacobian, const AlignState * __restrict__ align,
  const uint32_t * __restrict__ row_begin)
{
  scan_to_map_rows<SURFACE, SEARCH>(blockIdx.x, mi, P, k, pts, begin, count, count_stride, residual, jacobian, align, row_begin);
}

// Problem::Make of the localizer (loam_optimization_problem.hpp:62-84: edge rows and surface rows of the same scans against
// their two maps) in two launches.  The SEARCH: one wave per query, workgroups [0, x_edge) along x take edge points, the rest
// surface points (the short surface part runs beside the edge part instead of after it); it leaves the places of the k
// nearest map points in `nbr`, 16 words per row, and needs few registers -- every query of a scan is resident at once.  The
// ROWS: one THREAD per query turns the neighbours into Jacobian and residual.  (In one kernel -- the search by the wave, the
// row by its lane 0 -- the row's arithmetic ran at a 64th of the machine's rate in every one of the waves and its registers
// (the 16 x 3 QR of a surface row) halved the number of searches in flight: 67 us per iteration for one 64 x 1800 scan.)
struct RowsOfKind
{
  MapIndex mi;
  const float4 * pts;
  const uint32_t * begin, * count;
  uint32_t count_stride;
  double * residual, * jacobian;
  const uint32_t * row_begin;
  uint32_t * nbr;                          // [rows][kNearestMax]
  double * reach;                          // [rows] the distance of each row's last neighbour at the previous iteration, squared
};

__device__ __forceinline__ D3 to_map(const MapPose & P, D3 p)
{
  return {P.m[0] * p.x + P.m[1] * p.y + P.m[2] * p.z + P.m[3], P.m[4] * p.x + P.m[5] * p.y + P.m[6] * p.z + P.m[7],
    P.m[8] * p.x + P.m[9] * p.y + P.m[10] * p.z + P.m[11]};
}

constexpr int kSearchWaves = 4;          // queries per workgroup (nothing is shared between them: a workgroup of one wave each made
                                         // the dispatch of 4 500 workgroups the longest part of a single scan's search)
__global__ __launch_bounds__(64 * kSearchWaves) void map_search_kernel(
  RowsOfKind edge, RowsOfKind surface, uint32_t x_edge /* workgroups of edge queries */, uint32_t k, const AlignState * __restrict__ align,
  int iter)
{
  const uint32_t s = blockIdx.y;
  const bool surf = blockIdx.x >= x_edge;
  const RowsOfKind & R = surf ? surface : edge;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
  // (everything the wave needs before its first point, asked for at once: a wave's time is a chain of round trips to memory)
  const int32_t done = align[s].done;
  const MapPose P = align[s].pose;
  double prev[12];
#pragma unroll
  for (int a = 0; a < 12; a++) {prev[a] = align[s].prev_m[a];}
  // (the launch is sized from what the host knows of the clouds' lengths -- a bound, or what the previous call saw: the
  // workgroups of a kind stride over its queries)
  const uint32_t n = R.count[(size_t)s * R.count_stride], stride = (surf ? gridDim.x - x_edge : x_edge) * kSearchWaves;
  const uint32_t b = R.begin[s], rb = R.row_begin ? R.row_begin[s] : b;
  if (done) {return;}
  const uint32_t kk = k < (uint32_t)kNearestMax ? k : (uint32_t)kNearestMax;
  for (uint32_t i = (surf ? blockIdx.x - x_edge : blockIdx.x) * kSearchWaves + wave; i < n; i += stride) {
    const float4 pf = R.pts[b + i];
    const bool again = iter > 0;
    double was = 0.;
    if (again) {was = R.reach[rb + i];}
    const D3 p0{(double)pf.x, (double)pf.y, (double)pf.z};
    const D3 q = to_map(P, p0);
    // After the first iteration: the 16 points that were nearest then lay within sqrt(was) of where the query was; it has
    // moved by |q - q_before| since, so 16 points lie within the sum of the two now (the triangle inequality; a part in
    // 10^9 and 10^-12 on top for the roundings of this very estimate).  The search starts with that as its bar.
    double reach2 = 0.;
    if (again) {
      const D3 qb{prev[0] * p0.x + prev[1] * p0.y + prev[2] * p0.z + prev[3], prev[4] * p0.x + prev[5] * p0.y + prev[6] * p0.z + prev[7],
        prev[8] * p0.x + prev[9] * p0.y + prev[10] * p0.z + prev[11]};
      const D3 dq = d3_sub(q, qb);
      const double r = (sqrt(was) + sqrt(d3_dot(dq, dq))) * (1. + 1e-9) + 1e-12;
      reach2 = r * r;
    }
    double ld;
    uint32_t li;
    nearest_in_grid_wave(R.mi, q, kk, ld, li, again, reach2);
    if (lane < (uint32_t)kNearestMax) {R.nbr[(size_t)(rb + i) * kNearestMax + lane] = li;}
    if (lane == (uint32_t)kNearestMax - 1u) {R.reach[rb + i] = ld;}
  }
}

constexpr int kRowThreads = 64;
__global__ __launch_bounds__(kRowThreads) void rows_from_neighbours_kernel(
  RowsOfKind edge, RowsOfKind surface, uint32_t x_edge /* workgroups of edge rows */, uint32_t k, const AlignState * __restrict__ align)
{
  const uint32_t s = blockIdx.y;
  const bool surf = blockIdx.x >= x_edge;
  const RowsOfKind & R = surf ? surface : edge;
  // (the scan's state and extents asked for together, as in map_search_kernel)
  const int32_t done = align[s].done;
  const MapPose P = align[s].pose;
  const uint32_t n = R.count[(size_t)s * R.count_stride], stride = (surf ? gridDim.x - x_edge : x_edge) * kRowThreads;
  const uint32_t b = R.begin[s], rb = R.row_begin ? R.row_begin[s] : b;
  if (done) {return;}
  const uint32_t kk = k < (uint32_t)kNearestMax ? k : (uint32_t)kNearestMax;
  for (uint32_t i = (surf ? blockIdx.x - x_edge : blockIdx.x) * kRowThreads + threadIdx.x; i < n; i += stride) {
    const float4 pf = R.pts[b + i];
    const D3 p0{(double)pf.x, (double)pf.y, (double)pf.z};
    const D3 q = to_map(P, p0);
    uint32_t idx[kNearestMax];
    const uint4 * src = reinterpret_cast<const uint4 *>(R.nbr + (size_t)(rb + i) * kNearestMax);
#pragma unroll
    for (int j = 0; j < kNearestMax / 4; j++) {
      const uint4 v = src[j];
      idx[4 * j] = v.x; idx[4 * j + 1] = v.y; idx[4 * j + 2] = v.z; idx[4 * j + 3] = v.w;
    }
    if (surf) {
      double * J = R.jacobian + 7 * (size_t)(rb + i);
      row_from_neighbours<true>(P, p0, q, kk, idx, R.mi.pts, J, R.residual + (size_t)(rb + i));
      if (J[4] != 0. || J[5] != 0. || J[6] != 0.) {count_surface_row_with_a_plane(align, s);}
    } else {
      row_from_neighbours<false>(P, p0, q, kk, idx, R.mi.pts, R.jacobian + 21 * (size_t)(rb + i), R.residual + 3 * (size_t)(rb + i));
    }
  }
}

// ------------------------------------------------------------------------------------------
// Building a MapIndex (lfx_map_create): bounds, points per cell, the cells' first points by an exclusive scan, then the
// points into their cells.  The order of the points inside a cell is whatever the atomics give; nothing depends on it
// (the search orders equal distances by the original index, which travels in the w of every sorted point).
__device__ inline uint32_t float_order(float f)         // unsigned ints that order like the floats
{
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(256) void map_bounds_kernel(const float4 * __restrict__ pts, uint32_t n, uint32_t * __restrict__ bounds /* min xyz, max xyz */)
{
  uint32_t lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float4 p = pts[i];
    const uint32_t v[3] = {float_order(p.x), float_order(p.y), float_order(p.z)};
#pragma unroll
    for (int a = 0; a < 3; a++) {lo[a] = min(lo[a], v[a]); hi[a] = max(hi[a], v[a]);}
  }
#pragma unroll
  for (int a = 0; a < 3; a++) {
    for (int off = 32; off >= 1; off >>= 1) {
      lo[a] = min(lo[a], (uint32_t)__shfl_xor((int)lo[a], off, 64));
      hi[a] = max(hi[a], (uint32_t)__shfl_xor((int)hi[a], off, 64));
    }
    if ((threadIdx.x & 63) == 0) {atomicMin(&bounds[a], lo[a]); atomicMax(&bounds[3 + a], hi[a]);}
  }
}

__device__ inline size_t map_cell_of(const MapIndex & mi, float4 p)
{
  const int ix = min(max(cell_coordinate((double)p.x, mi.ox, mi.inv_h), 0), mi.nx - 1);
  const int iy = min(max(cell_coordinate((double)p.y, mi.oy, mi.inv_h), 0), mi.ny - 1);
  const int iz = min(max(cell_coordinate((double)p.z, mi.oz, mi.inv_h), 0), mi.nz - 1);
  return ((size_t)iz * mi.ny + iy) * mi.nx + ix;
}

__global__ __launch_bounds__(256) void map_count_kernel(MapIndex mi, const float4 * __restrict__ pts, uint32_t * __restrict__ cell_count)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < mi.n) {atomicAdd(&cell_count[map_cell_of(mi, pts[i])], 1u);}
}

constexpr int kScanThreads = 1024, kScanItems = 4 * kScanThreads;
// exclusive scan of one value per thread over the workgroup; sh: kScanThreads words; returns the thread's offset, total in `total`
__device__ inline uint32_t workgroup_exclusive_scan(uint32_t v, uint32_t * sh, uint32_t & total)
{
  const int tid = threadIdx.x;
  uint32_t incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t o = (uint32_t)__shfl_up((int)incl, off, 64);
    if ((tid & 63) >= off) {incl += o;}
  }
  if ((tid & 63) == 63) {sh[tid >> 6] = incl;}
  __syncthreads();
  if (tid < 64) {
    const uint32_t w = tid < kScanThreads / 64 ? sh[tid] : 0u;
    uint32_t wi = w;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t o = (uint32_t)__shfl_up((int)wi, off, 64);
      if (tid >= off) {wi += o;}
    }
    if (tid < kScanThreads / 64) {sh[64 + tid] = wi - w;}
    if (tid == kScanThreads / 64 - 1) {sh[128] = wi;}
  }
  __syncthreads();
  total = sh[128];
  const uint32_t r = sh[64 + (tid >> 6)] + incl - v;
  __syncthreads();
  return r;
}

__global__ __launch_bounds__(kScanThreads) void cell_block_sum_kernel(const uint32_t * __restrict__ cell_count, size_t cells, uint32_t * __restrict__ partial)
{
  __shared__ uint32_t sh[kScanThreads];
  const size_t base = (size_t)blockIdx.x * kScanItems + 4 * (size_t)threadIdx.x;
  uint32_t v = 0;
#pragma unroll
  for (int a = 0; a < 4; a++) {if (base + a < cells) {v += cell_count[base + a];}}
  uint32_t total;
  (void)workgroup_exclusive_scan(v, sh, total);
  if (threadIdx.x == 0) {partial[blockIdx.x] = total;}
}

__global__ __launch_bounds__(kScanThreads) void cell_partial_scan_kernel(uint32_t * __restrict__ partial, uint32_t n_blocks)
{
  __shared__ uint32_t sh[kScanThreads];
  const uint32_t per = (n_blocks + kScanThreads - 1) / kScanThreads;
  const uint32_t a = threadIdx.x * per, b = min(a + per, n_blocks);
  uint32_t v = 0;
  for (uint32_t i = a; i < b; i++) {v += partial[i];}
  uint32_t total;
  uint32_t run = workgroup_exclusive_scan(v, sh, total);
  for (uint32_t i = a; i < b; i++) {const uint32_t c = partial[i]; partial[i] = run; run += c;}
}

__global__ __launch_bounds__(kScanThreads) void cell_start_kernel(
  const uint32_t * __restrict__ cell_count, size_t cells, const uint32_t * __restrict__ partial, uint32_t * __restrict__ start, uint32_t n_points)
{
  __shared__ uint32_t sh[kScanThreads];
  const size_t base = (size_t)blockIdx.x * kScanItems + 4 * (size_t)threadIdx.x;
  uint32_t c[4];
  uint32_t v = 0;
#pragma unroll
  for (int a = 0; a < 4; a++) {c[a] = base + a < cells ? cell_count[base + a] : 0u; v += c[a];}
  uint32_t total;
  uint32_t run = partial[blockIdx.x] + workgroup_exclusive_scan(v, sh, total);
#pragma unroll
  for (int a = 0; a < 4; a++) {if (base + a < cells) {start[base + a] = run;} run += c[a];}
  if (blockIdx.x == 0 && threadIdx.x == 0) {start[cells] = n_points;}
}

__global__ __launch_bounds__(256) void map_scatter_kernel(
  MapIndex mi, const float4 * __restrict__ pts, uint32_t * __restrict__ cell_count, const uint32_t * __restrict__ start, float4 * __restrict__ sorted)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= mi.n) {return;}
  const float4 p = pts[i];
  const size_t cell = map_cell_of(mi, p);
  const uint32_t slot = atomicSub(&cell_count[cell], 1u) - 1u;
  sorted[start[cell] + slot] = make_float4(p.x, p.y, p.z, __uint_as_float(i));
}

// KDTreeEigen::NearestKSearch (localization/src/kdtree.cpp:44-68) for a batch of queries: per query the k nearest points
// of the map, ascending; neighbours [n][k][3] doubles (GetRows of the map), squared distances [n][k], indices [n][k]
// into the map as it was given.
template<int SEARCH>
__global__ __launch_bounds__(128) void map_nearest_kernel(
  MapIndex mi, const double * __restrict__ queries, uint32_t n, uint32_t k, double * __restrict__ neighbours,
  double * __restrict__ squared_distances, uint32_t * __restrict__ indices)
{
  constexpr int KM = kNearestMax, T = 128;
  const uint32_t i = SEARCH == kSearchGridWave ? blockIdx.x : blockIdx.x * T + threadIdx.x;
  const bool valid = i < n;
  const size_t at = valid ? i : 0u;
  const D3 q{queries[3 * at], queries[3 * at + 1], queries[3 * at + 2]};
  const uint32_t kk = k < (uint32_t)KM ? k : (uint32_t)KM;
  double dist[KM];
  uint32_t idx[KM];
  constexpr bool GRID = SEARCH != kSearchWholeMap;
  if (SEARCH == kSearchGridWave) {
    __shared__ double list_d[KM];
    __shared__ uint32_t list_i[KM];
    double ld;
    uint32_t li;
    nearest_in_grid_wave(mi, q, kk, ld, li);
    if (threadIdx.x < (uint32_t)KM) {list_d[threadIdx.x] = ld; list_i[threadIdx.x] = li;}
    __syncthreads();
    if (threadIdx.x != 0) {return;}
#pragma unroll
    for (int j = 0; j < KM; j++) {dist[j] = list_d[j]; idx[j] = list_i[j];}
  } else {
#pragma unroll
    for (int j = 0; j < KM; j++) {dist[j] = INFINITY; idx[j] = 0u;}
    __shared__ float4 tile[T];
    nearest_whole_map<T>(mi, q, dist, idx, tile);
  }
  if (!valid) {return;}
#pragma unroll
  for (int j = 0; j < KM; j++) {
    if ((uint32_t)j < kk) {
      const float4 m4 = mi.pts[idx[j]];
      const size_t o = (size_t)i * kk + j;
      if (neighbours) {neighbours[3 * o] = (double)m4.x; neighbours[3 * o + 1] = (double)m4.y; neighbours[3 * o + 2] = (double)m4.z;}
      if (squared_distances) {squared_distances[o] = dist[j];}
      if (indices) {indices[o] = GRID ? __float_as_uint(m4.w) : idx[j];}
    }
  }
}

// lfx_localize_batch: where voxel_downsample_kernel gave a cloud back unfiltered (status 1, nothing written), Downsample
// returns the input cloud (downsample.hpp:37-51 -> pcl::VoxelGrid::applyFilter), so the rows are built from all its points.
__global__ void downsample_passthrough_kernel(
  const float4 * __restrict__ pts, const uint32_t * __restrict__ begin, const uint32_t * __restrict__ count, uint32_t count_stride,
  float4 * __restrict__ out, uint32_t * __restrict__ out_count, const uint32_t * __restrict__ status,
  const uint32_t * __restrict__ edge_count, uint32_t * __restrict__ lengths /* pinned host memory or null: [scans][2] */)
{
  const uint32_t s = blockIdx.x;
  const uint32_t n = count[(size_t)s * count_stride], b = begin[s];
  // (what the next call sizes its launches by: this scan's edge points and downsampled surface points)
  if (lengths && threadIdx.x == 0) {lengths[2 * s] = edge_count[(size_t)s * count_stride]; lengths[2 * s + 1] = status[s] == 0u ? out_count[s] : n;}
  if (status[s] == 0u) {return;}
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
    const float4 p = pts[b + i];
    out[b + i] = make_float4(p.x, p.y, p.z, 1.f);
  }
  if (threadIdx.x == 0) {out_count[s] = n;}
}

// ------------------------------------------------------------------------------------------
// The optimizer around the rows: Optimizer::Run (localization/include/lidar_feature_localization/optimizer.hpp:79-123)
// as kernels, so that the iterations of a batch of scans run without a round trip to the host: align_begin_kernel, then
// per iteration the two row builds above, align_scale_kernel (errors, robust scale, weights) and align_update_kernel (the
// sums of WeightedUpdate, the 6 x 6 solve, the pose update and the three stopping tests); a finished scan's kernels return
// at once.  PARITY UNPINNED (Eigen's arithmetic; sums are taken in a fixed tree order here, not row by row).
constexpr int kAlignThreads = 256, kAlignKeysLds = 6144;
enum AlignCode : int32_t {kAlignConverged = 0, kAlignLargerError = 1, kAlignLargerScale = 2, kAlignMaxIteration = 3, kAlignEmpty = 4, kAlignNoPlane = 5};

// Eigen::Quaterniond(Matrix3d): the branch on the trace, then on the largest diagonal entry
__device__ inline void quaternion_of_rotation(const double (&m)[12], double & w, double (&v)[3])
{
  double t = m[0] + m[5] + m[10];
  if (t > 0.) {
    t = sqrt(t + 1.0);
    w = 0.5 * t;
    t = 0.5 / t;
    v[0] = (m[9] - m[6]) * t; v[1] = (m[2] - m[8]) * t; v[2] = (m[4] - m[1]) * t;
  } else {
    int i = 0;
    if (m[5] > m[0]) {i = 1;}
    if (m[10] > m[5 * i]) {i = 2;}
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = sqrt(m[5 * i] - m[5 * j] - m[5 * k] + 1.0);
    v[i] = 0.5 * t;
    t = 0.5 / t;
    w = (m[4 * k + j] - m[4 * j + k]) * t;
    v[j] = (m[4 * j + i] + m[4 * i + j]) * t;
    v[k] = (m[4 * k + i] + m[4 * i + k]) * t;
  }
}

// MakePose (posevec.cpp:47-55: q.toRotationMatrix(), q as it is) followed by what every Problem::Make starts with
__device__ inline void refresh_pose(AlignState & a)
{
  const double w = a.q[0], x = a.q[1], y = a.q[2], z = a.q[3];
  const double tx = 2. * x, ty = 2. * y, tz = 2. * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
  double (&m)[12] = a.pose.m;
  m[0] = 1. - (tyy + tzz); m[1] = txy - twz; m[2] = txz + twy; m[3] = a.t[0];
  m[4] = txy + twz; m[5] = 1. - (txx + tzz); m[6] = tyz - twx; m[7] = a.t[1];
  m[8] = txz - twy; m[9] = tyz + twx; m[10] = 1. - (txx + tyy); m[11] = a.t[2];
  double qw, v[3];
  quaternion_of_rotation(m, qw, v);
  a.pose.qw = qw; a.pose.qx = v[0]; a.pose.qy = v[1]; a.pose.qz = v[2];
}

// (initial: the caller's poses where the host put them -- pinned host memory, read from there; out: the results' place, likewise)
__global__ void align_begin_kernel(AlignState * __restrict__ states, const double * __restrict__ initial /* [n][12] */, uint32_t n,
  uint32_t * __restrict__ active, uint32_t * __restrict__ tickets, AlignOut * __restrict__ out)
{
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s == 0) {*active = n;}                            // scans still iterating
  if (s >= n) {return;}
  tickets[s] = 0u;
  out[s].done = 0;
  AlignState a;
  double m[12];
  for (int i = 0; i < 12; i++) {m[i] = initial[12 * (size_t)s + i];}
  double w, v[3];
  quaternion_of_rotation(m, w, v);                     // Eigen::Quaterniond q(initial_pose.linear())
  a.q[0] = w; a.q[1] = v[0]; a.q[2] = v[1]; a.q[3] = v[2];
  a.t[0] = m[3]; a.t[1] = m[7]; a.t[2] = m[11];
  a.prev_error = 1.7976931348623157e308; a.prev_scale = 1.7976931348623157e308;   // std::numeric_limits<double>::max()
  a.error = 0.; a.scale = 0.; a.cur_error = 0.; a.cur_scale = 0.; a.iteration = 0; a.code = kAlignMaxIteration; a.done = 0; a.surface_rows_with_plane = 0;
  refresh_pose(a);
  for (int i = 0; i < 12; i++) {a.prev_m[i] = a.pose.m[i];}
  states[s] = a;
}

// the end of a scan's Optimizer::Run: its OptimizationResult (optimization_result.hpp:43-79) and pose, for the kernels still
// to come (done) and for the host
__device__ inline void align_finish(AlignState & st, AlignOut & o, int iteration, double error, double scale, int code, uint32_t * active)
{
  st.iteration = iteration; st.error = error; st.scale = scale; st.code = code; st.done = 1;
  atomicSub(active, 1u);
  for (int i = 0; i < 12; i++) {o.pose[i] = st.pose.m[i];}
  o.error = error; o.scale = scale; o.iteration = iteration; o.code = code;
  __threadfence_system();
  *reinterpret_cast<volatile int32_t *>(&o.done) = 1;
}

// AlignmentProblem::Make (alignment.cpp:33-78), the problem the reference's optimizer tests run: rows [DRpDq(q, x), I],
// residual pose * x - y.  X, Y: [n][3] doubles of cloud s from record begin[s].
__global__ __launch_bounds__(128) void pair_rows_kernel(
  const double * __restrict__ X, const double * __restrict__ Y, const uint32_t * __restrict__ begin,
  const uint32_t * __restrict__ count, double * __restrict__ residual, double * __restrict__ jacobian,
  const AlignState * __restrict__ align)
{
  const uint32_t s = blockIdx.y;
  if (align[s].done) {return;}
  const uint32_t i = blockIdx.x * 128u + threadIdx.x;
  if (i >= count[s]) {return;}
  const MapPose P = align[s].pose;
  const size_t at = (size_t)begin[s] + i;
  const D3 x{X[3 * at], X[3 * at + 1], X[3 * at + 2]};
  double d[12];
  drp_dq(P, x, d);
  double * J = jacobian + 21 * at;
#pragma unroll
  for (int r = 0; r < 3; r++) {
#pragma unroll
    for (int c = 0; c < 4; c++) {J[7 * r + c] = d[4 * r + c];}
#pragma unroll
    for (int c = 0; c < 3; c++) {J[7 * r + 4 + c] = r == c ? 1. : 0.;}
  }
  residual[3 * at] = P.m[0] * x.x + P.m[1] * x.y + P.m[2] * x.z + P.m[3] - Y[3 * at];
  residual[3 * at + 1] = P.m[4] * x.x + P.m[5] * x.y + P.m[6] * x.z + P.m[7] - Y[3 * at + 1];
  residual[3 * at + 2] = P.m[8] * x.x + P.m[9] * x.y + P.m[10] * x.z + P.m[11] - Y[3 * at + 2];
}

// k-th smallest (0-based) of the n non-negative doubles v[0..n): most-significant-byte-first radix selection over their bit
// patterns (non-negative doubles order like their bits), a 256-bin histogram in LDS per byte.  Every thread of the
// workgroup calls it and gets the value.  sh: kSelectWords words of LDS, the first 256 ZERO on entry and zero again on
// return.  Two barriers per byte: the counts by all threads | wave 0 reads the 256 bins (four per lane), clears them for
// the next byte, finds the bin that holds rank k by a prefix sum along its lanes and publishes it | everyone reads that.
constexpr int kSelectWords = 576;         // 256 bins | 8 words of result | 48 words of the median's own | 256 "who counted here"

__device__ inline double workgroup_select(const double * __restrict__ v, uint32_t n, uint32_t k, uint32_t * sh)
{
  const int tid = threadIdx.x, T = blockDim.x;
  uint32_t * who = sh + 320;
  uint64_t prefix = 0, mask = 0;
  for (int shift = 56; shift >= 0; shift -= 8) {
    // (the counts: a wave whose lanes all hold the same byte adds their number in ONE atomic -- in the high bytes all the
    // values of a scan share two or three bins, and four thousand atomics on one LDS address take their turns.  Whoever
    // counts in a bin leaves its value's place there: a bin that ends with one value names it without another pass)
    for (uint32_t i0 = 0; i0 < n; i0 += T) {
      const uint32_t i = i0 + (uint32_t)tid;
      const uint64_t key = (uint64_t)__double_as_longlong(v[i < n ? i : 0u]);
      const bool match = i < n && (key & mask) == prefix;
      const uint32_t digit = (uint32_t)(key >> shift) & 255u;
      const uint64_t todo = __ballot(match);
      if (todo) {
        const int first = __ffsll((unsigned long long)todo) - 1;
        const uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)digit, first);
        if (__ballot(match && digit == d) == todo) {            // one bin for the whole wave: the high bytes
          if ((tid & 63) == first) {atomicAdd(&sh[d], (uint32_t)__popcll(todo)); who[d] = i;}
        } else if (match) {
          atomicAdd(&sh[digit], 1u);
          who[digit] = i;
        }
      }
    }
    __syncthreads();
    if (tid < 64) {
      const uint32_t h0 = sh[4 * tid], h1 = sh[4 * tid + 1], h2 = sh[4 * tid + 2], h3 = sh[4 * tid + 3];
      sh[4 * tid] = 0u; sh[4 * tid + 1] = 0u; sh[4 * tid + 2] = 0u; sh[4 * tid + 3] = 0u;
      const uint32_t incl = wave_inclusive_sum(h0 + h1 + h2 + h3);
      const uint32_t c0 = incl - (h0 + h1 + h2 + h3), c1 = c0 + h0, c2 = c1 + h1, c3 = c2 + h2;
      if (k >= c0 && k < incl) {                                // exactly one lane
        const uint32_t which = k < c1 ? 0u : (k < c2 ? 1u : (k < c3 ? 2u : 3u));
        const uint32_t bin = 4u * (uint32_t)tid + which;
        const uint64_t only = (uint64_t)__double_as_longlong(v[who[bin] < n ? who[bin] : 0u]);   // (the answer if the bin holds one value)
        sh[256] = bin;
        sh[257] = k - (which == 0u ? c0 : (which == 1u ? c1 : (which == 2u ? c2 : c3)));
        sh[258] = which == 0u ? h0 : (which == 1u ? h1 : (which == 2u ? h2 : h3));
        sh[259] = (uint32_t)only; sh[260] = (uint32_t)(only >> 32);
      }
    }
    __syncthreads();
    prefix |= (uint64_t)sh[256] << shift;
    mask |= 0xFFull << shift;
    k = sh[257];
    if (sh[258] == 1u) {                                        // one value carries this prefix: it is the answer
      return __longlong_as_double((long long)(((uint64_t)sh[260] << 32) | sh[259]));
    }
  }
  return __longlong_as_double((long long)prefix);
}

// Median (lib/src/stats.cpp:34-55) of non-negative values
__device__ inline double workgroup_median(const double * __restrict__ v, uint32_t n, uint32_t * sh)
{
  if (n & 1u) {return workgroup_select(v, n, (n - 1u) / 2u, sh);}
  // even n: the lower of the two middle values by selection, the upper one from it by one pass -- it is the same value
  // again if more than n / 2 values are <= it, the smallest larger value otherwise
  const double e1 = workgroup_select(v, n, n / 2u - 1u, sh);
  const int tid = threadIdx.x, T = blockDim.x;
  uint32_t not_above = 0;
  double next = INFINITY;
  for (uint32_t i = tid; i < n; i += T) {
    const double x = v[i];
    not_above += x <= e1 ? 1u : 0u;
    next = x > e1 && x < next ? x : next;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    not_above += (uint32_t)__shfl_xor((int)not_above, off, 64);
    const double o = __shfl_xor(next, off, 64);
    next = o < next ? o : next;
  }
  uint32_t * cnt = sh + 264;                                    // (behind the selection's words: its bins stay zero)
  double * shd = reinterpret_cast<double *>(sh + 280);
  if ((tid & 63) == 0) {cnt[tid >> 6] = not_above; shd[tid >> 6] = next;}
  __syncthreads();
  uint32_t total = 0;
  double e0 = INFINITY;
  for (int w = 0; w < T / 64; w++) {total += cnt[w]; e0 = shd[w] < e0 ? shd[w] : e0;}
  __syncthreads();
  if (total > n / 2u) {e0 = e1;}
  return (e0 + e1) / 2.;
}

// IsDegenerate (degenerate.cpp:32-37: some |eigenvalue| < threshold) of D = sum of J^T J.  D is positive semi-definite, so
// the test is "smallest eigenvalue < threshold", which is "D - threshold I is not positive definite": one Cholesky
// factorisation that meets a pivot <= 0 (Sylvester's criterion), instead of an eigen-decomposition on one thread.
__device__ inline bool is_degenerate7(const double * Din, double threshold)
{
  constexpr int n = 7;
  // a matrix with a NaN in it has no eigenvalue below the threshold (every comparison with NaN is false): not degenerate,
  // and the NaN goes on into the solve, as it does in the reference
  bool nan = false;
  for (int i = 0; i < n * n; i++) {nan = nan || Din[i] != Din[i];}
  if (nan) {return false;}
  double L[n * n];
  for (int j = 0; j < n; j++) {
    double s = Din[j * n + j] - threshold;
    for (int k = 0; k < j; k++) {s -= L[j * n + k] * L[j * n + k];}
    if (!(s > 0.)) {return true;}
    const double ljj = sqrt(s);
    L[j * n + j] = ljj;
    for (int i = j + 1; i < n; i++) {
      double v = Din[i * n + j];
      for (int k = 0; k < j; k++) {v -= L[i * n + k] * L[j * n + k];}
      L[i * n + j] = v / ljj;
    }
  }
  return false;
}

// CalcUpdate without the sums (optimizer.cpp:60-97): MakeM, the degenerate test on D, -(M^T A M).llt().solve(M^T b),
// AngleAxisToQuaternion.  D, A: 7 x 7 row-major; dq (w x y z), dt out.
__device__ inline void solve_update(const double (&q)[4], const double * D, const double * A, const double * b, double (&dq)[4], double (&dt)[3])
{
  double dx[6] = {0., 0., 0., 0., 0., 0.};
  if (!is_degenerate7(D, 0.1)) {
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    const double L[16] = {w, -x, -y, -z, x, w, -z, y, y, z, w, -x, z, -y, x, w};   // rotationlib LeftMultiplicationMatrix
    double M[42];
    for (int i = 0; i < 42; i++) {M[i] = 0.;}
    for (int r = 0; r < 4; r++) {for (int c = 0; c < 3; c++) {M[6 * r + c] = 0.5 * L[4 * r + 1 + c];}}
    for (int a = 0; a < 3; a++) {M[6 * (4 + a) + 3 + a] = 1.;}
    double AM[42], H[36], g[6];
    for (int r = 0; r < 7; r++) {
      for (int c = 0; c < 6; c++) {
        double s = 0.;
        for (int k = 0; k < 7; k++) {s += A[7 * r + k] * M[6 * k + c];}
        AM[6 * r + c] = s;
      }
    }
    for (int r = 0; r < 6; r++) {
      for (int c = 0; c < 6; c++) {
        double s = 0.;
        for (int k = 0; k < 7; k++) {s += M[6 * k + r] * AM[6 * k + c];}
        H[6 * r + c] = s;
      }
      double s = 0.;
      for (int k = 0; k < 7; k++) {s += M[6 * k + r] * b[k];}
      g[r] = s;
    }
    double Lc[36];
    for (int i = 0; i < 36; i++) {Lc[i] = 0.;}
    for (int j = 0; j < 6; j++) {
      double s = H[6 * j + j];
      for (int k = 0; k < j; k++) {s -= Lc[6 * j + k] * Lc[6 * j + k];}
      Lc[6 * j + j] = sqrt(s);
      for (int i = j + 1; i < 6; i++) {
        double v = H[6 * i + j];
        for (int k = 0; k < j; k++) {v -= Lc[6 * i + k] * Lc[6 * j + k];}
        Lc[6 * i + j] = v / Lc[6 * j + j];
      }
    }
    double yv[6], xv[6];
    for (int i = 0; i < 6; i++) {
      double v = g[i];
      for (int k = 0; k < i; k++) {v -= Lc[6 * i + k] * yv[k];}
      yv[i] = v / Lc[6 * i + i];
    }
    for (int i = 5; i >= 0; i--) {
      double v = yv[i];
      for (int k = i + 1; k < 6; k++) {v -= Lc[6 * k + i] * xv[k];}
      xv[i] = v / Lc[6 * i + i];
    }
    for (int a = 0; a < 6; a++) {dx[a] = -xv[a];}
  }
  const double k = sqrt(dx[0] * dx[0] + dx[1] * dx[1] + dx[2] * dx[2]);            // AngleAxisToQuaternion, posevec.cpp:32-45
  if (k < 1e-8) {
    dq[0] = 1.; dq[1] = 0.; dq[2] = 0.; dq[3] = 0.;
  } else {
    const double sn = sin(k / 2.);
    dq[0] = cos(k / 2.);
    for (int a = 0; a < 3; a++) {dq[1 + a] = (dx[a] / k) * sn;}
  }
  for (int a = 0; a < 3; a++) {dt[a] = dx[3 + a];}
}

// One iteration of Optimizer::Run after Problem::Make, in two kernels.  Rows of scan s: n3 = count3[s * stride3] residuals of
// dimension 3 (r3 / J3 from record begin3[s]: the edge rows, or the point pairs), then n1 = count1[...] of dimension 1
// (the surface rows; count1 may be null).  weights: one double per row, rows of scan s from begin3[s] + begin1[s] (also the
// selection's keys when a scan has more than kAlignKeysLds rows).
//
// align_scale_kernel, one workgroup of 1024 threads per scan: ComputeErrors, the error of the scan, Scale, ComputeWeights, and
// the two stopping tests that need nothing else (optimizer.hpp:97-108).  A scan of up to kAlignKeysLds rows keeps each
// thread's errors in registers -- all their loads in flight at once, read once -- and the selection's keys in LDS.
constexpr int kScaleThreads = 1024, kScaleItems = kAlignKeysLds / kScaleThreads;
__global__ __launch_bounds__(kScaleThreads) void align_scale_kernel(
  AlignState * __restrict__ states, int iter,
  const double * __restrict__ r3, const uint32_t * __restrict__ begin3, const uint32_t * __restrict__ count3, uint32_t stride3,
  const double * __restrict__ r1, const uint32_t * __restrict__ begin1, const uint32_t * __restrict__ count1, uint32_t stride1,
  double * __restrict__ weights, uint32_t * __restrict__ active, AlignOut * __restrict__ out)
{
  constexpr int T = kScaleThreads, E = kScaleItems;
  static_assert(kAlignKeysLds % kScaleThreads == 0, "whole items per thread");
  const uint32_t s = blockIdx.x;
  const int tid = threadIdx.x;
  AlignState & st = states[s];
  __shared__ __attribute__((aligned(8))) uint32_t sh[kSelectWords];
  __shared__ double part[T / 64];
  // (the scan's state and extents asked for together, whether or not there are rows of dimension 1)
  const int32_t done = st.done;
  const uint32_t * c1 = count1 ? count1 + (size_t)s * stride1 : count3, * s1 = count1 ? begin1 + s : begin3;
  const uint32_t n3 = count3[(size_t)s * stride3], b3 = begin3[s], n1_ = *c1, b1_ = *s1;
  if (done) {return;}
  const uint32_t n1 = count1 ? n1_ : 0u, b1 = count1 ? b1_ : 0u, n = n3 + n1;
  if (n == 0u) {                                            // EmptyInput (optimization_result.hpp:46-50)
    if (tid == 0) {align_finish(st, out[s], iter, 0., 0., kAlignEmpty, active);}
    return;
  }
  // surface points, and not one of their neighbourhoods spans a plane (a degenerate map: every surface row is the zero row):
  // no row says anything about the pose along the planes' normals -- a failure of its own kind, not "converged"
  const int32_t with_plane = st.surface_rows_with_plane;
  __syncthreads();
  if (tid == 0) {st.surface_rows_with_plane = 0;}           // (the next iteration's row kernels count afresh)
  if (n1 != 0u && with_plane == 0) {
    if (tid == 0) {align_finish(st, out[s], iter, 0., 0., kAlignNoPlane, active);}
    return;
  }
  double * w_out = weights + (size_t)b3 + b1;
  __shared__ double keys_lds[kAlignKeysLds];
  const bool small = n <= (uint32_t)kAlignKeysLds;
  double * key = small ? keys_lds : w_out;                   // (a longer scan: the keys pass through the weights' place)
  auto row_error = [&](uint32_t i) {                         // ComputeErrors (optimizer.cpp:99-107)
      if (i < n3) {
        const double * r = r3 + 3 * ((size_t)b3 + i);
        return r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
      }
      const double r = r1[(size_t)b1 + (i - n3)];
      return r * r;
    };
  if (tid < 256) {sh[tid] = 0u;}                              // (the selection's bins; the barrier after the keys covers this)
  double e_reg[E];
  double esum = 0.;
  if (small) {
    double ra[E], rb[E], rc[E];
#pragma unroll
    for (int k = 0; k < E; k++) {
      const uint32_t i = (uint32_t)tid + (uint32_t)k * T;
      ra[k] = 0.; rb[k] = 0.; rc[k] = 0.;
      if (i < n3) {
        const double * r = r3 + 3 * ((size_t)b3 + i);
        ra[k] = r[0]; rb[k] = r[1]; rc[k] = r[2];
      } else if (i < n) {
        ra[k] = r1[(size_t)b1 + (i - n3)];
      }
    }
#pragma unroll
    for (int k = 0; k < E; k++) {
      const uint32_t i = (uint32_t)tid + (uint32_t)k * T;
      e_reg[k] = i < n3 ? ra[k] * ra[k] + rb[k] * rb[k] + rc[k] * rc[k] : ra[k] * ra[k];
      if (i < n) {key[i] = e_reg[k]; esum += e_reg[k];}
    }
  } else {
    for (uint32_t i = tid; i < n; i += T) {const double e = row_error(i); key[i] = e; esum += e;}
  }
  __syncthreads();
  // Scale (robust.cpp:36-50): b * median(|e - median(e)|)
  const double median = workgroup_median(key, n, sh);
  if (small) {
#pragma unroll
    for (int k = 0; k < E; k++) {
      const uint32_t i = (uint32_t)tid + (uint32_t)k * T;
      if (i < n) {key[i] = fabs(e_reg[k] - median);}
    }
  } else {
    for (uint32_t i = tid; i < n; i += T) {key[i] = fabs(key[i] - median);}
  }
  __syncthreads();
  const double scale = 1.482602218505602 * workgroup_median(key, n, sh);
  // ComputeWeights (optimizer.cpp:120-127); HuberDerivative, robust.cpp:61-68
  if (small) {
#pragma unroll
    for (int k = 0; k < E; k++) {
      const uint32_t i = (uint32_t)tid + (uint32_t)k * T;
      const double en = e_reg[k] / (scale + 1e-16);
      if (i < n) {w_out[i] = en < 1.345 * 1.345 ? 1. : 1.345 / sqrt(en);}
    }
  } else {
    for (uint32_t i = tid; i < n; i += T) {
      const double en = row_error(i) / (scale + 1e-16);
      w_out[i] = en < 1.345 * 1.345 ? 1. : 1.345 / sqrt(en);
    }
  }
  // the error of the scan (errors.sum()), fixed tree order
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {esum += __shfl_xor(esum, off, 64);}
  if ((tid & 63) == 0) {part[tid >> 6] = esum;}
  __syncthreads();
  if (tid == 0) {
    double error = 0.;
    for (int wv = 0; wv < T / 64; wv++) {error += part[wv];}
    st.cur_error = error; st.cur_scale = scale;
    // the two tests of optimizer.hpp:97-108, in its order; a scan stopped here costs the update kernel nothing
    if (error > st.prev_error) {                             // LargerErrorThanPrevious
      align_finish(st, out[s], iter, error, scale, kAlignLargerError, active);
    } else {
      st.prev_error = error;
      if (scale > st.prev_scale) {                           // LargerScaleThanPrevious
        align_finish(st, out[s], iter, error, scale, kAlignLargerScale, active);
      } else {
        st.prev_scale = scale;
      }
    }
  }
}

// align_update_kernel, kAlignSlices workgroups per scan: the sums of WeightedUpdate (optimizer.cpp:40-64) -- D = sum J^T J,
// A = sum w J^T J, b = sum w J^T r over the 1 x 7 rows of the scan's Jacobians (the three rows of an edge residual lie one
// after the other and share a weight; the surface rows follow) -- as ONE product on the f64 matrix unit: four rows at a
// time, v_mfma_f64_16x16x4_f64 with the left operand [J | 0 | w J | 0]^T (16 x 4) and the right operand [J | r | 0] (4 x 16)
// accumulates D in rows 0-6, A in rows 8-14 and b in column 7 of rows 8-14 of its 16 x 16 tile: no accumulator per pair of
// columns in every thread, and no reduction across lanes (63 sums x 6 shuffle steps were half of this kernel's time).
// The waves' tiles are added in wave order, the slices' in slice order by the workgroup that finishes last, which then
// does the rest of the iteration on one thread: CalcUpdate, the pose, the last two stopping tests.
// (Operand and result lanes of the f64 form: A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15],
// C[row = (lane >> 4) + 4 reg][col = lane & 15], i.e. element reg * 64 + lane of a wave's result is C[16 row + col].)
constexpr int kAlignSlices = 16, kAlignTile = 256;
typedef double lfx_f64x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(kAlignThreads) void align_update_kernel(
  AlignState * __restrict__ states, int iter, int max_iter,
  const double * __restrict__ r3, const double * __restrict__ J3, const uint32_t * __restrict__ begin3,
  const uint32_t * __restrict__ count3, uint32_t stride3,
  const double * __restrict__ r1, const double * __restrict__ J1, const uint32_t * __restrict__ begin1,
  const uint32_t * __restrict__ count1, uint32_t stride1, const double * __restrict__ weights, double * __restrict__ partials,
  uint32_t * __restrict__ tickets, uint32_t * __restrict__ active, AlignOut * __restrict__ out)
{
  constexpr int T = kAlignThreads, W = T / 64, NS = kAlignTile, G = kAlignSlices;
  static_assert(T == NS, "one thread per element of the tile in the sums across waves and slices");
  const uint32_t s = blockIdx.y, g = blockIdx.x;
  const int tid = threadIdx.x;
  AlignState & st = states[s];
  __shared__ double part[W][NS];
  __shared__ double total[NS];
  __shared__ uint32_t last;
  // (the scan's state and extents asked for together, whether or not there are rows of dimension 1: one round trip, not four)
  const int32_t done = st.done;
  const uint32_t * c1 = count1 ? count1 + (size_t)s * stride1 : count3, * s1 = count1 ? begin1 + s : begin3;
  const uint32_t n3 = count3[(size_t)s * stride3], b3 = begin3[s], n1_ = *c1, b1_ = *s1;
  if (done) {return;}
  const uint32_t n1 = count1 ? n1_ : 0u, b1 = count1 ? b1_ : 0u;
  const double * key = weights + (size_t)b3 + b1;
  const uint32_t wave = __builtin_amdgcn_readfirstlane((uint32_t)tid >> 6), lane = (uint32_t)tid & 63u;
  lfx_f64x4 acc = {0., 0., 0., 0.};
  {
    const uint32_t m3 = 3u * n3, m_all = m3 + n1, groups = (m_all + 3u) / 4u;
    const double * Je = J3 + 21 * (size_t)b3, * re = r3 + 3 * (size_t)b3;
    const double * Js = J1 + 7 * (size_t)b1, * rs = r1 + (size_t)b1;
    const uint32_t k = lane >> 4, c = lane & 15u;
    const bool plain = c < 7u, weighted = c >= 8u && c < 15u;
    const uint32_t jc = plain ? c : (weighted ? c - 8u : 0u);          // the column of J this lane reads
    // (the loads of U groups are issued before the first product: a loop of one round trip to memory per step otherwise)
    constexpr int U = 12;
    for (uint32_t g0 = g * W + wave; g0 < groups; g0 += U * G * W) {
      double x[U], wv[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const uint32_t m = 4u * (g0 + (uint32_t)u * G * W) + k;
        const bool live = m < m_all;
        const uint32_t mm = live ? m : 0u;
        const bool edge = mm < m3;
        const double * J = edge ? Je + 7 * (size_t)mm : Js + 7 * (size_t)(mm - m3);
        const double * R = edge ? re + mm : rs + (mm - m3);
        // (every lane loads -- a lane with nothing to fetch reads row 0 -- and what it read is masked afterwards: loads under
        // a condition are waited for one by one)
        x[u] = *(c == 7u ? R : J + jc);                               // J[m][c] | r[m] | J[m][c - 8]
        wv[u] = key[edge ? mm / 3u : n3 + (mm - m3)];
        x[u] = live && c != 15u ? x[u] : 0.;
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const double a = plain ? x[u] : (weighted ? wv[u] * x[u] : 0.);
        const double b = weighted ? 0. : x[u];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
      }
    }
  }
  // fixed order: the waves of the workgroup, then (by the last workgroup) the slices
#pragma unroll
  for (int r = 0; r < 4; r++) {part[wave][64 * r + lane] = acc[r];}
  __syncthreads();
  // The slices' hand-over without a fence: a fence at agent scope writes back and invalidates the whole L2 of its XCD, twice
  // here.  Instead the slice's tile goes out in write-through stores at agent scope, the wave waits until they have left
  // (vmcnt), the barrier collects the waves and one thread takes the ticket; the last workgroup reads the tiles with loads
  // at agent scope, which do not look at its own L2's copy (cdna_hip_programming.md Guideline 16, as the bucketing kernel).
  double * mine = partials + ((size_t)s * G + g) * NS;
  {
    double v = 0.;
    for (int wv = 0; wv < W; wv++) {v += part[wv][tid];}
    __hip_atomic_store(&mine[tid], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {last = __hip_atomic_fetch_add(&tickets[s], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (uint32_t)G - 1u ? 1u : 0u;}
  __syncthreads();
  if (last == 0u) {return;}
  {
    double * all = partials + (size_t)s * G * NS;
    double v = 0.;
#pragma unroll
    for (int k = 0; k < G; k++) {v += __hip_atomic_load(&all[(size_t)k * NS + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);}
    total[tid] = v;
  }
  __syncthreads();
  if (tid != 0) {return;}
  tickets[s] = 0u;                                           // for the next iteration
  const double error = st.cur_error, scale = st.cur_scale;   // (both have passed align_scale_kernel's tests)
  double D[49], A[49], b[7];
  for (int a = 0; a < 7; a++) {                              // the upper triangles, mirrored
    for (int c = a; c < 7; c++) {
      D[7 * a + c] = total[16 * a + c]; D[7 * c + a] = total[16 * a + c];
      A[7 * a + c] = total[16 * (8 + a) + c]; A[7 * c + a] = total[16 * (8 + a) + c];
    }
    b[a] = total[16 * (8 + a) + 7];
  }
  double q[4] = {st.q[0], st.q[1], st.q[2], st.q[3]}, dq[4], dt[3];
  solve_update(q, D, A, b, dq, dt);
  st.q[0] = q[0] * dq[0] - q[1] * dq[1] - q[2] * dq[2] - q[3] * dq[3];          // q = q * dq
  st.q[1] = q[0] * dq[1] + q[1] * dq[0] + q[2] * dq[3] - q[3] * dq[2];
  st.q[2] = q[0] * dq[2] + q[2] * dq[0] + q[3] * dq[1] - q[1] * dq[3];
  st.q[3] = q[0] * dq[3] + q[3] * dq[0] + q[1] * dq[2] - q[2] * dq[1];
  st.t[0] += dt[0]; st.t[1] += dt[1]; st.t[2] += dt[2];
  for (int i = 0; i < 12; i++) {st.prev_m[i] = st.pose.m[i];}
  refresh_pose(st);
  const double nq = sqrt(dq[1] * dq[1] + dq[2] * dq[2] + dq[3] * dq[3]), nt = sqrt(dt[0] * dt[0] + dt[1] * dt[1] + dt[2] * dt[2]);
  if (nq < 1e-3 && nt < 1e-3) {                              // CheckConvergence (optimizer.cpp:35-38)
    align_finish(st, out[s], iter, error, scale, kAlignConverged, active);
  } else if (iter == max_iter - 1) {                         // ReachedMaximumIteration
    align_finish(st, out[s], max_iter, error, scale, kAlignMaxIteration, active);
  }
}

}  // namespace lfx
