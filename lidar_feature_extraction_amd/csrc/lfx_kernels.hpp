// lfx_kernels.hpp -- hand-written HIP kernels (gfx950 / MI355X) for the per-scan lidar feature
// extraction path.  Semantics follow /root/reference/extraction (file:line cited per routine);
// the structure does not: the reference walks rings and blocks sequentially on one CPU thread
// and decides edge/surface points by argsort + greedy suppression; here
//
//   ring_histogram / ring_scan / ring_scatter   stable counting sort of the scan's points by ring
//                                               (MakePointIndices, ring.hpp:114-125) into SoA x,y,index
//   ring_extract    one workgroup per ring: angle order check (+ LDS bitonic sort fallback),
//                   range, curvature, neighbour links, block labelling, occlusion / out-of-range /
//                   parallel-beam masks, in-ring compaction -- everything LDS resident
//   feature_compact packs the per-ring edge / surface lists into the scan's clouds
//
// Labelling without a sort.  The reference's per-block pass (label.hpp:72-95,113-134) visits
// points in curvature order and lets every pick suppress what its link-aware +-P fill reaches
// (fill.hpp:101-117).  "j reaches i" is symmetric (|i-j| <= P, same block, every link between
// them intact), so the picked set is the lexicographically first maximal independent set of the
// candidates in priority order.  That set is computed exactly by rounds of "a live candidate
// with no live higher-priority candidate in reach is picked; everything a pick reaches dies":
// each round is a few AND/shift operations on 32-bit windows of LDS bit arrays, and the
// priority comparisons (f64 curvature, index as tie-break) are done once per candidate.
//
// Floating point: every operation the reference performs in IEEE f64/f32 is performed here in the
// same type and order, unfused (contract off), so integer results (labels, index sets) are
// bit-exact and curvature is bit-equal.  The only libm call on the path, acos() in CalcRadian
// (math.cpp:34-46), is only ever compared with a threshold; the host turns that threshold into
// the equivalent bound on the cosine with the host's own acos (lfx_api.hip: cos_bound()).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace lfx
{

constexpr int kChunkPoints = 1024;     // points per workgroup in the ring bucketing kernels
constexpr int kChunkThreads = 256;
constexpr int kChunkSlots = kChunkPoints / kChunkThreads;
constexpr int kRings = 256;            // ring ids 0..255
constexpr uint32_t kSentinel = 0xFFFFFFFFu;

struct Layout { uint32_t step, ox, oy, oz, oring; };

struct Params
{
  int P;                 // convolution_padding
  int B;                 // n_blocks
  double cos_bound;      // IsNeighborXY(i,i+1)  <=>  cos_bound <= cos_angle <= 1   (neighbor.hpp:44-48)
  double dist_diff;      // distance_diff_threshold
  double pb_ratio;       // parallel_beam_min_range_ratio
  double edge_thr, surf_thr;
  double min_range, max_range;
};

// scan_info[s][4]
enum { kInfoRings = 0, kInfoError = 1, kInfoEdge = 2, kInfoSurface = 3 };

enum RingStatus : uint8_t
{
  kOk = 0, kSparse = 1, kTooFewConv = 2, kTooFewBlocks = 3, kBlockTooSmall = 4, kZeroNormPair = 5,
  kTooLarge = 7
};

enum Label : uint8_t
{
  kDefault = 0, kEdge = 1, kEdgeNeighbor = 2, kSurface = 3, kSurfaceNeighbor = 4, kOutOfRange = 5,
  kOccluded = 6, kParallelBeam = 7
};

// ------------------------------------------------------------------------------------------
// K0: ring histogram per 1024-point chunk.
__global__ __launch_bounds__(kChunkThreads) void ring_histogram_kernel(
  const uint8_t * __restrict__ pts, Layout L, const uint32_t * __restrict__ scan_begin,
  uint16_t * __restrict__ chunk_hist, uint32_t * __restrict__ scan_info, uint32_t max_chunks)
{
  const uint32_t s = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
  const uint32_t b = scan_begin[s], n = scan_begin[s + 1] - b;
  if (chunk * kChunkPoints >= n) {return;}
  __shared__ uint32_t h[kRings];
  h[tid] = 0;
  __syncthreads();
  bool bad = false;
#pragma unroll
  for (int i = 0; i < kChunkSlots; i++) {
    const uint32_t e = chunk * kChunkPoints + i * kChunkThreads + tid;
    if (e < n) {
      const uint32_t ring = *reinterpret_cast<const uint16_t *>(pts + (size_t)(b + e) * L.step + L.oring);
      if (ring >= kRings) {bad = true;} else {atomicAdd(&h[ring], 1u);}
    }
  }
  __syncthreads();
  chunk_hist[((size_t)s * max_chunks + chunk) * kRings + tid] = (uint16_t)h[tid];
  if (bad) {atomicOr(&scan_info[s * 4 + kInfoError], 1u);}
}

// ------------------------------------------------------------------------------------------
// K1: per scan, prefix the chunk histograms per ring, lay the rings out in ascending id, and
// list the non-empty rings (the "slots" ring_extract is launched over).
__global__ __launch_bounds__(kRings) void ring_scan_kernel(
  const uint32_t * __restrict__ scan_begin, const uint16_t * __restrict__ chunk_hist,
  uint32_t * __restrict__ chunk_base, uint32_t * __restrict__ ring_off_by_id,
  uint16_t * __restrict__ ring_id, uint32_t * __restrict__ ring_count, uint32_t * __restrict__ ring_offset,
  uint32_t * __restrict__ scan_info, uint32_t max_chunks)
{
  const uint32_t s = blockIdx.x, r = threadIdx.x;
  const uint32_t n = scan_begin[s + 1] - scan_begin[s];
  const uint32_t nchunks = (n + kChunkPoints - 1) / kChunkPoints;
  uint32_t acc = 0;
  for (uint32_t c = 0; c < nchunks; c++) {
    const size_t k = ((size_t)s * max_chunks + c) * kRings + r;
    const uint32_t v = chunk_hist[k];
    chunk_base[k] = acc;
    acc += v;
  }
  __shared__ uint32_t cnt[kRings], occ[kRings];
  cnt[r] = acc;
  occ[r] = acc ? 1u : 0u;
  __syncthreads();
  // inclusive Hillis-Steele scans over 256 entries (count -> offset, occupancy -> slot)
  for (uint32_t d = 1; d < kRings; d <<= 1) {
    const uint32_t a = r >= d ? cnt[r - d] : 0u, o = r >= d ? occ[r - d] : 0u;
    __syncthreads();
    cnt[r] += a;
    occ[r] += o;
    __syncthreads();
  }
  const uint32_t offset = cnt[r] - acc;
  ring_off_by_id[s * kRings + r] = offset;
  if (acc) {
    const uint32_t slot = occ[r] - 1;
    ring_id[s * kRings + slot] = (uint16_t)r;
    ring_count[s * kRings + slot] = acc;
    ring_offset[s * kRings + slot] = offset;
  }
  if (r == kRings - 1) {scan_info[s * 4 + kInfoRings] = occ[r];}
}

// ------------------------------------------------------------------------------------------
// K2: stable scatter of (x, y, original index) into ring-major SoA.  The rank of a point among
// the points of its ring inside the chunk comes from wave ballots (one per key bit), so the order
// of arrival is kept: position = ring offset + points of the ring in earlier chunks + rank.
__global__ __launch_bounds__(kChunkThreads) void ring_scatter_kernel(
  const uint8_t * __restrict__ pts, Layout L, const uint32_t * __restrict__ scan_begin,
  const uint32_t * __restrict__ chunk_base, const uint32_t * __restrict__ ring_off_by_id,
  float * __restrict__ sx, float * __restrict__ sy, uint32_t * __restrict__ sidx, uint32_t max_chunks)
{
  const uint32_t s = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
  const uint32_t b = scan_begin[s], n = scan_begin[s + 1] - b;
  if (chunk * kChunkPoints >= n) {return;}
  const uint32_t lane = tid & 63, wave = tid >> 6;
  constexpr int kGroups = kChunkSlots * (kChunkThreads / 64);      // (slot, wave) pairs in arrival order
  __shared__ uint16_t wcnt[kGroups][kRings];
  for (int i = tid; i < kGroups * kRings; i += kChunkThreads) {(&wcnt[0][0])[i] = 0;}
  __syncthreads();

  float x[kChunkSlots], y[kChunkSlots];
  uint32_t key[kChunkSlots], rank[kChunkSlots];
#pragma unroll
  for (int i = 0; i < kChunkSlots; i++) {
    const uint32_t e = chunk * kChunkPoints + i * kChunkThreads + tid;
    key[i] = kRings;                       // lanes past the end form their own group
    x[i] = y[i] = 0.f;
    if (e < n) {
      const uint8_t * p = pts + (size_t)(b + e) * L.step;
      x[i] = *reinterpret_cast<const float *>(p + L.ox);
      y[i] = *reinterpret_cast<const float *>(p + L.oy);
      const uint32_t ring = *reinterpret_cast<const uint16_t *>(p + L.oring);
      key[i] = ring < kRings ? ring : kRings;
    }
    uint64_t peers = ~0ull;
#pragma unroll
    for (int bit = 0; bit < 9; bit++) {
      const bool set = (key[i] >> bit) & 1u;
      const uint64_t m = __ballot(set);
      peers &= set ? m : ~m;
    }
    rank[i] = __popcll(peers & ((1ull << lane) - 1ull));
    if (rank[i] == 0 && key[i] < kRings) {wcnt[i * (kChunkThreads / 64) + wave][key[i]] = (uint16_t)__popcll(peers);}
  }
  __syncthreads();
  {
    uint32_t acc = 0;                      // thread = ring id: exclusive prefix over the arrival groups
    for (int g = 0; g < kGroups; g++) {
      const uint32_t v = wcnt[g][tid];
      wcnt[g][tid] = (uint16_t)acc;
      acc += v;
    }
  }
  __syncthreads();
  const uint32_t * cb = chunk_base + ((size_t)s * max_chunks + chunk) * kRings;
#pragma unroll
  for (int i = 0; i < kChunkSlots; i++) {
    if (key[i] < kRings) {
      const uint32_t pos = b + ring_off_by_id[s * kRings + key[i]] + cb[key[i]] +
        wcnt[i * (kChunkThreads / 64) + wave][key[i]] + rank[i];
      sx[pos] = x[i];
      sy[pos] = y[i];
      sidx[pos] = chunk * kChunkPoints + i * kChunkThreads + tid;
    }
  }
}

// ==========================================================================================
// Ring workspace in LDS.
//
// Bit arrays hold one bit per sorted position, 64 per word, with one zero word in front and
// behind so that 32-bit windows around any position can be read without bounds tests.
struct RingWork
{
  float * x;            // [cap]            aliased by hmask after the stencil phase
  float * y;            // [cap]            aliased by rmask
  uint32_t * idx;       // [cap]            original index (within the scan)
  double * r;           // [cap]  Range     range.hpp:52-56
  double * c;           // [cap]  curvature curvature.cpp:44-50
  uint8_t * lab;        // [cap]
  uint64_t * link;      // IsNeighborXY(i, i+1) on the whole ring
  uint64_t * llink;     // the same, cut at block boundaries and ring borders (label.hpp:157-159)
  uint64_t * cand, * alive, * sel, * selE, * covE, * selS, * covS, * jumpL, * jumpR, * featE, * featS;
  uint32_t * wbase;     // [2][cap/64] in-ring offsets of the per-word feature counts
  int * flags;          // [8]
  __device__ uint32_t * hmask() {return reinterpret_cast<uint32_t *>(x);}
  __device__ uint32_t * rmask() {return reinterpret_cast<uint32_t *>(y);}
};

constexpr int kBitArrays = 13;

__host__ __device__ inline size_t ring_lds_bytes(uint32_t cap)
{
  const size_t words = cap / 64 + 2;
  return (size_t)cap * (4 + 4 + 4 + 8 + 8 + 1) + kBitArrays * words * 8 + 2 * (cap / 64) * 4 + 8 * 4 + 64;
}

__device__ inline RingWork carve(uint8_t * base, uint32_t cap)
{
  RingWork w;
  const size_t words = cap / 64 + 2;
  uint8_t * p = base;
  w.r = reinterpret_cast<double *>(p); p += (size_t)cap * 8;
  w.c = reinterpret_cast<double *>(p); p += (size_t)cap * 8;
  uint64_t * bits = reinterpret_cast<uint64_t *>(p); p += kBitArrays * words * 8;
  w.link = bits + 0 * words; w.llink = bits + 1 * words; w.cand = bits + 2 * words;
  w.alive = bits + 3 * words; w.sel = bits + 4 * words; w.selE = bits + 5 * words;
  w.covE = bits + 6 * words; w.selS = bits + 7 * words; w.covS = bits + 8 * words;
  w.jumpL = bits + 9 * words; w.jumpR = bits + 10 * words; w.featE = bits + 11 * words;
  w.featS = bits + 12 * words;
  w.x = reinterpret_cast<float *>(p); p += (size_t)cap * 4;
  w.y = reinterpret_cast<float *>(p); p += (size_t)cap * 4;
  w.idx = reinterpret_cast<uint32_t *>(p); p += (size_t)cap * 4;
  w.wbase = reinterpret_cast<uint32_t *>(p); p += 2 * (size_t)(cap / 64) * 4;
  w.flags = reinterpret_cast<int *>(p); p += 8 * 4;
  w.lab = p;
  return w;
}

// 32-bit window of a bit array around position i: bit 16+d <-> position i+d, d in [-16, 15].
__device__ inline uint32_t window32(const uint64_t * bits, int i)
{
  const int o = i + 64 - 16;
  const int w = o >> 6, sh = o & 63;
  uint64_t v = bits[w] >> sh;
  if (sh) {v |= bits[w + 1] << (64 - sh);}
  return (uint32_t)v;
}

__device__ inline bool bit_at(const uint64_t * bits, int i)
{
  return (bits[(i >> 6) + 1] >> (i & 63)) & 1ull;
}

// One wave covers 64 consecutive positions starting at a multiple of 64: its ballot IS the word.
// Waves whose 64 positions lie wholly past `limit` (the ring length rounded up to 64) take part
// in the ballot but store nothing.
__device__ inline void store_word(uint64_t * bits, int i0 /* multiple of 64 */, int limit, bool pred)
{
  const uint64_t m = __ballot(pred);
  if ((threadIdx.x & 63) == 0 && i0 < limit) {bits[(i0 >> 6) + 1] = m;}
}

__device__ inline void or_word(uint64_t * bits, int i0, int limit, bool pred)
{
  const uint64_t m = __ballot(pred);
  if ((threadIdx.x & 63) == 0 && i0 < limit) {bits[(i0 >> 6) + 1] |= m;}
}

// AHasSmallerPolarAngleThanB for float fields (ring.hpp:54-99): the squares, the product of the
// y's and the determinant are evaluated in float, each operation rounded on its own.
__device__ inline bool polar_less(float ax, float ay, float bx, float by)
{
  if (ax == bx && ay == by) {return false;}
  const float lena = ax * ax + ay * ay;
  const float lenb = bx * bx + by * by;
  if (lena == 0.f) {
    if (by == 0.f) {return bx < 0.f;}
    return by > 0.f;
  }
  if (lenb == 0.f) {return ay < 0.f;}
  if (ay == 0.f) {return (ax >= 0.f) && (by >= 0.f);}
  if (by == 0.f) {return !((bx >= 0.f) && (ay >= 0.f));}
  if (ay * by > 0.f) {
    const float det = ax * by - ay * bx;
    return det > 0.f;
  }
  return ay < 0.f;
}

// total order used by the fallback sort: the predicate, then the original index (the order of
// arrival).  std::sort in the reference is unstable, so ties are unspecified there.
__device__ inline bool sort_less(float ax, float ay, uint32_t ai, float bx, float by, uint32_t bi)
{
  if (ai == kSentinel) {return false;}
  if (bi == kSentinel) {return true;}
  if (polar_less(ax, ay, bx, by)) {return true;}
  if (polar_less(bx, by, ax, ay)) {return false;}
  return ai < bi;
}

// Boundary j of the padded block range: index_range.cpp:60-66 with start=P, end=N-P.
__device__ inline int block_boundary(int N, int P, int B, int j)
{
  const double s = (double)P, e = (double)(N - P), n = (double)B;
  return (int)(s * (1. - j / n) + e * j / n);
}

// ------------------------------------------------------------------------------------------
// Angle order: verify that the ring as bucketed is strictly increasing under the predicate
// (then it IS the sorted order, whatever sort the reference runs); otherwise bitonic-sort it.
// Returns true when a sort was needed.  x, y, idx are LDS arrays of length >= M (pow2 >= N).
__device__ inline bool angle_sort(RingWork & w, int N, uint32_t cap)
{
  const int T = blockDim.x, tid = threadIdx.x;
  int bad = 0;
  for (int i = tid; i + 1 < N; i += T) {
    bad |= !polar_less(w.x[i], w.y[i], w.x[i + 1], w.y[i + 1]);
  }
  if (!__syncthreads_or(bad)) {return false;}
  uint32_t M = 1;
  while (M < (uint32_t)N) {M <<= 1;}
  for (uint32_t i = N + tid; i < M; i += T) {w.idx[i] = kSentinel; w.x[i] = 0.f; w.y[i] = 0.f;}
  __syncthreads();
  for (uint32_t k = 2; k <= M; k <<= 1) {
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      for (uint32_t t = tid; t < M / 2; t += T) {
        const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1));     // bit j clear
        const uint32_t p = i | j;
        const bool asc = (i & k) == 0;
        const float ax = w.x[i], ay = w.y[i], bx = w.x[p], by = w.y[p];
        const uint32_t ai = w.idx[i], bi = w.idx[p];
        const bool swap = asc ? sort_less(bx, by, bi, ax, ay, ai) : sort_less(ax, ay, ai, bx, by, bi);
        if (swap) {
          w.x[i] = bx; w.y[i] = by; w.idx[i] = bi;
          w.x[p] = ax; w.y[p] = ay; w.idx[p] = ai;
        }
      }
      __syncthreads();
    }
  }
  (void)cap;
  return true;
}

// ------------------------------------------------------------------------------------------
// Range, curvature, links.  `groups` (debug neighbour test) and `curv_in` (given curvature) are
// only used by the per-stage entry points.  Sets flags[1] when an adjacent pair is both (0,0).
__device__ inline void stencil_phase(
  RingWork & w, const Params & prm, int N, int Npad, const int32_t * groups, const double * curv_in,
  const double * range_in)
{
  const int T = blockDim.x, tid = threadIdx.x, P = prm.P;
  for (int i = tid; i < N; i += T) {
    const double x = (double)w.x[i], y = (double)w.y[i];
    w.r[i] = range_in ? range_in[i] : sqrt(x * x + y * y);           // math.hpp:36-39
  }
  __syncthreads();
  int zero_pair = 0;
  for (int i0 = 0; i0 < Npad; i0 += T) {
    const int i = i0 + tid;
    bool lk = false;
    if (i + 1 < N) {
      if (groups) {
        lk = groups[i] == groups[i + 1];                             // neighbor.hpp:124-127
      } else {
        const double r0 = w.r[i], r1 = w.r[i + 1];
        if (r0 == 0. && r1 == 0.) {zero_pair = 1;}                   // math.cpp:40-42 throws
        const double dot = (double)w.x[i] * (double)w.x[i + 1] + (double)w.y[i] * (double)w.y[i + 1];
        const double cosang = dot / (r0 * r1);
        lk = cosang >= prm.cos_bound && cosang <= 1.0;               // acos(cos) < threshold, NaN -> false
      }
    }
    store_word(w.link, i0 + (tid & ~63), Npad, lk);
    if (i < N) {
      double cv = 0.;
      if (curv_in) {
        cv = curv_in[i];
      } else if (i >= P && i < N - P) {                              // convolution.cpp:52-63: zero borders
        double sum = 0.;                                             // math.hpp:46-52: left to right from 0
        for (int k = -P; k <= P; k++) {
          const double wt = (k == 0) ? -2. * P : 1.;                 // curvature.cpp:36-42
          sum += w.r[i + k] * wt;
        }
        cv = sum * sum;                                              // curvature.cpp:47
      }
      w.c[i] = cv;
    }
  }
  if (zero_pair) {w.flags[1] = 1;}
}

// ------------------------------------------------------------------------------------------
// Block structure: mark block starts, cut the links at block ends, detect blocks of < 2 points.
// single_block: one block [0, N) (EdgeLabel::Assign on a bare array).
__device__ inline void block_phase(RingWork & w, const Params & prm, int N, int Npad, bool single_block)
{
  const int T = blockDim.x, tid = threadIdx.x, P = prm.P, B = prm.B;
  const int words = (Npad >> 6) + 2;
  for (int k = tid; k < words; k += T) {w.cand[k] = 0;}              // cand doubles as "is a block end" marks here
  __syncthreads();
  const int first = single_block ? 0 : P, last = single_block ? N : N - P;
  if (!single_block) {
    for (int j = tid; j < B; j += T) {
      const int b0 = block_boundary(N, P, B, j), b1 = block_boundary(N, P, B, j + 1);
      if (b1 - b0 < 2) {w.flags[2] = 1;}                             // neighbor.hpp:71-75 on the slice
      if (b1 - 1 >= 0) {atomicOr(reinterpret_cast<unsigned long long *>(&w.cand[((b1 - 1) >> 6) + 1]), 1ull << ((b1 - 1) & 63));}
    }
  } else if (tid == 0 && N < 2) {
    w.flags[2] = 1;
  }
  __syncthreads();
  for (int i0 = 0; i0 < Npad; i0 += T) {
    const int i = i0 + tid;
    bool lk = false;
    if (i >= first && i + 1 < last) {lk = bit_at(w.link, i) && !bit_at(w.cand, i);}
    store_word(w.llink, i0 + (tid & ~63), Npad, lk);
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------------
// One labelling pass over all blocks of the ring at once (EDGE: label.hpp:72-95 descending with
// c >= threshold; surface: label.hpp:113-134 ascending with c <= threshold on points the edge
// pass left Default).  Writes sel (picked) and cov (reached by a pick, the pick included).
template<bool EDGE>
__device__ inline void label_pass(
  RingWork & w, const Params & prm, int N, int Npad, bool single_block, uint64_t * selAll, uint64_t * covAll)
{
  const int T = blockDim.x, tid = threadIdx.x, P = prm.P;
  const int first = single_block ? 0 : P, last = single_block ? N : N - P;
  uint32_t * hm = w.hmask(), * rm = w.rmask();
  for (int i0 = 0; i0 < Npad; i0 += T) {
    const int i = i0 + tid;
    bool cd = false;
    if (i >= first && i < last) {
      const double c = w.c[i];
      cd = EDGE ? (c >= prm.edge_thr) : (c <= prm.surf_thr && !bit_at(w.covE, i));
    }
    const int w0 = i0 + (tid & ~63);
    store_word(w.cand, w0, Npad, cd);
    store_word(w.alive, w0, Npad, cd);
    store_word(selAll, w0, Npad, false);
    store_word(covAll, w0, Npad, false);
  }
  __syncthreads();
  for (int i = tid; i < N; i += T) {
    const uint32_t ll = window32(w.llink, i);                        // bit 16+d: link between i+d and i+d+1
    int L = __clz((int)~(ll << 16));                                 // intact links leftwards from i-1
    int R = __ffs((int)~(ll >> 16)) - 1;                             // intact links rightwards from i
    L = L < P ? L : P;
    R = R < P ? R : P;
    const uint32_t reach = ((1u << (L + R + 1)) - 1u) << (16 - L);   // fill.hpp:101-117 around i
    uint32_t higher = 0;
    if (bit_at(w.cand, i)) {
      const double ci = w.c[i];
      uint32_t m = window32(w.cand, i) & reach & ~(1u << 16);
      while (m) {
        const int b = __ffs((int)m) - 1;
        m &= m - 1;
        const int j = i + b - 16;
        const double cj = w.c[j];
        const bool first_j = EDGE ? (cj > ci || (cj == ci && j > i)) : (cj < ci || (cj == ci && j < i));
        if (first_j) {higher |= 1u << b;}
      }
    }
    hm[i] = higher;
    rm[i] = reach;
  }
  __syncthreads();
  for (;; ) {
    for (int i0 = 0; i0 < Npad; i0 += T) {
      const int i = i0 + tid;
      bool s = false;
      if (i < N && bit_at(w.alive, i)) {s = (window32(w.alive, i) & hm[i]) == 0;}
      const int w0 = i0 + (tid & ~63);
      store_word(w.sel, w0, Npad, s);
      or_word(selAll, w0, Npad, s);
    }
    __syncthreads();
    int any = 0;
    for (int i0 = 0; i0 < Npad; i0 += T) {
      const int i = i0 + tid;
      bool hit = false, live = false;
      if (i < N) {
        hit = (window32(w.sel, i) & rm[i]) != 0;
        live = bit_at(w.alive, i) && !hit;
      }
      const int w0 = i0 + (tid & ~63);
      or_word(covAll, w0, Npad, hit);
      store_word(w.alive, w0, Npad, live);
      any |= live;
    }
    if (!__syncthreads_or(any)) {break;}
  }
}

// ------------------------------------------------------------------------------------------
// Masks (feature_extraction.cpp:135-138, in this order, each overwriting) and the final label.
__device__ inline void mask_phase(RingWork & w, const Params & prm, int N, int Npad, uint32_t flags)
{
  const int T = blockDim.x, tid = threadIdx.x, P = prm.P;
  const bool do_occ = flags & 2u, do_oor = flags & 4u, do_pb = flags & 8u;
  for (int i0 = 0; i0 < Npad; i0 += T) {
    const int i = i0 + tid;
    bool jl = false, jr = false;
    if (do_occ) {
      // occlusion.hpp:44-57: i in [0, N-P-1), linked pair, far side to the right
      if (i + 1 < N && i < N - P - 1 && bit_at(w.link, i)) {jl = w.r[i + 1] > w.r[i] + prm.dist_diff;}
      // occlusion.hpp:67-79: i in [P+1, N-1], linked pair (i, i-1), far side to the left
      if (i < N && i >= P + 1 && bit_at(w.link, i - 1)) {jr = w.r[i - 1] > w.r[i] + prm.dist_diff;}
    }
    const int w0 = i0 + (tid & ~63);
    store_word(w.jumpL, w0, Npad, jl);
    store_word(w.jumpR, w0, Npad, jr);
  }
  __syncthreads();
  for (int i0 = 0; i0 < Npad; i0 += T) {
    const int i = i0 + tid;
    uint8_t lab = kDefault;
    if (i < N) {
      if (bit_at(w.selE, i)) {
        lab = kEdge;
      } else if (bit_at(w.selS, i)) {
        lab = kSurface;
      } else if (bit_at(w.covS, i)) {
        lab = kSurfaceNeighbor;
      } else if (bit_at(w.covE, i)) {
        lab = kEdgeNeighbor;
      }
      if (do_occ) {
        const uint32_t lk = window32(w.link, i);
        // FillFromLeft from a jump at i-k (k = 1..P+1) reaches i when links i-k+1 .. i-1 hold
        int Lr = __clz((int)~(lk << 16));
        Lr = Lr < P ? Lr : P;
        const uint32_t left = ((1u << (Lr + 1)) - 1u) << (15 - Lr);    // positions i-1 .. i-1-Lr
        // FillFromRight from a jump at i+k reaches i when links i .. i+k-2 hold
        int Rr = __ffs((int)~(lk >> 16)) - 1;
        Rr = Rr < P ? Rr : P;
        const uint32_t right = ((1u << (Rr + 1)) - 1u) << 16;          // positions i+1 .. i+1+Rr of the window around i+1
        if ((window32(w.jumpL, i) & left) || (window32(w.jumpR, i + 1) & right)) {lab = kOccluded;}
      }
      const double ri = w.r[i];
      if (do_oor && !(prm.min_range <= ri && ri <= prm.max_range)) {lab = kOutOfRange;}   // range.hpp:40-43
      if (do_pb && i >= 1 && i + 1 < N) {                                                 // parallel_beam.hpp:43-49
        const float ratio1 = (float)(fabs(w.r[i - 1] - ri) / ri);
        const float ratio2 = (float)(fabs(w.r[i + 1] - ri) / ri);
        if ((double)ratio1 > prm.pb_ratio && (double)ratio2 > prm.pb_ratio) {lab = kParallelBeam;}
      }
      w.lab[i] = lab;
    }
    const int w0 = i0 + (tid & ~63);
    store_word(w.featE, w0, Npad, lab == kEdge);
    store_word(w.featS, w0, Npad, lab == kSurface);
  }
  __syncthreads();
}

// Runs label + mask phases on a ring whose x, y (and idx) are in LDS.  Returns the ring status.
__device__ inline uint8_t process_ring(
  RingWork & w, const Params & prm, int N, uint32_t flags, const int32_t * groups, const double * curv_in,
  const double * range_in)
{
  const int T = blockDim.x, tid = threadIdx.x, P = prm.P;
  const int Npad = (N + 63) & ~63;
  const bool single_block = flags & 16u;
  const bool do_label = flags & 1u;
  // RemoveSparseRings (ring.cpp:46-59; also the least LabelOccludedPoints is defined for),
  // Convolution1D (convolution.cpp:39-43), IndexRange (index_range.cpp:35-40)
  if ((flags & 2u) && N < P + 1) {return kSparse;}
  if ((flags & 32u) && N < 2 * P + 1) {return kTooFewConv;}
  if (do_label && !single_block && N - 2 * P < prm.B) {return kTooFewBlocks;}
  const int words = (Npad >> 6) + 2;
  for (int k = tid; k < words; k += T) {
    w.link[k] = 0; w.llink[k] = 0; w.cand[k] = 0; w.alive[k] = 0; w.sel[k] = 0; w.selE[k] = 0; w.covE[k] = 0;
    w.selS[k] = 0; w.covS[k] = 0; w.jumpL[k] = 0; w.jumpR[k] = 0; w.featE[k] = 0; w.featS[k] = 0;
  }
  if (tid < 8) {w.flags[tid] = 0;}
  __syncthreads();
  stencil_phase(w, prm, N, Npad, groups, curv_in, range_in);
  __syncthreads();
  if (do_label) {
    block_phase(w, prm, N, Npad, single_block);
  }
  if (w.flags[1]) {return kZeroNormPair;}
  if (w.flags[2]) {return kBlockTooSmall;}
  if (do_label) {
    label_pass<true>(w, prm, N, Npad, single_block, w.selE, w.covE);
    label_pass<false>(w, prm, N, Npad, single_block, w.selS, w.covS);
  }
  mask_phase(w, prm, N, Npad, flags);
  return kOk;
}

// ------------------------------------------------------------------------------------------
// K3: one workgroup per (ring slot, scan).
__global__ __launch_bounds__(1024) void ring_extract_kernel(
  Params prm, uint32_t cap, const uint32_t * __restrict__ scan_begin, const uint32_t * __restrict__ scan_info,
  const uint32_t * __restrict__ ring_count, const uint32_t * __restrict__ ring_offset,
  float * __restrict__ sx, float * __restrict__ sy, uint32_t * __restrict__ sidx,
  uint8_t * __restrict__ label_s, double * __restrict__ curv_s, uint32_t * __restrict__ seg,
  uint8_t * __restrict__ ring_status, uint32_t * __restrict__ ring_nedge, uint32_t * __restrict__ ring_nsurf)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
  const uint32_t slot = blockIdx.x, s = blockIdx.y;
  if (slot >= scan_info[s * 4 + kInfoRings]) {return;}
  const int T = blockDim.x, tid = threadIdx.x;
  const int N = (int)ring_count[s * kRings + slot];
  const size_t off = (size_t)scan_begin[s] + ring_offset[s * kRings + slot];
  uint8_t status = kOk;
  RingWork w = carve(lds_raw, cap);
  if ((uint32_t)N > cap) {
    status = kTooLarge;
  } else {
    for (int i = tid; i < N; i += T) {
      w.x[i] = sx[off + i];
      w.y[i] = sy[off + i];
      w.idx[i] = sidx[off + i];
    }
    __syncthreads();
    if (angle_sort(w, N, cap)) {
      for (int i = tid; i < N; i += T) {
        sx[off + i] = w.x[i];
        sy[off + i] = w.y[i];
        sidx[off + i] = w.idx[i];
      }
    }
    __syncthreads();
    status = process_ring(w, prm, N, 47u, nullptr, nullptr, nullptr);
  }
  if (status != kOk) {
    // the ring contributes nothing (feature_extraction.cpp:116,154-156)
    for (int i = tid; i < N; i += T) {
      label_s[off + i] = kDefault;
      curv_s[off + i] = 0.;
    }
    if (tid == 0) {
      ring_status[s * kRings + slot] = status;
      ring_nedge[s * kRings + slot] = 0;
      ring_nsurf[s * kRings + slot] = 0;
    }
    return;
  }
  // in-ring compaction: edge positions ascending from the front of the ring's segment, surface
  // positions ascending from its back (edge + surface <= N, so they never meet)
  const int nwords = (N + 63) >> 6;
  if (tid < 64) {
    uint32_t ce = 0, cs = 0;
    for (int base = 0; base < nwords; base += 64) {
      const int k = base + tid;
      const uint32_t ne = k < nwords ? __popcll(w.featE[k + 1]) : 0u;
      const uint32_t ns = k < nwords ? __popcll(w.featS[k + 1]) : 0u;
      uint32_t ie = ne, is = ns;
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t te = __shfl_up(ie, d), ts = __shfl_up(is, d);
        if (tid >= d) {ie += te; is += ts;}
      }
      if (k < nwords) {
        w.wbase[k] = ce + ie - ne;
        w.wbase[cap / 64 + k] = cs + is - ns;
      }
      ce += __shfl(ie, 63);
      cs += __shfl(is, 63);
    }
    if (tid == 0) {
      ring_status[s * kRings + slot] = kOk;
      ring_nedge[s * kRings + slot] = ce;
      ring_nsurf[s * kRings + slot] = cs;
    }
  }
  __syncthreads();
  for (int i = tid; i < N; i += T) {
    const uint8_t lab = w.lab[i];
    label_s[off + i] = lab;
    curv_s[off + i] = w.c[i];
    const uint64_t below = (1ull << (i & 63)) - 1ull;
    if (lab == kEdge) {
      seg[off + w.wbase[i >> 6] + __popcll(w.featE[(i >> 6) + 1] & below)] = (uint32_t)i;
    } else if (lab == kSurface) {
      seg[off + N - 1 - (w.wbase[cap / 64 + (i >> 6)] + __popcll(w.featS[(i >> 6) + 1] & below))] = (uint32_t)i;
    }
  }
}

// ------------------------------------------------------------------------------------------
// K4: pack the per-ring lists into the scan's edge / surface clouds, rings ascending
// (AppendXYZIR, label.hpp:166-179: intensity <- (float)curvature).
__global__ __launch_bounds__(256) void feature_compact_kernel(
  const uint8_t * __restrict__ pts, Layout L, const uint32_t * __restrict__ scan_begin,
  uint32_t * __restrict__ scan_info, const uint32_t * __restrict__ ring_count,
  const uint32_t * __restrict__ ring_offset, const uint32_t * __restrict__ ring_nedge,
  const uint32_t * __restrict__ ring_nsurf, const float * __restrict__ sx, const float * __restrict__ sy,
  const uint32_t * __restrict__ sidx, const double * __restrict__ curv_s, const uint32_t * __restrict__ seg,
  float4 * __restrict__ edge_pts, uint32_t * __restrict__ edge_idx, float4 * __restrict__ surf_pts,
  uint32_t * __restrict__ surf_idx)
{
  const uint32_t slot = blockIdx.x, s = blockIdx.y, tid = threadIdx.x;
  const uint32_t nr = scan_info[s * 4 + kInfoRings];
  if (slot >= nr) {return;}
  __shared__ uint32_t red[2][256];
  red[0][tid] = (tid < slot) ? ring_nedge[s * kRings + tid] : 0u;
  red[1][tid] = (tid < slot) ? ring_nsurf[s * kRings + tid] : 0u;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if ((int)tid < d) {red[0][tid] += red[0][tid + d]; red[1][tid] += red[1][tid + d];}
    __syncthreads();
  }
  const uint32_t ebase = red[0][0], sbase = red[1][0];
  const uint32_t ne = ring_nedge[s * kRings + slot], ns = ring_nsurf[s * kRings + slot];
  const uint32_t N = ring_count[s * kRings + slot];
  const size_t b = scan_begin[s];
  const size_t off = b + ring_offset[s * kRings + slot];
  if (slot == nr - 1 && tid == 0) {
    scan_info[s * 4 + kInfoEdge] = ebase + ne;
    scan_info[s * 4 + kInfoSurface] = sbase + ns;
  }
  for (uint32_t k = tid; k < ne + ns; k += blockDim.x) {
    const bool edge = k < ne;
    const uint32_t q = edge ? k : k - ne;
    const uint32_t i = edge ? seg[off + q] : seg[off + N - 1 - q];
    const uint32_t orig = sidx[off + i];
    const float z = *reinterpret_cast<const float *>(pts + (b + orig) * L.step + L.oz);
    const float4 v = make_float4(sx[off + i], sy[off + i], z, (float)curv_s[off + i]);
    if (edge) {
      edge_pts[b + ebase + q] = v;
      edge_idx[b + ebase + q] = orig;
    } else {
      surf_pts[b + sbase + q] = v;
      surf_idx[b + sbase + q] = orig;
    }
  }
}

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
  uint32_t capacity)
{
  const uint32_t s = blockIdx.y;
  const uint32_t ne = scan_info[s * 4 + kInfoEdge], ns = scan_info[s * 4 + kInfoSurface];
  const size_t b = scan_begin[s];
  const uint32_t oe = offsets[s], os = offsets[batch + 1 + s];
  for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < ne + ns; k += gridDim.x * blockDim.x) {
    if (k < ne) {
      if (oe + k < capacity) {edge_out[oe + k] = edge_pts[b + k];}
    } else {
      const uint32_t q = k - ne;
      if (os + q < capacity) {surf_out[os + q] = surf_pts[b + q];}
    }
  }
}

// ------------------------------------------------------------------------------------------
// Per-stage kernel: one ring handed over as sorted x, y (lfx_stage_ring).
__global__ __launch_bounds__(1024) void ring_stage_kernel(
  Params prm, uint32_t cap, uint32_t flags, int N, const float * __restrict__ x, const float * __restrict__ y,
  const int32_t * __restrict__ groups, const double * __restrict__ curv_in,
  const double * __restrict__ range_in, double * __restrict__ range_out,
  double * __restrict__ curv_out, uint8_t * __restrict__ link_out, uint8_t * __restrict__ labels_out,
  int32_t * __restrict__ status_out)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
  const int T = blockDim.x, tid = threadIdx.x;
  RingWork w = carve(lds_raw, cap);
  for (int i = tid; i < N; i += T) {
    w.x[i] = x[i];
    w.y[i] = y[i];
    w.idx[i] = i;
    w.lab[i] = kDefault;
    w.r[i] = 0.;
    w.c[i] = 0.;
  }
  __syncthreads();
  const uint8_t status = process_ring(w, prm, N, flags, groups, curv_in, range_in);
  __syncthreads();
  for (int i = tid; i < N; i += T) {
    if (range_out) {range_out[i] = w.r[i];}
    if (curv_out) {curv_out[i] = status == kOk ? w.c[i] : 0.;}
    if (labels_out) {labels_out[i] = status == kOk ? w.lab[i] : (uint8_t)kDefault;}
    if (link_out && i + 1 < N) {link_out[i] = bit_at(w.link, i);}
  }
  if (tid == 0) {*status_out = status;}
}

// Convolution1D (convolution.cpp:35-66) for an arbitrary odd weight, one thread per output.
__global__ void convolution1d_kernel(
  const double * __restrict__ in, int n, const double * __restrict__ weight, int m, double * __restrict__ out)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) {return;}
  const int pad = (m - 1) / 2;
  double v = 0.;
  if (i >= pad && i < n - pad) {
    double sum = 0.;
    for (int k = 0; k < m; k++) {sum += in[i - pad + k] * weight[k];}
    v = sum;
  }
  out[i] = v;
}

}  // namespace lfx
