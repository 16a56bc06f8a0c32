// lfx_kernels_extract.hpp -- the extraction path around the unit kernels (lfx_kernels_unit.hpp): ring bucketing, order
// repair, the ring transforms of turned streams, the workgroup-per-ring slow path, compaction, download.  Overview:
// lfx_kernels_common.hpp.
#pragma once

#include "lfx_kernels_common.hpp"

#pragma clang fp contract(off)

namespace lfx
{

// ------------------------------------------------------------------------------------------
// What the batch's accumulators need before a batch (lfx_kernels_common.hpp: the counters, the per-scan flag words and the
// organised route's ring totals are kept clean by the batch before; scan_info and ring_count are WRITTEN by every route, never
// added to).  This kernel is what remains: the tables only the bucketing route dirties -- the look-back flags, the ring flags,
// the ring transforms -- over the scans a batch since the last reset may have touched, and the fall-back list of a batch
// that takes the bucketing route whole.  A stream the organised-scan kernel keeps taking never launches it.
__global__ __launch_bounds__(256) void batch_reset_kernel(
  uint32_t * __restrict__ chunk_flags, uint32_t n_flags, uint32_t * __restrict__ ring_flags, uint32_t n_rflags,
  uint32_t * __restrict__ counters /* this batch's set */, uint32_t * __restrict__ fb_list, uint32_t batch,
  uint32_t all_fall_back /* 1: every scan takes the bucketing route (the organised-scan kernel is not launched) */,
  uint32_t * __restrict__ xform /* [batch][256] ring transforms: identity unless ring_cut_kernel runs */)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
  for (uint32_t k = i; k < n_flags; k += stride) {chunk_flags[k] = 0;}
  for (uint32_t k = i; k < n_rflags; k += stride) {ring_flags[k] = 0; xform[k] = 0;}
  if (all_fall_back) {
    for (uint32_t k = i; k < batch; k += stride) {fb_list[k] = k;}
    if (i == 0) {counters[kCntFallback] = batch;}
  }
}

constexpr uint32_t kSpinLimit = 200000;   // ~50 ms of s_sleep polls before a look-back gives up

// Ring bucketing: stable scatter of (x, y | z | original index) into ring-major arrays.
// The only pass over the input.  A chunk publishes its per-ring counts, waits
// for the counts of the scan's earlier chunks (lower block index: already dispatched) and sums
// them -- no separate histogram pass.  Release / acquire at agent scope as
// cdna_hip_programming.md Guideline 16 prescribes; the spin is bounded and a timeout marks the
// scan as failed instead of hanging.  The rank of a point
// among the points of its ring inside the chunk comes from wave ballots (one per key bit), so the
// order of arrival is kept.  The chunk is first laid out ring-major in LDS; the stores to HBM then
// walk that layout, so a wave writes runs of consecutive positions (one run per ring) instead of
// 64 scattered dwords.
template<bool CANON>
__device__ __forceinline__ void scatter_chunk(
  uint32_t s, const uint8_t * __restrict__ pts, Layout L, const uint32_t * __restrict__ scan_begin,
  uint32_t * __restrict__ chunk_base, uint32_t * __restrict__ chunk_flags, uint32_t * __restrict__ ring_count,
  uint32_t * __restrict__ scan_info, uint32_t * __restrict__ scan_flags, float2 * __restrict__ sxy, float * __restrict__ sz,
  uint32_t * __restrict__ sidx, uint32_t max_chunks, uint32_t max_rings, uint32_t cap, uint32_t drop_zero,
  const uint16_t * __restrict__ ring_slot /* [65536] ring id -> slot (0xFFFF: not one of the sensor's), or nullptr: the id is the slot */)
{
  const uint32_t chunk = blockIdx.x, tid = threadIdx.x;
  const uint32_t b = scan_begin[s], n = scan_begin[s + 1] - b;
  if (chunk * kChunkPoints >= n) {return;}
  const uint32_t lane = tid & 63, wave = tid >> 6;
  const uint32_t n_here = (n - chunk * kChunkPoints) < (uint32_t)kChunkPoints ? (n - chunk * kChunkPoints) : (uint32_t)kChunkPoints;
  constexpr int kGroups = kChunkSlots * (kChunkThreads / 64);      // (slot, wave) pairs in arrival order
  __shared__ uint32_t cstart[kRings];                              // start of each ring inside the staged chunk
  __shared__ uint32_t gfill[kRings];                               // points of the ring in earlier chunks
  __shared__ uint32_t staged;                                      // points with a valid ring id in this chunk
  // staging, skewed by one element per 32: with column-major input a wave's 64 points go to 64
  // different rings, i.e. to staged positions one run length apart -- a power-of-two stride.
  // One LDS block serves two phases: first the (slot, wave) count table `wcnt` (16 KB) and the
  // look-back scratch `part` (4 KB behind it); then, once every thread holds its staging positions in
  // registers, the staging arrays over the same bytes (33 KB in all: four workgroups per CU).
  constexpr int kStage = kChunkPoints + kChunkPoints / 32 + 1;
  constexpr int kStageBytes = kStage * (8 + 4 + 2 + 1) + 16;
  __shared__ __attribute__((aligned(16))) uint8_t block[kStageBytes];
  uint16_t (*wcnt)[kRings] = reinterpret_cast<uint16_t (*)[kRings]>(block);
  uint32_t * part = reinterpret_cast<uint32_t *>(block + kGroups * kRings * 2);
  float2 * st_xy = reinterpret_cast<float2 *>(block);
  float * st_z = reinterpret_cast<float *>(block + kStage * 8);
  uint16_t * st_src = reinterpret_cast<uint16_t *>(block + kStage * 12);
  uint8_t * st_ring = block + kStage * 14;
  static_assert(kGroups * kRings * 2 + (kChunkThreads / 64) * kRings * 4 <= kStageBytes, "count table + look-back scratch must fit the staging block");
  static_assert(kChunkThreads >= kRings && kChunkThreads % 64 == 0, "one thread per ring id for the per-ring steps");
  for (int i = tid; i < kGroups * kRings; i += kChunkThreads) {(&wcnt[0][0])[i] = 0;}
  __syncthreads();

  float x[kChunkSlots], y[kChunkSlots], z[kChunkSlots];
  uint32_t key[kChunkSlots], rank[kChunkSlots];
  bool bad_ring = false;
#pragma unroll
  for (int i = 0; i < kChunkSlots; i++) {
    const uint32_t e = chunk * kChunkPoints + i * kChunkThreads + tid;
    key[i] = kRings;                       // lanes past the end form their own group
    x[i] = y[i] = z[i] = 0.f;
    if (e < n) {
      const uint8_t * p = pts + (size_t)(b + e) * L.step;
      uint32_t ring;
      if (CANON) {
        // PointXYZIR (point_type.hpp:62-86): x y z pad | intensity ring: one 16-byte and one 4-byte load
        const float4 v = *reinterpret_cast<const float4 *>(p);
        x[i] = v.x; y[i] = v.y; z[i] = v.z;
        ring = *reinterpret_cast<const uint32_t *>(p + 20) & 0xFFFFu;
      } else {
        x[i] = load_f32(p + L.ox, L.be);
        y[i] = load_f32(p + L.oy, L.be);
        z[i] = load_f32(p + L.oz, L.be);
        ring = load_ring(p + L.oring, L.rtype, L.be);
      }
      // (the reference buckets by whatever id a point carries, ring.hpp:114-125: ids that are not 0 .. rings-1 go through the
      // context's table to the slot of their rank among the sensor's ids)
      if (ring_slot != nullptr) {ring = ring < 65536u ? ring_slot[ring] : 0xFFFFu;}
      key[i] = ring < max_rings ? ring : kRings;
      if (ring >= max_rings) {bad_ring = true;}
      // the upstream converter's filter (point_type_converter/convert.py:162-163,192): all-zero points
      // are not part of the scan
      if (drop_zero && x[i] == 0.f && y[i] == 0.f && z[i] == 0.f) {key[i] = kRings;}
    }
  }
  const size_t row = (size_t)s * max_chunks;
  {
    // the per-ring counts are all the later chunks need: count with LDS atomics and publish them
    // before the ranking work, with write-through (sc1) stores, drain, then ONE lane sets the flag.
    // No L2 write-back fence: a release fence would flush every dirty line of the XCD's L2, i.e. the
    // other workgroups' scattered output (cdna_hip_programming.md Guideline 16, form R1).
    if (tid < kRings) {cstart[tid] = 0;}
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kChunkSlots; i++) {
      if (key[i] < kRings) {atomicAdd(&cstart[key[i]], 1u);}
    }
    __syncthreads();
    if (bad_ring) {atomicOr(&scan_flags[s], 1u);}
    if (tid < kRings) {
      __hip_atomic_store(&chunk_base[(row + chunk) * kRings + tid], cstart[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_store(&chunk_flags[row + chunk], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
#pragma unroll
  for (int i = 0; i < kChunkSlots; i++) {
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
  uint32_t mine = 0;
  if (tid < kRings) {
    uint32_t acc = 0;                      // thread = ring id: exclusive prefix over the arrival groups
    for (int g = 0; g < kGroups; g++) {
      const uint32_t v = wcnt[g][tid];
      wcnt[g][tid] = (uint16_t)acc;
      acc += v;
    }
    mine = acc;                            // points of ring `tid` in this chunk
    cstart[tid] = acc;
  }
  uint32_t before = 0;                     // points of ring `tid` in the scan's earlier chunks
  {
    bool timeout = false;
    for (uint32_t p = tid; p < chunk; p += kChunkThreads) {
      uint32_t spins = 0;
      while (__hip_atomic_load(&chunk_flags[row + p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > kSpinLimit) {timeout = true; break;}
      }
    }
    if (timeout) {atomicOr(&scan_flags[s], 4u);}
    __syncthreads();
    // every load of the published counts is an sc1 (L1-bypassing, agent-scope) load; four 16-byte
    // loads are kept in flight per thread (thread = 4 consecutive rings x one quarter of the chunks)
    {
      typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
      constexpr uint32_t kLb = kChunkThreads / 64;             // groups of 64 threads, each taking every kLb-th chunk
      const uint32_t grp = tid >> 6, quad = tid & 63;
      u32x4 acc = {0u, 0u, 0u, 0u};
      for (uint32_t p0 = 0; p0 < chunk; p0 += 4 * kLb) {
        const uint32_t pa = p0 + grp, pb = pa + kLb, pc = pa + 2 * kLb, pd = pa + 3 * kLb;
        const uint32_t * qa = chunk_base + (row + (pa < chunk ? pa : 0u)) * kRings + 4 * quad;
        const uint32_t * qb = chunk_base + (row + (pb < chunk ? pb : 0u)) * kRings + 4 * quad;
        const uint32_t * qc = chunk_base + (row + (pc < chunk ? pc : 0u)) * kRings + 4 * quad;
        const uint32_t * qd = chunk_base + (row + (pd < chunk ? pd : 0u)) * kRings + 4 * quad;
        u32x4 va, vb, vc, vd;
        asm volatile(
          "global_load_dwordx4 %0, %4, off sc1\n\t"
          "global_load_dwordx4 %1, %5, off sc1\n\t"
          "global_load_dwordx4 %2, %6, off sc1\n\t"
          "global_load_dwordx4 %3, %7, off sc1\n\t"
          "s_waitcnt vmcnt(0)"
          : "=&v"(va), "=&v"(vb), "=&v"(vc), "=&v"(vd)
          : "v"(qa), "v"(qb), "v"(qc), "v"(qd)
          : "memory");
        const u32x4 zero = {0u, 0u, 0u, 0u};
        acc += (pa < chunk ? va : zero) + (pb < chunk ? vb : zero) + (pc < chunk ? vc : zero) + (pd < chunk ? vd : zero);
      }
      __syncthreads();
      // part: [kLb][256] u32 behind the count table
      part[grp * kRings + 4 * quad + 0] = acc.x;
      part[grp * kRings + 4 * quad + 1] = acc.y;
      part[grp * kRings + 4 * quad + 2] = acc.z;
      part[grp * kRings + 4 * quad + 3] = acc.w;
      __syncthreads();
      if (tid < kRings) {
        for (uint32_t g = 0; g < kLb; g++) {before += part[g * kRings + tid];}
      }
      __syncthreads();
    }
    if ((chunk + 1) * (uint32_t)kChunkPoints >= n) {     // the scan's last chunk knows the ring totals
      if (tid < kRings) {ring_count[s * kRings + tid] = before + mine;}
      const uint32_t occupied = __syncthreads_count(tid < kRings && before + mine != 0u);
      if (tid == 0) {scan_info[s * 4 + kInfoRings] = occupied;}
    }
  }
  {
    // exclusive scan of the 256 ring counts: shuffles inside each wave, one exchange of the four wave
    // totals through LDS (two barriers instead of sixteen)
    uint32_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t t = __shfl_up(incl, d);
      if (lane >= (uint32_t)d) {incl += t;}
    }
    __syncthreads();                      // every thread has read its cstart[] entry by now
    if (lane == 63 && wave < kRings / 64) {gfill[wave] = incl;} // wave totals, parked in gfill[0..3] for a moment
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t v = 0; v < wave && v < kRings / 64; v++) {base += gfill[v];}
    const uint32_t total = gfill[0] + gfill[1] + gfill[2] + gfill[3];
    __syncthreads();
    if (tid < kRings) {
      cstart[tid] = base + incl - mine;
      gfill[tid] = before;
    }
    if (tid == 0) {staged = total;}
  }
  __syncthreads();
  uint32_t lp[kChunkSlots];
#pragma unroll
  for (int i = 0; i < kChunkSlots; i++) {
    lp[i] = 0;
    if (key[i] < kRings) {
      lp[i] = cstart[key[i]] + wcnt[i * (kChunkThreads / 64) + wave][key[i]] + rank[i];
      lp[i] += lp[i] >> 5;
    }
  }
  __syncthreads();                          // the count table is dead: its bytes become staging
#pragma unroll
  for (int i = 0; i < kChunkSlots; i++) {
    if (key[i] < kRings) {
      st_xy[lp[i]] = make_float2(x[i], y[i]);
      st_z[lp[i]] = z[i];
      st_src[lp[i]] = (uint16_t)(i * kChunkThreads + tid);
      st_ring[lp[i]] = (uint8_t)key[i];
    }
  }
  __syncthreads();
  // (a point with a ring id above 255 is not staged; the host rejects such a scan, LFX_ERR_RING_ID)
  const uint32_t n_staged = staged < n_here ? staged : n_here;
  for (uint32_t o = tid; o < n_staged; o += kChunkThreads) {
    const uint32_t os = o + (o >> 5);
    const uint32_t r = st_ring[os];
    const uint32_t k = gfill[r] + (o - cstart[r]);       // position inside the ring
    if (k < cap) {                                       // a ring longer than its capacity is reported, not stored
      const size_t pos = ring_base(s, r, max_rings, cap) + k;
      sxy[pos] = st_xy[os];
      sz[pos] = st_z[os];
      sidx[pos] = chunk * kChunkPoints + st_src[os];
    }
  }
}

// The scans to bucket are those on the fall-back list (every scan of the batch when the organised-scan kernel is
// not in use): workgroup (x, y) takes chunk x of list entries y, y + gridDim.y, ...  A chunk only ever waits for
// lower chunks of the same scan, i.e. for workgroups (x' < x, y) at the same step of their loop: no cycle.
// ONE = true: the grid has a row per scan of the batch (every scan may be on the list: a stream that is not organised), so a
// workgroup has one entry at most and no loop -- the loop costs the kernel 40 registers, a handful of spills and, measured
// on a ragged stream, a third of its speed.
template<bool CANON, bool ONE = false>
__global__ __launch_bounds__(kChunkThreads, ONE ? 1 : 3) void ring_scatter_kernel(
  const uint8_t * __restrict__ pts, Layout L, const uint32_t * __restrict__ scan_begin,
  uint32_t * __restrict__ chunk_base, uint32_t * __restrict__ chunk_flags, uint32_t * __restrict__ ring_count,
  uint32_t * __restrict__ scan_info, uint32_t * __restrict__ scan_flags, float2 * __restrict__ sxy, float * __restrict__ sz,
  uint32_t * __restrict__ sidx, uint32_t max_chunks, uint32_t max_rings, uint32_t cap, uint32_t drop_zero,
  const uint32_t * __restrict__ fb_count, const uint32_t * __restrict__ fb_list, const uint16_t * __restrict__ ring_slot)
{
  const uint32_t n_list = *fb_count;
  if (ONE) {
    if (blockIdx.y < n_list) {
      scatter_chunk<CANON>(fb_list[blockIdx.y], pts, L, scan_begin, chunk_base, chunk_flags, ring_count, scan_info, scan_flags, sxy, sz,
        sidx, max_chunks, max_rings, cap, drop_zero, ring_slot);
    }
    return;
  }
  for (uint32_t it = blockIdx.y; it < n_list; it += gridDim.y) {
    scatter_chunk<CANON>(fb_list[it], pts, L, scan_begin, chunk_base, chunk_flags, ring_count, scan_info, scan_flags, sxy, sz,
      sidx, max_chunks, max_rings, cap, drop_zero, ring_slot);
    __syncthreads();                        // the LDS blocks are reused by the next entry
  }
}

// ==========================================================================================
// Ring workspace in LDS (one workgroup = one ring of N <= cap points, cap a multiple of 64).
//
// Bit arrays hold one bit per sorted position, 64 per word, with one zero word in front and
// behind so that 32-bit windows around any position can be read without bounds tests.
struct RingWork
{
  double * r;           // [cap]  Range     range.hpp:52-56
  double * c;           // [cap]  curvature curvature.cpp:44-50
  float * x;            // [cap]  (the slow labelling path reuses x, y as its mask arrays)
  float * y;            // [cap]
  uint8_t * lab;        // [cap]  label after the block labelling, then the final label
  uint64_t * link;      // IsNeighborXY(i, i+1) on the whole ring
  uint64_t * llink;     // the same, cut at block boundaries and ring borders (label.hpp:157-159)
  uint64_t * cand, * alive, * sel, * selE, * covE, * selS, * covS, * jumpL, * jumpR, * featE, * featS;
  uint32_t * wbase;     // [2][cap/64] in-ring offsets of the per-word feature counts
  int * flags;          // [8]
  uint8_t * base;       // start of the workspace: scratch of the fallback angle sort
  __device__ uint32_t * hmask() {return reinterpret_cast<uint32_t *>(x);}
  __device__ uint32_t * rmask() {return reinterpret_cast<uint32_t *>(y);}
};

constexpr int kBitArrays = 13;
enum { kFlagUnsorted = 0, kFlagZeroPair = 1, kFlagBlockSmall = 2, kFlagBlockBig = 3, kFlagXYClobbered = 4 };

__host__ __device__ inline size_t ring_lds_bytes(uint32_t cap)
{
  const size_t words = cap / 64 + 2;
  return (size_t)cap * (8 + 8 + 4 + 4 + 1) + kBitArrays * words * 8 + 2 * (cap / 64) * 4 + 8 * 4 + 64;
}

__device__ inline RingWork carve(uint8_t * base, uint32_t cap)
{
  RingWork w;
  const size_t words = cap / 64 + 2;
  uint8_t * p = base;
  w.base = base;
  w.r = reinterpret_cast<double *>(p); p += (size_t)cap * 8;
  w.c = reinterpret_cast<double *>(p); p += (size_t)cap * 8;
  w.x = reinterpret_cast<float *>(p); p += (size_t)cap * 4;
  w.y = reinterpret_cast<float *>(p); p += (size_t)cap * 4;
  uint64_t * bits = reinterpret_cast<uint64_t *>(p); p += kBitArrays * words * 8;
  w.link = bits + 0 * words; w.llink = bits + 1 * words; w.cand = bits + 2 * words;
  w.alive = bits + 3 * words; w.sel = bits + 4 * words; w.selE = bits + 5 * words;
  w.covE = bits + 6 * words; w.selS = bits + 7 * words; w.covS = bits + 8 * words;
  w.jumpL = bits + 9 * words; w.jumpR = bits + 10 * words; w.featE = bits + 11 * words;
  w.featS = bits + 12 * words;
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

// 64 bits of a bit array starting at position g (any alignment, g >= -64).
__device__ inline uint64_t word_at(const uint64_t * bits, int g)
{
  const int o = g + 64;
  const int w = o >> 6, sh = o & 63;
  uint64_t v = bits[w] >> sh;
  if (sh) {v |= bits[w + 1] << (64 - sh);}
  return v;
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

// A 32-bit key that grows with the polar angle as polar_less orders it (y < 0 first, the positive x axis and points of
// zero length at 0, the negative x axis last): the pseudo-angle y / (|x| + |y|) unfolded over the four quadrants, in
// (-2, 2], as an unsigned integer of the same order.  Rounded f32 arithmetic: two points whose angles differ by less than a
// few ulp may come out equal or swapped -- whoever sorts by it verifies the result with the predicate.
__device__ inline float polar_pseudo_angle(float x, float y)
{
  const float len = x * x + y * y;                       // (what the predicate calls a zero point: its squared length rounds to 0)
  const float p = y * __builtin_amdgcn_rcpf(fabsf(x) + fabsf(y));
  const float t = x < 0.f ? (y < 0.f ? -2.f - p : 2.f - p) : p;
  return len == 0.f ? 0.f : t + 0.f;                     // (-0 -> +0)
}
__device__ inline uint32_t polar_key(float t /* polar_pseudo_angle */)
{
  const uint32_t u = __float_as_uint(t);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
// ... and one of kOrderBuckets equal slices of (-2, 2] (a non-decreasing function of the same number: buckets and keys agree)
constexpr int kOrderBuckets = 1024;
__device__ inline uint32_t polar_bucket(float t)
{
  const float b = (t + 2.f) * (0.25f * kOrderBuckets);
  return b >= (float)(kOrderBuckets - 1) ? (uint32_t)(kOrderBuckets - 1) : (b > 0.f ? (uint32_t)b : 0u);     // (a NaN lands in bucket 0)
}


// Pair t of step (k, j) of a bitonic sorting network over M = 2^m places in its ALL-ASCENDING form: the first step of
// every merge (j = k / 2) pairs place i with its mirror image inside the block of k places, i ^ (k - 1), the later ones
// with i | j; every exchange puts the smaller element at i < p.  Places beyond the N elements there are stand for +infinity
// and are never moved, so a caller skips the pairs with p >= N and stores N elements only.
__device__ inline void bitonic_partner(uint32_t t, uint32_t k, uint32_t j, uint32_t & i, uint32_t & p)
{
  i = ((t & ~(j - 1u)) << 1) | (t & (j - 1u));                        // bit j clear
  p = j == (k >> 1) ? (i ^ (k - 1u)) : (i | j);
}

// ------------------------------------------------------------------------------------------
// Angle order.  w.x / w.y hold the ring as bucketed.  If it is strictly increasing under the
// predicate it IS the sorted order (whatever sort the reference runs) and nothing is done.
// Otherwise the ring is bitonic-sorted (total order: predicate, then arrival index) in scratch
// laid over the whole workspace, written back to gx/gy/gidx, and w.x / w.y are refilled.
// Returns true when a sort was needed.
__device__ inline bool angle_sort(
  RingWork & w, int N, float2 * gxy, uint32_t * gidx /* the ring's slices in global memory */)
{
  const int T = blockDim.x, tid = threadIdx.x;
  int bad = 0;
  for (int i = tid; i + 1 < N; i += T) {
    bad |= !polar_less(w.x[i], w.y[i], w.x[i + 1], w.y[i + 1]);
  }
  if (!__syncthreads_or(bad)) {return false;}
  uint32_t M = 1;
  while (M < (uint32_t)N) {M <<= 1;}
  // 12 N <= 25 cap bytes: the scratch fits the workspace
  float * sx = reinterpret_cast<float *>(w.base);
  float * sy = sx + N;
  uint32_t * si = reinterpret_cast<uint32_t *>(sy + N);
  for (uint32_t i = tid; i < (uint32_t)N; i += T) {
    const float2 v = gxy[i];
    sx[i] = v.x;
    sy[i] = v.y;
    si[i] = gidx[i];
  }
  __syncthreads();
  // the bitonic network in the form whose every exchange puts the smaller element at the LOWER index (bitonic_partner):
  // the places N .. M - 1 of the power of two above N then stand for elements larger than any, which no exchange ever
  // moves -- they need no storage, and a ring's length is not bound to a power of two that fits the LDS
  for (uint32_t k = 2; k <= M; k <<= 1) {
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      for (uint32_t t = tid; t < M / 2; t += T) {
        uint32_t i, p;
        bitonic_partner(t, k, j, i, p);
        if (p < (uint32_t)N) {
          const float ax = sx[i], ay = sy[i], bx = sx[p], by = sy[p];
          const uint32_t ai = si[i], bi = si[p];
          if (sort_less(bx, by, bi, ax, ay, ai)) {
            sx[i] = bx; sy[i] = by; si[i] = bi;
            sx[p] = ax; sy[p] = ay; si[p] = ai;
          }
        }
      }
      __syncthreads();
    }
  }
  for (int i = tid; i < N; i += T) {
    gxy[i] = make_float2(sx[i], sy[i]);
    gidx[i] = si[i];
  }
  __syncthreads();          // all waves are done with the scratch; the workgroup's own global stores
                            // are visible to it after the barrier (workgroup-scope fence)
  for (int i = tid; i < N; i += T) {
    const float2 v = gxy[i];
    w.x[i] = v.x;
    w.y[i] = v.y;
  }
  __syncthreads();
  return true;
}

// ------------------------------------------------------------------------------------------
// Range, curvature, links.  `groups` (debug neighbour test), `curv_in` (given curvature) and
// `range_in` are only used by the per-stage entry points.
__device__ inline void stencil_phase(
  RingWork & w, const Params & prm, int N, int Npad, const int32_t * groups, const double * curv_in,
  const double * range_in)
{
  const int T = blockDim.x, tid = threadIdx.x, P = prm.P;
  for (int i = tid; i < N; i += T) {
    const double x = (double)w.x[i], y = (double)w.y[i];
    w.r[i] = range_in ? range_in[i] : sqrt(x * x + y * y);           // math.hpp:36-39
    w.lab[i] = kDefault;
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
  if (zero_pair) {w.flags[kFlagZeroPair] = 1;}
}

// ==========================================================================================
// Block labelling, fast path: one wave per block, everything in registers.
//
// Lane l of the wave owns the block's local positions q = 64k + l (k < K <= kWaveChunks).  A set
// of points is a K-word bitmask held wave-uniformly (the ballot of a per-lane predicate is
// exactly the word of its 64 positions), so a round of the pick/suppress iteration is ballots
// and funnel shifts -- no LDS, no barrier.
constexpr int kWaveChunks = 6;            // blocks of up to 384 points take the fast path

// window of a uniform word array around local position 64k + lane; W[0] is a zero pad word.
// In 32-bit halves U[2j], U[2j+1] of W[j]: the window starts at half 2(k+1)-1, 2(k+1) or 2(k+1)+1
// for lanes < 16, 16..47, >= 48, at bit (lane+16)&31 -- two selects between wave-uniform halves
// and one v_alignbit.
__device__ inline uint32_t uwindow(const uint64_t (&W)[kWaveChunks + 2], int k)
{
  const int lane = threadIdx.x & 63;
  const uint32_t u0 = (uint32_t)(W[k] >> 32), u1 = (uint32_t)W[k + 1], u2 = (uint32_t)(W[k + 1] >> 32),
    u3 = (uint32_t)W[k + 2];
  const uint32_t lo = lane < 16 ? u0 : (lane >= 48 ? u2 : u1);
  const uint32_t up = lane < 16 ? u1 : (lane >= 48 ? u3 : u2);
  return __builtin_amdgcn_alignbit(up, lo, (uint32_t)(lane + 16) & 31u);
}

// bit `lane` of a wave-uniform word, and the number of set bits below it, without 64-bit shifts
__device__ inline bool ubit(uint64_t w, int lane)
{
  const uint32_t half = lane < 32 ? (uint32_t)w : (uint32_t)(w >> 32);
  return (half >> (lane & 31)) & 1u;
}

__device__ inline uint32_t ubelow(uint64_t w, int lane)
{
  const uint32_t lo = (uint32_t)w, up = (uint32_t)(w >> 32);
  const uint32_t m = (1u << (lane & 31)) - 1u;
  return lane < 32 ? __popc(lo & m) : __popc(lo) + __popc(up & m);
}

// Order masks: bit 16+d of lt[k] (d = -P..P, d != 0) says that the neighbour at q+d comes BEFORE
// position q = 64k+lane in ascending curvature order with ties broken by the lower index -- the
// visiting order of the surface pass; the edge pass visits in exactly the reverse order
// (label.hpp:85-89 iterates the same argsort backwards), so its mask is the complement.
// cl[q] is the curvature at local position q; it may be read up to P positions outside [0, nloc):
// callers keep that addressable and the bits are masked by reach / candidates later.
template<int PT>
__device__ inline void order_masks(
  const double * cl_, const Params & prm, int nloc, int K, int lane, uint32_t (&lt)[kWaveChunks])
{
  const int P = PT > 0 ? PT : prm.P;
#pragma unroll
  for (int k = 0; k < kWaveChunks; k++) {
    lt[k] = 0;
    if (k < K) {
      const int q = 64 * k + lane;
      const int qc = q < nloc ? q : nloc - 1;
      const double ci = cl_[qc];
      uint32_t m = 0;
      if (PT > 0) {
#pragma unroll
        for (int d = 1; d <= (PT > 0 ? PT : 1); d++) {
          const double cl = cl_[qc - d], cr = cl_[qc + d];
          m |= (cl <= ci) ? (1u << (16 - d)) : 0u;        // left neighbour: lower index wins a tie
          m |= (cr < ci) ? (1u << (16 + d)) : 0u;
        }
      } else {
        for (int d = 1; d <= P; d++) {
          const double cl = cl_[qc - d], cr = cl_[qc + d];
          m |= (cl <= ci) ? (1u << (16 - d)) : 0u;
          m |= (cr < ci) ? (1u << (16 + d)) : 0u;
        }
      }
      lt[k] = m;
    }
  }
}

// One pass over one block.  `taken` (bit k = local position 64k+lane is no longer Default) is the
// per-lane state handed from the edge pass to the surface pass; the pass returns which of the
// lane's positions it picked (`sel`) and which it reached (`cov`, picks included).
// cl[q] is the curvature at local position q (q in [0, nloc)); `inblk` bit k says whether the
// lane's position 64k+lane belongs to the block being labelled.
template<bool EDGE, int PT>
__device__ inline void wave_pass(
  const double * cl_, const Params & prm, int nloc, uint32_t inblk, int K, int lane,
  const uint32_t (&reach)[kWaveChunks], const uint32_t (&lt)[kWaveChunks], uint32_t taken, uint32_t & sel,
  uint32_t & cov)
{
  uint64_t A[kWaveChunks + 2], S[kWaveChunks + 2];
  uint32_t H[kWaveChunks];
  sel = 0;
  cov = 0;
#pragma unroll
  for (int k = 0; k < kWaveChunks + 2; k++) {A[k] = 0; S[k] = 0;}
  uint64_t any = 0;
  uint32_t alive = 0;                      // the lane's own bits of A
#pragma unroll
  for (int k = 0; k < kWaveChunks; k++) {
    if (k < K) {
      bool cd = (inblk >> k) & 1u;
      const int q0 = 64 * k + lane;
      const double c0 = cl_[q0 < nloc ? q0 : nloc - 1];
      if (EDGE) {
        cd = cd && c0 >= prm.edge_thr;                                    // label.hpp:80-82
      } else {
        cd = cd && c0 <= prm.surf_thr && !((taken >> k) & 1u);            // label.hpp:119-121, still Default
      }
      A[k + 1] = __ballot(cd);
      alive |= (cd ? 1u : 0u) << k;
      any |= A[k + 1];
    }
  }
  if (any == 0) {return;}
  // priority masks: which candidates in reach are visited first
#pragma unroll
  for (int k = 0; k < kWaveChunks; k++) {
    H[k] = 0;
    if (k < K && A[k + 1] != 0) {
      const uint32_t m = uwindow(A, k) & reach[k] & ~(1u << 16);
      H[k] = (EDGE ? ~lt[k] : lt[k]) & m;
    }
  }
  // rounds: a live candidate with no live candidate of higher priority in reach is picked;
  // everything a pick reaches (the pick included) leaves the live set
  for (;; ) {
#pragma unroll
    for (int k = 0; k < kWaveChunks; k++) {
      if (k < K) {
        uint64_t s = 0;
        if (A[k + 1] != 0) {
          const bool pick = ((alive >> k) & 1u) && (uwindow(A, k) & H[k]) == 0;
          s = __ballot(pick);
          sel |= (pick ? 1u : 0u) << k;
        }
        S[k + 1] = s;
      }
    }
    uint64_t left = 0, picked = 0;
#pragma unroll
    for (int k = 0; k < kWaveChunks; k++) {picked |= S[k + 1];}
    // With a total order the live candidate of highest priority is always picked.  No pick at all
    // means the order is inconsistent (NaN curvature from non-finite input): stop instead of spinning.
    if (picked == 0) {break;}
#pragma unroll
    for (int k = 0; k < kWaveChunks; k++) {
      if (k < K) {
        if ((S[k] | S[k + 1] | S[k + 2]) != 0) {
          const bool hit = (uwindow(S, k) & reach[k]) != 0;
          const uint32_t hb = (hit ? 1u : 0u) << k;
          cov |= hb;
          alive &= ~hb;
          A[k + 1] &= ~__ballot(hit);
        }
        left |= A[k + 1];
      }
    }
    if (left == 0) {break;}
  }
}

template<int PT>
__device__ inline void label_blocks_wave(RingWork & w, const Params & prm, int N, bool single_block)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int P = PT > 0 ? PT : prm.P;
  const int nblocks = single_block ? 1 : prm.B;
  for (int j = wave; j < nblocks; j += nwaves) {
    const int b0 = single_block ? 0 : block_boundary(N, prm.P, prm.B, j);
    const int b1 = single_block ? N : block_boundary(N, prm.P, prm.B, j + 1);
    const int nb = b1 - b0;
    const int K = (nb + 63) >> 6;
    uint32_t reach[kWaveChunks];
    {
      uint64_t LL[kWaveChunks + 2];
#pragma unroll
      for (int k = 0; k < kWaveChunks + 2; k++) {LL[k] = 0;}
#pragma unroll
      for (int k = 0; k < kWaveChunks; k++) {
        if (k < K) {
          // links inside the block (label.hpp:157-159: the checker is sliced to the block)
          const int rem = nb - 1 - 64 * k;
          const uint64_t keep = rem >= 64 ? ~0ull : (rem <= 0 ? 0ull : ((1ull << rem) - 1ull));
          LL[k + 1] = word_at(w.link, b0 + 64 * k) & keep;
        }
      }
#pragma unroll
      for (int k = 0; k < kWaveChunks; k++) {
        reach[k] = 0;
        if (k < K) {
          const uint32_t ll = uwindow(LL, k);                    // bit 16+d: link between q+d and q+d+1
          int L = __clz((int)~(ll << 16));
          int R = __ffs((int)~(ll >> 16)) - 1;
          L = L < P ? L : P;
          R = R < P ? R : P;
          reach[k] = (64 * k + lane < nb) ? (((1u << (L + R + 1)) - 1u) << (16 - L)) : 0u;   // fill.hpp:101-117
        }
      }
    }
    uint32_t inblk = 0;
#pragma unroll
    for (int k = 0; k < kWaveChunks; k++) {inblk |= (64 * k + lane < nb ? 1u : 0u) << k;}
    uint32_t selE, covE, selS, covS;
    uint32_t lt[kWaveChunks];
    order_masks<PT>(w.c + b0, prm, nb, K, lane, lt);
    wave_pass<true, PT>(w.c + b0, prm, nb, inblk, K, lane, reach, lt, 0u, selE, covE);
    wave_pass<false, PT>(w.c + b0, prm, nb, inblk, K, lane, reach, lt, covE, selS, covS);
#pragma unroll
    for (int k = 0; k < kWaveChunks; k++) {
      if (k < K) {
        const int q = 64 * k + lane;
        if (q < nb) {
          uint8_t lab = kDefault;
          if ((selE >> k) & 1u) {
            lab = kEdge;
          } else if ((selS >> k) & 1u) {
            lab = kSurface;
          } else if ((covS >> k) & 1u) {
            lab = kSurfaceNeighbor;
          } else if ((covE >> k) & 1u) {
            lab = kEdgeNeighbor;
          }
          w.lab[b0 + q] = lab;
        }
      }
    }
  }
}

// Block sizes: a block of < 2 points makes the reference throw (neighbor.hpp:71-75 on the slice);
// a block of more than 64*kWaveChunks points takes the slow labelling path.
__device__ inline void block_check(RingWork & w, const Params & prm, int N, bool single_block)
{
  const int T = blockDim.x, tid = threadIdx.x;
  if (single_block) {
    if (tid == 0) {
      if (N < 2) {w.flags[kFlagBlockSmall] = 1;}
      if (N > 64 * kWaveChunks) {w.flags[kFlagBlockBig] = 1;}
    }
    return;
  }
  for (int j = tid; j < prm.B; j += T) {
    const int nb = block_boundary(N, prm.P, prm.B, j + 1) - block_boundary(N, prm.P, prm.B, j);
    if (nb < 2) {w.flags[kFlagBlockSmall] = 1;}
    if (nb > 64 * kWaveChunks) {w.flags[kFlagBlockBig] = 1;}
  }
}

// ==========================================================================================
// Block labelling, slow path (blocks longer than 512 points): the same iteration over the whole
// ring at once with the point sets as LDS bit arrays and a barrier between the half rounds.
__device__ inline void block_phase(RingWork & w, const Params & prm, int N, int Npad, bool single_block)
{
  const int T = blockDim.x, tid = threadIdx.x, P = prm.P, B = prm.B;
  const int words = (Npad >> 6) + 2;
  for (int k = tid; k < words; k += T) {w.cand[k] = 0;}              // cand doubles as "is a block end" marks here
  __syncthreads();
  const int first = single_block ? 0 : P, last = single_block ? N : N - P;
  if (!single_block) {
    for (int j = tid; j < B; j += T) {
      const int b1 = block_boundary(N, P, B, j + 1);
      if (b1 - 1 >= 0) {atomicOr(reinterpret_cast<unsigned long long *>(&w.cand[((b1 - 1) >> 6) + 1]), 1ull << ((b1 - 1) & 63));}
    }
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
    int picked = 0;
    for (int i0 = 0; i0 < Npad; i0 += T) {
      const int i = i0 + tid;
      bool s = false;
      if (i < N && bit_at(w.alive, i)) {s = (window32(w.alive, i) & hm[i]) == 0;}
      const int w0 = i0 + (tid & ~63);
      store_word(w.sel, w0, Npad, s);
      or_word(selAll, w0, Npad, s);
      picked |= s;
    }
    // no pick at all = inconsistent order (NaN curvature from non-finite input): stop, do not spin
    if (!__syncthreads_or(picked)) {break;}
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

// The same pass for a padding the 32-position windows do not span (P > kWindowPadding = 15; the reference accepts any
// padding > 0, hyper_parameter.hpp:45): the reach of a position is kept as its two run lengths and every test walks the
// positions in reach.  O(P) per position and round where the window form is O(1): the workgroup-per-ring kernel is the slow
// path already, and such a padding takes it for every ring.
template<bool EDGE>
__device__ inline void label_pass_wide(
  RingWork & w, const Params & prm, int N, int Npad, bool single_block, uint64_t * selAll, uint64_t * covAll)
{
  const int T = blockDim.x, tid = threadIdx.x, P = prm.P;
  const int first = single_block ? 0 : P, last = single_block ? N : N - P;
  uint32_t * reach = w.hmask();                                      // run of intact links leftwards | rightwards << 16
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
  for (int i = tid; i < N; i += T) {
    int L = 0, R = 0;                                                // fill.hpp:101-117 around i, links cut at the block ends (llink)
    while (L < P && i - 1 - L >= 0 && bit_at(w.llink, i - 1 - L)) {L++;}
    while (R < P && i + R < N && bit_at(w.llink, i + R)) {R++;}
    reach[i] = (uint32_t)L | ((uint32_t)R << 16);
  }
  __syncthreads();
  for (;; ) {
    int picked = 0;
    for (int i0 = 0; i0 < Npad; i0 += T) {
      const int i = i0 + tid;
      bool s = false;
      if (i < N && bit_at(w.alive, i)) {
        const int L = (int)(reach[i] & 0xFFFFu), R = (int)(reach[i] >> 16);
        const double ci = w.c[i];
        s = true;
        for (int j = i - L; j <= i + R && s; j++) {
          if (j != i && bit_at(w.alive, j)) {
            const double cj = w.c[j];
            if (EDGE ? (cj > ci || (cj == ci && j > i)) : (cj < ci || (cj == ci && j < i))) {s = false;}
          }
        }
      }
      const int w0 = i0 + (tid & ~63);
      store_word(w.sel, w0, Npad, s);
      or_word(selAll, w0, Npad, s);
      picked |= s;
    }
    if (!__syncthreads_or(picked)) {break;}
    int any = 0;
    for (int i0 = 0; i0 < Npad; i0 += T) {
      const int i = i0 + tid;
      bool hit = false, live = false;
      if (i < N) {
        const int L = (int)(reach[i] & 0xFFFFu), R = (int)(reach[i] >> 16);    // (j reaches i exactly when i reaches j)
        for (int j = i - L; j <= i + R && !hit; j++) {hit = bit_at(w.sel, j);}
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

constexpr int kWindowPadding = 15;        // the paddings the 32-position windows of the kernels span

__device__ inline void label_blocks_lds(RingWork & w, const Params & prm, int N, int Npad, bool single_block)
{
  const int T = blockDim.x, tid = threadIdx.x;
  block_phase(w, prm, N, Npad, single_block);
  if (tid == 0) {w.flags[kFlagXYClobbered] = 1;}
  if (prm.P > kWindowPadding) {
    label_pass_wide<true>(w, prm, N, Npad, single_block, w.selE, w.covE);
    label_pass_wide<false>(w, prm, N, Npad, single_block, w.selS, w.covS);
  } else {
    label_pass<true>(w, prm, N, Npad, single_block, w.selE, w.covE);
    label_pass<false>(w, prm, N, Npad, single_block, w.selS, w.covS);
  }
  for (int i = tid; i < N; i += T) {
    uint8_t lab = kDefault;
    if (bit_at(w.selE, i)) {
      lab = kEdge;
    } else if (bit_at(w.selS, i)) {
      lab = kSurface;
    } else if (bit_at(w.covS, i)) {
      lab = kSurfaceNeighbor;
    } else if (bit_at(w.covE, i)) {
      lab = kEdgeNeighbor;
    }
    w.lab[i] = lab;
  }
}

// ------------------------------------------------------------------------------------------
// Masks (feature_extraction.cpp:135-138, in this order, each overwriting) and the final label.
__device__ inline void mask_phase(RingWork & w, const Params & prm, int N, int Npad, uint32_t flags)
{
  const int T = blockDim.x, tid = threadIdx.x, P = prm.P;
  const bool do_occ = flags & 2u, do_oor = flags & 4u, do_pb = flags & 8u;
  if (do_occ) {
    for (int i0 = 0; i0 < Npad; i0 += T) {
      const int i = i0 + tid;
      bool jl = false, jr = false;
      // occlusion.hpp:44-57: i in [0, N-P-1), linked pair, far side to the right
      if (i + 1 < N && i < N - P - 1 && bit_at(w.link, i)) {jl = w.r[i + 1] > w.r[i] + prm.dist_diff;}
      // occlusion.hpp:67-79: i in [P+1, N-1], linked pair (i, i-1), far side to the left
      if (i < N && i >= P + 1 && bit_at(w.link, i - 1)) {jr = w.r[i - 1] > w.r[i] + prm.dist_diff;}
      const int w0 = i0 + (tid & ~63);
      store_word(w.jumpL, w0, Npad, jl);
      store_word(w.jumpR, w0, Npad, jr);
    }
    __syncthreads();
  }
  for (int i0 = 0; i0 < Npad; i0 += T) {
    const int i = i0 + tid;
    uint8_t lab = kDefault;
    if (i < N) {
      lab = w.lab[i];
      if (do_occ && P > kWindowPadding) {
        // (a padding beyond the windows: the same two fills by walking the positions)
        int Lr = 0, Rr = 0;
        while (Lr < P && i - 1 - Lr >= 0 && bit_at(w.link, i - 1 - Lr)) {Lr++;}
        while (Rr < P && i + Rr < N && bit_at(w.link, i + Rr)) {Rr++;}
        bool occ = false;
        for (int j = i - 1 - Lr; j <= i - 1 && !occ; j++) {occ = j >= 0 && bit_at(w.jumpL, j);}
        for (int j = i + 1; j <= i + 1 + Rr && !occ; j++) {occ = j < N && bit_at(w.jumpR, j);}
        if (occ) {lab = kOccluded;}
      } else if (do_occ) {
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

// Runs the stencil, labelling and mask phases on a ring whose x, y are in LDS.  Returns the status.
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
    w.link[k] = 0; w.jumpL[k] = 0; w.jumpR[k] = 0; w.featE[k] = 0; w.featS[k] = 0;
    w.llink[k] = 0; w.cand[k] = 0; w.alive[k] = 0; w.sel[k] = 0; w.selE[k] = 0; w.covE[k] = 0;
    w.selS[k] = 0; w.covS[k] = 0;
  }
  if (tid < 8 && tid != kFlagUnsorted) {w.flags[tid] = 0;}
  __syncthreads();
  stencil_phase(w, prm, N, Npad, groups, curv_in, range_in);
  if (do_label) {block_check(w, prm, N, single_block);}
  __syncthreads();
  if (w.flags[kFlagZeroPair]) {return kZeroNormPair;}
  if (w.flags[kFlagBlockSmall]) {return kBlockTooSmall;}
  if (do_label) {
    if (w.flags[kFlagBlockBig] || P > kWindowPadding) {
      label_blocks_lds(w, prm, N, Npad, single_block);
    } else if (P == 5) {
      label_blocks_wave<5>(w, prm, N, single_block);
    } else if (P == 2) {
      label_blocks_wave<2>(w, prm, N, single_block);
    } else {
      label_blocks_wave<0>(w, prm, N, single_block);
    }
    __syncthreads();
  }
  mask_phase(w, prm, N, Npad, flags);
  return kOk;
}

// The count pass of the HOLES form of the organised route (a driver that keeps the grid and writes invalid returns as
// (0, 0, 0) records; the zero filter drops them, convert.py:162-163,192): position k of ring r is then the ring's k-th
// VALID column, and a unit of ring_unit_org_kernel<.., HOLES> has to know where in the grid its positions lie -- and how long
// the ring is, for its block boundaries -- before it can load anything.  One workgroup per (group of four adjacent rings,
// scan) walks the group's columns once, lane = (column of a 16-column piece, ring of the group) as the unit kernel loads
// them, and leaves per ring the exclusive prefix of its valid returns over the pieces (cum16, 2 B per ring and piece: 15 KB
// per 64 x 1800 scan) and the ring's length (ring_count).  It reads every record's ring id, so the unit kernel need not: a
// scan that is not the grid it claims to be goes on the fall-back list here.
#ifndef LFX_COUNT_UNROLL
#define LFX_COUNT_UNROLL 4
#endif
constexpr int kCountUnroll = LFX_COUNT_UNROLL;       // (pieces in flight per wave; A/B: -DLFX_COUNT_UNROLL=n)
__global__ __launch_bounds__(256) void grid_count_kernel(
  const uint8_t * __restrict__ pts, const uint32_t * __restrict__ scan_begin, const uint32_t * __restrict__ geom, uint32_t R,
  uint32_t stride /* cum_stride(ring capacity) */, uint16_t * __restrict__ cum16, uint32_t * __restrict__ ring_count,
  const UnitTables * __restrict__ tab, uint32_t * __restrict__ counters, uint4 * __restrict__ desc /* [batch][R][B] */,
  uint32_t ring_cap, uint32_t max_span /* positions a unit wave holds: 64 x chunks */, uint32_t max_pieces /* pieces a workgroup loads */)
{
  __shared__ uint32_t piece_lo[kUnitMaxBlocks], piece_hi[kUnitMaxBlocks];
  __shared__ uint16_t cnt[4][LFX_MAX_RING_POINTS / kPieceCols + 64];
  __shared__ uint32_t flag[2];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = __builtin_amdgcn_readfirstlane(tid >> 6), s = blockIdx.y;
  uint32_t g = blockIdx.x;
  {
    // (the ring group is turned by the scan index exactly as ring_unit_org_kernel turns it: every XCD sees every group)
    const uint32_t groups = gridDim.x;
    const bool pairs = (groups & 15u) == 0u;
    if (pairs) {g = (g & ~15u) | ((g & 7u) << 1) | ((g >> 3) & 1u);}
    const uint32_t turn = (pairs ? 2u * s : s) & ((1u << (31 - __builtin_clz(groups))) - 1u);
    g += turn;
    g = g >= groups ? g - groups : g;
  }
  const uint32_t C = geom[s * kGeomStride];
  if (s == 0u && blockIdx.x == 0u && tid == 0u) {counters[kCntHolesRan] = 1u;}
  if (C == 0u) {                                      // not R rings x C columns (the host looked): the bucketing route's
    if (blockIdx.x == 0u && tid == 0u) {atomicOr(tab->scan_flags + s, (uint32_t)kScanCountFell); scan_falls_back(tab, s);}
    return;
  }
  if (tid < 2u) {flag[tid] = 0u;}
  if (tid < (uint32_t)kUnitMaxBlocks) {piece_lo[tid] = 0xFFFFFFFFu; piece_hi[tid] = 0u;}
  const uint32_t n_pieces = (C + kPieceCols - 1u) / kPieceCols;
  const uint32_t sub = lane & 3u, cq = lane >> 2, r0 = 4u * g, rr = r0 + sub, rload = rr < R ? rr : R - 1u;
  const uint8_t * const base = pts + (size_t)scan_begin[s] * 32u + (size_t)rload * 32u;
  const uint64_t ring_lanes = 0x1111111111111111ull << sub;
  uint64_t wrong = 0, zeros = 0;
  for (uint32_t t0 = wave; t0 < n_pieces; t0 += 4u * kCountUnroll) {
    float4 rec[kCountUnroll];
    uint32_t rw[kCountUnroll];
#pragma unroll
    for (int u = 0; u < kCountUnroll; u++) {
      const uint32_t col = (t0 + 4u * u) * kPieceCols + cq;
      const uint8_t * p = base + (size_t)(col < C ? col : C - 1u) * R * 32u;
      rec[u] = *reinterpret_cast<const float4 *>(p);
      rw[u] = *reinterpret_cast<const uint32_t *>(p + 20);
    }
#pragma unroll
    for (int u = 0; u < kCountUnroll; u++) {
      const uint32_t t = t0 + 4u * u;
      const uint32_t col = t * kPieceCols + cq;
      const bool in = col < C && rr < R;
      const bool zero = rec[u].x == 0.f && rec[u].y == 0.f && rec[u].z == 0.f;
      const uint64_t valid = __ballot(in && !zero);
      wrong |= __ballot(in && (rw[u] & 0xFFFFu) != rr);
      zeros |= __ballot(in && zero);
      if (t < n_pieces && cq == 0u) {cnt[sub][t] = (uint16_t)__popcll(valid & ring_lanes);}
    }
  }
  if (wrong != 0ull && lane == 0u) {flag[0] = 1u;}
  if (zeros != 0ull && lane == 0u) {flag[1] = 1u;}
  __syncthreads();
  if (flag[0] != 0u) {
    if (tid == 0u) {atomicOr(tab->scan_flags + s, (uint32_t)kScanCountFell); scan_falls_back(tab, s);}
    return;
  }
  if (flag[1] != 0u && tid == 0u) {atomicAdd(counters + kCntZeroGroups, 1u);}
  __syncthreads();                                    // (flag[0] is raised again below)
  // wave w: the exclusive prefix of ring r0 + w over the pieces (kept in LDS for the searches below)
  const uint32_t ring = r0 + wave;
  uint32_t N = 0;
  if (ring < R) {
    uint16_t * const row = cum16 + ((size_t)s * R + ring) * stride;
    uint32_t carry = 0;
    for (uint32_t p0 = 0; p0 < n_pieces; p0 += 64u) {
      const uint32_t p = p0 + lane;
      const uint32_t v = p < n_pieces ? cnt[wave][p] : 0u;
      const uint32_t incl = wave_inclusive_sum(v) + carry;
      if (p < n_pieces) {row[p] = (uint16_t)(incl - v); cnt[wave][p] = (uint16_t)(incl - v);}
      carry = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
    N = carry;
    if (lane == 0u) {
      row[n_pieces] = (uint16_t)N;
      cnt[wave][n_pieces] = (uint16_t)N;
      ring_count[s * kRings + ring] = N;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // lane j: the descriptor of unit (ring, block j) -- what unit_body works out for a unit of a bucketed ring, and the pieces
  // that hold its first and its last position
  const int P = tab->prm.P, B = tab->prm.B;
  bool defer = false;
  if (ring < R && (int)lane < B) {
    const int j = (int)lane, Ni = (int)N;
    uint32_t flags = N == 0u ? kHoleUnitDead : 0u;       // (a ring without a valid return is no ring of the scan)
    int b0 = 0, b1 = 0, ps = 0, pe = 0;
    if (flags == 0u && (Ni < 2 * P + 1 || Ni - 2 * P < B || N > ring_cap)) {flags = kHoleUnitDead; defer = true;}      // skip conditions, over-long rings: the bucketing route's
    if (flags == 0u) {
      b0 = block_boundary(Ni, P, B, j);
      b1 = block_boundary(Ni, P, B, j + 1);
      const int o0 = j == 0 ? 0 : b0, o1 = j == B - 1 ? Ni : b1;
      const int g0 = o0 - (P + 1), span = o1 + (P + 1) - g0;
      if (b1 - b0 < 2 || span > (int)max_span) {
        flags = kHoleUnitDead; defer = true;
      } else {
        const int first = g0 < 0 ? 0 : g0, last = (o1 + P + 1 < Ni ? o1 + P + 1 : Ni) - 1;
        // the piece of position x: the last p with prefix[p] <= x (prefix[0] = 0, prefix[n_pieces] = N > x)
        auto piece_of = [&](int x) {
            int lo = 0, hi = (int)n_pieces;
            while (hi - lo > 1) {
              const int mid = (lo + hi) >> 1;
              if ((int)cnt[wave][mid] <= x) {lo = mid;} else {hi = mid;}
            }
            return lo;
          };
        ps = piece_of(first);
        pe = piece_of(last);
        atomicMin(&piece_lo[j], (uint32_t)ps);
        atomicMax(&piece_hi[j], (uint32_t)pe);
      }
    }
    desc[((size_t)s * R + ring) * (uint32_t)B + (uint32_t)j] =
      make_uint4(N | ((uint32_t)b0 << 16), (uint32_t)b1 | (flags << 16), (uint32_t)ps | ((uint32_t)pe << 16), 0u);
  }
  if (__ballot(defer) != 0ull && lane == 0u) {flag[0] = 1u;}
  __syncthreads();
  // (more columns between the first and the last position of a block's four units than a workgroup loads: too many holes)
  if (tid < (uint32_t)B && piece_lo[tid] != 0xFFFFFFFFu && piece_hi[tid] - piece_lo[tid] + 1u > max_pieces) {flag[0] = 1u;}
  __syncthreads();
  if (flag[0] != 0u) {
    if (tid == 0u) {atomicOr(tab->scan_flags + s, (uint32_t)kScanCountFell); scan_falls_back(tab, s);}
    return;
  }
  // the scan is the unit kernel's, read in place as a grid with holes -- unless another ring group says otherwise
  // (kScanFellBack beside these bits: the bucketing route's after all)
  if (blockIdx.x == 0u && tid == 0u) {atomicOr(tab->scan_flags + s, (uint32_t)(kScanFused | kScanHoles));}
}

// The same count pass for LARGE batches, one workgroup per SCAN: the scan's records are read as they lie -- a wave's 64 lanes
// take 64 consecutive records, 2 KB in one piece, instead of sixteen 128-byte lines 2 KB apart -- and every valid record adds one
// to its (ring, piece) word in LDS; prefixes, ring lengths and unit descriptors as grid_count_kernel writes them.  Worth it
// where there are scans enough to fill the device with one workgroup each (the host: batch >= 256); grid_count_kernel's
// workgroup per (ring group, scan) stays for small batches.  LDS: (rings + 1) x (pieces + 1) words, given by the host.
constexpr int kScanCountThreads = 1024;
__global__ __launch_bounds__(kScanCountThreads) void scan_count_kernel(
  const uint8_t * __restrict__ pts, const uint32_t * __restrict__ scan_begin, const uint32_t * __restrict__ geom, uint32_t R,
  uint32_t stride /* cum_stride(ring capacity) */, uint16_t * __restrict__ cum16, uint32_t * __restrict__ ring_count,
  const UnitTables * __restrict__ tab, uint32_t * __restrict__ counters, uint4 * __restrict__ desc /* [batch][R][B] */,
  uint32_t ring_cap, uint32_t max_span, uint32_t max_pieces, uint32_t row_words /* words per ring's row in LDS: pieces of the longest scan + 1 */)
{
  extern __shared__ __attribute__((aligned(16))) uint32_t cnt[];      // [R][row_words]; behind it piece_lo / piece_hi [groups][B]
  __shared__ uint32_t flag[2];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = __builtin_amdgcn_readfirstlane(tid >> 6), s = blockIdx.x;
  const uint32_t C = geom[s * kGeomStride];
  if (s == 0u && tid == 0u) {counters[kCntHolesRan] = 1u;}
  if (C == 0u) {                                      // not R rings x C columns (the host looked): the bucketing route's
    if (tid == 0u) {atomicOr(tab->scan_flags + s, (uint32_t)kScanCountFell); scan_falls_back(tab, s);}
    return;
  }
  const int P = tab->prm.P, B = tab->prm.B;
  const uint32_t groups = (R + 3u) / 4u;
  uint32_t * const piece_lo = cnt + (size_t)R * row_words, * const piece_hi = piece_lo + groups * (uint32_t)B;
  for (uint32_t k = tid; k < R * row_words; k += kScanCountThreads) {cnt[k] = 0u;}
  for (uint32_t k = tid; k < groups * (uint32_t)B; k += kScanCountThreads) {piece_lo[k] = 0xFFFFFFFFu; piece_hi[k] = 0u;}
  if (tid < 2u) {flag[tid] = 0u;}
  __syncthreads();
  const uint32_t n_pieces = (C + kPieceCols - 1u) / kPieceCols, n = C * R;
  const uint8_t * const base = pts + (size_t)scan_begin[s] * 32u;
  // record e = column e / R, ring e mod R; a thread's records are 1 024 apart: both numbers move by constants
  const uint32_t dcol = (uint32_t)kScanCountThreads / R, dring = (uint32_t)kScanCountThreads % R;
  uint32_t col = tid / R, ring = tid % R;
  bool wrong = false, zeros = false;
  for (uint32_t e0 = tid; e0 < n; e0 += kCountUnroll * kScanCountThreads) {
    float4 rec[kCountUnroll];
    uint32_t rw[kCountUnroll];
#pragma unroll
    for (int u = 0; u < kCountUnroll; u++) {
      const uint32_t e = e0 + (uint32_t)u * kScanCountThreads;
      const uint8_t * p = base + (size_t)(e < n ? e : n - 1u) * 32u;
      rec[u] = *reinterpret_cast<const float4 *>(p);
      rw[u] = *reinterpret_cast<const uint32_t *>(p + 20);
    }
#pragma unroll
    for (int u = 0; u < kCountUnroll; u++) {
      const uint32_t e = e0 + (uint32_t)u * kScanCountThreads;
      if (e < n) {
        const bool zero = rec[u].x == 0.f && rec[u].y == 0.f && rec[u].z == 0.f;
        wrong = wrong || (rw[u] & 0xFFFFu) != ring;
        zeros = zeros || zero;
        if (!zero) {atomicAdd(&cnt[ring * row_words + (col >> 4)], 1u);}
      }
      col += dcol; ring += dring;
      if (ring >= R) {ring -= R; col += 1u;}
    }
  }
  if (__ballot(wrong) != 0ull && lane == 0u) {flag[0] = 1u;}
  if (__ballot(zeros) != 0ull && lane == 0u) {flag[1] = 1u;}
  __syncthreads();
  if (flag[0] != 0u) {
    if (tid == 0u) {atomicOr(tab->scan_flags + s, (uint32_t)kScanCountFell); scan_falls_back(tab, s);}
    return;
  }
  if (flag[1] != 0u && tid == 0u) {atomicAdd(counters + kCntZeroGroups, groups);}      // (counted in ring groups, as grid_count_kernel counts)
  // wave w: the exclusive prefixes of rings w, w + 16, ... over the pieces (kept in LDS for the searches below)
  for (uint32_t r = wave; r < R; r += kScanCountThreads / 64) {
    uint16_t * const row = cum16 + ((size_t)s * R + r) * stride;
    uint32_t * const lrow = cnt + (size_t)r * row_words;
    uint32_t carry = 0;
    for (uint32_t p0 = 0; p0 < n_pieces; p0 += 64u) {
      const uint32_t p = p0 + lane;
      const uint32_t v = p < n_pieces ? lrow[p] : 0u;
      const uint32_t incl = wave_inclusive_sum(v) + carry;
      if (p < n_pieces) {row[p] = (uint16_t)(incl - v); lrow[p] = incl - v;}
      carry = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
    if (lane == 0u) {
      row[n_pieces] = (uint16_t)carry;
      lrow[n_pieces] = carry;
      ring_count[s * kRings + r] = carry;
    }
  }
  __syncthreads();
  // thread (ring, block j): the unit's descriptor, as grid_count_kernel
  bool defer = false;
  for (uint32_t k = tid; k < R * (uint32_t)B; k += kScanCountThreads) {
    const uint32_t r = k / (uint32_t)B;
    const int j = (int)(k - r * (uint32_t)B);
    const uint32_t * const lrow = cnt + (size_t)r * row_words;
    const uint32_t N = lrow[n_pieces];
    const int Ni = (int)N;
    uint32_t flags = N == 0u ? kHoleUnitDead : 0u;       // (a ring without a valid return is no ring of the scan)
    int b0 = 0, b1 = 0, ps = 0, pe = 0;
    if (flags == 0u && (Ni < 2 * P + 1 || Ni - 2 * P < B || N > ring_cap)) {flags = kHoleUnitDead; defer = true;}
    if (flags == 0u) {
      b0 = block_boundary(Ni, P, B, j);
      b1 = block_boundary(Ni, P, B, j + 1);
      const int o0 = j == 0 ? 0 : b0, o1 = j == B - 1 ? Ni : b1;
      const int g0 = o0 - (P + 1), span = o1 + (P + 1) - g0;
      if (b1 - b0 < 2 || span > (int)max_span) {
        flags = kHoleUnitDead; defer = true;
      } else {
        const int first = g0 < 0 ? 0 : g0, last = (o1 + P + 1 < Ni ? o1 + P + 1 : Ni) - 1;
        auto piece_of = [&](int x) {
            int lo = 0, hi = (int)n_pieces;
            while (hi - lo > 1) {
              const int mid = (lo + hi) >> 1;
              if ((int)lrow[mid] <= x) {lo = mid;} else {hi = mid;}
            }
            return lo;
          };
        ps = piece_of(first);
        pe = piece_of(last);
        atomicMin(&piece_lo[(r >> 2) * (uint32_t)B + (uint32_t)j], (uint32_t)ps);
        atomicMax(&piece_hi[(r >> 2) * (uint32_t)B + (uint32_t)j], (uint32_t)pe);
      }
    }
    desc[((size_t)s * R + r) * (uint32_t)B + (uint32_t)j] =
      make_uint4(N | ((uint32_t)b0 << 16), (uint32_t)b1 | (flags << 16), (uint32_t)ps | ((uint32_t)pe << 16), 0u);
  }
  if (__ballot(defer) != 0ull && lane == 0u) {flag[0] = 1u;}
  __syncthreads();
  for (uint32_t k = tid; k < groups * (uint32_t)B; k += kScanCountThreads) {
    if (piece_lo[k] != 0xFFFFFFFFu && piece_hi[k] - piece_lo[k] + 1u > max_pieces) {flag[0] = 1u;}
  }
  __syncthreads();
  if (tid == 0u) {
    if (flag[0] != 0u) {
      atomicOr(tab->scan_flags + s, (uint32_t)kScanCountFell);
      scan_falls_back(tab, s);
    } else {
      atomicOr(tab->scan_flags + s, (uint32_t)(kScanFused | kScanHoles));
    }
  }
}

// The ring transforms of an organised stream whose rings do not arrive in angle order: a driver that starts its scans
// at another azimuth delivers every ring as a ROTATION of its sorted order, a clockwise sensor as its REVERSE (or
// both).  One wave per ring: direction by majority over 64 sampled adjacent pairs, then the column of the ring's
// smallest angle by a 64-ary search for the wrap (two or three rounds of loads) with the exact
// predicate (ring.hpp:54-99).  Nothing is moved: ring_unit_org_kernel<XF> applies the transform in its loads and still
// verifies every adjacent pair, so a ring that is not a rotation / reversal of its sorted order falls back as before.
constexpr int kCutThreads = 256, kCutWindow = 32;
__global__ __launch_bounds__(kCutThreads) void ring_cut_kernel(
  const uint8_t * __restrict__ pts, const uint32_t * __restrict__ scan_begin, uint32_t max_rings, uint32_t ring_cap,
  uint32_t * __restrict__ xform, uint32_t * __restrict__ counters)
{
  // One workgroup per scan.  Ring 0 gets the full search (one wave); the rings of one scan start within a few columns
  // of each other (per-laser azimuth offsets), so for the others the 32 columns around ring 0's wrap are read for ALL
  // rings at once -- whole columns, i.e. contiguous records -- and a ring whose wrap is not cleanly inside that window
  // gets the full search too.
  const uint32_t s = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (s == 0u && tid == 0u) {counters[kCntCutRan] = 1u;}                            // this batch's report was made with the transforms on
  const uint32_t first = scan_begin[s], n = scan_begin[s + 1] - first;
  const uint32_t C = n / max_rings;
  if (C * max_rings != n || C < 64u || C > ring_cap) {return;}                      // not this kernel's kind of scan: identity stays
  __shared__ uint32_t lead, n_hard;
  __shared__ uint32_t mask[kRings];
  __shared__ float2 first_xy[kRings];
  __shared__ uint16_t hard[kRings];
  auto xy = [&](uint32_t ring, uint32_t c) {
      return *reinterpret_cast<const float2 *>(pts + ((size_t)first + (size_t)c * max_rings + ring) * 32u);
    };
  // "column c lies beyond the wrap": forward rings -- its angle is below a[0]'s; reversed -- above it
  auto beyond = [&](bool rev, float2 a0, float2 v) {
      return rev ? polar_less(a0.x, a0.y, v.x, v.y) : polar_less(v.x, v.y, a0.x, a0.y);
    };
  // first column >= 1 beyond the wrap (C: none) -> transform
  auto to_xform = [&](bool rev, uint32_t hi) {
      return rev ? (kXformReversed | (hi - 1u)) : (hi < C ? hi : 0u);
    };
  auto full_search = [&](uint32_t ring) {                                            // by one wave
      // direction: adjacent pairs at 8 spread columns; all but (at most) one of them do not straddle the wrap
      bool up = false, down = false;
      if (lane < 8u) {
        const uint32_t cs = (lane * (C - 1u)) >> 3;                                    // in [0, C - 2]
        const float2 a = xy(ring, cs), b = xy(ring, cs + 1u);
        up = polar_less(a.x, a.y, b.x, b.y);
        down = polar_less(b.x, b.y, a.x, a.y);
      }
      const bool rev = __popcll(__ballot(down)) > __popcll(__ballot(up));
      const float2 a0 = xy(ring, 0u);
      uint32_t lo = 0u, hi = C;                                                       // P(lo) false, P(hi) true (C: sentinel)
      while (hi - lo > 1u) {
        const uint32_t step = (hi - lo + 63u) >> 6;
        const uint32_t c = lo + (lane + 1u) * step;
        bool by = true;
        if (c < hi) {by = beyond(rev, a0, xy(ring, c));}
        const uint64_t m = __ballot(by);                                              // lanes with c >= hi vote true: m != 0
        const uint32_t f = (uint32_t)__ffsll((long long)m) - 1u;
        const uint32_t nhi = lo + (f + 1u) * step;
        lo = lo + f * step;
        hi = nhi < hi ? nhi : hi;
      }
      return to_xform(rev, hi);
    };
  if (tid < kRings) {mask[tid] = 0u;}
  if (tid == 0u) {n_hard = 0u;}
  if (wave == 0u) {
    const uint32_t xf0 = full_search(0u);
    if (lane == 0u) {lead = xf0;}
  }
  for (uint32_t r = tid; r < max_rings; r += kCutThreads) {first_xy[r] = xy(r, 0u);}   // column 0: contiguous records
  __syncthreads();
  const uint32_t xf0 = lead;
  const bool rev0 = (xf0 & kXformReversed) != 0u;
  const uint32_t start0 = xf0 & ~kXformReversed;
  const uint32_t hi0 = rev0 ? start0 + 1u : (start0 == 0u ? C : start0);            // ring 0's first column beyond the wrap
  for (uint32_t e = tid; e < (uint32_t)kCutWindow * max_rings; e += kCutThreads) {
    const uint32_t w = e / max_rings, r = e % max_rings;
    const int c = (int)hi0 - kCutWindow / 2 + (int)w;
    bool by = c >= (int)C;                                                           // c <= 0: before the wrap by definition
    if (c >= 1 && c < (int)C) {by = beyond(rev0, first_xy[r], xy(r, (uint32_t)c));}
    if (by) {atomicOr(&mask[r], 1u << w);}
  }
  __syncthreads();
  for (uint32_t r = tid; r < max_rings; r += kCutThreads) {
    const uint32_t m = mask[r];
    const uint32_t f = m ? (uint32_t)__ffs((int)m) - 1u : 32u;
    if (r == 0u) {
      xform[s * kRings] = xf0;
      if (xf0 != 0u) {atomicAdd(counters + kCntTurned, 1u);}
    } else if (f >= 1u && f < 32u && m == (0xFFFFFFFFu << f)) {                      // one clean false -> true step inside the window
      const uint32_t xf = to_xform(rev0, (uint32_t)((int)hi0 - kCutWindow / 2 + (int)f));
      xform[s * kRings + r] = xf;
      if (xf != 0u) {atomicAdd(counters + kCntTurned, 1u);}
    } else {
      hard[atomicAdd(&n_hard, 1u)] = (uint16_t)r;
    }
  }
  __syncthreads();
  for (uint32_t k = wave; k < n_hard; k += kCutThreads / 64) {
    const uint32_t r = hard[k];
    const uint32_t xf = full_search(r);
    if (lane == 0u) {
      xform[s * kRings + r] = xf;
      if (xf != 0u) {atomicAdd(counters + kCntTurned, 1u);}
    }
  }
}

// ------------------------------------------------------------------------------------------
// Order repair.  One workgroup per ring the first pass found out of angle order.  A ring that is a
// ROTATION of its sorted order (a driver starting the scan at another azimuth), its REVERSE (a
// clockwise sensor) or both is put in order by index arithmetic; anything else by a bitonic sort
// in LDS with the exact predicate and the arrival index as tie-break (ring.hpp:54-112; canonical
// order where the reference's unstable std::sort leaves ties open).  The ring's slices of
// sxy / sz / sidx are rewritten in sorted order; rings whose only problem was the order go on the
// redo list (second pass of ring_unit_kernel), the others on the slow list.
__global__ __launch_bounds__(512) void ring_order_kernel(
  uint32_t cap, uint32_t max_rings, const uint32_t * __restrict__ ring_count, float2 * __restrict__ sxy,
  float * __restrict__ sz, uint32_t * __restrict__ sidx, uint32_t * __restrict__ ring_flags,
  const uint32_t * __restrict__ defer_count, const uint32_t * __restrict__ defer_list,
  uint32_t * __restrict__ redo_count, uint32_t * __restrict__ redo_list, uint32_t * __restrict__ slow_count,
  uint32_t * __restrict__ slow_list, uint32_t all_rings /* 1: run BEFORE the unit kernel over every ring of the scans on the fall-back list */,
  uint32_t * __restrict__ pre_fixed, uint32_t redo_cap, const uint32_t * __restrict__ fb_count,
  const uint32_t * __restrict__ fb_list, uint32_t list_cover /* all_rings = 0: list entries the first unit pass was launched for */)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
  const int T = blockDim.x, tid = threadIdx.x;
  if (!all_rings && fb_count != nullptr && tid == 0) {
    // scans on the fall-back list beyond what the first unit pass covered: all their rings to the workgroup-per-ring kernel
    const uint32_t n_list = *fb_count;
    if (n_list > list_cover) {
      const uint32_t extra = (n_list - list_cover) * max_rings;
      for (uint32_t item = blockIdx.x; item < extra; item += gridDim.x) {
        const uint32_t e = fb_list[list_cover + item / max_rings] * kRings + item % max_rings;
        if (ring_count[e] != 0u) {slow_list[atomicAdd(slow_count, 1u)] = e;}
      }
    }
  }
  const uint32_t M = cap;                             // (a multiple of 64; the sorting networks need no power of two: bitonic_partner)
  float * lx = reinterpret_cast<float *>(lds_raw);
  float * ly = lx + M;
  uint32_t * li = reinterpret_cast<uint32_t *>(ly + M);   // arrival index (tie-break)
  uint32_t * lp = li + M;                                 // position in the ring as bucketed
  float * lz = reinterpret_cast<float *>(lp + M);
  uint32_t * ls = reinterpret_cast<uint32_t *>(lz + M);   // sidx as bucketed
  int * cnt = reinterpret_cast<int *>(ls + M);            // [8] counters
  uint32_t * hist = reinterpret_cast<uint32_t *>(cnt + 8);  // [kOrderBuckets + 8]: bucket counts, then their prefix; the waves' totals
  // Two uses.  After the first unit pass (all_rings = 0): the rings on the defer list.  Before it (all_rings = 1,
  // switched on by the host while a stream keeps arriving out of order): every ring of the listed scans is put in order
  // here -- a rotation / reversal undone, anything else sorted -- so that the first pass takes the ring and no second
  // pass is needed; rings in order are only read.
  const uint32_t n_items = all_rings ? *fb_count * max_rings : *defer_count;
  for (uint32_t item = blockIdx.x; item < n_items; item += gridDim.x) {
    const uint32_t e = all_rings ? fb_list[item / max_rings] * kRings + item % max_rings : defer_list[item];
    const uint32_t s = e / kRings, slot = e % kRings;
    const int N = (int)ring_count[e];
    const uint32_t reason = all_rings ? (uint32_t)kDeferOrder : ring_flags[e];
    const size_t off = ring_base(s, slot, max_rings, cap);
    // (a ring this kernel already put in order is not sorted a second time: where the float predicate is not a
    // consistent order on nearly parallel points a second sort could differ)
    const bool fixable = (reason & kDeferOrder) && !(reason & kRingSorted) && N >= 2 && (uint32_t)N <= cap;
    if (fixable) {
      if (tid < 8) {cnt[tid] = tid == 1 || tid == 3 ? -1 : 0;}
      for (int i0 = 0; i0 < N; i0 += 4 * T) {           // twelve loads per thread in flight, then the LDS stores
        float2 v[4];
        float vz[4];
        uint32_t vs[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int i = i0 + u * T + tid;
          v[u] = make_float2(0.f, 0.f); vz[u] = 0.f; vs[u] = 0u;
          if (i < N) {v[u] = sxy[off + i]; vz[u] = sz[off + i]; vs[u] = sidx[off + i];}
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int i = i0 + u * T + tid;
          if (i < N) {lx[i] = v[u].x; ly[i] = v[u].y; lz[i] = vz[u]; ls[i] = vs[u];}
        }
      }
      __syncthreads();
      // cnt[0] / cnt[1]: number / position of pairs that are not strictly increasing,
      // cnt[2] / cnt[3]: the same for not strictly decreasing
      // (counted per wave with ballots: in a sorted or reversed ring EVERY pair fails one of the two tests, and
      // that many same-address LDS atomics serialise)
      for (int i0 = 0; i0 + 1 < N; i0 += T) {
        const int i = i0 + tid;
        bool up = false, down = false;
        if (i + 1 < N) {
          up = !polar_less(lx[i], ly[i], lx[i + 1], ly[i + 1]);
          down = !polar_less(lx[i + 1], ly[i + 1], lx[i], ly[i]);
        }
        const uint64_t mu = __ballot(up), md = __ballot(down);
        if ((tid & 63) == 0) {
          const int w0 = i0 + (tid & ~63);
          if (mu) {atomicAdd(&cnt[0], __popcll(mu)); atomicMax(&cnt[1], w0 + 63 - __clzll(mu));}
          if (md) {atomicAdd(&cnt[2], __popcll(md)); atomicMax(&cnt[3], w0 + 63 - __clzll(md));}
        }
      }
      __syncthreads();
      const int up_breaks = cnt[0], up_at = cnt[1], down_breaks = cnt[2], down_at = cnt[3];
      int mode = 0;                                     // 0 sort, 1 rotation, 2 reverse, 3 reversed rotation
      int cut = 0;
      __syncthreads();                                  // everyone has read cnt[0..3]; cnt[4] <- mode below
      if (up_breaks == 1 && polar_less(lx[up_at + 1], ly[up_at + 1], lx[up_at], ly[up_at]) &&
        polar_less(lx[N - 1], ly[N - 1], lx[0], ly[0]))
      {
        mode = 1; cut = up_at + 1;                      // sorted = [cut .. N-1] then [0 .. cut-1]
      } else if (down_breaks == 0) {
        mode = 2;                                       // strictly decreasing: sorted = reverse
      } else if (down_breaks == 1 && polar_less(lx[down_at], ly[down_at], lx[down_at + 1], ly[down_at + 1]) &&
        polar_less(lx[0], ly[0], lx[N - 1], ly[N - 1]))
      {
        mode = 3; cut = down_at + 1;                    // two decreasing runs: sorted = reverse([0..cut-1]) then reverse([cut..N-1])
      }
      if (tid == 0) {cnt[4] = mode;}
      if (all_rings && up_breaks == 0) {
        // in order already
      } else if (mode == 0) {
        uint32_t Ms = 1;
        while (Ms < (uint32_t)N) {Ms <<= 1;}
        // First by KEY: (polar_key, position as bucketed = order of arrival) as one 64-bit integer per point, sorted by
        // the same bitonic network -- two 8-byte LDS reads and an integer compare per exchange instead of six reads and
        // the predicate twice -- then every adjacent pair of the result is put to the exact predicate (with the arrival
        // index as tie-break: a strict total order, so a sequence whose adjacent pairs all pass IS the sorted one).  Only
        // a ring that fails (points a few ulp apart in angle) is sorted again by the predicate itself, below.
        uint64_t * k64 = reinterpret_cast<uint64_t *>(li);          // over li and lp
        // The keys are placed by COUNTING first: a histogram over kOrderBuckets slices of the angle, its prefix, every
        // key to its bucket's range -- the points of a ring are spread over the angle, a bucket holds one or two of them --
        // and what is left inside and between neighbouring buckets by odd-even exchanges of adjacent keys until a round
        // of both parities moves nothing: a dozen barriers where the bitonic network has sixty-six, each over the whole
        // ring.  A ring that is not done after kOrderRounds rounds (its points crowd into few buckets) takes the network.
        bool placed = false;
        if ((uint32_t)N <= 8u * (uint32_t)T) {
          constexpr int kOrderRounds = 8;
          uint32_t key[8], old[8], bkt[8];
          for (int b = tid; b < kOrderBuckets; b += T) {hist[b] = 0u;}
          __syncthreads();
#pragma unroll
          for (int u = 0; u < 8; u++) {
            const int i = tid + u * T;
            key[u] = 0u; old[u] = 0u; bkt[u] = 0u;
            if (i < N) {
              const float t = polar_pseudo_angle(lx[i], ly[i]);
              key[u] = polar_key(t);
              bkt[u] = polar_bucket(t);
              old[u] = atomicAdd(&hist[bkt[u]], 1u);
            }
          }
          __syncthreads();
          {
            // exclusive prefix of the bucket counts: kOrderBuckets / T per thread, along the lanes, across the waves
            constexpr int kPer = kOrderBuckets / 512;
            static_assert(kOrderBuckets % 512 == 0, "buckets per thread");
            uint32_t c[kPer], sum = 0;
            if (tid < 512) {
#pragma unroll
              for (int q = 0; q < kPer; q++) {c[q] = hist[tid * kPer + q]; sum += c[q];}
            }
            const uint32_t incl = wave_inclusive_sum(sum);
            if ((tid & 63) == 63) {hist[kOrderBuckets + (tid >> 6)] = incl;}
            __syncthreads();
            uint32_t base = 0;
            for (int w2 = 0; w2 < (tid >> 6); w2++) {base += hist[kOrderBuckets + w2];}
            if (tid < 512) {
              uint32_t run = base + incl - sum;
#pragma unroll
              for (int q = 0; q < kPer; q++) {hist[tid * kPer + q] = run; run += c[q];}
            }
            __syncthreads();
          }
#pragma unroll
          for (int u = 0; u < 8; u++) {
            const int i = tid + u * T;
            if (i < N) {k64[hist[bkt[u]] + old[u]] = ((uint64_t)key[u] << 32) | (uint32_t)i;}
          }
          __syncthreads();
          for (int round = 0; round < kOrderRounds && !placed; round++) {
            bool moved = false;
#pragma unroll
            for (int par = 0; par < 2; par++) {
              for (int i = 2 * tid + par; i + 1 < N; i += 2 * T) {
                const uint64_t a = k64[i], b = k64[i + 1];
                if (b < a) {k64[i] = b; k64[i + 1] = a; moved = true;}
              }
              __syncthreads();
            }
            placed = !__syncthreads_or(moved);
          }
        }
        if (!placed) {
        for (uint32_t i = tid; i < (uint32_t)N; i += T) {
          k64[i] = ((uint64_t)polar_key(polar_pseudo_angle(lx[i], ly[i])) << 32) | i;
        }
        __syncthreads();
        for (uint32_t k = 2; k <= Ms; k <<= 1) {                     // (all-ascending form over N elements: bitonic_partner)
          for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = tid; t < Ms / 2; t += T) {
              uint32_t i, p;
              bitonic_partner(t, k, j, i, p);
              if (p < (uint32_t)N) {
                const uint64_t a = k64[i], b = k64[p];
                if (b < a) {k64[i] = b; k64[p] = a;}
              }
            }
            __syncthreads();
          }
        }
        }
        bool out_of_order = false;
        for (int i = tid; i + 1 < N; i += T) {
          const uint32_t a = (uint32_t)k64[i], b = (uint32_t)k64[i + 1];
          if (!sort_less(lx[a], ly[a], ls[a], lx[b], ly[b], ls[b])) {out_of_order = true;}
        }
        if (!__syncthreads_or(out_of_order)) {
          for (int i = tid; i < N; i += T) {
            const uint32_t src = (uint32_t)k64[i];
            sxy[off + i] = make_float2(lx[src], ly[src]);
            sz[off + i] = lz[src];
            sidx[off + i] = ls[src];
          }
        } else {
        for (uint32_t i = tid; i < (uint32_t)N; i += T) {
          li[i] = ls[i];
          lp[i] = i;
        }
        __syncthreads();
        for (uint32_t k = 2; k <= Ms; k <<= 1) {
          for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = tid; t < Ms / 2; t += T) {
              uint32_t i, p;
              bitonic_partner(t, k, j, i, p);
              if (p < (uint32_t)N) {
                const float ax = lx[i], ay = ly[i], bx = lx[p], by = ly[p];
                const uint32_t ai = li[i], bi = li[p];
                if (sort_less(bx, by, bi, ax, ay, ai)) {
                  const uint32_t pa = lp[i], pb = lp[p];
                  lx[i] = bx; ly[i] = by; li[i] = bi; lp[i] = pb;
                  lx[p] = ax; ly[p] = ay; li[p] = ai; lp[p] = pa;
                }
              }
            }
            __syncthreads();
          }
        }
        for (int i = tid; i < N; i += T) {
          const uint32_t src = lp[i];
          sxy[off + i] = make_float2(lx[i], ly[i]);
          sz[off + i] = lz[src];
          sidx[off + i] = li[i];
        }
        }
      } else {
        for (int i = tid; i < N; i += T) {
          int src;
          if (mode == 1) {
            src = i + cut < N ? i + cut : i + cut - N;
          } else if (mode == 2) {
            src = N - 1 - i;
          } else {
            src = i < cut ? cut - 1 - i : N - 1 - (i - cut);
          }
          sxy[off + i] = make_float2(lx[src], ly[src]);
          sz[off + i] = lz[src];
          sidx[off + i] = ls[src];
        }
      }
    }
    if (all_rings) {
      if (tid == 0 && fixable && cnt[0] != 0) {
        ring_flags[e] = kRingSorted;
        atomicAdd(pre_fixed, 1u);
      }
      __syncthreads();
      continue;
    }
    if (tid == 0) {
      // kRingSorted: the ring is sorted exactly once from its bucketed order, also when it ends up in
      // the workgroup-per-ring kernel later (a second sort could differ where the float predicate is
      // not a consistent order on nearly parallel points)
      // (the second pass is launched for redo_cap rings -- what earlier batches needed, with room to spare;
      // a ring beyond that takes the workgroup-per-ring kernel: slower, same result)
      uint32_t at = 0xFFFFFFFFu;
      if (fixable && reason == kDeferOrder) {at = atomicAdd(redo_count, 1u);}
      if (at < redo_cap) {
        ring_flags[e] = kRingSorted;
        redo_list[at] = e;
      } else {
        if (fixable) {ring_flags[e] = reason | kRingSorted;}
        slow_list[atomicAdd(slow_count, 1u)] = e;
      }
    }
    __syncthreads();
  }
}

__host__ __device__ inline size_t order_lds_bytes(uint32_t cap)
{
  return (size_t)cap * 24 + 64 + (kOrderBuckets + 8) * 4;      // (the networks store a ring's N elements, not the power of two above)
}

// ------------------------------------------------------------------------------------------
// Ring kernel, slow path: one workgroup per ring, ring resident in LDS.  Takes the rings the fast path
// deferred (use_list) or every ring of the batch (n_blocks > 64, debugging).  Writes the ring's
// feature records as ONE segment (edge from the front of the ring, surface from its back) and
// marks the ring so that feature_compact_kernel reads it that way.
struct RingExtractArgs
{
  Params prm;
  uint32_t cap, stage_flags, max_rings;
  const uint8_t * __restrict__ pts;
  Layout L;
  const uint32_t * __restrict__ scan_begin;
  const uint32_t * __restrict__ ring_count;
  float2 * __restrict__ sxy;
  const float * __restrict__ sz;
  uint32_t * __restrict__ sidx;
  uint8_t * __restrict__ label_s;
  double * __restrict__ curv_s;
  float4 * __restrict__ rec_pts;
  uint32_t * __restrict__ rec_idx;
  uint8_t * __restrict__ ring_status;
  uint32_t * __restrict__ unit_ne, * __restrict__ unit_ns, * __restrict__ unit_span, * __restrict__ ring_flags;
};

// One ring (scan s, ring id slot) by the whole workgroup.  sorted_already: ring_order_kernel has put the ring in angle
// order; mark: leave the ring's flag word saying the ring was taken here (the list routes; the fall-back tail leaves the
// flags as it found them, so that no reset has to follow it).
__device__ inline void extract_ring(const RingExtractArgs & a, uint8_t * lds_raw, uint32_t s, uint32_t slot, bool sorted_already, bool mark)
{
  const int T = blockDim.x, tid = threadIdx.x;
  const uint32_t cap = a.cap, max_rings = a.max_rings;
  const int N = (int)a.ring_count[s * kRings + slot];
  const size_t sb = a.scan_begin[s];
  const size_t off = ring_base(s, slot, max_rings, cap);
  const size_t ui = ((size_t)s * kRings + slot) * kUnitMaxBlocks;
  uint8_t status = kOk;
  bool resorted = false;
  RingWork w = carve(lds_raw, cap);
  if ((uint32_t)N > cap) {
    status = kTooLarge;
  } else {
    for (int i = tid; i < N; i += T) {
      const float2 v = a.sxy[off + i];
      w.x[i] = v.x;
      w.y[i] = v.y;
    }
    __syncthreads();
    if (!sorted_already) {
      resorted = angle_sort(w, N, a.sxy + off, a.sidx + off);
    }
    status = process_ring(w, a.prm, N, a.stage_flags, nullptr, nullptr, nullptr);
  }
  // the whole ring is ONE unit with one segment of records: [0, N)
  if (tid < kUnitMaxBlocks) {a.unit_ne[ui + tid] = 0; a.unit_ns[ui + tid] = 0; a.unit_span[ui + tid] = 0;}
  if (tid == 0) {
    if (mark) {a.ring_flags[s * kRings + slot] = 1u;}
    a.unit_span[ui] = ((uint32_t)((uint32_t)N < cap ? N : (int)cap) << 16);
  }
  if (status != kOk) {
    // the ring contributes nothing (feature_extraction.cpp:116,154-156)
    const int stored = (uint32_t)N < cap ? N : (int)cap;
    for (int i = tid; i < stored; i += T) {
      a.label_s[off + i] = kDefault;
      if (a.curv_s != nullptr) {a.curv_s[off + i] = 0.;}
    }
    if (tid == 0) {a.ring_status[s * kRings + slot] = status;}
    __syncthreads();
    return;
  }
  __syncthreads();
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
      a.ring_status[s * kRings + slot] = kOk;
      a.unit_ne[ui] = ce;
      a.unit_ns[ui] = cs;
    }
  }
  __syncthreads();
  const bool reload_xy = w.flags[kFlagXYClobbered] != 0;
  for (int i = tid; i < N; i += T) {
    const uint8_t lab = w.lab[i];
    const double c = w.c[i];
    a.label_s[off + i] = lab;
    if (a.curv_s != nullptr) {a.curv_s[off + i] = c;}
    if (lab == kEdge || lab == kSurface) {
      const uint64_t below = (1ull << (i & 63)) - 1ull;
      const uint32_t orig = a.sidx[off + i];
      // AppendXYZIR (label.hpp:166-179): x, y, z and intensity <- (float)curvature
      const float z = resorted ? load_f32(a.pts + (sb + orig) * a.L.step + a.L.oz, a.L.be) : a.sz[off + i];
      float x = w.x[i], y = w.y[i];
      if (reload_xy) {const float2 v = a.sxy[off + i]; x = v.x; y = v.y;}
      size_t at;
      if (lab == kEdge) {
        at = off + w.wbase[i >> 6] + __popcll(w.featE[(i >> 6) + 1] & below);
      } else {
        at = off + N - 1 - (w.wbase[cap / 64 + (i >> 6)] + __popcll(w.featS[(i >> 6) + 1] & below));
      }
      a.rec_pts[at] = make_float4(x, y, z, (float)c);
      a.rec_idx[at] = orig;
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(512) void ring_extract_kernel(
  RingExtractArgs a, uint32_t use_list, const uint32_t * __restrict__ slow_count, const uint32_t * __restrict__ slow_list)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
  const uint32_t max_rings = a.max_rings;
  // use_list: 0 = every ring of the batch, grid (max_rings, batch); 1 = the rings on the slow list; 2 = every ring of
  // the SCANS listed (slow_count / slow_list are the fall-back list then)
  const uint32_t n_items = use_list == 2u ? *slow_count * max_rings : (use_list ? *slow_count : 1u);
  for (uint32_t item = use_list ? blockIdx.x : 0u; item < n_items; item += use_list ? gridDim.x : 1u) {
    uint32_t slot, s;
    if (use_list == 2u) {
      s = slow_list[item / max_rings];
      slot = item % max_rings;
      if (a.ring_count[s * kRings + slot] == 0u) {continue;}
    } else if (use_list) {
      const uint32_t e = slow_list[item];
      s = e / kRings;
      slot = e % kRings;
    } else {
      slot = blockIdx.x;
      s = blockIdx.y;
      if (slot >= max_rings || a.ring_count[s * kRings + slot] == 0u) {return;}
    }
    extract_ring(a, lds_raw, s, slot, use_list && (a.ring_flags[s * kRings + slot] & kRingSorted), true);
  }
}

// ------------------------------------------------------------------------------------------
// The fall-back tail of the organised route: ONE launch behind ring_unit_org_kernel that does nothing -- its first
// instruction reads the length of the fall-back list -- on the batches of a stream that kernel keeps taking, and redoes the
// odd scan out whole otherwise: workgroup (x, y) buckets chunk x of list entries y, y + gridDim.y, ... exactly as
// ring_scatter_kernel does; the workgroup that finishes a scan's LAST chunk (a ticket per scan) then takes the scan's rings
// one after the other as ring_extract_kernel would, and leaves the look-back flags and the ticket as it found them: zero.
// Nobody waits for anybody beyond the look-back of the bucketing itself, whatever the grid.  (Until round 6 this was
// ring_scatter_kernel + ring_extract_kernel, two launches of ~6 us that found an empty list; a stream that does fall back
// is moved to the five-launch route by choose_route after its first report.)
struct ScatterArgs
{
  const uint8_t * __restrict__ pts;
  Layout L;
  const uint32_t * __restrict__ scan_begin;
  uint32_t * __restrict__ chunk_base, * __restrict__ chunk_flags, * __restrict__ ring_count, * __restrict__ scan_info, * __restrict__ scan_flags;
  float2 * __restrict__ sxy;
  float * __restrict__ sz;
  uint32_t * __restrict__ sidx;
  uint32_t max_chunks, max_rings, cap, drop_zero;
};
constexpr uint32_t kTailMaxTurns = 128;       // list entries a workgroup may have to take (the host sizes the grid for it)

template<bool CANON>
__global__ __launch_bounds__(kChunkThreads, 1) void fallback_tail_kernel(
  ScatterArgs sc, RingExtractArgs ex, const uint32_t * __restrict__ fb_count, const uint32_t * __restrict__ fb_list,
  uint32_t * __restrict__ tail_ticket /* [max_batch], zero between batches */)
{
  const uint32_t n_list = *fb_count;
  if (n_list == 0u) {return;}
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
  __shared__ uint32_t mine[kTailMaxTurns / 32];
  const uint32_t tid = threadIdx.x;
  if (tid < kTailMaxTurns / 32) {mine[tid] = 0u;}
  __syncthreads();
  uint32_t turn = 0;
  for (uint32_t it = blockIdx.y; it < n_list && turn < kTailMaxTurns; it += gridDim.y, turn++) {
    const uint32_t s = fb_list[it];
    const uint32_t n = sc.scan_begin[s + 1] - sc.scan_begin[s];
    const uint32_t n_chunks = (n + kChunkPoints - 1) / kChunkPoints;
    if (blockIdx.x < n_chunks) {
      scatter_chunk<CANON>(s, sc.pts, sc.L, sc.scan_begin, sc.chunk_base, sc.chunk_flags, sc.ring_count, sc.scan_info, sc.scan_flags,
        sc.sxy, sc.sz, sc.sidx, sc.max_chunks, sc.max_rings, sc.cap, sc.drop_zero, nullptr);
      // everything this chunk staged is out before its ticket is drawn; the last to draw sees every chunk's
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __syncthreads();
      if (tid == 0u) {
        const uint32_t t = __hip_atomic_fetch_add(&tail_ticket[s], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (t == n_chunks - 1u) {mine[turn >> 5] |= 1u << (turn & 31u);}
      }
    }
    __syncthreads();                        // the LDS blocks are reused by the next entry
  }
  turn = 0;
  for (uint32_t it = blockIdx.y; it < n_list && turn < kTailMaxTurns; it += gridDim.y, turn++) {
    if (((mine[turn >> 5] >> (turn & 31u)) & 1u) == 0u) {continue;}
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const uint32_t s = fb_list[it];
    for (uint32_t slot = 0; slot < sc.max_rings; slot++) {
      if (ex.ring_count[s * kRings + slot] != 0u) {extract_ring(ex, lds_raw, s, slot, false, false);}
    }
    const uint32_t n = sc.scan_begin[s + 1] - sc.scan_begin[s];
    const uint32_t n_chunks = (n + kChunkPoints - 1) / kChunkPoints;
    for (uint32_t k = tid; k < n_chunks; k += blockDim.x) {sc.chunk_flags[(size_t)s * sc.max_chunks + k] = 0u;}
    if (tid == 0u) {tail_ticket[s] = 0u;}
  }
}

// ------------------------------------------------------------------------------------------
// Compaction: per-unit records -> the scan's dense edge / surface clouds, one workgroup (row of workgroups) per scan.
// Step 1: per scan, ring totals and their exclusive prefix (rings ascending).
__global__ __launch_bounds__(kRings) void ring_totals_kernel(
  uint32_t * __restrict__ scan_info, const uint32_t * __restrict__ ring_count,
  const uint32_t * __restrict__ unit_ne, const uint32_t * __restrict__ unit_ns,
  uint32_t * __restrict__ ring_nedge, uint32_t * __restrict__ ring_nsurf, uint32_t * __restrict__ ring_ebase,
  uint32_t * __restrict__ ring_sbase, uint32_t n_units, uint32_t max_rings)
{
  const uint32_t slot = threadIdx.x, s = blockIdx.x;
  const uint32_t lane = slot & 63u, wave = slot >> 6;
  __shared__ uint32_t we[kRings / 64], wf[kRings / 64];
  {
    uint32_t e = 0, f = 0;
    if (slot < max_rings && ring_count[s * kRings + slot] != 0u) {
      const size_t ui = ((size_t)s * kRings + slot) * kUnitMaxBlocks;
      for (uint32_t j = 0; j < n_units; j++) {e += unit_ne[ui + j]; f += unit_ns[ui + j];}
    }
    // prefix over the 256 ring ids: along the lanes of each wave (DPP), then the four wave totals through LDS
    uint32_t ie = wave_inclusive_sum(e), jf = wave_inclusive_sum(f);
    if (lane == 63u) {we[wave] = ie; wf[wave] = jf;}
    __syncthreads();
    uint32_t tote = 0, totf = 0;
    for (uint32_t w = 0; w < kRings / 64; w++) {
      if (w < wave) {ie += we[w]; jf += wf[w];}
      tote += we[w];
      totf += wf[w];
    }
    ring_nedge[s * kRings + slot] = e;
    ring_nsurf[s * kRings + slot] = f;
    ring_ebase[s * kRings + slot] = ie - e;
    ring_sbase[s * kRings + slot] = jf - f;
    if (slot == kRings - 1) {
      scan_info[s * 4 + kInfoEdge] = tote;
      scan_info[s * 4 + kInfoSurface] = totf;
    }
  }
}

// Step 2: copy the feature records into the scan's edge / surface clouds: rings ascending, inside a
// ring angle ascending (units ascending; a slow-path ring is one unit).  One wave per ring; the ring's records
// are taken as ONE sequence over its units, four per lane in flight, so that the copy waits for memory once per
// 256 records instead of once per unit.
__global__ __launch_bounds__(256) void feature_compact_kernel(
  uint32_t n_units, uint32_t cap, const uint32_t * __restrict__ scan_begin,
  const uint32_t * __restrict__ ring_count, const uint32_t * __restrict__ ring_ebase,
  const uint32_t * __restrict__ ring_sbase, const uint32_t * __restrict__ unit_ne,
  const uint32_t * __restrict__ unit_ns, const uint32_t * __restrict__ unit_span,
  const float4 * __restrict__ rec_pts, const uint32_t * __restrict__ rec_idx, float4 * __restrict__ edge_pts,
  uint32_t * __restrict__ edge_idx, float4 * __restrict__ surf_pts, uint32_t * __restrict__ surf_idx,
  uint32_t max_rings, uint32_t * __restrict__ scan_info /* the scan's flags and (where ring_ebase == nullptr) totals are written here */,
  const uint32_t * __restrict__ counters, uint32_t * __restrict__ report /* pinned host memory [1 + kCounters + 1], or nullptr */,
  uint32_t serial /* of this batch, never 0 */, const uint32_t * __restrict__ ring_nedge, const uint32_t * __restrict__ ring_nsurf,
  const float4 * __restrict__ rec32 /* the record slots of the unit kernels' units (UnitTables), or nullptr */,
  uint32_t slot_places /* per slot: rec_slot_places() of the context's unit kernels */,
  const uint32_t * __restrict__ scan_flags /* [batch]: what the batch's kernels said about each scan (errors, route) */,
  uint32_t * __restrict__ ring_count_w /* rows of empty scans are zeroed here (no route writes them) */,
  uint32_t batch, uint32_t fused_ran /* the report's kCntBatch and kCntFusedRan: the host's own */,
  uint32_t * __restrict__ next_counters, uint32_t * __restrict__ next_flags, uint32_t * __restrict__ next_nedge,
  uint32_t * __restrict__ next_nsurf, uint32_t next_scans /* the OTHER set of accumulators (lfx_kernels_common.hpp kParityCounters): zeroed for the next batch */)
{
  const uint32_t lane = threadIdx.x & 63, s = blockIdx.y;
  // the batch's last kernel also hands what the batch reports about its stream to the host (the next batches' route is
  // chosen from it, lfx_api.hip choose_route): a dozen words written straight into pinned memory -- as a copy of its own
  // it would put another engine's work between this batch's kernels and the next one's
  // (nobody waits for it and no event says it has arrived: the block carries the batch's serial number before and after
  // the counters, written in this order with system-scope fences between, and the host reads it the other way round)
  if (report && blockIdx.x == 0 && blockIdx.y == 0) {
    if (threadIdx.x == 0) {
      __hip_atomic_store(report, serial, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __threadfence_system();
    }
    __syncthreads();
    if (threadIdx.x < kCounters) {
      const uint32_t v = threadIdx.x == kCntBatch ? batch : (threadIdx.x == kCntFusedRan ? fused_ran : counters[threadIdx.x]);
      __hip_atomic_store(report + 1 + threadIdx.x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __threadfence_system();
    }
    __syncthreads();
    // (a release at system scope: the closing serial must not be seen before the counters, whoever stored them)
    if (threadIdx.x == 0) {
      __threadfence_system();
      __hip_atomic_store(report + 1 + kCounters, serial, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  const uint32_t slot = blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // the next batch's accumulators (nobody is using that set now): every wave its ring's totals of the scans s, s + gridDim.y, ...
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < kParityCounters) {next_counters[threadIdx.x] = 0u;}
  if (lane == 0 && slot < (uint32_t)kRings) {
    for (uint32_t s2 = s; s2 < next_scans; s2 += gridDim.y) {
      next_nedge[s2 * kRings + slot] = 0u;
      next_nsurf[s2 * kRings + slot] = 0u;
      if (slot == 0u) {next_flags[s2] = 0u;}
    }
  }
  if (slot >= max_rings) {return;}
  {
    // Everything a wave needs before it can ask for its records is asked for at ONCE -- the scan's first point and route, the
    // ring's length, the totals of the rings before it, its units' counts and spans: a wave's life was five memory round
    // trips in series (15 us; the kernel ran at 3.7 TB/s of its own bytes, round 5), three of which only decided which of
    // the others to make.  (The totals of the earlier rings are read from the organised route's table whether or not the
    // scan took that route: it is a valid address either way, and the other route's sums follow below.)
    const size_t ui = ((size_t)s * kRings + slot) * kUnitMaxBlocks;
    const size_t b = scan_begin[s];
    const uint32_t n_scan = scan_begin[s + 1] - (uint32_t)b;
    const uint32_t err = scan_flags[s];
    uint32_t n_ring = ring_count[s * kRings + slot];
    uint32_t ne_k = 0, ns_k = 0, span_k = 0;               // lane j holds unit j's entries (n_units <= 64)
    if (lane < n_units) {
      ne_k = unit_ne[ui + lane];
      ns_k = unit_ns[ui + lane];
      span_k = unit_span[ui + lane];
    }
    uint32_t e_before = 0, f_before = 0, e_base = 0, f_base = 0;
    if (ring_ebase) {
      e_base = ring_ebase[s * kRings + slot];
      f_base = ring_sbase[s * kRings + slot];
    } else {
      for (uint32_t r = lane; r < slot; r += 64) {
        e_before += ring_nedge[s * kRings + r];
        f_before += ring_nsurf[s * kRings + r];
      }
    }
    asm volatile ("" : "+v"(ne_k), "+v"(ns_k), "+v"(span_k), "+v"(e_before), "+v"(f_before), "+v"(e_base), "+v"(f_base));    // (every load out before the first branch on one of them)
    // The scan's flag word becomes part of its results here (the batch's kernels OR into a word of their own set, which the
    // NEXT batch's compaction zeroes again; scan_info itself is only ever written).  An empty scan has been through no kernel
    // that writes its rows: no rings, no features -- and what the tables still hold of an earlier batch is not its.
    // (a holes scan is published without kScanFused: lfx_kernels_common.hpp scan_took_holes)
    const uint32_t err_pub = (err & kScanHoles) != 0u ? (err & ~(uint32_t)kScanFused) : err;
    if (n_scan == 0u) {
      if (lane == 0) {
        ring_count_w[s * kRings + slot] = 0u;
        if (slot == 0u) {
          scan_info[s * 4 + kInfoRings] = 0u; scan_info[s * 4 + kInfoError] = err_pub;
          scan_info[s * 4 + kInfoEdge] = 0u; scan_info[s * 4 + kInfoSurface] = 0u;
        }
      }
      return;
    }
    if (slot == 0u && lane == 0) {scan_info[s * 4 + kInfoError] = err_pub;}
    if (slot == max_rings - 1u && (err & (kScanHoles | kScanFellBack)) == kScanHoles) {
      // a grid with holes read in place: its rings are those that kept a point (the other routes write the number themselves)
      uint32_t occupied = 0;
      for (uint32_t r = lane; r < max_rings; r += 64) {occupied += ring_count[s * kRings + r] != 0u ? 1u : 0u;}
      occupied = (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_sum(occupied), 63);
      if (lane == 0) {scan_info[s * 4 + kInfoRings] = occupied;}
    }
    // a unit the unit kernels labelled keeps its records in its slot (points, then indices, in rank order: edges then surfaces;
    // beyond the slot's places at their ranks in the old arrays; the top bit of its span says so); a ring the workgroup-per-ring kernel
    // took in rec_pts / rec_idx, edges from the front of its positions and surfaces from their back
    const bool by_ring = scan_is_organised(err);
    size_t eb, fb;
    if (ring_ebase) {
      eb = b + e_base;
      fb = b + f_base;
    } else {
      // No ring_totals_kernel ahead of this one.  A scan the organised-scan kernel took: its units added their counts to
      // their rings' totals (unit_body), lane r, r + 64, ... takes ring r < slot.  Any other scan (a scan that kernel gave up,
      // redone by the bucketing route; a small batch on the bucketing route): the same sums from the unit tables
      // themselves.  The last ring's wave also writes the scan's totals.
      uint32_t e = e_before, f = f_before;
      if (!by_ring) {
        e = 0; f = 0;
        for (uint32_t r = lane; r < slot; r += 64) {
          if (ring_count[s * kRings + r] != 0u) {
            const size_t ur = ((size_t)s * kRings + r) * kUnitMaxBlocks;
            for (uint32_t j = 0; j < n_units; j++) {e += unit_ne[ur + j]; f += unit_ns[ur + j];}
          }
        }
      }
      e = (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_sum(e), 63);
      f = (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_sum(f), 63);
      eb = b + e;
      fb = b + f;
      if (slot == max_rings - 1u) {
        // (this ring's own totals: the sum of its units' counts, already here)
        uint32_t oe = n_ring != 0u || by_ring ? ne_k : 0u, of = n_ring != 0u || by_ring ? ns_k : 0u;
        oe = (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_sum(oe), 63);
        of = (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_sum(of), 63);
        if (lane == 0) {
          scan_info[s * 4 + kInfoEdge] = e + oe;
          scan_info[s * 4 + kInfoSurface] = f + of;
        }
      }
    }
    if (n_ring == 0u) {return;}
    const size_t off = ring_base(s, slot, max_rings, cap);
    // Lane u holds unit u's counts: their prefixes along the lanes say where in the ring's sequence of records (units
    // ascending; inside a unit the edges, then the surfaces) each unit's begin.  A record finds its unit by counting the units
    // that end at or before it (a wave-uniform loop of one scalar read and one compare per unit) and takes that unit's six
    // numbers from its lane -- until round 6 every record ran the whole address arithmetic once per unit and kept the one
    // that fitted: 600 vector and 780 scalar instructions per wave, a third of what a 16-ring sensor's compaction took.
    const uint32_t tot_k = ne_k + ns_k;
    const uint32_t incl = wave_inclusive_sum(tot_k), e_incl = wave_inclusive_sum(ne_k), s_incl = wave_inclusive_sum(ns_k);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    const uint32_t excl = incl - tot_k, e_excl = e_incl - ne_k, s_excl = s_incl - ns_k;
    for (uint32_t t0 = 0; t0 < total; t0 += 256) {
      size_t src[4], dst[4];
      uint32_t idx_at[4];                                    // a slot's index plane lies behind its points: dwords from the record's point to its index
      bool edge[4], valid[4], in_slot[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const uint32_t t = t0 + 64 * i + lane;
        valid[i] = t < total;
        uint32_t u = 0;
        for (uint32_t v = 0; v + 1u < n_units; v++) {u += t >= (uint32_t)__builtin_amdgcn_readlane((int)incl, (int)v) ? 1u : 0u;}
        const uint32_t ne = (uint32_t)__shfl((int)ne_k, (int)u), ns = (uint32_t)__shfl((int)ns_k, (int)u), span = (uint32_t)__shfl((int)span_k, (int)u);
        const uint32_t cum = (uint32_t)__shfl((int)excl, (int)u), ecum = (uint32_t)__shfl((int)e_excl, (int)u), scum = (uint32_t)__shfl((int)s_excl, (int)u);
        const size_t first = off + (span & 0xFFFFu), last = off + ((span >> 16) & 0x7FFFu);
        const bool slots = (span & kUnitRecordsInSlot) != 0u;        // a unit of the unit kernels: the first slot_places records in its slot
        const uint32_t q = t - cum;                          // index inside unit u
        edge[i] = q < ne;
        in_slot[i] = slots && q < slot_places;
        idx_at[i] = 4u * (ne + ns < slot_places ? ne + ns : slot_places) - 3u * q;
        if (slots) {
          src[i] = in_slot[i] ? (((size_t)s * max_rings + slot) * n_units + u) * (slot_places * (kRecBytes / 4u)) + 4u * q : first + q;    // (slot: in dwords)
        } else {
          src[i] = edge[i] ? first + q : last - 1 - (q - ne);
        }
        dst[i] = edge[i] ? eb + ecum + q : fb + scum + (q - ne);
        if (!valid[i]) {src[i] = off; dst[i] = b; in_slot[i] = false;}
      }
      float4 rp[4];
      uint32_t ri[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        rp[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        ri[i] = 0;
        if (valid[i]) {
          if (in_slot[i]) {
            const uint32_t * w = reinterpret_cast<const uint32_t *>(rec32) + src[i];          // the point; its index 4 n - 3 q dwords on (n records in the slot)
            rp[i] = *reinterpret_cast<const float4 *>(w);
            ri[i] = w[idx_at[i]];
          } else {
            rp[i] = rec_pts[src[i]];
            ri[i] = rec_idx[src[i]];
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 4; i++) {
        if (valid[i]) {
          if (edge[i]) {
            edge_pts[dst[i]] = rp[i];
            edge_idx[dst[i]] = ri[i];
          } else {
            surf_pts[dst[i]] = rp[i];
            surf_idx[dst[i]] = ri[i];
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Download path: the per-point outputs of the scans of a batch from the ring-major layout.  sorted_index goes to a dense
// array, rings ascending; labels and curvature go straight to the CALLER's point order (label of input point k
// at [k]) -- the scatter by original index is done here rather than in a host loop after the copy.  The host
// zeroes d_label / d_curv first: points the zero filter dropped, or beyond a ring's capacity, stay Default / 0.
// One workgroup per (ring id, scan).
__global__ __launch_bounds__(256) void densify_kernel(
  uint32_t first, uint32_t p0, const uint32_t * __restrict__ scan_begin, uint32_t max_rings, uint32_t cap,
  const uint32_t * __restrict__ ring_count,
  const uint8_t * __restrict__ label_s, const double * __restrict__ curv_s, const uint32_t * __restrict__ sidx,
  uint8_t * __restrict__ d_label, double * __restrict__ d_curv, uint32_t * __restrict__ d_sidx,
  const uint32_t * __restrict__ scan_info, const uint32_t * __restrict__ xform)
{
  // scan first + blockIdx.y of the batch; its outputs from record scan_begin[s] - p0 of the three arrays
  const uint32_t s = first + blockIdx.y;
  const uint32_t at = scan_begin[s] - p0, n_points = scan_begin[s + 1] - scan_begin[s];
  d_label += at; d_curv += at; d_sidx += at;
  const uint32_t ring = blockIdx.x, tid = threadIdx.x;
  const bool org = scan_is_grid(scan_info[s * 4 + kInfoError]);          // position i of the ring is point i * rings + ring (holes form: sidx says)
  __shared__ uint32_t part[256];
  part[tid] = (tid < ring && tid < max_rings) ? ring_count[s * kRings + tid] : 0u;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if ((int)tid < d) {part[tid] += part[tid + d];}
    __syncthreads();
  }
  const uint32_t dense = part[0];
  uint32_t N = ring_count[s * kRings + ring];
  N = N < cap ? N : cap;                       // an over-long ring holds only `cap` stored points
  const size_t off = ring_base(s, ring, max_rings, cap);
  const uint32_t Nfull = ring_count[s * kRings + ring];
  for (uint32_t i = tid; i < Nfull; i += blockDim.x) {
    const bool stored = i < N;
    const uint32_t orig = stored ? (org ? ring_column(xform[s * kRings + ring], i, Nfull) * max_rings + ring : sidx[off + i]) : 0xFFFFFFFFu;
    d_sidx[dense + i] = orig;
    if (orig < n_points) {
      d_label[orig] = label_s[off + i];
      if (curv_s != nullptr) {d_curv[orig] = curv_s[off + i];}
    }
  }
}

// ------------------------------------------------------------------------------------------
// Synchronous host API: what the caller always gets back -- per scan the header (scan_info, ring counts, ring status)
// and the edge / surface clouds with their index lists -- written by the device STRAIGHT INTO PINNED HOST MEMORY, so
// that only the bytes that exist cross PCIe and the host waits once, for the stream, not for sizes first.
// Host block: headers [count][kResultHeaderBytes], then edge_pts, surf_pts (float4), edge_idx, surf_idx (u32), each
// addressed like the device arrays relative to the first scan's first point (p0).
constexpr uint32_t kResultHeaderBytes = 16 + kRings * 4 + kRings;       // scan_info[4], ring_count[256], ring_status[256]
__global__ __launch_bounds__(256) void result_pack_kernel(
  uint32_t first, uint32_t p0, const uint32_t * __restrict__ scan_begin, const uint32_t * __restrict__ scan_info,
  const uint32_t * __restrict__ ring_count, const uint8_t * __restrict__ ring_status,
  const float4 * __restrict__ edge_pts, const uint32_t * __restrict__ edge_idx, const float4 * __restrict__ surf_pts,
  const uint32_t * __restrict__ surf_idx, uint8_t * __restrict__ h_hdr, float4 * __restrict__ h_edge_pts,
  float4 * __restrict__ h_surf_pts, uint32_t * __restrict__ h_edge_idx, uint32_t * __restrict__ h_surf_idx)
{
  const uint32_t k = blockIdx.y, s = first + k, tid = threadIdx.x;
  if (blockIdx.x == 0) {
    uint32_t * hdr = reinterpret_cast<uint32_t *>(h_hdr + (size_t)k * kResultHeaderBytes);
    if (tid < 4) {hdr[tid] = scan_info[s * 4 + tid];}
    hdr[4 + tid] = ring_count[s * kRings + tid];
    reinterpret_cast<uint8_t *>(hdr + 4 + kRings)[tid] = ring_status[s * kRings + tid];
  }
  const uint32_t ne = scan_info[s * 4 + kInfoEdge], ns = scan_info[s * 4 + kInfoSurface];
  const size_t b = scan_begin[s], hb = b - p0;
  for (uint32_t i = blockIdx.x * blockDim.x + tid; i < ne + ns; i += gridDim.x * blockDim.x) {
    if (i < ne) {
      h_edge_pts[hb + i] = edge_pts[b + i];
      h_edge_idx[hb + i] = edge_idx[b + i];
    } else {
      h_surf_pts[hb + i - ne] = surf_pts[b + i - ne];
      h_surf_idx[hb + i - ne] = surf_idx[b + i - ne];
    }
  }
}

// ------------------------------------------------------------------------------------------
// Per-stage kernel: one ring handed over as sorted x, y (lfx_stage_ring).
__global__ __launch_bounds__(512) void ring_stage_kernel(
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
