// lfx_kernels_common.hpp -- what every kernel header shares (record layout, parameters, scan_info bits, ring-major
// addressing).  The hand-written HIP kernels (gfx950 / MI355X) of the per-scan lidar feature extraction path are in
// lfx_kernels_extract.hpp; wire formats, Downsample and the localizer in lfx_kernels_wire / _downsample / _localize.hpp.  Semantics follow /root/reference/extraction (file:line cited per routine);
// the structure does not: the reference walks rings and blocks sequentially on one CPU thread
// and decides edge/surface points by argsort + greedy suppression.  Per batch of scans:
//
//   ring_unit_org_kernel    the path of an ORGANISED scan (a driver's column-major R x C grid, rings in angle order): a
//                           workgroup = one block of four adjacent rings, one ring per wave, reads the 32-byte records
//                           in place (four lanes share a 128-byte line) and does everything ring_unit_kernel does;
//                           ring_cut_kernel first finds the rings' rotation / reversal where a stream needs it
//   ring_scatter_kernel     any other scan: the ONE pass over the input, a stable counting sort of the points by ring
//                           (MakePointIndices, ring.hpp:114-125) into ring-major arrays, the prefix
//                           over earlier chunks obtained by look-back inside the launch
//   ring_unit_kernel        one WAVE per (ring, block) of a bucketed scan: angle-order check, range,
//                           curvature, links, block labelling, occlusion / out-of-range /
//                           parallel-beam masks, per-unit feature records; no workgroup barrier;
//                           point sets as ballot words in LDS bit arrays (one read + v_alignbit per
//                           32-position window); instantiated per span (3..6 chunks of 64) and, for the
//                           reference's default parameter set, with the thresholds as literals
//   ring_order_kernel       order repair: a ring that arrives as a rotation / reversal of its angle order
//                           is put in order by index arithmetic, anything else by an LDS bitonic sort;
//                           after the first unit pass (then a second pass takes the repaired rings) or,
//                           while a stream keeps arriving rotated, ahead of it over every ring
//   ring_extract_kernel     slow path for what neither pass takes (skip conditions, blocks that do not
//                           fit a wave, exactly tied directions): one workgroup per ring in LDS
//   ring_totals_kernel, feature_compact_kernel   per-unit records -> the scan's edge / surface clouds (small batches: the
//                           second alone)
//
// Labelling without a sort.  The reference's per-block pass (label.hpp:72-95,113-134) visits
// points in curvature order and lets every pick suppress what its link-aware +-P fill reaches
// (fill.hpp:101-117).  "j reaches i" is symmetric (|i-j| <= P, same block, every link between
// them intact), so the picked set is the lexicographically first maximal independent set of the
// candidates in priority order.  That set is computed exactly by rounds of "a live candidate
// with no live higher-priority candidate in reach is picked; everything a pick reaches dies":
// each round is a few AND/shift operations on 32-bit windows of point-set bit masks, and the
// priority comparisons (f64 curvature, index as tie-break) are done once per point.
//
// Floating point: every operation the reference performs in IEEE f64/f32 is performed here in the
// same type and order, unfused (contract off), so integer results (labels, index sets) are
// bit-exact and curvature is bit-equal.  Threshold tests on quotients are pre-classified with
// cheaper arithmetic and fall back to the exact division next to the threshold (unit_body).  The
// only libm call on the path, acos() in CalcRadian (math.cpp:34-46), is only ever compared with a
// threshold; the host turns that threshold into the equivalent bound on the cosine with the host's
// own acos (lfx_api.hip: cos_bound()).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace lfx
{

constexpr int kChunkPoints = 2048;     // points per workgroup in the ring bucketing kernels
constexpr int kChunkThreads = 256;
constexpr int kChunkSlots = kChunkPoints / kChunkThreads;
constexpr int kRings = 256;            // ring ids 0..255
constexpr uint32_t kSentinel = 0xFFFFFFFFu;

struct Layout { uint32_t step, ox, oy, oz, oring, rtype, be; };   // rtype: PointField datatype of ring; be: big-endian

// One f32 field / the ring field of a record as a PointCloud2 describes them.  Byte-wise access to
// fields that are not naturally aligned is left to the hardware (unaligned dword loads are legal).
__device__ inline float load_f32(const uint8_t * p, uint32_t be)
{
  uint32_t v = *reinterpret_cast<const uint32_t *>(p);
  if (be) {v = __builtin_bswap32(v);}
  return __uint_as_float(v);
}

__device__ inline uint32_t load_ring(const uint8_t * p, uint32_t rtype, uint32_t be)
{
  switch (rtype) {
    case 1: return (uint32_t)(int32_t)*reinterpret_cast<const int8_t *>(p);          // negative ids end up >= max_rings
    case 2: return *p;
    case 3: {uint16_t v = *reinterpret_cast<const uint16_t *>(p); if (be) {v = __builtin_bswap16(v);} return (uint32_t)(int32_t)(int16_t)v;}
    case 5:
    case 6: {uint32_t v = *reinterpret_cast<const uint32_t *>(p); if (be) {v = __builtin_bswap32(v);} return v;}
    default: {uint16_t v = *reinterpret_cast<const uint16_t *>(p); if (be) {v = __builtin_bswap16(v);} return v;}
  }
}

struct Params
{
  int P;                 // convolution_padding
  int B;                 // n_blocks
  double cos_bound;      // IsNeighborXY(i,i+1)  <=>  cos_bound <= cos_angle <= 1   (neighbor.hpp:44-48)
  float cos_bound_f;     // (float)cos_bound, +-inf kept: the f32 pre-filter of the same test
  double dist_diff;      // distance_diff_threshold
  double pb_ratio;       // parallel_beam_min_range_ratio
  float pb_ratio_f;      // (float)pb_ratio for the f32 pre-filter
  double edge_thr, surf_thr;
  double min_range, max_range;
};

// scan_info[s][4]
enum { kInfoRings = 0, kInfoError = 1, kInfoEdge = 2, kInfoSurface = 3 };
// bits of scan_info[s][kInfoError]: errors, and the route the scan took.  kScanFused: the organised-scan kernel took
// the scan (ring r's position k IS input point k * rings + r: nothing was staged, sxy / sz / sidx hold nothing for
// it); kScanFellBack: that kernel (or the host) handed the scan to the bucketing route, whose staged arrays are valid.
// kScanHoles (beside kScanFused): the scan is a grid whose invalid returns are (0, 0, 0) records, taken in place by the holes
// form of the organised-scan kernel: position k of ring r is the ring's k-th VALID column, its original index is in sidx
// (nothing else was staged: x, y, z come from the record).  kScanZeroFell / kScanCountFell: bookkeeping of the kernels.
enum : uint32_t { kErrRingId = 1u, kErrTimeout = 4u, kScanFused = 0x100u, kScanFellBack = 0x200u, kScanOrderFell = 0x400u,
                  kScanHoles = 0x800u, kScanZeroFell = 0x1000u, kScanCountFell = 0x2000u };
__host__ __device__ inline bool scan_is_organised(uint32_t err) {return (err & (kScanFused | kScanFellBack)) == kScanFused;}
// The PUBLISHED word of a holes scan (scan_info, feature_compact_kernel) carries kScanHoles WITHOUT kScanFused: to a caller that
// tests (bits & 0x300) == 0x100 (include/lfx.h, LFX_SCAN_ORGANISED) such a scan is one whose sorted_index is valid -- it is.
__host__ __device__ inline bool scan_took_holes(uint32_t err) {return (err & (kScanHoles | kScanFellBack)) == kScanHoles;}
// ... and its positions are its columns (no index array): the plain organised form
__host__ __device__ inline bool scan_is_grid(uint32_t err) {return (err & (kScanFused | kScanFellBack | kScanHoles)) == kScanFused;}
// counters[kCounters] (one block per batch parity, see kParityCounters): rings deferred by the first unit pass, repaired after it, sent to the workgroup-per-ring
// kernel, repaired before it; scans on the fall-back list; whether the organised-scan kernel ran; scans in the batch
// ... scans the organised-scan kernel gave up on because a ring was not in angle order (the rest of the pattern held);
// rings ring_cut_kernel found rotated / reversed; whether it ran
// ... scans the plain organised-scan kernel gave up for a (0, 0, 0) record alone (zero filter on); whether the holes form
// ran; (holes form) ring groups that held such a record
enum { kCntDefer = 0, kCntRedo = 1, kCntSlow = 2, kCntPreFixed = 3, kCntFallback = 4, kCntFusedRan = 5, kCntBatch = 6,
       kCntOrderFell = 7, kCntTurned = 8, kCntCutRan = 9, kCntZeroFell = 10, kCntHolesRan = 11, kCntZeroGroups = 12, kCounters = 14 };
// What a batch accumulates into with atomics -- the counters, the per-scan flag words, the organised route's ring totals --
// exists TWICE: batch k uses set k mod 2, and its last kernel (feature_compact_kernel) leaves the other set zeroed for batch
// k + 1.  No reset launch stands in front of a batch then: on the organised route a step is the unit kernel, the fall-back
// tail and the compaction (round 6; the reset kernel was 5 us of every step, 5 % of a 128 x 2048 x 32 one).
constexpr uint32_t kParityCounters = 16;                  // words per set of counters
// Ring transform of an organised scan (ring_cut_kernel): position k of the ring is column (start + k) mod C, or
// (start - k) mod C for a clockwise sensor; 0 = the ring arrives in angle order.
constexpr uint32_t kXformReversed = 0x80000000u;
__host__ __device__ inline uint32_t ring_column(uint32_t xf, uint32_t k, uint32_t C)
{
  const uint32_t start = xf & ~kXformReversed;
  if (xf & kXformReversed) {return k <= start ? start - k : start + C - k;}
  return start + k < C ? start + k : start + k - C;
}

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
// Ring-major layout: ring `r` of scan `s` owns positions [((s * max_rings) + r) * cap, + cap) of every
// per-point array (fixed capacity per ring id, so no global prefix over rings is needed).
__host__ __device__ inline size_t ring_base(uint32_t s, uint32_t ring, uint32_t max_rings, uint32_t cap)
{
  return ((size_t)s * max_rings + ring) * cap;
}

// inclusive prefix sum along the 64 lanes of a wave with DPP moves (row shifts inside the rows of 16 lanes, then the two
// row broadcasts): eight VALU instructions where six ds_bpermute round trips through the LDS crossbar were ~700 cycles
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t x)
{
  uint32_t t = x;
  t += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, true);      // row_shr:1
  t += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, true);      // row_shr:2
  t += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x113, 0xF, 0xF, true);      // row_shr:3
  t += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x114, 0xF, 0xE, true);      // row_shr:4, banks 1-3
  t += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x118, 0xF, 0xC, true);      // row_shr:8, banks 2-3
  t += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x142, 0xA, 0xF, true);      // row_bcast:15 into rows 1 and 3
  t += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x143, 0xC, 0xF, true);      // row_bcast:31 into rows 2 and 3
  return t;
}


// ------------------------------------------------------------------------------------------
// Shared by the unit kernels (lfx_kernels_unit.hpp) and the kernels around them (lfx_kernels_extract.hpp)
constexpr int kUnitWaves = 4;
constexpr int kUnitMaxBlocks = 64;        // per-unit count tables are sized for n_blocks <= 64
// Per scan, from the host with the scan's first index (the organised-scan kernel): [0] columns per ring, 0 if the scan is
// not max_rings x columns with the columns within the ring capacity; [1 + j] boundary j of its rings' blocks, j = 0 .. B
constexpr int kGeomStride = kUnitMaxBlocks + 2;
enum { kDeferOrder = 1u, kDeferOther = 2u, kDeferMask = 3u, kRingSorted = 4u /* put in order by ring_order_kernel */ };

// The holes form (grid_count_kernel + ring_unit_org_kernel<.., HOLES>): valid returns are counted per ring and PIECE of 16
// columns; cum16[scan][ring][p] = valid returns of the ring in pieces 0 .. p - 1 (entry ceil(C / 16) = the ring's length).
// Per (ring, block) unit grid_count_kernel also leaves a DESCRIPTOR -- the ring's length, the block's two boundaries
// (index_range.cpp:60-66), the first and the last piece that hold a position the unit needs -- so that a unit's head is four
// scalar loads (its workgroup's four rings) and no search: {N | b0 << 16, b1 | flags << 16, first piece | last piece << 16, 0}.
constexpr uint32_t kHoleUnitDead = 1u;                // flags: the ring has no unit here (no valid return at all)
constexpr int kPieceCols = 16;
__host__ __device__ inline uint32_t cum_stride(uint32_t ring_cap) {return ring_cap / kPieceCols + 4u;}      // entries per row
constexpr int holes_loads(int chunks) {return chunks + 2;}       // pieces of 16 columns x 4 rings a wave loads: 64 (CH + 2) columns per unit

constexpr int kUnitMaxChunks = 12;        // the long form: blocks of up to 768 positions (rings of up to ~4 500 points in 6 blocks)

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


// Boundary j of the padded block range: index_range.cpp:60-66 with start=P, end=N-P.
__host__ __device__ inline int block_boundary(int N, int P, int B, int j)
{
  const double s = (double)P, e = (double)(N - P), n = (double)B;
  return (int)(s * (1. - j / n) + e * j / n);
}

// Output tables of the unit kernel, read through one pointer: sixteen kernel-argument pointers held in
// scalar registers from the first instruction on crowd out the wave-uniform masks the kernel works
// with (the scalar file is the scarce one here).  The entries are fetched where they are first needed.
struct UnitTables
{
  uint8_t * label_s;
  double * curv_s;
  float4 * rec_pts;
  uint32_t * rec_idx;
  uint8_t * ring_status;
  uint32_t * unit_ne, * unit_ns, * unit_span;
  uint32_t * ring_flags;
  uint32_t * scan_info, * fb_count, * fb_list;      // organised-scan kernel: a scan it cannot take goes on the fall-back list
  uint32_t * scan_flags;                            // [batch]: the batch's error / route bits while it runs (feature_compact_kernel moves them into scan_info)
  uint32_t * sidx;                                  // holes form of the organised-scan kernel: original index per ring position
  const uint16_t * cum16;                           // ... grid_count_kernel's prefix table, rows of cum_stride(cap) entries, [batch][max_rings]
  const uint4 * hole_desc;                          // ... and its unit descriptors, [batch][max_rings][n_blocks]
  uint32_t * ring_nedge, * ring_nsurf;              // organised-scan kernel: every unit adds its counts to its ring's (feature_compact_kernel)
  float4 * rec32;                                   // the unit kernels' record slots, [batch][max_rings][n_blocks] x rec_slot_places() x kRecBytes
  Params prm;                                       // the thresholds, for the kernels that do not have them as literals (read where a stage needs them)
};
// A slot of a (ring, block) unit the unit kernels labelled (both routes): up to rec_slot_places() records {x, y, z, (float)c} and, right behind the n that are there, their n original indices (one run of 20 n bytes);
// edges then surfaces, each in position order.  What does not fit lies at its rank in rec_pts / rec_idx from the unit's
// first owned position.
// Places per slot: 64 (the default parameters leave ~33 features in a unit of ~300 positions), 128 where the padding is
// 1 or 2 -- a pick silences P positions either side, so the launch file's P = 2 leaves ~85 -- and the unit's LDS can stage them
// (5 chunks or more); one number for the unit kernels of a context and its compaction (lfx_api.hip).
__host__ __device__ constexpr uint32_t rec_slot_places(int padding_compiled_for, int chunks)
{
  return padding_compiled_for > 0 && padding_compiled_for <= 2 && chunks >= 5 ? 128u : 64u;
}
// A scan the organised-scan kernels cannot take (ring pattern, point count, angle order, a skip condition, a unit that
// does not fit a wave) is flagged once and appended to the fall-back list: the bucketing route then redoes it whole.
// why: kScanOrderFell / kScanZeroFell are counted (once per scan) for the host's choice of route -- a stream whose rings are
// rotated / reversed gets ring_cut_kernel, a grid with (0, 0, 0) records the holes form.
__device__ inline void scan_falls_back(const UnitTables * __restrict__ tab, uint32_t s, uint32_t why = 0u)
{
  const uint32_t bits = kScanFellBack | why;
  const uint32_t old = atomicOr(tab->scan_flags + s, bits);
  if ((old & kScanFellBack) == 0u) {tab->fb_list[atomicAdd(tab->fb_count, 1u)] = s;}
  if ((why & kScanOrderFell) != 0u && (old & kScanOrderFell) == 0u) {atomicAdd(tab->fb_count + (kCntOrderFell - kCntFallback), 1u);}
  if ((why & kScanZeroFell) != 0u && (old & kScanZeroFell) == 0u) {atomicAdd(tab->fb_count + (kCntZeroFell - kCntFallback), 1u);}
}

constexpr uint32_t kRecBytes = 20u;                       // per place: a 16-byte point and a 4-byte index
constexpr uint32_t kUnitRecordsInSlot = 0x80000000u;      // in unit_span: written by the unit kernels, not by the workgroup-per-ring kernel

}  // namespace lfx
