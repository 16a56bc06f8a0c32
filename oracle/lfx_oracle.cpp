// lfx_oracle.cpp -- CPU oracle (TEST INFRASTRUCTURE ONLY; see lfx_oracle.h).
//
// A restatement, on flat arrays, of the reference's per-scan extraction path
// (tier4/lidar_feature_extraction, extraction/).  It keeps the reference's evaluation
// strategy where that strategy is observable or costs time: neighbour tests are evaluated
// lazily and recomputed on every call (sqrt, sqrt, divide, acos), ranges are recomputed on
// every use, each block is argsorted twice with std::sort, block slices copy their index
// lists.  Compile with -ffp-contract=off: the reference is built for baseline x86-64
// (extraction/CMakeLists.txt:6-7, no -march), so no operation in it is fused.
//
// Citations are file:line relative to /root/reference/.

#include "lfx_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace
{

// The reference signals "skip this ring" with std::invalid_argument
// (feature_extraction.cpp:154-156).  The oracle attaches which site raised it.
struct RingSkip : public std::invalid_argument
{
  RingSkip(int c, const char * what)
  : std::invalid_argument(what), code(c) {}
  int code;
};

using Labels = std::vector<uint8_t>;

// ------------------------------------------------------------------ math.hpp / math.cpp
inline double XYNormD(double x, double y)   // math.hpp:36-39
{
  return std::sqrt(x * x + y * y);
}

double CalcRadianD(double x1, double y1, double x2, double y2)   // math.cpp:34-46
{
  const double dot = x1 * x2 + y1 * y2;
  const double n1 = XYNormD(x1, y1);
  const double n2 = XYNormD(x2, y2);
  if (n1 == 0 && n2 == 0) {
    throw RingSkip(ORC_RING_ZERO_NORM_PAIR, "All input values are zero. Angle cannot be calculated");
  }
  return std::acos(dot / (n1 * n2));
}

double InnerProductD(const double * a, const double * a_end, const double * b)   // math.hpp:43-53
{
  double sum = 0.;
  for (; a != a_end; ++a, ++b) {
    sum += (*a) * (*b);
  }
  return sum;
}

// ------------------------------------------------------------------ convolution / curvature
std::vector<double> Conv1D(const std::vector<double> & input, const std::vector<double> & weight)
{
  // convolution.cpp:35-66
  if (input.size() < weight.size()) {
    throw RingSkip(ORC_RING_TOO_FEW_CONV, "Input array size cannot be smaller than weight size");
  }
  const int pad = (static_cast<int>(weight.size()) - 1) / 2;
  const int n_valid = static_cast<int>(input.size()) - 2 * pad;
  std::vector<double> out(input.size());
  for (int i = 0; i < pad; i++) {out[i] = 0.;}
  for (int i = 0; i < n_valid; i++) {
    const double * w0 = input.data() + i;
    out[pad + i] = InnerProductD(w0, w0 + weight.size(), weight.data());
  }
  for (int i = 0; i < pad; i++) {out[n_valid + pad + i] = 0.;}
  return out;
}

std::vector<double> Weight(int padding)   // curvature.cpp:36-42
{
  std::vector<double> w(padding * 2 + 1, 1.);
  w.at(padding) = -2. * padding;
  return w;
}

std::vector<double> Curvature(const std::vector<double> & range, int padding)   // curvature.cpp:44-50
{
  std::vector<double> c = Conv1D(range, Weight(padding));
  for (double & v : c) {v = v * v;}
  return c;
}

// ------------------------------------------------------------------ algorithm.hpp:44-71
std::vector<int> ArgsortD(const double * v, int n, bool canonical, int64_t * ties)
{
  std::vector<int> idx(n);
  for (int i = 0; i < n; i++) {idx[i] = i;}
  if (canonical) {
    std::sort(idx.begin(), idx.end(), [v](int l, int r) {return v[l] < v[r] || (v[l] == v[r] && l < r);});
  } else {
    std::sort(idx.begin(), idx.end(), [v](int l, int r) {return v[l] < v[r];});
  }
  if (ties) {
    for (int i = 0; i + 1 < n; i++) {
      if (v[idx[i]] == v[idx[i + 1]]) {(*ties)++;}
    }
  }
  return idx;
}

// ------------------------------------------------------------------ index_range.cpp:32-79
struct BlockRange
{
  BlockRange(int start, int end, int n_blocks)
  : start_(start), end_(end), n_(n_blocks)
  {
    if (end - start < n_blocks) {
      throw RingSkip(ORC_RING_TOO_FEW_BLOCKS, "end_index - start_index cannot be smaller than n_blocks");
    }
  }
  int Boundary(int j) const   // index_range.cpp:60-66
  {
    const double s = static_cast<double>(start_);
    const double e = static_cast<double>(end_);
    const double n = static_cast<double>(n_);
    return static_cast<int>(s * (1. - j / n) + e * j / n);
  }
  int start_, end_, n_;
};

// ------------------------------------------------------------------ ring.hpp:54-99
// F is the arithmetic type of the point fields: the reference's template evaluates the
// products, the sum and the determinant in the field type (float for PointXYZIR, double
// for the struct used by test_ring.cpp) before widening to double.
template<typename F>
bool PolarLess(F ax, F ay, F bx, F by)
{
  if (ax == bx && ay == by) {return false;}
  const double lena = ax * ax + ay * ay;
  const double lenb = bx * bx + by * by;
  if (lena == 0) {
    if (by == 0) {return bx < 0;}
    return by > 0;
  }
  if (lenb == 0) {return ay < 0;}
  if (ay == 0) {return (ax >= 0) && (by >= 0);}
  if (by == 0) {return !((bx >= 0) && (ay >= 0));}
  if (ay * by > 0) {
    const double det = ax * by - ay * bx;
    return det > 0;
  }
  return ay < 0;
}

// ------------------------------------------------------------------ point access
// x and y of point k live at bx + k*stride and by + k*stride (AoS or two flat arrays).
struct XYSource
{
  const uint8_t * bx;
  const uint8_t * by;
  size_t stride;
  size_t n;
  float X(size_t k) const {float v; std::memcpy(&v, bx + k * stride, 4); return v;}
  float Y(size_t k) const {float v; std::memcpy(&v, by + k * stride, 4); return v;}
};

// mapped_points.hpp:40-72 -- a view cloud[indices[i]]; Slice copies the index list.
class Mapped
{
public:
  Mapped(const XYSource & src, std::vector<int> indices)
  : src_(src), indices_(std::move(indices)) {}
  int size() const {return static_cast<int>(indices_.size());}
  void at(int i, float & x, float & y) const
  {
    const size_t k = static_cast<size_t>(indices_.at(i));   // vector::at -> std::out_of_range
    if (k >= src_.n) {throw std::out_of_range("cloud index");}
    x = src_.X(k);
    y = src_.Y(k);
  }
  Mapped Slice(int begin, int end) const
  {
    return Mapped(src_, std::vector<int>(indices_.begin() + begin, indices_.begin() + end));
  }

private:
  XYSource src_;
  std::vector<int> indices_;
};

// ------------------------------------------------------------------ neighbor.hpp:50-136
class NeighborBase
{
public:
  virtual ~NeighborBase() {}
  virtual bool operator()(int i, int j) const = 0;
  virtual int size() const = 0;
};

class NeighborXY : public NeighborBase   // neighbor.hpp:64-114
{
public:
  NeighborXY(const Mapped & pts, double radian_threshold)
  : pts_(pts), thr_(radian_threshold)
  {
    if (pts.size() < 2) {
      throw RingSkip(ORC_RING_BLOCK_TOO_SMALL, "The input point size cannot be smaller than 2");
    }
  }
  bool operator()(int i, int j) const override
  {
    if (i < 0 || i >= size() || j < 0 || j >= size()) {throw std::out_of_range("neighbor index");}
    float x1, y1, x2, y2;
    pts_.at(i, x1, y1);
    pts_.at(j, x2, y2);
    return CalcRadianD(x1, y1, x2, y2) < thr_;   // neighbor.hpp:44-48
  }
  int size() const override {return pts_.size();}
  NeighborXY Slice(int begin, int end) const {return NeighborXY(pts_.Slice(begin, end), thr_);}

private:
  Mapped pts_;
  double thr_;
};

class NeighborDebug : public NeighborBase   // neighbor.hpp:116-136
{
public:
  NeighborDebug(const int * groups, int n)
  : g_(groups, groups + n) {}
  bool operator()(int i, int j) const override {return g_.at(i) == g_.at(j);}
  int size() const override {return static_cast<int>(g_.size());}

private:
  std::vector<int> g_;
};

// range.hpp:45-74 -- recomputed on every call, like the reference.
class RangeOf
{
public:
  explicit RangeOf(const Mapped & pts)
  : pts_(pts) {}
  double operator()(int i) const
  {
    float x, y;
    pts_.at(i, x, y);
    return XYNormD(x, y);
  }
  std::vector<double> All(int begin, int end) const   // range.hpp:58-65 (ignores begin, as the reference does)
  {
    std::vector<double> r(end - begin);
    for (unsigned int i = 0; i < r.size(); i++) {r.at(i) = (*this)(i);}
    return r;
  }
  int size() const {return pts_.size();}

private:
  Mapped pts_;
};

// ------------------------------------------------------------------ fill.hpp:40-117
// `labels` points at the first label of the (sliced) view, n = its length.
void FillLeft(uint8_t * labels, int n, const NeighborBase & nb, int begin, int end, uint8_t label)
{
  if (end > n) {throw RingSkip(ORC_RING_OTHER, "end_index > labels.size()");}
  if (begin < 0) {throw RingSkip(ORC_RING_OTHER, "begin_index < 0");}
  for (int i = begin; i < end - 1; i++) {
    labels[i] = label;
    if (!nb(i, i + 1)) {return;}
  }
  if (end - 1 < 0 || end - 1 >= n) {throw std::out_of_range("labels.at");}
  labels[end - 1] = label;
}

void FillRight(uint8_t * labels, int n, const NeighborBase & nb, int begin, int end, uint8_t label)
{
  if (end >= n) {throw RingSkip(ORC_RING_OTHER, "end_index >= labels.size()");}
  if (begin < -1) {throw RingSkip(ORC_RING_OTHER, "begin_index < -1");}
  for (int i = end; i > begin + 1; i--) {
    labels[i] = label;
    if (!nb(i, i - 1)) {return;}
  }
  if (begin + 1 < 0 || begin + 1 >= n) {throw std::out_of_range("labels.at");}
  labels[begin + 1] = label;
}

void FillAround(uint8_t * labels, int n, const NeighborBase & nb, int index, int padding, uint8_t label)
{
  const int lo = std::max(-1, index - padding - 1);
  const int hi = std::min(index + 1 + padding, n);
  FillRight(labels, n, nb, lo, index, label);
  FillLeft(labels, n, nb, index, hi, label);
}

// ------------------------------------------------------------------ label.hpp:61-139
void EdgeAssign(
  uint8_t * labels, const double * curv, int n, const NeighborBase & nb, int padding, double thr,
  bool canonical, int64_t * ties)
{
  const std::vector<int> order = ArgsortD(curv, n, canonical, ties);
  for (auto it = order.rbegin(); it != order.rend(); ++it) {
    const int i = *it;
    if (!(labels[i] == ORC_LABEL_DEFAULT && curv[i] >= thr)) {continue;}
    FillAround(labels, n, nb, i, padding, ORC_LABEL_EDGE_NEIGHBOR);
    labels[i] = ORC_LABEL_EDGE;
  }
}

void SurfaceAssign(
  uint8_t * labels, const double * curv, int n, const NeighborBase & nb, int padding, double thr,
  bool canonical)
{
  const std::vector<int> order = ArgsortD(curv, n, canonical, nullptr);
  for (const int i : order) {
    if (!(labels[i] == ORC_LABEL_DEFAULT && curv[i] <= thr)) {continue;}
    FillAround(labels, n, nb, i, padding, ORC_LABEL_SURFACE_NEIGHBOR);
    labels[i] = ORC_LABEL_SURFACE;
  }
}

// label.hpp:141-164
void AssignBlocks(
  Labels & labels, const std::vector<double> & curv, const NeighborXY & nb, const BlockRange & blocks,
  int padding, double edge_thr, double surf_thr, bool canonical, int64_t * ties)
{
  for (int j = 0; j < blocks.n_; j++) {
    const int b = blocks.Boundary(j);
    const int e = blocks.Boundary(j + 1);
    const NeighborXY sliced = nb.Slice(b, e);
    EdgeAssign(labels.data() + b, curv.data() + b, e - b, sliced, padding, edge_thr, canonical, ties);
    SurfaceAssign(labels.data() + b, curv.data() + b, e - b, sliced, padding, surf_thr, canonical);
  }
}

// ------------------------------------------------------------------ occlusion.hpp:37-91
void OccludedFromLeft(Labels & labels, const NeighborBase & nb, const RangeOf & range, unsigned int padding, double d)
{
  const int n = static_cast<int>(labels.size());
  for (unsigned int i = 0; i < labels.size() - padding - 1; i++) {
    if (!nb(i, i + 1)) {continue;}
    const double r0 = range(i);
    const double r1 = range(i + 1);
    if (r1 > r0 + d) {
      FillLeft(labels.data(), n, nb, i + 1, i + padding + 2, ORC_LABEL_OCCLUDED);
    }
  }
}

void OccludedFromRight(Labels & labels, const NeighborBase & nb, const RangeOf & range, unsigned int padding, double d)
{
  const int n = static_cast<int>(labels.size());
  for (unsigned int i = labels.size() - 1; i >= padding + 1; i--) {
    if (!nb(i, i - 1)) {continue;}
    const double r1 = range(i - 1);
    const double r0 = range(i);
    if (r1 > r0 + d) {
      FillRight(labels.data(), n, nb, static_cast<int>(i - padding - 2), i - 1, ORC_LABEL_OCCLUDED);
    }
  }
}

// out_of_range.hpp:36-48
void OutOfRange(Labels & labels, const RangeOf & range, double lo, double hi)
{
  for (int i = 0; i < range.size(); i++) {
    const double v = range(i);
    if (!(lo <= v && v <= hi)) {labels.at(i) = ORC_LABEL_OUT_OF_RANGE;}   // range.hpp:40-43
  }
}

// parallel_beam.hpp:36-51 -- the ratios are narrowed to float before the compare.
void ParallelBeam(Labels & labels, const RangeOf & range, double thr)
{
  const std::vector<double> r = range.All(0, static_cast<int>(labels.size()));
  for (unsigned int i = 1; i + 1 < labels.size(); i++) {
    const float ratio1 = std::abs(r.at(i - 1) - r.at(i)) / r.at(i);
    const float ratio2 = std::abs(r.at(i + 1) - r.at(i)) / r.at(i);
    if (ratio1 > thr && ratio2 > thr) {labels.at(i) = ORC_LABEL_PARALLEL_BEAM;}
  }
}

// helpers for the stage-level C entry points -----------------------------------------
struct StageNeighbor
{
  StageNeighbor(const int * groups, const float * x, const float * y, int n, double thr)
  {
    if (groups) {
      nb.reset(new NeighborDebug(groups, n));
    } else {
      XYSource s{reinterpret_cast<const uint8_t *>(x), reinterpret_cast<const uint8_t *>(y), 4, static_cast<size_t>(n)};
      std::vector<int> idx(n);
      for (int i = 0; i < n; i++) {idx[i] = i;}
      nb.reset(new NeighborXY(Mapped(s, idx), thr));
    }
  }
  std::unique_ptr<NeighborBase> nb;
};

template<typename Fn>
int Guard(Fn && fn)
{
  try {
    fn();
  } catch (const std::invalid_argument &) {
    return 1;
  } catch (const std::out_of_range &) {
    return 2;
  }
  return 0;
}

Mapped MapAll(const float * x, const float * y, int n)
{
  XYSource s{reinterpret_cast<const uint8_t *>(x), reinterpret_cast<const uint8_t *>(y), 4, static_cast<size_t>(n)};
  std::vector<int> idx(n);
  for (int i = 0; i < n; i++) {idx[i] = i;}
  return Mapped(s, idx);
}

}  // namespace

// ======================================================================= C entry points
extern "C" {

double orc_xy_norm(double x, double y) {return XYNormD(x, y);}

int orc_calc_radian(double x1, double y1, double x2, double y2, double * out)
{
  return Guard([&] {*out = CalcRadianD(x1, y1, x2, y2);});
}

double orc_inner_product(const double * a, const double * b, int n) {return InnerProductD(a, a + n, b);}

int orc_convolution1d(const double * input, int n, const double * weight, int m, double * out)
{
  return Guard(
    [&] {
      const auto r = Conv1D(std::vector<double>(input, input + n), std::vector<double>(weight, weight + m));
      std::copy(r.begin(), r.end(), out);
    });
}

void orc_make_weight(int padding, double * out)
{
  const auto w = Weight(padding);
  std::copy(w.begin(), w.end(), out);
}

int orc_calc_curvature(const double * range, int n, int padding, double * out)
{
  return Guard(
    [&] {
      const auto c = Curvature(std::vector<double>(range, range + n), padding);
      std::copy(c.begin(), c.end(), out);
    });
}

void orc_argsort(const double * values, int n, int * out)
{
  const auto idx = ArgsortD(values, n, false, nullptr);
  std::copy(idx.begin(), idx.end(), out);
}

int orc_index_range(int start, int end, int n_blocks, int * bounds)
{
  return Guard(
    [&] {
      const BlockRange r(start, end, n_blocks);
      for (int j = 0; j <= n_blocks; j++) {bounds[j] = r.Boundary(j);}
    });
}

int orc_padded_index_range(int size, int n_blocks, int padding, int * bounds)
{
  return orc_index_range(padding, size - padding, n_blocks, bounds);   // index_range.hpp:59-66
}

int orc_polar_less_f64(double ax, double ay, double bx, double by) {return PolarLess<double>(ax, ay, bx, by);}
int orc_polar_less_f32(float ax, float ay, float bx, float by) {return PolarLess<float>(ax, ay, bx, by);}

void orc_sort_by_atan2_f64(const double * x, const double * y, int n, int * indices)
{
  std::sort(indices, indices + n, [&](int a, int b) {return PolarLess<double>(x[a], y[a], x[b], y[b]);});
}

int orc_is_neighbor_xy(float x1, float y1, float x2, float y2, double radian_threshold, int * out)
{
  return Guard([&] {*out = CalcRadianD(x1, y1, x2, y2) < radian_threshold;});
}

int orc_is_in_inclusive_range(double v, double lo, double hi) {return lo <= v && v <= hi;}

int orc_fill_from_left(
  uint8_t * labels, int n, const int * groups, const float * x, const float * y, double thr,
  int begin, int end, uint8_t label)
{
  return Guard([&] {StageNeighbor s(groups, x, y, n, thr); FillLeft(labels, n, *s.nb, begin, end, label);});
}

int orc_fill_from_right(
  uint8_t * labels, int n, const int * groups, const float * x, const float * y, double thr,
  int begin, int end, uint8_t label)
{
  return Guard([&] {StageNeighbor s(groups, x, y, n, thr); FillRight(labels, n, *s.nb, begin, end, label);});
}

int orc_fill_neighbors(
  uint8_t * labels, int n, const int * groups, const float * x, const float * y, double thr,
  int index, int padding, uint8_t label)
{
  return Guard([&] {StageNeighbor s(groups, x, y, n, thr); FillAround(labels, n, *s.nb, index, padding, label);});
}

int orc_edge_label_assign(
  uint8_t * labels, const double * curvature, int n, const int * groups, const float * x, const float * y,
  double thr, int padding, double threshold)
{
  return Guard(
    [&] {
      StageNeighbor s(groups, x, y, n, thr);
      EdgeAssign(labels, curvature, n, *s.nb, padding, threshold, false, nullptr);
    });
}

int orc_surface_label_assign(
  uint8_t * labels, const double * curvature, int n, const int * groups, const float * x, const float * y,
  double thr, int padding, double threshold)
{
  return Guard(
    [&] {
      StageNeighbor s(groups, x, y, n, thr);
      SurfaceAssign(labels, curvature, n, *s.nb, padding, threshold, false);
    });
}

int orc_assign_label(
  uint8_t * labels, const double * curvature, int n, const float * x, const float * y, double thr,
  int n_blocks, int padding, double edge_threshold, double surface_threshold)
{
  return Guard(
    [&] {
      Labels l(labels, labels + n);
      const std::vector<double> c(curvature, curvature + n);
      const NeighborXY nb(MapAll(x, y, n), thr);
      const BlockRange blocks(padding, n - padding, n_blocks);
      AssignBlocks(l, c, nb, blocks, padding, edge_threshold, surface_threshold, false, nullptr);
      std::copy(l.begin(), l.end(), labels);
    });
}

int orc_label_occluded(
  uint8_t * labels, int n, const float * x, const float * y, double thr, int padding, double d)
{
  return Guard(
    [&] {
      Labels l(labels, labels + n);
      const Mapped pts = MapAll(x, y, n);
      const NeighborXY nb(pts, thr);
      const RangeOf range(pts);
      OccludedFromLeft(l, nb, range, padding, d);
      OccludedFromRight(l, nb, range, padding, d);
      std::copy(l.begin(), l.end(), labels);
    });
}

void orc_label_out_of_range(uint8_t * labels, int n, const float * x, const float * y, double lo, double hi)
{
  Labels l(labels, labels + n);
  OutOfRange(l, RangeOf(MapAll(x, y, n)), lo, hi);
  std::copy(l.begin(), l.end(), labels);
}

void orc_label_parallel_beam(uint8_t * labels, int n, const float * x, const float * y, double thr)
{
  Labels l(labels, labels + n);
  ParallelBeam(l, RangeOf(MapAll(x, y, n)), thr);
  std::copy(l.begin(), l.end(), labels);
}

// range_message.hpp:37-83: "{} (which is {}) OP {} (which is {})"
int orc_range_message(int kind, const char * value_name, const char * range_name, long long value, long long range, char * buf, size_t len)
{
  static const char * const op[4] = {">=", "<=", ">", "<"};
  if (kind < 0 || kind > 3) {return -1;}
  return std::snprintf(buf, len, "%s (which is %lld) %s %s (which is %lld)", value_name, value, op[kind], range_name, range);
}

void orc_irange(int size, int * out)   // iterator.cpp:33-36
{
  for (int i = 0; i < size; i++) {out[i] = i;}
}

// mapped_points.hpp:40-72: a view cloud[indices[i]]; Slice(begin, end) copies indices[begin, end)
int orc_mapped_points_at(const double * cloud_y, int n_cloud, const int * indices, int n_indices, int begin, int end, int i, double * out, int * size)
{
  if (begin < 0 || end > n_indices || begin > end) {return 1;}
  const std::vector<int> sliced(indices + begin, indices + end);      // mapped_points.hpp:63-67
  *size = static_cast<int>(sliced.size());                            // :69-71
  if (i < 0 || i >= *size) {return 1;}
  const int k = sliced[i];
  if (k < 0 || k >= n_cloud) {return 1;}
  *out = cloud_y[k];                                                  // :58-61
  return 0;
}

// what() of the exception that makes the node abandon a ring (feature_extraction.cpp:154-156)
int orc_ring_message(int status, int n, const orc_params * p, char * buf, size_t len)
{
  const int P = p->padding, B = p->n_blocks;
  switch (status) {
    case ORC_RING_TOO_FEW_CONV:      // convolution.cpp:40-41
      return std::snprintf(buf, len, "Input array size %d cannot be smaller than weight size %d", n, 2 * P + 1);
    case ORC_RING_TOO_FEW_BLOCKS:    // index_range.cpp:36-38 (the closing parenthesis is missing there too)
      return std::snprintf(buf, len, "end_index - start_index (which is %d) cannot be smaller than n_blocks (which is %d", n - 2 * P, B);
    case ORC_RING_BLOCK_TOO_SMALL: { // neighbor.hpp:72-73, on the first block slice with fewer than 2 points
      const BlockRange r(P, n - P, B);
      for (int j = 0; j < B; j++) {
        const int size = r.Boundary(j + 1) - r.Boundary(j);
        if (size < 2) {return std::snprintf(buf, len, "The input point size (which is %d) cannot be smaller than 2", size);}
      }
      return std::snprintf(buf, len, "%s", "");
    }
    case ORC_RING_ZERO_NORM_PAIR:    // math.cpp:41
      return std::snprintf(buf, len, "All input values are zero. Angle cannot be calculated");
    default:
      return std::snprintf(buf, len, "%s", "");
  }
}

// pcl::VoxelGrid<PointXYZ>::applyFilter (PCL 1.12.1, filters/impl/voxel_grid.hpp), see lfx_oracle.h: PARITY UNPINNED
int orc_voxel_downsample(const float * points, int n, float leaf, float * out, int * n_out)
{
  *n_out = 0;
  if (n <= 0) {return 0;}
  float mn[3] = {points[0], points[1], points[2]}, mx[3] = {points[0], points[1], points[2]};
  for (int i = 1; i < n; i++) {                                  // getMinMax3D
    for (int a = 0; a < 3; a++) {
      mn[a] = std::min(mn[a], points[4 * i + a]);
      mx[a] = std::max(mx[a], points[4 * i + a]);
    }
  }
  const float inv = 1.0f / leaf;                                 // inverse_leaf_size_ = Ones / leaf_size_
  const long long dx = static_cast<long long>((mx[0] - mn[0]) * inv) + 1, dy = static_cast<long long>((mx[1] - mn[1]) * inv) + 1,
    dz = static_cast<long long>((mx[2] - mn[2]) * inv) + 1;
  // "Leaf size is too small for the input dataset": the product against INT_MAX, factor by factor (three extents of a few
  // million cells overflow 64 bits: found by the sanitizer build, `make asan`)
  const long long lim = 2147483647LL;
  if (dx > lim || dy > lim || dz > lim || dx * dy > lim || dx * dy * dz > lim) {return 1;}
  int min_b[3], div_b[3];
  for (int a = 0; a < 3; a++) {
    min_b[a] = static_cast<int>(std::floor(mn[a] * inv));
    div_b[a] = static_cast<int>(std::floor(mx[a] * inv)) - min_b[a] + 1;
  }
  const int mul1 = div_b[0], mul2 = div_b[0] * div_b[1];
  std::vector<std::pair<unsigned, int>> cells(n);
  for (int i = 0; i < n; i++) {
    const int i0 = static_cast<int>(std::floor(points[4 * i] * inv) - static_cast<float>(min_b[0]));
    const int i1 = static_cast<int>(std::floor(points[4 * i + 1] * inv) - static_cast<float>(min_b[1]));
    const int i2 = static_cast<int>(std::floor(points[4 * i + 2] * inv) - static_cast<float>(min_b[2]));
    cells[i] = {static_cast<unsigned>(i0 + i1 * mul1 + i2 * mul2), i};
  }
  std::sort(cells.begin(), cells.end());                          // (cell, input index): canonical order inside a cell
  int m = 0;
  for (int a = 0; a < n; ) {
    int b = a;
    float sx = 0.f, sy = 0.f, sz = 0.f;                           // AccumulatorXYZ: Eigen::Vector3f sum
    while (b < n && cells[b].first == cells[a].first) {
      const float * p = points + 4 * cells[b].second;
      sx += p[0]; sy += p[1]; sz += p[2];
      b++;
    }
    const float cnt = static_cast<float>(b - a);
    out[4 * m] = sx / cnt; out[4 * m + 1] = sy / cnt; out[4 * m + 2] = sz / cnt; out[4 * m + 3] = 1.0f;
    m++;
    a = b;
  }
  *n_out = m;
  return 0;
}

void orc_label_to_color(uint8_t label, uint8_t rgb[3])   // color_points.cpp:39-68
{
  static const uint8_t table[8][3] = {
    {255, 255, 255}, {255, 0, 0}, {255, 63, 0}, {255, 0, 0}, {255, 63, 0}, {127, 127, 127},
    {255, 0, 255}, {0, 255, 0}};
  const uint8_t * c = table[label & 7];
  rgb[0] = c[0]; rgb[1] = c[1]; rgb[2] = c[2];
}

// ---------------------------------------------------------------------------------------
// Whole scan: feature_extraction.cpp:114-157.
int orc_extract(
  const void * points, size_t n, size_t stride, size_t off_x, size_t off_y, size_t off_z,
  size_t off_ring, const orc_params * p, int canonical_ties,
  uint8_t * labels_out, double * curvature_out, int32_t * sorted_index,
  int32_t * ring_id, int32_t * ring_count, int32_t * ring_status, int32_t max_rings,
  int32_t * n_rings, int32_t * edge_index, int32_t * n_edge, int32_t * surface_index,
  int32_t * n_surface, float * edge_points, float * surface_points, int64_t * ties)
{
  if (!points && n) {return -1;}
  if (!p || p->padding <= 0 || p->n_blocks <= 0) {return -1;}
  const uint8_t * base = static_cast<const uint8_t *>(points);
  const XYSource src{base + off_x, base + off_y, stride, n};
  const bool canonical = canonical_ties != 0;
  int64_t angle_ties = 0, curv_ties = 0;

  if (labels_out) {std::memset(labels_out, ORC_LABEL_DEFAULT, n);}
  if (curvature_out) {std::fill(curvature_out, curvature_out + n, 0.);}

  // ring.hpp:114-125 (MakePointIndices).  The reference iterates an unordered_map; only
  // the per-ring content is defined, so the oracle walks rings in ascending id.
  std::map<int, std::vector<int>> rings;
  for (size_t i = 0; i < n; i++) {
    uint16_t r;
    std::memcpy(&r, base + i * stride + off_ring, 2);
    rings[r].push_back(static_cast<int>(i));
  }

  // ring.hpp:101-112,131-139 (SortByAtan2 on every ring, std::sort, float predicate)
  for (auto & kv : rings) {
    std::vector<int> & idx = kv.second;
    auto less = [&](int a, int b) {return PolarLess<float>(src.X(a), src.Y(a), src.X(b), src.Y(b));};
    if (canonical) {
      std::sort(idx.begin(), idx.end(), [&](int a, int b) {return less(a, b) || (!less(b, a) && a < b);});
    } else {
      std::sort(idx.begin(), idx.end(), less);
    }
    for (size_t i = 0; i + 1 < idx.size(); i++) {
      if (!less(idx[i], idx[i + 1]) && !less(idx[i + 1], idx[i])) {angle_ties++;}
    }
  }

  const double radian_threshold = p->neighbor_degree_threshold * M_PI / 180.0;   // degree_to_radian.hpp:34-37
  int32_t ring_slot = 0, ne = 0, ns = 0;
  size_t cursor = 0;

  for (const auto & kv : rings) {
    const std::vector<int> & indices = kv.second;
    const int N = static_cast<int>(indices.size());
    int status = ORC_RING_OK;
    if (sorted_index) {
      for (int i = 0; i < N; i++) {sorted_index[cursor + i] = indices[i];}
    }
    cursor += N;

    if (N < p->padding + 1) {   // ring.cpp:46-59, called with padding+1 (feature_extraction.cpp:116)
      status = ORC_RING_SPARSE;
    } else {
      const Mapped ref_points(src, indices);
      const NeighborXY is_neighbor(ref_points, radian_threshold);   // outside the try in the reference; N>=2 here
      const RangeOf range(ref_points);
      try {
        Labels labels(N, ORC_LABEL_DEFAULT);                                   // label.hpp:56-59
        const std::vector<double> ranges = range.All(0, range.size());         // :129
        const std::vector<double> curvature = Curvature(ranges, p->padding);   // :130
        const BlockRange blocks(p->padding, N - p->padding, p->n_blocks);      // :131
        AssignBlocks(labels, curvature, is_neighbor, blocks, p->padding, p->edge_threshold,
          p->surface_threshold, canonical, &curv_ties);                        // :133
        OccludedFromLeft(labels, is_neighbor, range, p->padding, p->distance_diff_threshold);   // :135
        OccludedFromRight(labels, is_neighbor, range, p->padding, p->distance_diff_threshold);
        OutOfRange(labels, range, p->min_range, p->max_range);                 // :137
        ParallelBeam(labels, range, p->parallel_beam_min_range_ratio);         // :138

        // :142-151 -- compaction of Edge and Surface; intensity <- (float)curvature (label.hpp:176)
        for (int i = 0; i < N; i++) {
          const int k = indices[i];
          if (labels_out) {labels_out[k] = labels[i];}
          if (curvature_out) {curvature_out[k] = curvature[i];}
          float z;
          std::memcpy(&z, base + static_cast<size_t>(k) * stride + off_z, 4);
          if (labels[i] == ORC_LABEL_EDGE) {
            if (edge_index) {edge_index[ne] = k;}
            if (edge_points) {
              float * q = edge_points + 4 * static_cast<size_t>(ne);
              q[0] = src.X(k); q[1] = src.Y(k); q[2] = z; q[3] = static_cast<float>(curvature[i]);
            }
            ne++;
          } else if (labels[i] == ORC_LABEL_SURFACE) {
            if (surface_index) {surface_index[ns] = k;}
            if (surface_points) {
              float * q = surface_points + 4 * static_cast<size_t>(ns);
              q[0] = src.X(k); q[1] = src.Y(k); q[2] = z; q[3] = static_cast<float>(curvature[i]);
            }
            ns++;
          }
        }
      } catch (const RingSkip & e) {
        status = e.code;      // feature_extraction.cpp:154-156: warn, ring contributes nothing
      } catch (const std::invalid_argument &) {
        status = ORC_RING_OTHER;
      } catch (const std::out_of_range &) {
        status = ORC_RING_OTHER;   // would terminate the reference node (not caught there)
      }
    }
    if (ring_slot < max_rings) {
      if (ring_id) {ring_id[ring_slot] = kv.first;}
      if (ring_count) {ring_count[ring_slot] = N;}
      if (ring_status) {ring_status[ring_slot] = status;}
    }
    ring_slot++;
  }
  if (n_rings) {*n_rings = ring_slot;}
  if (n_edge) {*n_edge = ne;}
  if (n_surface) {*n_surface = ns;}
  if (ties) {ties[0] = angle_ties; ties[1] = curv_ties;}
  return 0;
}

}  // extern "C"
