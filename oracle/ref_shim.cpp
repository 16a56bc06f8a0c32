// ref_shim.cpp -- extern "C" doorways onto the REFERENCE's own compiled functions.
//
// TEST INFRASTRUCTURE ONLY.  This file holds no algorithm: each function forwards to a
// function compiled from the reference's own source file where it lies under
// /root/reference (see Makefile target `ref`; output oracle/_ref/libref_pieces.so).
// Only three reference translation units build with what this image holds
// (math.cpp, convolution.cpp, index_range.cpp -- the last two need {fmt}, whose headers
// ship inside the image's torch wheel); everything else on the path needs PCL /
// range-v3 / boost / rclcpp and is unbuildable here.  tests/ use these doorways to pin
// the oracle's restatement of those pieces bit-for-bit on random inputs.
#include <stdexcept>
#include <vector>

#include "lidar_feature_extraction/convolution.hpp"
#include "lidar_feature_extraction/index_range.hpp"
#include "lidar_feature_extraction/math.hpp"

extern "C" {

double ref_xy_norm(double x, double y) {return XYNorm(x, y);}

int ref_calc_radian(double x1, double y1, double x2, double y2, double * out)
{
  try {
    *out = CalcRadian(x1, y1, x2, y2);
  } catch (const std::invalid_argument &) {
    return 1;
  }
  return 0;
}

double ref_inner_product(const double * a, const double * b, int n)
{
  return InnerProduct(a, a + n, b);
}

int ref_convolution1d(const double * input, int n, const double * weight, int m, double * out)
{
  try {
    const std::vector<double> r =
      Convolution1D(std::vector<double>(input, input + n), std::vector<double>(weight, weight + m));
    for (size_t i = 0; i < r.size(); i++) {out[i] = r[i];}
  } catch (const std::invalid_argument &) {
    return 1;
  }
  return 0;
}

int ref_padded_index_range(int size, int n_blocks, int padding, int * bounds)
{
  try {
    const PaddedIndexRange r(size, n_blocks, padding);
    for (int j = 0; j < n_blocks; j++) {
      bounds[j] = r.Begin(j);
      bounds[j + 1] = r.End(j);
    }
  } catch (const std::invalid_argument &) {
    return 1;
  }
  return 0;
}

}  // extern "C"
