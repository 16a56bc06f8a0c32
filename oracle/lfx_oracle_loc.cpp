// lfx_oracle_loc.cpp -- CPU oracle for the FIRST STEP OF THE CONSUMER of the extraction path:
// the scan-to-map residual build of the reference's localization package.  TEST INFRASTRUCTURE ONLY (see lfx_oracle.h).
//
// Restated from /root/reference/localization: edge.hpp:61-131 + src/edge.cpp:38-97 (Edge::Make, CalcMeanAndCovariance,
// TripletCross, MakeEdgeJacobianRow, MakeEdgeResidual, PrincipalIsReliable), surface.hpp:37-145 (Surface::
// MakeFromDownsampled, EstimatePlaneCoefficients, SignedPointPlaneDistance, MakeJacobianRow), kdtree.hpp / src/kdtree.cpp
// (NearestKSearch), math.hpp:36-40 (SolveLinear), rotationlib/src/jacobian/quaternion.cpp:35-52 (DRpDq),
// rotationlib/src/hat.cpp:35-43 (Hat), lib/src/algorithm.cpp:33-50 (SortThreeValues).
//
// PARITY UNPINNED beyond the vectors of localization/test/test_edge.cpp and test_math.cpp: the arithmetic underneath is
// third party and absent from this image -- Eigen (sums of colwise().mean(), D^T D, SelfAdjointEigenSolver::
// computeDirect, householderQr, Quaternion(Matrix3)) and nanoflann 1.x (un-vendored submodule: exact L2 k-nearest
// search, order of equidistant neighbours unspecified).  This file uses plain sequential double arithmetic, a Jacobi
// eigen-iteration (deliberately NOT the closed form the HIP path uses, so that the two check each other), Householder QR,
// and breaks distance ties by the lower map index.  Comparisons with it are by tolerance, never by bit pattern.
#include "lfx_oracle.h"

#include <algorithm>
#include <cmath>
#include <utility>
#include <vector>

namespace
{
struct V3 { double x, y, z; };
V3 operator+(V3 a, V3 b) {return {a.x + b.x, a.y + b.y, a.z + b.z};}
V3 operator-(V3 a, V3 b) {return {a.x - b.x, a.y - b.y, a.z - b.z};}
V3 operator*(double s, V3 a) {return {s * a.x, s * a.y, s * a.z};}
V3 cross(V3 a, V3 b) {return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};}
double dot(V3 a, V3 b) {return a.x * b.x + a.y * b.y + a.z * b.z;}

// Eigen::Quaterniond(Matrix3d) (Eigen/src/Geometry/Quaternion.h, quaternionbase_assign_impl<Other, 3, 3>); m row-major
void QuaternionFromRotation(const double * m, double & w, double q[3])
{
  auto M = [&](int r, int c) {return m[3 * r + c];};
  double t = M(0, 0) + M(1, 1) + M(2, 2);
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
}

V3 Transform(const double * pose, V3 p)   // Isometry3d * Vector3d; pose = [R | t] row-major 3 x 4
{
  return {pose[0] * p.x + pose[1] * p.y + pose[2] * p.z + pose[3], pose[4] * p.x + pose[5] * p.y + pose[6] * p.z + pose[7],
    pose[8] * p.x + pose[9] * p.y + pose[10] * p.z + pose[11]};
}

// rotationlib::DRpDq (jacobian/quaternion.cpp:35-52), out 3 x 4 row-major
void DRpDq(double w, const double v_[3], V3 p, double out[12])
{
  const V3 v{v_[0], v_[1], v_[2]};
  const V3 c0 = w * p + cross(v, p);
  const double vp = dot(v, p);
  const double K[9] = {0., -p.z, p.y, p.z, 0., -p.x, -p.y, p.x, 0.};      // Hat(p), hat.cpp:35-43
  const double vv[3] = {v.x, v.y, v.z}, pp[3] = {p.x, p.y, p.z};
  const double c0v[3] = {c0.x, c0.y, c0.z};
  for (int r = 0; r < 3; r++) {
    out[4 * r] = 2. * c0v[r];
    for (int c = 0; c < 3; c++) {
      out[4 * r + 1 + c] = 2. * ((r == c ? vp : 0.) + vv[r] * pp[c] - pp[r] * vv[c] - w * K[3 * r + c]);
    }
  }
}

// exact k nearest map points of q (squared L2, ascending; ties by the lower index)
void Nearest(const float * map, int n_map, V3 q, int k, std::vector<int> & idx)
{
  std::vector<std::pair<double, int>> d(n_map);
  for (int i = 0; i < n_map; i++) {
    const double dx = (double)map[4 * i] - q.x, dy = (double)map[4 * i + 1] - q.y, dz = (double)map[4 * i + 2] - q.z;
    d[i] = {dx * dx + dy * dy + dz * dz, i};
  }
  std::partial_sort(d.begin(), d.begin() + k, d.end());
  idx.resize(k);
  for (int i = 0; i < k; i++) {idx[i] = d[i].second;}
}

void MeanCov(const double * X, int n, double mean[3], double cov[9])   // edge.cpp:43-49
{
  for (int a = 0; a < 3; a++) {
    double s = 0.;
    for (int i = 0; i < n; i++) {s += X[3 * i + a];}
    mean[a] = s / n;
  }
  for (int a = 0; a < 3; a++) {
    for (int b = 0; b < 3; b++) {
      double s = 0.;
      for (int i = 0; i < n; i++) {s += (X[3 * i + a] - mean[a]) * (X[3 * i + b] - mean[b]);}
      cov[3 * a + b] = s / n;
    }
  }
}

// eigen-decomposition of a symmetric 3 x 3 matrix by cyclic Jacobi rotations; eigenvalues ascending, eigenvectors as
// COLUMNS of V (row-major storage), each of unit length -- the contract of SelfAdjointEigenSolver (edge.cpp:59-64)
void Jacobi3(const double * C, double ev[3], double V[9])
{
  double A[3][3], Q[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int r = 0; r < 3; r++) {for (int c = 0; c < 3; c++) {A[r][c] = C[3 * r + c];}}
  for (int sweep = 0; sweep < 60; sweep++) {
    const double off = std::fabs(A[0][1]) + std::fabs(A[0][2]) + std::fabs(A[1][2]);
    if (off == 0.) {break;}
    for (int p = 0; p < 2; p++) {
      for (int q = p + 1; q < 3; q++) {
        if (A[p][q] == 0.) {continue;}
        const double theta = (A[q][q] - A[p][p]) / (2. * A[p][q]);
        const double t = (theta >= 0. ? 1. : -1.) / (std::fabs(theta) + std::sqrt(theta * theta + 1.));
        const double c = 1. / std::sqrt(t * t + 1.), s = t * c;
        for (int k = 0; k < 3; k++) {
          const double akp = A[k][p], akq = A[k][q];
          A[k][p] = c * akp - s * akq; A[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; k++) {
          const double apk = A[p][k], aqk = A[q][k];
          A[p][k] = c * apk - s * aqk; A[q][k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; k++) {
          const double qkp = Q[k][p], qkq = Q[k][q];
          Q[k][p] = c * qkp - s * qkq; Q[k][q] = s * qkp + c * qkq;
        }
      }
    }
  }
  int order[3] = {0, 1, 2};
  std::sort(order, order + 3, [&](int a, int b) {return A[a][a] < A[b][b];});
  for (int c = 0; c < 3; c++) {
    ev[c] = A[order[c]][order[c]];
    for (int r = 0; r < 3; r++) {V[3 * r + c] = Q[r][order[c]];}
  }
}

// least squares A x = b by Householder QR (math.hpp:36-40: A.householderQr().solve(b)); A rows x cols row-major, rows >= cols
void SolveLinear(std::vector<double> A, int rows, int cols, std::vector<double> b, double * x)
{
  for (int c = 0; c < cols; c++) {
    double norm = 0.;
    for (int r = c; r < rows; r++) {norm += A[r * cols + c] * A[r * cols + c];}
    norm = std::sqrt(norm);
    if (norm == 0.) {continue;}
    const double alpha = A[c * cols + c] > 0. ? -norm : norm;
    std::vector<double> v(rows, 0.);
    for (int r = c; r < rows; r++) {v[r] = A[r * cols + c];}
    v[c] -= alpha;
    double vv = 0.;
    for (int r = c; r < rows; r++) {vv += v[r] * v[r];}
    if (vv == 0.) {continue;}
    for (int cc = c; cc < cols; cc++) {
      double s = 0.;
      for (int r = c; r < rows; r++) {s += v[r] * A[r * cols + cc];}
      s = 2. * s / vv;
      for (int r = c; r < rows; r++) {A[r * cols + cc] -= s * v[r];}
    }
    double s = 0.;
    for (int r = c; r < rows; r++) {s += v[r] * b[r];}
    s = 2. * s / vv;
    for (int r = c; r < rows; r++) {b[r] -= s * v[r];}
  }
  for (int c = cols - 1; c >= 0; c--) {
    double s = b[c];
    for (int cc = c + 1; cc < cols; cc++) {s -= A[c * cols + cc] * x[cc];}
    x[c] = s / A[c * cols + c];
  }
}
}  // namespace

extern "C" {

void orc_loc_triplet_cross(const double * p0, const double * p1, const double * p2, double * out)   // edge.cpp:51-57
{
  const V3 a{p0[0], p0[1], p0[2]}, b{p1[0], p1[1], p1[2]}, c{p2[0], p2[1], p2[2]};
  const V3 r = cross(c - b, cross(a - b, a - c));
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

void orc_loc_mean_cov(const double * X, int n, double * mean, double * cov) {MeanCov(X, n, mean, cov);}

void orc_loc_principal(const double * cov, double * eigenvalues, double * eigenvectors) {Jacobi3(cov, eigenvalues, eigenvectors);}

int orc_loc_principal_is_reliable(const double * ev)   // edge.cpp:92-96 + lib/src/algorithm.cpp:33-50
{
  double a = ev[0], b = ev[1], c = ev[2];
  if (a > b) {std::swap(a, b);}
  if (b > c) {std::swap(b, c);}
  if (a > b) {std::swap(a, b);}
  return c > b * 3.0;
}

void orc_loc_solve_linear(const double * A, int rows, int cols, const double * b, double * x)
{
  SolveLinear(std::vector<double>(A, A + rows * cols), rows, cols, std::vector<double>(b, b + rows), x);
}

void orc_loc_quaternion(const double * R, double * wxyz)
{
  QuaternionFromRotation(R, wxyz[0], wxyz + 1);
}

// Edge::Make (edge.hpp:86-124).  map / points: records of 4 floats; pose: [R | t] row-major 3 x 4 (point_to_map);
// residual [n][3], jacobian [n][3][7] row-major.
void orc_loc_edge_residuals(const float * map, int n_map, const double * pose, int k, const float * points, int n,
  double * residual, double * jacobian)
{
  const double R[9] = {pose[0], pose[1], pose[2], pose[4], pose[5], pose[6], pose[8], pose[9], pose[10]};
  double w, v[3];
  QuaternionFromRotation(R, w, v);
  std::vector<int> idx;
  std::vector<double> X(3 * k);
  for (int i = 0; i < n; i++) {
    const V3 p0{(double)points[4 * i], (double)points[4 * i + 1], (double)points[4 * i + 2]};
    const V3 query = Transform(pose, p0);
    Nearest(map, n_map, query, k, idx);
    for (int j = 0; j < k; j++) {for (int a = 0; a < 3; a++) {X[3 * j + a] = (double)map[4 * idx[j] + a];}}
    double mean[3], cov[9], ev[3], V[9];
    MeanCov(X.data(), k, mean, cov);
    Jacobi3(cov, ev, V);
    const V3 principal{V[2], V[5], V[8]};                               // eigenvectors.col(2)
    const V3 m{mean[0], mean[1], mean[2]};
    const V3 p1 = m - principal, p2 = m + principal;
    double d[12];
    DRpDq(w, v, p0, d);
    const V3 e = p2 - p1;
    const double K[9] = {0., -e.z, e.y, e.z, 0., -e.x, -e.y, e.x, 0.};   // Hat(p2 - p1)
    double * J = jacobian + 21 * i;
    for (int r = 0; r < 3; r++) {
      for (int c = 0; c < 4; c++) {
        J[7 * r + c] = K[3 * r] * d[c] + K[3 * r + 1] * d[4 + c] + K[3 * r + 2] * d[8 + c];
      }
      for (int c = 0; c < 3; c++) {J[7 * r + 4 + c] = K[3 * r + c];}
    }
    const V3 p = Transform(pose, p0);
    const V3 r = cross(p - p1, p - p2);                                 // MakeEdgeResidual, edge.cpp:76-84
    residual[3 * i] = r.x; residual[3 * i + 1] = r.y; residual[3 * i + 2] = r.z;
  }
}

// Surface::MakeFromDownsampled (surface.hpp:116-139).  residual [n], jacobian [n][7].
void orc_loc_surface_residuals(const float * map, int n_map, const double * pose, int k, const float * points, int n,
  double * residual, double * jacobian)
{
  const double R[9] = {pose[0], pose[1], pose[2], pose[4], pose[5], pose[6], pose[8], pose[9], pose[10]};
  double w, v[3];
  QuaternionFromRotation(R, w, v);
  std::vector<int> idx;
  std::vector<double> X(3 * k), g(k, -1.0);                              // plane_bias = 1.0: g = -1
  for (int i = 0; i < n; i++) {
    const V3 p{(double)points[4 * i], (double)points[4 * i + 1], (double)points[4 * i + 2]};
    const V3 on_map = Transform(pose, p);
    Nearest(map, n_map, on_map, k, idx);
    for (int j = 0; j < k; j++) {for (int a = 0; a < 3; a++) {X[3 * j + a] = (double)map[4 * idx[j] + a];}}
    double wv[3];
    SolveLinear(X, k, 3, g, wv);                                         // EstimatePlaneCoefficients, surface.hpp:78-83
    const double norm = std::sqrt(wv[0] * wv[0] + wv[1] * wv[1] + wv[2] * wv[2]);
    const double u[3] = {wv[0] / norm, wv[1] / norm, wv[2] / norm};
    double d[12];
    DRpDq(w, v, p, d);
    double * J = jacobian + 7 * i;
    for (int c = 0; c < 4; c++) {J[c] = u[0] * d[c] + u[1] * d[4 + c] + u[2] * d[8 + c];}
    for (int c = 0; c < 3; c++) {J[4 + c] = u[c];}
    residual[i] = (wv[0] * on_map.x + wv[1] * on_map.y + wv[2] * on_map.z + 1.0) / norm;   // SignedPointPlaneDistance
  }
}

}  // extern "C"
