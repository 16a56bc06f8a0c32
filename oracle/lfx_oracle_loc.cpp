// lfx_oracle_loc.cpp -- CPU oracle for the FIRST STEP OF THE CONSUMER of the extraction path:
// the scan-to-map residual build of the reference's localization package.  TEST INFRASTRUCTURE ONLY (see lfx_oracle.h).
//
// Restated from /root/reference/localization: edge.hpp:61-131 + src/edge.cpp:38-97 (Edge::Make, CalcMeanAndCovariance,
// TripletCross, MakeEdgeJacobianRow, MakeEdgeResidual, PrincipalIsReliable), surface.hpp:37-145 (Surface::
// MakeFromDownsampled, EstimatePlaneCoefficients, SignedPointPlaneDistance, MakeJacobianRow), kdtree.hpp / src/kdtree.cpp
// (NearestKSearch), math.hpp:36-40 (SolveLinear), rotationlib/src/jacobian/quaternion.cpp:35-52 (DRpDq),
// rotationlib/src/hat.cpp:35-43 (Hat), lib/src/algorithm.cpp:33-50 (SortThreeValues).
//
// ... and of the optimizer around it (optimizer.hpp:70-127 Optimizer::Run, src/optimizer.cpp:35-127 CheckConvergence,
// WeightedUpdate, MakeM, CalcUpdate, ComputeErrors, NormalizeErrorScale, ComputeWeights; src/robust.cpp:36-71;
// src/degenerate.cpp:32-37; src/posevec.cpp:32-55; lib/src/stats.cpp:34-67 Median; src/alignment.cpp:33-78 the
// point-pair problem the reference's optimizer tests run; rotationlib/src/quaternion.cpp:45-60).
//
// PARITY UNPINNED beyond the vectors of localization/test/test_edge.cpp, test_math.cpp, test_robust.cpp,
// test_degenerate.cpp, test_posevec.cpp and test_optimizer.cpp: the arithmetic underneath is
// third party and absent from this image -- Eigen (sums of colwise().mean(), D^T D, SelfAdjointEigenSolver::
// computeDirect, householderQr, Quaternion(Matrix3)) and nanoflann 1.x (un-vendored submodule: exact L2 k-nearest
// search, order of equidistant neighbours unspecified).  This file uses plain sequential double arithmetic, a Jacobi
// eigen-iteration (deliberately NOT the closed form the HIP path uses, so that the two check each other), Householder QR,
// and breaks distance ties by the lower map index.  Comparisons with it are by tolerance, never by bit pattern.
#include "lfx_oracle.h"

#include <algorithm>
#include <cmath>
#include <functional>
#include <limits>
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
// Returns false where R has a (numerically) zero pivot: no solution worth the name; Eigen divides by it all the same, what
// comes out cannot be known here (its arithmetic is not in the image).
bool SolveLinear(std::vector<double> A, int rows, int cols, std::vector<double> b, double * x)
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
  double rmax = 0., rmin = std::numeric_limits<double>::infinity();
  for (int c = 0; c < cols; c++) {
    rmax = std::max(rmax, std::fabs(A[c * cols + c]));
    rmin = std::min(rmin, std::fabs(A[c * cols + c]));
  }
  for (int c = cols - 1; c >= 0; c--) {
    double s = b[c];
    for (int cc = c + 1; cc < cols; cc++) {s -= A[c * cols + cc] * x[cc];}
    x[c] = s / A[c * cols + c];
  }
  return rmin > 1e-9 * rmax;
}

// ---- the optimizer (optimizer.hpp / src/optimizer.cpp) ---------------------------------------------------------------
double MedianOf(std::vector<double> v)                                  // lib/src/stats.cpp:34-55
{
  const size_t size = v.size();
  if (size % 2 == 1) {
    const size_t n = (size - 1) / 2;
    std::nth_element(v.begin(), v.begin() + n, v.end());
    return v[n];
  }
  const size_t n = size / 2;
  std::nth_element(v.begin(), v.begin() + n, v.end());
  const double e0 = v[n];
  std::nth_element(v.begin(), v.begin() + n - 1, v.end());
  const double e1 = v[n - 1];
  return (e0 + e1) / 2.;
}

double Mad(const std::vector<double> & v)                               // robust.cpp:36-40
{
  const double median = MedianOf(v);
  std::vector<double> dev(v.size());
  for (size_t i = 0; i < v.size(); i++) {dev[i] = std::fabs(v[i] - median);}
  return MedianOf(dev);
}

double ScaleOf(const std::vector<double> & v) {return 1.482602218505602 * Mad(v);}   // robust.cpp:42-50

double HuberOf(double e, double k) {return e < k * k ? e : 2 * k * std::sqrt(e) - k * k;}       // robust.cpp:52-59
double HuberDerivativeOf(double e, double k) {return e < k * k ? 1. : k / std::sqrt(e);}       // robust.cpp:61-68

// eigenvalues of a symmetric n x n matrix (row-major) by cyclic Jacobi rotations, ascending
void JacobiEigenvalues(std::vector<double> A, int n, std::vector<double> & ev)
{
  for (int sweep = 0; sweep < 100; sweep++) {
    double off = 0., diag = 0.;
    for (int i = 0; i < n; i++) {
      diag += A[i * n + i] * A[i * n + i];
      for (int j = i + 1; j < n; j++) {off += A[i * n + j] * A[i * n + j];}
    }
    if (off <= 1e-30 * diag || off == 0.) {break;}
    for (int p = 0; p < n; p++) {
      for (int q = p + 1; q < n; q++) {
        const double apq = A[p * n + q];
        if (apq == 0.) {continue;}
        const double theta = (A[q * n + q] - A[p * n + p]) / (2. * apq);
        const double t = (theta >= 0. ? 1. : -1.) / (std::fabs(theta) + std::sqrt(theta * theta + 1.));
        const double c = 1. / std::sqrt(t * t + 1.), s = t * c;
        for (int k = 0; k < n; k++) {                                   // A <- A G
          const double akp = A[k * n + p], akq = A[k * n + q];
          A[k * n + p] = c * akp - s * akq;
          A[k * n + q] = s * akp + c * akq;
        }
        for (int k = 0; k < n; k++) {                                   // A <- G^T A
          const double apk = A[p * n + k], aqk = A[q * n + k];
          A[p * n + k] = c * apk - s * aqk;
          A[q * n + k] = s * apk + c * aqk;
        }
      }
    }
  }
  ev.resize(n);
  for (int i = 0; i < n; i++) {ev[i] = A[i * n + i];}
  std::sort(ev.begin(), ev.end());
}

bool IsDegenerateOf(const std::vector<double> & C, int n, double threshold)      // degenerate.cpp:32-37
{
  std::vector<double> ev;
  JacobiEigenvalues(C, n, ev);
  for (double e : ev) {if (std::fabs(e) < threshold) {return true;}}
  return false;
}

void RotationOfQuaternion(const double q[4] /* w x y z */, double R[9])    // Eigen QuaternionBase::toRotationMatrix
{
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  const double tx = 2. * x, ty = 2. * y, tz = 2. * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1. - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1. - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1. - (txx + tyy);
}

void PoseOf(const double q[4], const double t[3], double pose[12])         // posevec.cpp:47-55 MakePose
{
  double R[9];
  RotationOfQuaternion(q, R);
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) {pose[4 * r + c] = R[3 * r + c];}
    pose[4 * r + 3] = t[r];
  }
}

void AngleAxisToQuaternionOf(const double theta[3], double q[4])           // posevec.cpp:32-45
{
  const double k = std::sqrt(theta[0] * theta[0] + theta[1] * theta[1] + theta[2] * theta[2]);
  if (k < 1e-8) {q[0] = 1.; q[1] = q[2] = q[3] = 0.; return;}
  const double s = std::sin(k / 2.);
  q[0] = std::cos(k / 2.);
  for (int a = 0; a < 3; a++) {q[1 + a] = (theta[a] / k) * s;}
}

void MakeMOf(const double q[4], double M[42])                              // optimizer.cpp:73-84, 7 x 6 row-major
{
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  const double L[16] = {w, -x, -y, -z, x, w, -z, y, y, z, w, -x, z, -y, x, w};   // rotationlib quaternion.cpp:45-60
  for (int i = 0; i < 42; i++) {M[i] = 0.;}
  for (int r = 0; r < 4; r++) {for (int c = 0; c < 3; c++) {M[6 * r + c] = 0.5 * L[4 * r + 1 + c];}}
  for (int a = 0; a < 3; a++) {M[6 * (4 + a) + 3 + a] = 1.;}
}

// rows: residuals of dimension 3 (r3 [n3][3], J3 [n3][3][7]) followed by residuals of dimension 1 (r1 [n1], J1 [n1][7])
struct RowSet
{
  std::vector<double> J3, r3, J1, r1;
  size_t size() const {return r3.size() / 3 + r1.size();}
};

// WeightedUpdate (optimizer.cpp:40-71); weights in row order
bool WeightedUpdateOf(const double M[42], const std::vector<double> & weights, const RowSet & rows, double dx[6])
{
  std::vector<double> D(49, 0.), A(49, 0.);
  double b[7] = {0., 0., 0., 0., 0., 0., 0.};
  const size_t n3 = rows.r3.size() / 3;
  for (size_t i = 0; i < rows.size(); i++) {
    const int dim = i < n3 ? 3 : 1;
    const double * J = i < n3 ? &rows.J3[21 * i] : &rows.J1[7 * (i - n3)];
    const double * r = i < n3 ? &rows.r3[3 * i] : &rows.r1[i - n3];
    double JtJ[49], Jtr[7];
    for (int a = 0; a < 7; a++) {
      for (int c = 0; c < 7; c++) {
        double s = 0.;
        for (int k = 0; k < dim; k++) {s += J[7 * k + a] * J[7 * k + c];}
        JtJ[7 * a + c] = s;
      }
      double s = 0.;
      for (int k = 0; k < dim; k++) {s += J[7 * k + a] * r[k];}
      Jtr[a] = s;
    }
    for (int a = 0; a < 49; a++) {D[a] += JtJ[a]; A[a] += weights[i] * JtJ[a];}
    for (int a = 0; a < 7; a++) {b[a] += weights[i] * Jtr[a];}
  }
  for (int a = 0; a < 6; a++) {dx[a] = 0.;}
  if (IsDegenerateOf(D, 7, 0.1)) {return false;}
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
  // llt().solve: H = L L^T
  double Lc[36] = {0.};
  for (int j = 0; j < 6; j++) {
    double s = H[6 * j + j];
    for (int k = 0; k < j; k++) {s -= Lc[6 * j + k] * Lc[6 * j + k];}
    Lc[6 * j + j] = std::sqrt(s);
    for (int i = j + 1; i < 6; i++) {
      double v = H[6 * i + j];
      for (int k = 0; k < j; k++) {v -= Lc[6 * i + k] * Lc[6 * j + k];}
      Lc[6 * i + j] = v / Lc[6 * j + j];
    }
  }
  double y[6], x[6];
  for (int i = 0; i < 6; i++) {
    double v = g[i];
    for (int k = 0; k < i; k++) {v -= Lc[6 * i + k] * y[k];}
    y[i] = v / Lc[6 * i + i];
  }
  for (int i = 5; i >= 0; i--) {
    double v = y[i];
    for (int k = i + 1; k < 6; k++) {v -= Lc[6 * k + i] * x[k];}
    x[i] = v / Lc[6 * i + i];
  }
  for (int a = 0; a < 6; a++) {dx[a] = -x[a];}
  return true;
}

// Optimizer::Run (optimizer.hpp:79-123).  make(pose, rows) is ProblemType::Make.  result: pose[12], error, error_scale,
// iteration, code (0 converged, 1 larger error, 2 larger scale, 3 maximum iteration, 4 empty input, 5 no surface row with a
// plane: not the reference's, see below); returns success.
int RunOptimizer(const std::function<void(const double *, RowSet &)> & make, const double * initial_pose, int max_iter,
  double * pose_out, double * error_out, double * scale_out, int * iteration_out, int * code_out)
{
  const double R0[9] = {initial_pose[0], initial_pose[1], initial_pose[2], initial_pose[4], initial_pose[5], initial_pose[6],
    initial_pose[8], initial_pose[9], initial_pose[10]};
  double q[4], t[3] = {initial_pose[3], initial_pose[7], initial_pose[11]};
  QuaternionFromRotation(R0, q[0], q + 1);
  double prev_scale = std::numeric_limits<double>::max(), prev_error = std::numeric_limits<double>::max();
  auto finish = [&](int iteration, double error, double scale, int code) {
      PoseOf(q, t, pose_out);
      *error_out = error; *scale_out = scale; *iteration_out = iteration; *code_out = code;
      return code <= 2 ? 1 : 0;
    };
  RowSet rows;
  for (int iter = 0; iter < max_iter; iter++) {
    double pose[12];
    PoseOf(q, t, pose);
    make(pose, rows);
    if (rows.size() == 0) {return finish(iter, 0., 0., 4);}
    const size_t n3 = rows.r3.size() / 3;
    {
      // Not the reference's: where a surface neighbourhood spans no plane, surface.hpp:78-83 solves with a zero pivot and
      // Eigen hands back NaN (the reference then runs to its iteration limit with a NaN pose); that arithmetic is not
      // available here, such a row is the zero row (SurfaceRows below), and a scan whose surface rows are ALL zero rows ends
      // with a failure of its own (code 5) instead of reading rows that say nothing as "converged".
      const size_t n1 = rows.r1.size();
      size_t with_plane = 0;
      for (size_t i = 0; i < n1; i++) {
        const double * J = &rows.J1[7 * i];
        if (J[4] != 0. || J[5] != 0. || J[6] != 0.) {with_plane++;}
      }
      if (n1 != 0 && with_plane == 0) {return finish(iter, 0., 0., 5);}
    }
    std::vector<double> errors(rows.size());
    for (size_t i = 0; i < rows.size(); i++) {
      if (i < n3) {
        const double * r = &rows.r3[3 * i];
        errors[i] = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
      } else {
        errors[i] = rows.r1[i - n3] * rows.r1[i - n3];
      }
    }
    const double scale = ScaleOf(errors);
    double error = 0.;
    for (double e : errors) {error += e;}
    if (error > prev_error) {return finish(iter, error, scale, 1);}
    prev_error = error;
    if (scale > prev_scale) {return finish(iter, error, scale, 2);}
    prev_scale = scale;
    std::vector<double> weights(errors.size());
    for (size_t i = 0; i < errors.size(); i++) {weights[i] = HuberDerivativeOf(errors[i] / (scale + 1e-16), 1.345);}
    double M[42], dx[6], dq[4];
    MakeMOf(q, M);
    WeightedUpdateOf(M, weights, rows, dx);
    AngleAxisToQuaternionOf(dx, dq);
    const double qn[4] = {                                               // q * dq (Eigen quaternion product)
      q[0] * dq[0] - q[1] * dq[1] - q[2] * dq[2] - q[3] * dq[3],
      q[0] * dq[1] + q[1] * dq[0] + q[2] * dq[3] - q[3] * dq[2],
      q[0] * dq[2] + q[2] * dq[0] + q[3] * dq[1] - q[1] * dq[3],
      q[0] * dq[3] + q[3] * dq[0] + q[1] * dq[2] - q[2] * dq[1]};
    for (int a = 0; a < 4; a++) {q[a] = qn[a];}
    for (int a = 0; a < 3; a++) {t[a] += dx[3 + a];}
    const double nq = std::sqrt(dq[1] * dq[1] + dq[2] * dq[2] + dq[3] * dq[3]);
    const double nt = std::sqrt(dx[3] * dx[3] + dx[4] * dx[4] + dx[5] * dx[5]);
    if (nq < 1e-3 && nt < 1e-3) {return finish(iter, error, scale, 0);}   // CheckConvergence, optimizer.cpp:35-38
  }
  return finish(max_iter, prev_error, prev_scale, 3);
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
    const bool plane = SolveLinear(X, k, 3, g, wv);                      // EstimatePlaneCoefficients, surface.hpp:78-83
    double * J = jacobian + 7 * i;
    if (!plane) {
      // the k neighbours coincide or lie on one line: the zero row (weight 0 in every sum of the optimizer; see
      // scan_to_map_row in lfx_kernels_localize.hpp).  What Eigen returns here is not available.
      for (int c = 0; c < 7; c++) {J[c] = 0.;}
      residual[i] = 0.;
      continue;
    }
    const double norm = std::sqrt(wv[0] * wv[0] + wv[1] * wv[1] + wv[2] * wv[2]);
    const double u[3] = {wv[0] / norm, wv[1] / norm, wv[2] / norm};
    double d[12];
    DRpDq(w, v, p, d);
    for (int c = 0; c < 4; c++) {J[c] = u[0] * d[c] + u[1] * d[4 + c] + u[2] * d[8 + c];}
    for (int c = 0; c < 3; c++) {J[4 + c] = u[c];}
    residual[i] = (wv[0] * on_map.x + wv[1] * on_map.y + wv[2] * on_map.z + 1.0) / norm;   // SignedPointPlaneDistance
  }
}

// KDTreeEigen::NearestKSearch (src/kdtree.cpp:44-68): neighbours [k][3], squared distances [k], indices [k]
void orc_loc_nearest(const float * map, int n_map, const double * query, int k, double * neighbours, double * squared_distances, int * indices)
{
  std::vector<int> idx;
  const V3 q{query[0], query[1], query[2]};
  Nearest(map, n_map, q, k, idx);
  for (int j = 0; j < k; j++) {
    const double dx = (double)map[4 * idx[j]] - q.x, dy = (double)map[4 * idx[j] + 1] - q.y, dz = (double)map[4 * idx[j] + 2] - q.z;
    for (int a = 0; a < 3; a++) {neighbours[3 * j + a] = (double)map[4 * idx[j] + a];}
    squared_distances[j] = dx * dx + dy * dy + dz * dz;
    indices[j] = idx[j];
  }
}

void orc_loc_drp_dq(const double * wxyz, const double * p, double * out /* 3 x 4 */)   // rotationlib jacobian/quaternion.cpp:35-52
{
  DRpDq(wxyz[0], wxyz + 1, V3{p[0], p[1], p[2]}, out);
}

double orc_loc_median(const double * v, int n) {return MedianOf(std::vector<double>(v, v + n));}
double orc_loc_mad(const double * v, int n) {return Mad(std::vector<double>(v, v + n));}
double orc_loc_scale(const double * v, int n) {return ScaleOf(std::vector<double>(v, v + n));}
double orc_loc_huber(double e, double k) {return HuberOf(e, k);}
double orc_loc_huber_derivative(double e, double k) {return HuberDerivativeOf(e, k);}
int orc_loc_is_degenerate(const double * C, int n, double threshold) {return IsDegenerateOf(std::vector<double>(C, C + n * n), n, threshold);}
void orc_loc_angle_axis_to_quaternion(const double * theta, double * wxyz) {AngleAxisToQuaternionOf(theta, wxyz);}
void orc_loc_rotation_matrix(const double * wxyz, double * R) {RotationOfQuaternion(wxyz, R);}
void orc_loc_make_m(const double * wxyz, double * M) {MakeMOf(wxyz, M);}

// one CalcUpdate (optimizer.cpp:86-97) of the point-pair problem: dq [4] (w x y z), dt [3]
void orc_loc_pairs_update(const double * X, const double * Y, int n, const double * pose, double * dq, double * dt);

// AlignmentProblem (alignment.cpp:33-78) under Optimizer::Run: X, Y [n][3]
int orc_loc_optimize_pairs(const double * X, const double * Y, int n, const double * initial_pose, int max_iter,
  double * pose_out, double * error_out, double * scale_out, int * iteration_out, int * code_out)
{
  auto make = [&](const double * pose, RowSet & rows) {
      const double R[9] = {pose[0], pose[1], pose[2], pose[4], pose[5], pose[6], pose[8], pose[9], pose[10]};
      double w, v[3];
      QuaternionFromRotation(R, w, v);
      rows.J3.assign(21 * (size_t)n, 0.); rows.r3.resize(3 * (size_t)n); rows.J1.clear(); rows.r1.clear();
      for (int i = 0; i < n; i++) {
        const V3 x{X[3 * i], X[3 * i + 1], X[3 * i + 2]};
        double d[12];
        DRpDq(w, v, x, d);
        for (int r = 0; r < 3; r++) {
          for (int c = 0; c < 4; c++) {rows.J3[21 * i + 7 * r + c] = d[4 * r + c];}
          rows.J3[21 * i + 7 * r + 4 + r] = 1.;
        }
        const V3 p = Transform(pose, x);
        rows.r3[3 * i] = p.x - Y[3 * i]; rows.r3[3 * i + 1] = p.y - Y[3 * i + 1]; rows.r3[3 * i + 2] = p.z - Y[3 * i + 2];
      }
    };
  return RunOptimizer(make, initial_pose, max_iter, pose_out, error_out, scale_out, iteration_out, code_out);
}

// LOAMOptimizationProblem::Make (loam_optimization_problem.hpp:62-84) under Optimizer::Run (localizer.hpp:76); the surface
// cloud is the one AFTER Downsample (surface.hpp:111), edge rows first
int orc_loc_optimize_scan(const float * edge_map, int n_edge_map, const float * surface_map, int n_surface_map, int k,
  const float * edge_points, int n_edge, const float * surface_points, int n_surface, const double * initial_pose, int max_iter,
  double * pose_out, double * error_out, double * scale_out, int * iteration_out, int * code_out)
{
  auto make = [&](const double * pose, RowSet & rows) {
      rows.J3.resize(21 * (size_t)n_edge); rows.r3.resize(3 * (size_t)n_edge);
      rows.J1.resize(7 * (size_t)n_surface); rows.r1.resize((size_t)n_surface);
      orc_loc_edge_residuals(edge_map, n_edge_map, pose, k, edge_points, n_edge, rows.r3.data(), rows.J3.data());
      orc_loc_surface_residuals(surface_map, n_surface_map, pose, k, surface_points, n_surface, rows.r1.data(), rows.J1.data());
    };
  return RunOptimizer(make, initial_pose, max_iter, pose_out, error_out, scale_out, iteration_out, code_out);
}

void orc_loc_pairs_update(const double * X, const double * Y, int n, const double * pose, double * dq, double * dt)
{
  const double R[9] = {pose[0], pose[1], pose[2], pose[4], pose[5], pose[6], pose[8], pose[9], pose[10]};
  double q[4];
  QuaternionFromRotation(R, q[0], q + 1);
  RowSet rows;
  rows.J3.assign(21 * (size_t)n, 0.); rows.r3.resize(3 * (size_t)n);
  std::vector<double> errors(n);
  for (int i = 0; i < n; i++) {
    const V3 x{X[3 * i], X[3 * i + 1], X[3 * i + 2]};
    double d[12];
    DRpDq(q[0], q + 1, x, d);
    for (int r = 0; r < 3; r++) {
      for (int c = 0; c < 4; c++) {rows.J3[21 * i + 7 * r + c] = d[4 * r + c];}
      rows.J3[21 * i + 7 * r + 4 + r] = 1.;
    }
    const V3 p = Transform(pose, x);
    rows.r3[3 * i] = p.x - Y[3 * i]; rows.r3[3 * i + 1] = p.y - Y[3 * i + 1]; rows.r3[3 * i + 2] = p.z - Y[3 * i + 2];
    errors[i] = rows.r3[3 * i] * rows.r3[3 * i] + rows.r3[3 * i + 1] * rows.r3[3 * i + 1] + rows.r3[3 * i + 2] * rows.r3[3 * i + 2];
  }
  const double scale = ScaleOf(errors);
  std::vector<double> weights(n);
  for (int i = 0; i < n; i++) {weights[i] = HuberDerivativeOf(errors[i] / (scale + 1e-16), 1.345);}
  double M[42], dx[6];
  MakeMOf(q, M);
  WeightedUpdateOf(M, weights, rows, dx);
  AngleAxisToQuaternionOf(dx, dq);
  for (int a = 0; a < 3; a++) {dt[a] = dx[3 + a];}
}

}  // extern "C"
