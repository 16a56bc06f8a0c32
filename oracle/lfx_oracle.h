/*
 * lfx_oracle.h -- CPU oracle for the per-scan feature-extraction hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a CPU restatement of the reference algorithm
 * (tier4/lidar_feature_extraction, extraction/ package).  It exists to CHECK the HIP
 * product path and to be timed as the `cpu_baseline` leg of bench.py.  Nothing in the
 * product (lidar_feature_extraction_amd/, include/) may include, link, import or call
 * anything in this directory; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg do.
 *
 * Parity pinning: every function below is checked against the known-answer vectors of
 * the reference's own unit tests (the .cpp files of extraction/test/, restated as data in
 * tests/golden/reference_unit_vectors.json) and, where the reference source compiles
 * from its own files with what this image holds (math.cpp, convolution.cpp,
 * index_range.cpp), against that compiled reference code (oracle/_ref, see Makefile).
 * The full per-scan path (label.hpp, fill.hpp, occlusion.hpp, ring.hpp ...) needs PCL,
 * range-v3, boost and rclcpp, which this image lacks: it is unbuildable here, so the
 * whole-scan behaviour is pinned by the unit vectors of each stage, not by a run of the
 * reference node.
 *
 * All citations are file:line relative to /root/reference/.
 */
#ifndef LFX_ORACLE_H_
#define LFX_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* extraction/include/lidar_feature_extraction/hyper_parameter.hpp:32-65 */
typedef struct orc_params {
  int32_t padding;                       /* convolution_padding            */
  double neighbor_degree_threshold;      /* degrees                        */
  double distance_diff_threshold;
  double parallel_beam_min_range_ratio;
  double edge_threshold;
  double surface_threshold;
  double min_range;
  double max_range;
  int32_t n_blocks;
} orc_params;

/* extraction/include/lidar_feature_extraction/point_label.hpp:32-42 */
enum {
  ORC_LABEL_DEFAULT = 0, ORC_LABEL_EDGE = 1, ORC_LABEL_EDGE_NEIGHBOR = 2,
  ORC_LABEL_SURFACE = 3, ORC_LABEL_SURFACE_NEIGHBOR = 4, ORC_LABEL_OUT_OF_RANGE = 5,
  ORC_LABEL_OCCLUDED = 6, ORC_LABEL_PARALLEL_BEAM = 7
};

/* per-ring outcome; every non-zero value = "ring contributes nothing"
 * (extraction/app/feature_extraction.cpp:116,126,154-156) */
enum {
  ORC_RING_OK = 0,
  ORC_RING_SPARSE = 1,            /* N < padding+1   ring.cpp:46-59 (RemoveSparseRings)       */
  ORC_RING_TOO_FEW_CONV = 2,      /* N < 2P+1        convolution.cpp:39-43                    */
  ORC_RING_TOO_FEW_BLOCKS = 3,    /* N-2P < n_blocks index_range.cpp:35-40                    */
  ORC_RING_BLOCK_TOO_SMALL = 4,   /* a block slice has < 2 points  neighbor.hpp:71-75         */
  ORC_RING_ZERO_NORM_PAIR = 5,    /* adjacent pair both (0,0) in xy  math.cpp:40-42           */
  ORC_RING_OTHER = 6
};

/* ---- stage level (each mirrors one reference function; used for the unit vectors) ---- */
double orc_xy_norm(double x, double y);                                   /* math.hpp:36-39 */
int orc_calc_radian(double x1, double y1, double x2, double y2, double *out); /* math.cpp:34-46; returns 1 when the reference throws */
double orc_inner_product(const double *a, const double *b, int n);        /* math.hpp:43-53 */
int orc_convolution1d(const double *input, int n, const double *weight, int m, double *out); /* convolution.cpp:35-66; 1 = throws */
void orc_make_weight(int padding, double *out /* 2P+1 */);                /* curvature.cpp:36-42 */
int orc_calc_curvature(const double *range, int n, int padding, double *out); /* curvature.cpp:44-50; 1 = throws */
void orc_argsort(const double *values, int n, int *out);                  /* algorithm.hpp:65-71 */
int orc_index_range(int start, int end, int n_blocks, int *bounds /* n_blocks+1 */); /* index_range.cpp:32-66; 1 = throws */
int orc_padded_index_range(int size, int n_blocks, int padding, int *bounds); /* index_range.hpp:59-66 */
int orc_polar_less_f64(double ax, double ay, double bx, double by);      /* ring.hpp:54-99 with double fields (test_ring.cpp) */
int orc_polar_less_f32(float ax, float ay, float bx, float by);          /* ring.hpp:54-99 with float fields (PointXYZIR) */
void orc_sort_by_atan2_f64(const double *x, const double *y, int n, int *indices /* in/out, n */); /* ring.hpp:101-112 */
int orc_is_neighbor_xy(float x1, float y1, float x2, float y2, double radian_threshold, int *out); /* neighbor.hpp:44-48 */
int orc_is_in_inclusive_range(double v, double min, double max);         /* range.hpp:40-43 */

/* Neighbour checker handed to the fill / label stages: either the debug checker
 * (groups != NULL; neighbor.hpp:116-136) or the XY checker over float points
 * (neighbor.hpp:64-114).  Return value of every stage: 0 ok, 1 the reference throws
 * std::invalid_argument, 2 the reference throws std::out_of_range. */
int orc_fill_from_left(uint8_t *labels, int n, const int *groups, const float *x, const float *y,
                       double radian_threshold, int begin, int end, uint8_t label);   /* fill.hpp:40-68 */
int orc_fill_from_right(uint8_t *labels, int n, const int *groups, const float *x, const float *y,
                        double radian_threshold, int begin, int end, uint8_t label);  /* fill.hpp:70-99 */
int orc_fill_neighbors(uint8_t *labels, int n, const int *groups, const float *x, const float *y,
                       double radian_threshold, int index, int padding, uint8_t label); /* fill.hpp:101-117 */
int orc_edge_label_assign(uint8_t *labels, const double *curvature, int n, const int *groups,
                          const float *x, const float *y, double radian_threshold,
                          int padding, double threshold);                             /* label.hpp:61-100 */
int orc_surface_label_assign(uint8_t *labels, const double *curvature, int n, const int *groups,
                             const float *x, const float *y, double radian_threshold,
                             int padding, double threshold);                          /* label.hpp:102-139 */
int orc_assign_label(uint8_t *labels, const double *curvature, int n, const float *x, const float *y,
                     double radian_threshold, int n_blocks, int padding,
                     double edge_threshold, double surface_threshold);                /* label.hpp:141-164 */
int orc_label_occluded(uint8_t *labels, int n, const float *x, const float *y, double radian_threshold,
                       int padding, double distance_diff_threshold);                  /* occlusion.hpp:37-91 */
void orc_label_out_of_range(uint8_t *labels, int n, const float *x, const float *y,
                            double min_range, double max_range);                      /* out_of_range.hpp:36-48 */
void orc_label_parallel_beam(uint8_t *labels, int n, const float *x, const float *y,
                             double range_ratio_threshold);                           /* parallel_beam.hpp:36-51 */
/* range_message.hpp:37-83: kind 0 LargerThanOrEqualTo, 1 SmallerThanOrEqualTo, 2 LargerThan, 3 SmallerThan; returns the length */
int orc_range_message(int kind, const char *value_name, const char *range_name, long long value, long long range, char *buf, size_t len);
void orc_irange(int size, int *out);                                                  /* iterator.cpp:33-36 */
/* MappedPoints (mapped_points.hpp:40-72) over a cloud given by its y values: at(i) of Slice(begin, end) of the view; 1 = out of range */
int orc_mapped_points_at(const double *cloud_y, int n_cloud, const int *indices, int n_indices, int begin, int end, int i, double *out, int *size);
/* what() of the std::invalid_argument a ring is abandoned with (status = ORC_RING_*), for a ring of n points; returns the length */
int orc_ring_message(int status, int n, const orc_params *p, char *buf, size_t len);
/* Voxel-grid Downsample (lib/include/lidar_feature_library/downsample.hpp:37-51 = pcl::VoxelGrid<T> with one leaf size,
 * applied to the surface scan at localization/include/lidar_feature_localization/surface.hpp:111).  PARITY UNPINNED: the
 * arithmetic is PCL's (third party, not under /root/reference, not in this image; ROS 2 Humble ships PCL 1.12.1); this
 * restates the published algorithm of filters/include/pcl/filters/impl/voxel_grid.hpp (applyFilter): float bounds,
 * inverse leaf size as float, cell index = floor(x * inv) - min cell, cells in ascending linear index, centroid =
 * float sum / count.  PCL leaves the order of the float sum inside a cell to an unstable sort; here (and in the HIP
 * path) it is ascending input index.  points / out: records of 4 floats (x, y, z, 1).  Returns 0, or 1 where PCL gives
 * up (leaf too small for the cloud's extent: index would overflow int). */
int orc_voxel_downsample(const float *points, int n, float leaf, float *out /* capacity n */, int *n_out);
/* --- localization: the scan-to-map residual build that consumes the two clouds; lfx_oracle_loc.cpp.
 * PARITY UNPINNED beyond the vectors of localization/test/test_edge.cpp, test_math.cpp (Eigen + nanoflann underneath). */
void orc_loc_triplet_cross(const double *p0, const double *p1, const double *p2, double *out);          /* edge.cpp:51-57 */
void orc_loc_mean_cov(const double *X /* [n][3] */, int n, double *mean /* 3 */, double *cov /* 9 */);   /* edge.cpp:38-49 */
void orc_loc_principal(const double *cov, double *eigenvalues /* ascending */, double *eigenvectors /* columns, row-major */); /* edge.cpp:59-64 */
int orc_loc_principal_is_reliable(const double *eigenvalues);                                           /* edge.cpp:92-96 */
void orc_loc_solve_linear(const double *A, int rows, int cols, const double *b, double *x);             /* math.hpp:36-40 */
void orc_loc_quaternion(const double *R /* row-major */, double *wxyz);                                 /* Eigen::Quaterniond(R) */
void orc_loc_edge_residuals(const float *map, int n_map, const double *pose /* [R|t] 3x4 */, int k, const float *points,
                            int n, double *residual /* [n][3] */, double *jacobian /* [n][3][7] */);     /* edge.hpp:86-124 */
void orc_loc_surface_residuals(const float *map, int n_map, const double *pose, int k, const float *points, int n,
                               double *residual /* [n] */, double *jacobian /* [n][7] */);              /* surface.hpp:116-139 */
/* the optimizer around those rows (optimizer.hpp:70-127, src/optimizer.cpp, robust.cpp, degenerate.cpp, posevec.cpp,
 * lib/src/stats.cpp, alignment.cpp); pinned by test_robust.cpp, test_degenerate.cpp, test_posevec.cpp, test_optimizer.cpp.
 * code: 0 converged, 1 error larger than before, 2 scale larger than before (success), 3 maximum iteration, 4 empty input. */
void orc_loc_nearest(const float *map, int n_map, const double *query, int k, double *neighbours /* [k][3] */,
                     double *squared_distances, int *indices);                                           /* kdtree.cpp:44-68 */
void orc_loc_drp_dq(const double *wxyz, const double *p, double *out /* 3 x 4 row-major */);          /* rotationlib jacobian/quaternion.cpp:35-52 */
double orc_loc_median(const double *v, int n);                                                          /* stats.cpp:34-55 */
double orc_loc_mad(const double *v, int n);                                                             /* robust.cpp:36-40 */
double orc_loc_scale(const double *v, int n);                                                           /* robust.cpp:42-50 */
double orc_loc_huber(double e, double k);                                                               /* robust.cpp:52-59 */
double orc_loc_huber_derivative(double e, double k);                                                    /* robust.cpp:61-68 */
int orc_loc_is_degenerate(const double *C /* [n][n] */, int n, double threshold);                       /* degenerate.cpp:32-37 */
void orc_loc_angle_axis_to_quaternion(const double *theta, double *wxyz);                               /* posevec.cpp:32-45 */
void orc_loc_rotation_matrix(const double *wxyz, double *R /* row-major */);                            /* Eigen toRotationMatrix */
void orc_loc_make_m(const double *wxyz, double *M /* [7][6] */);                                        /* optimizer.cpp:73-84 */
void orc_loc_pairs_update(const double *X, const double *Y, int n, const double *pose, double *dq /* wxyz */, double *dt);
int orc_loc_optimize_pairs(const double *X /* [n][3] */, const double *Y, int n, const double *initial_pose, int max_iter,
                           double *pose_out, double *error_out, double *scale_out, int *iteration_out, int *code_out);
int orc_loc_optimize_scan(const float *edge_map, int n_edge_map, const float *surface_map, int n_surface_map, int k,
                          const float *edge_points, int n_edge, const float *surface_points, int n_surface,
                          const double *initial_pose, int max_iter, double *pose_out, double *error_out, double *scale_out,
                          int *iteration_out, int *code_out);
void orc_label_to_color(uint8_t label, uint8_t rgb[3]);                               /* color_points.cpp:39-68 */

/* ---- whole scan: the body of FeatureExtraction::Callback, feature_extraction.cpp:114-157 ----
 * Input: n points, `stride` bytes apart; x,y,z f32 and ring u16 at the given byte offsets
 * (PointXYZIR: stride 32, x0 y4 z8 ring20; lib/.../point_type.hpp:62-86).
 * Outputs (caller allocated, any may be NULL):
 *   labels[n], curvature[n]      addressed by ORIGINAL point index (Default / 0.0 for skipped rings)
 *   sorted_index[n]              rings ascending, each ring angle-sorted (ring.hpp:141-147)
 *   ring_id/ring_count/ring_status[max_rings]  ascending ring id; *n_rings entries
 *   edge_index/surface_index[n]  original indices, canonical order (ring asc, angle asc)
 *   edge_points/surface_points   4 floats per point: x, y, z, (float)curvature (label.hpp:166-179)
 *   ties[2]                      [0] adjacent pairs the angle predicate calls equal, [1] adjacent
 *                                equal curvatures met inside a block argsort (unspecified order
 *                                in the reference: std::sort is unstable)
 * canonical_ties != 0: break both kinds of tie by the lower index first (what the HIP path
 * defines); 0: leave them to std::sort exactly as the reference does.
 * Returns 0, or -1 on bad arguments. */
int orc_extract(const void *points, size_t n, size_t stride, size_t off_x, size_t off_y, size_t off_z,
                size_t off_ring, const orc_params *params, int canonical_ties,
                uint8_t *labels, double *curvature, int32_t *sorted_index,
                int32_t *ring_id, int32_t *ring_count, int32_t *ring_status, int32_t max_rings,
                int32_t *n_rings, int32_t *edge_index, int32_t *n_edge, int32_t *surface_index,
                int32_t *n_surface, float *edge_points, float *surface_points, int64_t *ties);

#ifdef __cplusplus
}
#endif
#endif  /* LFX_ORACLE_H_ */
