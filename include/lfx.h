/*
 * lfx.h -- C ABI of the MI355X-native lidar feature extraction hot path.
 *
 * The reference (tier4/lidar_feature_extraction) has no plugin/FFI boundary: its per-scan
 * extraction is the C++ body of FeatureExtraction::Callback,
 *   /root/reference/extraction/app/feature_extraction.cpp:114-157
 * between GetPointCloud<PointXYZIR> (:94) and ToPointXYZ/ToRosMsg (:161-166).  This header is
 * the drop-in boundary for exactly those lines: PointXYZIR points in (32-byte AoS,
 * lib/include/lidar_feature_library/point_type.hpp:62-86), edge + surface clouds, per-point
 * labels and curvature out.  Plain pointers and sizes only; no C++ or torch types.
 * INTEGRATION.md shows the replacement of lines 114-157 that binds these entry points.
 *
 * Threading: a context is NOT thread-safe; use one context per GPU / per calling thread
 * (the reference's caller is a single-threaded executor, feature_extraction.cpp:185).
 * Every function returns LFX_OK (0) or a negative lfx_error; lfx_last_error() gives text.
 * No exception crosses this boundary: the reference's per-ring std::invalid_argument
 * (feature_extraction.cpp:154-156: warn, ring contributes nothing) becomes ring_status[].
 */
#ifndef LFX_H_
#define LFX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LFX_VERSION 1
#define LFX_MAX_PADDING 63          /* convolution_padding: up to 15 the window kernels (the fast routes); 16 .. 63 every ring takes the
                                    * workgroup-per-ring kernel, whose labelling then walks the positions in reach (slow, same results) */
#define LFX_MAX_RING_ID 65535       /* a ring id is the uint16 of PointXYZIR (point_type.hpp:62-86) */
#define LFX_MAX_RINGS 256           /* distinct ring ids a context takes (every spinning lidar fielded today has fewer) */
#define LFX_MAX_RING_POINTS 4608    /* points of one ring must fit one workgroup's LDS (25 B each); 6 blocks of the unit kernels' long form */

/* The nine node parameters: extraction/include/lidar_feature_extraction/hyper_parameter.hpp:32-65
 * (same names, same units; the neighbour threshold is in DEGREES, converted as
 * lib/include/lidar_feature_library/degree_to_radian.hpp:34-37 does). */
typedef struct lfx_params {
  int32_t padding;                        /* convolution_padding            default 5    */
  double neighbor_degree_threshold;       /*                                default 2.0  */
  double distance_diff_threshold;         /*                                default 0.3  */
  double parallel_beam_min_range_ratio;   /*                                default 0.02 */
  double edge_threshold;                  /*                                default 0.05 */
  double surface_threshold;               /*                                default 0.05 */
  double min_range;                       /*                                default 0.1  */
  double max_range;                       /*                                default 100  */
  int32_t n_blocks;                       /*                                default 6    */
} lfx_params;

/* sensor_msgs/msg/PointField datatype codes (the upstream converter's table,
 * point_type_converter/convert.py:41-53). */
enum lfx_field_type {
  LFX_FIELD_INT8 = 1, LFX_FIELD_UINT8 = 2, LFX_FIELD_INT16 = 3, LFX_FIELD_UINT16 = 4, LFX_FIELD_INT32 = 5,
  LFX_FIELD_UINT32 = 6, LFX_FIELD_FLOAT32 = 7, LFX_FIELD_FLOAT64 = 8
};

/* Where x, y, z (f32) and ring sit inside one point record, as a PointCloud2 describes it.
 * PointXYZIR is {32, 0, 4, 8, 20, LFX_FIELD_UINT16, 0} (point_type.hpp:62-86; convert.py:134-145 of
 * point_type_converter).  ring_datatype: any integer PointField type (0 = UINT16; an Ouster driver
 * publishes UINT8, test_convert.py:42-60); big_endian: the message's is_bigendian flag.  With these
 * the library reads a driver's cloud directly -- the repack the upstream converter node does on the
 * CPU (convert.py:183-212) is folded into the bucketing kernel's loads. */
typedef struct lfx_layout {
  uint32_t point_step, off_x, off_y, off_z, off_ring;
  uint32_t ring_datatype, big_endian;
} lfx_layout;

/* One entry of PointCloud2.fields */
typedef struct lfx_point_field {
  const char *name;
  uint32_t offset;
  uint8_t datatype;               /* lfx_field_type */
  uint32_t count;
} lfx_point_field;

typedef struct lfx_config {
  uint32_t struct_size;           /* sizeof(lfx_config) as the CALLER was compiled with it: fields beyond it (added to the end
                                   * of this struct by a later header) read as zero, so that a caller built against an
                                   * older header keeps working.  0 is refused (an uninitialised struct).               */
  uint32_t max_points_per_scan;   /* capacity of one scan                                   */
  uint32_t max_batch;             /* scans per lfx_extract_batch* call                      */
  uint32_t max_points_per_ring;   /* 0 = LFX_MAX_RING_POINTS; rounded up to a multiple of 64.  The sensor's real
                                   * column count here lets the ring kernel run its smallest (fastest) variant */
  uint32_t max_rings;             /* ring ids are 0 .. max_rings-1 (a sensor's ring count); 0 = 256 */
  uint32_t drop_zero_points;      /* 1: points with x = y = z = 0 are not part of the scan -- the filter the
                                   * upstream converter applies (point_type_converter/convert.py:162-163,192) */
  lfx_layout layout;              /* all-zero = PointXYZIR                                  */
  uint32_t outputs;               /* LFX_OUT_* mask: what lfx_extract / lfx_extract_batch bring back to the host.
                                   * 0 = LFX_OUT_ALL.  The edge / surface clouds (with their index lists) and the ring
                                   * table always come back; labels, curvature and sorted_index are per-point arrays
                                   * (13 bytes per point over PCIe) that the node itself does not consume
                                   * (feature_extraction.cpp:161-170 publishes the two clouds; labels only feed the
                                   * colored_scan debug cloud): a caller that does not need them leaves them out.
                                   * Without LFX_OUT_CURVATURE the per-point curvature is not PRODUCED either (the
                                   * device view's curvature_sorted is NULL; the feature points still carry theirs as
                                   * intensity): 8 of the 9 bytes the kernels write per point, +5 % scans/s          */
  uint32_t stream_hint;           /* LFX_STREAM_*: what the caller knows about the order its driver publishes in.  The
                                   * library finds the route for a stream from what the first batches report (nothing to
                                   * configure); a hint only spares the FIRST batch of a stream the slower route         */
  const uint16_t *ring_ids;       /* the sensor's ring ids where they are not 0 .. max_rings-1 (the reference buckets by
                                   * whatever uint16 a point carries, ring.hpp:114-125): n_ring_ids distinct ids, at most
                                   * LFX_MAX_RINGS of them; results list rings by id ascending.  NULL: ids 0 .. max_rings-1 --
                                   * and the host entry points (lfx_extract*) look the ids of a scan up themselves when a
                                   * point carries another one (lfx_set_ring_ids does the same for the device path)       */
  uint32_t n_ring_ids;
} lfx_config;

#define LFX_STREAM_UNKNOWN 0u      /* start on the organised route, adapt                                               */
#define LFX_STREAM_TURNED_RINGS 1u /* a grid whose rings arrive rotated / reversed (scans not cut at -pi, a clockwise
                                    * sensor): find the rings' transforms from the first batch on                        */
#define LFX_STREAM_NO_GRID 2u      /* records missing or in arbitrary order: the bucketing route from the first batch on */
#define LFX_STREAM_GRID_WITH_HOLES 3u /* a grid whose invalid returns are (0, 0, 0) records, with drop_zero_points set (what the
                                    * reference's converter filters, convert.py:162-163,192): count the valid returns per ring
                                    * first and read the grid in place, from the first batch on                          */

#define LFX_OUT_FEATURES 1u        /* always on */
#define LFX_OUT_LABELS 2u
#define LFX_OUT_CURVATURE 4u
#define LFX_OUT_SORTED_INDEX 8u
#define LFX_OUT_ALL 15u

/* PointLabel values: extraction/include/lidar_feature_extraction/point_label.hpp:32-42 */
enum lfx_label {
  LFX_LABEL_DEFAULT = 0, LFX_LABEL_EDGE = 1, LFX_LABEL_EDGE_NEIGHBOR = 2, LFX_LABEL_SURFACE = 3,
  LFX_LABEL_SURFACE_NEIGHBOR = 4, LFX_LABEL_OUT_OF_RANGE = 5, LFX_LABEL_OCCLUDED = 6,
  LFX_LABEL_PARALLEL_BEAM = 7
};

/* Per-ring outcome.  Non-zero = the ring contributes no label, curvature or feature point,
 * exactly as a ring the reference removes (RemoveSparseRings, ring.cpp:46-59) or abandons on
 * std::invalid_argument (feature_extraction.cpp:154-156). */
enum lfx_ring_status {
  LFX_RING_OK = 0,
  LFX_RING_SPARSE = 1,            /* N < padding+1                 ring.cpp:46-59             */
  LFX_RING_TOO_FEW_CONV = 2,      /* N < 2*padding+1               convolution.cpp:39-43      */
  LFX_RING_TOO_FEW_BLOCKS = 3,    /* N - 2*padding < n_blocks      index_range.cpp:35-40      */
  LFX_RING_BLOCK_TOO_SMALL = 4,   /* a block holds < 2 points      neighbor.hpp:71-75         */
  LFX_RING_ZERO_NORM_PAIR = 5,    /* adjacent points both (0,0)    math.cpp:40-42             */
  LFX_RING_TOO_LARGE = 7          /* N > max_points_per_ring (no reference counterpart)       */
};

enum lfx_error {
  LFX_OK = 0,
  LFX_ERR_INVALID_ARGUMENT = -1,
  LFX_ERR_NO_DEVICE = -2,         /* no HIP device / kernel image: the product has NO CPU fallback */
  LFX_ERR_HIP = -3,
  LFX_ERR_CAPACITY = -4,          /* more points / scans than the context was created for     */
  LFX_ERR_RING_ID = -5,           /* a point carries a ring id the context does not know (lfx_config.ring_ids,
                                   * lfx_set_ring_ids), or a scan more than LFX_MAX_RINGS distinct ones */
  LFX_ERR_OUT_OF_MEMORY = -6,
  LFX_ERR_NO_RING_FIELD = -7,     /* the cloud has no "ring" field: RingIsAvailable (ring.cpp:36-44) is false and the
                                   * node shuts down (feature_extraction.cpp:103-108)            */
  LFX_ERR_UNSUPPORTED_FIELD = -8  /* x / y / z missing or not FLOAT32, ring not an integer, field outside point_step */
};

typedef struct lfx_ctx lfx_ctx;

/* One scan's results in host memory (pinned, owned by the context, valid until its next call). */
typedef struct lfx_scan_result {
  uint32_t n_points;
  const uint8_t *labels;          /* [n_points] lfx_label, addressed by ORIGINAL point index; NULL unless LFX_OUT_LABELS  */
  const double *curvature;        /* [n_points] f64, original index (curvature.cpp:44-50); NULL unless LFX_OUT_CURVATURE  */
  const uint32_t *sorted_index;   /* [n_sorted] rings ascending, angle ascending inside a ring (ring.hpp:141-147); NULL unless LFX_OUT_SORTED_INDEX */
  uint32_t n_sorted;              /* = n_points, less the points drop_zero_points removed                  */
  uint32_t n_rings;
  const uint16_t *ring_id;        /* [n_rings] ascending                                        */
  const uint32_t *ring_count;     /* [n_rings] points of the ring                               */
  const uint32_t *ring_offset;    /* [n_rings] start of the ring inside sorted_index            */
  const uint8_t *ring_status;     /* [n_rings] lfx_ring_status                                  */
  uint32_t n_edge;
  const float *edge_points;       /* [n_edge][4] x, y, z, (float)curvature  (label.hpp:166-179) */
  const uint32_t *edge_index;     /* [n_edge] original point index; ring asc, angle asc         */
  uint32_t n_surface;
  const float *surface_points;    /* [n_surface][4]                                             */
  const uint32_t *surface_index;
} lfx_scan_result;

/* Device-resident results of the last lfx_extract_batch_device call (device pointers owned by
 * the context).  Per-point outputs are RING-MAJOR with a fixed capacity per ring slot (the ring id; with lfx_config.ring_ids /
 * lfx_set_ring_ids the id's rank among the sensor's ids): ring r of
 * scan s owns positions [(s * max_rings + r) * ring_capacity, + ring_count[s][r]) of labels_sorted,
 * curvature_sorted and sorted_index, angle ascending.  The feature clouds are dense: scan s owns the
 * first n_edge / n_surface records from scan_begin[s] (scan_info[s][2], [3]).
 * scan_info[s][1] carries error bits (1: a ring id the context does not know, 4: bucketing timed out -- lfx_batch_status turns
 * them into a return code) and the route the scan took: (bits & LFX_SCAN_ROUTE_MASK) == LFX_SCAN_ORGANISED means the
 * scan arrived column-major with ring == index mod max_rings and was read in place: position k of ring r IS input
 * point k * max_rings + r and sorted_index holds nothing for that scan.  Any other value: sorted_index holds every ring
 * position's original index -- a bucketed scan, or (bit LFX_SCAN_GRID_WITH_HOLES, without LFX_SCAN_ORGANISED) a grid with
 * (0, 0, 0) records that the zero filter dropped, read in place (lfx_scan_routes: 3). */
#define LFX_SCAN_ROUTE_MASK 0x300u
#define LFX_SCAN_ORGANISED 0x100u
#define LFX_SCAN_GRID_WITH_HOLES 0x800u
typedef struct lfx_device_view {
  uint32_t batch;
  uint32_t max_rings;             /* ring ids the layout has room for                            */
  uint32_t ring_capacity;         /* positions per ring (max_points_per_ring rounded up to 64)   */
  const uint32_t *scan_begin;     /* device [batch+1] (in points)                                */
  const uint8_t *labels_sorted;   /* device: label of ring position k                            */
  const double *curvature_sorted; /* device; NULL in a context created without LFX_OUT_CURVATURE */
  const uint32_t *sorted_index;   /* device: original index (within its scan) of ring position k */
  const uint32_t *scan_info;      /* device [batch][4]: occupied rings, error bits, n_edge, n_surface */
  const uint32_t *ring_count;     /* device [batch][256] by ring slot                            */
  const uint8_t *ring_status;     /* device [batch][256] by ring slot (valid where ring_count > 0) */
  const float *edge_points;       /* device [total][4]                                           */
  const uint32_t *edge_index;
  const float *surface_points;
  const uint32_t *surface_index;
} lfx_device_view;

/* --- parameters ------------------------------------------------------------------------ */
void lfx_default_params(lfx_params *p);   /* code defaults, hyper_parameter.hpp:35-43 */
void lfx_launch_params(lfx_params *p);    /* lidar_feature_launch/config/lidar_feature_extraction.param.yaml:3-10 */

/* --- context --------------------------------------------------------------------------- */
/* Replaces the node's construction of HyperParameters / EdgeLabel / SurfaceLabel
 * (feature_extraction.cpp:68-72).  Fails with LFX_ERR_NO_DEVICE when no MI355X is present. */
int lfx_create(lfx_ctx **ctx, int device_id, const lfx_params *params, const lfx_config *config);
void lfx_destroy(lfx_ctx *ctx);
const char *lfx_last_error(const lfx_ctx *ctx);   /* ctx may be NULL: error of the last failed lfx_create */
const char *lfx_status_string(int ring_status);
/* The text the reference's exception carries when it abandons a ring of n_points points for `ring_status` -- what the
 * node logs with RCLCPP_WARN(e.what()), feature_extraction.cpp:154-156: convolution.cpp:40-41, index_range.cpp:36-38,
 * neighbor.hpp:72-73, math.cpp:41.  Empty for LFX_RING_OK, LFX_RING_SPARSE (RemoveSparseRings drops the ring silently,
 * ring.cpp:46-59) and LFX_RING_TOO_LARGE (no counterpart).  Returns the length written (snprintf semantics). */
int lfx_ring_message(int ring_status, uint32_t n_points, const lfx_params *params, char *buf, size_t len);
/* Optional log callback: where the node logs, the library calls back.  For every ring a host-API call (lfx_extract,
 * lfx_extract_batch, lfx_extract_wait) finds abandoned on std::invalid_argument it calls `cb(LFX_LOG_WARN, text, user)` with
 * the text of lfx_ring_message -- the node's RCLCPP_WARN(e.what()), feature_extraction.cpp:154-156, once per ring as there
 * (a ring RemoveSparseRings drops makes no sound, ring.cpp:46-59; LFX_RING_TOO_LARGE, which has no counterpart, is reported
 * with the library's own text).  Called on the caller's thread, inside the call that brought the results; cb = NULL
 * switches it off (the default).  Nothing is ever printed by the library itself. */
#define LFX_LOG_WARN 1
typedef void (*lfx_log_fn)(int level, const char *message, void *user);
int lfx_set_log_callback(lfx_ctx *ctx, lfx_log_fn cb, void *user);
/* RangeMessage* of range_message.hpp:37-83, the texts of the reference's bounds errors ("i (which is 39) >= max (which
 * is 30)"): kind 0 LargerThanOrEqualTo, 1 SmallerThanOrEqualTo, 2 LargerThan, 3 SmallerThan.  Returns the length, -1
 * for an unknown kind. */
int lfx_range_message(int kind, const char *value_name, const char *range_name, long long value, long long range,
                      char *buf, size_t len);

/* --- the operator: feature_extraction.cpp:114-157 ---------------------------------------- */
/* Host points in, host results out (synchronous; H2D + kernels + D2H). */
int lfx_extract(lfx_ctx *ctx, const void *points, size_t n_points, lfx_scan_result *out);
/* The same in two halves, for a caller that keeps the next scan coming while this one is on the device (the node's
 * callback, feature_extraction.cpp:92-171, called by a spinning executor, :185): lfx_extract_submit queues the upload of
 * `points` (on a stream of its own, so that it runs beside the kernels of the scan before), the kernels and the download,
 * and returns at once with a ticket; lfx_extract_wait(ticket) waits for that scan alone and fills `out`.  Two scans may
 * be in flight: a third lfx_extract_submit before the first lfx_extract_wait fails with LFX_ERR_INVALID_ARGUMENT.
 * Tickets are waited for in the order they were issued.  `points`: pageable memory is copied at once and may be reused
 * when lfx_extract_submit returns; pinned memory (lfx_host_alloc) is read by DMA and must stay untouched until that
 * ticket's lfx_extract_wait returns.  `out` and what it points to stay valid until the SECOND lfx_extract_submit after
 * this lfx_extract_wait (each of the two slots has a result block of its own).  Not to be mixed with lfx_extract /
 * lfx_extract_batch / lfx_extract_batch_device while a ticket is outstanding (LFX_ERR_INVALID_ARGUMENT). */
int lfx_extract_submit(lfx_ctx *ctx, const void *points, size_t n_points, uint64_t *ticket);
int lfx_extract_wait(lfx_ctx *ctx, uint64_t ticket, lfx_scan_result *out);
int lfx_extract_batch(lfx_ctx *ctx, const void *const *points, const size_t *n_points, uint32_t batch,
                      lfx_scan_result *out /* [batch] */);

/* Device points in, results stay on the device (asynchronous on `stream`, a hipStream_t or NULL).
 * d_points: the scans' point records back to back; n_points: host array [batch]. */
int lfx_extract_batch_device(lfx_ctx *ctx, const void *d_points, const uint32_t *n_points, uint32_t batch,
                             void *stream);
int lfx_device_results(const lfx_ctx *ctx, lfx_device_view *view);
/* What lfx_extract reports through its return code, for callers of the device-resident path: waits for `stream`,
 * reads the last batch's scan_info and returns LFX_ERR_RING_ID / LFX_ERR_HIP if any scan carries an error bit
 * (the clouds of such a scan are not to be used), else LFX_OK.  first_bad (may be NULL): index of the first such scan. */
int lfx_batch_status(lfx_ctx *ctx, void *stream, uint32_t *first_bad);
/* The sensor's ring ids for a context created without lfx_config.ring_ids (n distinct ids, n <= the context's max_rings;
 * any order: results list rings by id ascending).  ids = NULL: back to 0 .. max_rings-1.  Takes effect with the next
 * batch; ids other than 0 .. n-1 go through the bucketing route (the organised-scan kernel reads ring r at column
 * offset r).  Waits for the context's stream. */
int lfx_set_ring_ids(lfx_ctx *ctx, const uint16_t *ids, uint32_t n);
/* Which route each scan of the last batch took (diagnostics; waits for `stream`): routes[s] = 1 read in place by the
 * organised-scan kernel, 2 the same through per-ring transforms (rings rotated / reversed in the stream), 3 read in place
 * as a grid with (0, 0, 0) records that the zero filter dropped (sorted_index holds its points' indices), 0 bucketed. */
int lfx_scan_routes(lfx_ctx *ctx, void *stream, uint8_t *routes /* [batch] */);

/* Pinned host memory for the caller's point buffers: lfx_extract reads a buffer allocated here by DMA (3.7 MB in
 * ~70 us for a 64 x 1800 scan); any other host pointer is first copied, chunk by chunk, through the context's own
 * pinned staging buffer (CPU memcpy speed).  Free with lfx_host_free before or after lfx_destroy. */
int lfx_host_alloc(lfx_ctx *ctx, size_t bytes, void **out);
void lfx_host_free(lfx_ctx *ctx, void *ptr);

/* --- PointCloud2 on either side of the operator ---------------------------- */
/* The record layout for lfx_config from a message's field list: what pcl::fromROSMsg<PointXYZIR>
 * (ros_msg.hpp:72-78) needs from it -- x, y, z as FLOAT32 -- plus the ring channel the node insists on.
 * Other fields (intensity, time, reflectivity ...) are skipped, as the upstream converter's field filter
 * does (convert.py:113-121).  Returns LFX_ERR_NO_RING_FIELD / LFX_ERR_UNSUPPORTED_FIELD. */
int lfx_layout_from_fields(const lfx_point_field *fields, uint32_t n_fields, uint32_t point_step, int is_bigendian,
                           lfx_layout *out);
/* The last device batch's edge and surface clouds as pcl::PointXYZ wire records (point_step 16:
 * x, y, z, 1.0f -- what ToPointXYZ + toROSMsg publish as scan_edge / scan_surface,
 * feature_extraction.cpp:163-170), packed back to back in scan order exactly as lfx_pack_features
 * does (same offsets table). */
int lfx_pack_xyz(lfx_ctx *ctx, float *d_edge_out, float *d_surface_out, uint32_t *d_offsets_out,
                 size_t capacity_points, void *stream);
/* The same clouds as tight x, y, z triples (12 bytes per point, [capacity_points][3] floats): the least that has
 * to cross xGMI when the clouds of several GPUs are gathered to one (gather.py, bench.py). */
int lfx_pack_xyz12(lfx_ctx *ctx, float *d_edge_out, float *d_surface_out, uint32_t *d_offsets_out,
                   size_t capacity_points, void *stream);
/* colored_scan (feature_extraction.cpp:153,161) of the last device batch as pcl::PointXYZRGB wire records
 * (point_step 32: x, y, z, 1.0f | rgb bit-cast to float, 0, 0, 0; rgb = 0xFF<<24 | r<<16 | g<<8 | b with the
 * table of color_points.cpp:39-68): for every scan the points of each ring that was labelled (status
 * LFX_RING_OK), rings ascending, angle ascending -- the reference appends ring by ring and skips a ring it
 * abandons.  d_offsets_out u32 [batch+1]: exclusive prefix of the per-scan point counts. */
int lfx_pack_colored(lfx_ctx *ctx, float *d_colored_out, uint32_t *d_offsets_out, size_t capacity_points,
                     void *stream);
/* Pack the last device batch's edge and surface clouds back to back, in scan order, into
 * caller-provided DEVICE buffers (what one rank hands to the multi-GPU gather):
 * d_edge_out / d_surface_out [capacity_points][4] floats (16-byte aligned); d_offsets_out u32
 * [2][batch+1]: exclusive prefix of the per-scan edge counts, then of the surface counts
 * (entry [batch] = total).  Asynchronous on `stream`. */
int lfx_pack_features(lfx_ctx *ctx, float *d_edge_out, float *d_surface_out, uint32_t *d_offsets_out,
                      size_t capacity_points, void *stream);
/* Copy scan `scan` of the last device batch to host memory (synchronises the stream). */
int lfx_download_scan(lfx_ctx *ctx, uint32_t scan, void *stream, lfx_scan_result *out);

/* --- multi-GPU: one process per GPU, scans sharded scan i -> rank i mod N, clouds gathered to one rank ------------- */
/* The reference node is stateless per message (feature_extraction.cpp:173-179), so scans are independent units and the
 * only exchange is the gather of the variable-length edge / surface clouds.  It runs over RCCL directly (librccl is
 * opened at the first call; point-to-point send / recv over the direct xGMI links, no ring, no reduction).
 * Bootstrap as with NCCL: rank 0 makes an id (lfx_comm_unique_id), the caller hands it to every rank by whatever means
 * it has (MPI, a socket, torch.distributed ...), every rank calls lfx_comm_create with it. */
#define LFX_COMM_ID_BYTES 128
typedef struct lfx_comm lfx_comm;
int lfx_comm_unique_id(uint8_t id[LFX_COMM_ID_BYTES]);
int lfx_comm_create(lfx_ctx *ctx, const uint8_t id[LFX_COMM_ID_BYTES], int rank, int world, lfx_comm **out);
void lfx_comm_destroy(lfx_comm *comm);
/* Gather of one step, in two halves so that a caller can run it one step behind the extraction (bench.py, gather.py):
 *   lfx_gather_counts   queues on `stream`: the all-gather of this rank's two totals (from d_offsets as written by
 *                       lfx_pack_xyz12 / lfx_pack_xyz / lfx_pack_features: entries [batch] and [2*batch+1]) and their copy
 *                       to pinned host memory.  Returns at once.
 *   lfx_gather_payload  waits (host) for those totals, then queues on `stream` one grouped send / recv: every rank
 *                       sends its first n_edge and n_surface records (floats_per_point floats each: 3 for xyz12, 4
 *                       for xyz / features) and its offsets table to `dst`; dst receives them rank after rank into
 *                       d_edge_all / d_surface_all (capacity_points records each; its own part is a device copy) and
 *                       d_offsets_all [world][2][batch+1].  counts_out (host, [world][2], may be NULL) receives the
 *                       totals of every rank on every rank; rank r's clouds start at the sum of the counts before it.
 *                       capacity_points is the destination's, given by EVERY rank (the same value): if the gathered
 *                       clouds do not fit, every rank returns LFX_ERR_CAPACITY and nothing is sent.
 * lfx_gather = both, one after the other.  Every rank must make the same sequence of calls. */
int lfx_gather_counts(lfx_ctx *ctx, lfx_comm *comm, const uint32_t *d_offsets, uint32_t batch, void *stream);
/* What this rank has posted on the communicator so far: [0] sends, [1] receives, [2] bytes sent, [3] bytes received,
 * [4] all-gathers of totals.  `dst` of lfx_gather_payload may change from call to call (every rank gives the same). */
#define LFX_COMM_STATS 5
int lfx_comm_stats(const lfx_comm *comm, uint64_t out[LFX_COMM_STATS]);
/* Two steps' exchanges as ONE group on ONE communicator: step A's clouds travel to steps[0].dst, step B's to steps[1].dst,
 * every rank posting its sends and receives of both between one ncclGroupStart / ncclGroupEnd -- two of a rank's xGMI links
 * carry data at once (a sender reaches a destination over one link), with no second communicator whose kernels could
 * start in another order on another rank.  Each step's totals come from the slot its lfx_gather_counts_slot call named
 * (LFX_GATHER_SLOTS = 2 may be out at once; lfx_gather_counts = slot 0).  n_steps = 1 or 2; batch, floats_per_point and
 * capacity_points are common to the steps; if either step's clouds do not fit, every rank returns LFX_ERR_CAPACITY and
 * nothing of either is sent.  Every rank must make the same sequence of calls with the same dst and slot values. */
#define LFX_GATHER_SLOTS 2
typedef struct lfx_gather_step
{
  int dst;                          /* destination rank of this step                                              */
  uint32_t slot;                    /* which lfx_gather_counts_slot call carries its totals                        */
  const float *d_edge, *d_surface;  /* this rank's packed clouds of the step (lfx_pack_xyz12 / _xyz / _features)   */
  const uint32_t *d_offsets;
  float *d_edge_all, *d_surface_all;    /* where this rank is dst: capacity_points records each; else may be NULL  */
  uint32_t *d_offsets_all;              /* [world][2][batch+1]                                                      */
  uint64_t *counts_out;                 /* host [world][2] or NULL: every rank's totals of the step                 */
} lfx_gather_step;
int lfx_gather_counts_slot(lfx_ctx *ctx, lfx_comm *comm, uint32_t slot, const uint32_t *d_offsets, uint32_t batch, void *stream);
int lfx_gather_payload2(lfx_ctx *ctx, lfx_comm *comm, const lfx_gather_step *steps, uint32_t n_steps, uint32_t batch,
                        uint32_t floats_per_point, size_t capacity_points, void *stream);
int lfx_gather_payload(lfx_ctx *ctx, lfx_comm *comm, int dst, const float *d_edge, const float *d_surface,
                       const uint32_t *d_offsets, uint32_t batch, uint32_t floats_per_point, float *d_edge_all,
                       float *d_surface_all, uint32_t *d_offsets_all, size_t capacity_points, uint64_t *counts_out,
                       void *stream);
int lfx_gather(lfx_ctx *ctx, lfx_comm *comm, int dst, const float *d_edge, const float *d_surface,
               const uint32_t *d_offsets, uint32_t batch, uint32_t floats_per_point, float *d_edge_all,
               float *d_surface_all, uint32_t *d_offsets_all, size_t capacity_points, uint64_t *counts_out, void *stream);

/* --- voxel-grid Downsample ------------------------------------------------------------------------- */
/* Downsample<T>(cloud, leaf) of lib/include/lidar_feature_library/downsample.hpp:37-51 (pcl::VoxelGrid with one leaf
 * size), which the localizer applies to scan_surface before it builds residuals (localization/.../surface.hpp:111),
 * for a batch of clouds that are already on the device.  Cloud s = d_count[s * count_stride] records of 4 floats
 * (x, y, z, -) from record d_begin[s] of d_points; its downsampled cloud (x, y, z, 1: pcl::PointXYZ) is written from
 * record d_begin[s] of d_out, cells in ascending cell index, d_out_count[s] records; d_status[s] = 1 where PCL gives
 * the cloud back unfiltered because the leaf is too small for its extent (nothing is written then).  PARITY UNPINNED:
 * VoxelGrid's arithmetic is PCL's, a third-party library that is neither under the reference tree nor in this image;
 * implemented from its published algorithm (PCL 1.12.1), points of a cell summed in input order.  Asynchronous. */
int lfx_voxel_downsample(lfx_ctx *ctx, const float *d_points, const uint32_t *d_begin, const uint32_t *d_count,
                         uint32_t count_stride, uint32_t n_clouds, size_t total_points, float leaf, float *d_out,
                         uint32_t *d_out_count, uint32_t *d_status, void *stream);
/* The same for the surface clouds of the last device batch (scan s: the scan's n_surface points): d_out laid out like
 * lfx_device_view::surface_points, d_out_count / d_status [batch]. */
int lfx_downsample_surface(lfx_ctx *ctx, float leaf, float *d_out, uint32_t *d_out_count, uint32_t *d_status, void *stream);

/* --- the map a scan is matched against ---------------------------------------------------------------- */
/* KDTreeEigen (localization/include/lidar_feature_localization/kdtree.hpp:50-63, src/kdtree.cpp:37-68; MakeKDTree :66-71):
 * built once per map, answers exact k-nearest queries.  lfx_map_create copies n_points records of 4 floats (x, y, z, -)
 * from the device into an index of its own: a uniform grid of cubic cells of cell_size (map units; grown if the map's
 * extent would need more than 2^25 cells), the points sorted by cell -- or, with cell_size 0, no grid: every query reads
 * the whole map (small maps; the check of the grid).  Both answer alike: neighbours by ascending squared distance,
 * equal distances by the lower index in the map as given (nanoflann leaves that order open).  Points must be finite.
 * Synchronous on `stream`.  A map belongs to the device of the context that made it and outlives nothing: destroy it
 * before the context's device is reset. */
typedef struct lfx_map lfx_map;
int lfx_map_create(lfx_ctx *ctx, const float *d_points, uint32_t n_points, float cell_size, lfx_map **out, void *stream);
/* The same from host memory (the points are staged through a temporary device buffer). */
int lfx_map_create_host(lfx_ctx *ctx, const float *points, uint32_t n_points, float cell_size, lfx_map **out, void *stream);
void lfx_map_destroy(lfx_map *map);
int lfx_map_info(const lfx_map *map, uint32_t *n_points, float *cell_size /* 0: no grid */, int32_t dims[3]);
/* KDTreeEigen::NearestKSearch (src/kdtree.cpp:44-68) for n_queries queries of 3 doubles on the device: per query the k
 * nearest points of the map -- d_neighbours [n][k][3] doubles (GetRows of the map), d_squared_distances [n][k],
 * d_indices [n][k] into the map as given; any of the three may be NULL.  k <= 16.  Asynchronous. */
int lfx_map_nearest(lfx_ctx *ctx, const lfx_map *map, const double *d_queries, uint32_t n_queries, uint32_t k,
                    double *d_neighbours, double *d_squared_distances, uint32_t *d_indices, void *stream);

/* --- scan-to-map residual build (first slice) --------------------------------------------------------- */
/* What the reference's localizer does first with scan_edge / scan_surface, on clouds that are already on the device:
 *   LFX_RESIDUAL_EDGE     Edge::Make (localization/include/lidar_feature_localization/edge.hpp:86-124): per point the k
 *                         nearest points of the edge map, their mean and principal direction, residual[3] =
 *                         (p - p1) x (p - p2) and the 3 x 7 row [Hat(p2 - p1) DRpDq(q, p0), Hat(p2 - p1)];
 *   LFX_RESIDUAL_SURFACE  Surface::MakeFromDownsampled (surface.hpp:116-139; downsample first: lfx_downsample_surface):
 *                         the plane X w = -1 through the k nearest points of the surface map, residual[1] = signed
 *                         point-plane distance and the 1 x 7 row [u^T DRpDq(q, p), u^T], u = w / |w|.
 * pose: point_to_map as [R | t], row-major 3 x 4 doubles (host); clouds as in lfx_voxel_downsample; outputs addressed
 * like the points (record d_begin[s] + i): d_residual 3 (edge) or 1 (surface) doubles per point, d_jacobian 21 or 7
 * doubles per point, row-major.  n_neighbors <= 16 (the localizer uses 15).  Exact nearest-neighbour search (the
 * reference's nanoflann KD-tree is exact too).  PARITY UNPINNED: Eigen's and nanoflann's arithmetic is not available
 * here; tolerance-level agreement with the CPU restatement, edge rows up to the sign of the principal direction (see
 * DESIGN.md).  Asynchronous. */
#define LFX_RESIDUAL_EDGE 0
#define LFX_RESIDUAL_SURFACE 1
int lfx_scan_to_map_residuals(lfx_ctx *ctx, int kind, const lfx_map *map, const double pose[12], uint32_t n_neighbors,
                              const float *d_points, const uint32_t *d_begin, const uint32_t *d_count,
                              uint32_t count_stride, uint32_t n_clouds, uint32_t max_points_per_cloud,
                              double *d_residual, double *d_jacobian, void *stream);
/* The same for the edge clouds of the last device batch (outputs laid out like lfx_device_view::edge_points). */
int lfx_edge_residuals(lfx_ctx *ctx, const lfx_map *map, const double pose[12], uint32_t n_neighbors,
                       double *d_residual, double *d_jacobian, void *stream);

/* --- the optimizer around those rows (second slice) ---------------------------------------------------- */
/* Optimizer<LOAMOptimizationProblem, EdgeSurfaceScan>::Run (localization/include/lidar_feature_localization/
 * optimizer.hpp:79-123, as Localizer::Update calls it, localizer.hpp:76) for a batch of scans against one pair of maps
 * (lfx_map_create: the reference builds its two KD-trees in the problem's constructor, loam_optimization_problem.hpp:54-60),
 * every scan from its own initial pose, all iterations on the device: per iteration Problem::Make (the two row builds
 * above, edge rows first: loam_optimization_problem.hpp:62-84), ComputeErrors, NormalizeErrorScale (Scale = 1.4826 *
 * median absolute deviation, robust.cpp:36-50), ComputeWeights (HuberDerivative, k = 1.345), WeightedUpdate (sums of
 * J^T J, w J^T J, w J^T r; IsDegenerate(D, 0.1) -> no step; -(M^T A M).llt().solve(M^T b), optimizer.cpp:40-71),
 * q <- q * AngleAxisToQuaternion(dx[0:3]), t <- t + dx[3:6], and the stopping tests in the reference's order (error
 * larger than before, scale larger than before, |dq.vec| and |dt| < 1e-3, max_iter).  A scan that has stopped costs no
 * further work.  The surface clouds are the ones AFTER Downsample (surface.hpp:111; lfx_downsample_surface, leaf 1.0).
 * max_*_points_per_cloud must be at least the longest cloud's count (they size the launches; lfx_localize_batch reads the
 * counts back itself), total_*_points the extent of the point arrays in records (rows are addressed like the points).
 * initial_poses: [n_clouds][12] host doubles ([R | t] row-major); results: [n_clouds], host.  Synchronous on `stream`.
 * PARITY UNPINNED (Eigen / nanoflann / PCL arithmetic underneath; see lfx_scan_to_map_residuals): results agree with the
 * CPU restatement to tolerance, and the restatement passes the reference's own optimizer tests. */
#define LFX_ALIGN_CONVERGED 0      /* "Optimization successfully converged"           success */
#define LFX_ALIGN_LARGER_ERROR 1   /* "The error is larger than previous iteration"   success */
#define LFX_ALIGN_LARGER_SCALE 2   /* "The scale is larger than previous iteration"   success */
#define LFX_ALIGN_MAX_ITERATION 3  /* "The iteration reached the maximum value"       failure */
#define LFX_ALIGN_EMPTY_INPUT 4    /* "The input data is empty"                       failure */
#define LFX_ALIGN_NO_PLANE 5       /* "No surface neighbourhood spans a plane"        failure: not a status of the reference --
                                    * every surface row of the scan was a zero row (its k nearest map points coincide, lie on
                                    * one line or on a plane through the origin: surface.hpp:78-83 has a zero pivot there and
                                    * Eigen's solve() hands back NaN, with which the reference runs to its iteration limit) */
#define LFX_ALIGN_SUCCESS(code) ((code) <= LFX_ALIGN_LARGER_SCALE)
typedef struct lfx_align_result {   /* OptimizationResult, optimization_result.hpp:35-43 */
  double pose[12];                  /* [R | t] row-major 3 x 4 */
  double error;                     /* sum of squared residuals at the last Problem::Make */
  double error_scale;
  int32_t iteration;
  int32_t code;                     /* LFX_ALIGN_*; text: lfx_align_message */
} lfx_align_result;
const char *lfx_align_message(int code);
int lfx_scan_to_map_align(lfx_ctx *ctx, const lfx_map *edge_map, const lfx_map *surface_map, uint32_t n_neighbors, int max_iter,
                          const float *d_edge_points, const uint32_t *d_edge_begin, const uint32_t *d_edge_count,
                          uint32_t edge_count_stride, uint32_t max_edge_points_per_cloud, size_t total_edge_points,
                          const float *d_surface_points, const uint32_t *d_surface_begin, const uint32_t *d_surface_count,
                          uint32_t surface_count_stride, uint32_t max_surface_points_per_cloud, size_t total_surface_points,
                          uint32_t n_clouds, const double *initial_poses, lfx_align_result *results, void *stream);
/* The same loop on AlignmentProblem (localization/src/alignment.cpp:33-78: rows [DRpDq(q, x), I], residual pose * x - y),
 * the problem the reference's optimizer tests run (localization/test/test_optimizer.cpp).  d_source / d_target: records
 * of 3 doubles on the device, cloud s = d_count[s] records from record d_begin[s]. */
int lfx_align_point_pairs(lfx_ctx *ctx, const double *d_source, const double *d_target, const uint32_t *d_begin,
                          const uint32_t *d_count, uint32_t max_points_per_cloud, size_t total_points, uint32_t n_clouds,
                          int max_iter, const double *initial_poses, lfx_align_result *results, void *stream);
/* Localizer::Update (localizer.hpp:71-80) for every scan of the last device batch: Downsample(scan_surface, surface_leaf)
 * (the reference uses 1.0), then lfx_scan_to_map_align on scan_edge and the downsampled cloud; nothing leaves the device
 * but the results. */
int lfx_localize_batch(lfx_ctx *ctx, const lfx_map *edge_map, const lfx_map *surface_map, uint32_t n_neighbors, int max_iter,
                       float surface_leaf, uint32_t n_scans /* = scans of the last batch: sizes initial_poses[n][12], results[n] */,
                       const double *initial_poses, lfx_align_result *results, void *stream);

/* Localizer::Update for one scan whose two clouds are on the host -- the consumer in a process of its own, handed
 * scan_edge / scan_surface as published (records of 4 floats: x, y, z, -; pcl::PointXYZ on the wire): upload, Downsample
 * of the surface cloud, lfx_scan_to_map_align.  Synchronous. */
int lfx_localize_host(lfx_ctx *ctx, const lfx_map *edge_map, const lfx_map *surface_map, uint32_t n_neighbors, int max_iter,
                      float surface_leaf, const float *edge_points, uint32_t n_edge, const float *surface_points,
                      uint32_t n_surface, const double initial_pose[12], lfx_align_result *result, void *stream);

/* --- per-stage entry points (device-backed mirrors of the reference's free functions) ----- */
/* One ring given as angle-sorted x[n], y[n] host arrays; every stage runs the same device
 * routines the fused ring kernel runs.  Optional inputs may be NULL.
 *   groups     replaces the XY neighbour test by NeighborCheckDebug (neighbor.hpp:116-136)
 *   curvature_in  use these values instead of computing them (EdgeLabel/SurfaceLabel tests)
 * flags select what runs; outputs may be NULL. */
#define LFX_STAGE_LABEL 1u          /* AssignLabel              label.hpp:141-164          */
#define LFX_STAGE_OCCLUSION 2u      /* LabelOccludedPoints      occlusion.hpp:81-91        */
#define LFX_STAGE_OUT_OF_RANGE 4u   /* LabelOutOfRange          out_of_range.hpp:36-48     */
#define LFX_STAGE_PARALLEL_BEAM 8u  /* LabelParallelBeamPoints  parallel_beam.hpp:36-51    */
#define LFX_STAGE_SINGLE_BLOCK 16u  /* label the whole array as one block without borders (EdgeLabel::Assign, label.hpp:72-95) */
#define LFX_STAGE_CURVATURE 32u     /* CalcCurvature must succeed (N >= 2P+1)  curvature.cpp:44-50 */
#define LFX_STAGE_ALL 47u           /* what the fused ring kernel runs */
int lfx_stage_ring(lfx_ctx *ctx, const lfx_params *params, uint32_t flags, uint32_t n, const float *x,
                   const float *y, const int32_t *groups, const double *curvature_in,
                   const double *range_in /* use these ranges instead of sqrt(x^2+y^2) (CalcCurvature tests) */,
                   double *range_out /* [n]  Range, range.hpp:45-74 */,
                   double *curvature_out /* [n]  CalcCurvature, curvature.cpp:44-50 */,
                   uint8_t *link_out /* [n-1] IsNeighborXY(i,i+1), neighbor.hpp:44-48 */,
                   uint8_t *labels_out /* [n] */, int32_t *ring_status_out);
/* Convolution1D (convolution.cpp:35-66) for any odd weight; returns LFX_ERR_INVALID_ARGUMENT where the reference throws. */
int lfx_stage_convolution1d(lfx_ctx *ctx, const double *input, uint32_t n, const double *weight, uint32_t m,
                            double *out);
/* ExtractAngleSortedRings (ring.hpp:141-147) alone: per-ring angle-sorted original indices. */
int lfx_stage_ring_projection(lfx_ctx *ctx, const void *points, size_t n_points, uint32_t *sorted_index,
                              uint32_t *n_rings, uint16_t *ring_id /* [256] */, uint32_t *ring_count /* [256] */);

/* --- colored_scan (debug cloud of the node, feature_extraction.cpp:153,161) ---------------- */
/* LabelToColor, extraction/src/color_points.cpp:39-68: rgb of one PointLabel; returns
 * LFX_ERR_INVALID_ARGUMENT for a value that is not a label (the reference throws). */
int lfx_label_to_color(uint8_t label, uint8_t rgb[3]);
/* ColorPointsByLabel (color_points.hpp:60-74) for a whole scan on the host: out[i] = {x, y, z, rgb packed
 * as PCL does (0xFF << 24 | r << 16 | g << 8 | b, bit-cast to float; a = 255 is PointXYZRGB's default)} for input point i, from the labels lfx_extract
 * returned.  points: the scan's records (layout as given to lfx_create); out: n_points * 4 floats. */
int lfx_color_points_by_label(const lfx_ctx *ctx, const void *points, size_t n_points, const uint8_t *labels,
                              float *out);

/* --- the route selection, as a function ---------------------------------------------------- */
/* Which kernels a batch is given depends on what the batches before it reported about the stream (organised scans are read
 * in place, anything else is bucketed first; rotated / reversed rings get their transforms found first; see
 * INTEGRATION.md 3).  This is that decision with nothing around it -- no context, no device: `report` = the counters a
 * batch leaves behind ([0] rings deferred by the first unit pass, [1] repaired after it, [2] sent to the workgroup-per-ring
 * kernel, [3] repaired before it, [4] scans on the fall-back list, [5] whether the organised-scan kernel ran, [6] scans in
 * the batch, [7] scans given up for the angle order of their rings alone, [8] rings found rotated / reversed, [9] whether
 * the transforms were looked for, [10] scans given up for a (0, 0, 0) record alone (zero filter on), [11] whether the holes
 * form ran, [12] ring groups that held such a record then), `report_rings` = rings of that batch (0 = no report yet),
 * `state` in and out = {rings transformed, every scan bucketed, batches until the organised route is tried again, order
 * repair first, grid with holes}, `choice` out = {organised-scan kernel first, with ring transforms, fall-back list entries
 * launched for, one-launch tail, order repair before the first unit pass, rings the second unit pass is launched for, the
 * holes form (count pass first)}.  tests/test_route_choice.py drives it. */
#define LFX_ROUTE_REPORT_WORDS 14
#define LFX_ROUTE_STATE_WORDS 5
#define LFX_ROUTE_CHOICE_WORDS 7
int lfx_route_choice(const uint32_t report[LFX_ROUTE_REPORT_WORDS], uint32_t report_rings, uint32_t state[LFX_ROUTE_STATE_WORDS],
                     int organised_possible, uint32_t batch, uint32_t max_rings, uint32_t choice[LFX_ROUTE_CHOICE_WORDS]);

/* --- measurement ------------------------------------------------------------------------- */
#define LFX_N_KERNELS 12  /* ring_scatter, ring_unit, ring_order, ring_unit (second pass), ring_extract (the bucketing route), ring_totals, feature_compact (compaction), ring_unit_org (organised scans), ring_cut (transforms of rotated / reversed rings), fallback_tail (the organised route's tail), grid_count (valid returns per ring and column piece of a grid with holes), batch_reset */
int lfx_set_profiling(lfx_ctx *ctx, int enabled);
/* Record the events around every n-th batch only (default 1).  The event pairs between the kernels of a batch
 * cost ~7 % of the device-resident throughput at 64x1800x256; sampled, the durations stay live and the cost goes. */
int lfx_set_profiling_interval(lfx_ctx *ctx, uint32_t every_n_batches);
/* Sum of HIP-event durations per kernel since profiling was enabled, and launches counted. */
int lfx_kernel_times(lfx_ctx *ctx, double ms[LFX_N_KERNELS], uint64_t launches[LFX_N_KERNELS]);
const char *lfx_kernel_name(int k);
/* What THIS device gives at this moment, so that a rate measured on it can be told from a rate measured on another box of the
 * same model: copy_gbs = bytes read + bytes written per second by a plain float4 copy of `bytes` bytes (0: 1 GiB), the best
 * of three; clock_mhz = the shader clock a wave sees while every SIMD of the device runs a chain of dependent 32-bit adds
 * (a SIMD-32 takes a wave in two cycles, four waves share it).  Allocates and frees 2 x bytes of device memory; about 10 ms; on `stream` (a hipStream_t, or NULL). */
int lfx_box_calibration(lfx_ctx *ctx, size_t bytes, void *stream, double *copy_gbs, double *clock_mhz);

#ifdef __cplusplus
}
#endif
#endif  /* LFX_H_ */
