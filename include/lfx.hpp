// lfx.hpp -- C++ host side over the C ABI (lfx.h): what a maintainer of the reference node calls.
//
// The reference operator is the body of FeatureExtraction::Callback,
// /root/reference/extraction/app/feature_extraction.cpp:114-157; this header gives it a name,
// lfx::FeatureExtraction::ExtractFeatures(cloud), with the reference's types: PointXYZIR in
// (lib/include/lidar_feature_library/point_type.hpp:62-86), edge and surface PointXYZIR clouds out
// (intensity = (float)curvature, label.hpp:166-179), plus the per-point labels and curvature.
// Header only; link with liblfx.so.  Errors of the C ABI become lfx::Error; per-ring conditions
// the reference reports with RCLCPP_WARN (feature_extraction.cpp:154-156) are in ring_status.
#ifndef LFX_HPP_
#define LFX_HPP_

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "lfx.h"

namespace lfx
{

struct alignas(16) PointXYZIR   // same 32-byte layout as the reference's PCL point type
{
  float x, y, z, pad;
  float intensity;
  std::uint16_t ring;
  std::uint8_t reserved[10];
};
static_assert(sizeof(PointXYZIR) == 32, "PointXYZIR must be 32 bytes");

struct Error : std::runtime_error
{
  Error(int c, const std::string & what)
  : std::runtime_error(what), code(c) {}
  int code;
};

struct HyperParameters : lfx_params   // hyper_parameter.hpp:32-65
{
  HyperParameters() {lfx_default_params(this);}
  static HyperParameters LaunchYaml() {HyperParameters p; lfx_launch_params(&p); return p;}
};

struct RingInfo
{
  std::uint16_t id;
  std::uint32_t count, offset;
  lfx_ring_status status;
};

struct Features
{
  std::vector<PointXYZIR> edge, surface;        // rings ascending, angle ascending inside a ring
  std::vector<std::uint32_t> edge_index, surface_index;   // original point indices
  std::vector<std::uint8_t> labels;             // PointLabel per input point (point_label.hpp:32-42)
  std::vector<double> curvature;                // per input point
  std::vector<std::uint32_t> sorted_index;      // ExtractAngleSortedRings, ring.hpp:141-147
  std::vector<RingInfo> rings;
};

class FeatureExtraction
{
public:
  // max_rings: the sensor's ring count (ring ids 0 .. max_rings-1).  Given, a driver's column-major scan is read in
  // place by the organised-scan kernel (no bucketing pass); 0 = unknown (256 ring ids, every scan is bucketed).
  // outputs: LFX_OUT_* mask of the per-point arrays to bring back besides the two clouds; what the node publishes
  // (feature_extraction.cpp:161-170) needs none of them, its colored_scan debug cloud needs LFX_OUT_LABELS.
  explicit FeatureExtraction(
    const HyperParameters & params = HyperParameters(), int device = 0,
    std::uint32_t max_points_per_scan = 262144, std::uint32_t max_points_per_ring = 0,
    std::uint32_t max_rings = 0, std::uint32_t outputs = LFX_OUT_ALL)
  {
    lfx_config cfg{};
    cfg.struct_size = sizeof(cfg);
    cfg.max_points_per_scan = max_points_per_scan;
    cfg.max_batch = 1;
    cfg.max_points_per_ring = max_points_per_ring;
    cfg.max_rings = max_rings;
    cfg.outputs = outputs;
    const int rc = lfx_create(&ctx_, device, &params, &cfg);
    if (rc != LFX_OK) {throw Error(rc, lfx_last_error(nullptr));}
  }
  ~FeatureExtraction()
  {
    for (void * p : pinned_) {lfx_host_free(ctx_, p);}
    lfx_destroy(ctx_);
  }
  FeatureExtraction(const FeatureExtraction &) = delete;
  FeatureExtraction & operator=(const FeatureExtraction &) = delete;

  // The sensor's ring ids where they are not 0 .. rings-1 (the reference buckets by whatever uint16 a point carries,
  // ring.hpp:114-125).  ExtractFeatures finds them by itself; a caller that knows them spares the first scan a second run.
  void SetRingIds(const std::vector<std::uint16_t> & ids) const
  {
    const int rc = lfx_set_ring_ids(ctx_, ids.empty() ? nullptr : ids.data(), static_cast<std::uint32_t>(ids.size()));
    if (rc != LFX_OK) {throw Error(rc, lfx_last_error(ctx_));}
  }

  // A point buffer in pinned host memory, owned by this object: lfx_extract reads it by DMA (a buffer from anywhere
  // else is first copied through the context's staging buffer).  Let GetPointCloud fill it.
  PointXYZIR * PinnedPoints(std::size_t capacity)
  {
    void * p = nullptr;
    const int rc = lfx_host_alloc(ctx_, capacity * sizeof(PointXYZIR), &p);
    if (rc != LFX_OK) {throw Error(rc, lfx_last_error(ctx_));}
    pinned_.push_back(p);
    return static_cast<PointXYZIR *>(p);
  }

  // feature_extraction.cpp:114-157 for one cloud, results left where the library put them (pinned host memory owned by
  // the context, valid until the next call): no copies.
  lfx_scan_result ExtractFeaturesView(const PointXYZIR * points, std::size_t n) const
  {
    lfx_scan_result r{};
    const int rc = lfx_extract(ctx_, points, n, &r);
    if (rc != LFX_OK) {throw Error(rc, lfx_last_error(ctx_));}
    return r;
  }

  // The pipelined pair (lfx_extract_submit / lfx_extract_wait): Submit returns at once with a ticket, Wait gives that
  // scan's view.  Two scans may be in flight; the upload of one runs beside the kernels of the one before.  For the
  // node: Submit the cloud that just arrived, Wait for (and publish) the one submitted by the previous callback.
  std::uint64_t Submit(const PointXYZIR * points, std::size_t n) const
  {
    std::uint64_t ticket = 0;
    const int rc = lfx_extract_submit(ctx_, points, n, &ticket);
    if (rc != LFX_OK) {throw Error(rc, lfx_last_error(ctx_));}
    return ticket;
  }
  lfx_scan_result Wait(std::uint64_t ticket) const
  {
    lfx_scan_result r{};
    const int rc = lfx_extract_wait(ctx_, ticket, &r);
    if (rc != LFX_OK) {throw Error(rc, lfx_last_error(ctx_));}
    return r;
  }

  // The same, copied into containers the caller keeps.
  Features ExtractFeatures(const PointXYZIR * points, std::size_t n) const
  {
    const lfx_scan_result r = ExtractFeaturesView(points, n);
    Features f;
    if (r.labels) {f.labels.assign(r.labels, r.labels + r.n_points);}
    if (r.curvature) {f.curvature.assign(r.curvature, r.curvature + r.n_points);}
    if (r.sorted_index) {f.sorted_index.assign(r.sorted_index, r.sorted_index + r.n_sorted);}
    f.edge_index.assign(r.edge_index, r.edge_index + r.n_edge);
    f.surface_index.assign(r.surface_index, r.surface_index + r.n_surface);
    fill(f.edge, r.edge_points, r.edge_index, r.n_edge, points);
    fill(f.surface, r.surface_points, r.surface_index, r.n_surface, points);
    for (std::uint32_t k = 0; k < r.n_rings; k++) {
      f.rings.push_back(
        RingInfo{r.ring_id[k], r.ring_count[k], r.ring_offset[k], static_cast<lfx_ring_status>(r.ring_status[k])});
    }
    return f;
  }
  Features ExtractFeatures(const std::vector<PointXYZIR> & cloud) const
  {
    return ExtractFeatures(cloud.data(), cloud.size());
  }
  // colored_scan of the node (feature_extraction.cpp:153): x, y, z + packed rgb per input point.
  std::vector<float> ColorPointsByLabel(const PointXYZIR * points, std::size_t n, const Features & f) const
  {
    std::vector<float> out(4 * n);
    const int rc = lfx_color_points_by_label(ctx_, points, n, f.labels.data(), out.data());
    if (rc != LFX_OK) {throw Error(rc, "invalid label");}
    return out;
  }
  lfx_ctx * handle() const {return ctx_;}

private:
  static void fill(
    std::vector<PointXYZIR> & out, const float * pts, const std::uint32_t * idx, std::uint32_t n,
    const PointXYZIR * in)
  {
    out.resize(n);
    for (std::uint32_t k = 0; k < n; k++) {
      PointXYZIR q{};
      q.x = pts[4 * k]; q.y = pts[4 * k + 1]; q.z = pts[4 * k + 2]; q.pad = 1.0f;
      q.intensity = pts[4 * k + 3];            // (float)curvature, label.hpp:176
      q.ring = in[idx[k]].ring;
      out[k] = q;
    }
  }
  lfx_ctx * ctx_ = nullptr;
  std::vector<void *> pinned_;
};

// The consumer of the two clouds: Localizer of the reference's localization package (localization/include/
// lidar_feature_localization/localizer.hpp:48-95) -- maps built once (there: two KD-trees in the problem's constructor),
// Init(pose), Update(scan) -> success, Get() -> pose.  Poses are [R | t], row-major 3 x 4.
class Localizer
{
public:
  // edge_map / surface_map: records of 4 floats (x, y, z, -) on the host; cell_size: the grid behind the nearest-neighbour
  // search (lfx_map_create).  `fx` must outlive the localizer.
  Localizer(
    const FeatureExtraction & fx, const std::vector<float> & edge_map, const std::vector<float> & surface_map,
    int max_iter = 20, float cell_size = 1.0f)
  : ctx_(fx.handle()), max_iter_(max_iter)
  {
    int rc = lfx_map_create_host(ctx_, edge_map.data(), static_cast<std::uint32_t>(edge_map.size() / 4), cell_size, &edge_, nullptr);
    if (rc == LFX_OK) {
      rc = lfx_map_create_host(ctx_, surface_map.data(), static_cast<std::uint32_t>(surface_map.size() / 4), cell_size, &surface_, nullptr);
    }
    if (rc != LFX_OK) {
      lfx_map_destroy(edge_);
      throw Error(rc, lfx_last_error(ctx_));
    }
    const double identity[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    for (int i = 0; i < 12; i++) {last_.pose[i] = identity[i];}
  }
  ~Localizer() {lfx_map_destroy(edge_); lfx_map_destroy(surface_);}
  Localizer(const Localizer &) = delete;
  Localizer & operator=(const Localizer &) = delete;

  void Init(const double pose[12])
  {
    for (int i = 0; i < 12; i++) {last_.pose[i] = pose[i];}
    initialized_ = true;
  }
  bool IsInitialized() const {return initialized_;}
  const double * Get() const {return last_.pose;}
  const lfx_align_result & Result() const {return last_;}      // OptimizationResult of the last Update

  // Update with the scan the FeatureExtraction was last given (its clouds are still on the device: nothing is copied)
  bool Update()
  {
    lfx_align_result r{};
    const int rc = lfx_localize_batch(ctx_, edge_, surface_, kNeighbors, max_iter_, 1.0f, 1u, last_.pose, &r, nullptr);
    if (rc != LFX_OK) {throw Error(rc, lfx_last_error(ctx_));}
    last_ = r;
    return LFX_ALIGN_SUCCESS(r.code);
  }
  // Update with clouds received from elsewhere (scan_edge / scan_surface as published: 4 floats per point)
  bool Update(const float * edge, std::uint32_t n_edge, const float * surface, std::uint32_t n_surface)
  {
    lfx_align_result r{};
    const int rc = lfx_localize_host(ctx_, edge_, surface_, kNeighbors, max_iter_, 1.0f, edge, n_edge, surface, n_surface, last_.pose, &r, nullptr);
    if (rc != LFX_OK) {throw Error(rc, lfx_last_error(ctx_));}
    last_ = r;
    return LFX_ALIGN_SUCCESS(r.code);
  }

private:
  static constexpr std::uint32_t kNeighbors = 15;               // N_NEIGHBORS, localizer.hpp:46
  lfx_ctx * ctx_;
  int max_iter_;
  lfx_map * edge_ = nullptr, * surface_ = nullptr;
  lfx_align_result last_{};
  bool initialized_ = false;
};

}  // namespace lfx
#endif  // LFX_HPP_
