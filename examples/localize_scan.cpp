// localize_scan.cpp -- the consumer's side without ROS: lfx::Localizer (the drop-in for the reference's Localizer,
// localization/include/lidar_feature_localization/localizer.hpp:48-95) over the two clouds of a scan.
//
//   localize_scan EDGE_MAP SURFACE_MAP SCAN RINGS COLS OUT [host]
//     EDGE_MAP, SURFACE_MAP   raw records of 4 floats (x, y, z, -)
//     SCAN                    raw 32-byte PointXYZIR records (point_type.hpp:62-86)
//     OUT                     two results, each 12 doubles pose [R | t], error, error_scale (doubles), iteration, code
//                             (int32): first Update() on the extraction's own device clouds, then (with "host") Update on
//                             the clouds as a separate consumer would receive them
//   The initial pose is a fixed small offset from the identity; tests/test_cpp_host.py compares with the CPU oracle.
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#include "lfx.hpp"

namespace
{
template<typename T>
std::vector<T> slurp(const char * path)
{
  std::FILE * f = std::fopen(path, "rb");
  if (!f) {throw std::runtime_error(std::string("cannot open ") + path);}
  std::fseek(f, 0, SEEK_END);
  const long bytes = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  std::vector<T> v(static_cast<std::size_t>(bytes) / sizeof(T));
  if (!v.empty() && std::fread(v.data(), sizeof(T), v.size(), f) != v.size()) {std::fclose(f); throw std::runtime_error("short read");}
  std::fclose(f);
  return v;
}

void put(std::FILE * f, const lfx_align_result & r)
{
  std::fwrite(r.pose, sizeof(double), 12, f);
  std::fwrite(&r.error, sizeof(double), 1, f);
  std::fwrite(&r.error_scale, sizeof(double), 1, f);
  std::fwrite(&r.iteration, sizeof(std::int32_t), 1, f);
  std::fwrite(&r.code, sizeof(std::int32_t), 1, f);
}
}  // namespace

int main(int argc, char ** argv)
{
  try {
    if (argc < 7) {
      std::fprintf(stderr, "usage: localize_scan EDGE_MAP SURFACE_MAP SCAN RINGS COLS OUT [host]\n");
      return 2;
    }
    const std::vector<float> edge_map = slurp<float>(argv[1]), surface_map = slurp<float>(argv[2]);
    const std::vector<lfx::PointXYZIR> cloud = slurp<lfx::PointXYZIR>(argv[3]);
    const std::uint32_t rings = static_cast<std::uint32_t>(std::stoul(argv[4])), cols = static_cast<std::uint32_t>(std::stoul(argv[5]));
    lfx::FeatureExtraction extraction(lfx::HyperParameters(), 0, static_cast<std::uint32_t>(cloud.size()), cols, rings, 0);
    lfx::Localizer localizer(extraction, edge_map, surface_map, 20, 1.0f);
    const double initial[12] = {1, 0, 0, 0.02, 0, 1, 0, -0.015, 0, 0, 1, 0.01};
    std::FILE * out = std::fopen(argv[6], "wb");
    if (!out) {throw std::runtime_error("cannot open the output file");}
    // the node's order of things: features of the scan, then the pose from them
    const lfx_scan_result view = extraction.ExtractFeaturesView(cloud.data(), cloud.size());
    localizer.Init(initial);
    const bool ok = localizer.Update();
    put(out, localizer.Result());
    std::printf("update %s: iteration %d, %s\n", ok ? "succeeded" : "failed", localizer.Result().iteration, lfx_align_message(localizer.Result().code));
    if (argc > 7 && std::string(argv[7]) == "host") {
      // scan_edge / scan_surface as a separate consumer gets them: x, y, z of the feature points, 4 floats per point
      std::vector<float> edge(view.edge_points, view.edge_points + 4 * static_cast<std::size_t>(view.n_edge));
      std::vector<float> surface(view.surface_points, view.surface_points + 4 * static_cast<std::size_t>(view.n_surface));
      localizer.Init(initial);
      localizer.Update(edge.data(), view.n_edge, surface.data(), view.n_surface);
      put(out, localizer.Result());
    }
    std::fclose(out);
    return 0;
  } catch (const std::exception & e) {
    std::fprintf(stderr, "localize_scan: %s\n", e.what());
    return 1;
  }
}
