// extract_scan.cpp -- the reference node's per-scan work without ROS: a PointXYZIR cloud goes through
// lfx::FeatureExtraction::ExtractFeatures (the drop-in for feature_extraction.cpp:114-157) and what the node would
// publish comes back.  Build: see INTEGRATION.md.
//
//   extract_scan                               a built-in synthetic 16 x 900 scan, summary on stdout
//   extract_scan IN OUT RINGS [launch]         IN: raw 32-byte PointXYZIR records (point_type.hpp:62-86); RINGS: the
//                                              sensor's ring count (0 = unknown); "launch": the launch-yaml parameters.
//                                              OUT (little endian): u32 "LFX1", n, n_edge, n_surface, n_rings;
//                                              labels u8[n] padded to 4; curvature f64[n]; edge_index u32[n_edge];
//                                              surface_index u32[n_surface]; edge cloud, surface cloud as 32-byte
//                                              PointXYZIR records; ring table {u16 id, u16 status, u32 count}[n_rings]
// tests/test_cpp_host.py runs the second form in a child process and compares OUT with the CPU oracle.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "lfx.hpp"

namespace
{
std::vector<lfx::PointXYZIR> synthetic()
{
  const int rings = 16, cols = 900;
  std::vector<lfx::PointXYZIR> cloud;
  unsigned int seed = 1234;
  for (int c = 0; c < cols; c++) {
    const double az = -M_PI + 2.0 * M_PI * (c + 0.5) / cols;
    for (int r = 0; r < rings; r++) {
      seed = seed * 1664525u + 1013904223u;
      const double noise = 0.01 * ((seed >> 8) / 16777216.0 - 0.5);
      const double range = ((c / 60) % 2 ? 6.0 : 9.0) + noise;      // walls at two depths: edges + occlusions
      lfx::PointXYZIR p{};
      p.x = static_cast<float>(range * std::cos(az));
      p.y = static_cast<float>(range * std::sin(az));
      p.z = static_cast<float>(range * std::tan((r - 7.5) * 0.035));
      p.pad = 1.0f;
      p.ring = static_cast<std::uint16_t>(r);
      cloud.push_back(p);
    }
  }
  return cloud;
}

template<typename T>
void put(std::FILE * f, const T * p, std::size_t n)
{
  if (n && std::fwrite(p, sizeof(T), n, f) != n) {throw std::runtime_error("short write");}
}
}  // namespace

int main(int argc, char ** argv)
{
  try {
    if (argc < 4) {
      const std::vector<lfx::PointXYZIR> cloud = synthetic();
      lfx::FeatureExtraction extraction(lfx::HyperParameters(), 0, static_cast<std::uint32_t>(cloud.size()), 900, 16);
      const lfx::Features f = extraction.ExtractFeatures(cloud);
      std::printf("points %zu  scan_edge %zu  scan_surface %zu  rings %zu\n",
        cloud.size(), f.edge.size(), f.surface.size(), f.rings.size());
      for (const auto & ring : f.rings) {
        if (ring.status != LFX_RING_OK) {std::printf("ring %u skipped: %s\n", ring.id, lfx_status_string(ring.status));}
      }
      return 0;
    }
    std::FILE * in = std::fopen(argv[1], "rb");
    if (!in) {std::fprintf(stderr, "cannot open %s\n", argv[1]); return 2;}
    std::fseek(in, 0, SEEK_END);
    const long bytes = std::ftell(in);
    std::fseek(in, 0, SEEK_SET);
    const std::size_t n = static_cast<std::size_t>(bytes) / sizeof(lfx::PointXYZIR);
    const std::uint32_t rings = static_cast<std::uint32_t>(std::stoul(argv[3]));
    const lfx::HyperParameters params = (argc > 4 && std::string(argv[4]) == "launch") ? lfx::HyperParameters::LaunchYaml() :
      lfx::HyperParameters();
    lfx::FeatureExtraction extraction(params, 0, static_cast<std::uint32_t>(n ? n : 1), 0, rings);
    lfx::PointXYZIR * cloud = extraction.PinnedPoints(n ? n : 1);          // what GetPointCloud would fill
    if (n && std::fread(cloud, sizeof(lfx::PointXYZIR), n, in) != n) {std::fprintf(stderr, "short read\n"); return 2;}
    std::fclose(in);
    const lfx::Features f = extraction.ExtractFeatures(cloud, n);
    std::FILE * out = std::fopen(argv[2], "wb");
    if (!out) {std::fprintf(stderr, "cannot open %s\n", argv[2]); return 2;}
    const std::uint32_t head[5] = {0x3158464Cu /* "LFX1" */, static_cast<std::uint32_t>(n),
      static_cast<std::uint32_t>(f.edge.size()), static_cast<std::uint32_t>(f.surface.size()),
      static_cast<std::uint32_t>(f.rings.size())};
    put(out, head, 5);
    put(out, f.labels.data(), f.labels.size());
    const std::uint8_t zero[4] = {0, 0, 0, 0};
    put(out, zero, (4 - f.labels.size() % 4) % 4);
    put(out, f.curvature.data(), f.curvature.size());
    put(out, f.edge_index.data(), f.edge_index.size());
    put(out, f.surface_index.data(), f.surface_index.size());
    put(out, f.edge.data(), f.edge.size());
    put(out, f.surface.data(), f.surface.size());
    for (const auto & ring : f.rings) {
      const std::uint16_t a[2] = {ring.id, static_cast<std::uint16_t>(ring.status)};
      put(out, a, 2);
      put(out, &ring.count, 1);
    }
    std::fclose(out);
    std::printf("points %zu  scan_edge %zu  scan_surface %zu  rings %zu\n", n, f.edge.size(), f.surface.size(), f.rings.size());
  } catch (const lfx::Error & e) {
    std::fprintf(stderr, "lfx error %d: %s\n", e.code, e.what());
    return 1;
  } catch (const std::exception & e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
