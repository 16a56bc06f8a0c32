// extract_scan.cpp -- the reference node's per-scan work without ROS: build a PointXYZIR cloud,
// call lfx::FeatureExtraction::ExtractFeatures (the drop-in for feature_extraction.cpp:114-157),
// print what the node would publish.  Build: see INTEGRATION.md.
#include <cmath>
#include <cstdio>
#include <vector>

#include "lfx.hpp"

int main()
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
  try {
    lfx::FeatureExtraction extraction(lfx::HyperParameters(), 0, static_cast<std::uint32_t>(cloud.size()));
    const lfx::Features f = extraction.ExtractFeatures(cloud);
    std::printf("points %zu  scan_edge %zu  scan_surface %zu  rings %zu\n",
      cloud.size(), f.edge.size(), f.surface.size(), f.rings.size());
    for (const auto & ring : f.rings) {
      if (ring.status != LFX_RING_OK) {std::printf("ring %u skipped: %s\n", ring.id, lfx_status_string(ring.status));}
    }
  } catch (const lfx::Error & e) {
    std::fprintf(stderr, "lfx error %d: %s\n", e.code, e.what());
    return 1;
  }
  return 0;
}
