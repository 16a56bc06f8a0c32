#!/usr/bin/env python3
"""PCIe-inclusive rate of the synchronous host API (lfx_extract_batch: pageable host buffers in,
host results out, incl. densify + un-permute).  Not bench.py's `value`; recorded in DESIGN.md."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run from anywhere: the repo root holds the packages
import sys
import time

from lidar_feature_extraction_amd import FeatureExtraction, make_scan

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
clouds = [make_scan(64, 1800, seed=1234 + i) for i in range(batch)]
fx = FeatureExtraction(device=0, max_points_per_scan=64 * 1800, max_batch=batch, max_points_per_ring=2048, max_rings=64)
fx.extract_batch(clouds)
t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    out = fx.extract_batch(clouds)
dt = time.perf_counter() - t0
print("host-in/host-out: %.1f scans/s (%.3f ms/scan), batch %d, %d edge + %d surface in scan 0" % (
    batch * reps / dt, 1e3 * dt / (batch * reps), batch, len(out[0].edge_index), len(out[0].surface_index)))
fx.close()
