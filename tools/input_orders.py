#!/usr/bin/env python3
"""Throughput of the device-resident path for input orders other than "every ring arrives angle
sorted": rotated scan start, clockwise sensor, shuffled points.  (bench.py measures the sorted case.)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run from anywhere: the repo root holds the packages
import sys
import time

import numpy as np
import torch

from lidar_feature_extraction_amd import FeatureExtraction, make_scan, concat

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
variants = {"sorted": {}, "rotated": {"start_col": 517}, "reversed": {"reverse": True}, "ragged": {"drop_fraction": 0.05},
            "shuffled": {"shuffle": True}}
for name, kw in variants.items():
    clouds = [make_scan(64, 1800, seed=1234 + i, **kw) for i in range(8)]
    tiled = [clouds[j % 8] for j in range(batch)]
    d = torch.from_numpy(concat(tiled).view(np.uint8)).to("cuda:0")
    n = np.array([len(c) for c in tiled], np.uint32)
    fx = FeatureExtraction(device=0, max_points_per_scan=64 * 1800, max_batch=batch, max_points_per_ring=2048, max_rings=64)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(8):                       # (the library settles on a route from what the first batches report)
        fx.extract_batch_device(d.data_ptr(), n, st)
        torch.cuda.synchronize()
    torch.cuda.synchronize()
    fx.set_profiling(True)
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        fx.extract_batch_device(d.data_ptr(), n, st)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kt = {k: round(1e3 * ms / max(c, 1), 1) for k, (ms, c) in fx.kernel_times().items() if c}
    print("%-9s %9.0f scans/s   %s" % (name, batch * reps / dt, kt))
    sys.stdout.flush()
    fx.close()
