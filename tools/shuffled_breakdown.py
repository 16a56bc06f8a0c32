"""Diagnostic: kernel times (total ms, launches) of a shuffled 64 x 1800 stream, 256 scans per step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time
import numpy as np
import torch
from lidar_feature_extraction_amd import FeatureExtraction, make_scan, concat
batch = 256
clouds = [make_scan(64, 1800, seed=1234 + i, shuffle=True) for i in range(8)]
tiled = [clouds[i % 8] for i in range(batch)]
d = torch.from_numpy(concat(tiled).view(np.uint8)).cuda()
n = np.array([len(c) for c in tiled], np.uint32)
fx = FeatureExtraction(device=0, max_points_per_scan=len(clouds[0]), max_batch=batch, max_points_per_ring=1800, max_rings=64)
st = torch.cuda.current_stream().cuda_stream
for _ in range(4):
    fx.extract_batch_device(d.data_ptr(), n, st)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    fx.extract_batch_device(d.data_ptr(), n, st)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
print("ms per step %.3f  scans/s %.0f" % (1e3 * dt, batch / dt))
fx.set_profiling(True)
for _ in range(4):
    fx.extract_batch_device(d.data_ptr(), n, st)
torch.cuda.synchronize()
for k, (ms, cnt) in fx.kernel_times().items():
    if cnt:
        print("%-34s %8.1f us per step  (%d launches per step)" % (k, 1e3 * ms / 4, cnt // 4))
