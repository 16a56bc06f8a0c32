"""tools/submit_timing.py -- where the pipelined host API spends its time per scan: the host's own work inside
lfx_extract_submit (nothing waits for the device there), lfx_extract_wait, and the loop as bench.py times it."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidar_feature_extraction_amd import FeatureExtraction, HyperParameters, make_scan
from lidar_feature_extraction_amd import binding as LB

clouds = [make_scan(64, 1800, seed=1234 + i) for i in range(8)]
fe = FeatureExtraction(HyperParameters(), device=0, max_points_per_scan=64 * 1800, max_batch=16, max_points_per_ring=1800, max_rings=64,
                       outputs=LB.OUT_FEATURES)
pinned = [fe.pinned_like(c) for c in clouds]
for _ in range(20):
    fe.wait(fe.submit(pinned[0]), raw=True)
n = 400
t_sub, t_wait = 0.0, 0.0
depth = 2                                                       # scans in flight (the API allows two)
t0 = time.perf_counter()
pending = [fe.submit(pinned[j % 8]) for j in range(depth - 1)]
for j in range(depth - 1, n):
    a = time.perf_counter()
    pending.append(fe.submit(pinned[j % 8]))
    b = time.perf_counter()
    fe.wait(pending.pop(0), raw=True)
    c = time.perf_counter()
    t_sub += b - a
    t_wait += c - b
while pending:
    fe.wait(pending.pop(0), raw=True)
total = time.perf_counter() - t0
print("per scan: loop %.1f us, submit %.1f us (host work), wait %.1f us" % (1e6 * total / n, 1e6 * t_sub / (n - 1), 1e6 * t_wait / (n - 1)))
# the submit alone, device idle between (one at a time): the host's share without any back-pressure
t_alone = 0.0
for j in range(100):
    a = time.perf_counter()
    tk = fe.submit(pinned[j % 8])
    t_alone += time.perf_counter() - a
    fe.wait(tk, raw=True)
print("submit alone (device idle): %.1f us" % (1e6 * t_alone / 100))
fe.close()
