"""Diagnostic: where a unit wave's lifetime goes.  Build the kernels with -DLFX_STAMPS
(make -C lidar_feature_extraction_amd/csrc stamps), run this on the GPU box:
    LFX_LIB_PATH=$PWD/lidar_feature_extraction_amd/_lib/liblfx_stamps.so python tools/stamps.py
It runs batches of 64x1800 scans and prints the median shader cycles between consecutive stage
stamps of the unit kernel (units of scan 0).  Not part of the product or the tests."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run from anywhere: the repo root holds the packages
import ctypes as C
import numpy as np
import torch
from lidar_feature_extraction_amd import FeatureExtraction, HyperParameters, make_scan, concat
from lidar_feature_extraction_amd import binding as B

NAMES = ["entry->checks", "boundaries", "A load", "B order+range", "C links+jumps", "D occlusion+reach",
         "E curvature", "F order masks", "F edge pass", "F surface pass", "G parallel beam", "G labels+publish",
         "G look-back", "G label/curv stores", "G feature points"]
batch, rings, cols = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 64, 1800
clouds = [make_scan(rings, cols, seed=1234 + i) for i in range(8)]
tiled = [clouds[i % 8] for i in range(batch)]
d = torch.from_numpy(concat(tiled).view(np.uint8)).cuda()
n = np.array([len(c) for c in tiled], np.uint32)
fx = FeatureExtraction(HyperParameters(), device=0, max_points_per_scan=len(clouds[0]), max_batch=batch,
                       max_points_per_ring=cols, max_rings=rings)
for _ in range(5):
    fx.extract_batch_device(d.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
L = B.load()
L.lfx_debug_read_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
total = 384 * 16
buf = (C.c_ulonglong * total)()
assert L.lfx_debug_read_stamps(buf, total) == total
t = np.frombuffer(buf, dtype=np.uint64).reshape(384, 16).astype(np.int64)
ok = t[:, 14] > t[:, 0]
info = t[ok][:, 15]
polls = info & 0xFFFF
xcc = (info >> 16) & 15
wg = info >> 32
print('XCDs the scan ran on:', sorted(set(xcc.tolist())), ' workgroup indices mod 8:', sorted(set((wg % 8).tolist())))
for u in range(0, 48, 1):
    print('unit %3d wg %6d xcc %d' % (u, wg[u], xcc[u]))
t = t[ok]
print("units stamped:", len(t))
life = t[:, 14] - t[:, 0]
print("lifetime (stamp 0 -> 14): median %d  p10 %d  p90 %d shader cycles" % (np.median(life), np.percentile(life, 10), np.percentile(life, 90)))
for k in range(14):
    dt = t[:, k + 1] - t[:, k]
    print("%-22s median %6d  mean %6d  p90 %6d  (%4.1f %%)" % (NAMES[k + 1], np.median(dt), dt.mean(), np.percentile(dt, 90), 100.0 * dt.mean() / life.mean()))
print("re-polls of the look-back: mean %.2f  max %d  units with any %d" % (polls.mean(), polls.max(), int((polls > 0).sum())))
# the scan's units against the clock: first start, the spread of the moments the counts are published, last end
t0 = t[:, 0].min()
print("starts %d..%d  publishes (stamp 11) %d..%d  ends %d..%d (cycles after the scan's first wave began)" % (
    0, t[:, 0].max() - t0, t[:, 11].min() - t0, t[:, 11].max() - t0, t[:, 14].min() - t0, t[:, 14].max() - t0))
order = np.argsort(np.arange(len(t)))
lb = t[:, 12] - t[:, 11]
for lo in range(0, len(t), 48):
    print("units %3d..%3d  look-back mean %6d  start %6d  publish %6d" % (lo, min(lo + 47, len(t) - 1), lb[lo:lo + 48].mean(),
          (t[lo:lo + 48, 0] - t0).mean(), (t[lo:lo + 48, 11] - t0).mean()))
