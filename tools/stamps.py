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

# (round 5, rows path: stage D runs with the order masks, after stage E -- the slot between stamps 4 and 5 holds only a wave
# sync; the labels' way back to the chunk form belongs to the surface pass's slot)
NAMES = ["entry->checks", "boundaries", "A load", "B range", "C order+links+jumps+range+beam", "(sync)",
         "E curvature", "F order masks + D occlusion+reach", "F edge pass", "F surface pass + labels word", "G labels+curvature+records"]
def _opt(name, default):
    return int(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


batch, rings, cols = _opt("--batch", 256), _opt("--rings", 64), _opt("--cols", 1800)      # (the stamped scan is LFX_STAMP_SCAN = 128: batch > 128)
clouds = [make_scan(rings, cols, seed=1234 + i, vfov_deg=22.5 if rings >= 128 else 15.0) for i in range(8)]
# --holes FRACTION: that share of the returns written as (0, 0, 0) records, zero filter on: the HOLES form of the kernel
holes = float(sys.argv[sys.argv.index("--holes") + 1]) if "--holes" in sys.argv else 0.0
if holes > 0.0:
    for j, c in enumerate(clouds):
        gone = np.random.Generator(np.random.PCG64(99 + j)).uniform(0.0, 1.0, len(c)) < holes
        for f in ("x", "y", "z"):
            c[f][gone] = 0.0
tiled = [clouds[i % 8] for i in range(batch)]
d = torch.from_numpy(concat(tiled).view(np.uint8)).cuda()
n = np.array([len(c) for c in tiled], np.uint32)
fx = FeatureExtraction(HyperParameters(), device=0, max_points_per_scan=len(clouds[0]), max_batch=batch,
                       max_points_per_ring=cols, max_rings=rings, drop_zero_points=holes > 0.0,
                       stream_hint=B.STREAM_GRID_WITH_HOLES if holes > 0.0 else 0)
for _ in range(5):
    fx.extract_batch_device(d.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
L = B.load()
L.lfx_debug_read_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
total = 384 * 16
buf = (C.c_ulonglong * total)()
assert L.lfx_debug_read_stamps(buf, total) == total
t = np.frombuffer(buf, dtype=np.uint64).reshape(384, 16).astype(np.int64)
ok = t[:, 10] > t[:, 0]
t = t[ok]
print("units stamped:", len(t))
if "--skew" in sys.argv:
    # REFCLK (100 MHz, one counter for the chip: slots 11 and 12) at the start and the end of every unit of the scan
    t0, t1 = t[:, 11], t[:, 12]
    first = t0.min()
    print("REFCLK ticks of 10 ns; the scan's units: first start 0, last start %d, first end %d, last end %d" % (t0.max() - first, t1.min() - first, t1.max() - first))
    wait_all = t1.max() - t1
    print("an end that waited for the scan's LAST unit would wait: median %.2f us, p90 %.2f us, max %.2f us" % (np.median(wait_all) / 100.0, np.percentile(wait_all, 90) / 100.0, wait_all.max() / 100.0))
    # ... for every unit before it in the order rings ascending, blocks ascending (the units stamped are in that order)
    run = np.maximum.accumulate(t1)
    before = np.concatenate([[t1[0]], run[:-1]])
    wait_pre = np.maximum(before - t1, 0)
    print("... for the units BEFORE it (rings ascending, blocks ascending): median %.2f us, p90 %.2f us, max %.2f us" % (np.median(wait_pre) / 100.0, np.percentile(wait_pre, 90) / 100.0, wait_pre.max() / 100.0))
    print("unit lifetime by the same clock: median %.2f us" % (np.median(t1 - t0) / 100.0))
    byx = {}
    for u in range(len(t)):
        byx.setdefault((u // 6 // 4) % 8, []).append(t1[u] - first)
    print("mean end per ring group mod 8 (the XCDs, before the groups are turned): " + " ".join("%d:%.1f" % (k, np.mean(v) / 100.0) for k, v in sorted(byx.items())))
if holes > 0.0:
    # the holes form's stage A: 0 = head loads asked for, 13 = geometry and piece range known (the prefix rows have arrived),
    # 14 = records loaded and scattered into the slabs, 1 = hand-over barrier passed
    for a, b, what in ((0, 13, "head: prefix rows + geometry + piece range"), (13, 14, "record loads + scatter"), (14, 1, "hand-over barrier")):
        dt = t[:, b] - t[:, a]
        print("%-44s median %6d  p90 %6d" % (what, np.median(dt), np.percentile(dt, 90)))
life = t[:, 10] - t[:, 0]
print("lifetime (stamp 0 -> 10): median %d  p10 %d  p90 %d shader cycles" % (np.median(life), np.percentile(life, 10), np.percentile(life, 90)))
stages = {}
for k in range(10):
    dt = t[:, k + 1] - t[:, k]
    print("%-36s median %6d  p90 %6d  (%4.1f %%)" % (NAMES[k + 1], np.median(dt), np.percentile(dt, 90), 100.0 * np.median(dt) / np.median(life)))
    stages[NAMES[k + 1]] = {"median": int(np.median(dt)), "p90": int(np.percentile(dt, 90))}
if "--json" in sys.argv:
    import json
    out = sys.argv[sys.argv.index("--json") + 1]
    json.dump({"note": "shader-clock stamps (s_memtime) at the stage boundaries of ring_unit_org_kernel<0, 5, false>, the %d units of one 64 x 1800 scan in a "
                       "batch of %d, diagnostic build (make stamps); cycles between consecutive stamps" % (len(t), batch),
               "units": int(len(t)), "life": {"median": int(np.median(life)), "p10": int(np.percentile(life, 10)), "p90": int(np.percentile(life, 90))},
               "stages": stages}, open(out, "w"), indent=1)
if "--by-ring" in sys.argv:
    # a unit's stamps sit at index ring * B + j (B = 6 blocks): life and the two pick passes per group of four rings, and per
    # XCD as the grid's x index lands on them (ring group mod 8)
    B_ = 6
    full = np.frombuffer(buf, dtype=np.uint64).reshape(384, 16).astype(np.int64)
    ring = np.arange(384) // B_
    lifef = full[:, 10] - full[:, 0]
    picks = full[:, 9] - full[:, 7]
    print("group  life(mean)  picks(mean)  picks(max)")
    for g in range(16):
        m = (ring // 4 == g) & (full[:, 10] > full[:, 0])
        print("%5d  %10.0f  %11.0f  %10d" % (g, lifef[m].mean(), picks[m].mean(), picks[m].max()))
    for x in range(8):
        m = ((ring // 4) % 8 == x) & (full[:, 10] > full[:, 0])
        print("xcd %d: life sum %d" % (x, lifef[m].sum()))
    print("stage means per group (columns: stages 0->1 ... 9->10), then start time of the group's first unit relative to the scan's first")
    t0 = full[full[:, 0] > 0, 0].min()
    for g in range(16):
        m = (ring // 4 == g) & (full[:, 10] > full[:, 0])
        print("%2d " % g + " ".join("%6.0f" % (full[m, k + 1] - full[m, k]).mean() for k in range(10)) + "   start %d" % (full[m, 0].min() - t0))
