import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from lidar_feature_extraction_amd import FeatureExtraction, make_scan, concat
batch = 256
clouds = [make_scan(64, 1800, seed=1234 + i, start_col=517) for i in range(16)]
tiled = [clouds[j % 16] for j in range(batch)]
d = torch.from_numpy(concat(tiled).view(np.uint8)).to("cuda:0")
n = np.array([len(c) for c in tiled], np.uint32)
fx = FeatureExtraction(device=0, max_points_per_scan=64 * 1800, max_batch=batch, max_points_per_ring=1800, max_rings=64)
st = torch.cuda.current_stream().cuda_stream
fx.set_profiling(True)
for rep in range(12):
    fx.extract_batch_device(d.data_ptr(), n, st)
    r = fx.scan_routes(batch, st)
    print(rep, np.bincount(r, minlength=3).tolist(), {k.replace("ring_", "").replace("_kernel", ""): (round(v[0], 3), v[1]) for k, v in fx.kernel_times().items() if v[1]})
