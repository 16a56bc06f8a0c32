"""Randomised sweep of the whole consumer chain on the device (extraction -> Downsample -> map search -> residual rows ->
optimizer: lfx_localize_batch) against the CPU restatement chain (oracle extract -> orc_voxel_downsample ->
orc_loc_optimize_scan): random sensor shapes, maps made of the features of 2-4 other scans, random start poses, leaf and cell
sizes, neighbour counts and iteration limits.  Same stopping reason and iteration and the pose to 1e-6 -- or, where a
stopping test sits on a tie of two successive errors, within one iteration and 2e-3 (counted).  On the GPU box:
    python tools/stress_localize.py [N] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_align_gpu import _oracle_scan, _downsample, _pose  # noqa: E402


def main():
    import torch
    from lidar_feature_extraction_amd import FeatureExtraction, make_scan, concat
    from oracle import binding as OB
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 13
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda", 0)
    t0 = time.time()
    scans_done, ties, codes = 0, 0, {}
    for case in range(n_cases):
        rings = int(rng.choice([8, 16, 32]))
        cols = int(rng.integers(300, 1100))
        batch = int(rng.integers(1, 5))
        base = int(rng.integers(1, 1 << 20))
        clouds = [make_scan(rings, cols, seed=base + s) for s in range(batch)]
        want = [OB.extract(c, canonical_ties=False) for c in clouds]
        maps = [OB.extract(make_scan(rings, cols, seed=base + 100 + s), canonical_ties=False) for s in range(int(rng.integers(2, 5)))]
        edge_map = np.ascontiguousarray(np.concatenate([m["edge_points"] for m in maps]), np.float32)
        surf_map = np.ascontiguousarray(np.concatenate([m["surface_points"] for m in maps]), np.float32)
        k = int(rng.integers(5, 16))
        if len(edge_map) < k or len(surf_map) < k:
            continue
        max_iter = int(rng.choice([1, 3, 8, 20]))
        leaf = float(rng.choice([0.5, 1.0, 2.0]))
        cell_e, cell_s = float(10.0 ** rng.uniform(-0.7, 0.7)), float(10.0 ** rng.uniform(-0.7, 0.7))
        fx = FeatureExtraction(device=0, max_points_per_scan=rings * cols, max_batch=batch, max_points_per_ring=cols, max_rings=rings)
        d = torch.from_numpy(concat(clouds).view(np.uint8).copy()).to(dev)
        stream = torch.cuda.current_stream().cuda_stream
        fx.extract_batch_device(d.data_ptr(), [len(c) for c in clouds], stream)
        d_emap, d_smap = torch.from_numpy(edge_map).to(dev), torch.from_numpy(surf_map).to(dev)
        emap, smap = fx.make_map(d_emap.data_ptr(), len(edge_map), cell_e, stream), fx.make_map(d_smap.data_ptr(), len(surf_map), cell_s, stream)
        amp = float(10.0 ** rng.uniform(-3.5, -1.5))
        poses = np.stack([_pose(rng.normal(0, amp, 3), rng.normal(0, 8 * amp, 3)) for _ in range(batch)])
        got = fx.localize_batch(emap, smap, poses, k, max_iter, leaf, stream)
        for s in range(batch):
            w = _oracle_scan(edge_map, surf_map, k, want[s]["edge_points"], _downsample(want[s]["surface_points"], leaf), poses[s], max_iter)
            g = got[s]
            scans_done += 1
            codes[g["code"]] = codes.get(g["code"], 0) + 1
            what = "case %d scan %d (%dx%d, k %d, max_iter %d, leaf %g, cells %.2f %.2f)" % (case, s, rings, cols, k, max_iter, leaf, cell_e, cell_s)
            if np.isnan(w["pose"]).any() or np.isnan(g["pose"]).any():
                assert np.isnan(w["pose"]).any() == np.isnan(g["pose"]).any(), (what, g, w)
                continue
            if (g["code"], g["iteration"]) == (w["code"], w["iteration"]):
                assert np.abs(g["pose"] - w["pose"]).max() <= 1e-6 * (1 + np.abs(w["pose"]).max()), (what, g, w)
            else:
                ties += 1
                assert abs(g["iteration"] - w["iteration"]) <= 1 and g["success"] == w["success"], (what, g, w)
                assert np.abs(g["pose"] - w["pose"]).max() < 2e-3, (what, g, w)
        emap.close()
        smap.close()
        fx.close()
        if case % 10 == 9:
            print("%d cases, %d scans ok (%d at a tie), %.0f s" % (case + 1, scans_done, ties, time.time() - t0), flush=True)
    print("all %d scans of %d cases ok in %.0f s (seed %d; stopping reasons %s; %d at a tie)" % (
        scans_done, n_cases, time.time() - t0, seed, dict(sorted(codes.items())), ties))


if __name__ == "__main__":
    main()
