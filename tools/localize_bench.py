"""tools/localize_bench.py -- time of lfx_localize_batch (Localizer::Update for a batch of scans) on the
GPU box, beside the CPU restatement (oracle, exhaustive neighbour search, one core) on a few scans.

  python3 tools/localize_bench.py [--rings 64] [--cols 1800] [--batch 64] [--map-scans 40] [--cell 1.0] [--steps 5]

The maps are the edge / surface features of `map-scans` other scans of the synthetic scene, each moved to a pose of its own
along a track (so that the map is larger than one scan's surroundings, as a localizer's map is)."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(rings=64, cols=1800, batch=64, map_scans=40, cell=1.0, steps=5, max_iter=20, cpu_scans=0, whole_map=False, device=0, kd_scans=0):
    """The measurement as a function (bench.py reports it beside the extraction numbers)."""
    return _measure(argparse.Namespace(rings=rings, cols=cols, batch=batch, map_scans=map_scans, cell=cell, steps=steps, max_iter=max_iter,
                                       cpu_scans=cpu_scans, whole_map=whole_map, device=device, kd_scans=kd_scans))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rings", type=int, default=64)
    ap.add_argument("--cols", type=int, default=1800)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--map-scans", type=int, default=40)
    ap.add_argument("--cell", type=float, default=1.0)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--max-iter", type=int, default=20)
    ap.add_argument("--cpu-scans", type=int, default=2)
    ap.add_argument("--whole-map", action="store_true", help="also time maps without a grid")
    ap.add_argument("--kd-scans", type=int, default=2, help="scans of the KD-tree host baseline (scipy.spatial.cKDTree, one thread)")
    a = ap.parse_args()
    a.device = 0
    print(json.dumps(_measure(a)))


def _measure(a):
    import torch
    from lidar_feature_extraction_amd import FeatureExtraction, make_scan, concat
    dev = torch.device("cuda", a.device)
    rng = np.random.default_rng(1)
    fx = FeatureExtraction(device=a.device, max_points_per_scan=a.rings * a.cols, max_batch=max(a.batch, 8), max_points_per_ring=a.cols,
                           max_rings=a.rings)
    stream = torch.cuda.current_stream().cuda_stream

    def features(clouds):
        d = torch.from_numpy(concat(clouds).view(np.uint8).copy()).to(dev)
        fx.extract_batch_device(d.data_ptr(), [len(c) for c in clouds], stream)
        total = sum(len(c) for c in clouds)
        d_e = torch.zeros((total, 4), dtype=torch.float32, device=dev)
        d_s = torch.zeros((total, 4), dtype=torch.float32, device=dev)
        d_o = torch.zeros(2 * len(clouds) + 2, dtype=torch.int32, device=dev)
        fx.pack_xyz(d_e.data_ptr(), d_s.data_ptr(), d_o.data_ptr(), total, stream)
        torch.cuda.synchronize()
        o = d_o.cpu().numpy()
        n = len(clouds)
        eo, so = o[:n + 1], o[n + 1:2 * n + 2]
        e, s = d_e.cpu().numpy(), d_s.cpu().numpy()
        return [e[eo[i]:eo[i + 1]] for i in range(n)], [s[so[i]:so[i + 1]] for i in range(n)]

    # the maps: features of other scans, each shifted along a track
    edge_parts, surf_parts = [], []
    for at in range(0, a.map_scans, 8):
        clouds = [make_scan(a.rings, a.cols, seed=9000 + at + i) for i in range(min(8, a.map_scans - at))]
        e, s = features(clouds)
        for i in range(len(clouds)):
            shift = np.array([3.0 * (at + i), 0.5 * ((at + i) % 5), 0, 0], np.float32)
            edge_parts.append(e[i] + shift)
            surf_parts.append(s[i] + shift)
    edge_map = np.ascontiguousarray(np.concatenate(edge_parts))
    surf_map = np.ascontiguousarray(np.concatenate(surf_parts))
    d_emap, d_smap = torch.from_numpy(edge_map).to(dev), torch.from_numpy(surf_map).to(dev)
    t0 = time.perf_counter()
    emap, smap = fx.make_map(d_emap.data_ptr(), len(edge_map), a.cell, stream), fx.make_map(d_smap.data_ptr(), len(surf_map), a.cell, stream)
    t_build = time.perf_counter() - t0
    # the scans to localize: the first positions of the track, each from a perturbed pose
    clouds = [make_scan(a.rings, a.cols, seed=9000 + (i % a.map_scans)) for i in range(a.batch)]
    d = torch.from_numpy(concat(clouds).view(np.uint8).copy()).to(dev)
    n_points = [len(c) for c in clouds]

    def pose(i):
        th = rng.normal(0, 0.004, 3)
        k = np.linalg.norm(th)
        u = th / k
        K = np.array([[0, -u[2], u[1]], [u[2], 0, -u[0]], [-u[1], u[0], 0]])
        R = np.eye(3) + np.sin(k) * K + (1 - np.cos(k)) * K @ K
        t = np.array([3.0 * (i % a.map_scans), 0.5 * ((i % a.map_scans) % 5), 0.0]) + rng.normal(0, 0.03, 3)
        return np.hstack([R, t.reshape(3, 1)])
    poses = np.stack([pose(i) for i in range(a.batch)])
    out = {"rings": a.rings, "cols": a.cols, "batch": a.batch, "edge_map_points": len(edge_map), "surface_map_points": len(surf_map),
           "cell": a.cell, "map_dims": [emap.info()["dims"], smap.info()["dims"]], "map_build_ms": round(1e3 * t_build, 2), "max_iter": a.max_iter}

    def timed(em, sm, steps):
        fx.extract_batch_device(d.data_ptr(), n_points, stream)
        res = fx.localize_batch(em, sm, poses, 15, a.max_iter, 1.0, stream)          # warm-up
        torch.cuda.synchronize()
        t = []
        for _ in range(steps):
            t0 = time.perf_counter()
            res = fx.localize_batch(em, sm, poses, 15, a.max_iter, 1.0, stream)
            t.append(time.perf_counter() - t0)
        return res, float(np.median(t))
    res, t_grid = timed(emap, smap, a.steps)
    it = np.array([r["iteration"] for r in res])
    out["localize_ms_per_batch"] = round(1e3 * t_grid, 3)
    out["localize_ms_per_scan"] = round(1e3 * t_grid / a.batch, 4)
    out["iterations_mean"] = float(it.mean())
    out["codes"] = {str(c): int(sum(r["code"] == c for r in res)) for c in range(5)}
    out["pose_error_after"] = float(np.median([np.abs(r["pose"][:, 3] - [3.0 * (i % a.map_scans), 0.5 * ((i % a.map_scans) % 5), 0]).max()
                                               for i, r in enumerate(res)]))
    if a.whole_map:
        em0, sm0 = fx.make_map(d_emap.data_ptr(), len(edge_map), 0.0, stream), fx.make_map(d_smap.data_ptr(), len(surf_map), 0.0, stream)
        res0, t_whole = timed(em0, sm0, max(1, a.steps // 3))
        out["whole_map_ms_per_scan"] = round(1e3 * t_whole / a.batch, 4)
        out["whole_map_same_bits"] = all(x["pose"].tobytes() == y["pose"].tobytes() for x, y in zip(res, res0))
    if a.cpu_scans:
        from oracle import binding as OB
        L = OB.lib()
        PD, PF = C.POINTER(C.c_double), C.POINTER(C.c_float)
        e, s = features(clouds[:a.cpu_scans])
        t_cpu, same = [], []
        for i in range(a.cpu_scans):
            pts = np.ascontiguousarray(s[i], np.float32)
            down, n_down = np.zeros_like(pts), C.c_int(0)
            L.orc_voxel_downsample(OB.ptr(pts, PF), len(pts), C.c_float(1.0), OB.ptr(down, PF), C.byref(n_down))
            down = np.ascontiguousarray(down[:n_down.value])
            ee = np.ascontiguousarray(e[i], np.float32)
            po, err, sc, itn, code = np.zeros(12), C.c_double(), C.c_double(), C.c_int(), C.c_int()
            p0 = np.ascontiguousarray(poses[i])
            t0 = time.perf_counter()
            L.orc_loc_optimize_scan(OB.ptr(edge_map, PF), len(edge_map), OB.ptr(surf_map, PF), len(surf_map), 15, OB.ptr(ee, PF), len(ee),
                                    OB.ptr(down, PF), len(down), OB.ptr(p0, PD), a.max_iter, OB.ptr(po, PD), C.byref(err), C.byref(sc),
                                    C.byref(itn), C.byref(code))
            t_cpu.append(time.perf_counter() - t0)
            same.append(float(np.abs(po.reshape(3, 4) - res[i]["pose"]).max()))
        out["cpu_oracle_ms_per_scan"] = round(1e3 * float(np.mean(t_cpu)), 1)
        out["cpu_oracle_note"] = "exhaustive neighbour search, one core; not the reference's KD-tree"
        out["max_pose_difference_to_oracle"] = max(same)
    a.kd_scans = min(getattr(a, "kd_scans", 0), a.batch)
    if a.kd_scans:
        # A KD-tree on the host, the structure the reference searches (kdtree.hpp:50-71 builds a nanoflann tree per map):
        # scipy's cKDTree, one thread, the same queries -- every edge point and every downsampled surface point of a scan,
        # k = 15, once per iteration the device needed.  Only the search: the rows and the solve of Optimizer::Run are not
        # in it, so this is a LOWER bound of the reference's Update on this host.
        from scipy.spatial import cKDTree
        from oracle import binding as OB
        L = OB.lib()
        PF = C.POINTER(C.c_float)
        t0 = time.perf_counter()
        te, ts = cKDTree(edge_map[:, :3].astype(np.float64)), cKDTree(surf_map[:, :3].astype(np.float64))
        t_tree = time.perf_counter() - t0
        e, s = features(clouds[:a.kd_scans])
        t_kd, n_q, visited = [], [], []
        cells_e = _cell_counts(edge_map, a.cell)
        cells_s = _cell_counts(surf_map, a.cell)
        for i in range(a.kd_scans):
            pts = np.ascontiguousarray(s[i], np.float32)
            down, n_down = np.zeros_like(pts), C.c_int(0)
            L.orc_voxel_downsample(OB.ptr(pts, PF), len(pts), C.c_float(1.0), OB.ptr(down, PF), C.byref(n_down))
            R, tr = res[i]["pose"][:, :3], res[i]["pose"][:, 3]
            qe = e[i][:, :3].astype(np.float64) @ R.T + tr
            qs = down[:n_down.value, :3].astype(np.float64) @ R.T + tr
            iters = max(1, int(res[i]["iteration"]) + 1)
            t0 = time.perf_counter()
            for _ in range(iters):
                te.query(qe, k=15, workers=1)
                ts.query(qs, k=15, workers=1)
            t_kd.append(time.perf_counter() - t0)
            n_q.append((len(qe) + len(qs)) * iters)
            visited.append((_cube_points(cells_e, qe, a.cell) + _cube_points(cells_s, qs, a.cell)) * iters)
        out["kdtree_host"] = {"ms_per_scan": round(1e3 * float(np.mean(t_kd)), 2), "tree_build_ms": round(1e3 * t_tree, 1),
                              "queries_per_scan": int(np.mean(n_q)), "cores": 1,
                              "what": "scipy.spatial.cKDTree, k = 15, every edge and downsampled surface point, once per iteration the device "
                                      "needed; the search only (rows and solve not included): a lower bound of the reference's Update"}
        # what the device's search has to read: the map points of the 3 x 3 x 3 cells around each query (the first cube of
        # nearest_in_grid_wave; it grows only where 16 neighbours are not inside it), 16 bytes per point
        out["search_bytes_per_scan"] = int(16 * np.mean(visited))
    emap.close()
    smap.close()
    fx.close()
    return out


def _cell_counts(map_points, cell):
    """Occupancy of the grid cells of a map: {(ix, iy, iz): points}."""
    idx = np.floor(map_points[:, :3].astype(np.float64) / cell).astype(np.int64)
    keys, counts = np.unique(idx, axis=0, return_counts=True)
    return {tuple(k): int(c) for k, c in zip(keys.tolist(), counts.tolist())}


def _cube_points(cells, queries, cell):
    """Map points in the 27 cells around each query, summed over the queries."""
    idx = np.floor(queries / cell).astype(np.int64)
    total = 0
    offs = [(dx, dy, dz) for dx in (-1, 0, 1) for dy in (-1, 0, 1) for dz in (-1, 0, 1)]
    uniq, n = np.unique(idx, axis=0, return_counts=True)
    for k, m in zip(uniq.tolist(), n.tolist()):
        total += m * sum(cells.get((k[0] + o[0], k[1] + o[1], k[2] + o[2]), 0) for o in offs)
    return total


if __name__ == "__main__":
    main()
