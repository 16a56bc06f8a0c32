"""Randomised sweep of the grid search (lfx_map_*): random maps (uniform, clustered, planar, collinear, lattices with ties,
duplicated points), cell sizes over three decades, k in 1..16, queries inside, around and far from the map; every answer
(indices into the map as given, squared distances bit for bit) against an exhaustive search in numpy.  On the GPU box:
    python tools/stress_map.py [N] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def exhaustive(pts4, queries, k, chunk=256):
    m = pts4[:, :3].astype(np.float64)
    idx = np.zeros((len(queries), k), np.uint32)
    dist = np.zeros((len(queries), k))
    order0 = np.arange(len(m))
    for a in range(0, len(queries), chunk):
        q = queries[a:a + chunk]
        dx = m[None, :, 0] - q[:, None, 0]
        dy = m[None, :, 1] - q[:, None, 1]
        dz = m[None, :, 2] - q[:, None, 2]
        d = dx * dx + dy * dy + dz * dz
        for i in range(len(q)):
            o = np.lexsort((order0, d[i]))[:k]
            idx[a + i], dist[a + i] = o, d[i][o]
    return idx, dist


def draw_map(rng):
    kind = rng.choice(["uniform", "clustered", "planar", "collinear", "lattice", "duplicates", "tiny"])
    n = int(rng.integers(20, 6000))
    scale = float(10.0 ** rng.uniform(-1, 2.5))
    if kind == "uniform":
        p = rng.uniform(-1, 1, (n, 3)) * scale
    elif kind == "clustered":
        c = rng.uniform(-1, 1, (int(rng.integers(1, 12)), 3)) * scale
        p = c[rng.integers(0, len(c), n)] + rng.normal(0, scale * 10.0 ** rng.uniform(-3, -0.5), (n, 3))
    elif kind == "planar":
        p = rng.uniform(-1, 1, (n, 3)) * scale * [1, 1, 0]
        p[:, 2] = float(rng.normal(0, scale))
    elif kind == "collinear":
        p = np.outer(rng.uniform(-1, 1, n), rng.normal(0, 1, 3)) * scale
    elif kind == "lattice":
        g = np.stack(np.meshgrid(np.arange(int(rng.integers(2, 14))), np.arange(int(rng.integers(1, 10))), np.arange(int(rng.integers(1, 6))),
                                 indexing="ij"), -1).reshape(-1, 3).astype(np.float64)
        p = g[rng.permutation(len(g))] * float(rng.choice([1.0, 0.5, 2.0]))
    elif kind == "duplicates":
        base = rng.uniform(-1, 1, (max(n // 4, 17), 3)) * scale
        p = base[rng.integers(0, len(base), n)]
    else:
        p = rng.uniform(-1, 1, (int(rng.integers(16, 40)), 3)) * scale
    out = np.zeros((len(p), 4), np.float32)
    out[:, :3] = p
    return str(kind), out


def main():
    import torch
    from lidar_feature_extraction_amd import FeatureExtraction
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    rng = np.random.default_rng(seed)
    fx = FeatureExtraction(device=0, max_points_per_scan=1024, max_batch=1)
    dev = torch.device("cuda", 0)
    seen = {}
    t0 = time.time()
    for case in range(n_cases):
        kind, pts = draw_map(rng)
        seen[kind] = seen.get(kind, 0) + 1
        lo, hi = pts[:, :3].min(0).astype(np.float64), pts[:, :3].max(0).astype(np.float64)
        span = np.maximum(hi - lo, 1e-3)
        nq = int(rng.integers(1, 400))
        queries = np.concatenate([
            pts[rng.integers(0, len(pts), nq), :3].astype(np.float64),
            pts[rng.integers(0, len(pts), nq), :3].astype(np.float64) + rng.normal(0, 0.05, (nq, 3)) * span,
            rng.uniform(lo - span, hi + span, (nq, 3)),
            (lo + hi) / 2 + rng.normal(0, 1, (3, 3)) * span * 1e3])
        k = int(rng.integers(1, min(16, len(pts)) + 1))
        want_idx, want_dist = exhaustive(pts, queries, k)
        d_pts = torch.from_numpy(pts).to(dev)
        d_q = torch.from_numpy(np.ascontiguousarray(queries)).to(dev)
        for cell in (float(10.0 ** rng.uniform(-2, 2)) * float(span.max()) / 10.0, float(span.max()) * 3.0):
            m = fx.make_map(d_pts.data_ptr(), len(pts), cell)
            d_d = torch.zeros((len(queries), k), dtype=torch.float64, device=dev)
            d_i = torch.zeros((len(queries), k), dtype=torch.int32, device=dev)
            m.nearest(d_q.data_ptr(), len(queries), k, 0, d_d.data_ptr(), d_i.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            idx, dist = d_i.cpu().numpy().view(np.uint32), d_d.cpu().numpy()
            info = m.info()
            m.close()
            assert np.array_equal(idx, want_idx), "case %d (%s, n %d, k %d, cell %g -> %g, dims %s): indices differ at query %s" % (
                case, kind, len(pts), k, cell, info["cell_size"], info["dims"], np.nonzero((idx != want_idx).any(1))[0][:5])
            assert dist.tobytes() == want_dist.tobytes(), "case %d: distances differ" % case
        if case % 50 == 49:
            print("%d cases ok, %.0f s" % (case + 1, time.time() - t0), flush=True)
    print("all %d cases ok in %.0f s (seed %d; maps drawn: %s)" % (n_cases, time.time() - t0, seed, seen))
    fx.close()


if __name__ == "__main__":
    main()
