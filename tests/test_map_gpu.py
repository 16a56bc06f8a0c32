"""lfx_map (the place of the reference's KDTreeEigen, localization/include/lidar_feature_localization/kdtree.hpp:50-71):
the vectors of localization/test/test_kdtree.cpp through the device, and the grid index against an exhaustive search in
numpy -- the k nearest points, ascending squared distance, equal distances by the lower map index -- bit for bit: the
distances are plain double arithmetic on both sides."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _exhaustive(points, queries, k):
    m = points[:, :3].astype(np.float64)
    idx = np.zeros((len(queries), k), np.uint32)
    dist = np.zeros((len(queries), k))
    for i, q in enumerate(queries):
        dx, dy, dz = m[:, 0] - q[0], m[:, 1] - q[1], m[:, 2] - q[2]
        d = dx * dx + dy * dy + dz * dz
        order = np.lexsort((np.arange(len(m)), d))[:k]
        idx[i], dist[i] = order, d[order]
    return idx, dist


def _query(fx, scan_map, queries, k):
    import torch
    dev = torch.device("cuda", 0)
    n = len(queries)
    d_q = torch.from_numpy(np.ascontiguousarray(queries, np.float64)).to(dev)
    d_x = torch.zeros((n, k, 3), dtype=torch.float64, device=dev)
    d_d = torch.zeros((n, k), dtype=torch.float64, device=dev)
    d_i = torch.zeros((n, k), dtype=torch.int32, device=dev)
    scan_map.nearest(d_q.data_ptr(), n, k, d_x.data_ptr(), d_d.data_ptr(), d_i.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return d_x.cpu().numpy(), d_d.cpu().numpy(), d_i.cpu().numpy().view(np.uint32)


def test_reference_kdtree_vectors_through_the_device(refvec):
    import torch
    from lidar_feature_extraction_amd import FeatureExtraction
    g = refvec["loc_kdtree"]
    pts = np.zeros((len(g["points"]), 4), np.float32)
    pts[:, :3] = g["points"]
    fx = FeatureExtraction(device=0, max_points_per_scan=1024, max_batch=1)
    d_pts = torch.from_numpy(pts).to("cuda:0")
    for cell in (0.0, 1.0, 0.25, 100.0):
        m = fx.make_map(d_pts.data_ptr(), len(pts), cell)
        assert m.info()["n_points"] == 4 and (m.info()["cell_size"] > 0) == (cell > 0)
        for c in g["cases"]:
            X, dist, _ = _query(fx, m, [g["query"]], c["k"])
            assert X[0].tolist() == c["X"] and dist[0].tolist() == c["squared_distances"]      # EXPECT_EQ(norm, 0), ElementsAre
        m.close()
    fx.close()


@pytest.mark.parametrize("case", ["clustered", "lattice-ties", "flat", "huge-extent", "single-cell"])
def test_grid_against_exhaustive_search(case):
    import torch
    from lidar_feature_extraction_amd import FeatureExtraction
    rng = np.random.default_rng({"clustered": 1, "lattice-ties": 2, "flat": 3, "huge-extent": 4, "single-cell": 5}[case])
    if case == "clustered":
        centres = rng.uniform(-60, 60, (40, 3)) * [1, 1, 0.1]
        pts = np.concatenate([c + rng.normal(0, rng.uniform(0.2, 3), (rng.integers(20, 800), 3)) for c in centres])
        cells = [1.0, 0.3, 4.0]
    elif case == "lattice-ties":                      # many equal distances, duplicates of whole points
        g = np.stack(np.meshgrid(np.arange(12), np.arange(9), np.arange(5), indexing="ij"), -1).reshape(-1, 3).astype(np.float64)
        pts = np.concatenate([g, g[::7], g[::11]])
        pts = pts[rng.permutation(len(pts))]
        cells = [1.0, 0.5, 2.5]
    elif case == "flat":                              # a plane: one layer of cells
        pts = np.concatenate([rng.uniform(-80, 80, (6000, 2)), np.zeros((6000, 1))], 1)
        cells = [1.0, 5.0]
    elif case == "huge-extent":                       # the asked cell size would need far more than 2^25 cells: it grows
        pts = np.concatenate([rng.uniform(-4000, 4000, (5000, 3)), rng.normal(0, 1, (3000, 3))])
        cells = [0.5]
    else:
        pts = rng.uniform(0, 1, (300, 3))
        cells = [50.0]
    pts4 = np.zeros((len(pts), 4), np.float32)
    pts4[:, :3] = pts
    lo, hi = pts4[:, :3].min(0).astype(np.float64), pts4[:, :3].max(0).astype(np.float64)
    span = hi - lo
    queries = np.concatenate([
        pts4[rng.choice(len(pts4), 300), :3].astype(np.float64),                     # on map points (distance 0, ties with duplicates)
        pts4[rng.choice(len(pts4), 300), :3].astype(np.float64) + rng.normal(0, 0.4, (300, 3)),
        rng.uniform(lo - 0.2 * span - 1, hi + 0.2 * span + 1, (300, 3)),            # anywhere in and around the box
        lo + span * rng.uniform(0, 1, (20, 3)) + [[1e4, 0, 0]],                      # far outside
        np.array([lo, hi, (lo + hi) / 2, lo - 1e-3, hi + 1e-3])])                    # corners of the grid
    if case == "lattice-ties":
        queries = np.concatenate([queries, np.array([[5.5, 4.5, 2.5], [5.0, 4.0, 2.0], [0.5, 0.5, 0.5], [-3.0, 4.0, 2.0]])])
    fx = FeatureExtraction(device=0, max_points_per_scan=1024, max_batch=1)
    d_pts = torch.from_numpy(pts4).to("cuda:0")
    for k in (15, 1, 16):
        want_idx, want_dist = _exhaustive(pts4, queries, k)
        whole = fx.make_map(d_pts.data_ptr(), len(pts4), 0.0)
        X, dist, idx = _query(fx, whole, queries, k)
        assert np.array_equal(idx, want_idx) and dist.tobytes() == want_dist.tobytes()
        whole.close()
        for cell in cells:
            m = fx.make_map(d_pts.data_ptr(), len(pts4), cell)
            info = m.info()
            assert np.prod(np.array(info["dims"], np.float64)) <= 2 ** 25 and info["cell_size"] >= np.float32(cell)
            if case == "huge-extent":
                assert info["cell_size"] > cell
            X, dist, idx = _query(fx, m, queries, k)
            bad = np.nonzero((idx != want_idx).any(1))[0]
            assert len(bad) == 0, (case, cell, k, bad[:5], idx[bad[:2]], want_idx[bad[:2]])
            assert dist.tobytes() == want_dist.tobytes()
            assert np.array_equal(X, pts4[idx.astype(np.int64), :3].astype(np.float64))
            m.close()
    fx.close()


def test_map_arguments():
    import torch
    from lidar_feature_extraction_amd import FeatureExtraction
    from lidar_feature_extraction_amd.binding import LfxError
    fx = FeatureExtraction(device=0, max_points_per_scan=1024, max_batch=1)
    pts = np.zeros((10, 4), np.float32)
    pts[:, 0] = np.arange(10)
    d_pts = torch.from_numpy(pts).to("cuda:0")
    with pytest.raises(LfxError):
        fx.make_map(d_pts.data_ptr(), 10, -1.0)
    with pytest.raises(LfxError):
        fx.make_map(d_pts.data_ptr(), 0, 1.0)
    bad = pts.copy()
    bad[3, 1] = np.inf
    d_bad = torch.from_numpy(bad).to("cuda:0")
    with pytest.raises(LfxError):
        fx.make_map(d_bad.data_ptr(), 10, 1.0)
    m = fx.make_map(d_pts.data_ptr(), 10, 1.0)
    d_q = torch.zeros((1, 3), dtype=torch.float64, device="cuda:0")
    with pytest.raises(LfxError):
        m.nearest(d_q.data_ptr(), 1, 11)            # more neighbours than the map has points
    with pytest.raises(LfxError):
        m.nearest(d_q.data_ptr(), 1, 17)
    m.close()
    fx.close()


def test_many_queries_and_small_k():
    """20 000 queries in one call (one wave each), k = 2 and 7, against scipy's exact KD-tree where distances are distinct."""
    import torch
    from scipy.spatial import cKDTree
    from lidar_feature_extraction_amd import FeatureExtraction
    rng = np.random.default_rng(9)
    pts = np.zeros((30000, 4), np.float32)
    pts[:, :3] = rng.normal(0, 8, (30000, 3)) * [1, 1, 0.2]
    queries = rng.normal(0, 9, (20000, 3)) * [1, 1, 0.2]
    fx = FeatureExtraction(device=0, max_points_per_scan=1024, max_batch=1)
    d_pts = torch.from_numpy(pts).to("cuda:0")
    m = fx.make_map(d_pts.data_ptr(), len(pts), 0.8)
    tree = cKDTree(pts[:, :3].astype(np.float64))
    for k in (2, 7):
        X, dist, idx = _query(fx, m, queries, k)
        wd, wi = tree.query(queries, k)
        assert np.array_equal(idx.astype(np.int64), wi)
        assert np.allclose(dist, wd * wd, rtol=1e-12, atol=0)
    m.close()
    fx.close()
