"""lfx_scan_to_map_residuals / lfx_edge_residuals (first slice: the reference localizer's Edge::Make and
Surface::MakeFromDownsampled on the device) against the CPU restatement, oracle/lfx_oracle_loc.cpp.  Tolerance, not
bits: Eigen's and nanoflann's arithmetic is not in the image (parity unpinned); the HIP path finds the principal
direction in closed form, the oracle by Jacobi iteration, so the two check each other.  Edge rows are compared up to the
sign of the principal direction (residual and Jacobian row flip together)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PD, PF = C.POINTER(C.c_double), C.POINTER(C.c_float)


def _pose(rng, angle=0.05, shift=0.2):
    ax = rng.standard_normal(3)
    ax /= np.linalg.norm(ax)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    R = np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * K @ K
    return np.ascontiguousarray(np.hstack([R, rng.uniform(-shift, shift, (3, 1))]))


def _same_up_to_sign(res, jac, wres, wjac, tol):
    n = len(res)
    scale_r = np.abs(wres).max() + 1e-30
    scale_j = np.abs(wjac).max() + 1e-30
    for i in range(n):
        a = max(np.abs(res[i] - wres[i]).max() / scale_r, np.abs(jac[i] - wjac[i]).max() / scale_j)
        b = max(np.abs(res[i] + wres[i]).max() / scale_r, np.abs(jac[i] + wjac[i]).max() / scale_j)
        assert min(a, b) < tol, (i, a, b)


def test_edge_and_surface_rows_of_an_extracted_batch():
    import torch
    from lidar_feature_extraction_amd import FeatureExtraction, make_scan, concat
    from oracle import binding as OB
    L = OB.lib()
    rng = np.random.default_rng(21)
    rings, cols, batch, k = 32, 1024, 3, 15
    dev = torch.device("cuda", 0)
    clouds = [make_scan(rings, cols, seed=7300 + s) for s in range(batch)]
    want = [OB.extract(c, canonical_ties=False) for c in clouds]
    # maps: the features of another, nearby scan of the same scene (what a localizer's map holds), a little denser
    ref = OB.extract(make_scan(rings, cols, seed=7399), canonical_ties=False)
    edge_map = np.ascontiguousarray(np.concatenate([ref["edge_points"], want[0]["edge_points"]]), np.float32)
    surf_map = np.ascontiguousarray(np.concatenate([ref["surface_points"], want[1]["surface_points"]]), np.float32)
    pose = _pose(rng)
    fx = FeatureExtraction(device=0, max_points_per_scan=rings * cols, max_batch=batch, max_points_per_ring=cols, max_rings=rings)
    d = torch.from_numpy(concat(clouds).view(np.uint8).copy()).to(dev)
    stream = torch.cuda.current_stream().cuda_stream
    fx.extract_batch_device(d.data_ptr(), [len(c) for c in clouds], stream)
    total = sum(len(c) for c in clouds)
    d_emap, d_smap = torch.from_numpy(edge_map).to(dev), torch.from_numpy(surf_map).to(dev)
    # ---- edge rows, straight from the batch's edge clouds
    d_res = torch.zeros((total, 3), dtype=torch.float64, device=dev)
    d_jac = torch.zeros((total, 21), dtype=torch.float64, device=dev)
    emap, smap = fx.make_map(d_emap.data_ptr(), len(edge_map), 1.0, stream), fx.make_map(d_smap.data_ptr(), len(surf_map), 0.0, stream)
    fx.edge_residuals(emap, pose, k, d_res.data_ptr(), d_jac.data_ptr(), stream)
    # ---- surface rows: downsample (surface.hpp:111), then the planes
    d_down = torch.zeros((total, 4), dtype=torch.float32, device=dev)
    d_dn = torch.zeros(batch, dtype=torch.int32, device=dev)
    d_ds = torch.zeros(batch, dtype=torch.int32, device=dev)
    fx.downsample_surface(1.0, d_down.data_ptr(), d_dn.data_ptr(), d_ds.data_ptr(), stream)
    d_sres = torch.zeros(total, dtype=torch.float64, device=dev)
    d_sjac = torch.zeros((total, 7), dtype=torch.float64, device=dev)
    view = fx.device_view()
    fx.scan_to_map_residuals(1, smap, pose, k, d_down.data_ptr(), view.scan_begin, d_dn.data_ptr(), 1,
                             batch, rings * cols, d_sres.data_ptr(), d_sjac.data_ptr(), stream)
    torch.cuda.synchronize()
    res, jac = d_res.cpu().numpy(), d_jac.cpu().numpy()
    sres, sjac, down, dn = d_sres.cpu().numpy(), d_sjac.cpu().numpy(), d_down.cpu().numpy(), d_dn.cpu().numpy()
    at = 0
    for s, c in enumerate(clouds):
        e = np.ascontiguousarray(want[s]["edge_points"], np.float32)
        ne = len(e)
        wres, wjac = np.zeros((ne, 3)), np.zeros((ne, 21))
        L.orc_loc_edge_residuals(OB.ptr(edge_map, PF), len(edge_map), OB.ptr(pose, PD), k, OB.ptr(e, PF), ne, OB.ptr(wres, PD), OB.ptr(wjac, PD))
        assert ne > 100
        # points whose neighbourhood has no clear direction (two nearly equal leading eigenvalues) are ill-posed for any
        # eigen-solver; the reference keeps them too, but two solvers may return different directions there: skip those
        keep = []
        for i in range(ne):
            dd = ((edge_map[:, :3].astype(np.float64) - (pose[:, :3] @ e[i, :3].astype(np.float64) + pose[:, 3])) ** 2).sum(1)
            nb = edge_map[np.argsort(dd, kind="stable")[:k], :3].astype(np.float64)
            ev = np.linalg.eigvalsh(np.cov(nb.T, bias=True))
            if ev[2] - ev[1] > 1e-3 * ev[2]:
                keep.append(i)
        keep = np.array(keep)
        assert len(keep) > 0.8 * ne
        _same_up_to_sign(res[at:at + ne][keep], jac[at:at + ne][keep], wres[keep], wjac[keep], 1e-7)
        m = int(dn[s])
        dpts = np.ascontiguousarray(down[at:at + m])
        wsr, wsj = np.zeros(m), np.zeros((m, 7))
        L.orc_loc_surface_residuals(OB.ptr(surf_map, PF), len(surf_map), OB.ptr(pose, PD), k, OB.ptr(dpts, PF), m, OB.ptr(wsr, PD), OB.ptr(wsj, PD))
        assert m > 50
        assert np.allclose(sres[at:at + m], wsr, rtol=1e-7, atol=1e-9 * (np.abs(wsr).max() + 1))
        assert np.allclose(sjac[at:at + m], wsj, rtol=1e-7, atol=1e-8 * (np.abs(wsj).max() + 1))
        at += len(c)
    fx.close()


def test_reference_vectors_through_the_device():
    """The geometry pinned by localization/test/test_edge.cpp, through the kernels: a map of points on the x axis gives
    the principal direction (1, 0, 0) (test_edge.cpp:81-92), so a scan point p has the residual (p - p1) x (p - p2) with
    p1, p2 = centre -/+ (1, 0, 0) (:151-176); points on the map's line / plane have zero residual."""
    import torch
    from lidar_feature_extraction_amd import FeatureExtraction
    dev = torch.device("cuda", 0)
    fx = FeatureExtraction(device=0, max_points_per_scan=1024, max_batch=1)
    line = np.zeros((5, 4), np.float32)
    line[:, 0] = [0, 1, 2, 3, 4]                                     # X of test_edge.cpp:159-165
    pts = np.array([[2, 1, 0, 1], [0.5, 0, 0, 1]], np.float32)       # p0 of :177, and a point on the line
    pose = np.hstack([np.eye(3), np.array([[3.0], [2.0], [1.0]])])   # theta0 = 0, t0 = (3, 2, 1), :167-168
    d_map, d_pts = torch.from_numpy(line).to(dev), torch.from_numpy(pts).to(dev)
    line_map = fx.make_map(d_map.data_ptr(), 5, 0.5)
    d_b = torch.zeros(1, dtype=torch.int32, device=dev)
    d_n = torch.tensor([2], dtype=torch.int32, device=dev)
    d_res = torch.zeros((2, 3), dtype=torch.float64, device=dev)
    d_jac = torch.zeros((2, 21), dtype=torch.float64, device=dev)
    fx.scan_to_map_residuals(0, line_map, pose, 5, d_pts.data_ptr(), d_b.data_ptr(), d_n.data_ptr(), 1, 1, 2,
                             d_res.data_ptr(), d_jac.data_ptr(), 0)
    torch.cuda.synchronize()
    res, jac = d_res.cpu().numpy(), d_jac.cpu().numpy().reshape(2, 3, 7)
    centre = np.array([2.0, 0, 0])
    p1, p2 = centre - [1, 0, 0], centre + [1, 0, 0]
    p = pts[0, :3].astype(np.float64) + [3, 2, 1]
    want = np.cross(p - p1, p - p2)
    assert min(np.abs(res[0] - want).max(), np.abs(res[0] + want).max()) < 1e-12
    K = np.array([[0, 0, 0], [0, 0, -2.0], [0, 2.0, 0]])             # Hat(p2 - p1) = Hat((2, 0, 0))
    assert min(np.abs(jac[0][:, 4:] - K).max(), np.abs(jac[0][:, 4:] + K).max()) < 1e-12
    # identity pose: the second point lies on the line
    pose0 = np.hstack([np.eye(3), np.zeros((3, 1))])
    fx.scan_to_map_residuals(0, line_map, pose0, 5, d_pts.data_ptr(), d_b.data_ptr(), d_n.data_ptr(), 1, 1, 2,
                             d_res.data_ptr(), d_jac.data_ptr(), 0)
    torch.cuda.synchronize()
    assert np.abs(d_res.cpu().numpy()[1]).max() < 1e-12
    fx.close()


def test_surface_neighbourhoods_without_a_plane_get_the_zero_row():
    """surface.hpp:78-83 solves X w = -1 by `householderQr().solve` whatever the rank of X.  Where the k nearest map
    points coincide, or lie on one line, R has a zero pivot; what Eigen returns then cannot be known here (its arithmetic is
    not in the image), so the row is given weight 0: residual 0 and u = 0, on the device and in the oracle alike -- it adds
    nothing to any sum of the optimizer, and a caller can tell it (every other surface row has |u| = 1).  Three queries:
    beside 20 coincident map points, beside 20 points on a line along (1, 1, 0), beside a proper patch of a plane."""
    import torch
    from lidar_feature_extraction_amd import FeatureExtraction
    from oracle import binding as OB
    L = OB.lib()
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(5)
    k = 15
    same = np.tile(np.array([[10.0, 0.0, 1.0, 1.0]], np.float32), (20, 1))
    t = np.arange(20, dtype=np.float32)
    line = np.stack([-10 + 0.25 * t, 5 + 0.25 * t, np.full(20, 2.0, np.float32), np.ones(20, np.float32)], 1)
    patch = np.zeros((40, 4), np.float32)
    patch[:, :2] = rng.uniform(-1, 1, (40, 2))
    patch[:, 2] = 0.3 * patch[:, 0] - 7.0
    surf_map = np.ascontiguousarray(np.concatenate([same, line, patch]), np.float32)
    pts = np.array([[10.2, 0.1, 1.1, 1], [-8.0, 7.1, 2.2, 1], [0.1, 0.2, -6.5, 1]], np.float32)
    pose = np.hstack([np.eye(3), np.zeros((3, 1))])
    fx = FeatureExtraction(device=0, max_points_per_scan=1024, max_batch=1)
    d_map, d_pts = torch.from_numpy(surf_map).to(dev), torch.from_numpy(pts).to(dev)
    for cell in (0.0, 1.0):                    # without and with the grid index
        m = fx.make_map(d_map.data_ptr(), len(surf_map), cell)
        d_b = torch.zeros(1, dtype=torch.int32, device=dev)
        d_n = torch.tensor([3], dtype=torch.int32, device=dev)
        d_res = torch.full((3,), 7.0, dtype=torch.float64, device=dev)
        d_jac = torch.full((3, 7), 7.0, dtype=torch.float64, device=dev)
        fx.scan_to_map_residuals(1, m, pose, k, d_pts.data_ptr(), d_b.data_ptr(), d_n.data_ptr(), 1, 1, 3, d_res.data_ptr(), d_jac.data_ptr(), 0)
        torch.cuda.synchronize()
        res, jac = d_res.cpu().numpy(), d_jac.cpu().numpy()
        wres, wjac = np.zeros(3), np.zeros((3, 7))
        L.orc_loc_surface_residuals(OB.ptr(surf_map, PF), len(surf_map), OB.ptr(pose, PD), k, OB.ptr(pts, PF), 3, OB.ptr(wres, PD), OB.ptr(wjac, PD))
        for i in (0, 1):
            assert res[i] == 0.0 and not jac[i].any(), (cell, i, res[i], jac[i])
            assert wres[i] == 0.0 and not wjac[i].any()
        assert abs(np.linalg.norm(jac[2, 4:]) - 1.0) < 1e-12 and np.isfinite(res[2])
        assert np.allclose(res[2], wres[2], rtol=1e-9, atol=1e-12) and np.allclose(jac[2], wjac[2], rtol=1e-9, atol=1e-12)
        # the plane z = 0.3 x - 7: the query's signed distance to it
        want = abs(0.3 * pts[2, 0] - pts[2, 2] - 7.0) / np.sqrt(1 + 0.09)
        assert abs(abs(res[2]) - want) < 1e-5
    fx.close()
