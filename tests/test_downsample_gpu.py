"""lfx_voxel_downsample / lfx_downsample_surface (Downsample = pcl::VoxelGrid, downsample.hpp:37-51, applied
to scan_surface at localization/.../surface.hpp:111) against the oracle's restatement of the PCL algorithm, bit for bit
(both sum the points of a cell in input order).  Parity with PCL itself is unpinned: see oracle/lfx_oracle.h."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _oracle(cloud, leaf):
    from oracle import binding as OB
    L = OB.lib()
    cloud = np.ascontiguousarray(cloud, np.float32)
    out = np.zeros_like(cloud)
    n = C.c_int(0)
    rc = L.orc_voxel_downsample(OB.ptr(cloud, C.POINTER(C.c_float)), len(cloud), leaf, OB.ptr(out, C.POINTER(C.c_float)), C.byref(n))
    return rc, out[:n.value].copy()


def test_voxel_downsample_random_clouds():
    import torch
    from lidar_feature_extraction_amd import FeatureExtraction
    rng = np.random.default_rng(11)
    sizes = [1, 2, 0, 777, 5000, 40000, 3, 1025, 130000]
    clouds = []
    for k, n in enumerate(sizes):
        c = np.ones((n, 4), np.float32)
        scale = [3.0, 30.0, 120.0][k % 3]
        c[:, :3] = (rng.standard_normal((n, 3)) * scale).astype(np.float32)
        c[:, 2] *= 0.1
        clouds.append(c)
    clouds.append(np.array([[0, 0, 0, 1], [4000, 4000, 4000, 1]], np.float32))       # leaf too small for this one
    begin = np.zeros(len(clouds), np.uint32)
    begin[1:] = np.cumsum([len(c) for c in clouds])[:-1]
    count = np.array([len(c) for c in clouds], np.uint32)
    total = int(count.sum())
    dev = torch.device("cuda", 0)
    d_pts = torch.from_numpy(np.concatenate(clouds)).to(dev)
    d_begin = torch.from_numpy(begin.astype(np.int32)).to(dev)
    d_count = torch.from_numpy(count.astype(np.int32)).to(dev)
    d_out = torch.zeros((total, 4), dtype=torch.float32, device=dev)
    d_n = torch.zeros(len(clouds), dtype=torch.int32, device=dev)
    d_st = torch.zeros(len(clouds), dtype=torch.int32, device=dev)
    fx = FeatureExtraction(device=0, max_points_per_scan=1024, max_batch=1)
    stream = torch.cuda.current_stream().cuda_stream
    for leaf in (1.0, 0.25, 0.01):
        fx.voxel_downsample(d_pts.data_ptr(), d_begin.data_ptr(), d_count.data_ptr(), 1, len(clouds), total, leaf, d_out.data_ptr(),
                            d_n.data_ptr(), d_st.data_ptr(), stream)
        torch.cuda.synchronize()
        out, n_out, st = d_out.cpu().numpy(), d_n.cpu().numpy(), d_st.cpu().numpy()
        for k, c in enumerate(clouds):
            rc, want = _oracle(c, leaf)
            assert st[k] == rc, (leaf, k)
            if rc == 0:
                assert n_out[k] == len(want), (leaf, k, n_out[k], len(want))
                assert out[begin[k]:begin[k] + len(want)].tobytes() == want.tobytes(), (leaf, k)
    fx.close()


def test_downsample_of_the_surface_clouds_of_a_batch():
    """The localizer's use: Downsample(scan_surface, 1.0) (surface.hpp:111), chained on the device behind the extraction."""
    import torch
    from lidar_feature_extraction_amd import FeatureExtraction, make_scan, concat
    from oracle import binding as OB
    rings, cols = 32, 1024
    clouds = [make_scan(rings, cols, seed=6100 + k, drop_fraction=(0.1 if k == 2 else 0.0)) for k in range(4)]
    dev = torch.device("cuda", 0)
    fx = FeatureExtraction(device=0, max_points_per_scan=rings * cols, max_batch=4, max_points_per_ring=cols, max_rings=rings)
    d = torch.from_numpy(concat(clouds).view(np.uint8).copy()).to(dev)
    stream = torch.cuda.current_stream().cuda_stream
    fx.extract_batch_device(d.data_ptr(), [len(c) for c in clouds], stream)
    total = sum(len(c) for c in clouds)
    d_out = torch.zeros((total, 4), dtype=torch.float32, device=dev)
    d_n = torch.zeros(4, dtype=torch.int32, device=dev)
    d_st = torch.zeros(4, dtype=torch.int32, device=dev)
    fx.downsample_surface(1.0, d_out.data_ptr(), d_n.data_ptr(), d_st.data_ptr(), stream)
    torch.cuda.synchronize()
    out, n_out = d_out.cpu().numpy(), d_n.cpu().numpy()
    at = 0
    for k, c in enumerate(clouds):
        w = OB.extract(c, canonical_ties=False)
        surf = w["surface_points"].copy()
        rc, want = _oracle(surf, 1.0)
        assert rc == 0 and n_out[k] == len(want) and 0 < len(want) < len(surf)
        assert out[at:at + len(want)].tobytes() == want.tobytes(), k
        at += len(c)
    fx.close()
