"""The C++ host side (include/lfx.hpp over the C ABI) as a maintainer of the reference node would use it: the compiled
examples/extract_scan is run in a fresh child process on a scan written to a file, and what it writes back -- labels,
curvature, edge / surface index lists and the two PointXYZIR clouds (feature_extraction.cpp:142-151) -- is compared
with the CPU oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "lidar_feature_extraction_amd", "_lib", "extract_scan")

from lidar_feature_extraction_amd import HyperParameters, make_scan, POINT_DTYPE  # noqa: E402


def _read_result(path):
    raw = open(path, "rb").read()
    head = np.frombuffer(raw, np.uint32, 5)
    assert head[0] == 0x3158464C
    n, ne, ns, nr = (int(v) for v in head[1:])
    at = 20
    labels = np.frombuffer(raw, np.uint8, n, at)
    at += (n + 3) // 4 * 4
    curvature = np.frombuffer(raw, np.float64, n, at)
    at += 8 * n
    edge_index = np.frombuffer(raw, np.uint32, ne, at)
    at += 4 * ne
    surface_index = np.frombuffer(raw, np.uint32, ns, at)
    at += 4 * ns
    edge = np.frombuffer(raw, POINT_DTYPE, ne, at)
    at += 32 * ne
    surface = np.frombuffer(raw, POINT_DTYPE, ns, at)
    at += 32 * ns
    rings = np.frombuffer(raw, np.dtype([("id", "<u2"), ("status", "<u2"), ("count", "<u4")]), nr, at)
    assert at + 8 * nr == len(raw)
    return dict(labels=labels, curvature=curvature, edge_index=edge_index, surface_index=surface_index, edge=edge,
                surface=surface, rings=rings)


def test_example_builds_without_a_gpu():
    """(CPU) __graft_entry__.build() compiles and links the C++ caller of the boundary."""
    if not os.path.exists(EXE):
        import __graft_entry__
        __graft_entry__.build()
    assert os.access(EXE, os.X_OK)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["organised-defaults", "ragged-launch", "rings-unknown"])
def test_cpp_host_side_against_the_oracle(case, tmp_path):
    from oracle import binding as OB
    if case == "organised-defaults":
        cloud, rings, hp, extra = make_scan(16, 900, seed=1300), 16, HyperParameters(), []
    elif case == "ragged-launch":
        cloud, rings, hp, extra = make_scan(32, 1024, seed=1301, drop_fraction=0.1), 32, HyperParameters.launch_yaml(), ["launch"]
    else:
        cloud, rings, hp, extra = make_scan(16, 1200, seed=1302), 0, HyperParameters(), []
    src, dst = str(tmp_path / "scan.bin"), str(tmp_path / "features.bin")
    cloud.tofile(src)
    r = subprocess.run([EXE, src, dst, str(rings)] + extra, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got = _read_result(dst)
    op = OB.Params(hp.padding, hp.neighbor_degree_threshold, hp.distance_diff_threshold, hp.parallel_beam_min_range_ratio,
                   hp.edge_threshold, hp.surface_threshold, hp.min_range, hp.max_range, hp.n_blocks)
    want = OB.extract(cloud, op, canonical_ties=False)
    assert want["angle_ties"] == 0 and want["curvature_ties"] == 0
    assert np.array_equal(got["labels"], want["labels"])
    assert got["curvature"].tobytes() == want["curvature"].tobytes()
    assert np.array_equal(got["edge_index"], want["edge_index"].astype(np.uint32))
    assert np.array_equal(got["surface_index"], want["surface_index"].astype(np.uint32))
    assert got["rings"]["id"].tolist() == want["ring_id"].tolist() and got["rings"]["count"].tolist() == want["ring_count"].tolist()
    assert np.array_equal(got["rings"]["status"] != 0, want["ring_status"] != 0)
    # the clouds the node appends to (AppendXYZIR, label.hpp:166-179): x, y, z, intensity <- (float)curvature, ring
    for name, idx in (("edge", want["edge_index"]), ("surface", want["surface_index"])):
        g = got[name]
        assert np.array_equal(g["x"], cloud["x"][idx]) and np.array_equal(g["y"], cloud["y"][idx]) and np.array_equal(g["z"], cloud["z"][idx])
        assert np.array_equal(g["intensity"], want["curvature"][idx].astype(np.float32))
        assert np.array_equal(g["ring"], cloud["ring"][idx])
    assert len(got["edge"]) > 0 and len(got["surface"]) > 0
    assert "scan_edge %d" % len(want["edge_index"]) in r.stdout


LOCALIZE = os.path.join(ROOT, "lidar_feature_extraction_amd", "_lib", "localize_scan")


def test_localizer_example_builds_without_a_gpu():
    if not os.path.exists(LOCALIZE):
        import __graft_entry__
        __graft_entry__.build()
    assert os.access(LOCALIZE, os.X_OK)


@pytest.mark.gpu
def test_cpp_localizer_against_the_oracle(tmp_path):
    """lfx::Localizer (include/lfx.hpp; the reference's Localizer, localizer.hpp:48-95) in a C++ process of its own: Init,
    Update on the extraction's device clouds, Update on host clouds; both poses against the oracle chain extract ->
    Downsample -> Optimizer::Run (tolerance: parity unpinned, see tests/test_align_gpu.py)."""
    import ctypes as C
    from oracle import binding as OB
    from tests.test_align_gpu import _oracle_scan, _downsample
    rings, cols = 32, 1024
    cloud = make_scan(rings, cols, seed=7700)
    want = OB.extract(cloud, canonical_ties=False)
    maps = [OB.extract(make_scan(rings, cols, seed=s), canonical_ties=False) for s in (7790, 7791)]
    edge_map = np.ascontiguousarray(np.concatenate([m["edge_points"] for m in maps]), np.float32)
    surf_map = np.ascontiguousarray(np.concatenate([m["surface_points"] for m in maps]), np.float32)
    paths = [str(tmp_path / n) for n in ("edge_map.bin", "surface_map.bin", "scan.bin", "poses.bin")]
    edge_map.tofile(paths[0]); surf_map.tofile(paths[1]); cloud.tofile(paths[2])
    r = subprocess.run([LOCALIZE, paths[0], paths[1], paths[2], str(rings), str(cols), paths[3], "host"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    raw = open(paths[3], "rb").read()
    rec = np.dtype([("pose", "<f8", 12), ("error", "<f8"), ("scale", "<f8"), ("iteration", "<i4"), ("code", "<i4")])
    got = np.frombuffer(raw, rec)
    assert len(got) == 2
    initial = np.array([[1, 0, 0, 0.02], [0, 1, 0, -0.015], [0, 0, 1, 0.01]], np.float64)
    w = _oracle_scan(edge_map, surf_map, 15, want["edge_points"], _downsample(want["surface_points"], 1.0), initial, 20)
    for g in got:
        assert abs(int(g["iteration"]) - w["iteration"]) <= 1 and (int(g["code"]) <= 2) == w["success"]
        assert np.abs(g["pose"].reshape(3, 4) - w["pose"]).max() < (1e-6 if int(g["iteration"]) == w["iteration"] else 2e-3)
    # the two ways in give the same clouds to the same optimizer
    assert got[0]["pose"].tobytes() == got[1]["pose"].tobytes() and got[0]["iteration"] == got[1]["iteration"]
    assert ("update succeeded" in r.stdout) == w["success"]
