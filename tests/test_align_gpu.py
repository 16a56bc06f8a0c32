"""lfx_scan_to_map_align / lfx_align_point_pairs / lfx_localize_batch (second slice: the reference
localizer's Optimizer::Run on the device, optimizer.hpp:79-123) against the CPU restatement in oracle/lfx_oracle_loc.cpp,
which itself passes the reference's optimizer tests (tests/test_oracle_localization.py).  The scenarios of
localization/test/test_optimizer.cpp run through the device with the reference's own bounds.  Tolerance, not bits:
Eigen's, nanoflann's and PCL's arithmetic is not in the image (parity unpinned), and the device sums rows in a tree."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PD, PF, PI = C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_int)


def _rotation(axis_angle):
    th = np.asarray(axis_angle, np.float64)
    k = np.linalg.norm(th)
    if k == 0:
        return np.eye(3)
    u = th / k
    K = np.array([[0, -u[2], u[1]], [u[2], 0, -u[0]], [-u[1], u[0], 0]])
    return np.eye(3) + np.sin(k) * K + (1 - np.cos(k)) * K @ K


def _pose(axis_angle, t):
    return np.ascontiguousarray(np.hstack([_rotation(axis_angle), np.asarray(t, np.float64).reshape(3, 1)]))


def _pose_of_quaternion(q_unnormalised, t):
    from oracle import binding as OB
    q = np.ascontiguousarray(np.asarray(q_unnormalised, np.float64) / np.linalg.norm(q_unnormalised))
    R = np.zeros(9)
    OB.lib().orc_loc_rotation_matrix(OB.ptr(q, PD), OB.ptr(R, PD))
    return np.ascontiguousarray(np.hstack([R.reshape(3, 3), np.asarray(t, np.float64).reshape(3, 1)]))


def _oracle_pairs(X, Y, pose, max_iter):
    from oracle import binding as OB
    X, Y, pose = (np.ascontiguousarray(a, np.float64) for a in (X, Y, pose))
    out, err, scale, it, code = np.zeros(12), C.c_double(), C.c_double(), C.c_int(), C.c_int()
    ok = OB.lib().orc_loc_optimize_pairs(OB.ptr(X, PD), OB.ptr(Y, PD), len(X), OB.ptr(pose, PD), max_iter, OB.ptr(out, PD),
                                         C.byref(err), C.byref(scale), C.byref(it), C.byref(code))
    return dict(pose=out.reshape(3, 4), error=err.value, error_scale=scale.value, iteration=it.value, code=code.value, success=bool(ok))


def _oracle_scan(edge_map, surf_map, k, edge, surf_down, pose, max_iter):
    from oracle import binding as OB
    edge_map, surf_map, edge, surf_down = (np.ascontiguousarray(a, np.float32) for a in (edge_map, surf_map, edge, surf_down))
    pose = np.ascontiguousarray(pose, np.float64)
    out, err, scale, it, code = np.zeros(12), C.c_double(), C.c_double(), C.c_int(), C.c_int()
    ok = OB.lib().orc_loc_optimize_scan(OB.ptr(edge_map, PF), len(edge_map), OB.ptr(surf_map, PF), len(surf_map), k,
                                        OB.ptr(edge, PF), len(edge), OB.ptr(surf_down, PF), len(surf_down), OB.ptr(pose, PD),
                                        max_iter, OB.ptr(out, PD), C.byref(err), C.byref(scale), C.byref(it), C.byref(code))
    return dict(pose=out.reshape(3, 4), error=err.value, error_scale=scale.value, iteration=it.value, code=code.value, success=bool(ok))


def _same_result(got, want, what, pose_tol=1e-8, rel=1e-7):
    assert (got["code"], got["iteration"], got["success"]) == (want["code"], want["iteration"], want["success"]), (what, got, want)
    assert np.abs(got["pose"] - want["pose"]).max() <= pose_tol * (1 + np.abs(want["pose"]).max()), (what, got["pose"], want["pose"])
    assert abs(got["error"] - want["error"]) <= rel * abs(want["error"]) + 1e-18, (what, got["error"], want["error"])
    assert abs(got["error_scale"] - want["error_scale"]) <= rel * abs(want["error_scale"]) + 1e-18, what


def _run_pairs(fx, problems, max_iter):
    """problems: list of (X, Y, initial pose); one device call for all of them."""
    import torch
    dev = torch.device("cuda", 0)
    counts = np.array([len(p[0]) for p in problems], np.int32)
    begins = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.int32)
    total = int(counts.sum())
    X = np.concatenate([np.asarray(p[0], np.float64).reshape(-1, 3) for p in problems]) if total else np.zeros((0, 3))
    Y = np.concatenate([np.asarray(p[1], np.float64).reshape(-1, 3) for p in problems]) if total else np.zeros((0, 3))
    dX = torch.from_numpy(np.ascontiguousarray(np.vstack([X, np.zeros((1, 3))]))).to(dev)
    dY = torch.from_numpy(np.ascontiguousarray(np.vstack([Y, np.zeros((1, 3))]))).to(dev)
    db, dn = torch.from_numpy(begins).to(dev), torch.from_numpy(counts).to(dev)
    poses = np.stack([p[2] for p in problems])
    return fx.align_point_pairs(dX.data_ptr(), dY.data_ptr(), db.data_ptr(), dn.data_ptr(), int(counts.max()), total, max_iter, poses,
                                torch.cuda.current_stream().cuda_stream)


def test_reference_optimizer_scenarios_through_the_device(refvec):
    """localization/test/test_optimizer.cpp:53-242 (SimpleDatasetConvergenceCheck, ShouldReturnFalseForEmptyData,
    ShouldReturnFalseWhenNoConvergence) with the device in place of Optimizer<AlignmentProblem>::Run."""
    from lidar_feature_extraction_amd import FeatureExtraction
    g = refvec["loc_alignment"]
    true = _pose_of_quaternion(g["q_true_wxyz_unnormalised"], g["t_true"])
    X = np.asarray(g["X"], np.float64)
    Y = X @ true[:, :3].T + true[:, 3]
    fx = FeatureExtraction(device=0, max_points_per_scan=1024, max_batch=1)
    b = g["bounds"]
    runs = g["runs"]
    problems = [(X, Y, _pose_of_quaternion(r["q_wxyz_unnormalised"], r["t"])) for r in runs]
    got = _run_pairs(fx, problems, g["max_iter"])                     # all four starts in one call
    for run, r, pr in zip(runs, got, problems):
        assert r["success"] == b["success"], (run["name"], r)
        if "iteration_eq" in run:
            assert r["iteration"] == run["iteration_eq"]
        else:
            assert r["iteration"] < run["iteration_lt"], (run["name"], r["iteration"])
        assert r["error"] < b["error_lt"] and r["error_scale"] < b["error_scale_lt"]
        assert np.linalg.norm(true[:, :3] - r["pose"][:, :3]) <= b["rotation_norm_le"]
        assert np.linalg.norm(true[:, 3] - r["pose"][:, 3]) <= b["translation_norm_le"]
        want = _oracle_pairs(pr[0], pr[1], pr[2], g["max_iter"])
        # exact data: after the first step error and scale are zero to rounding, so which of the three successful stopping
        # tests fires first is rounding too; the reference's test asks for success, the iteration bound and the pose
        assert r["success"] == want["success"] and abs(r["iteration"] - want["iteration"]) <= 1
        assert np.abs(r["pose"] - want["pose"]).max() < 1e-9
    ident = _pose([0, 0, 0], [0, 0, 0])
    e = g["empty"]
    r = _run_pairs(fx, [(np.zeros((0, 3)), np.zeros((0, 3)), ident)], 10)[0]
    assert (r["iteration"], r["success"], r["error"], r["error_scale"]) == (e["iteration"], e["success"], e["error"], e["error_scale"])
    assert r["message"] == "The input data is empty"
    nc = g["no_convergence"]
    rng = np.random.default_rng(3)
    Xn, Yn = rng.normal(*nc["x"], (nc["n"], 3)), rng.normal(*nc["y"], (nc["n"], 3))
    r = _run_pairs(fx, [(Xn, Yn, ident)], nc["max_iter"])[0]
    assert r["iteration"] == nc["iteration"] and r["success"] == nc["success"] and r["message"] == "The iteration reached the maximum value"
    assert r["error"] > nc["error_gt"] and r["error_scale"] > nc["error_scale_gt"]
    _same_result(r, _oracle_pairs(Xn, Yn, ident, nc["max_iter"]), "no convergence")
    # WeightedUpdate returns zero when D is degenerate (:313-328): no step, which CheckConvergence reads as converged
    r = _run_pairs(fx, [(np.zeros((1, 3)), np.ones((1, 3)), ident)], 10)[0]
    assert (r["code"], r["iteration"]) == (0, 0) and np.array_equal(r["pose"], ident)
    fx.close()


def test_batches_of_point_pair_problems_against_the_oracle():
    """Random problems of different sizes in one call (ragged batch, outliers so that the Huber weights matter, one empty
    cloud in the middle); every result equals the oracle's: stopping reason, iteration, pose, error, scale."""
    from lidar_feature_extraction_amd import FeatureExtraction
    rng = np.random.default_rng(11)
    fx = FeatureExtraction(device=0, max_points_per_scan=1024, max_batch=1)
    problems = []
    for i in range(24):
        n = 0 if i == 7 else int(rng.integers(4, 3000))
        X = rng.uniform(-20, 20, (n, 3))
        true = _pose(rng.normal(0, 0.3, 3), rng.normal(0, 2, 3))
        Y = X @ true[:, :3].T + true[:, 3] + rng.normal(0, 0.02, (n, 3))
        if n > 10:
            out = rng.choice(n, n // 10, replace=False)
            Y[out] += rng.normal(0, 3.0, (len(out), 3))
        start = _pose(rng.normal(0, 0.2, 3), rng.normal(0, 1, 3))
        problems.append((X, Y, start))
    for max_iter in (20, 3):
        got = _run_pairs(fx, problems, max_iter)
        codes = set()
        for i, (r, pr) in enumerate(zip(got, problems)):
            want = _oracle_pairs(pr[0], pr[1], pr[2], max_iter)
            _same_result(r, want, "problem %d, max_iter %d" % (i, max_iter))
            codes.add(r["code"])
        assert 4 in codes and len(codes) >= 2
    fx.close()


def _scene(rings, cols, seeds):
    from lidar_feature_extraction_amd import make_scan
    from oracle import binding as OB
    clouds = [make_scan(rings, cols, seed=s) for s in seeds]
    return clouds, [OB.extract(c, canonical_ties=False) for c in clouds]


def _downsample(points, leaf):
    from oracle import binding as OB
    pts = np.ascontiguousarray(points, np.float32)
    out, n_out = np.zeros_like(pts), C.c_int(0)
    rc = OB.lib().orc_voxel_downsample(OB.ptr(pts, PF), len(pts), C.c_float(leaf), OB.ptr(out, PF), C.byref(n_out))
    return pts.copy() if rc else np.ascontiguousarray(out[:n_out.value])


@pytest.mark.parametrize("batch", [4, 1])
def test_localize_batch_against_the_oracle(batch):
    """Localizer::Update for a batch straight after extraction: maps = the features of other scans of the scene, every scan
    from its own perturbed pose.  Against the oracle chain (extract -> Downsample -> Optimizer::Run).  A batch, and one scan
    (the streaming case)."""
    import torch
    from lidar_feature_extraction_amd import FeatureExtraction, concat
    rng = np.random.default_rng(31)
    rings, cols, k, max_iter = 32, 1024, 15, 20
    clouds, want = _scene(rings, cols, [7500 + s for s in range(batch)])
    _, maps = _scene(rings, cols, [7590, 7591, 7592])
    edge_map = np.ascontiguousarray(np.concatenate([m["edge_points"] for m in maps]), np.float32)
    surf_map = np.ascontiguousarray(np.concatenate([m["surface_points"] for m in maps]), np.float32)
    dev = torch.device("cuda", 0)
    fx = FeatureExtraction(device=0, max_points_per_scan=rings * cols, max_batch=batch, max_points_per_ring=cols, max_rings=rings)
    d = torch.from_numpy(concat(clouds).view(np.uint8).copy()).to(dev)
    stream = torch.cuda.current_stream().cuda_stream
    fx.extract_batch_device(d.data_ptr(), [len(c) for c in clouds], stream)
    d_emap, d_smap = torch.from_numpy(edge_map).to(dev), torch.from_numpy(surf_map).to(dev)
    poses = np.stack([_pose(rng.normal(0, 0.004, 3), rng.normal(0, 0.03, 3)) for _ in range(batch)])
    if batch > 1:
        poses[0] = _pose([0, 0, 0], [0, 0, 0])
    emap, smap = fx.make_map(d_emap.data_ptr(), len(edge_map), 1.0, stream), fx.make_map(d_smap.data_ptr(), len(surf_map), 2.0, stream)
    got = fx.localize_batch(emap, smap, poses, k, max_iter, 1.0, stream)
    # the same with maps that have no grid (every query reads the whole map): the same neighbours, so the same bits
    emap0, smap0 = fx.make_map(d_emap.data_ptr(), len(edge_map), 0.0, stream), fx.make_map(d_smap.data_ptr(), len(surf_map), 0.0, stream)
    got0 = fx.localize_batch(emap0, smap0, poses, k, max_iter, 1.0, stream)
    for a, b in zip(got, got0):
        assert a["pose"].tobytes() == b["pose"].tobytes() and (a["code"], a["iteration"], a["error"], a["error_scale"]) == (
            b["code"], b["iteration"], b["error"], b["error_scale"])
    moved = 0
    for s in range(batch):
        down = _downsample(want[s]["surface_points"], 1.0)
        w = _oracle_scan(edge_map, surf_map, k, want[s]["edge_points"], down, poses[s], max_iter)
        # a stopping test that compares this iteration's error with the last one's may fall either way when the two are
        # equal to rounding; the poses agree all the same
        if (got[s]["code"], got[s]["iteration"]) != (w["code"], w["iteration"]):
            assert abs(got[s]["iteration"] - w["iteration"]) <= 1, (s, got[s], w)
            assert np.abs(got[s]["pose"] - w["pose"]).max() < 2e-3, (s, got[s], w)
        else:
            _same_result(got[s], w, "scan %d" % s, pose_tol=1e-6, rel=1e-5)
        assert got[s]["success"] == w["success"]
        moved += int(np.abs(got[s]["pose"] - poses[s]).max() > 1e-4)
    assert moved >= max(batch - 1, 1)
    # a pose array that is not one pose per scan of the batch is refused (it sizes the result array too)
    with pytest.raises(Exception, match="n_scans"):
        fx.localize_batch(emap, smap, np.concatenate([poses, poses[:1]]), k, max_iter, 1.0, stream)
    fx.close()


def test_localize_calls_in_a_row_and_both_ways_of_sizing_the_rows():
    """lfx_localize_batch asks the device nothing before it aligns a few scans: launches sized by the previous call's clouds,
    as many iterations queued as the previous call needed, rows laid out by a bound.  So: a call after a much smaller one
    (short clouds, two iterations) must give what a fresh context gives; and a batch large enough for the other layout (the
    clouds' lengths fetched first, rows packed) must give every scan what it gets alone -- bit for bit, the arithmetic of a
    scan does not depend on its neighbours."""
    import torch
    from lidar_feature_extraction_amd import FeatureExtraction, concat, make_scan
    rng = np.random.default_rng(77)
    rings, cols, k = 32, 1024, 15
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    _, maps = _scene(rings, cols, [7590, 7591, 7592])
    edge_map = np.ascontiguousarray(np.concatenate([m["edge_points"] for m in maps]), np.float32)
    surf_map = np.ascontiguousarray(np.concatenate([m["surface_points"] for m in maps]), np.float32)
    d_emap, d_smap = torch.from_numpy(edge_map).to(dev), torch.from_numpy(surf_map).to(dev)
    full = [make_scan(rings, cols, seed=7600 + s) for s in range(8)]
    small = full[0][:4 * cols].copy()                      # the first four columns' worth of points: a few features per ring
    poses = np.stack([_pose(rng.normal(0, 0.004, 3), rng.normal(0, 0.03, 3)) for _ in range(8)])

    def alone(fx, emap, smap, cloud, pose, max_iter=20):
        d = torch.from_numpy(cloud.view(np.uint8).copy()).to(dev)
        fx.extract_batch_device(d.data_ptr(), [len(cloud)], stream)
        return fx.localize_batch(emap, smap, pose[None], k, max_iter, 1.0, stream)[0]

    def same(a, b):
        return a["pose"].tobytes() == b["pose"].tobytes() and (a["code"], a["iteration"], a["error"], a["error_scale"]) == (
            b["code"], b["iteration"], b["error"], b["error_scale"])

    batch = 44                                             # 44 x 32 768 points x 400 bytes: past the bound's budget of 192 MB
    fx = FeatureExtraction(device=0, max_points_per_scan=rings * cols, max_batch=batch, max_points_per_ring=cols, max_rings=rings)
    emap, smap = fx.make_map(d_emap.data_ptr(), len(edge_map), 1.0, stream), fx.make_map(d_smap.data_ptr(), len(surf_map), 2.0, stream)
    fresh = alone(fx, emap, smap, full[3], poses[3])
    assert fresh["iteration"] >= 2
    alone(fx, emap, smap, small, poses[0], max_iter=2)     # leaves short guesses behind
    again = alone(fx, emap, smap, full[3], poses[3])
    assert same(fresh, again), (fresh, again)
    clouds = [full[s % 8] for s in range(batch)]
    d = torch.from_numpy(concat(clouds).view(np.uint8).copy()).to(dev)
    fx.extract_batch_device(d.data_ptr(), [len(c) for c in clouds], stream)
    got = fx.localize_batch(emap, smap, np.stack([poses[s % 8] for s in range(batch)]), k, 20, 1.0, stream)
    for s in (0, 3, 13, 43):
        assert same(got[s], got[s % 8]), s
        assert same(got[s], alone(fx, emap, smap, full[s % 8], poses[s % 8])), s
    fx.close()


def test_localize_batch_with_a_leaf_too_small_for_the_cloud():
    """PCL's VoxelGrid hands a cloud back unfiltered when the leaf is too small for its extent (downsample.hpp:37-51): the rows
    are then built from every surface point.  lfx_localize_batch does that copy inside its Downsample launch."""
    import torch
    from lidar_feature_extraction_amd import FeatureExtraction, concat
    rng = np.random.default_rng(5)
    rings, cols, k, max_iter, leaf = 16, 900, 15, 3, 1e-7
    clouds, want = _scene(rings, cols, [7700])
    _, maps = _scene(rings, cols, [7790, 7791])
    edge_map = np.ascontiguousarray(np.concatenate([m["edge_points"] for m in maps]), np.float32)
    surf_map = np.ascontiguousarray(np.concatenate([m["surface_points"] for m in maps]), np.float32)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    fx = FeatureExtraction(device=0, max_points_per_scan=rings * cols, max_batch=1, max_points_per_ring=cols, max_rings=rings)
    d = torch.from_numpy(concat(clouds).view(np.uint8).copy()).to(dev)
    fx.extract_batch_device(d.data_ptr(), [len(c) for c in clouds], stream)
    d_emap, d_smap = torch.from_numpy(edge_map).to(dev), torch.from_numpy(surf_map).to(dev)
    emap, smap = fx.make_map(d_emap.data_ptr(), len(edge_map), 1.0, stream), fx.make_map(d_smap.data_ptr(), len(surf_map), 1.0, stream)
    pose = _pose(rng.normal(0, 0.004, 3), rng.normal(0, 0.03, 3))
    got = fx.localize_batch(emap, smap, pose[None], k, max_iter, leaf, stream)[0]
    down = _downsample(want[0]["surface_points"], leaf)
    assert len(down) == len(want[0]["surface_points"])                  # unfiltered
    w = _oracle_scan(edge_map, surf_map, k, want[0]["edge_points"], down, pose, max_iter)
    _same_result(got, w, "unfiltered", pose_tol=1e-6, rel=1e-5)
    fx.close()


def test_scan_to_map_align_on_caller_clouds_and_its_arguments():
    """The general entry: clouds the caller lays out (ragged, one scan without surface points), fewer iterations; argument
    checks."""
    import torch
    from lidar_feature_extraction_amd import FeatureExtraction
    from lidar_feature_extraction_amd.binding import LfxError
    rng = np.random.default_rng(41)
    rings, cols, k = 16, 900, 15
    _, maps = _scene(rings, cols, [7690, 7691])
    _, scans = _scene(rings, cols, [7600, 7601, 7602])
    edge_map = np.ascontiguousarray(np.concatenate([m["edge_points"] for m in maps]), np.float32)
    surf_map = np.ascontiguousarray(np.concatenate([m["surface_points"] for m in maps]), np.float32)
    edges = [np.ascontiguousarray(s["edge_points"], np.float32) for s in scans]
    surfs = [_downsample(s["surface_points"], 1.0) for s in scans]
    surfs[1] = surfs[1][:0]
    edges[2] = edges[2][: len(edges[2]) // 2]
    dev = torch.device("cuda", 0)
    fx = FeatureExtraction(device=0, max_points_per_scan=1024, max_batch=1)

    def lay(parts):
        n = np.array([len(p) for p in parts], np.int32)
        b = np.concatenate([[0], np.cumsum(n)[:-1]]).astype(np.int32)
        return torch.from_numpy(np.concatenate(parts + [np.zeros((1, 4), np.float32)])).to(dev), torch.from_numpy(b).to(dev), torch.from_numpy(n).to(dev), n
    d_e, d_eb, d_en, en = lay(edges)
    d_s, d_sb, d_sn, sn = lay(surfs)
    d_emap, d_smap = torch.from_numpy(edge_map).to(dev), torch.from_numpy(surf_map).to(dev)
    poses = np.stack([_pose(rng.normal(0, 0.003, 3), rng.normal(0, 0.02, 3)) for _ in range(3)])
    emap, smap = fx.make_map(d_emap.data_ptr(), len(edge_map), 0.7), fx.make_map(d_smap.data_ptr(), len(surf_map), 1.5)
    tiny = fx.make_map(d_emap.data_ptr(), 5, 1.0)
    for max_iter in (20, 2):
        got = fx.scan_to_map_align(emap, smap, k, max_iter, d_e.data_ptr(),
                                   d_eb.data_ptr(), d_en.data_ptr(), 1, int(en.max()), int(en.sum()), d_s.data_ptr(), d_sb.data_ptr(),
                                   d_sn.data_ptr(), 1, int(sn.max()), int(sn.sum()), poses, 0)
        for s in range(3):
            w = _oracle_scan(edge_map, surf_map, k, edges[s], surfs[s], poses[s], max_iter)
            if (got[s]["code"], got[s]["iteration"]) != (w["code"], w["iteration"]):
                assert abs(got[s]["iteration"] - w["iteration"]) <= 1 and np.abs(got[s]["pose"] - w["pose"]).max() < 2e-3, (s, got[s], w)
            else:
                _same_result(got[s], w, "scan %d, max_iter %d" % (s, max_iter), pose_tol=1e-6, rel=1e-5)
    with pytest.raises(LfxError):
        fx.scan_to_map_align(emap, smap, k, 0, d_e.data_ptr(), d_eb.data_ptr(),
                             d_en.data_ptr(), 1, 1, 1, d_s.data_ptr(), d_sb.data_ptr(), d_sn.data_ptr(), 1, 1, 1, poses, 0)
    with pytest.raises(LfxError):
        fx.scan_to_map_align(tiny, smap, k, 5, d_e.data_ptr(), d_eb.data_ptr(),
                             d_en.data_ptr(), 1, 1, 1, d_s.data_ptr(), d_sb.data_ptr(), d_sn.data_ptr(), 1, 1, 1, poses, 0)
    fx.close()


def test_large_problem_degenerate_map_and_host_clouds():
    """(1) A scan with more rows than the step kernel keeps in LDS (its selection then runs over global memory);
    (2) maps that leave the pose unconstrained -- one plane, no edges worth the name: WeightedUpdate returns zero
    (IsDegenerate, optimizer.cpp:66-68), which CheckConvergence reads as converged at iteration 0 with the pose untouched;
    (3) lfx_localize_host: clouds from the host, empty clouds included."""
    from lidar_feature_extraction_amd import FeatureExtraction
    rng = np.random.default_rng(51)
    fx = FeatureExtraction(device=0, max_points_per_scan=1024, max_batch=1)
    # (1)
    n = 9000
    X = rng.uniform(-30, 30, (n, 3))
    true = _pose(rng.normal(0, 0.2, 3), rng.normal(0, 1, 3))
    Y = X @ true[:, :3].T + true[:, 3] + rng.normal(0, 0.05, (n, 3))
    Y[rng.choice(n, 900, replace=False)] += rng.normal(0, 4.0, (900, 3))
    start = _pose(rng.normal(0, 0.1, 3), rng.normal(0, 0.5, 3))
    got = _run_pairs(fx, [(X, Y, start), (X[:100], Y[:100], start)], 20)
    _same_result(got[0], _oracle_pairs(X, Y, start, 20), "9000 pairs")
    _same_result(got[1], _oracle_pairs(X[:100], Y[:100], start, 20), "100 pairs beside them")
    # (2) a plane z = -2 as both maps; scan points a little above it
    plane = np.zeros((4000, 4), np.float32)
    plane[:, :2] = rng.uniform(-20, 20, (4000, 2))
    plane[:, 2] = -2.0
    scan_surface = np.zeros((300, 4), np.float32)
    scan_surface[:, :2] = rng.uniform(-15, 15, (300, 2))
    scan_surface[:, 2] = -1.95
    emap, smap = fx.make_map_from_host(plane, 1.0), fx.make_map_from_host(plane, 1.0)
    ident = _pose([0, 0, 0], [0, 0, 0])
    none = np.zeros((0, 4), np.float32)
    for leaf in (1000.0, 0.5):                              # a handful of rows, then hundreds: D = sum J^T J has rank 3 either way
        r = fx.localize_host(emap, smap, none, scan_surface, ident, 15, 20, leaf)
        w = _oracle_scan(plane, plane, 15, none, _downsample(scan_surface, leaf), ident, 20)
        assert (r["code"], r["iteration"]) == (w["code"], w["iteration"]) == (0, 0), (leaf, r, w)
        assert np.array_equal(r["pose"], ident) and r["success"]
    emap.close()
    smap.close()
    # a plane THROUGH THE ORIGIN cannot be written as w . x = -1 (surface.hpp:78-83): X has a zero column, R a zero pivot.  The
    # reference solves all the same (math.hpp:39: householderQr().solve, no rank check), gets NaN rows and runs to its iteration
    # limit with a NaN pose: a FAILURE.  What Eigen hands back there is not available here; such rows are zero rows (weight 0,
    # tests/test_residuals_gpu.py), and a scan whose surface rows are ALL zero rows ends with a failure of its own kind --
    # LFX_ALIGN_NO_PLANE, pose untouched -- in the library and in the oracle: not "converged" (a caller that gates on `success`
    # must not accept a pose no row supported)
    plane[:, 2] = 0.0
    scan_surface[:, 2] = 0.05
    emap, smap = fx.make_map_from_host(plane, 1.0), fx.make_map_from_host(plane, 1.0)
    r = fx.localize_host(emap, smap, none, scan_surface, ident, 15, 7, 1000.0)
    w = _oracle_scan(plane, plane, 15, none, _downsample(scan_surface, 1000.0), ident, 7)
    assert (r["code"], r["iteration"], r["success"]) == (w["code"], w["iteration"], w["success"]) == (5, 0, False), (r, w)
    assert r["message"] == "No surface neighbourhood spans a plane"
    assert np.array_equal(r["pose"], ident) and np.array_equal(w["pose"], ident)
    # (3) nothing at all: EmptyInput
    r = fx.localize_host(emap, smap, none, none, ident)
    assert (r["code"], r["iteration"], r["success"], r["error"]) == (4, 0, False, 0.0) and r["message"] == "The input data is empty"
    emap.close()
    smap.close()
    fx.close()
