"""Pins the oracle of the consumer's first step (oracle/lfx_oracle_loc.cpp: the scan-to-map residual build of the
reference's localization package) against the vectors of localization/test/test_edge.cpp and
test_math.cpp (restated in tests/golden/reference_unit_vectors.json) and against numpy's own eigen-decomposition and
least squares.  Beyond those vectors parity is unpinned (Eigen and nanoflann are not in the image).  CPU only."""
import ctypes as C

import numpy as np

from oracle import binding as B

L = B.lib()
PD, PF = C.POINTER(C.c_double), C.POINTER(C.c_float)


def d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def test_triplet_cross(refvec):
    for c in refvec["loc_triplet_cross"]["cases"]:
        out = np.zeros(3)
        L.orc_loc_triplet_cross(B.ptr(d(c["p0"]), PD), B.ptr(d(c["p1"]), PD), B.ptr(d(c["p2"]), PD), B.ptr(out, PD))
        assert out.tolist() == c["expect"]


def test_center_and_covariance(refvec):
    X = d(refvec["loc_center"]["X"])
    mean, cov = np.zeros(3), np.zeros(9)
    L.orc_loc_mean_cov(B.ptr(X, PD), len(X), B.ptr(mean, PD), B.ptr(cov, PD))
    assert mean.tolist() == refvec["loc_center"]["expect"]
    X = d(refvec["loc_mean_cov"]["X"])
    L.orc_loc_mean_cov(B.ptr(X, PD), len(X), B.ptr(mean, PD), B.ptr(cov, PD))
    assert (cov.reshape(3, 3) * 4).tolist() == refvec["loc_mean_cov"]["expect_cov_times_4"]      # EXPECT_EQ(norm, 0)


def test_principal_components(refvec):
    rng = np.random.default_rng(1)
    ev, V = np.zeros(3), np.zeros(9)
    for _ in range(200):
        A = rng.uniform(-1, 1, (3, 3))
        Cm = d(A @ A.T)
        L.orc_loc_principal(B.ptr(Cm, PD), B.ptr(ev, PD), B.ptr(V, PD))
        Vm = V.reshape(3, 3)
        assert ev[0] <= ev[1] <= ev[2]                                                   # test_edge.cpp:67-68
        assert np.linalg.norm(Cm - Vm @ np.diag(ev) @ np.linalg.inv(Vm)) <= 1e-4          # :73
        assert np.linalg.norm(Cm @ Vm[:, 2] - ev[2] * Vm[:, 2]) <= 1e-4                   # :77
        w, U = np.linalg.eigh(Cm)
        assert np.allclose(ev, w, rtol=1e-12, atol=1e-14)
        assert abs(abs(Vm[:, 2] @ U[:, 2]) - 1.0) < 1e-9
    line = refvec["loc_principal_line"]
    X = d(line["X"])
    mean, cov = np.zeros(3), np.zeros(9)
    L.orc_loc_mean_cov(B.ptr(X, PD), len(X), B.ptr(mean, PD), B.ptr(cov, PD))
    L.orc_loc_principal(B.ptr(cov, PD), B.ptr(ev, PD), B.ptr(V, PD))
    u = V.reshape(3, 3)[:, 2]
    assert min(np.linalg.norm(u - line["expect_direction"]), np.linalg.norm(u + line["expect_direction"])) <= line["tolerance"]
    assert abs(ev[0]) <= 1e-8 and abs(ev[1]) <= 1e-8 and ev[2] > 0
    for c in refvec["loc_principal_is_reliable"]["cases"]:
        assert bool(L.orc_loc_principal_is_reliable(B.ptr(d(c["ev"]), PD))) == c["expect"]


def test_solve_linear(refvec):
    for c in refvec["loc_solve_linear"]["cases"]:
        A, b = d(c["A"]), d(c["b"])
        x = np.zeros(A.shape[1])
        L.orc_loc_solve_linear(B.ptr(A, PD), A.shape[0], A.shape[1], B.ptr(b, PD), B.ptr(x, PD))
        assert np.linalg.norm(x - c["expect"]) <= refvec["loc_solve_linear"]["tolerance"]
    rng = np.random.default_rng(2)
    for _ in range(100):
        A, b = d(rng.standard_normal((15, 3))), d(rng.standard_normal(15))
        x = np.zeros(3)
        L.orc_loc_solve_linear(B.ptr(A, PD), 15, 3, B.ptr(b, PD), B.ptr(x, PD))
        assert np.allclose(x, np.linalg.lstsq(A, b, rcond=None)[0], rtol=1e-10, atol=1e-12)


def _pose(rng):
    q = rng.standard_normal(4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    return q, np.ascontiguousarray(np.hstack([R, rng.uniform(-2, 2, (3, 1))]))


def test_quaternion_of_a_rotation():
    rng = np.random.default_rng(3)
    for _ in range(200):
        q, pose = _pose(rng)
        out = np.zeros(4)
        R = np.ascontiguousarray(pose[:, :3])
        L.orc_loc_quaternion(B.ptr(R, PD), B.ptr(out, PD))
        assert min(np.linalg.norm(out - q), np.linalg.norm(out + q)) < 1e-12


def test_edge_jacobian_approximates_the_residual(refvec):
    """test_edge.cpp:144-192 (ApproximateError): r(theta + d) ~ r(theta) + J M d; here by central differences on the
    quaternion and the translation, which is what the 3 x 7 row [K DRpDq, K] differentiates."""
    rng = np.random.default_rng(4)
    line = np.zeros((40, 4), np.float32)
    line[:, 0] = np.linspace(0, 4, 40)
    line[:, 1:3] = rng.normal(0, 0.01, (40, 2))
    pts = np.array([[2, 1, 0, 1], [1.0, -0.5, 0.3, 1]], np.float32)
    q, pose = _pose(rng)
    res, jac = np.zeros((2, 3)), np.zeros((2, 3, 7))
    L.orc_loc_edge_residuals(B.ptr(line, PF), len(line), B.ptr(pose, PD), 5, B.ptr(pts, PF), 2, B.ptr(res, PD), B.ptr(jac, PD))
    eps = 1e-6
    for a in range(3):                                          # translation columns 4..6
        dp = pose.copy()
        dp[a, 3] += eps
        r1 = np.zeros((2, 3))
        j1 = np.zeros((2, 3, 7))
        L.orc_loc_edge_residuals(B.ptr(line, PF), len(line), B.ptr(dp, PD), 5, B.ptr(pts, PF), 2, B.ptr(r1, PD), B.ptr(j1, PD))
        assert np.allclose((r1 - res) / eps, jac[:, :, 4 + a], atol=1e-4)
    assert np.all(np.isfinite(jac)) and np.linalg.norm(res) > 0


def test_residuals_of_points_on_the_map_vanish():
    """A scan point that lies on the map's line / plane has a zero residual (the geometry of edge.cpp:76-84 and
    surface.hpp:40-44), whatever the pose that puts it there."""
    rng = np.random.default_rng(5)
    q, pose = _pose(rng)
    R, t = pose[:, :3], pose[:, 3]
    line = np.zeros((60, 4), np.float32)
    line[:, 0] = np.linspace(-3, 3, 60)
    plane = np.zeros((400, 4), np.float32)
    gx, gy = np.meshgrid(np.linspace(-2, 2, 20), np.linspace(-2, 2, 20))
    plane[:, 0], plane[:, 1], plane[:, 2] = gx.ravel(), gy.ravel(), 1.5
    on_line = np.array([[0.33, 0, 0], [-1.21, 0, 0]])
    on_plane = np.array([[0.4, -0.3, 1.5], [1.1, 0.9, 1.5]])
    for world, cloud, fn, shape in ((on_line, line, L.orc_loc_edge_residuals, (2, 3)), (on_plane, plane, L.orc_loc_surface_residuals, (2,))):
        local = np.ones((2, 4), np.float32)
        local[:, :3] = ((world - t) @ R).astype(np.float32)       # R^T (p - t)
        res = np.zeros(shape)
        jac = np.zeros((2, 21 if len(shape) == 2 else 7))
        fn(B.ptr(cloud, PF), len(cloud), B.ptr(pose, PD), 5, B.ptr(local, PF), 2, B.ptr(res, PD), B.ptr(jac, PD))
        assert np.all(np.abs(res) < 1e-5)


# ---- the optimizer around the rows (optimizer.hpp / src/optimizer.cpp, robust.cpp, degenerate.cpp, posevec.cpp) ----------
PI = C.POINTER(C.c_int)


def test_median_absolute_deviation_and_scale(refvec):
    for c in refvec["loc_mad"]["cases"]:
        v = d(c["v"])
        assert L.orc_loc_mad(B.ptr(v, PD), len(v)) == c["expect"]                        # EXPECT_EQ
        assert L.orc_loc_median(B.ptr(v, PD), len(v)) == float(np.median(v))
    s = refvec["loc_scale_normal"]
    e = np.random.default_rng(5).normal(s["mean"], s["stddev"], s["n"])
    nf = float(s["n"])
    assert abs(np.sqrt((nf - 1) / nf) * s["stddev"] - L.orc_loc_scale(B.ptr(e, PD), len(e))) <= s["tolerance"]


def test_huber_and_its_derivative(refvec):
    k = refvec["loc_huber"]["k"]
    for c in refvec["loc_huber"]["cases"]:
        assert L.orc_loc_huber(c["r"] * c["r"], k) == c["expect"]
    hd = refvec["loc_huber_derivative"]
    for c in hd["cases"]:
        num = (L.orc_loc_huber(c["e"] + c["h"], hd["k"]) - L.orc_loc_huber(c["e"], hd["k"])) / c["h"]
        assert abs(num - L.orc_loc_huber_derivative(c["e"], hd["k"])) < hd["tolerance"]


def test_is_degenerate(refvec):
    g = refvec["loc_is_degenerate"]
    Cm = d(g["C"])
    for c in g["cases"]:
        assert bool(L.orc_loc_is_degenerate(B.ptr(Cm, PD), 3, c["threshold"])) == c["expect"]
    rng = np.random.default_rng(2)
    for n in (3, 6, 7):
        for _ in range(50):
            A = rng.normal(size=(n, n))
            S = d(A + A.T)
            w = np.linalg.eigvalsh(S)
            thr = float(rng.uniform(0.05, 1.0))
            if np.min(np.abs(np.abs(w) - thr)) < 1e-9:
                continue
            assert bool(L.orc_loc_is_degenerate(B.ptr(S, PD), n, thr)) == bool((np.abs(w) < thr).any())


def _rotation(q):
    R = np.zeros(9)
    L.orc_loc_rotation_matrix(B.ptr(d(q), PD), B.ptr(R, PD))
    return R.reshape(3, 3)


def test_angle_axis_and_make_m(refvec):
    g = refvec["loc_angle_axis"]
    q = np.zeros(4)
    L.orc_loc_angle_axis_to_quaternion(B.ptr(d(g["cases"][0]["theta"]), PD), B.ptr(q, PD))
    assert np.linalg.norm(q) == 1.0 and q[0] == 1.0
    theta = d(g["cases"][1]["theta"])
    L.orc_loc_angle_axis_to_quaternion(B.ptr(theta, PD), B.ptr(q, PD))
    k = np.linalg.norm(theta)
    u = theta / k
    K = np.array([[0, -u[2], u[1]], [u[2], 0, -u[0]], [-u[1], u[0], 0]])
    E = np.eye(3) + np.sin(k) * K + (1 - np.cos(k)) * K @ K                              # AngleAxisd(k, u).toRotationMatrix()
    assert abs(np.linalg.norm(q) - 1.0) <= g["tolerance"] and np.linalg.norm(_rotation(q) - E) <= g["tolerance"]
    for c in refvec["loc_make_m"]["cases"]:
        M = np.zeros(42)
        L.orc_loc_make_m(B.ptr(d(c["q_wxyz"]), PD), B.ptr(M, PD))
        assert np.linalg.norm(M.reshape(7, 6) - d(c["expect"])) <= refvec["loc_make_m"]["tolerance"]


def _pose_of(q_unnormalised, t):
    q = d(q_unnormalised) / np.linalg.norm(q_unnormalised)
    return d(np.hstack([_rotation(q), d(t).reshape(3, 1)]).reshape(-1))


def run_pairs(X, Y, pose, max_iter):
    out, err, scale, it, code = np.zeros(12), C.c_double(), C.c_double(), C.c_int(), C.c_int()
    ok = L.orc_loc_optimize_pairs(B.ptr(d(X), PD), B.ptr(d(Y), PD), len(X), B.ptr(d(pose), PD), max_iter, B.ptr(out, PD),
                                  C.byref(err), C.byref(scale), C.byref(it), C.byref(code))
    return dict(pose=out.reshape(3, 4), error=err.value, scale=scale.value, iteration=it.value, code=code.value, success=bool(ok))


def test_optimizer_on_the_alignment_problem(refvec):
    """localization/test/test_optimizer.cpp:53-242 on the restated Optimizer::Run + AlignmentProblem."""
    g = refvec["loc_alignment"]
    true = _pose_of(g["q_true_wxyz_unnormalised"], g["t_true"]).reshape(3, 4)
    X = d(g["X"])
    Y = d(X @ true[:, :3].T + true[:, 3])
    # one CalcUpdate lowers the error (:73-96)
    one = g["one_update"]
    pose0 = _pose_of(one["q_wxyz_unnormalised"], one["t"])
    dq, dt = np.zeros(4), np.zeros(3)
    L.orc_loc_pairs_update(B.ptr(X, PD), B.ptr(Y, PD), len(X), B.ptr(pose0, PD), B.ptr(dq, PD), B.ptr(dt, PD))
    q0 = d(one["q_wxyz_unnormalised"]) / np.linalg.norm(one["q_wxyz_unnormalised"])
    w0, v0, w1, v1 = q0[0], q0[1:], dq[0], dq[1:]
    q1 = np.concatenate([[w0 * w1 - v0 @ v1], w0 * v1 + w1 * v0 + np.cross(v0, v1)])
    R0, R1 = pose0.reshape(3, 4)[:, :3], _rotation(q1)
    e0 = np.sum((X @ R0.T + d(one["t"]) - Y) ** 2)
    e1 = np.sum((X @ R1.T + d(one["t"]) + dt - Y) ** 2)
    assert e1 < e0
    b = g["bounds"]
    for run in g["runs"]:
        r = run_pairs(X, Y, _pose_of(run["q_wxyz_unnormalised"], run["t"]), g["max_iter"])
        assert r["success"] == b["success"], run["name"]
        if "iteration_eq" in run:
            assert r["iteration"] == run["iteration_eq"]
        else:
            assert r["iteration"] < run["iteration_lt"], (run["name"], r["iteration"])
        assert r["error"] < b["error_lt"] and r["scale"] < b["error_scale_lt"]
        assert np.linalg.norm(true[:, :3] - r["pose"][:, :3]) <= b["rotation_norm_le"]
        assert np.linalg.norm(true[:, 3] - r["pose"][:, 3]) <= b["translation_norm_le"]
    ident = d(np.hstack([np.eye(3), np.zeros((3, 1))]).reshape(-1))
    r = run_pairs(np.zeros((0, 3)), np.zeros((0, 3)), ident, 10)                           # ShouldReturnFalseForEmptyData
    e = g["empty"]
    assert (r["iteration"], r["success"], r["error"], r["scale"]) == (e["iteration"], e["success"], e["error"], e["error_scale"])
    nc = g["no_convergence"]                                                              # ShouldReturnFalseWhenNoConvergence
    rng = np.random.default_rng(3)
    r = run_pairs(rng.normal(*nc["x"], (nc["n"], 3)), rng.normal(*nc["y"], (nc["n"], 3)), ident, nc["max_iter"])
    assert r["iteration"] == nc["iteration"] and r["success"] == nc["success"]
    assert r["error"] > nc["error_gt"] and r["scale"] > nc["error_scale_gt"]
    # WeightedUpdate returns zero when D is degenerate (:313-328): one point pair leaves the rotation about it free
    dq, dt = np.zeros(4), np.zeros(3)
    L.orc_loc_pairs_update(B.ptr(d([[0, 0, 0]]), PD), B.ptr(d([[1, 1, 1]]), PD), 1, B.ptr(ident, PD), B.ptr(dq, PD), B.ptr(dt, PD))
    assert dq.tolist() == [1, 0, 0, 0] and dt.tolist() == [0, 0, 0]


def test_nearest_k_search(refvec):
    """localization/test/test_kdtree.cpp:37-77 on the restated exact search."""
    g = refvec["loc_kdtree"]
    pts = np.zeros((len(g["points"]), 4), np.float32)
    pts[:, :3] = g["points"]
    for c in g["cases"]:
        k = c["k"]
        X, dist, idx = np.zeros((k, 3)), np.zeros(k), np.zeros(k, np.int32)
        L.orc_loc_nearest(B.ptr(pts, PF), len(pts), B.ptr(d(g["query"]), PD), k, B.ptr(X, PD), B.ptr(dist, PD), B.ptr(idx, PI))
        assert X.tolist() == c["X"] and dist.tolist() == c["squared_distances"]


def _qmul(a, b):
    w0, v0, w1, v1 = a[0], np.asarray(a[1:]), b[0], np.asarray(b[1:])
    return np.concatenate([[w0 * w1 - v0 @ v1], w0 * v1 + w1 * v0 + np.cross(v0, v1)])


def test_drp_dq_and_left_multiplication(refvec):
    """rotationlib/test/test_jacobian_quaternion.cpp:67-82 (DRpDq against the quaternion-product form) and
    test_quaternion.cpp:40-48 (LeftMultiplicationMatrix, the matrix inside MakeM) on the restatement."""
    g = refvec["loc_drp_dq"]
    q = d(g["q_wxyz_unnormalised"]) / np.linalg.norm(g["q_wxyz_unnormalised"])
    p = d(g["p"])
    J = np.zeros(12)
    L.orc_loc_drp_dq(B.ptr(q, PD), B.ptr(p, PD), B.ptr(J, PD))

    def Q(a):
        w, x, y, z = a
        return np.array([[w, -x, -y, -z], [x, w, -z, y], [y, z, w, -x], [z, -y, x, w]])

    def P(a):
        w, x, y, z = a
        return np.array([[w, -x, -y, -z], [x, w, z, -y], [y, -z, w, x], [z, y, -x, w]])
    u = np.concatenate([[0.0], p])
    q_inv = np.concatenate([[q[0]], -q[1:]])                                   # unit quaternion
    D = Q(_qmul(q, u)) @ np.diag([1.0, -1.0, -1.0, -1.0]) + P(_qmul(u, q_inv))
    assert np.abs(D[1:] - J.reshape(3, 4)).max() <= 1e-14
    rng = np.random.default_rng(4)
    for _ in range(20):
        q1, q2 = rng.uniform(-1, 1, 4), rng.uniform(-1, 1, 4)
        M = np.zeros(42)
        L.orc_loc_make_m(B.ptr(d(q1), PD), B.ptr(M, PD))
        left_cols = 2.0 * M.reshape(7, 6)[:4, :3]                              # columns 1..3 of LeftMultiplicationMatrix(q1)
        assert np.abs(left_cols @ q2[1:] + q1 * q2[0] - _qmul(q1, q2)).max() <= 1e-12
