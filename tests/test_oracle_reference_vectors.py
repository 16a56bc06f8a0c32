"""Pins the CPU oracle (oracle/lfx_oracle.cpp) against the reference's own unit-test vectors
(tests/golden/reference_unit_vectors.json, restated from /root/reference/extraction/test/*.cpp)
and, where the reference's own translation units build here (oracle/_ref), against that
compiled reference code bit-for-bit on random inputs.  CPU only."""
import ctypes as C
import math

import numpy as np
import pytest

from oracle import binding as B

L = B.lib()
LAB = B.LABEL
PD, PI, PF, PU8 = (C.POINTER(t) for t in (C.c_double, C.c_int, C.c_float, C.c_uint8))


def d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def names(labels):
    return [B.LABEL_NAMES[v] for v in labels]


def xy(points):
    p = np.asarray(points, dtype=np.float32).reshape(-1, 2)
    return np.ascontiguousarray(p[:, 0]), np.ascontiguousarray(p[:, 1])


def test_make_weight(refvec):
    for c in refvec["make_weight"]["cases"]:
        out = np.zeros(2 * c["padding"] + 1)
        L.orc_make_weight(c["padding"], B.ptr(out, PD))
        assert out.tolist() == c["expect"]


def test_calc_curvature(refvec):
    for c in refvec["calc_curvature"]["cases"]:
        r = d(c["range"])
        out = np.full(len(r), -1.0)
        assert L.orc_calc_curvature(B.ptr(r, PD), len(r), c["padding"], B.ptr(out, PD)) == 0
        assert out.tolist() == c["expect"]


def test_convolution1d(refvec):
    for c in refvec["convolution1d"]["cases"]:
        i, w = d(c["input"]), d(c["weight"])
        out = np.full(len(i), -7.0)
        rc = L.orc_convolution1d(B.ptr(i, PD), len(i), B.ptr(w, PD), len(w), B.ptr(out, PD))
        if c.get("throws"):
            assert rc == 1
        else:
            assert rc == 0 and out.tolist() == c["expect"]


def test_math(refvec):
    for c in refvec["xy_norm"]["cases"]:
        assert L.orc_xy_norm(c["x"], c["y"]) == c["expect"]
    tol = refvec["calc_radian"]["tolerance"]
    for c in refvec["calc_radian"]["cases"]:
        out = C.c_double(0)
        rc = L.orc_calc_radian(*c["args"], C.byref(out))
        if c.get("throws"):
            assert rc == 1
        else:
            assert rc == 0 and abs(out.value - c["expect_pi_times"] * math.pi) < tol
    for c in refvec["inner_product"]["cases"]:
        a, b = d(c["a"]), d(c["b"])
        assert L.orc_inner_product(B.ptr(a, PD), B.ptr(b, PD), len(a)) == c["expect"]


def test_argsort(refvec):
    for c in refvec["argsort"]["cases"]:
        v = d(c["values"])
        out = np.zeros(len(v), np.int32)
        L.orc_argsort(B.ptr(v, PD), len(v), B.ptr(out, PI))
        assert out.tolist() == c["expect"]


def test_index_range(refvec):
    for c in refvec["index_range"]["cases"]:
        out = np.zeros(c["n_blocks"] + 1, np.int32)
        rc = L.orc_index_range(c["start"], c["end"], c["n_blocks"], B.ptr(out, PI))
        if c.get("throws"):
            assert rc == 1
        else:
            assert rc == 0 and out.tolist() == c["bounds"]
    for c in refvec["padded_index_range"]["cases"]:
        out = np.zeros(c["n_blocks"] + 1, np.int32)
        assert L.orc_padded_index_range(c["size"], c["n_blocks"], c["padding"], B.ptr(out, PI)) == 0
        assert out.tolist() == c["bounds"]


def test_range(refvec):
    for c in refvec["is_in_inclusive_range"]["cases"]:
        assert bool(L.orc_is_in_inclusive_range(c["v"], c["min"], c["max"])) == c["expect"]
    for x, y in refvec["range"]["points_xy"]:
        assert abs(L.orc_xy_norm(np.float32(x), np.float32(y)) - math.sqrt(x * x + y * y)) < refvec["range"]["tolerance"]


def test_out_of_range(refvec):
    for c in refvec["out_of_range"]["cases"]:
        x, y = xy(c["points_xy"])
        lab = np.zeros(len(x), np.uint8)
        L.orc_label_out_of_range(B.ptr(lab, PU8), len(x), B.ptr(x, PF), B.ptr(y, PF), c["min_range"], c["max_range"])
        assert names(lab) == c["expect"]


def test_parallel_beam(refvec):
    for c in refvec["parallel_beam"]["cases"]:
        x, y = xy(c["points_xy"])
        lab = np.zeros(len(x), np.uint8)
        L.orc_label_parallel_beam(B.ptr(lab, PU8), len(x), B.ptr(x, PF), B.ptr(y, PF), c["threshold"])
        assert names(lab) == c["expect"]


@pytest.mark.parametrize("which", ["fill_from_left", "fill_from_right"])
def test_fill_directional(refvec, which):
    fn = getattr(L, "orc_" + which)
    for c in refvec[which]["cases"]:
        g = np.asarray(c["groups"], np.int32)
        lab = np.zeros(len(g), np.uint8)
        rc = fn(B.ptr(lab, PU8), len(g), B.ptr(g, PI), None, None, 0.0, c["begin"], c["end"], LAB[c["label"]])
        if c.get("throws"):
            assert rc == 1
        else:
            assert rc == 0 and names(lab) == c["expect"]


def test_fill_neighbors(refvec):
    for c in refvec["fill_neighbors"]["cases"]:
        g = np.asarray(c["groups"], np.int32)
        lab = np.zeros(len(g), np.uint8)
        rc = L.orc_fill_neighbors(B.ptr(lab, PU8), len(g), B.ptr(g, PI), None, None, 0.0, c["index"], c["padding"],
                                  LAB["EdgeNeighbor"])
        assert rc == 0 and names(lab) == c["expect"]


def test_edge_label(refvec):
    for c in refvec["edge_label"]["cases"]:
        g = np.asarray(c["groups"], np.int32)
        cv = d(c["curvature"])
        lab = np.zeros(len(g), np.uint8)
        rc = L.orc_edge_label_assign(B.ptr(lab, PU8), B.ptr(cv, PD), len(g), B.ptr(g, PI), None, None, 0.0,
                                     c["padding"], c["threshold"])
        assert rc == 0 and names(lab) == c["expect"]


def test_occlusion(refvec):
    o = refvec["occlusion"]
    for c in o["cases"]:
        x, y = xy(c["points_xy"])
        lab = np.zeros(len(x), np.uint8)
        rc = L.orc_label_occluded(B.ptr(lab, PU8), len(x), B.ptr(x, PF), B.ptr(y, PF), o["neighbor_radian_threshold"],
                                  c["padding"], o["distance_threshold"])
        assert rc == 0 and names(lab) == c["expect"]


def test_neighbor(refvec):
    for c in refvec["is_neighbor_xy"]["cases"]:
        thr = c["threshold"] if "threshold" in c else math.pi / 2 + c["threshold_pi_half_plus"]
        out = C.c_int(0)
        assert L.orc_is_neighbor_xy(*c["p0"], *c["p1"], thr, C.byref(out)) == 0
        assert bool(out.value) == c["expect"]
    n = refvec["neighbor_check_xy"]
    pts = n["points_xy"]
    for c in n["cases"]:
        thr = c["threshold"] if "threshold" in c else math.pi / 4 + c["threshold_pi_quarter_plus"]
        sl = pts[c["slice"][0]:c["slice"][1]] if "slice" in c else pts
        i, j = c["pair"]
        out = C.c_int(0)
        assert L.orc_is_neighbor_xy(*sl[i], *sl[j], thr, C.byref(out)) == 0
        assert bool(out.value) == c["expect"]
    # a 1-point checker throws (neighbor.hpp:71-75): reached through a stage that builds one
    x, y = xy(n["too_few_points_throws"]["points_xy"])
    lab = np.zeros(1, np.uint8)
    assert L.orc_label_occluded(B.ptr(lab, PU8), 1, B.ptr(x, PF), B.ptr(y, PF), 0.0, 1, 1.0) == 1


def test_polar_less(refvec):
    for a, b in refvec["polar_less_specific"]["pairs"]:
        want = math.atan2(a[1], a[0]) < math.atan2(b[1], b[0])
        assert bool(L.orc_polar_less_f64(a[0], a[1], b[0], b[1])) == want, (a, b)
        assert bool(L.orc_polar_less_f32(a[0], a[1], b[0], b[1])) == want, (a, b)
    n = refvec["polar_less_random"]["n"]
    rng = np.random.default_rng(0)
    p = rng.uniform(-1.0, 1.0, size=(n, 2))
    x, y = d(p[:, 0]), d(p[:, 1])
    idx = np.arange(n, dtype=np.int32)
    L.orc_sort_by_atan2_f64(B.ptr(x, PD), B.ptr(y, PD), n, B.ptr(idx, PI))
    want = np.argsort(np.arctan2(y, x), kind="stable")
    assert np.array_equal(idx, want)


def test_sort_by_atan2_and_rings(refvec):
    s = refvec["sort_by_atan2"]
    p = np.asarray(s["points_xy"], np.float64)
    idx = np.arange(len(p), dtype=np.int32)
    L.orc_sort_by_atan2_f64(B.ptr(d(p[:, 0]), PD), B.ptr(d(p[:, 1]), PD), len(p), B.ptr(idx, PI))
    assert idx.tolist() == s["expect"]

    e = refvec["extract_angle_sorted_rings"]
    pts = np.zeros(len(e["points_ring_xy"]), B.POINT_DTYPE)
    for k, (r, px, py) in enumerate(e["points_ring_xy"]):
        pts[k]["ring"], pts[k]["x"], pts[k]["y"] = r, px, py
    # padding 1 keeps every ring (>= 2 points); the projection does not depend on it
    prm = B.default_params()
    prm.padding = 1
    out = B.extract(pts, prm, canonical_ties=False)
    off = 0
    for rid, cnt in zip(out["ring_id"], out["ring_count"]):
        assert out["sorted_index"][off:off + cnt].tolist() == e["expect"][str(rid)]
        off += cnt

    rs = refvec["remove_sparse_rings"]
    pts = []
    for rid, size in rs["ring_sizes"].items():
        for k in range(size):
            pts.append((int(rid), math.cos(0.1 * k + 0.05), math.sin(0.1 * k + 0.05)))
    cloud = np.zeros(len(pts), B.POINT_DTYPE)
    for k, (r, px, py) in enumerate(pts):
        cloud[k]["ring"], cloud[k]["x"], cloud[k]["y"] = r, px, py
    for c in rs["cases"]:
        prm = B.default_params()
        prm.padding = c["n_min_points"] - 1      # RemoveSparseRings(rings, padding + 1), feature_extraction.cpp:116
        out = B.extract(cloud, prm)
        kept = [int(r) for r, s in zip(out["ring_id"], out["ring_status"]) if s != 1]
        assert kept == c["kept"]


def test_append_xyzir_narrowing(refvec):
    """label.hpp:166-179: intensity <- (float)curvature; exercised through the whole-scan output."""
    a = refvec["append_xyzir"]
    for (x, y, z, _i, _r), c, want in zip(a["points_xyzir"], a["curvature"], a["expect_xyzir"]):
        assert [np.float32(x), np.float32(y), np.float32(z), np.float32(c)] == want[:4]


def test_label_to_color(refvec):
    for name, rgb in refvec["label_to_color"]["expect_rgb"].items():
        out = (C.c_uint8 * 3)()
        L.orc_label_to_color(LAB[name], out)
        assert list(out) == rgb


# ------------------------------------------------------------------ against compiled reference pieces
needs_ref = pytest.mark.skipif(B.ref_pieces() is None, reason="oracle/_ref not built (no /root/reference here)")


@needs_ref
def test_ref_pieces_bit_exact():
    R = B.ref_pieces()
    rng = np.random.default_rng(1)
    # XYNorm / CalcRadian on float-valued inputs, including degenerate ones
    v = rng.normal(0, 30, size=(20000, 4)).astype(np.float32).astype(np.float64)
    v[:50, :2] = 0.0
    v[50:100, 2:] = 0.0
    v[100:150] = 0.0
    v[150:200, 2:] = v[150:200, :2] * 3.0     # same direction: cos may round above 1 -> NaN
    for a in v:
        assert L.orc_xy_norm(a[0], a[1]) == R.ref_xy_norm(a[0], a[1])
        o1, o2 = C.c_double(0), C.c_double(0)
        r1 = L.orc_calc_radian(*a, C.byref(o1))
        r2 = R.ref_calc_radian(*a, C.byref(o2))
        assert r1 == r2
        if r1 == 0:
            assert np.float64(o1.value).tobytes() == np.float64(o2.value).tobytes()
    # Convolution1D with curvature weights: bit-exact accumulation order
    for pad in (1, 2, 3, 5, 8):
        w = np.zeros(2 * pad + 1)
        L.orc_make_weight(pad, B.ptr(w, PD))
        for n in (2 * pad, 2 * pad + 1, 64, 257):
            r = np.abs(rng.normal(20, 15, n))
            o1, o2 = np.zeros(n), np.zeros(n)
            rc1 = L.orc_convolution1d(B.ptr(r, PD), n, B.ptr(w, PD), len(w), B.ptr(o1, PD))
            rc2 = R.ref_convolution1d(B.ptr(r, PD), n, B.ptr(w, PD), len(w), B.ptr(o2, PD))
            assert rc1 == rc2
            if rc1 == 0:
                assert o1.tobytes() == o2.tobytes()
                c = np.zeros(n)
                assert L.orc_calc_curvature(B.ptr(r, PD), n, pad, B.ptr(c, PD)) == 0
                assert c.tobytes() == (o2 * o2).tobytes()
    # PaddedIndexRange boundaries over a sweep of sizes
    for pad in (1, 2, 5):
        for nb in (1, 3, 6, 7):
            for size in list(range(2 * pad, 2 * pad + 40)) + [900, 1800, 2048, 4001]:
                b1 = np.zeros(nb + 1, np.int32)
                b2 = np.zeros(nb + 1, np.int32)
                rc1 = L.orc_padded_index_range(size, nb, pad, B.ptr(b1, PI))
                rc2 = R.ref_padded_index_range(size, nb, pad, B.ptr(b2, PI))
                assert rc1 == rc2
                if rc1 == 0:
                    assert b1.tolist() == b2.tolist()


# ------------------------------------------------------------------ helpers and error texts
KINDS = {"LargerThanOrEqualTo": 0, "SmallerThanOrEqualTo": 1, "LargerThan": 2, "SmallerThan": 3}


def _text(fn, *args):
    buf = C.create_string_buffer(256)
    n = fn(*args, buf, 256)
    assert 0 <= n < 256
    return buf.value.decode()


def test_range_message(refvec):
    for c in refvec["range_message"]["cases"]:
        got = _text(L.orc_range_message, KINDS[c["kind"]], c["value_name"].encode(), c["range_name"].encode(), c["value"], c["range"])
        assert got == c["expect"]


def test_irange(refvec):
    for c in refvec["irange"]["cases"]:
        out = np.full(c["size"], -1, np.int32)
        L.orc_irange(c["size"], B.ptr(out, PI))
        assert out.tolist() == c["expect"]


def test_mapped_points(refvec):
    m = refvec["mapped_points"]
    y, idx = d(m["cloud_y"]), np.ascontiguousarray(m["indices"], dtype=np.int32)

    def view(begin, end):
        out, size, vals = C.c_double(0), C.c_int(0), []
        k = 0
        while L.orc_mapped_points_at(B.ptr(y, PD), len(y), B.ptr(idx, PI), len(idx), begin, end, k, C.byref(out), C.byref(size)) == 0:
            vals.append(out.value)
            k += 1
        return size.value, vals

    assert view(0, len(idx)) == (m["expect_size"], m["expect_y"])
    s = m["slice"]
    assert view(s["begin"], s["end"]) == (s["expect_size"], s["expect_y"])


def test_throw_texts_of_abandoned_rings(refvec):
    """The text of the exception a ring is abandoned with, for the ring's own numbers; the convolution text is the
    one test_convolution.cpp:61-70 pins."""
    t = refvec["throw_texts"]
    p = B.default_params()                                   # P = 5, B = 6
    assert _text(L.orc_ring_message, 2, 9, C.byref(p)) == t["too_few_for_convolution"].format(n=9, m=11)
    assert _text(L.orc_ring_message, 3, 14, C.byref(p)) == t["too_few_for_blocks"].format(d=4, b=6)
    assert _text(L.orc_ring_message, 4, 17, C.byref(p)) == t["block_too_small"].format(n=1)
    assert _text(L.orc_ring_message, 5, 400, C.byref(p)) == t["zero_norm_pair"]
    assert _text(L.orc_ring_message, 1, 3, C.byref(p)) == ""
    p3 = B.Params(1, 2.0, 0.3, 0.02, 0.05, 0.05, 0.1, 100.0, 6)
    conv = [c for c in refvec["convolution1d"]["cases"] if c.get("throws")][0]
    assert _text(L.orc_ring_message, 2, len(conv["input"]), C.byref(p3)) == conv["message"]


# ------------------------------------------------------------------ voxel-grid Downsample (parity unpinned)
def test_voxel_downsample_known_answers():
    """Downsample = pcl::VoxelGrid with one leaf size (downsample.hpp:37-51).  PCL is not in the image and the
    reference holds no test of it: these are hand-computed cases of the published algorithm (cells of `leaf` metres
    anchored at multiples of the leaf, one centroid per occupied cell, cells in ascending x-fastest index order)."""
    pts = np.array([[0.1, 0.1, 0.1, 1], [0.3, 0.5, 0.9, 1],          # cell (0, 0, 0)
                    [1.5, 0.2, 0.2, 1],                               # cell (1, 0, 0)
                    [0.2, 1.2, 0.1, 1], [0.4, 1.4, 0.3, 1], [0.6, 1.9, 0.5, 1],   # cell (0, 1, 0)
                    [-0.5, 0.5, 0.5, 1],                              # cell (-1, 0, 0): the grid's origin moves to it
                    [0.5, 0.5, 2.5, 1]], np.float32)                  # cell (0, 0, 2)
    out = np.zeros_like(pts)
    n_out = C.c_int(0)
    assert L.orc_voxel_downsample(B.ptr(pts, PF), len(pts), 1.0, B.ptr(out, PF), C.byref(n_out)) == 0
    f = np.float32
    want = [[-0.5, 0.5, 0.5, 1], [(f(0.1) + f(0.3)) / f(2), (f(0.1) + f(0.5)) / f(2), (f(0.1) + f(0.9)) / f(2), 1], [1.5, 0.2, 0.2, 1],
            [(f(0.2) + f(0.4) + f(0.6)) / f(3), (f(1.2) + f(1.4) + f(1.9)) / f(3), (f(0.1) + f(0.3) + f(0.5)) / f(3), 1],
            [0.5, 0.5, 2.5, 1]]
    assert n_out.value == 5
    assert out[:5].tolist() == np.array(want, np.float32).tolist()
    # a leaf far too small for the extent: PCL warns and gives the cloud back; here: status 1
    far = np.array([[0, 0, 0, 1], [4000, 4000, 4000, 1]], np.float32)
    assert L.orc_voxel_downsample(B.ptr(far, PF), 2, 0.001, B.ptr(out, PF), C.byref(n_out)) == 1
    # properties on a random cloud: every point lies in the cell of exactly one centroid, counts add up
    rng = np.random.default_rng(3)
    cloud = np.ones((5000, 4), np.float32)
    cloud[:, :3] = rng.uniform(-20, 20, (5000, 3)).astype(np.float32)
    out = np.zeros_like(cloud)
    assert L.orc_voxel_downsample(B.ptr(cloud, PF), len(cloud), 2.0, B.ptr(out, PF), C.byref(n_out)) == 0
    cells = np.floor(cloud[:, :3] * np.float32(0.5)).astype(np.int64)
    uniq = np.unique(cells, axis=0)
    assert n_out.value == len(uniq)
    got_cells = np.floor(out[:n_out.value, :3] * np.float32(0.5)).astype(np.int64)
    assert len(np.unique(got_cells, axis=0)) == n_out.value
    order = np.lexsort((uniq[:, 0], uniq[:, 1], uniq[:, 2]))             # x fastest, then y, then z
    assert np.array_equal(got_cells, uniq[order])
