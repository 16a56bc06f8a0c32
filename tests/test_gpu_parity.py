"""Parity tests proper: the HIP path, called through the C ABI (ctypes), against the CPU oracle on
the same seeded inputs, plus the reference's own unit-test vectors run through the device stages.
All marked gpu; run on the MI355X box with `pytest -m gpu`."""
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from lidar_feature_extraction_amd import FeatureExtraction, HyperParameters, make_scan, POINT_DTYPE  # noqa: E402
from lidar_feature_extraction_amd import binding as LB, synth  # noqa: E402
from oracle import binding as OB  # noqa: E402
from tests.parity import assert_scan_equal, status_codes_equal_where_single_cause  # noqa: E402

LAB = OB.LABEL


def oracle_params(hp):
    return OB.Params(hp.padding, hp.neighbor_degree_threshold, hp.distance_diff_threshold,
                     hp.parallel_beam_min_range_ratio, hp.edge_threshold, hp.surface_threshold,
                     hp.min_range, hp.max_range, hp.n_blocks)


@pytest.fixture(scope="module")
def fx():
    f = FeatureExtraction(device=0, max_points_per_scan=262144, max_batch=8)
    yield f
    f.close()


def names(labels):
    return [OB.LABEL_NAMES[v] for v in labels]


def xy(points):
    p = np.asarray(points, dtype=np.float32).reshape(-1, 2)
    return np.ascontiguousarray(p[:, 0]), np.ascontiguousarray(p[:, 1])


# ------------------------------------------------------------------ whole scans vs the oracle
PARAM_SETS = {"defaults": HyperParameters(), "launch_yaml": HyperParameters.launch_yaml()}


@pytest.mark.parametrize("pname", list(PARAM_SETS))
@pytest.mark.parametrize("shape", [(16, 900, 15.0), (16, 1800, 15.0), (64, 1800, 15.0), (128, 2048, 22.5)])
def test_scan_parity(shape, pname):
    rings, cols, vfov = shape
    hp = PARAM_SETS[pname]
    f = FeatureExtraction(hp, device=0, max_points_per_scan=rings * cols, max_batch=2,
                          max_points_per_ring=cols, max_rings=rings)
    clouds = [make_scan(rings, cols, seed=1234 + i, vfov_deg=vfov) for i in range(2)]
    got = f.extract_batch(clouds)
    for i, c in enumerate(clouds):
        want = OB.extract(c, oracle_params(hp), canonical_ties=False)   # std::sort exactly as the reference
        assert want["angle_ties"] == 0 and want["curvature_ties"] == 0
        assert_scan_equal(got[i], want, "%dx%d/%s/scan%d" % (rings, cols, pname, i))
        assert len(got[i].edge_index) > 0 and len(got[i].surface_index) > 0
    f.close()


@pytest.mark.parametrize("variant", ["ragged", "shuffled", "rotated", "reversed", "reversed_rotated", "ragged_rotated",
                                     "ragged_shuffled"])
def test_scan_parity_input_orders(fx, variant):
    kw = {"ragged": dict(drop_fraction=0.13), "shuffled": dict(shuffle=True), "rotated": dict(start_col=517),
          "reversed": dict(reverse=True), "reversed_rotated": dict(reverse=True, start_col=333),
          "ragged_rotated": dict(drop_fraction=0.2, start_col=901),
          "ragged_shuffled": dict(drop_fraction=0.3, shuffle=True)}[variant]
    c = make_scan(32, 1024, seed=77, **kw)
    got = fx.ExtractFeatures(c)
    want = OB.extract(c, canonical_ties=False)
    assert_scan_equal(got, want, variant)


INPUT_ORDER_KW = {"sorted": dict(), "ragged": dict(drop_fraction=0.13), "shuffled": dict(shuffle=True), "rotated": dict(start_col=517),
                  "reversed": dict(reverse=True), "reversed_rotated": dict(reverse=True, start_col=333),
                  "ragged_rotated": dict(drop_fraction=0.2, start_col=901)}


@pytest.mark.parametrize("force", ["1", "0", None])
def test_order_repair_before_the_first_pass(force):
    """A stream that keeps arriving rotated / reversed gets its rings put in order BEFORE the unit kernel
    (ring_order_kernel over every ring) instead of after a failed first pass; the library switches that on from
    the repair counts of earlier batches (force=None: same context fed the same kind of batch three times, so the
    later calls take the other route), LFX_DEBUG_PRE_ORDER pins it.  Same results on every route."""
    import os
    clouds = {k: make_scan(16, 1024, seed=85, **kw) for k, kw in INPUT_ORDER_KW.items()}
    want = {k: OB.extract(c, canonical_ties=False) for k, c in clouds.items()}
    if force is not None:
        os.environ["LFX_DEBUG_PRE_ORDER"] = force
    try:
        f = FeatureExtraction(device=0, max_points_per_scan=16 * 1024, max_batch=4, max_points_per_ring=1024, max_rings=16)
    finally:
        os.environ.pop("LFX_DEBUG_PRE_ORDER", None)
    for k, c in clouds.items():
        for rep in range(3):
            got = f.extract_batch([c, clouds["sorted"], c])
            assert_scan_equal(got[0], want[k], "%s/pre-order %s/rep %d" % (k, force, rep))
            assert_scan_equal(got[1], want["sorted"], "sorted beside %s/pre-order %s/rep %d" % (k, force, rep))
            assert_scan_equal(got[2], want[k], "%s (second copy)/pre-order %s/rep %d" % (k, force, rep))
    f.close()


def test_second_pass_capacity_overflow_goes_to_the_slow_path():
    """The second unit pass is launched for a number of rings estimated from earlier batches; rings beyond it are
    handed to the workgroup-per-ring kernel by the order-repair kernel.  Forced here to 3 rings."""
    import os
    c = make_scan(16, 1024, seed=86, start_col=300)
    want = OB.extract(c, canonical_ties=False)
    os.environ["LFX_DEBUG_REDO_CAP"] = "3"
    os.environ["LFX_DEBUG_PRE_ORDER"] = "0"
    try:
        f = FeatureExtraction(device=0, max_points_per_scan=16 * 1024, max_batch=2, max_points_per_ring=1024, max_rings=16)
    finally:
        del os.environ["LFX_DEBUG_REDO_CAP"], os.environ["LFX_DEBUG_PRE_ORDER"]
    for rep in range(2):
        got = f.extract_batch([c, c])
        assert_scan_equal(got[0], want, "redo cap 3, rep %d" % rep)
        assert_scan_equal(got[1], want, "redo cap 3, second scan, rep %d" % rep)
    f.close()


def _with_env(env, make):
    import os
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return make()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("force", ["1", "0", None, "short-tail", "transforms"])
def test_organised_scan_kernel_and_its_fall_back(force):
    """A context that knows the sensor's ring count reads a driver's column-major scan directly (organised-scan
    kernel: no bucketing pass); every scan that is not of that form -- ragged, rotated, reversed, shuffled, ring ids
    that do not follow the column pattern, a zero-norm pair, a ring count that is not the sensor's -- falls back to
    the bucketing route inside the same call.  Mixed batches, both routes pinned (LFX_DEBUG_FUSED) and the library's
    own choice (None: the same context sees three kinds of stream); same results everywhere."""
    R, Ccols = 12, 700                     # 12 rings: three groups of four; 10-ring scans below leave a group half empty
    clouds = {k: make_scan(R, Ccols, seed=300 + i, **kw) for i, (k, kw) in enumerate(INPUT_ORDER_KW.items())}
    swapped = make_scan(R, Ccols, seed=320)
    col = swapped[R * 100:R * 101].copy()
    swapped[R * 100], swapped[R * 100 + 1] = col[1], col[0]            # two records of one column change places
    clouds["swapped_in_column"] = swapped
    pair = make_scan(R, Ccols, seed=321)
    k = np.nonzero(pair["ring"] == 5)[0]
    for a in (k[200], k[201]):
        pair["x"][a] = 0.0
        pair["y"][a] = 0.0
    clouds["zero_norm_pair"] = pair                                    # ring 5 is abandoned (math.cpp:40-42), the others are not
    clouds["fewer_rings"] = make_scan(10, Ccols, seed=322)             # 7000 points: not a multiple of 12
    clouds["other_shape_same_count"] = make_scan(6, 2 * Ccols, seed=323)   # 8400 points = 12 x 700, ring ids 0..5 only
    clouds["short"] = make_scan(R, 9, seed=324, spikes=False)                        # 9 columns: too few for the convolution
    want = {k: OB.extract(c, canonical_ties=False) for k, c in clouds.items()}
    want["zero_norm_pair"] = OB.extract(pair, canonical_ties=True)      # the two (0, 0) points tie under the angle predicate
    assert want["zero_norm_pair"]["ring_status"].tolist().count(0) == R - 1
    # "short-tail": the form the bucketing route takes while no scan has been falling back (bucketing, then the
    # workgroup-per-ring kernel over every ring of the scans that did), pinned for every batch
    # "transforms": every ring's rotation / reversal is found first (ring_cut_kernel) and applied in the kernel's loads
    env = {} if force is None else ({"LFX_DEBUG_FUSED": "1", "LFX_DEBUG_SHORT_TAIL": "1"} if force == "short-tail" else
                                    {"LFX_DEBUG_FUSED": "1", "LFX_DEBUG_XFORM": "1"} if force == "transforms" else {"LFX_DEBUG_FUSED": force})
    f = _with_env(env, lambda: FeatureExtraction(device=0, max_points_per_scan=R * 2 * Ccols, max_batch=6,
                                                 max_points_per_ring=2 * Ccols, max_rings=R))
    for name, c in clouds.items():
        for rep in range(3):
            got = f.extract_batch([c, clouds["sorted"], c, clouds["sorted"], clouds["sorted"]])
            routes = f.scan_routes(5)
            if force == "transforms":
                # rotated / reversed rings stay on the organised route; what is not R x C in ring order does not
                turned = name in ("rotated", "reversed", "reversed_rotated")
                assert routes.tolist() == ([2, 2, 2, 2, 2] if turned or name in ("sorted",) else [0, 2, 0, 2, 2]), (name, routes)
            if force == "0":
                assert not routes.any()
            for i, key in enumerate([name, "sorted", name, "sorted", "sorted"]):
                assert_scan_equal(got[i], want[key], "%s[%d]/fused %s/rep %d%s" % (key, i, force, rep, "[ties]" if want[key]["angle_ties"] else ""))
    f.close()


@pytest.mark.parametrize("kind", ["rotated", "reversed", "reversed_rotated"])
def test_stream_of_turned_rings_stays_on_the_organised_route(kind):
    """A driver that starts its scans at another azimuth (every ring a rotation of its angle order), a clockwise sensor
    (reversed), or both: the first batch falls back for its angle order, the library then finds the rings' transforms
    ahead of the kernel (ring_cut_kernel) and the stream is read in place again; when the stream arrives in order
    again the transforms are dropped.  Rings of one scan start at different columns (per-laser azimuth offsets)."""
    R, Ccols, nb = 16, 900, 4
    kw = INPUT_ORDER_KW[kind]
    turned = [make_scan(R, Ccols, seed=1600 + i, **kw) for i in range(nb)]
    # per-ring offsets: rotate the records of some rings by another column or two inside the column-major layout
    for c in turned:
        grid = c.reshape(Ccols, R)
        for r, shift in ((3, 1), (7, 2), (12, 5)):
            grid[:, r] = np.roll(grid[:, r], shift)
    plain = [make_scan(R, Ccols, seed=1700 + i) for i in range(nb)]
    want_t = [OB.extract(c, canonical_ties=False) for c in turned]
    want_p = [OB.extract(c, canonical_ties=False) for c in plain]
    f = FeatureExtraction(device=0, max_points_per_scan=R * Ccols, max_batch=nb, max_points_per_ring=Ccols, max_rings=R)
    seen = []
    for rep in range(4):
        got = f.extract_batch(turned)
        seen.append(f.scan_routes(nb).tolist())
        for i in range(nb):
            assert_scan_equal(got[i], want_t[i], "%s scan %d, rep %d" % (kind, i, rep))
    assert seen[0] == [0] * nb and seen[-1] == [2] * nb, seen
    for rep in range(4):
        got = f.extract_batch(plain)
        seen.append(f.scan_routes(nb).tolist())
        for i in range(nb):
            assert_scan_equal(got[i], want_p[i], "in order again, scan %d, rep %d" % (i, rep))
    assert seen[-1] == [1] * nb, seen
    f.close()


def test_route_follows_the_stream_while_the_host_runs_ahead():
    """A caller that queues batch after batch without ever waiting (a loop of lfx_extract_batch_device calls: the bench, a
    pipeline): the report of a batch reaches the host through pinned memory with the batch's serial number before and after
    it, and the next batch's route is chosen from the newest block that is whole -- not from an event of the batch just
    queued, which such a caller would never find passed.  A stream of rotated scans (three batches, waited for: the ring
    transforms are on) turns into one in angle order: a hundred and twenty batches, no synchronise, the host far ahead of the device -- the
    later ones are read without the transforms (route 1) although the host never waited for any of them."""
    import torch
    R, Ccols, nb = 16, 900, 1024
    st = torch.cuda.current_stream().cuda_stream
    f = FeatureExtraction(device=0, max_points_per_scan=R * Ccols, max_batch=nb, max_points_per_ring=Ccols, max_rings=R)

    def resident(scans):
        tiled = [scans[i % len(scans)] for i in range(nb)]
        return torch.from_numpy(synth.concat(tiled).view(np.uint8).copy()).cuda(), np.array([len(c) for c in tiled], np.uint32)

    turned = [make_scan(R, Ccols, seed=2600 + i, start_col=211) for i in range(8)]
    plain = [make_scan(R, Ccols, seed=2700 + i) for i in range(8)]
    d_t, n_t = resident(turned)
    d_p, n_p = resident(plain)
    for _ in range(3):
        f.extract_batch_device(d_t.data_ptr(), n_t, st)
        routes = f.scan_routes(nb, st).tolist()
    assert routes == [2] * nb, sorted(set(routes))
    for _ in range(120):      # (a few suffice when the device keeps up; the margin is for a slow box)
        f.extract_batch_device(d_p.data_ptr(), n_p, st)
    routes = f.scan_routes(nb, st).tolist()
    assert routes == [1] * nb, "the route did not follow the stream while the host ran ahead: %s" % sorted(set(routes))
    assert_scan_equal(f.download(3, st), OB.extract(plain[3], canonical_ties=False), "scan 3 of the last batch")
    f.close()


def test_batches_of_changing_size_and_stream_on_one_context():
    """No reset launch stands in front of an organised batch (round 6): the batch's accumulators exist twice and every batch's
    compaction zeroes the other set over the scans the batch before last dirtied there; the bucketing route's tables are
    reset when a batch takes that route or follows one that did.  Batch sizes and streams change from call to call here --
    large after small, organised after shuffled after organised, a scan that falls back in an otherwise clean batch, an
    empty scan, zero records with the filter on -- and every scan of every batch must equal the oracle."""
    import torch
    from lidar_feature_extraction_amd import concat
    R, C = 16, 600
    st = torch.cuda.current_stream().cuda_stream
    f = FeatureExtraction(device=0, max_points_per_scan=R * C, max_batch=12, max_points_per_ring=C, max_rings=R, drop_zero_points=True)
    plain = [make_scan(R, C, seed=3100 + i) for i in range(12)]
    shuffled = [make_scan(R, C, seed=3200 + i, shuffle=True) for i in range(12)]
    ragged = [make_scan(R, C, seed=3300 + i, drop_fraction=0.1) for i in range(12)]
    rotated = [make_scan(R, C, seed=3400 + i, start_col=77) for i in range(12)]
    zeroed = [_zeroed(make_scan(R, C, seed=3500 + i), 0.06, 70 + i) for i in range(12)]
    empty = plain[0][:0]
    want = {}

    def oracle_of(c, zero=None):
        key = (c.ctypes.data, len(c))
        if key not in want:
            keep = np.arange(len(c)) if zero is None else np.nonzero(~zero)[0]
            want[key] = (OB.extract(np.ascontiguousarray(c[keep]), canonical_ties=False), keep, np.zeros(len(c), bool) if zero is None else zero)
        return want[key]

    def run(scans, zeros=None, what=""):
        d = torch.from_numpy(concat(scans).view(np.uint8).copy()).cuda() if sum(len(c) for c in scans) else torch.zeros(32, dtype=torch.uint8).cuda()
        f.extract_batch_device(d.data_ptr(), np.array([len(c) for c in scans], np.uint32), st)
        for k, c in enumerate(scans):
            g = f.download(k, st)
            if len(c) == 0:
                assert len(g.labels) == 0 and len(g.edge_index) == 0 and len(g.surface_index) == 0 and len(g.ring_id) == 0, what + ": empty scan %d" % k
                continue
            w, keep, z = oracle_of(c, None if zeros is None else zeros[k])
            from tests.parity import assert_filtered_equal
            assert_filtered_equal(g, w, keep, z, "%s scan %d" % (what, k))
        torch.cuda.synchronize()

    run(plain[:12], what="12 organised")
    run(plain[:12], what="12 organised again (one-launch tail from here on)")
    run(plain[:3], what="3 organised after 12")
    run(plain[:12], what="12 organised after 3 (the other set was dirtied over 12 scans two batches ago)")
    run([plain[0], rotated[1], plain[2], empty, plain[4]], what="a rotated scan and an empty one among organised ones (the tail redoes the rotated one)")
    run(plain[:12], what="12 organised after a fall-back")
    run(shuffled[:5], what="5 shuffled (the list routes: the reset kernel runs)")
    run(shuffled[:5], what="5 shuffled again")
    for k in range(4):
        run(shuffled[:7], what="7 shuffled, batch %d (the stream moves to the bucketing route)" % k)
    run(plain[:12], what="12 organised after bucketing batches")
    for k in range(20):
        run(plain[:2 + (5 * k) % 11], what="organised, changing sizes, batch %d" % k)
    run([c for c, _ in zeroed[:6]], [z for _, z in zeroed[:6]], what="6 with zero records (plain form refuses them)")
    for k in range(4):
        run([c for c, _ in zeroed[:1 + 3 * k]], [z for _, z in zeroed[:1 + 3 * k]], what="zero records, batch %d (the holes form after the report)" % k)
    run(ragged[:9], what="9 ragged after the holes form")
    run(plain[:12], what="12 organised at the end")
    run([empty, empty], what="two empty scans")
    run(plain[:4], what="4 organised after an empty batch")
    f.close()


@pytest.mark.parametrize("tail", [None, "short"])
def test_organised_scan_kernel_more_fall_backs_than_expected(tail):
    """The bucketing route is launched for as many scans as earlier batches sent to it (plus a few); scans beyond
    that are still bucketed, and their rings are handed to the workgroup-per-ring kernel.  32 rotated scans arrive
    at a context that has seen nothing (room for 8), then again (room for all), then 32 organised ones.
    "short": the same with the two-launch tail pinned (the form the route has while nothing has been falling back, i.e.
    exactly when a guess is too small): every list-driven kernel has to cope with a list four times its grid."""
    if tail == "short":
        os.environ["LFX_DEBUG_FUSED"], os.environ["LFX_DEBUG_SHORT_TAIL"] = "1", "1"
    try:
        _more_fall_backs_than_expected()
    finally:
        os.environ.pop("LFX_DEBUG_FUSED", None)
        os.environ.pop("LFX_DEBUG_SHORT_TAIL", None)


def _more_fall_backs_than_expected():
    R, Ccols, nb = 8, 600, 32
    rot = [make_scan(R, Ccols, seed=700 + i, start_col=37 + 11 * i) for i in range(nb)]
    srt = [make_scan(R, Ccols, seed=800 + i) for i in range(nb)]
    want_rot = [OB.extract(c, canonical_ties=False) for c in rot]
    want_srt = [OB.extract(c, canonical_ties=False) for c in srt]
    f = FeatureExtraction(device=0, max_points_per_scan=R * Ccols, max_batch=nb, max_points_per_ring=Ccols, max_rings=R)
    for rep in range(3):
        got = f.extract_batch(rot)
        for i in range(nb):
            assert_scan_equal(got[i], want_rot[i], "rotated %d, rep %d" % (i, rep))
    for rep in range(2):
        got = f.extract_batch(srt)
        for i in range(nb):
            assert_scan_equal(got[i], want_srt[i], "organised %d, rep %d" % (i, rep))
    f.close()


def test_organised_scan_kernel_zero_point_filter_and_colored_scan():
    """With drop_zero_points a (0,0,0) record is not part of the scan, so a scan that holds one is not organised;
    colored_scan of an organised scan is built from the input records (nothing was staged)."""
    import torch
    R, Ccols = 16, 500
    c = make_scan(R, Ccols, seed=910)
    z = c.copy()
    for k in (R * 40 + 3, R * 41 + 3, R * 300 + 9):
        z["x"][k] = 0.0
        z["y"][k] = 0.0
        z["z"][k] = 0.0
    keep = np.nonzero(~((z["x"] == 0) & (z["y"] == 0) & (z["z"] == 0)))[0]
    want_c = OB.extract(c, canonical_ties=False)
    want_z = OB.extract(np.ascontiguousarray(z[keep]), canonical_ties=False)
    f = FeatureExtraction(device=0, max_points_per_scan=len(c), max_batch=2, max_points_per_ring=Ccols, max_rings=R, drop_zero_points=True)
    got = f.extract_batch([c, z])
    assert_scan_equal(got[0], want_c, "organised, filter on")
    assert np.array_equal(got[1].labels[keep], want_z["labels"]) and not got[1].labels[np.setdiff1d(np.arange(len(z)), keep)].any()
    assert np.array_equal(got[1].edge_index, keep[want_z["edge_index"]].astype(np.uint32))
    assert np.array_equal(got[1].surface_index, keep[want_z["surface_index"]].astype(np.uint32))
    # colored_scan of the pair, on the device
    host = synth.concat([c, z]).view(np.uint8)
    dev = torch.from_numpy(host.copy()).to("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    f.extract_batch_device(dev.data_ptr(), [len(c), len(z)], stream)
    cap = 2 * len(c)
    col = torch.zeros((cap, 8), dtype=torch.float32, device="cuda:0")
    coffs = torch.zeros(3, dtype=torch.int32, device="cuda:0")
    f.pack_colored(col.data_ptr(), coffs.data_ptr(), cap, stream)
    torch.cuda.synchronize()
    col, coffs = col.cpu().numpy(), coffs.cpu().numpy()
    order = want_c["sorted_index"]
    assert coffs[1] - coffs[0] == len(c)
    g = col[coffs[0]:coffs[1]]
    assert np.array_equal(g[:, 0], c["x"][order]) and np.array_equal(g[:, 1], c["y"][order]) and np.array_equal(g[:, 2], c["z"][order])
    rgba = g[:, 4].copy().view(np.uint32)
    lab = want_c["labels"][order]
    for v in np.unique(lab):
        rgb = np.zeros(3, np.uint8)
        assert LB.load().lfx_label_to_color(int(v), rgb.ctypes.data_as(LB.C.POINTER(LB.C.c_uint8))) == 0
        assert np.all((rgba[lab == v] >> 16) & 255 == rgb[0]) and np.all(rgba[lab == v] & 255 == rgb[2])
    f.close()


def test_context_without_the_per_point_curvature():
    """lfx_config.outputs without LFX_OUT_CURVATURE: the kernels do not write the per-point curvature at all (8 of the 9
    bytes they store per point); everything else -- labels, index sets, the two clouds with the curvature of their points
    as intensity, the ring projection -- is what a full context gives, on the organised route, the bucketing route (a
    shuffled and a ragged scan in the batch) and the workgroup-per-ring kernel (a ring of 3 000 points in 2 blocks)."""
    from lidar_feature_extraction_amd import binding as LB
    clouds = [make_scan(16, 900, seed=7300), make_scan(16, 900, seed=7301, shuffle=True), make_scan(16, 900, seed=7302, drop_fraction=0.05)]
    f = FeatureExtraction(device=0, max_points_per_scan=16 * 900, max_batch=3, max_points_per_ring=900, max_rings=16,
                          outputs=LB.OUT_FEATURES | LB.OUT_LABELS | LB.OUT_SORTED_INDEX)
    for rep in range(2):
        got = f.extract_batch(clouds)
        for i, c in enumerate(clouds):
            w = OB.extract(c, canonical_ties=False)
            assert len(got[i].curvature) == 0
            assert np.array_equal(got[i].labels, w["labels"]), "scan %d labels" % i
            assert np.array_equal(got[i].sorted_index, w["sorted_index"].astype(np.uint32))
            assert np.array_equal(got[i].edge_index, w["edge_index"].astype(np.uint32))
            assert np.array_equal(got[i].surface_index, w["surface_index"].astype(np.uint32))
            assert got[i].edge_points.tobytes() == w["edge_points"].tobytes() and got[i].surface_points.tobytes() == w["surface_points"].tobytes()
    assert not f.device_view().curvature_sorted
    f.close()
    hp = HyperParameters(n_blocks=2)
    c = make_scan(2, 3000, seed=7303)
    f = FeatureExtraction(hp, device=0, max_points_per_scan=len(c), max_batch=1, outputs=LB.OUT_FEATURES | LB.OUT_LABELS)
    g, w = f.ExtractFeatures(c), OB.extract(c, oracle_params(hp), canonical_ties=False)
    assert np.array_equal(g.labels, w["labels"]) and g.edge_points.tobytes() == w["edge_points"].tobytes()
    assert g.surface_points.tobytes() == w["surface_points"].tobytes() and len(g.curvature) == 0
    f.close()


def test_contexts_of_different_ring_capacity_side_by_side():
    """Two lidars of different width in one process: the dynamic-LDS limits of the workgroup-per-ring kernels are per
    function, not per context, so a later, smaller context must not lower them for an earlier, larger one."""
    big = make_scan(2, 3900, seed=1501, shuffle=True)            # rings of 3900 points: sorted and labelled by the workgroup-per-ring kernels
    small = make_scan(16, 300, seed=1502, shuffle=True)
    f_big = FeatureExtraction(device=0, max_points_per_scan=len(big), max_batch=1)                          # ring capacity 4096
    f_small = FeatureExtraction(device=0, max_points_per_scan=len(small), max_batch=1, max_points_per_ring=320, max_rings=16)
    want_big, want_small = OB.extract(big, canonical_ties=False), OB.extract(small, canonical_ties=False)
    for rep in range(2):
        assert_scan_equal(f_small.ExtractFeatures(small), want_small, "small context, rep %d" % rep)
        assert_scan_equal(f_big.ExtractFeatures(big), want_big, "large context after the small one, rep %d" % rep)
    f_small.close()
    assert_scan_equal(f_big.ExtractFeatures(big), want_big, "large context after the small one was destroyed")
    f_big.close()


def _cloud(ring, x, y, z=None):
    c = np.zeros(len(ring), POINT_DTYPE)
    c["ring"], c["x"], c["y"] = ring, x, y
    c["z"] = 0.5 if z is None else z
    c["pad"] = 1.0
    return c


def test_ring_skip_conditions(fx):
    """Rings the reference removes or abandons contribute nothing (feature_extraction.cpp:116,154-156)."""
    rng = np.random.default_rng(5)
    parts = []

    def ring_pts(rid, n, r0=8.0):
        az = np.sort(rng.uniform(-3.0, 3.0, n))
        r = r0 + 0.01 * rng.standard_normal(n)
        return rid, r * np.cos(az), r * np.sin(az)

    sizes = {0: 400, 1: 3, 2: 9, 3: 14, 4: 16, 5: 21, 7: 300, 9: 5}   # P=5,B=6: <6 sparse, <11 conv, <16 blocks, 16..21 block<2
    ring, x, y = [], [], []
    for rid, n in sizes.items():
        r_, x_, y_ = ring_pts(rid, n)
        ring += [r_] * n
        x += list(x_)
        y += list(y_)
    # ring 7: two adjacent points exactly (0,0) in xy -> CalcRadian throws (math.cpp:40-42)
    c = _cloud(np.array(ring, np.uint16), np.array(x, np.float32), np.array(y, np.float32))
    sel7 = np.nonzero(c["ring"] == 7)[0]
    c["x"][sel7[100]] = 0.0
    c["y"][sel7[100]] = 0.0
    c["x"][sel7[101]] = 0.0
    c["y"][sel7[101]] = 0.0
    c["z"][sel7[101]] = 0.7
    perm = rng.permutation(len(c))
    c = c[perm]
    got = fx.ExtractFeatures(c)
    want = OB.extract(c, canonical_ties=True)
    assert want["ring_status"].tolist().count(0) == 1
    # two identical (0,0) points tie under the angle predicate: canonical (index) order on both sides
    assert_scan_equal(got, want, "skip[ties]")
    status_codes_equal_where_single_cause(got, want)
    st = dict(zip(got.ring_id.tolist(), got.ring_status.tolist()))
    assert st[1] == 1 and st[9] == 1 and st[2] == 2 and st[3] == 3 and st[7] == 5 and st[0] == 0
    assert st[4] != 0 and st[5] != 0
    # the log callback: one warning per abandoned ring with the text of the reference's exception -- what the node hands to
    # RCLCPP_WARN (feature_extraction.cpp:154-156); the two sparse rings are dropped without a sound (ring.cpp:46-59)
    import ctypes as C
    heard = []
    FN = C.CFUNCTYPE(None, C.c_int, C.c_char_p, C.c_void_p)
    cb = FN(lambda level, text, user: heard.append((level, text.decode())))
    L = LB.load()
    L.lfx_set_log_callback.argtypes = [C.c_void_p, FN, C.c_void_p]
    assert L.lfx_set_log_callback(fx._ctx, cb, None) == 0
    fx.ExtractFeatures(c)
    assert L.lfx_set_log_callback(fx._ctx, FN(0), None) == 0
    fx.ExtractFeatures(c)                                         # switched off again: nothing more is heard
    assert len(heard) == 5 and all(level == 1 for level, _ in heard), heard
    texts = sorted(t for _, t in heard)
    assert "All input values are zero. Angle cannot be calculated" in texts                        # ring 7, math.cpp:41
    assert "Input array size 9 cannot be smaller than weight size 11" in texts                     # ring 2, convolution.cpp:40-41
    assert "end_index - start_index (which is 4) cannot be smaller than n_blocks (which is 6" in texts     # ring 3, index_range.cpp:36-38
    assert sum(t.startswith("The input point size (which is ") for t in texts) == 2                # rings 4 and 5, neighbor.hpp:72-73


def test_one_zero_norm_point_breaks_links_only(fx):
    """A single (0,0) point: cos = 0/0 = NaN, acos(NaN) < thr is false -> link broken, ring kept."""
    c = make_scan(4, 600, seed=9)
    k = np.nonzero(c["ring"] == 2)[0][300]
    c["x"][k] = 0.0
    c["y"][k] = 0.0
    got = fx.ExtractFeatures(c)
    want = OB.extract(c, canonical_ties=False)
    assert want["ring_status"].tolist() == [0, 0, 0, 0]
    assert_scan_equal(got, want, "one-zero")


def test_batch_of_ragged_scans_and_empty_scan(fx):
    clouds = [make_scan(16, 500, seed=3), make_scan(8, 1200, seed=4, drop_fraction=0.2),
              np.zeros(0, POINT_DTYPE), make_scan(24, 300, seed=5, shuffle=True), make_scan(2, 64, seed=6)]
    got = fx.extract_batch(clouds)
    for i, c in enumerate(clouds):
        if len(c) == 0:
            assert len(got[i].labels) == 0 and len(got[i].edge_index) == 0 and len(got[i].ring_id) == 0
            continue
        assert_scan_equal(got[i], OB.extract(c, canonical_ties=False), "batch%d" % i)


def test_duplicate_points_canonical_ties(fx):
    """Exact duplicates tie under the angle predicate; the HIP path orders ties by arrival index."""
    base = make_scan(8, 400, seed=21)
    c = np.zeros(len(base) + 40, POINT_DTYPE)
    c[:len(base)] = base
    c[len(base):] = base[100:140]
    got = fx.ExtractFeatures(c)
    want = OB.extract(c, canonical_ties=True)
    assert want["angle_ties"] > 0
    assert_scan_equal(got, want, "dups[ties]")


def test_curvature_ties_canonical(fx):
    """Noise-free symmetric data gives exactly equal curvatures inside a block; lower index first."""
    n = 600
    az = -math.pi + 2 * math.pi * (np.arange(n) + 0.5) / n
    r = 5.0 + 1.0 * (np.arange(n) % 7 == 0)        # exact repeating pattern -> exact ties
    c = _cloud(np.zeros(n, np.uint16), (r * np.cos(az)).astype(np.float32), (r * np.sin(az)).astype(np.float32))
    got = fx.ExtractFeatures(c)
    want = OB.extract(c, canonical_ties=True)
    assert_scan_equal(got, want, "curvties[ties]")


def test_a_stray_ring_id_is_a_ring_of_its_own():
    """One point with an id no other point has: the reference gives it a ring of its own (ring.hpp:114-125), which
    RemoveSparseRings then drops (ring.cpp:46-59).  (Until round 6: LFX_ERR_RING_ID for the whole call.)  On the device path,
    where the library has no copy of the records to look ids up in, the error stands until the caller names the ids."""
    import torch
    c = make_scan(4, 200, seed=1)
    c["ring"][17] = 300
    f = FeatureExtraction(device=0, max_points_per_scan=len(c), max_batch=1, max_rings=8)
    want = OB.extract(c, canonical_ties=False)
    assert 300 in want["ring_id"].tolist()
    assert_scan_equal(f.ExtractFeatures(c), want, "stray id")
    f.close()
    # ... and ids looked up in one odd scan do not keep a stream of ids 0 .. rings-1 off the organised route: when a later scan
    # carries an id the table does not hold and both sets together exceed the context's rings, the new scan's ids stand alone
    odd = make_scan(8, 300, seed=5)
    odd["ring"][odd["ring"] == 7] = 300
    plain8 = make_scan(8, 300, seed=6)
    f = FeatureExtraction(device=0, max_points_per_scan=len(odd), max_batch=1, max_points_per_ring=300, max_rings=8)
    assert_scan_equal(f.ExtractFeatures(odd), OB.extract(odd, canonical_ties=False), "ring 7 numbered 300")
    routes = []
    for k in range(4):
        g = f.ExtractFeatures(plain8)
        routes.append(int(f.scan_routes(1, 0)[0]))
    assert routes[-1] == 1, routes
    assert_scan_equal(g, OB.extract(plain8, canonical_ties=False), "ids 0 .. 7 again")
    f.close()
    f = FeatureExtraction(device=0, max_points_per_scan=len(c), max_batch=1, max_rings=8)
    d = torch.from_numpy(c.view(np.uint8)).to("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    f.extract_batch_device(d.data_ptr(), np.array([len(c)], np.uint32), st)
    with pytest.raises(LB.LfxError) as e:
        f.batch_status(st)
    assert e.value.code == -5
    f.set_ring_ids([0, 1, 2, 3, 300])
    f.extract_batch_device(d.data_ptr(), np.array([len(c)], np.uint32), st)
    f.batch_status(st)
    assert_scan_equal(f.download(0, st), want, "stray id, device path, ids named")
    f.close()


def test_capacity_errors():
    f = FeatureExtraction(device=0, max_points_per_scan=1000, max_batch=1)
    with pytest.raises(LB.LfxError) as e:
        f.ExtractFeatures(make_scan(4, 300, seed=1))
    assert e.value.code == -4
    with pytest.raises(LB.LfxError):
        f.extract_batch([make_scan(2, 100), make_scan(2, 100)])
    f.close()


def test_ring_longer_than_capacity_is_reported():
    c = make_scan(2, 3000, seed=8)
    f = FeatureExtraction(device=0, max_points_per_scan=len(c), max_batch=1, max_points_per_ring=2048)
    got = f.ExtractFeatures(c)
    assert got.ring_status.tolist() == [7, 7] and not got.labels.any() and len(got.edge_index) == 0
    f.close()


def test_device_resident_path_with_torch(fx):
    import torch
    clouds = [make_scan(16, 900, seed=40 + i) for i in range(3)]
    host = synth.concat(clouds).view(np.uint8)
    dev = torch.from_numpy(host.copy()).to("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    fx.extract_batch_device(dev.data_ptr(), [len(c) for c in clouds], stream)
    torch.cuda.synchronize()
    for i, c in enumerate(clouds):
        assert_scan_equal(fx.download(i, stream), OB.extract(c, canonical_ties=False), "device%d" % i)


def test_large_batch_compaction_with_the_totals_kernel():
    """A batch of more than 8 192 rings (600 scans x 16 rings) compacts with ring_totals_kernel ahead of
    feature_compact_kernel -- bench.py's form; the other tests' batches are small enough for the compaction kernel to find
    the rings' places itself.  Every scan's clouds must be what the same scan gives alone."""
    import torch
    unique = [make_scan(16, 900, seed=4100 + i) for i in range(6)]
    clouds = [unique[i % 6] for i in range(600)]
    f = FeatureExtraction(device=0, max_points_per_scan=16 * 900, max_batch=600, max_points_per_ring=900, max_rings=16)
    dev = torch.from_numpy(synth.concat(clouds).view(np.uint8).copy()).to("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(2):                                   # (the second call runs on what the first one reported)
        f.extract_batch_device(dev.data_ptr(), [len(c) for c in clouds], stream)
    torch.cuda.synchronize()
    want = [OB.extract(c, canonical_ties=False) for c in unique]
    for i in (0, 1, 5, 299, 598, 599):
        assert_scan_equal(f.download(i, stream), want[i % 6], "scan %d of 600" % i)
    for i in range(0, 600, 7):
        g = f.download(i, stream)
        assert np.array_equal(g.edge_index, want[i % 6]["edge_index"].astype(np.uint32)), i
        assert np.array_equal(g.surface_points[:, :3], want[i % 6]["surface_points"][:, :3]), i
    f.close()


def test_large_random_parameters(fx):
    """Other paddings / block counts / thresholds, incl. the largest supported padding."""
    rng = np.random.default_rng(11)
    c = make_scan(6, 1500, seed=31, drop_fraction=0.05)
    for pad, nb in [(1, 1), (2, 3), (3, 10), (8, 6), (15, 4), (7, 97)]:
        hp = HyperParameters(padding=pad, n_blocks=nb, neighbor_degree_threshold=float(rng.uniform(0.3, 4.0)),
                             distance_diff_threshold=float(rng.uniform(0.05, 1.0)),
                             parallel_beam_min_range_ratio=float(rng.uniform(0.001, 0.1)),
                             edge_threshold=float(rng.uniform(0.001, 1.0)), surface_threshold=float(rng.uniform(0.001, 1.0)),
                             min_range=0.5, max_range=9.0)
        f = FeatureExtraction(hp, device=0, max_points_per_scan=len(c), max_batch=1)
        assert_scan_equal(f.ExtractFeatures(c), OB.extract(c, oracle_params(hp), canonical_ties=False), "P%d/B%d" % (pad, nb))
        f.close()


def test_long_blocks_take_the_lds_labelling_path(fx):
    """Blocks longer than 512 points are labelled by the LDS/barrier variant of the same iteration."""
    c = make_scan(3, 3600, seed=17, drop_fraction=0.02)
    for nb in (1, 2, 5, 6, 7):
        hp = HyperParameters(n_blocks=nb)
        f = FeatureExtraction(hp, device=0, max_points_per_scan=len(c), max_batch=1)
        assert_scan_equal(f.ExtractFeatures(c), OB.extract(c, oracle_params(hp), canonical_ties=False), "B%d" % nb)
        f.close()
    # and an unsorted ring that needs the fallback angle sort at the same time
    c2 = make_scan(2, 3000, seed=18, shuffle=True)
    hp = HyperParameters(n_blocks=3)
    f = FeatureExtraction(hp, device=0, max_points_per_scan=len(c2), max_batch=1)
    assert_scan_equal(f.ExtractFeatures(c2), OB.extract(c2, oracle_params(hp), canonical_ties=False), "B3-shuffled")
    f.close()


@pytest.mark.parametrize("organised", [True, False])
def test_long_rings_take_the_long_form_of_the_unit_kernel(organised):
    """Rings of more than ~2 230 points (6 blocks) have units of more than 384 positions: the 12-chunk form of the unit
    kernels takes them (blocks of up to 768 positions, rings of up to ~4 510 points) -- a 0.1-degree sensor's 3 600 columns
    -- on the organised route and on the bucketing route (records shuffled, some dropped), with the reference's default
    thresholds and with others, rings turned; a 7-block setting and a ring capacity of the maximum.  Rings too long even
    for that (4 096 points in 4 blocks, 4 608 in 6) are still the workgroup-per-ring kernel's: the same results."""
    rng = np.random.default_rng(5)
    for rings, cols, hp, kw in [(8, 3600, HyperParameters(), {}), (4, 4090, HyperParameters(), {}),
                                (8, 3000, HyperParameters(edge_threshold=0.1, surface_threshold=0.02), {}),
                                (4, 3600, HyperParameters(n_blocks=7), {}), (4, 3600, HyperParameters(), {"start_col": 700}),
                                (4, 4096, HyperParameters(n_blocks=4), {}),
                                # an HDL-64E at 5 Hz: ~4 500 points per ring (the reference has no cap, ring.hpp:114-125), and the cap itself
                                (4, 4500, HyperParameters(), {}), (2, 4608, HyperParameters(), {}), (2, 4500, HyperParameters.launch_yaml(), {})]:
        clouds = [make_scan(rings, cols, seed=int(rng.integers(1 << 30)), drop_fraction=0.0 if organised else 0.03,
                            shuffle=not organised, **kw) for _ in range(3)]
        f = FeatureExtraction(hp, device=0, max_points_per_scan=rings * cols, max_batch=3, max_points_per_ring=cols, max_rings=rings)
        for _ in range(2 if kw else 1):                      # (turned rings: the ring transforms come with the second batch)
            got = f.extract_batch(clouds)
        for i, c in enumerate(clouds):
            assert_scan_equal(got[i], OB.extract(c, oracle_params(hp), canonical_ties=False), "%dx%d/%d" % (rings, cols, i))
        if organised and hp.n_blocks >= 6 and cols <= 4500:     # (4 608 points in 6 blocks: units of more than 768 positions)
            assert list(f.scan_routes(3)) == [2 if kw else 1] * 3, "read in place by the organised-scan kernel"
        f.close()


@pytest.mark.parametrize("padding", [16, 23, 40, 63])
def test_paddings_beyond_the_windows(padding):
    """The reference accepts any convolution_padding > 0 (hyper_parameter.hpp:45-53).  Up to 15 the kernels' 32-position windows
    span a pick's reach; beyond that (16 .. LFX_MAX_PADDING = 63) every ring takes the workgroup-per-ring kernel, whose labelling
    and occlusion fills then walk the positions in reach.  Same results as the oracle: sorted and shuffled input, a ring too short
    for the padding (skipped as the reference skips it), the launch file's thresholds."""
    for hp in (HyperParameters(padding=padding), HyperParameters(padding=padding, neighbor_degree_threshold=3.0, edge_threshold=50.0, max_range=1000.0)):
        clouds = [make_scan(8, 1200, seed=4100 + padding), make_scan(8, 700, seed=4200 + padding, shuffle=True),
                  make_scan(4, 2 * padding + 3, seed=4300 + padding, spikes=False)]
        f = FeatureExtraction(hp, device=0, max_points_per_scan=8 * 1200, max_batch=3, max_points_per_ring=1200, max_rings=8)
        got = f.extract_batch(clouds)
        for i, c in enumerate(clouds):
            want = OB.extract(c, oracle_params(hp), canonical_ties=False)
            assert_scan_equal(got[i], want, "padding %d, scan %d" % (padding, i))
        assert any(len(g.edge_index) + len(g.surface_index) > 0 for g in got)
        f.close()


@pytest.mark.parametrize("chunks", [3, 4, 5, 6, 12])
def test_every_unit_kernel_variant(chunks):
    """The wave-per-unit kernel is instantiated for spans of 3..6 chunks of 64 positions and the host picks
    one from max_points_per_ring; here each variant is forced in turn.  900-column rings fit all of them;
    1800-column rings do not fit the 3- and 4-chunk variants, whose units then hand the ring to the
    workgroup-per-ring kernel -- same results either way."""
    import os
    clouds = [make_scan(16, 900, seed=80), make_scan(16, 1800, seed=81), make_scan(16, 1500, seed=82, drop_fraction=0.07)]
    os.environ["LFX_DEBUG_UNIT_CHUNKS"] = str(chunks)
    try:
        f = FeatureExtraction(device=0, max_points_per_scan=16 * 1800, max_batch=3, max_rings=16)
    finally:
        del os.environ["LFX_DEBUG_UNIT_CHUNKS"]
    got = f.extract_batch(clouds)
    for i, c in enumerate(clouds):
        assert_scan_equal(got[i], OB.extract(c, canonical_ties=False), "chunks%d/%d" % (chunks, i))
    f.close()


def test_organised_kernel_gives_a_scan_up_half_way():
    """A scan that breaks the organised pattern in a LATE unit: by then the earlier units of the scan have written their
    labels, curvatures, per-unit feature records and their counts into their rings' totals.  The unit that finds the break
    puts the scan on the fall-back list (scan_falls_back); the bucketing route redoes it whole in the same call and
    overwrites all of that -- the unit tables, the ring counts, and the compaction takes the bucketing route's tables for
    such a scan (feature_compact_kernel, by_ring).  Scans in order, one such scan, rotated rings."""
    f = FeatureExtraction(device=0, max_points_per_scan=64 * 1800, max_batch=4, max_points_per_ring=1800, max_rings=64)
    clouds = [make_scan(64, 1800, seed=880 + i) for i in range(4)]
    clouds[2] = clouds[2].copy()
    a, b = 64 * 1000 + 5, 64 * 1001 + 5                 # two neighbouring records of ring 5 change places, in a late unit:
    clouds[2][[a, b]] = clouds[2][[b, a]]               # the angle order breaks there and the scan is given up half-way
    for got, c in zip(f.extract_batch(clouds), clouds):
        assert_scan_equal(got, OB.extract(c, canonical_ties=False))
    rot = [make_scan(64, 1800, seed=890 + i, start_col=300) for i in range(4)]
    for _ in range(3):                                  # (the ring transforms are switched on from the second batch on)
        res = f.extract_batch(rot)
    for got, c in zip(res, rot):
        assert_scan_equal(got, OB.extract(c, canonical_ties=False))
    f.close()


@pytest.mark.parametrize("env", ["LFX_DEBUG_NO_FAST_PATH", "LFX_DEBUG_GENERIC_THRESHOLDS", "LFX_DEBUG_TOTALS_KERNEL"])
def test_fallback_paths_give_the_same_results(env):
    """The workgroup-per-ring kernel for every ring is kept as the fallback of the wave-per-unit kernel; with the
    reference's default thresholds the unit kernel runs a variant that has them as literals, here switched off; batches of
    more than 8 192 rings compact with ring_totals_kernel ahead of feature_compact_kernel (small ones without it), here
    forced for a small batch.  Same results."""
    import os
    clouds = [make_scan(16, 1200, seed=70), make_scan(16, 1200, seed=71, drop_fraction=0.1), make_scan(8, 700, seed=72, shuffle=True)]
    os.environ[env] = "1"
    try:
        f = FeatureExtraction(device=0, max_points_per_scan=16 * 1200, max_batch=3, max_rings=16)
    finally:
        del os.environ[env]
    got = f.extract_batch(clouds)
    for i, c in enumerate(clouds):
        assert_scan_equal(got[i], OB.extract(c, canonical_ties=False), "%s/%d" % (env, i))
    f.close()


def test_zero_point_filter_matches_the_upstream_converter():
    """drop_zero_points: all-zero points are dropped before extraction, as point_type_converter does
    (convert.py:162-163,192; pinned there by test_convert.py).  Result = oracle on the filtered cloud."""
    rng = np.random.default_rng(8)
    c = make_scan(16, 1000, seed=55)
    zero = rng.uniform(0, 1, len(c)) < 0.03
    c["x"][zero] = 0.0
    c["y"][zero] = 0.0
    c["z"][zero] = 0.0
    keep = np.nonzero(~zero)[0]
    filtered = np.ascontiguousarray(c[keep])
    want = OB.extract(filtered, canonical_ties=False)
    f = FeatureExtraction(device=0, max_points_per_scan=len(c), max_batch=1, max_rings=16, drop_zero_points=True)
    got = f.ExtractFeatures(c)
    f.close()
    assert len(got.sorted_index) == len(keep)
    assert np.array_equal(got.sorted_index, keep[want["sorted_index"]].astype(np.uint32))
    assert np.array_equal(got.labels[keep], want["labels"]) and not got.labels[zero].any()
    assert got.curvature[keep].tobytes() == want["curvature"].tobytes()
    assert np.array_equal(got.edge_index, keep[want["edge_index"]].astype(np.uint32))
    assert np.array_equal(got.surface_index, keep[want["surface_index"]].astype(np.uint32))
    assert got.edge_points.tobytes() == want["edge_points"].tobytes()
    # without the option the zero points stay in (and make their rings' adjacent zero pairs skip conditions)
    f2 = FeatureExtraction(device=0, max_points_per_scan=len(c), max_batch=1, max_rings=16)
    assert len(f2.ExtractFeatures(c).sorted_index) == len(c)
    f2.close()


def test_ring_ids_are_whatever_uint16_the_points_carry():
    """The reference buckets by the id a point carries (an unordered_map<int, ...> over a uint16 field, ring.hpp:114-125,
    point_type.hpp:62-86): ids need not be 0 .. rings-1.  The host entry point looks the ids of such a scan up itself; the
    device path is told (lfx_config.ring_ids / lfx_set_ring_ids).  LFX_ERR_RING_ID is left for more distinct ids than a
    context takes."""
    import torch
    base = make_scan(3, 900, seed=77)
    c = base.copy()
    c["ring"] = np.array([0, 300, 65535], np.uint16)[base["ring"]]
    want = OB.extract(c, canonical_ties=False)
    assert want["ring_id"].tolist()[:3] == [0, 300, 65535]
    f = FeatureExtraction(device=0, max_points_per_scan=len(c), max_batch=2, max_points_per_ring=900, max_rings=3)
    got = f.ExtractFeatures(c)                       # (the first call finds an id it does not know, looks them up, runs again)
    assert_scan_equal(got, want, "looked up")
    assert_scan_equal(f.ExtractFeatures(c), want, "second call")
    assert_scan_equal(f.ExtractFeatures(base), OB.extract(base, canonical_ties=False), "a scan with ids 0 .. 2 next: looked up again")
    f.close()
    # told at creation: the device path, ids in any order
    f = FeatureExtraction(device=0, max_points_per_scan=len(c), max_batch=2, max_points_per_ring=900, max_rings=3, ring_ids=[65535, 0, 300])
    d = torch.from_numpy(c.view(np.uint8)).to("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    f.extract_batch_device(d.data_ptr(), np.array([len(c)], np.uint32), st)
    assert_scan_equal(f.download(0, st), want, "device path, ids given")
    f.set_ring_ids(None)                             # ... and taken back: ids 0 .. 2 again, the organised route with them
    d0 = torch.from_numpy(base.view(np.uint8)).to("cuda:0")
    for _ in range(2):
        f.extract_batch_device(d0.data_ptr(), np.array([len(base)], np.uint32), st)
    assert f.scan_routes(1, st).tolist() == [1]
    assert_scan_equal(f.download(0, st), OB.extract(base, canonical_ties=False), "ids taken back")
    f.close()
    # more distinct ids than the context takes: the error that is left
    many = make_scan(8, 400, seed=78)
    many["ring"] = (many["ring"].astype(np.uint32) * 1000 + 7).astype(np.uint16)
    f = FeatureExtraction(device=0, max_points_per_scan=len(many), max_batch=1, max_points_per_ring=400, max_rings=4)
    with pytest.raises(LB.LfxError) as e:
        f.ExtractFeatures(many)
    assert e.value.code == -5 and "distinct ring ids" in str(e.value)
    f.close()
    f = FeatureExtraction(device=0, max_points_per_scan=len(many), max_batch=1, max_points_per_ring=400, max_rings=8)
    assert_scan_equal(f.ExtractFeatures(many), OB.extract(many, canonical_ties=False), "eight ids of a thousand apart")
    f.close()


def _zeroed(c, fraction, seed, columns=None, rings=None):
    """A copy of grid scan c with a share of its returns written as (0, 0, 0) records (or whole columns / rings of them)."""
    c = c.copy()
    rng = np.random.default_rng(seed)
    zero = rng.uniform(0, 1, len(c)) < fraction
    n_rings = int(c["ring"].max()) + 1
    if columns is not None:
        col = np.arange(len(c)) // n_rings
        zero |= np.isin(col, columns)
    if rings is not None:
        zero |= np.isin(c["ring"], rings)
    for f in ("x", "y", "z"):
        c[f][zero] = 0.0
    return c, zero


def _check_filtered(got, c, zero, hp=None, ctx=""):
    keep = np.nonzero(~zero)[0]
    op = None if hp is None else OB.Params(hp.padding, hp.neighbor_degree_threshold, hp.distance_diff_threshold, hp.parallel_beam_min_range_ratio,
                                           hp.edge_threshold, hp.surface_threshold, hp.min_range, hp.max_range, hp.n_blocks)
    want = OB.extract(np.ascontiguousarray(c[keep]), op, canonical_ties=False)
    if want["angle_ties"] or want["curvature_ties"]:      # (a record moved to another ring may share its new neighbour's direction exactly)
        want = OB.extract(np.ascontiguousarray(c[keep]), op, canonical_ties=True)
    assert np.array_equal(got.sorted_index, keep[want["sorted_index"]].astype(np.uint32)), ctx + ": ring projection"
    assert got.ring_count.tolist() == want["ring_count"].tolist(), ctx + ": ring counts"
    assert np.array_equal(got.ring_status != 0, want["ring_status"] != 0), ctx + ": skipped rings"
    assert np.array_equal(got.labels[keep], want["labels"]) and not got.labels[zero].any(), ctx + ": labels"
    assert got.curvature[keep].tobytes() == want["curvature"].tobytes(), ctx + ": curvature bits"
    assert np.array_equal(got.edge_index, keep[want["edge_index"]].astype(np.uint32)), ctx + ": edge index set"
    assert np.array_equal(got.surface_index, keep[want["surface_index"]].astype(np.uint32)), ctx + ": surface index set"
    assert got.edge_points.tobytes() == want["edge_points"].tobytes() and got.surface_points.tobytes() == want["surface_points"].tobytes(), ctx + ": clouds"


@pytest.mark.parametrize("shape,params", [((64, 1800), "defaults"), ((16, 900), "defaults"), ((16, 1800), "launch_yaml"), ((128, 2048), "defaults"),
                                          ((13, 700), "defaults"), ((32, 3600), "defaults")])
def test_grid_with_holes_is_read_in_place(shape, params):
    """A driver that keeps the grid and writes invalid returns as (0, 0, 0) records, zero filter on (convert.py:162-163,192):
    the HOLES form of the organised route -- grid_count_kernel, then the unit kernel loading by column and compacting in its
    slabs -- gives what the reference gives on the filtered cloud, and the scans are reported as read in place (route 3)."""
    import torch
    from lidar_feature_extraction_amd import concat
    R, C = shape
    hp = HyperParameters.launch_yaml() if params == "launch_yaml" else HyperParameters()
    scans, zeros = [], []
    for k, frac in enumerate((0.05, 0.0, 0.12, 0.02)):
        c, z = _zeroed(make_scan(R, C, seed=300 + k, vfov_deg=22.5 if R >= 128 else 15.0), frac, 17 + k)
        scans.append(c)
        zeros.append(z)
    f = FeatureExtraction(hp, device=0, max_points_per_scan=R * C, max_batch=len(scans), max_points_per_ring=C, max_rings=R,
                          drop_zero_points=True, stream_hint=LB.STREAM_GRID_WITH_HOLES)
    dev = torch.device("cuda", 0)
    d = torch.from_numpy(concat(scans).view(np.uint8)).to(dev)
    n = np.array([len(c) for c in scans], np.uint32)
    st = torch.cuda.current_stream().cuda_stream
    for rep in range(2):                               # (twice: the second batch runs on the other set of accumulators)
        f.extract_batch_device(d.data_ptr(), n, st)
        routes = f.scan_routes(len(scans), st)
        assert routes.tolist() == [3] * len(scans), "every scan read in place as a grid with holes: %s" % routes.tolist()
        for k in range(len(scans)):
            _check_filtered(f.download(k, st), scans[k], zeros[k], hp, "%dx%d %s scan %d rep %d" % (R, C, params, k, rep))
    f.close()


@pytest.mark.parametrize("shape", [(64, 1800), (16, 900), (13, 700), (128, 2048), (32, 3600)])
def test_grid_with_holes_count_pass_by_scan(shape):
    """The count pass of large batches (scan_count_kernel: one workgroup per scan reading its records as they lie) against the
    one of small batches (grid_count_kernel): pinned on for a batch of six here -- plain holes, a ring gone, a ring left too
    short (the bucketing route's), a stretch of empty columns wider than a unit loads (the bucketing route's), no holes at all,
    a record with another ring id than its place's (the bucketing route's) -- every scan equal to the oracle on the filtered cloud."""
    import torch
    from lidar_feature_extraction_amd import concat
    R, C = shape
    scans, zeros = [], []
    base = [make_scan(R, C, seed=600 + k, vfov_deg=22.5 if R >= 128 else 15.0) for k in range(6)]
    c, z = _zeroed(base[0], 0.05, 1); scans.append(c); zeros.append(z)
    c, z = _zeroed(base[1], 0.03, 2, rings=[R // 2]); scans.append(c); zeros.append(z)
    c, z = _zeroed(base[2], 0.02, 3); keepers = np.nonzero((c["ring"] == 1) & ~z)[0][8:]
    for fld in ("x", "y", "z"):
        c[fld][keepers] = 0.0
    z = z.copy(); z[keepers] = True
    scans.append(c); zeros.append(z)
    c, z = _zeroed(base[3], 0.02, 4, columns=np.arange(C // 4, C // 4 + C // 3)); scans.append(c); zeros.append(z)
    c, z = _zeroed(base[4], 0.0, 5); scans.append(c); zeros.append(z)
    c, z = _zeroed(base[5], 0.04, 6); c["ring"][3 * R + 2] = (c["ring"][3 * R + 2] + 1) % R; scans.append(c); zeros.append(z)
    os.environ["LFX_DEBUG_SCAN_COUNT_FROM"] = "1"
    try:
        f = FeatureExtraction(device=0, max_points_per_scan=R * C, max_batch=len(scans), max_points_per_ring=C, max_rings=R,
                              drop_zero_points=True, stream_hint=LB.STREAM_GRID_WITH_HOLES)
    finally:
        os.environ.pop("LFX_DEBUG_SCAN_COUNT_FROM", None)
    d = torch.from_numpy(concat(scans).view(np.uint8)).to("cuda:0")
    n = np.array([len(c) for c in scans], np.uint32)
    st = torch.cuda.current_stream().cuda_stream
    f.extract_batch_device(d.data_ptr(), n, st)
    routes = f.scan_routes(len(scans), st).tolist()
    assert routes == [3, 3, 0, 0, 3, 0], routes
    for k in range(len(scans)):
        _check_filtered(f.download(k, st), scans[k], zeros[k], None, "%dx%d scan %d" % (R, C, k))
    f.close()


def test_grid_with_holes_found_without_a_hint_and_left_again():
    """Without the hint the first batches fall back for their zero records (the plain organised form refuses them), the report
    says why, and the route moves to the holes form; a stream that stops having holes moves back to the plain form."""
    import torch
    from lidar_feature_extraction_amd import concat
    R, C, nb = 16, 900, 8
    holes = [_zeroed(make_scan(R, C, seed=400 + k), 0.05, 40 + k) for k in range(nb)]
    clean = [make_scan(R, C, seed=400 + k) for k in range(nb)]
    f = FeatureExtraction(device=0, max_points_per_scan=R * C, max_batch=nb, max_points_per_ring=C, max_rings=R, drop_zero_points=True)
    dev = torch.device("cuda", 0)
    d_holes = torch.from_numpy(concat([c for c, _ in holes]).view(np.uint8)).to(dev)
    d_clean = torch.from_numpy(concat(clean).view(np.uint8)).to(dev)
    n = np.full(nb, R * C, np.uint32)
    st = torch.cuda.current_stream().cuda_stream
    seen = []
    for k in range(6):
        f.extract_batch_device(d_holes.data_ptr(), n, st)
        seen.append(int(f.scan_routes(nb, st)[0]))
        _check_filtered(f.download(3, st), holes[3][0], holes[3][1], None, "batch %d" % k)
    assert seen[0] == 0 and seen[-1] == 3, "bucketed first, then read in place as a grid with holes: %s" % seen
    seen = []
    for k in range(6):
        f.extract_batch_device(d_clean.data_ptr(), n, st)
        seen.append(int(f.scan_routes(nb, st)[0]))
        assert_scan_equal(f.download(5, st), OB.extract(clean[5], canonical_ties=False), "clean batch %d" % k)
    assert seen[-1] == 1, "back to the plain organised form: %s" % seen
    f.close()


def test_grid_with_holes_odd_scans_out():
    """What the holes form hands to the bucketing route inside the same call: a ring left too short by its holes (a skip
    condition), a stretch of columns without a single return (more columns than a unit loads), a record with a ring id
    that is not its place's, a scan that is not a grid at all -- and beside them a scan whose whole RING is gone (a ring of
    no points is no ring: still read in place)."""
    import torch
    from lidar_feature_extraction_amd import concat
    R, C = 16, 1200
    base = [make_scan(R, C, seed=500 + k) for k in range(6)]
    scans, zeros = [], []
    c, z = _zeroed(base[0], 0.04, 1); scans.append(c); zeros.append(z)                       # plain holes
    c, z = _zeroed(base[1], 0.04, 2, rings=[5]); scans.append(c); zeros.append(z)            # ring 5 is gone
    c, z = _zeroed(base[2], 0.02, 3, columns=np.arange(300, 700)); scans.append(c); zeros.append(z)      # 400 empty columns
    c, z = _zeroed(base[3], 0.02, 4); keepers = np.nonzero((c["ring"] == 7) & ~z)[0][8:]     # ring 7 keeps 8 points: a skip condition
    for fld in ("x", "y", "z"):
        c[fld][keepers] = 0.0
    z = z.copy(); z[keepers] = True
    scans.append(c); zeros.append(z)
    c, z = _zeroed(base[4], 0.03, 5); c["ring"][1234] = (c["ring"][1234] + 3) % R; scans.append(c); zeros.append(z)   # a wrong ring id
    c, z = _zeroed(base[5], 0.03, 6); scans.append(np.ascontiguousarray(c[:-5])); zeros.append(z[:-5])                  # not R x C
    f = FeatureExtraction(device=0, max_points_per_scan=R * C, max_batch=len(scans), max_points_per_ring=C, max_rings=R,
                          drop_zero_points=True, stream_hint=LB.STREAM_GRID_WITH_HOLES)
    dev = torch.device("cuda", 0)
    d = torch.from_numpy(concat(scans).view(np.uint8)).to(dev)
    n = np.array([len(c) for c in scans], np.uint32)
    st = torch.cuda.current_stream().cuda_stream
    for rep in range(3):
        f.extract_batch_device(d.data_ptr(), n, st)
        routes = f.scan_routes(len(scans), st).tolist()
        # (first batch: the hint's route.  Four of six scans fell back, so the report moves the stream to the bucketing route
        # for every scan -- choose_route's "mostly not organised" -- and the results must not care)
        assert rep > 0 or (routes[0] == 3 and routes[1] == 3 and routes[2:] == [0, 0, 0, 0]), routes
        for k in range(len(scans)):
            _check_filtered(f.download(k, st), scans[k], zeros[k], None, "scan %d rep %d" % (k, rep))
    f.close()


@pytest.mark.timeout(120)
def test_non_finite_input_terminates_and_leaves_other_rings_alone(fx):
    """The node requires a dense cloud (feature_extraction.cpp:96-101); NaN / inf coordinates make the
    curvature order inconsistent.  The kernels must still terminate (the pick rounds stop when a round
    picks nothing) and rings without such points must be unaffected."""
    c = make_scan(8, 900, seed=61)
    bad = c.copy()
    r3 = np.nonzero(bad["ring"] == 3)[0]
    r5 = np.nonzero(bad["ring"] == 5)[0]
    bad["x"][r3[100:140]] = np.nan
    bad["y"][r3[400]] = np.inf
    bad["x"][r5[10:700:7]] = np.nan           # NaN curvature almost everywhere in ring 5
    got = fx.ExtractFeatures(bad)             # must return
    ref = fx.ExtractFeatures(c)
    clean = ~np.isin(bad["ring"], [3, 5])
    assert np.array_equal(got.labels[clean], ref.labels[clean])
    assert got.curvature[clean].tobytes() == ref.curvature[clean].tobytes()


def test_other_record_layouts():
    """lfx_layout: any point_step / field offsets (what a PointCloud2 message describes), not only the
    32-byte PointXYZIR record."""
    c = make_scan(16, 800, seed=66, drop_fraction=0.05, shuffle=True)     # shuffled: the sorting path reads z by index
    want = OB.extract(c, canonical_ties=False)
    wide = np.dtype({"names": ["t", "ring", "z", "y", "junk", "x"], "formats": ["<f8", "<u2", "<f4", "<f4", "<u4", "<f4"],
                     "offsets": [0, 10, 12, 20, 28, 40], "itemsize": 48})
    w = np.zeros(len(c), wide)
    w["x"], w["y"], w["z"], w["ring"] = c["x"], c["y"], c["z"], c["ring"]
    w["t"], w["junk"] = np.arange(len(c)), 0xDEADBEEF
    f = FeatureExtraction(device=0, max_points_per_scan=len(c), max_batch=1, max_rings=16, layout=(48, 40, 20, 12, 10))
    assert_scan_equal(f.ExtractFeatures(w), want, "48-byte records")
    f.close()
    tight = np.dtype({"names": ["x", "y", "z", "ring"], "formats": ["<f4", "<f4", "<f4", "<u2"], "offsets": [0, 4, 8, 12],
                      "itemsize": 16})
    t = np.zeros(len(c), tight)
    t["x"], t["y"], t["z"], t["ring"] = c["x"], c["y"], c["z"], c["ring"]
    f = FeatureExtraction(device=0, max_points_per_scan=len(c), max_batch=1, max_rings=16, layout=(16, 0, 4, 8, 12))
    assert_scan_equal(f.ExtractFeatures(t), want, "16-byte records")
    f.close()
    with pytest.raises(LB.LfxError):
        FeatureExtraction(device=0, max_points_per_scan=100, layout=(32, 0, 4, 8, 31))      # ring field leaves the record


def test_driver_clouds_are_read_directly():
    """the repack of the upstream converter node (convert.py:183-212: any field order,
    ring as UINT8, either byte order, all-zero points dropped) happens in the bucketing kernel's loads.
    The expected result is the oracle on the PointXYZIR cloud the converter would have emitted."""
    from lidar_feature_extraction_amd import layout_from_fields
    c = make_scan(16, 900, seed=91, drop_fraction=0.03)
    z = np.zeros(40, POINT_DTYPE)                       # invalid returns of the driver: (0, 0, 0)
    z["ring"] = np.arange(40) % 16
    raw = synth.concat([c[:5000], z, c[5000:]])
    want = OB.extract(c, canonical_ties=False)          # the converter drops the zero points
    # the Ouster-like record of test_convert.py:42-60: ring UINT8 at 26, point_step 48
    ouster = np.dtype({"names": ["x", "y", "z", "intensity", "t", "reflectivity", "ring", "noise", "range"],
                       "formats": ["<f4", "<f4", "<f4", "<f4", "<u4", "<u2", "u1", "<u2", "<u4"],
                       "offsets": [0, 4, 8, 16, 20, 24, 26, 28, 32], "itemsize": 48})
    fields = [("x", 0, 7, 1), ("y", 4, 7, 1), ("z", 8, 7, 1), ("intensity", 16, 7, 1), ("t", 20, 6, 1),
              ("reflectivity", 24, 4, 1), ("ring", 26, 2, 1), ("noise", 28, 4, 1), ("range", 32, 6, 1)]
    o = np.zeros(len(raw), ouster)
    for k in ("x", "y", "z", "intensity", "ring"):
        o[k] = raw[k]
    o["t"], o["noise"] = np.arange(len(raw)), 7
    f = FeatureExtraction(device=0, max_points_per_scan=len(raw), max_batch=1, max_rings=16, drop_zero_points=True,
                          layout=layout_from_fields(fields, 48))
    got = f.ExtractFeatures(o)
    keep = np.nonzero(~((raw["x"] == 0) & (raw["y"] == 0) & (raw["z"] == 0)))[0]
    assert np.array_equal(got.labels[keep], want["labels"])
    assert got.curvature[keep].tobytes() == want["curvature"].tobytes()
    assert np.array_equal(keep[want["edge_index"]], got.edge_index) and np.array_equal(keep[want["surface_index"]], got.surface_index)
    f.close()
    # big-endian message, 22-byte records (fields not naturally aligned), ring INT32
    be = np.dtype({"names": ["x", "ring", "y", "z"], "formats": [">f4", ">i4", ">f4", ">f4"], "offsets": [0, 4, 10, 14],
                   "itemsize": 22})
    b = np.zeros(len(c), be)
    for k in ("x", "y", "z", "ring"):
        b[k] = c[k]
    lay = layout_from_fields([("x", 0, 7, 1), ("ring", 4, 5, 1), ("y", 10, 7, 1), ("z", 14, 7, 1)], 22, is_bigendian=True)
    f = FeatureExtraction(device=0, max_points_per_scan=len(c), max_batch=1, max_rings=16, layout=lay)
    assert_scan_equal(f.ExtractFeatures(b), want, "big-endian 22-byte records")
    f.close()
    with pytest.raises(LB.LfxError) as e:
        layout_from_fields([("x", 0, 7, 1), ("y", 4, 7, 1), ("z", 8, 7, 1), ("intensity", 16, 7, 1)], 32)
    assert e.value.code == -7                            # no ring channel: the node shuts down (feature_extraction.cpp:103-108)


def test_wire_payloads_of_the_three_published_clouds(fx):
    """scan_edge / scan_surface as pcl::PointXYZ records and colored_scan as pcl::PointXYZRGB records
    (feature_extraction.cpp:153-170), built on the device; expected values from the oracle's labels."""
    import torch
    clouds = [make_scan(16, 900, seed=95), make_scan(16, 900, seed=96, drop_fraction=0.05)]
    few = clouds[1]["ring"] == 7
    clouds[1] = synth.concat([clouds[1][~few], clouds[1][few][:4]])    # ring 7: 4 points, removed as sparse
    host = synth.concat(clouds).view(np.uint8)
    dev = torch.from_numpy(host.copy()).to("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    fx.extract_batch_device(dev.data_ptr(), [len(c) for c in clouds], stream)
    cap = sum(len(c) for c in clouds)
    e = torch.zeros((cap, 4), dtype=torch.float32, device="cuda:0")
    s = torch.zeros((cap, 4), dtype=torch.float32, device="cuda:0")
    offs = torch.zeros(2 * 3, dtype=torch.int32, device="cuda:0")
    col = torch.zeros((cap, 8), dtype=torch.float32, device="cuda:0")
    coffs = torch.zeros(3, dtype=torch.int32, device="cuda:0")
    fx.pack_xyz(e.data_ptr(), s.data_ptr(), offs.data_ptr(), cap, stream)
    fx.pack_colored(col.data_ptr(), coffs.data_ptr(), cap, stream)
    e12 = torch.zeros((cap, 3), dtype=torch.float32, device="cuda:0")
    s12 = torch.zeros((cap, 3), dtype=torch.float32, device="cuda:0")
    offs12 = torch.zeros(2 * 3, dtype=torch.int32, device="cuda:0")
    fx.pack_xyz12(e12.data_ptr(), s12.data_ptr(), offs12.data_ptr(), cap, stream)
    torch.cuda.synchronize()
    e, s, offs, col, coffs = e.cpu().numpy(), s.cpu().numpy(), offs.cpu().numpy(), col.cpu().numpy(), coffs.cpu().numpy()
    assert np.array_equal(offs12.cpu().numpy(), offs)                      # the tight gather payload: same points, 12 bytes each
    assert np.array_equal(e12.cpu().numpy()[:offs[2]], e[:offs[2], :3]) and np.array_equal(s12.cpu().numpy()[:offs[5]], s[:offs[5], :3])
    for i, c in enumerate(clouds):
        want = OB.extract(c, canonical_ties=False)
        for arr, o0, o1, idx in ((e, offs[i], offs[i + 1], want["edge_index"]), (s, offs[3 + i], offs[3 + i + 1], want["surface_index"])):
            assert o1 - o0 == len(idx)
            assert np.array_equal(arr[o0:o1, 0], c["x"][idx]) and np.array_equal(arr[o0:o1, 1], c["y"][idx])
            assert np.array_equal(arr[o0:o1, 2], c["z"][idx]) and np.all(arr[o0:o1, 3] == 1.0)
        # colored_scan: the labelled rings' points, rings ascending, angle ascending
        order = []
        at = 0
        for rid, cnt, st in zip(want["ring_id"], want["ring_count"], want["ring_status"]):
            if st == 0:
                order.extend(want["sorted_index"][at:at + cnt])
            at += cnt
        order = np.asarray(order, dtype=np.int64)
        got = col[coffs[i]:coffs[i + 1]]
        assert len(got) == len(order)
        assert np.array_equal(got[:, 0], c["x"][order]) and np.array_equal(got[:, 1], c["y"][order])
        assert np.array_equal(got[:, 2], c["z"][order]) and np.all(got[:, 3] == 1.0) and np.all(got[:, 5:] == 0.0)
        rgba = got[:, 4].copy().view(np.uint32)
        assert np.all(rgba >> 24 == 255)
        lab = want["labels"][order]
        for v in np.unique(lab):
            rgb = np.zeros(3, np.uint8)
            assert LB.load().lfx_label_to_color(int(v), rgb.ctypes.data_as(LB.C.POINTER(LB.C.c_uint8))) == 0
            m = lab == v
            assert np.all((rgba[m] >> 16) & 255 == rgb[0]) and np.all((rgba[m] >> 8) & 255 == rgb[1]) and np.all(rgba[m] & 255 == rgb[2])
    w1 = OB.extract(clouds[1], canonical_ties=False)
    assert 7 not in [int(r) for r, st in zip(w1["ring_id"], w1["ring_status"]) if st == 0]    # the sparse ring is left out


def test_wire_payloads_and_route_word_of_a_grid_with_holes():
    """The consumers of a scan read in place as a grid with zero records (route 3): colored_scan takes x, y, z from the
    input records and the index from sorted_index; the published route word carries LFX_SCAN_GRID_WITH_HOLES without
    LFX_SCAN_ORGANISED, so that a caller written against the two-route contract ((bits & 0x300) == 0x100: positions are
    columns, no index array) treats it as a scan whose sorted_index is valid -- which it is."""
    import torch
    R, C = 16, 900
    c, zero = _zeroed(make_scan(R, C, seed=97), 0.05, 9)
    keep = np.nonzero(~zero)[0]
    want = OB.extract(np.ascontiguousarray(c[keep]), canonical_ties=False)
    f = FeatureExtraction(device=0, max_points_per_scan=len(c), max_batch=1, max_points_per_ring=C, max_rings=R, drop_zero_points=True,
                          stream_hint=LB.STREAM_GRID_WITH_HOLES)
    dev = torch.from_numpy(c.view(np.uint8).copy()).to("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    f.extract_batch_device(dev.data_ptr(), [len(c)], st)
    assert f.scan_routes(1, st).tolist() == [3]
    v = f.device_view()
    info = torch.zeros(4, dtype=torch.int32, device="cuda:0")
    hip = LB.C.CDLL("libamdhip64.so")
    hip.hipMemcpy(LB.C.c_void_p(info.data_ptr()), LB.C.cast(v.scan_info, LB.C.c_void_p), 16, 3)
    bits = int(info.cpu().numpy().view(np.uint32)[1])
    assert (bits & 0x300) != 0x100 and (bits & 0x800) == 0x800, hex(bits)
    sidx = torch.zeros(R * ((C + 63) // 64 * 64), dtype=torch.int32, device="cuda:0")
    hip.hipMemcpy(LB.C.c_void_p(sidx.data_ptr()), LB.C.cast(v.sorted_index, LB.C.c_void_p), sidx.numel() * 4, 3)
    cap = v.ring_capacity
    sidx = sidx.cpu().numpy().view(np.uint32)
    at = 0
    for rid, cnt in zip(want["ring_id"], want["ring_count"]):
        assert np.array_equal(sidx[rid * cap:rid * cap + cnt], keep[want["sorted_index"][at:at + cnt]].astype(np.uint32)), "sorted_index of ring %d" % rid
        at += cnt
    n = len(c)
    col = torch.zeros((n, 8), dtype=torch.float32, device="cuda:0")
    coffs = torch.zeros(2, dtype=torch.int32, device="cuda:0")
    f.pack_colored(col.data_ptr(), coffs.data_ptr(), n, st)
    e = torch.zeros((n, 4), dtype=torch.float32, device="cuda:0")
    s2 = torch.zeros((n, 4), dtype=torch.float32, device="cuda:0")
    offs = torch.zeros(4, dtype=torch.int32, device="cuda:0")
    f.pack_xyz(e.data_ptr(), s2.data_ptr(), offs.data_ptr(), n, st)
    torch.cuda.synchronize()
    col, coffs, e, s2, offs = col.cpu().numpy(), coffs.cpu().numpy(), e.cpu().numpy(), s2.cpu().numpy(), offs.cpu().numpy()
    order = []
    at = 0
    for rid, cnt, stt in zip(want["ring_id"], want["ring_count"], want["ring_status"]):
        if stt == 0:
            order.extend(keep[want["sorted_index"][at:at + cnt]])
        at += cnt
    order = np.asarray(order, dtype=np.int64)
    got = col[coffs[0]:coffs[1]]
    assert len(got) == len(order)
    assert np.array_equal(got[:, 0], c["x"][order]) and np.array_equal(got[:, 1], c["y"][order]) and np.array_equal(got[:, 2], c["z"][order])
    ei, si = keep[want["edge_index"]], keep[want["surface_index"]]
    assert offs[1] - offs[0] == len(ei) and offs[3] - offs[2] == len(si)
    assert np.array_equal(e[offs[0]:offs[1], 0], c["x"][ei]) and np.array_equal(s2[offs[2]:offs[3], 2], c["z"][si])
    f.close()


def test_ring_id_beyond_max_rings_is_an_error():
    c = make_scan(8, 300, seed=3)
    f = FeatureExtraction(device=0, max_points_per_scan=len(c), max_batch=1, max_rings=4)
    with pytest.raises(LB.LfxError) as e:
        f.ExtractFeatures(c)
    assert e.value.code == -5
    f.close()


def test_many_chunks_look_back():
    """A long scan (128 chunks of 2048 points) exercises the look-back over many predecessors."""
    c = make_scan(128, 2048, seed=90, vfov_deg=22.5)
    f = FeatureExtraction(device=0, max_points_per_scan=len(c), max_batch=2, max_points_per_ring=2048, max_rings=128)
    got = f.extract_batch([c, c[:100000]])
    assert_scan_equal(got[0], OB.extract(c, canonical_ties=False), "128x2048")
    c2 = np.ascontiguousarray(c[:100000])
    assert_scan_equal(got[1], OB.extract(c2, canonical_ties=False), "128x2048-truncated")
    f.close()


@pytest.mark.parametrize("smallest_step", [1e-6, 1e-9])
def test_values_next_to_the_thresholds(fx, smallest_step):
    """The link and parallel-beam tests are pre-classified with cheap f32 arithmetic and only values
    next to a threshold take the exact f64 division.  Rings built to sit next to every threshold:
    angular steps around neighbor_degree_threshold and around zero, range ratios around
    parallel_beam_min_range_ratio, range jumps around distance_diff_threshold, ranges around min/max."""
    rng = np.random.default_rng(123)
    hp = HyperParameters()
    theta = math.radians(hp.neighbor_degree_threshold)
    rings = []
    for rid in range(12):
        n = 1500
        steps = np.full(n, math.radians(0.2))
        k = rng.integers(0, n, 500)
        # steps within a few 1e-8 rad of the threshold angle, and tiny steps down to 1e-9 rad
        steps[k[:250]] = theta + rng.integers(-40, 41, 250) * 1e-9
        steps[k[250:]] = 10.0 ** rng.uniform(math.log10(smallest_step), -4, 250)
        az = -3.0 + np.cumsum(steps)
        az = az[az < 3.1]
        n = len(az)
        r = np.full(n, 8.0) + 0.01 * rng.standard_normal(n)
        j = rng.integers(1, n - 1, 300)
        # both neighbours differ by ratio * (1 + tiny) -> parallel-beam test next to its threshold
        eps = rng.integers(-30, 31, 300) * 1e-8
        r[j] = r[j - 1] / (1.0 + hp.parallel_beam_min_range_ratio * (1.0 + eps))
        r[j + 1] = r[j - 1]
        j2 = rng.integers(1, n - 1, 200)
        r[j2] = r[j2 - 1] + hp.distance_diff_threshold + rng.integers(-20, 21, 200) * 1e-8     # occlusion jumps
        j3 = rng.integers(0, n, 100)
        r[j3[:50]] = hp.min_range + rng.integers(-20, 21, 50) * 1e-9
        r[j3[50:]] = hp.max_range + rng.integers(-20, 21, 50) * 1e-7
        rings.append((rid, (r * np.cos(az)).astype(np.float32), (r * np.sin(az)).astype(np.float32)))
    ring = np.concatenate([np.full(len(x), rid, np.uint16) for rid, x, _y in rings])
    c = _cloud(ring, np.concatenate([x for _r, x, _y in rings]), np.concatenate([y for _r, _x, y in rings]))
    want = OB.extract(c, oracle_params(hp), canonical_ties=True)
    if smallest_step >= 1e-6:
        assert want["angle_ties"] == 0          # every ring is strictly angle-sorted: the wave-per-unit path takes them
    else:
        assert want["angle_ties"] > 0           # directions that coincide in f32: the sorting (slow) path takes those rings
    got = fx.ExtractFeatures(c)
    assert_scan_equal(got, want, "thresholds[ties]")
    counts = np.bincount(got.labels, minlength=8)
    assert counts[5] > 0 and counts[6] > 0 and counts[7] > 0      # out-of-range, occluded, parallel-beam all hit


def test_f64_sqrt_and_divide_are_correctly_rounded(fx):
    """Range (math.hpp:36-39) and the link cosine feed orderings: they must equal IEEE results."""
    rng = np.random.default_rng(2)
    n = 4096
    x = (rng.standard_normal(n) * 10 ** rng.uniform(-3, 3, n)).astype(np.float32)
    y = (rng.standard_normal(n) * 10 ** rng.uniform(-3, 3, n)).astype(np.float32)
    out = fx.stage_ring(x, y, 0)
    want = np.sqrt(x.astype(np.float64) ** 2 + y.astype(np.float64) ** 2)
    assert out["range"].tobytes() == want.tobytes()


# ------------------------------------------------------------------ properties at full size
def test_full_size_properties():
    """64x1800 (BASELINE config) x 8 scans: size-independent properties of the outputs."""
    hp = HyperParameters()
    clouds = [make_scan(64, 1800, seed=500 + i) for i in range(8)]
    f = FeatureExtraction(hp, device=0, max_points_per_scan=64 * 1800, max_batch=8, max_points_per_ring=2048, max_rings=64)
    got = f.extract_batch(clouds)
    again = f.extract_batch(clouds)
    for c, g, g2 in zip(clouds, got, again):
        n = len(c)
        assert g.labels.tobytes() == g2.labels.tobytes() and g.curvature.tobytes() == g2.curvature.tobytes()   # deterministic
        assert np.array_equal(np.sort(g.sorted_index), np.arange(n))                                           # a permutation
        assert np.array_equal(np.nonzero(g.labels == 1)[0], np.sort(g.edge_index))
        assert np.array_equal(np.nonzero(g.labels == 3)[0], np.sort(g.surface_index))
        # ring-major, angle-ascending emission order
        pos = np.empty(n, np.int64)
        pos[g.sorted_index] = np.arange(n)
        assert np.all(np.diff(pos[g.edge_index]) > 0) and np.all(np.diff(pos[g.surface_index]) > 0)
        ang = np.arctan2(c["y"][g.sorted_index].astype(np.float64), c["x"][g.sorted_index].astype(np.float64))
        for off, cnt in zip(g.ring_offset, g.ring_count):
            assert np.all(np.diff(ang[off:off + cnt]) > 0)
        # feature clouds carry the input coordinates and the narrowed curvature
        assert np.array_equal(g.edge_points[:, 0], c["x"][g.edge_index]) and np.array_equal(g.edge_points[:, 2], c["z"][g.edge_index])
        assert np.array_equal(g.edge_points[:, 3], g.curvature[g.edge_index].astype(np.float32))
        assert np.all(g.curvature[g.edge_index] >= hp.edge_threshold) and np.all(g.curvature[g.surface_index] <= hp.surface_threshold)
        # no two picks of one kind within reach: adjacent picks in a ring are > padding apart unless a link is broken
        # (checked exactly by the oracle comparison below on one scan)
    assert_scan_equal(got[3], OB.extract(clouds[3], canonical_ties=False), "full-size")
    f.close()


def _scan_properties(c, g, hp, ctx):
    """Size-independent properties of one scan's outputs (no oracle)."""
    n = len(c)
    assert np.array_equal(np.sort(g.sorted_index), np.arange(n)), ctx                                   # a permutation
    assert np.array_equal(np.nonzero(g.labels == 1)[0], np.sort(g.edge_index)), ctx
    assert np.array_equal(np.nonzero(g.labels == 3)[0], np.sort(g.surface_index)), ctx
    pos = np.empty(n, np.int64)
    pos[g.sorted_index] = np.arange(n)
    assert np.all(np.diff(pos[g.edge_index]) > 0) and np.all(np.diff(pos[g.surface_index]) > 0), ctx     # emission order
    assert np.array_equal(c["ring"][g.sorted_index], np.repeat(g.ring_id, g.ring_count)), ctx           # ring-major
    assert np.array_equal(g.edge_points[:, 0], c["x"][g.edge_index]) and np.array_equal(g.edge_points[:, 1], c["y"][g.edge_index]), ctx
    assert np.array_equal(g.edge_points[:, 2], c["z"][g.edge_index]), ctx
    assert np.array_equal(g.surface_points[:, 2], c["z"][g.surface_index]), ctx
    assert np.array_equal(g.edge_points[:, 3], g.curvature[g.edge_index].astype(np.float32)), ctx
    assert np.array_equal(g.surface_points[:, 3], g.curvature[g.surface_index].astype(np.float32)), ctx
    assert np.all(g.curvature[g.edge_index] >= hp.edge_threshold) and np.all(g.curvature[g.surface_index] <= hp.surface_threshold), ctx


@pytest.mark.parametrize("pname", list(PARAM_SETS))
def test_config_os1_128x2048_batch32(pname):
    """BASELINE.json configs[3]: OS1-128-shaped 128-ring x 2048-column scans, batch = 32 scans in ONE call
    (8.4 M points: 128 bucketing chunks per scan, ~3.7 GB of context scratch).  Both parameter sets.  Every scan:
    size-independent properties and equality with a second run; scans 0, 9, 18, 27 and 31: the CPU oracle."""
    hp = PARAM_SETS[pname]
    rings, cols, batch = 128, 2048, 32
    clouds = [make_scan(rings, cols, seed=4000 + i, vfov_deg=22.5) for i in range(batch)]
    f = FeatureExtraction(hp, device=0, max_points_per_scan=rings * cols, max_batch=batch, max_points_per_ring=cols,
                          max_rings=rings)
    got = f.extract_batch(clouds)
    again = f.extract_batch(clouds)
    f.close()
    for i, (c, g, g2) in enumerate(zip(clouds, got, again)):
        ctx = "128x2048x32/%s/scan%d" % (pname, i)
        assert g.labels.tobytes() == g2.labels.tobytes() and g.curvature.tobytes() == g2.curvature.tobytes(), ctx
        assert np.array_equal(g.edge_index, g2.edge_index) and np.array_equal(g.surface_index, g2.surface_index), ctx
        assert len(g.ring_id) == rings and g.ring_count.tolist() == [cols] * rings and not g.ring_status.any(), ctx
        _scan_properties(c, g, hp, ctx)
        assert len(g.edge_index) > 0 and len(g.surface_index) > 0, ctx
    for i in (0, 9, 18, 27, 31):
        want = OB.extract(clouds[i], oracle_params(hp), canonical_ties=False)
        assert want["angle_ties"] == 0 and want["curvature_ties"] == 0
        assert_scan_equal(got[i], want, "128x2048x32/%s/scan%d" % (pname, i))


@pytest.mark.timeout(900)
def test_stress_slice():
    """A seeded 200-draw slice of tools/stress.py: random sensor shapes (4-64 rings x 150-2600 columns), input
    orders (sorted, rotated, reversed, both, shuffled, ragged), noise levels and all nine hyper-parameters; every
    output against the oracle, each batch extracted twice (the second call may take another route)."""
    from tests.stress_cases import ORDERS, run_cases
    seen = run_cases(200, seed=20261004)
    assert set(seen) == set(ORDERS), seen


# ------------------------------------------------------------------ reference unit vectors on the device
def test_refvec_curvature_convolution(fx, refvec):
    for c in refvec["calc_curvature"]["cases"]:
        r = np.asarray(c["range"], np.float64)
        hp = HyperParameters(padding=c["padding"])
        out = fx.stage_ring(r.astype(np.float32), np.zeros(len(r), np.float32), LB.STAGE_CURVATURE, hp, range_in=r)
        assert out["status"] == 0 and out["curvature"].tolist() == c["expect"]
    for c in refvec["make_weight"]["cases"]:
        # MakeWeight is the stencil the device applies: an impulse response reveals it
        p = c["padding"]
        imp = np.zeros(4 * p + 1)
        imp[2 * p] = 1.0
        w = fx.convolution1d(imp, np.asarray(c["expect"], np.float64))
        assert w[p:3 * p + 1].tolist() == c["expect"][::-1]
    for c in refvec["convolution1d"]["cases"]:
        if c.get("throws"):
            with pytest.raises(LB.LfxError) as e:
                fx.convolution1d(c["input"], c["weight"])
            assert c["message"] in str(e.value)          # the throw text pinned by test_convolution.cpp:61-70
        else:
            assert fx.convolution1d(c["input"], c["weight"]).tolist() == c["expect"]


def test_refvec_range_and_neighbor(fx, refvec):
    for c in refvec["xy_norm"]["cases"]:
        out = fx.stage_ring([c["x"], 1.0], [c["y"], 1.0], 0)
        assert out["range"][0] == c["expect"]
    pts = refvec["range"]["points_xy"]
    x, y = xy(pts)
    out = fx.stage_ring(x, y, 0)
    for (px, py), r in zip(pts, out["range"]):
        assert abs(r - math.sqrt(px * px + py * py)) < refvec["range"]["tolerance"]
    for c in refvec["is_neighbor_xy"]["cases"]:
        thr = c["threshold"] if "threshold" in c else math.pi / 2 + c["threshold_pi_half_plus"]
        hp = HyperParameters(neighbor_degree_threshold=math.degrees(thr))
        x, y = xy([c["p0"], c["p1"]])
        assert bool(fx.stage_ring(x, y, 0, hp)["link"][0]) == c["expect"]
    n = refvec["neighbor_check_xy"]
    for c in n["cases"]:
        thr = c["threshold"] if "threshold" in c else math.pi / 4 + c["threshold_pi_quarter_plus"]
        sl = n["points_xy"][c["slice"][0]:c["slice"][1]] if "slice" in c else n["points_xy"]
        i, j = c["pair"]
        assert j == i + 1
        x, y = xy(sl)
        hp = HyperParameters(neighbor_degree_threshold=math.degrees(thr))
        assert bool(fx.stage_ring(x, y, 0, hp)["link"][i]) == c["expect"]
    # CalcRadian vectors (test_math.cpp:42-65) through the link test: acos(...) < expected +- tol
    tol = refvec["calc_radian"]["tolerance"]
    for c in refvec["calc_radian"]["cases"]:
        a = c["args"]
        x, y = xy([[a[0], a[1]], [a[2], a[3]]])
        if c.get("throws"):
            out = fx.stage_ring(x, y, LB.STAGE_OCCLUSION, HyperParameters(padding=1))
            assert out["status"] == 5
            continue
        want = c["expect_pi_times"] * math.pi
        above = fx.stage_ring(x, y, 0, HyperParameters(neighbor_degree_threshold=math.degrees(want + tol)))["link"][0]
        assert bool(above)
        if want - tol > 0:
            below = fx.stage_ring(x, y, 0, HyperParameters(neighbor_degree_threshold=math.degrees(want - tol)))["link"][0]
            assert not bool(below)


def test_refvec_masks(fx, refvec):
    for c in refvec["out_of_range"]["cases"]:
        x, y = xy(c["points_xy"])
        hp = HyperParameters(min_range=c["min_range"], max_range=c["max_range"])
        assert names(fx.stage_ring(x, y, LB.STAGE_OUT_OF_RANGE, hp)["labels"]) == c["expect"]
    for c in refvec["parallel_beam"]["cases"]:
        x, y = xy(c["points_xy"])
        hp = HyperParameters(parallel_beam_min_range_ratio=c["threshold"])
        assert names(fx.stage_ring(x, y, LB.STAGE_PARALLEL_BEAM, hp)["labels"]) == c["expect"]
    o = refvec["occlusion"]
    for c in o["cases"]:
        x, y = xy(c["points_xy"])
        hp = HyperParameters(padding=c["padding"], neighbor_degree_threshold=math.degrees(o["neighbor_radian_threshold"]),
                             distance_diff_threshold=o["distance_threshold"])
        out = fx.stage_ring(x, y, LB.STAGE_OCCLUSION, hp)
        assert out["status"] == 0 and names(out["labels"]) == c["expect"]
    # a checker over a single point throws (test_neighbor.cpp:144-162): status, not labels
    x, y = xy(refvec["neighbor_check_xy"]["too_few_points_throws"]["points_xy"])
    assert fx.stage_ring(x, y, LB.STAGE_OCCLUSION, HyperParameters(padding=1))["status"] != 0


def test_refvec_labelling(fx, refvec):
    flags = LB.STAGE_LABEL | LB.STAGE_SINGLE_BLOCK
    for c in refvec["edge_label"]["cases"]:
        n = len(c["groups"])
        hp = HyperParameters(padding=c["padding"], edge_threshold=c["threshold"], surface_threshold=1e-300)
        cv = np.asarray(c["curvature"], np.float64) + 1.0        # keep every value above the surface threshold
        hp.edge_threshold = c["threshold"] + 1.0
        out = fx.stage_ring(np.ones(n), np.zeros(n), flags, hp, groups=c["groups"], curvature_in=cv)
        assert out["status"] == 0 and names(out["labels"]) == c["expect"]
    # FillNeighbors (fill.hpp:101-117) = what one pick does: make `index` the only candidate
    for c in refvec["fill_neighbors"]["cases"]:
        n = len(c["groups"])
        cv = np.full(n, 1.0)
        cv[c["index"]] = 5.0
        hp = HyperParameters(padding=c["padding"], edge_threshold=3.0, surface_threshold=1e-300)
        out = fx.stage_ring(np.ones(n), np.zeros(n), flags, hp, groups=c["groups"], curvature_in=cv)
        lab = names(out["labels"])
        assert lab[c["index"]] == "Edge"
        lab[c["index"]] = "EdgeNeighbor"
        assert lab == c["expect"]
    # FillFromLeft / FillFromRight (fill.hpp:40-99) are the two halves of that fill; their vectors
    # are replayed as picks at the fill's origin with the other half cut off by a group change
    for which in ("fill_from_left", "fill_from_right"):
        for c in refvec[which]["cases"]:
            if c.get("throws") or c["label"] == "Default":
                continue
            g = list(c["groups"])
            n = len(g)
            if which == "fill_from_left":
                origin, length = c["begin"], c["end"] - c["begin"]
                g2 = [(-1 - k) for k in range(origin)] + g[origin:]          # nothing to the left is linked
            else:
                origin, length = c["end"], c["end"] - c["begin"]
                g2 = g[:origin + 1] + [(-1 - k) for k in range(n - origin - 1)]
            cv = np.full(n, 1.0)
            cv[origin] = 5.0
            hp = HyperParameters(padding=max(length - 1, 1), edge_threshold=3.0, surface_threshold=1e-300)
            if length - 1 < 1:
                continue
            out = fx.stage_ring(np.ones(n), np.zeros(n), flags, hp, groups=g2, curvature_in=cv)
            lab = ["Edge" if v in ("Edge", "EdgeNeighbor") else v for v in names(out["labels"])]
            assert lab == c["expect"], (which, c)


def test_refvec_argsort_order_through_labelling(fx, refvec):
    """Argsort vectors (test_algorithm.cpp:36-49): the device never sorts, but visiting order is
    observable: with padding covering the whole array and all links intact, the single pick of an
    edge pass is the argmax (last of argsort), of a surface pass the argmin (first of argsort)."""
    flags = LB.STAGE_LABEL | LB.STAGE_SINGLE_BLOCK
    for c in refvec["argsort"]["cases"]:
        v = np.asarray(c["values"], np.float64) + 1.0
        n = len(v)
        g = np.zeros(n, np.int32)
        hp = HyperParameters(padding=n, edge_threshold=0.5, surface_threshold=1e-300)
        lab = fx.stage_ring(np.ones(n), np.zeros(n), flags, hp, groups=g, curvature_in=v)["labels"]
        if len(set(c["values"])) == n:
            assert int(np.nonzero(lab == LAB["Edge"])[0][0]) == c["expect"][-1]
        hp = HyperParameters(padding=n, edge_threshold=1e300, surface_threshold=1e300)
        lab = fx.stage_ring(np.ones(n), np.zeros(n), flags, hp, groups=g, curvature_in=v)["labels"]
        assert int(np.nonzero(lab == LAB["Surface"])[0][0]) == c["expect"][0]


def test_refvec_index_range_through_labelling(fx, refvec):
    """PaddedIndexRange vectors (test_index_range.cpp:147-176): block boundaries are observable as
    the places where a fill stops although every link is intact."""
    for c in refvec["padded_index_range"]["cases"]:
        n, nb, pad, bounds = c["size"], c["n_blocks"], c["padding"], c["bounds"]
        az = np.linspace(-1.0, 1.0, n)
        x, y = (5 * np.cos(az)).astype(np.float32), (5 * np.sin(az)).astype(np.float32)
        hp = HyperParameters(padding=pad, n_blocks=nb, neighbor_degree_threshold=90.0, edge_threshold=1e300,
                             surface_threshold=1e300)
        out = fx.stage_ring(x, y, LB.STAGE_LABEL | LB.STAGE_CURVATURE, hp)
        lab = out["labels"]
        assert out["status"] == 0
        assert not lab[:bounds[0]].any() and not lab[bounds[-1]:].any()      # borders stay Default
        assert lab[bounds[0]:bounds[-1]].all()                               # every block point is labelled
        oc = OB.lib()
        want = np.zeros(n, np.uint8)
        cv = np.ascontiguousarray(out["curvature"])
        import ctypes as C
        assert oc.orc_assign_label(want.ctypes.data_as(C.POINTER(C.c_uint8)), cv.ctypes.data_as(C.POINTER(C.c_double)), n,
                                   x.ctypes.data_as(C.POINTER(C.c_float)), y.ctypes.data_as(C.POINTER(C.c_float)),
                                   math.radians(90.0), nb, pad, 1e300, 1e300) == 0
        assert lab.tolist() == want.tolist()


def test_refvec_ring_projection(fx, refvec):
    e = refvec["extract_angle_sorted_rings"]
    pts = np.zeros(len(e["points_ring_xy"]), POINT_DTYPE)
    for k, (r, px, py) in enumerate(e["points_ring_xy"]):
        pts[k]["ring"], pts[k]["x"], pts[k]["y"] = r, px, py
    got = fx.ring_projection(pts)
    assert {str(k): v.tolist() for k, v in got.items()} == e["expect"]
    s = refvec["sort_by_atan2"]
    pts = np.zeros(len(s["points_xy"]), POINT_DTYPE)
    for k, (px, py) in enumerate(s["points_xy"]):
        pts[k]["x"], pts[k]["y"] = px, py
    assert fx.ring_projection(pts)[0].tolist() == s["expect"]
    # the predicate against atan2 on the special pairs (test_ring.cpp:47-101): sort each pair
    for a, b in refvec["polar_less_specific"]["pairs"]:
        pts = np.zeros(2, POINT_DTYPE)
        pts[0]["x"], pts[0]["y"], pts[1]["x"], pts[1]["y"] = a[0], a[1], b[0], b[1]
        order = fx.ring_projection(pts)[0].tolist()
        ta, tb = math.atan2(a[1], a[0]), math.atan2(b[1], b[0])
        if ta < tb:
            assert order == [0, 1], (a, b)
        elif tb < ta:
            assert order == [1, 0], (a, b)
    # 10000 random points (test_ring.cpp:103-127), in float: equal to sorting by atan2
    n = refvec["polar_less_random"]["n"]
    rng = np.random.default_rng(0)
    for chunk in range(3):
        m = 4000 if chunk < 2 else n - 8000
        pts = np.zeros(m, POINT_DTYPE)
        pts["x"] = rng.uniform(-1, 1, m).astype(np.float32)
        pts["y"] = rng.uniform(-1, 1, m).astype(np.float32)
        order = fx.ring_projection(pts)[0]
        want = np.argsort(np.arctan2(pts["y"].astype(np.float64), pts["x"].astype(np.float64)), kind="stable")
        assert np.array_equal(order, want)
    rs = refvec["remove_sparse_rings"]
    ring, x, y = [], [], []
    for rid, size in rs["ring_sizes"].items():
        for k in range(size):
            ring.append(int(rid)); x.append(math.cos(0.1 * k + 0.05)); y.append(math.sin(0.1 * k + 0.05))
    cloud = _cloud(np.array(ring, np.uint16), np.array(x, np.float32), np.array(y, np.float32))
    for c in rs["cases"]:
        f = FeatureExtraction(HyperParameters(padding=c["n_min_points"] - 1), device=0, max_points_per_scan=64, max_batch=1)
        g = f.ExtractFeatures(cloud)
        assert [int(r) for r, s in zip(g.ring_id, g.ring_status) if s != 1] == c["kept"]
        f.close()


def test_refvec_append_xyzir(fx, refvec):
    """AppendXYZIR (test_label.cpp:55-74): the feature clouds carry x,y,z and (float)curvature."""
    c = make_scan(4, 400, seed=12)
    g = fx.ExtractFeatures(c)
    assert len(g.edge_index) > 0
    assert np.array_equal(g.edge_points[:, 3], g.curvature[g.edge_index].astype(np.float32))
    assert np.array_equal(g.edge_points[:, 1], c["y"][g.edge_index])
    a = refvec["append_xyzir"]
    for (x, y, z, _i, _r), cv, want in zip(a["points_xyzir"], a["curvature"], a["expect_xyzir"]):
        assert [np.float32(x), np.float32(y), np.float32(z), np.float32(cv)] == want[:4]


def test_stream_hint_spares_the_first_batch_the_slow_route():
    """lfx_config.stream_hint: a caller that knows its driver's order says so, and the FIRST batch already takes the route the
    library would otherwise reach from the reports of a batch or two (lfx_scan_routes: 2 = in place through ring transforms,
    0 = bucketed, 1 = in place)."""
    rot = [make_scan(32, 1024, seed=8100 + k, start_col=300) for k in range(3)]
    want = [OB.extract(c, canonical_ties=False) for c in rot]
    for hint, first in ((LB.STREAM_UNKNOWN, [0, 0, 0]), (LB.STREAM_TURNED_RINGS, [2, 2, 2]), (LB.STREAM_NO_GRID, [0, 0, 0])):
        f = FeatureExtraction(device=0, max_points_per_scan=32 * 1024, max_batch=3, max_points_per_ring=1024, max_rings=32, stream_hint=hint)
        res = f.extract_batch(rot)
        assert f.scan_routes(3).tolist() == first, (hint, f.scan_routes(3).tolist())
        for got, w in zip(res, want):
            assert_scan_equal(got, w)
        if hint == LB.STREAM_NO_GRID:
            # (no organised-scan launch at all for such a stream: the kernel's time stays zero)
            f.set_profiling(True)
            f.extract_batch(rot)
            assert f.kernel_times()["ring_unit_org_kernel"][1] == 0
        f.close()
    with pytest.raises(LB.LfxError):
        FeatureExtraction(device=0, max_points_per_scan=1000, stream_hint=7)
