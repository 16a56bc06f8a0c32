"""Randomised parity cases shared by tests/test_gpu_parity.py (a seeded slice under pytest -m gpu) and
tools/stress.py (the long sweep): random sensor shapes, input orders, sensor noise and all nine
hyper-parameters; every output of the HIP path against the CPU oracle (tests/parity.assert_scan_equal)."""
import time

import numpy as np

from lidar_feature_extraction_amd import FeatureExtraction, HyperParameters, make_scan
from oracle import binding as OB
from lidar_feature_extraction_amd import binding as LB
from tests.parity import assert_scan_equal, assert_filtered_equal

# ("zeros": a grid whose invalid returns are (0, 0, 0) records, zero filter on -- the holes form of the organised route)
ORDERS = ["sorted", "rotated", "reversed", "revrot", "shuffled", "ragged", "ragrot", "zeros", "zeros"]


def draw(rng):
    """One case: (rings, cols, HyperParameters, order, make_scan kwargs, seed, sigma, exact_cap)."""
    rings = int(rng.choice([4, 8, 16, 32, 64]))
    cols = int(rng.integers(150, 2600))
    if rng.integers(0, 6) == 0:                # long rings: the 12-chunk form of the unit kernels, or (few blocks) the workgroup-per-ring kernel
        cols = int(rng.integers(2600, 4609))        # (up to LFX_MAX_RING_POINTS)
        rings = min(rings, 16)
    P = int(rng.choice([1, 2, 3, 5, 5, 5, 8, 15, 20, 33]))      # (16 and up: beyond the windows, the workgroup-per-ring kernel for every ring)
    B = int(rng.choice([1, 2, 3, 6, 6, 6, 9, 17, 40]))
    hp = HyperParameters(padding=P, n_blocks=B,
                         neighbor_degree_threshold=float(rng.uniform(0.5, 6.0)),
                         distance_diff_threshold=float(rng.uniform(0.05, 1.0)),
                         parallel_beam_min_range_ratio=float(rng.uniform(0.005, 0.2)),
                         edge_threshold=float(rng.choice([0.01, 0.05, 0.1, 0.5])),
                         surface_threshold=float(rng.choice([0.001, 0.01, 0.1])),
                         min_range=float(rng.uniform(0.05, 1.0)), max_range=float(rng.choice([50.0, 100.0, 1000.0])))
    kw = {}
    order = str(rng.choice(ORDERS))
    if order in ("rotated", "revrot", "ragrot"):
        kw["start_col"] = int(rng.integers(1, cols))
    if order in ("reversed", "revrot"):
        kw["reverse"] = True
    if order == "shuffled":
        kw["shuffle"] = True
    if order in ("ragged", "ragrot"):
        kw["drop_fraction"] = float(rng.uniform(0.01, 0.4))
    if order == "zeros":
        # (share of the returns zeroed, a stretch of columns without a return, whether the context is told)
        kw["_zeros"] = (float(rng.choice([0.0, 0.01, 0.05, 0.05, 0.15, 0.3])), int(rng.integers(0, 3)) == 0, int(rng.integers(0, 2)) == 0)
    seed = int(rng.integers(1, 1 << 30))
    sigma = float(rng.choice([0.01, 0.002, 0.03]))
    exact_cap = bool(rng.integers(0, 2))
    return rings, cols, hp, order, kw, seed, sigma, exact_cap


def run_case(case, rng):
    import os
    rings, cols, hp, order, kw, seed, sigma, exact_cap = draw(rng)
    zeros = kw.pop("_zeros", None)
    clouds = [make_scan(rings, cols, seed=seed + i, sigma=sigma, **kw) for i in range(2)]
    if zeros is not None:
        return run_zeros_case(case, rng, rings, cols, hp, clouds, zeros, seed, exact_cap)
    # a third of the contexts that know the sensor look for rotated / reversed rings from the first batch on
    pin = bool(rng.integers(0, 3) == 0)
    if pin:
        os.environ["LFX_DEBUG_XFORM"] = "1"
    try:
        f = FeatureExtraction(hp, device=0, max_points_per_scan=rings * cols, max_batch=2,
                              max_points_per_ring=cols if exact_cap else 0, max_rings=rings if exact_cap else 0)
    finally:
        os.environ.pop("LFX_DEBUG_XFORM", None)
    op = OB.Params(hp.padding, hp.neighbor_degree_threshold, hp.distance_diff_threshold, hp.parallel_beam_min_range_ratio,
                   hp.edge_threshold, hp.surface_threshold, hp.min_range, hp.max_range, hp.n_blocks)
    want = []
    for c in clouds:
        w = OB.extract(c, op, canonical_ties=False)
        if w["angle_ties"] or w["curvature_ties"]:
            w = OB.extract(c, op, canonical_ties=True)
        want.append(w)
    try:
        for rep in range(2):                       # the second call may take another route (order pre-pass, path choice)
            got = f.extract_batch(clouds)
            for i in range(len(clouds)):
                assert_scan_equal(got[i], want[i], "case %d: %dx%d P%d B%d %s rep%d scan%d seed%d" % (
                    case, rings, cols, hp.padding, hp.n_blocks, order, rep, i, seed))
    finally:
        f.close()
    return order


def run_zeros_case(case, rng, rings, cols, hp, clouds, zeros, seed, exact_cap):
    """A grid with (0, 0, 0) records against the oracle on the filtered cloud (assert_filtered_equal)."""
    fraction, gap, hinted = zeros
    masks = []
    for i, c in enumerate(clouds):
        z = rng.uniform(0.0, 1.0, len(c)) < fraction
        if gap:
            lo = int(rng.integers(0, cols))
            width = int(rng.integers(1, max(2, cols // 3)))
            z |= (np.arange(len(c)) // rings >= lo) & (np.arange(len(c)) // rings < lo + width) & (rng.uniform(0.0, 1.0, len(c)) < 0.9)
        for f in ("x", "y", "z"):
            c[f][z] = 0.0
        masks.append(z)
    import os
    by_scan = bool(rng.integers(0, 2))             # the count pass of large batches (one workgroup per scan), pinned on for these two scans
    if by_scan:
        os.environ["LFX_DEBUG_SCAN_COUNT_FROM"] = "1"
    try:
        f = FeatureExtraction(hp, device=0, max_points_per_scan=rings * cols, max_batch=2, max_points_per_ring=cols if exact_cap else 0,
                              max_rings=rings, drop_zero_points=True, stream_hint=LB.STREAM_GRID_WITH_HOLES if hinted else 0)
    finally:
        os.environ.pop("LFX_DEBUG_SCAN_COUNT_FROM", None)
    op = OB.Params(hp.padding, hp.neighbor_degree_threshold, hp.distance_diff_threshold, hp.parallel_beam_min_range_ratio,
                   hp.edge_threshold, hp.surface_threshold, hp.min_range, hp.max_range, hp.n_blocks)
    keeps = [np.nonzero(~z)[0] for z in masks]
    want = []
    for c, keep in zip(clouds, keeps):
        w = OB.extract(np.ascontiguousarray(c[keep]), op, canonical_ties=False)
        if w["angle_ties"] or w["curvature_ties"]:
            w = OB.extract(np.ascontiguousarray(c[keep]), op, canonical_ties=True)
        want.append(w)
    try:
        for rep in range(3):                       # (the route follows the report: plain form, holes form, the bucketing route)
            got = f.extract_batch(clouds)
            for i in range(len(clouds)):
                assert_filtered_equal(got[i], want[i], keeps[i], masks[i], "case %d: %dx%d P%d B%d zeros %.2f%s%s rep%d scan%d seed%d[ties]" % (
                    case, rings, cols, hp.padding, hp.n_blocks, fraction, " gap" if gap else "", " hinted" if hinted else "", rep, i, seed))
    finally:
        f.close()
    return "zeros"


def run_cases(n_cases, seed=7, report_every=0):
    rng = np.random.default_rng(seed)
    t0 = time.time()
    seen = {}
    for case in range(n_cases):
        o = run_case(case, rng)
        seen[o] = seen.get(o, 0) + 1
        if report_every and case % report_every == report_every - 1:
            print("%d cases ok, %.0f s" % (case + 1, time.time() - t0), flush=True)
    return seen
