"""Randomised parity cases shared by tests/test_gpu_parity.py (a seeded slice under pytest -m gpu) and
tools/stress.py (the long sweep): random sensor shapes, input orders, sensor noise and all nine
hyper-parameters; every output of the HIP path against the CPU oracle (tests/parity.assert_scan_equal)."""
import time

import numpy as np

from lidar_feature_extraction_amd import FeatureExtraction, HyperParameters, make_scan
from oracle import binding as OB
from tests.parity import assert_scan_equal

ORDERS = ["sorted", "rotated", "reversed", "revrot", "shuffled", "ragged", "ragrot"]


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
    seed = int(rng.integers(1, 1 << 30))
    sigma = float(rng.choice([0.01, 0.002, 0.03]))
    exact_cap = bool(rng.integers(0, 2))
    return rings, cols, hp, order, kw, seed, sigma, exact_cap


def run_case(case, rng):
    import os
    rings, cols, hp, order, kw, seed, sigma, exact_cap = draw(rng)
    clouds = [make_scan(rings, cols, seed=seed + i, sigma=sigma, **kw) for i in range(2)]
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
